// pw_x3 — the pointwise (1x1x1, stride 1) layers of the SlowFast encoder in the contract-grade split-plane arithmetic
// (see conv_x3.hip for the number format):  y = act( W x + b [+ r] )  on (hi, lo) plane pairs.
//
// Why a second x3 kernel: these layers — the bottlenecks' reducing / expanding convs (+ residual), 42 % of the x3
// encoder's time on the general tile — are HBM-bound (bytes/row = 4 (K + N [+ N]), two planes each), and the general
// tile serves them phase by phase (operands -> LDS -> MFMA -> fp32 staging -> residual -> stores, two barriers per tile,
// 2 workgroups per CU) at 2.5-3.7 TB/s.  Here, as in the bf16 path's pw_chain.hip, a WAVE owns 16 positions from load to
// store and nothing is staged: the input fragments (lane = position l & 15, k-group l >> 4: 16 contiguous bytes of a row)
// are the MFMA B operand as loaded; the weights are the first operand, resident in LDS as MFMA fragments for the whole
// launch (persistent workgroups), with their output rows permuted in the packing so that a lane ends up holding 8
// CONSECUTIVE channels of its position — one 16-byte store per plane, one 16-byte load per plane of the residual; no
// barrier in the position loop, 8 waves per CU keep ~20 KB of loads in flight each.
// A workgroup holds the fragments of NT1 x 16 output channels (both planes: NT1 x K1S x 2 KB <= 128 KB); wider layers
// are split into channel chunks handled by workgroups that are adjacent in the grid AND on one XCD (block b and b + 8
// share an XCD), so the chunks' re-reads of the same input rows meet in that XCD's L2.
// v_mfma_f32_16x16x32_{f16,bf16}: three passes per product, wl*xh + wh*xl + wh*xh, fp32 accumulate.
#include <stdlib.h>

#include <type_traits>

#include "avt_common.h"
#include "split_planes.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct PxArgs {
  const uint16_t* xh;  // [M, ldx] input planes
  const uint16_t* xl;
  const uint16_t* rh;  // [M, ldr] residual planes or NULL
  const uint16_t* rl;
  uint16_t* yh;        // [M, ldy]
  uint16_t* yl;
  const i32x4* wh;     // [N/16][K1S][64 lanes] fragments (fused_slowfast.pack_pw_planes)
  const i32x4* wl;
  const float* bias;   // [N]
  const float* wscale; // [N] or NULL
  int M, ldx, ldr, ldy, k1c, ntiles, relu;
  int n_chunks, n_rg;  // channel chunks per row group, row groups (grid = 8-aligned n_rg * n_chunks)
  // temporal-tap form (avt_lateral_x3, round 4): the layer is Conv3d [taps,1,1] with temporal stride tst and padding tpad — the
  // lateral fast -> slow connections ([7,1,1] stride 4).  Output row p = ((b * To + to) * HW + pos); its K axis is tap-major
  // (k = dt * Cin + c, the convolution's own weight order), so 8-channel chunk ch of the operand is channels 8 (ch % cin8) .. + 7
  // of INPUT row ((b * T + to * tst - tpad + ch / cin8) * HW + pos): the same streaming kernel with a gathered operand — no im2col
  // pass, no tap table; frames outside [0, T) and chunks past the last tap are zero operands.  taps = 1: plain pointwise
  int taps, cin8, T, To, HW, tst, tpad;
  int n_valid;  // output channels that exist (the last 16-channel tile pair may be padding: cout 16 runs as one pair)
};

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(i32x4 w, i32x4 x, f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}

constexpr int PX_NW = 8;
// phase-skip diagnostic (tools/probe_pw_phases.sh builds one library per -DAVT_PW_DBG_CONST=n: 1 no stores, 2 no MFMAs,
// 4 no residual loads); the shipped library is built with 0
#ifndef AVT_PW_DBG_CONST
#define AVT_PW_DBG_CONST 0
#endif
#define PW_SKIP(bit) (((AVT_PW_DBG_CONST) & (bit)) != 0)
#ifndef PW_IL
#define PW_IL 2  // accumulators interleaved in the MFMA loop (NT1 is even)
#endif

template <int K1S, int NT1, bool F16>
__global__ __launch_bounds__(PX_NW * 64) void pw_x3_kernel(PxArgs a) {
  constexpr int NC = NT1 * 16;  // output channels of this workgroup's chunk
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* whl = lds;                                   // [NT1][K1S] fragments of 1 KB
  char* wll = whl + NT1 * K1S * 1024;
  float* bl = reinterpret_cast<float*>(wll + NT1 * K1S * 1024);  // [NC] bias, [NC] scale
  float* sl = bl + NC;

  // block -> (row group, channel chunk): the chunks of one row group are consecutive blocks of ONE XCD
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int chunk = j % a.n_chunks;
  const int rg = (j / a.n_chunks) * 8 + xcd;
  if (rg >= a.n_rg) return;
  const int c0 = chunk * NC;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int f0 = chunk * NT1 * K1S;  // first fragment of the chunk
  for (int f = wid; f < NT1 * K1S; f += PX_NW) {
    *reinterpret_cast<i32x4*>(whl + f * 1024 + lane * 16) = a.wh[(f0 + f) * 64 + lane];
    *reinterpret_cast<i32x4*>(wll + f * 1024 + lane * 16) = a.wl[(f0 + f) * 64 + lane];
  }
  for (int i = tid; i < NC; i += PX_NW * 64) {
    bl[i] = a.bias ? a.bias[c0 + i] : 0.0f;
    sl[i] = a.wscale ? a.wscale[c0 + i] : 1.0f;
  }
  __syncthreads();
  const bool has_res = a.rh != nullptr;

  for (int tile = rg * PX_NW + wid; tile < a.ntiles; tile += a.n_rg * PX_NW) {
    const int p = tile * 16 + l15;
    const bool ok = p < a.M;
    int lofs = lane * 16;  // opaque per tile: keeps the loop-invariant fragment reads from being hoisted into registers
    asm volatile("" : "+v"(lofs));
    const int64_t pc = ok ? p : a.M - 1;
    i32x4 xh[K1S], xl[K1S];
    if (a.taps > 1) {  // uniform: the temporal-tap form
      const int bto = (int)pc / a.HW, pos = (int)pc - bto * a.HW;
      const int b = bto / a.To, to = bto - b * a.To;
#pragma unroll
      for (int ks = 0; ks < K1S; ++ks) {
        const int ch = 4 * ks + q;
        const int dt = ch / a.cin8, cc = ch - dt * a.cin8;
        const int t_in = to * a.tst - a.tpad + dt;
        const bool valid = dt < a.taps && (unsigned)t_in < (unsigned)a.T;
        const int64_t o = (((int64_t)b * a.T + (valid ? t_in : 0)) * a.HW + pos) * a.ldx + cc * 8;
        const i32x4 vh = *reinterpret_cast<const i32x4*>(a.xh + o), vl = *reinterpret_cast<const i32x4*>(a.xl + o);
        xh[ks] = valid ? vh : i32x4{0, 0, 0, 0};
        xl[ks] = valid ? vl : i32x4{0, 0, 0, 0};
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < K1S; ++ks) {
      int ch = 4 * ks + q;  // past the row's end the weights are zero: any finite value will do
      ch = ch < a.k1c ? ch : a.k1c - 1;
      xh[ks] = *reinterpret_cast<const i32x4*>(a.xh + pc * a.ldx + ch * 8);
      xl[ks] = *reinterpret_cast<const i32x4*>(a.xl + pc * a.ldx + ch * 8);
    }
    }
    uint4 rfh[NT1 / 2], rfl[NT1 / 2];
    if (has_res && !PW_SKIP(4)) {
#pragma unroll
      for (int jj = 0; jj < NT1 / 2; ++jj) {
        const int64_t o = pc * a.ldr + c0 + 32 * jj + 8 * q;
        rfh[jj] = *reinterpret_cast<const uint4*>(a.rh + o);
        rfl[jj] = *reinterpret_cast<const uint4*>(a.rl + o);
      }
    }
    f32x4 acc[NT1];
#pragma unroll
    for (int n = 0; n < NT1; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < K1S; ++ks) {
#pragma unroll
      for (int n = 0; n < NT1; n += PW_IL) {
        // PW_IL accumulators in turn (round 4): every accumulator still sees its own three terms in the order wl*xh, wh*xl, wh*xh
        // (bit-identical results), but no MFMA reads the result of the one issued just before it: -4.5 % on 256 -> 1024 + residual
        i32x4 wh[PW_IL], wl[PW_IL];
#pragma unroll
        for (int u = 0; u < PW_IL; ++u) {
          const int f = (n + u) * K1S + ks;
          wh[u] = *reinterpret_cast<const i32x4*>(whl + f * 1024 + lofs);
          wl[u] = *reinterpret_cast<const i32x4*>(wll + f * 1024 + lofs);
        }
        if (!PW_SKIP(2)) {
#pragma unroll
          for (int u = 0; u < PW_IL; ++u) acc[n + u] = mfma16<F16>(wl[u], xh[ks], acc[n + u]);
#pragma unroll
          for (int u = 0; u < PW_IL; ++u) acc[n + u] = mfma16<F16>(wh[u], xl[ks], acc[n + u]);
#pragma unroll
          for (int u = 0; u < PW_IL; ++u) acc[n + u] = mfma16<F16>(wh[u], xh[ks], acc[n + u]);
        } else {
#pragma unroll
          for (int u = 0; u < PW_IL; ++u)
            acc[n + u][0] += __builtin_bit_cast(float, wl[u][0] ^ xh[ks][0] ^ wh[u][1] ^ xl[ks][1]);  // keep the operands alive
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < NT1 / 2; ++jj) {  // tiles 2jj, 2jj+1 -> channels c0 + 32 jj + 8 q .. + 7
      const int cl = 32 * jj + 8 * q;
      const float4 ba = *reinterpret_cast<const float4*>(bl + cl), bb = *reinterpret_cast<const float4*>(bl + cl + 4);
      const float4 sa = *reinterpret_cast<const float4*>(sl + cl), sb = *reinterpret_cast<const float4*>(sl + cl + 4);
      float v[8] = {acc[2 * jj][0] * sa.x + ba.x,     acc[2 * jj][1] * sa.y + ba.y,
                    acc[2 * jj][2] * sa.z + ba.z,     acc[2 * jj][3] * sa.w + ba.w,
                    acc[2 * jj + 1][0] * sb.x + bb.x, acc[2 * jj + 1][1] * sb.y + bb.y,
                    acc[2 * jj + 1][2] * sb.z + bb.z, acc[2 * jj + 1][3] * sb.w + bb.w};
      if (has_res) {
        const uint32_t* ph = reinterpret_cast<const uint32_t*>(&rfh[jj]);
        const uint32_t* pl = reinterpret_cast<const uint32_t*>(&rfl[jj]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const avt::f32x2 r = avt::join2<F16>(ph[e], pl[e]);
          v[2 * e] += r.x;
          v[2 * e + 1] += r.y;
        }
      }
      if (a.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = avt::relu_keep_nan(v[i]);
      }
      uint4 oh, ol;
      avt::split2<F16>(v[0], v[1], oh.x, ol.x);
      avt::split2<F16>(v[2], v[3], oh.y, ol.y);
      avt::split2<F16>(v[4], v[5], oh.z, ol.z);
      avt::split2<F16>(v[6], v[7], oh.w, ol.w);
      if (ok && c0 + cl < a.n_valid && (!PW_SKIP(1) || (oh.x ^ ol.x) == 0x12345678u)) {
        const int64_t o = pc * a.ldy + c0 + cl;
        *reinterpret_cast<uint4*>(a.yh + o) = oh;
        *reinterpret_cast<uint4*>(a.yl + o) = ol;
      }
    }
  }
}

// ---- chained form: y = act(W1 x + b1 [+ r]) AND z = relu(W2 y + b2) in one pass over the rows ---------------------------------
// The slow pathway's  c of block i (+ residual + ReLU)  ->  a of block i + 1 (+ ReLU)  (the x3 counterpart of pw_chain.hip):
// the first GEMM's D layout — a lane ends with 8 consecutive channels of its position, already split into the two planes it
// stores — IS the second GEMM's B-operand layout (k-group q of k-step jj = channels 32 jj + 8 q .. + 7), so y goes from the
// epilogue's registers into the next MFMAs without touching LDS or HBM again: the a layer's read of y (256 channels x 2 planes
// per row, a third of the pair's bytes) disappears.  z is computed from the SAME rounded planes the unchained a layer would
// read, in the same k order: bit-identical to the two launches.  One chunk holds all N1 channels (res2: 64 -> 256 -> 64:
// 64 + 64 KB of fragments, one 8-wave workgroup per CU).
struct PcArgs {
  PxArgs p;            // first layer (wh / wl / bias / wscale / relu of layer 1; n_chunks == 1)
  const i32x4* w2h;    // [N2/16][K2S][64 lanes] fragments of layer 2 over K2 = N1
  const i32x4* w2l;
  const float* bias2;
  const float* wscale2;
  uint16_t* zh;        // [M, ldz]
  uint16_t* zl;
  int ldz;
};

template <int K1S, int NT1, int NT2, bool F16>
__global__ __launch_bounds__(PX_NW * 64) void pw_chain_x3_kernel(PcArgs c) {
  const PxArgs& a = c.p;
  constexpr int NC = NT1 * 16, K2S = NT1 / 2, NC2 = NT2 * 16;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* whl = lds;
  char* wll = whl + NT1 * K1S * 1024;
  char* w2hl = wll + NT1 * K1S * 1024;
  char* w2ll = w2hl + NT2 * K2S * 1024;
  float* bl = reinterpret_cast<float*>(w2ll + NT2 * K2S * 1024);  // [NC] bias, [NC] scale, [NC2] bias2, [NC2] scale2
  float* sl = bl + NC;
  float* b2l = sl + NC;
  float* s2l = b2l + NC2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  for (int f = wid; f < NT1 * K1S; f += PX_NW) {
    *reinterpret_cast<i32x4*>(whl + f * 1024 + lane * 16) = a.wh[f * 64 + lane];
    *reinterpret_cast<i32x4*>(wll + f * 1024 + lane * 16) = a.wl[f * 64 + lane];
  }
  for (int f = wid; f < NT2 * K2S; f += PX_NW) {
    *reinterpret_cast<i32x4*>(w2hl + f * 1024 + lane * 16) = c.w2h[f * 64 + lane];
    *reinterpret_cast<i32x4*>(w2ll + f * 1024 + lane * 16) = c.w2l[f * 64 + lane];
  }
  for (int i = tid; i < NC; i += PX_NW * 64) {
    bl[i] = a.bias ? a.bias[i] : 0.0f;
    sl[i] = a.wscale ? a.wscale[i] : 1.0f;
  }
  for (int i = tid; i < NC2; i += PX_NW * 64) {
    b2l[i] = c.bias2 ? c.bias2[i] : 0.0f;
    s2l[i] = c.wscale2 ? c.wscale2[i] : 1.0f;
  }
  __syncthreads();
  const bool has_res = a.rh != nullptr;

  for (int tile = blockIdx.x * PX_NW + wid; tile < a.ntiles; tile += gridDim.x * PX_NW) {
    const int p = tile * 16 + l15;
    const bool ok = p < a.M;
    int lofs = lane * 16;
    asm volatile("" : "+v"(lofs));
    const int64_t pc = ok ? p : a.M - 1;
    i32x4 xh[K1S], xl[K1S];
#pragma unroll
    for (int ks = 0; ks < K1S; ++ks) {
      int ch = 4 * ks + q;
      ch = ch < a.k1c ? ch : a.k1c - 1;
      xh[ks] = *reinterpret_cast<const i32x4*>(a.xh + pc * a.ldx + ch * 8);
      xl[ks] = *reinterpret_cast<const i32x4*>(a.xl + pc * a.ldx + ch * 8);
    }
    uint4 rfh[NT1 / 2], rfl[NT1 / 2];
    if (has_res) {
#pragma unroll
      for (int jj = 0; jj < NT1 / 2; ++jj) {
        const int64_t o = pc * a.ldr + 32 * jj + 8 * q;
        rfh[jj] = *reinterpret_cast<const uint4*>(a.rh + o);
        rfl[jj] = *reinterpret_cast<const uint4*>(a.rl + o);
      }
    }
    f32x4 acc2[NT2];
#pragma unroll
    for (int n = 0; n < NT2; ++n) acc2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < NT1 / 2; ++jj) {  // tiles 2jj, 2jj+1 of layer 1 -> channels 32 jj + 8 q .. + 7 = k-step jj of layer 2
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < K1S; ++ks) {
        const int f0 = (2 * jj) * K1S + ks, f1 = (2 * jj + 1) * K1S + ks;
        const i32x4 w0h = *reinterpret_cast<const i32x4*>(whl + f0 * 1024 + lofs), w0l = *reinterpret_cast<const i32x4*>(wll + f0 * 1024 + lofs);
        const i32x4 w1h = *reinterpret_cast<const i32x4*>(whl + f1 * 1024 + lofs), w1l = *reinterpret_cast<const i32x4*>(wll + f1 * 1024 + lofs);
        a0 = mfma16<F16>(w0l, xh[ks], a0);
        a0 = mfma16<F16>(w0h, xl[ks], a0);
        a0 = mfma16<F16>(w0h, xh[ks], a0);
        a1 = mfma16<F16>(w1l, xh[ks], a1);
        a1 = mfma16<F16>(w1h, xl[ks], a1);
        a1 = mfma16<F16>(w1h, xh[ks], a1);
      }
      const int cl = 32 * jj + 8 * q;
      const float4 ba = *reinterpret_cast<const float4*>(bl + cl), bb = *reinterpret_cast<const float4*>(bl + cl + 4);
      const float4 sa = *reinterpret_cast<const float4*>(sl + cl), sb = *reinterpret_cast<const float4*>(sl + cl + 4);
      float v[8] = {a0[0] * sa.x + ba.x, a0[1] * sa.y + ba.y, a0[2] * sa.z + ba.z, a0[3] * sa.w + ba.w,
                    a1[0] * sb.x + bb.x, a1[1] * sb.y + bb.y, a1[2] * sb.z + bb.z, a1[3] * sb.w + bb.w};
      if (has_res) {
        const uint32_t* ph = reinterpret_cast<const uint32_t*>(&rfh[jj]);
        const uint32_t* pl = reinterpret_cast<const uint32_t*>(&rfl[jj]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const avt::f32x2 r = avt::join2<F16>(ph[e], pl[e]);
          v[2 * e] += r.x;
          v[2 * e + 1] += r.y;
        }
      }
      if (a.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = avt::relu_keep_nan(v[i]);
      }
      uint4 oh, ol;
      avt::split2<F16>(v[0], v[1], oh.x, ol.x);
      avt::split2<F16>(v[2], v[3], oh.y, ol.y);
      avt::split2<F16>(v[4], v[5], oh.z, ol.z);
      avt::split2<F16>(v[6], v[7], oh.w, ol.w);
      if (ok) {
        const int64_t o = pc * a.ldy + cl;
        *reinterpret_cast<uint4*>(a.yh + o) = oh;
        *reinterpret_cast<uint4*>(a.yl + o) = ol;
      }
      // layer 2, k-step jj: the planes just stored are the operand
      const i32x4 yh = __builtin_bit_cast(i32x4, oh), yl = __builtin_bit_cast(i32x4, ol);
#pragma unroll
      for (int n = 0; n < NT2; ++n) {
        const int f = n * K2S + jj;
        const i32x4 wh = *reinterpret_cast<const i32x4*>(w2hl + f * 1024 + lofs), wl = *reinterpret_cast<const i32x4*>(w2ll + f * 1024 + lofs);
        acc2[n] = mfma16<F16>(wl, yh, acc2[n]);
        acc2[n] = mfma16<F16>(wh, yl, acc2[n]);
        acc2[n] = mfma16<F16>(wh, yh, acc2[n]);
      }
    }
#pragma unroll
    for (int jj = 0; jj < NT2 / 2; ++jj) {
      const int cl = 32 * jj + 8 * q;
      const float4 ba = *reinterpret_cast<const float4*>(b2l + cl), bb = *reinterpret_cast<const float4*>(b2l + cl + 4);
      const float4 sa = *reinterpret_cast<const float4*>(s2l + cl), sb = *reinterpret_cast<const float4*>(s2l + cl + 4);
      float v[8] = {acc2[2 * jj][0] * sa.x + ba.x,     acc2[2 * jj][1] * sa.y + ba.y,
                    acc2[2 * jj][2] * sa.z + ba.z,     acc2[2 * jj][3] * sa.w + ba.w,
                    acc2[2 * jj + 1][0] * sb.x + bb.x, acc2[2 * jj + 1][1] * sb.y + bb.y,
                    acc2[2 * jj + 1][2] * sb.z + bb.z, acc2[2 * jj + 1][3] * sb.w + bb.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = avt::relu_keep_nan(v[i]);
      uint4 oh, ol;
      avt::split2<F16>(v[0], v[1], oh.x, ol.x);
      avt::split2<F16>(v[2], v[3], oh.y, ol.y);
      avt::split2<F16>(v[4], v[5], oh.z, ol.z);
      avt::split2<F16>(v[6], v[7], oh.w, ol.w);
      if (ok) {
        const int64_t o = pc * c.ldz + cl;
        *reinterpret_cast<uint4*>(c.zh + o) = oh;
        *reinterpret_cast<uint4*>(c.zl + o) = ol;
      }
    }
  }
}

// ---- the TRAINING form: fp32 rows in, fp32 rows out (train_ops.conv3d's pointwise layers: forward and stride-1 input gradient) -----
// The expanding 1x1x1 convolutions of the bottlenecks (64 -> 256, 128 -> 512, 256 -> 1024 and the fast pathway's 8 -> 32 ... 64 -> 256)
// and the input gradients of the reducing ones are the widest-output layers of the step, and K is one to four 64-wide steps: on
// the 128 x 128 tile a workgroup's prologue and epilogue were 26 + 29 % of its cycles and the layer ran at 1.1 TB/s (128 -> 512 at
// 28^2: 1.86 ms per 128 clips, profiles/r04/train_pw_f32_ab.log).  Same streaming structure as pw_x3_kernel: a wave owns 16 positions
// from load to store; its fp32 operand chunks (8 consecutive channels = two 16-byte loads) are split into the two planes in registers
// and are the MFMA B operand; the weight planes arrive in their plain [N][K] layout (they change every optimizer step: no host
// packing) and the prologue of the persistent workgroup lays them out as fragments in LDS; the epilogue scales (fp16 planes' row
// scales), adds the optional fp32 `add` rows and stores 8 consecutive channels per lane as two non-temporal 16-byte stores.
struct PfArgs {
  const float* x;        // [M, ldx]
  const float* add;      // [M, lda] or NULL
  float* y;              // [M, ldy]
  const uint16_t* wh;    // [N][K] plain planes
  const uint16_t* wl;
  const float* wscale;   // [N] or NULL
  int M, ldx, lda, ldy, K, k1c, ntiles, n_chunks, n_rg;
  // STATS (round 5; avt_pw_x3_f32_stats): the output feeds a train-mode BatchNorm — blockIdx.y walks its replica groups (slabs of
  // M rows each: `M` / `ntiles` are PER GROUP then), and every wave leaves the per-channel sum / sum of squares of the rows it
  // stored as one row of partials in bn_train.hip's channel_sums layout (row (group * n_rg + rg) * 8 + wave)
  double* stat_part;
  int n_total;           // channels of the whole layer (the BatchNorm's C)
  // STATS = 2 (avt_pw_x3_f32_bwdstats): the launch is an input gradient whose result is the OUTPUT gradient of a train-mode
  // BatchNorm (+ ReLU) — see conv_args.h (bst_*): the BatchNorm's input rows [M, ldy], its saved statistics [groups][N], scale,
  // shift (mask recomputed) or saved mask bits; the result is stored MASKED and the partial rows hold the sums of g and g * xhat
  const float* bst_x;
  const float* bst_mean;
  const float* bst_invstd;
  const float* bst_gamma;
  const float* bst_beta;
  const uint8_t* bst_mask;
  int bst_relu;
};

// pass width (channels) of the STATS transpose: the per-wave LDS scratch must fit beside the weight fragments
template <int K1S, int NT1>
struct PfStat {
  static constexpr int NC = NT1 * 16;
  static constexpr int PW = NC < 64 ? NC : ((2 * NT1 * K1S * 1024 >= 96 * 1024) ? 32 : 64);
  static constexpr int RS = PW + 4;                      // floats per scratch row (padded)
  static constexpr int BYTES = PX_NW * 16 * RS * 4;      // all waves
};

template <int K1S, int NT1, bool F16, int STATS = 0>
__global__ __launch_bounds__(PX_NW * 64) void pw_x3_f32_kernel(PfArgs a) {
  constexpr int NC = NT1 * 16;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* whl = lds;
  char* wll = whl + NT1 * K1S * 1024;
  float* sl = reinterpret_cast<float*>(wll + NT1 * K1S * 1024);  // [NC] scale

  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int chunk = j % a.n_chunks;
  const int rg = (j / a.n_chunks) * 8 + xcd;
  if (rg >= a.n_rg) return;
  const int c0 = chunk * NC;
  if constexpr (STATS != 0) {  // this workgroup's group: its slab of the rows
    const int64_t g = blockIdx.y;
    a.x += g * a.M * a.ldx;
    a.y += g * a.M * a.ldy;
    if constexpr (STATS == 2) {
      if (a.add) a.add += g * a.M * a.lda;
      a.bst_x += g * a.M * a.ldy;
      if (a.bst_mask) a.bst_mask += g * a.M * a.ldy / 4;
    }
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  // fragments from the plain planes (fused_slowfast.pack_pw_planes' order): fragment (nt, ks), lane (n = l15, q): 8 consecutive k of
  // output row 32 (nt / 2) + 8 (n >> 2) + 4 (nt % 2) + (n & 3), k = 32 ks + 8 q ..; k past the row's end = zeros
  for (int f = wid; f < NT1 * K1S; f += PX_NW) {
    const int nt = f / K1S, ks = f - nt * K1S;
    const int row = c0 + 32 * (nt >> 1) + 8 * (l15 >> 2) + 4 * (nt & 1) + (l15 & 3);
    const int col = 32 * ks + 8 * q;
    i32x4 vh = {0, 0, 0, 0}, vl = {0, 0, 0, 0};
    if (col < a.K) {
      vh = *reinterpret_cast<const i32x4*>(a.wh + (int64_t)row * a.K + col);
      vl = *reinterpret_cast<const i32x4*>(a.wl + (int64_t)row * a.K + col);
    }
    *reinterpret_cast<i32x4*>(whl + f * 1024 + lane * 16) = vh;
    *reinterpret_cast<i32x4*>(wll + f * 1024 + lane * 16) = vl;
  }
  for (int i = tid; i < NC; i += PX_NW * 64) sl[i] = a.wscale ? a.wscale[c0 + i] : 1.0f;
  // STATS = 2: this (group, chunk)'s BatchNorm coefficients [mean | invstd * gamma | beta] behind the statistics scratch
  float* kc = reinterpret_cast<float*>(reinterpret_cast<char*>(sl) + NC * 4 + PfStat<K1S, NT1>::BYTES);
  if constexpr (STATS == 2) {
    for (int i = tid; i < NC; i += PX_NW * 64) {
      const size_t gc = (size_t)blockIdx.y * a.n_total + c0 + i;
      kc[i] = a.bst_mean[gc];
      kc[NC + i] = a.bst_invstd[gc] * a.bst_gamma[c0 + i];  // the forward's scale: the same product
      kc[2 * NC + i] = a.bst_beta ? a.bst_beta[c0 + i] : 0.0f;
    }
  }
  __syncthreads();
  const bool has_add = STATS != 1 && a.add != nullptr;
  // STATS: per-wave scratch [16 rows][PW channels] behind the scale floats; lane l owns column l of every pass (PW = 32: column
  // l & 31, rows 8 (l >> 5) ..), its sums over the wave's tiles in fp64 (fp32 over the 16 / 8 rows of one tile)
  typedef PfStat<K1S, NT1> PS;
  constexpr int NPASS = NC / PS::PW;
  float* scr = reinterpret_cast<float*>(reinterpret_cast<char*>(sl) + NC * 4) + wid * 16 * PS::RS;
  double st_s[STATS != 0 ? NPASS : 1], st_q[STATS != 0 ? NPASS : 1];
  if constexpr (STATS != 0) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) st_s[ps] = st_q[ps] = 0.0;
  }

  for (int tile = rg * PX_NW + wid; tile < a.ntiles; tile += a.n_rg * PX_NW) {
    const int p = tile * 16 + l15;
    const bool ok = p < a.M;
    int lofs = lane * 16;  // opaque per tile: keeps the loop-invariant fragment reads from being hoisted into registers
    asm volatile("" : "+v"(lofs));
    int kz = 0;            // ... and the STATS = 2 coefficient reads
    asm volatile("" : "+v"(kz));
    const int64_t pc = ok ? p : a.M - 1;
    float4 xa[K1S], xb[K1S];
#pragma unroll
    for (int ks = 0; ks < K1S; ++ks) {
      int ch = 4 * ks + q;  // past the row's end the weights are zero: any finite value will do
      ch = ch < a.k1c ? ch : a.k1c - 1;
      const float* px = a.x + pc * a.ldx + ch * 8;
      xa[ks] = *reinterpret_cast<const float4*>(px);
      xb[ks] = *reinterpret_cast<const float4*>(px + 4);
    }
    // the epilogue's operands (`add` rows, STATS = 2: the BatchNorm input's octets + mask nibbles), requested ahead of their use.
    // STATS = 2 with 16 tiles per wave: in TWO batches — the second after the MFMAs, when the operand registers are free (all at
    // once, the kernel spilled 40-115 registers)
    constexpr int HB = (STATS == 2 && NT1 > 8) ? NT1 / 4 : NT1 / 2;
    float4 ra[NT1 / 2], rb[NT1 / 2];
    float4 bxa[STATS == 2 ? NT1 / 2 : 1], bxb[STATS == 2 ? NT1 / 2 : 1];
    unsigned bbits[STATS == 2 ? NT1 / 2 : 1];
    auto epi_request = [&](auto j0, auto j1) {  // (compile-time bounds: the arrays stay in registers)
#pragma unroll
      for (int jj = decltype(j0)::value; jj < decltype(j1)::value; ++jj) {
        if (has_add) {
          const float* pr = a.add + pc * a.lda + c0 + 32 * jj + 8 * q;
          ra[jj] = *reinterpret_cast<const float4*>(pr);
          rb[jj] = *reinterpret_cast<const float4*>(pr + 4);
        }
        if constexpr (STATS == 2) {
          const int64_t o = pc * a.ldy + c0 + 32 * jj + 8 * q;
          bxa[jj] = *reinterpret_cast<const float4*>(a.bst_x + o);
          bxb[jj] = *reinterpret_cast<const float4*>(a.bst_x + o + 4);
          const unsigned v = a.bst_mask ? *reinterpret_cast<const uint16_t*>(a.bst_mask + (o >> 2)) : 0u;
          bbits[jj] = (v & 0xFu) | ((v >> 4) & 0xF0u);
        }
      }
    };
    epi_request(std::integral_constant<int, 0>{}, std::integral_constant<int, HB>{});
    i32x4 xh[K1S], xl[K1S];
#pragma unroll
    for (int ks = 0; ks < K1S; ++ks) {
      const float f8[8] = {xa[ks].x, xa[ks].y, xa[ks].z, xa[ks].w, xb[ks].x, xb[ks].y, xb[ks].z, xb[ks].w};
      uint4 h, l;
      avt::split8<F16>(f8, h, l);
      xh[ks] = __builtin_bit_cast(i32x4, h);
      xl[ks] = __builtin_bit_cast(i32x4, l);
    }
    f32x4 acc[NT1];
    auto mfma_part = [&](auto n0_, auto n1_) {  // output tiles n0 .. n1 - 1: every accumulator sees its own terms in the one order
#pragma unroll
      for (int n = decltype(n0_)::value; n < decltype(n1_)::value; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < K1S; ++ks) {
#pragma unroll
        for (int n = decltype(n0_)::value; n < decltype(n1_)::value; n += PW_IL) {
          i32x4 wh[PW_IL], wl[PW_IL];
#pragma unroll
          for (int u = 0; u < PW_IL; ++u) {
            const int f = (n + u) * K1S + ks;
            wh[u] = *reinterpret_cast<const i32x4*>(whl + f * 1024 + lofs);
            wl[u] = *reinterpret_cast<const i32x4*>(wll + f * 1024 + lofs);
          }
#pragma unroll
          for (int u = 0; u < PW_IL; ++u) acc[n + u] = mfma16<F16>(wl[u], xh[ks], acc[n + u]);
#pragma unroll
          for (int u = 0; u < PW_IL; ++u) acc[n + u] = mfma16<F16>(wh[u], xl[ks], acc[n + u]);
#pragma unroll
          for (int u = 0; u < PW_IL; ++u) acc[n + u] = mfma16<F16>(wh[u], xh[ks], acc[n + u]);
        }
      }
    };
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    f32x4n hv[4];  // (STATS = 2: the second statistic's addends of one pass)
    auto epilogue = [&](auto j0_, auto j1_) {
#pragma unroll
    for (int jj = decltype(j0_)::value; jj < decltype(j1_)::value; ++jj) {  // tiles 2jj, 2jj+1 -> channels c0 + 32 jj + 8 q .. + 7
      const int cl = 32 * jj + 8 * q;
      f32x4n o0, o1;
      if constexpr (STATS == 2) {  // (gradients: bf16 planes, no row scales — and no 4 NT1 scale registers held across the tile loop)
        o0 = f32x4n{acc[2 * jj][0], acc[2 * jj][1], acc[2 * jj][2], acc[2 * jj][3]};
        o1 = f32x4n{acc[2 * jj + 1][0], acc[2 * jj + 1][1], acc[2 * jj + 1][2], acc[2 * jj + 1][3]};
      } else {
        const float4 sa = *reinterpret_cast<const float4*>(sl + cl), sb = *reinterpret_cast<const float4*>(sl + cl + 4);
        o0 = f32x4n{acc[2 * jj][0] * sa.x, acc[2 * jj][1] * sa.y, acc[2 * jj][2] * sa.z, acc[2 * jj][3] * sa.w};
        o1 = f32x4n{acc[2 * jj + 1][0] * sb.x, acc[2 * jj + 1][1] * sb.y, acc[2 * jj + 1][2] * sb.z, acc[2 * jj + 1][3] * sb.w};
      }
      if (has_add) {
        o0 += f32x4n{ra[jj].x, ra[jj].y, ra[jj].z, ra[jj].w};
        o1 += f32x4n{rb[jj].x, rb[jj].y, rb[jj].z, rb[jj].w};
      }
      f32x4n v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;  // STATS = 2: g * (x - mean), the second statistic's addends
      if constexpr (STATS == 2) {  // o becomes g = mask * dz (what is stored)
        // (coefficients through an offset that is opaque per tile: read at constant addresses they are loop-invariant, and the
        //  compiler kept all 24 x NT1 / 2 of them in registers across the tile loop — 54-108 spilled registers)
        const float* kct = kc + kz;
        const float4 m0 = *reinterpret_cast<const float4*>(kct + cl), m1 = *reinterpret_cast<const float4*>(kct + cl + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(kct + NC + cl), s1 = *reinterpret_cast<const float4*>(kct + NC + cl + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(kct + 2 * NC + cl), b1 = *reinterpret_cast<const float4*>(kct + 2 * NC + cl + 4);
        const float xc[8] = {bxa[jj].x - m0.x, bxa[jj].y - m0.y, bxa[jj].z - m0.z, bxa[jj].w - m0.w,
                             bxb[jj].x - m1.x, bxb[jj].y - m1.y, bxb[jj].z - m1.z, bxb[jj].w - m1.w};
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float gg[8] = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]}, vv[8];
        // (uniform choices folded into the select's operands: see bst_apply in conv_args.h)
        const bool by_bits = a.bst_relu != 0 && a.bst_mask != nullptr, by_x = a.bst_relu != 0 && a.bst_mask == nullptr;
        const unsigned mbits = by_bits ? bbits[jj] : 0xFFu;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float act = by_x ? xc[e] * sc[e] + be[e] : 1.0f;
          const float sel = (((mbits >> e) & 1u) != 0u && act > 0.0f) ? 1.0f : 0.0f;
          gg[e] = sel != 0.0f ? gg[e] : 0.0f;
          vv[e] = gg[e] * xc[e];
        }
        o0 = f32x4n{gg[0], gg[1], gg[2], gg[3]};
        o1 = f32x4n{gg[4], gg[5], gg[6], gg[7]};
        v0 = f32x4n{vv[0], vv[1], vv[2], vv[3]};
        v1 = f32x4n{vv[4], vv[5], vv[6], vv[7]};
      }
      if (ok) {
        float* po = a.y + pc * a.ldy + c0 + cl;
        __builtin_nontemporal_store(o0, reinterpret_cast<f32x4n*>(po));
        __builtin_nontemporal_store(o1, reinterpret_cast<f32x4n*>(po + 4));
      }
      if constexpr (STATS == 2) {
        // as below, with two blocks per pass: g for the first statistic, g * (x - mean) for the second (held in registers — at
        // most two octets — until the first block has been read back)
        constexpr int JPP = PS::PW / 32;
        const int jl = jj % JPP;
        if (!ok) o0 = o1 = v0 = v1 = f32x4n{0.f, 0.f, 0.f, 0.f};
        float* pw_ = scr + l15 * PS::RS + 32 * jl + 8 * q;
        *reinterpret_cast<f32x4n*>(pw_) = o0;
        *reinterpret_cast<f32x4n*>(pw_ + 4) = o1;
        if (jl == 0) { hv[0] = v0; hv[1] = v1; } else { hv[2] = v0; hv[3] = v1; }
        if (jl == JPP - 1) {
          constexpr int NR = PS::PW == 64 ? 16 : 8;
          const float* pr = scr + (PS::PW == 64 ? lane : (lane & 31) + (lane >> 5) * 8 * PS::RS);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          float ts = 0.f, tq = 0.f;
          if (PS::PW >= 32 || lane < PS::PW) {
#pragma unroll
            for (int r = 0; r < NR; ++r) ts += pr[r * PS::RS];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < JPP; ++k) {
            float* pv = scr + l15 * PS::RS + 32 * k + 8 * q;
            *reinterpret_cast<f32x4n*>(pv) = hv[2 * k];
            *reinterpret_cast<f32x4n*>(pv + 4) = hv[2 * k + 1];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (PS::PW >= 32 || lane < PS::PW) {
#pragma unroll
            for (int r = 0; r < NR; ++r) tq += pr[r * PS::RS];
          }
          st_s[jj / JPP] += (double)ts;
          st_q[jj / JPP] += (double)tq;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      if constexpr (STATS == 1) {
        // the 16 x PW block of this pass goes through the wave's scratch: written as it lies (row = position, 8 consecutive
        // channels per lane; rows past the slab as zeros), read back one COLUMN per lane — LDS operations of one wave complete in
        // order, so a wait on the counter is the only synchronisation; no cross-lane traffic, 8 accumulator registers per pass
        constexpr int JPP = PS::PW / 32;  // jj's per pass
        const int jl = jj % JPP;
        if (!ok) o0 = o1 = f32x4n{0.f, 0.f, 0.f, 0.f};
        float* pw_ = scr + l15 * PS::RS + 32 * jl + 8 * q;
        *reinterpret_cast<f32x4n*>(pw_) = o0;
        *reinterpret_cast<f32x4n*>(pw_ + 4) = o1;
        if (jl == JPP - 1) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          constexpr int NR = PS::PW == 64 ? 16 : 8;
          const float* pr = scr + (PS::PW == 64 ? lane : (lane & 31) + (lane >> 5) * 8 * PS::RS);
          float ts = 0.f, tq = 0.f;
          if (PS::PW >= 32 || lane < PS::PW) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              const float v = pr[r * PS::RS];
              ts += v;
              tq += v * v;
            }
          }
          st_s[jj / JPP] += (double)ts;
          st_q[jj / JPP] += (double)tq;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the reads are done before the next pass overwrites the block)
        }
      }
    }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, HB> IH;
    typedef std::integral_constant<int, NT1 / 2> IE;
    if constexpr (HB < NT1 / 2) {  // (STATS = 2, 16 tiles) two halves: the second half's epilogue operands are requested under its MFMAs
      // (sched_barrier: the scheduler otherwise lifts the second batch's loads and MFMAs over the first half's epilogue, and the
      //  register pressure of the unsplit form is back)
      mfma_part(I0{}, std::integral_constant<int, 2 * HB>{});
      __builtin_amdgcn_sched_barrier(0);
      epilogue(I0{}, IH{});
      __builtin_amdgcn_sched_barrier(0);
      epi_request(IH{}, IE{});
      mfma_part(std::integral_constant<int, 2 * HB>{}, std::integral_constant<int, NT1>{});
      __builtin_amdgcn_sched_barrier(0);
      epilogue(IH{}, IE{});
    } else {
      mfma_part(I0{}, std::integral_constant<int, NT1>{});
      epilogue(I0{}, IE{});
    }
  }
  if constexpr (STATS != 0) {  // this wave's row of partials: channel c0 + PW * pass + column
    const int C = a.n_total, q4 = C / 4, nq = q4 < 256 ? q4 : 256, unit = q4 > 256 ? q4 / 256 : 1;
    const size_t row = ((size_t)blockIdx.y * a.n_rg + rg) * PX_NW + wid;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      double s0 = st_s[ps], s1 = st_q[ps];
      if (PS::PW == 32) {  // the two row halves of a 32-wide pass meet here
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
      }
      if (lane < PS::PW) {
        const int c = c0 + ps * PS::PW + lane, quad = c >> 2, e = c & 3;
        if constexpr (STATS == 2) s1 *= (double)a.bst_invstd[(size_t)blockIdx.y * C + c];  // sum of g * (x - mean) -> of g * xhat
        double* dst = a.stat_part + (row * unit + (quad >> 8)) * nq * 8 + (size_t)(quad % nq) * 8 + e;
        dst[0] = s0;
        dst[4] = s1;
      }
    }
  }
}

// row groups (persistent workgroups along the rows) of one launch: as many workgroups as stay resident, over `groups` slabs
int pf_row_groups(int lds_bytes, int n_chunks, int ntiles, int groups) {
  const int per_cu = lds_bytes > 80 * 1024 ? 1 : (lds_bytes > 52 * 1024 ? 2 : 3);
  int n_rg = (256 * per_cu) / (n_chunks * groups);
  n_rg = n_rg < 8 ? 8 : (n_rg / 8) * 8;
  const int max_rg = (ntiles + PX_NW - 1) / PX_NW;
  if (n_rg > ((max_rg + 7) / 8) * 8) n_rg = ((max_rg + 7) / 8) * 8;
  return n_rg;
}

template <int K1S, int NT1>
constexpr int pf_lds_bytes(int stats) {  // 0 none, 1 forward statistics, 2 backward (+ the coefficient table)
  return 2 * NT1 * K1S * 1024 + NT1 * 16 * 4 + (stats ? PfStat<K1S, NT1>::BYTES : 0) + (stats == 2 ? 3 * NT1 * 16 * 4 : 0);
}

template <int K1S, int NT1, bool F16, int MODE>
int launch_pf_stats(PfArgs& a, hipStream_t st, int groups) {  // blockIdx.y = the BatchNorm's replica group, a.M / a.ntiles per group
  constexpr int lds_bytes = pf_lds_bytes<K1S, NT1>(MODE);
  static_assert(lds_bytes <= 160 * 1024, "weights + the statistics scratch fit the LDS");
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pw_x3_f32_kernel<K1S, NT1, F16, MODE>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("avt_pw_x3_f32_stats: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  a.n_rg = pf_row_groups(lds_bytes, a.n_chunks, a.ntiles, groups);
  hipLaunchKernelGGL((pw_x3_f32_kernel<K1S, NT1, F16, MODE>), dim3((unsigned)(a.n_rg * a.n_chunks), (unsigned)groups), dim3(PX_NW * 64),
                     lds_bytes, st, a);
  return avt::check_launch("avt_pw_x3_f32_stats");
}

template <int K1S, int NT1, bool F16>
int launch_pf(PfArgs& a, hipStream_t st, int groups = 0) {
  if (groups > 0) {
    if (a.bst_x) {
      if constexpr (!F16) return launch_pf_stats<K1S, NT1, false, 2>(a, st, groups);  // (gradients: bf16 planes)
      avt::set_error("avt_pw_x3_f32_bwdstats: bf16 planes only");
      return AVT_ERR_UNSUPPORTED;
    }
    return launch_pf_stats<K1S, NT1, F16, 1>(a, st, groups);
  }
  constexpr int lds_bytes = pf_lds_bytes<K1S, NT1>(0);
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pw_x3_f32_kernel<K1S, NT1, F16>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("avt_pw_x3_f32: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  a.n_rg = pf_row_groups(lds_bytes, a.n_chunks, a.ntiles, 1);
  hipLaunchKernelGGL((pw_x3_f32_kernel<K1S, NT1, F16>), dim3((unsigned)(a.n_rg * a.n_chunks)), dim3(PX_NW * 64), lds_bytes, st, a);
  return avt::check_launch("avt_pw_x3_f32");
}

// LDS bytes of the (K1S, NT1) instance, for the row-group query (avt_pw_x3_f32_stat_rows)
int pf_lds_of(int k1s, int nt1, int stats) {
#define PF_CASE(K, N) if (k1s == K && nt1 == N) return pf_lds_bytes<K, N>(stats);
  PF_CASE(1, 16) PF_CASE(1, 8) PF_CASE(1, 4) PF_CASE(1, 2) PF_CASE(2, 16) PF_CASE(2, 8) PF_CASE(2, 4) PF_CASE(2, 2)
  PF_CASE(4, 16) PF_CASE(4, 8) PF_CASE(4, 4) PF_CASE(4, 2) PF_CASE(8, 8) PF_CASE(8, 4) PF_CASE(8, 2)
#undef PF_CASE
  return 0;
}

template <int K1S, bool F16>
int dispatch_pf_nt(PfArgs& a, int nt1, hipStream_t st, int groups) {
  switch (nt1) {
    case 16: if constexpr (K1S <= 4) return launch_pf<K1S, 16, F16>(a, st, groups); break;
    case 8: if constexpr (K1S <= 8) return launch_pf<K1S, 8, F16>(a, st, groups); break;
    case 4: return launch_pf<K1S, 4, F16>(a, st, groups);
    case 2: return launch_pf<K1S, 2, F16>(a, st, groups);
  }
  avt::set_error("avt_pw_x3_f32: no kernel for K steps %d, tiles %d", K1S, nt1);
  return AVT_ERR_UNSUPPORTED;
}

template <bool F16>
int dispatch_pf_k(PfArgs& a, int k1s, int nt1, hipStream_t st, int groups = 0) {
  switch (k1s) {
    case 1: return dispatch_pf_nt<1, F16>(a, nt1, st, groups);
    case 2: return dispatch_pf_nt<2, F16>(a, nt1, st, groups);
    case 4: return dispatch_pf_nt<4, F16>(a, nt1, st, groups);
    case 8: return dispatch_pf_nt<8, F16>(a, nt1, st, groups);
  }
  avt::set_error("avt_pw_x3_f32: unsupported K (%d steps of 32)", k1s);
  return AVT_ERR_UNSUPPORTED;
}

// tiles per workgroup chunk for (K1S, N): the widest of 16 / 8 / 4 / 2 that divides N/16 and keeps both planes' fragments
// within 128 KB of LDS; 0 = unsupported
int pick_nt1(int k1s, int n) {
  if (n % 32) return 0;
  const int nt = n / 16;
  for (int t : {16, 8, 4, 2})
    if (nt % t == 0 && 2 * t * k1s <= 128) return t;
  return 0;
}

template <int K1S, int NT1, bool F16>
int launch_px(PxArgs& a, hipStream_t st) {
  constexpr int lds_bytes = 2 * NT1 * K1S * 1024 + 2 * NT1 * 16 * 4;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pw_x3_kernel<K1S, NT1, F16>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("avt_pw_x3: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  // persistent workgroups: as many as stay resident (LDS-limited), split into row groups x channel chunks
  const int per_cu = lds_bytes > 80 * 1024 ? 1 : (lds_bytes > 52 * 1024 ? 2 : 3);
  int n_rg = (256 * per_cu) / a.n_chunks;
  n_rg = n_rg < 8 ? 8 : (n_rg / 8) * 8;
  const int max_rg = (a.ntiles + PX_NW - 1) / PX_NW;
  if (n_rg > ((max_rg + 7) / 8) * 8) n_rg = ((max_rg + 7) / 8) * 8;
  a.n_rg = n_rg;  // a multiple of 8: blocks of one XCD (b % 8) form whole row groups
  hipLaunchKernelGGL((pw_x3_kernel<K1S, NT1, F16>), dim3((unsigned)(n_rg * a.n_chunks)), dim3(PX_NW * 64), lds_bytes, st, a);
  return avt::check_launch("avt_pw_x3");
}

template <int K1S, bool F16>
int dispatch_nt(PxArgs& a, int nt1, hipStream_t st) {
  switch (nt1) {
    case 16: if constexpr (K1S <= 4) return launch_px<K1S, 16, F16>(a, st); break;
    case 8: if constexpr (K1S <= 8) return launch_px<K1S, 8, F16>(a, st); break;
    case 4: return launch_px<K1S, 4, F16>(a, st);
    case 2: return launch_px<K1S, 2, F16>(a, st);
  }
  avt::set_error("avt_pw_x3: no kernel for K steps %d, tiles %d", K1S, nt1);
  return AVT_ERR_UNSUPPORTED;
}

template <bool F16>
int dispatch_k(PxArgs& a, int k1s, int nt1, hipStream_t st) {
  switch (k1s) {
    case 1: return dispatch_nt<1, F16>(a, nt1, st);
    case 2: return dispatch_nt<2, F16>(a, nt1, st);
    case 3: return dispatch_nt<3, F16>(a, nt1, st);
    case 4: return dispatch_nt<4, F16>(a, nt1, st);
    case 5: return dispatch_nt<5, F16>(a, nt1, st);
    case 7: return dispatch_nt<7, F16>(a, nt1, st);
    case 8: return dispatch_nt<8, F16>(a, nt1, st);
    case 10: return dispatch_nt<10, F16>(a, nt1, st);
    case 14: return dispatch_nt<14, F16>(a, nt1, st);
    case 16: return dispatch_nt<16, F16>(a, nt1, st);
  }
  avt::set_error("avt_pw_x3: unsupported K (%d steps of 32)", k1s);
  return AVT_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int avt_pw_x3_supported(int k, int n) {
  const int k1s = (k + 31) / 32;
  if (k % 8 || !(k1s == 1 || k1s == 2 || k1s == 3 || k1s == 4 || k1s == 5 || k1s == 8 || k1s == 10 || k1s == 16)) return 0;
  const int nt1 = pick_nt1(k1s, n);
  if (!nt1) return 0;
  if ((nt1 == 16 && k1s > 4) || (nt1 == 8 && k1s > 8)) return 0;
  return n / (16 * nt1) <= 8 ? 1 : 0;  // more than 8 channel chunks re-read the input too often: the general tile is better
}

extern "C" int avt_pw_x3(const void* x_hi, const void* x_lo, int ldx, int k, const void* w_hi, const void* w_lo, const float* bias,
                         const float* wscale, const void* res_hi, const void* res_lo, int ldr, void* y_hi, void* y_lo, int ldy,
                         int n, int64_t m, int relu, int plane_dtype, void* stream) {
  AVT_REQUIRE(x_hi && x_lo && w_hi && w_lo && y_hi && y_lo && (!res_hi == !res_lo), "avt_pw_x3: NULL pointer / half a plane pair");
  AVT_REQUIRE(avt_pw_x3_supported(k, n), "avt_pw_x3: unsupported layer K=%d N=%d", k, n);
  AVT_REQUIRE(m > 0 && m < (1ll << 31) - 16 && ldx >= k && ldy >= n && ldx % 8 == 0 && ldy % 8 == 0 && (!res_hi || (ldr >= n && ldr % 8 == 0)),
              "avt_pw_x3: bad sizes / leading dimensions");
  AVT_REQUIRE(avt::aligned16(x_hi) && avt::aligned16(x_lo) && avt::aligned16(w_hi) && avt::aligned16(w_lo) && avt::aligned16(y_hi) &&
                  avt::aligned16(y_lo) && (!res_hi || (avt::aligned16(res_hi) && avt::aligned16(res_lo))) &&
                  (!bias || avt::aligned16(bias)) && (!wscale || avt::aligned16(wscale)),
              "avt_pw_x3: pointers must be 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_pw_x3: bad plane_dtype");
  const int k1s = (k + 31) / 32, nt1 = pick_nt1(k1s, n);
  PxArgs a;
  a.xh = static_cast<const uint16_t*>(x_hi);
  a.xl = static_cast<const uint16_t*>(x_lo);
  a.rh = static_cast<const uint16_t*>(res_hi);
  a.rl = static_cast<const uint16_t*>(res_lo);
  a.yh = static_cast<uint16_t*>(y_hi);
  a.yl = static_cast<uint16_t*>(y_lo);
  a.wh = static_cast<const i32x4*>(w_hi);
  a.wl = static_cast<const i32x4*>(w_lo);
  a.bias = bias;
  a.wscale = wscale;
  a.M = (int)m;
  a.ldx = ldx;
  a.ldr = ldr;
  a.ldy = ldy;
  a.k1c = k / 8;
  a.ntiles = (int)((m + 15) / 16);
  a.relu = relu;
  a.n_chunks = n / (16 * nt1);
  a.taps = 1;
  a.cin8 = a.T = a.To = a.HW = a.tst = a.tpad = 0;
  a.n_valid = n;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return plane_dtype == AVT_X3_F16 ? dispatch_k<true>(a, k1s, nt1, st) : dispatch_k<false>(a, k1s, nt1, st);
}

// Conv3d [kt,1,1], temporal stride st, padding pt, on plane pairs: the lateral fast -> slow connections (FuseFastToSlow: [7,1,1]
// stride 4 + BN + ReLU of the third-party SlowFast model the reference runs per clip window, models/models.py:335, 399)
extern "C" int avt_lateral_x3_supported(int cin, int cout, int kt) {
  const int k = kt * cin, k1s = (k + 31) / 32, n = (cout + 31) / 32 * 32;
  if (cin % 8 || kt < 2 || !(k1s == 2 || k1s == 7 || k1s == 14)) return 0;
  const int nt1 = pick_nt1(k1s, n);
  return (nt1 && cout % 16 == 0 && n / (16 * nt1) <= 8) ? 1 : 0;
}

extern "C" int avt_lateral_x3(const void* x_hi, const void* x_lo, int ldx, int cin, const void* w_hi, const void* w_lo, const float* bias,
                              const float* wscale, void* y_hi, void* y_lo, int ldy, int cout, int batch, int t, int hw, int kt, int st,
                              int pt, int relu, int plane_dtype, void* stream) {
  AVT_REQUIRE(x_hi && x_lo && w_hi && w_lo && y_hi && y_lo, "avt_lateral_x3: NULL pointer");
  AVT_REQUIRE(avt_lateral_x3_supported(cin, cout, kt), "avt_lateral_x3: unsupported layer Cin=%d Cout=%d kt=%d", cin, cout, kt);
  AVT_REQUIRE(batch > 0 && t > 0 && hw > 0 && st > 0 && pt >= 0 && pt < kt && ldx >= cin && ldy >= cout && ldx % 8 == 0 && ldy % 8 == 0,
              "avt_lateral_x3: bad sizes / leading dimensions");
  AVT_REQUIRE(avt::aligned16(x_hi) && avt::aligned16(x_lo) && avt::aligned16(w_hi) && avt::aligned16(w_lo) && avt::aligned16(y_hi) &&
                  avt::aligned16(y_lo) && (!bias || avt::aligned16(bias)) && (!wscale || avt::aligned16(wscale)),
              "avt_lateral_x3: pointers must be 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_lateral_x3: bad plane_dtype");
  const int to = (t + 2 * pt - kt) / st + 1;
  const int64_t m = (int64_t)batch * to * hw;
  AVT_REQUIRE(to > 0 && m < (1ll << 31) - 16, "avt_lateral_x3: no output frames / too many rows");
  const int k = kt * cin, k1s = (k + 31) / 32, n = (cout + 31) / 32 * 32, nt1 = pick_nt1(k1s, n);
  PxArgs a;
  a.xh = static_cast<const uint16_t*>(x_hi);
  a.xl = static_cast<const uint16_t*>(x_lo);
  a.rh = a.rl = nullptr;
  a.yh = static_cast<uint16_t*>(y_hi);
  a.yl = static_cast<uint16_t*>(y_lo);
  a.wh = static_cast<const i32x4*>(w_hi);  // fused_slowfast.pack_pw_planes over the conv's own [Cout (padded to 32), kt * Cin] rows
  a.wl = static_cast<const i32x4*>(w_lo);
  a.bias = bias;      // [n] (padded with the weights)
  a.wscale = wscale;
  a.M = (int)m;
  a.ldx = ldx;
  a.ldr = 0;
  a.ldy = ldy;
  a.k1c = k / 8;
  a.ntiles = (int)((m + 15) / 16);
  a.relu = relu;
  a.n_chunks = n / (16 * nt1);
  a.taps = kt;
  a.cin8 = cin / 8;
  a.T = t;
  a.To = to;
  a.HW = hw;
  a.tst = st;
  a.tpad = pt;
  a.n_valid = cout;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return plane_dtype == AVT_X3_F16 ? dispatch_k<true>(a, k1s, nt1, s) : dispatch_k<false>(a, k1s, nt1, s);
}

// (k1, n1, n2) of the chained form: the slow res2 pair 64 -> 256 (+ residual) -> 64
extern "C" int avt_pw_chain_x3_supported(int k1, int n1, int n2) { return (k1 == 64 && n1 == 256 && n2 == 64) ? 1 : 0; }

extern "C" int avt_pw_chain_x3(const void* x_hi, const void* x_lo, int ldx, int k1, const void* w1_hi, const void* w1_lo, const float* bias1,
                               const float* wscale1, const void* res_hi, const void* res_lo, int ldr, void* y_hi, void* y_lo, int ldy,
                               int n1, int relu1, const void* w2_hi, const void* w2_lo, const float* bias2, const float* wscale2,
                               void* z_hi, void* z_lo, int ldz, int n2, int64_t m, int plane_dtype, void* stream) {
  AVT_REQUIRE(x_hi && x_lo && w1_hi && w1_lo && y_hi && y_lo && w2_hi && w2_lo && z_hi && z_lo && (!res_hi == !res_lo),
              "avt_pw_chain_x3: NULL pointer / half a plane pair");
  AVT_REQUIRE(avt_pw_chain_x3_supported(k1, n1, n2), "avt_pw_chain_x3: unsupported pair K1=%d N1=%d N2=%d (64 -> 256 -> 64)", k1, n1, n2);
  AVT_REQUIRE(m > 0 && m < (1ll << 31) - 16 && ldx >= k1 && ldy >= n1 && ldz >= n2 && ldx % 8 == 0 && ldy % 8 == 0 && ldz % 8 == 0 &&
                  (!res_hi || (ldr >= n1 && ldr % 8 == 0)),
              "avt_pw_chain_x3: bad sizes / leading dimensions");
  AVT_REQUIRE(avt::aligned16(x_hi) && avt::aligned16(x_lo) && avt::aligned16(w1_hi) && avt::aligned16(w1_lo) && avt::aligned16(y_hi) &&
                  avt::aligned16(y_lo) && avt::aligned16(w2_hi) && avt::aligned16(w2_lo) && avt::aligned16(z_hi) && avt::aligned16(z_lo) &&
                  (!res_hi || (avt::aligned16(res_hi) && avt::aligned16(res_lo))) && (!bias1 || avt::aligned16(bias1)) &&
                  (!wscale1 || avt::aligned16(wscale1)) && (!bias2 || avt::aligned16(bias2)) && (!wscale2 || avt::aligned16(wscale2)),
              "avt_pw_chain_x3: pointers must be 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_pw_chain_x3: bad plane_dtype");
  PcArgs c;
  PxArgs& a = c.p;
  a.xh = static_cast<const uint16_t*>(x_hi);
  a.xl = static_cast<const uint16_t*>(x_lo);
  a.rh = static_cast<const uint16_t*>(res_hi);
  a.rl = static_cast<const uint16_t*>(res_lo);
  a.yh = static_cast<uint16_t*>(y_hi);
  a.yl = static_cast<uint16_t*>(y_lo);
  a.wh = static_cast<const i32x4*>(w1_hi);
  a.wl = static_cast<const i32x4*>(w1_lo);
  a.bias = bias1;
  a.wscale = wscale1;
  a.M = (int)m;
  a.ldx = ldx;
  a.ldr = ldr;
  a.ldy = ldy;
  a.k1c = k1 / 8;
  a.ntiles = (int)((m + 15) / 16);
  a.relu = relu1;
  a.n_chunks = 1;
  a.n_rg = 0;
  a.taps = 1;
  a.cin8 = a.T = a.To = a.HW = a.tst = a.tpad = 0;
  a.n_valid = n1;
  c.w2h = static_cast<const i32x4*>(w2_hi);
  c.w2l = static_cast<const i32x4*>(w2_lo);
  c.bias2 = bias2;
  c.wscale2 = wscale2;
  c.zh = static_cast<uint16_t*>(z_hi);
  c.zl = static_cast<uint16_t*>(z_lo);
  c.ldz = ldz;
  constexpr int K1S = 2, NT1 = 16, NT2 = 4;
  constexpr int lds_bytes = 2 * NT1 * K1S * 1024 + 2 * NT2 * (NT1 / 2) * 1024 + (2 * NT1 * 16 + 2 * NT2 * 16) * 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(pw_chain_x3_kernel<K1S, NT1, NT2, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  static const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(pw_chain_x3_kernel<K1S, NT1, NT2, false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    avt::set_error("avt_pw_chain_x3: hipFuncSetAttribute(%d B LDS) failed", lds_bytes);
    return AVT_ERR_LAUNCH;
  }
  int grid = (a.ntiles + PX_NW - 1) / PX_NW;
  if (grid > 256) grid = 256;  // persistent: one workgroup per CU (129 KB of fragments)
  if (plane_dtype == AVT_X3_F16)
    hipLaunchKernelGGL((pw_chain_x3_kernel<K1S, NT1, NT2, true>), dim3((unsigned)grid), dim3(PX_NW * 64), lds_bytes, st, c);
  else
    hipLaunchKernelGGL((pw_chain_x3_kernel<K1S, NT1, NT2, false>), dim3((unsigned)grid), dim3(PX_NW * 64), lds_bytes, st, c);
  return avt::check_launch("avt_pw_chain_x3");
}

// The training form of avt_pw_x3 (see include/avt.h): fp32 rows in / out, plain [n][k] weight planes
extern "C" int avt_pw_x3_f32_supported(int k, int n) {
  const int k1s = (k + 31) / 32;
  if (k % 8 || n % 32 || !(k1s == 1 || k1s == 2 || k1s == 4 || k1s == 8)) return 0;
  const int nt1 = pick_nt1(k1s, n);
  if (!nt1) return 0;
  if ((nt1 == 16 && k1s > 4) || (nt1 == 8 && k1s > 8)) return 0;
  return n / (16 * nt1) <= 8 ? 1 : 0;
}

extern "C" int avt_pw_x3_f32(const float* x, int ldx, int k, const void* w_hi, const void* w_lo, const float* wscale, const float* add,
                             int lda, float* y, int ldy, int n, int64_t m, int plane_dtype, void* stream) {
  AVT_REQUIRE(x && w_hi && w_lo && y, "avt_pw_x3_f32: NULL pointer");
  AVT_REQUIRE(avt_pw_x3_f32_supported(k, n), "avt_pw_x3_f32: unsupported layer K=%d N=%d", k, n);
  AVT_REQUIRE(m > 0 && m < (1ll << 31) - 16 && ldx >= k && ldy >= n && ldx % 4 == 0 && ldy % 4 == 0 && (!add || (lda >= n && lda % 4 == 0)),
              "avt_pw_x3_f32: bad sizes / leading dimensions");
  AVT_REQUIRE(avt::aligned16(x) && avt::aligned16(w_hi) && avt::aligned16(w_lo) && avt::aligned16(y) && (!add || avt::aligned16(add)) &&
                  (!wscale || avt::aligned16(wscale)),
              "avt_pw_x3_f32: pointers must be 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_pw_x3_f32: bad plane_dtype");
  const int k1s = (k + 31) / 32, nt1 = pick_nt1(k1s, n);
  PfArgs a;
  a.x = x; a.add = add; a.y = y;
  a.wh = static_cast<const uint16_t*>(w_hi);
  a.wl = static_cast<const uint16_t*>(w_lo);
  a.wscale = wscale;
  a.M = (int)m; a.ldx = ldx; a.lda = lda; a.ldy = ldy; a.K = k; a.k1c = k / 8;
  a.ntiles = (int)((m + 15) / 16);
  a.n_chunks = n / (16 * nt1);
  a.stat_part = nullptr;
  a.n_total = n;
  a.bst_x = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return plane_dtype == AVT_X3_F16 ? dispatch_pf_k<true>(a, k1s, nt1, st) : dispatch_pf_k<false>(a, k1s, nt1, st);
}

// ... leaving the train-mode BatchNorm statistics of its output behind (include/avt.h; the streaming counterpart of
// avt_conv3d_igemm_x3_f32_stats): rows of partials per group = 8 x the row groups of the launch
extern "C" int avt_pw_x3_f32_stat_rows(int k, int n, int64_t m, int groups) {
  if (!avt_pw_x3_f32_supported(k, n) || groups < 1 || m <= 0 || m % groups) return -1;
  const int k1s = (k + 31) / 32, nt1 = pick_nt1(k1s, n);
  const int ntiles = (int)((m / groups + 15) / 16);
  return pf_row_groups(pf_lds_of(k1s, nt1, 1), n / (16 * nt1), ntiles, groups) * PX_NW;
}

extern "C" int avt_pw_x3_f32_stats(const float* x, int ldx, int k, const void* w_hi, const void* w_lo, const float* wscale, float* y, int ldy,
                                   int n, int64_t m, int plane_dtype, void* stat_part, int groups, void* stream) {
  AVT_REQUIRE(x && w_hi && w_lo && y && stat_part, "avt_pw_x3_f32_stats: NULL pointer");
  AVT_REQUIRE(avt_pw_x3_f32_supported(k, n), "avt_pw_x3_f32_stats: unsupported layer K=%d N=%d", k, n);
  AVT_REQUIRE(groups >= 1 && groups <= 65535 && m > 0 && m % groups == 0 && m < (1ll << 31) - 16 && ldx >= k && ldy >= n && ldx % 4 == 0 &&
                  ldy % 4 == 0 && n >= 8 && (n & (n - 1)) == 0 && n <= 4096,
              "avt_pw_x3_f32_stats: bad sizes / leading dimensions / groups (n a power of two: the BatchNorm's domain)");
  AVT_REQUIRE(avt::aligned16(x) && avt::aligned16(w_hi) && avt::aligned16(w_lo) && avt::aligned16(y) && avt::aligned16(stat_part) &&
                  (!wscale || avt::aligned16(wscale)),
              "avt_pw_x3_f32_stats: pointers must be 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_pw_x3_f32_stats: bad plane_dtype");
  const int k1s = (k + 31) / 32, nt1 = pick_nt1(k1s, n);
  PfArgs a;
  a.x = x; a.add = nullptr; a.y = y;
  a.wh = static_cast<const uint16_t*>(w_hi);
  a.wl = static_cast<const uint16_t*>(w_lo);
  a.wscale = wscale;
  a.M = (int)(m / groups); a.ldx = ldx; a.lda = 0; a.ldy = ldy; a.K = k; a.k1c = k / 8;
  a.ntiles = (a.M + 15) / 16;
  a.n_chunks = n / (16 * nt1);
  a.stat_part = static_cast<double*>(stat_part);
  a.n_total = n;
  a.bst_x = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return plane_dtype == AVT_X3_F16 ? dispatch_pf_k<true>(a, k1s, nt1, st, groups) : dispatch_pf_k<false>(a, k1s, nt1, st, groups);
}

// The streaming input gradient whose result is a train-mode BatchNorm's OUTPUT gradient (include/avt.h; the counterpart of
// avt_conv3d_igemm_x3_f32_bwdstats for the pointwise layers)
extern "C" int avt_pw_x3_f32_bwdstats_rows(int k, int n, int64_t m, int groups) {
  if (!avt_pw_x3_f32_supported(k, n) || groups < 1 || m <= 0 || m % groups || n < 8 || (n & (n - 1)) || n > 4096) return -1;
  int k1s = (k + 31) / 32, nt1 = pick_nt1(k1s, n);
  if (nt1 == 16) nt1 = 8;  // (as the launch below)
  const int ntiles = (int)((m / groups + 15) / 16);
  return pf_row_groups(pf_lds_of(k1s, nt1, 2), n / (16 * nt1), ntiles, groups) * PX_NW;
}

extern "C" int avt_pw_x3_f32_bwdstats(const float* x, int ldx, int k, const void* w_hi, const void* w_lo, const float* add, int lda, float* y,
                                      int ldy, int n, int64_t m, int plane_dtype, const float* bn_x, const float* bn_mean,
                                      const float* bn_invstd, const float* bn_gamma, const float* bn_beta, const void* bn_mask, int relu,
                                      void* stat_part, int groups, void* stream) {
  AVT_REQUIRE(x && w_hi && w_lo && y && stat_part && bn_x && bn_mean && bn_invstd && bn_gamma, "avt_pw_x3_f32_bwdstats: NULL pointer");
  AVT_REQUIRE(avt_pw_x3_f32_bwdstats_rows(k, n, m, groups) > 0, "avt_pw_x3_f32_bwdstats: unsupported layer K=%d N=%d / %lld rows in %d groups",
              k, n, (long long)m, groups);
  AVT_REQUIRE(m < (1ll << 31) - 16 && ldx >= k && ldy == n && ldx % 4 == 0 && (!add || (lda >= n && lda % 4 == 0)) &&
                  (!relu || bn_mask || bn_beta) && plane_dtype == AVT_X3_BF16,
              "avt_pw_x3_f32_bwdstats: contiguous output rows (ldy == n), bf16 planes, a mask or beta for the ReLU");
  AVT_REQUIRE(avt::aligned16(x) && avt::aligned16(w_hi) && avt::aligned16(w_lo) && avt::aligned16(y) && avt::aligned16(stat_part) &&
                  avt::aligned16(bn_x) && (!add || avt::aligned16(add)),
              "avt_pw_x3_f32_bwdstats: pointers must be 16-byte aligned");
  int k1s = (k + 31) / 32, nt1 = pick_nt1(k1s, n);
  // (16 tiles per wave spill 13-84 registers in this mode, and a scratch reload's s_waitcnt vmcnt(0) serializes the tile's loads: 8
  //  tiles and twice the chunks — the dy rows read twice, a small operand here — are 10 % (K = 64) to 27 % (K = 128) faster)
  if (nt1 == 16) nt1 = 8;
  PfArgs a;
  a.x = x; a.add = add; a.y = y;
  a.wh = static_cast<const uint16_t*>(w_hi);
  a.wl = static_cast<const uint16_t*>(w_lo);
  a.wscale = nullptr;
  a.M = (int)(m / groups); a.ldx = ldx; a.lda = lda; a.ldy = ldy; a.K = k; a.k1c = k / 8;
  a.ntiles = (a.M + 15) / 16;
  a.n_chunks = n / (16 * nt1);
  a.stat_part = static_cast<double*>(stat_part);
  a.n_total = n;
  a.bst_x = bn_x; a.bst_mean = bn_mean; a.bst_invstd = bn_invstd; a.bst_gamma = bn_gamma; a.bst_beta = bn_beta;
  a.bst_mask = static_cast<const uint8_t*>(bn_mask); a.bst_relu = relu;
  return dispatch_pf_k<false>(a, k1s, nt1, static_cast<hipStream_t>(stream), groups);
}
