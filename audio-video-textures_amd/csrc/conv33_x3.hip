// conv33_x3 — the [1,3,3] 64 -> 64, stride-1 convolution of the slow pathway's res2 bottlenecks (BN + ReLU folded in) in the
// contract-grade split-plane arithmetic (conv_x3.hip's number format), the x3 counterpart of the bf16 path's conv33_c64.hip
// (blocks of the third-party SlowFast model the reference runs per clip window, contrastive_video_textures/models/models.py:
// 335, 399).  On the general tile this layer ran at 0.29 of the x3 roof: with K = 576 and only 64 output channels a 128 x 64
// tile moves 48 KB of operands per 64-wide K-step from L2 into the LDS for half a wide tile's flops, nine times over the
// same activation rows (one gather per tap).
// Here nothing is staged: a WAVE owns 32 consecutive positions of a frame row-major and all 64 output channels; the
// activation fragments of tap (dy, dx) are the MFMA B operand AS LOADED — lane (position l & 31, k-half l >> 5) reads the
// 16 bytes of its shifted position's channel chunk straight from global memory (the nine taps re-read the same three rows:
// L1 / L2 hits; taps that fall off the frame carry an out-of-bounds buffer offset and arrive as the zero padding) — and
// the weights, 147 KB as MFMA fragments of both planes, live in the LDS of a persistent 8-wave workgroup for the whole launch.
// No barrier after the prologue; the next row's loads are in flight under the current row's 72 MFMAs.
// The output rows are permuted in the packing so that a lane ends with 16 channels as two 16-byte stores per plane that sit
// next to its partner lane's: a position's 32 channels of a tile are one contiguous 64 bytes per plane per instruction pair.
// Roofline: MFMA at 1/3 of the f16 / bf16 peak (2 * M * 576 * 64 flop counted once).
#include <stdlib.h>

#include <type_traits>

#include "avt_common.h"
#include "split_planes.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFFF0u;

template <bool F16>
__device__ __forceinline__ f32x16 mfma(i32x4 w, i32x4 x, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}

struct C33Args {
  const uint16_t* xh;
  const uint16_t* xl;
  uint16_t* oh;
  uint16_t* ol;
  const i32x4* wf;     // [9 taps][4 k-slices][2 n-tiles][2 planes][64 lanes]
  const float* coef;   // [scale 64 | bias 64] in the PACKED channel order's inverse: indexed by output channel
  int M, H, W, ldi, ldo, relu, ntiles;
  unsigned x_bytes, o_bytes;
};

constexpr int NW = 8;  // 8 waves = two per SIMD (two row sets of operands in flight: ~200 registers)
constexpr int NFR = 9 * 4 * 2 * 2;  // fragments of 1 KB

// lane i <- lane i - 1 / lane i + 1 over the whole wave (DPP wave_shr:1 / wave_shl:1)
__device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }

template <bool F16>
__global__ __launch_bounds__(NW * 64, 2) void conv33_x3_kernel(C33Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  for (int f = wid; f < NFR; f += NW) *reinterpret_cast<i32x4*>(lds + f * 1024 + lane * 16) = a.wf[f * 64 + lane];
  float* cf = reinterpret_cast<float*>(lds + NFR * 1024);
  for (int i = tid; i < 128; i += NW * 64) cf[i] = a.coef[i];
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)a.xh, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)a.xl, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t roh = __builtin_amdgcn_make_buffer_rsrc((void*)a.oh, 0, a.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rol = __builtin_amdgcn_make_buffer_rsrc((void*)a.ol, 0, a.o_bytes, 0x00020000);
  const int HW = a.H * a.W;
  const int stride_t = gridDim.x * NW;
  const bool ledge = lr == 0, redge = lr == 31;

  // One ROW of taps (dy fixed) = two load sets: C, this lane's own column of row y + dy, and E, the column outside the tile
  // for the two edge lanes (lane 0 of a half: x - 1; lane 31: x + 1; every other lane carries an out-of-bounds offset and
  // costs no bandwidth).  The dx = -1 / +1 operands of the interior lanes are C shifted by one lane (DPP): a row costs one
  // pass over its activations instead of three, the tile 3.2 passes instead of 9 — the first form of this kernel loaded every
  // tap by itself and ran at the L2 -> CU rate (208 TFLOP/s; the general tile 241).
  struct Row {
    i32x4 c[4][2], e[4][2];  // [k-slice][plane]
  };
  struct Geo {  // this lane's position in a tile
    unsigned base;  // byte offset of (position, channel chunk lh)
    int y, x;
    bool in;
  };
  auto geo_of = [&](int tile) {
    Geo g;
    const int p = tile * 32 + lr;
    const int rem = p % HW;
    g.y = rem / a.W;
    g.x = rem - g.y * a.W;
    g.in = p < a.M;
    g.base = ((unsigned)p * (unsigned)a.ldi + (unsigned)(lh * 8)) * 2u;
    return g;
  };
  auto load_row = [&](Row& r, const Geo& g, int dy) {
    const bool rowok = g.in && (unsigned)(g.y + dy) < (unsigned)a.H;
    const unsigned oc = rowok ? g.base + (unsigned)(dy * a.W * a.ldi * 2) : kOob;  // (negative shifts wrap modulo 2^32: exact)
    const bool eok = rowok && ((ledge && g.x > 0) || (redge && g.x < a.W - 1));
    const unsigned oe = eok ? oc + (unsigned)((ledge ? -1 : 1) * a.ldi * 2) : kOob;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int o1 = (int)(oc != kOob ? oc + (unsigned)(k * 32) : kOob), o2 = (int)(oe != kOob ? oe + (unsigned)(k * 32) : kOob);
      r.c[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rxh, o1, 0, 0);
      r.c[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rxl, o1, 0, 0);
      r.e[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rxh, o2, 0, 0);
      r.e[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rxl, o2, 0, 0);
    }
  };
  // the three taps of a row into the accumulators
  auto mul_row = [&](f32x16* acc, const Row& r, const Geo& g, int dy) {
    const bool lz = g.x == 0, rz = g.x == a.W - 1;  // the neighbour is the frame's zero padding
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      const int tap = (dy + 1) * 3 + dxi;
      int lofs = lane * 16;  // opaque per tap: the fragment reads are not hoisted out of the tile loop into registers
      asm volatile("" : "+v"(lofs));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        i32x4 xo[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          if (dxi == 1) {
            xo[pl] = r.c[k][pl];
          } else {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const int sh = dxi == 0 ? wave_shr1(r.c[k][pl][d]) : wave_shl1(r.c[k][pl][d]);
              const bool edge = dxi == 0 ? ledge : redge, zero = dxi == 0 ? lz : rz;
              xo[pl][d] = edge ? r.e[k][pl][d] : (zero ? 0 : sh);
            }
          }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int f = ((tap * 4 + k) * 2 + n) * 2;
          const i32x4 wh = *reinterpret_cast<const i32x4*>(lds + f * 1024 + lofs);
          const i32x4 wl = *reinterpret_cast<const i32x4*>(lds + (f + 1) * 1024 + lofs);
          acc[n] = mfma<F16>(wl, xo[0], acc[n]);  // small terms first
          acc[n] = mfma<F16>(wh, xo[1], acc[n]);
          acc[n] = mfma<F16>(wh, xo[0], acc[n]);
        }
      }
    }
  };
  // epilogue: D row rho = (r & 3) + 8 (r >> 2) + 4 lh of n-tile n holds channel 32 n + (2 (r >> 3) + lh) * 8 + (r & 7) (packing)
  auto epilogue = [&](const f32x16* ac, int tl) {
    const int pp = tl * 32 + lr;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int c0 = 32 * n + (2 * g + lh) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = ac[n][8 * g + e] * cf[c0 + e] + cf[64 + c0 + e];
          if (a.relu) v[e] = avt::relu_keep_nan(v[e]);
        }
        uint4 oh, ol;
        avt::split2<F16>(v[0], v[1], oh.x, ol.x);
        avt::split2<F16>(v[2], v[3], oh.y, ol.y);
        avt::split2<F16>(v[4], v[5], oh.z, ol.z);
        avt::split2<F16>(v[6], v[7], oh.w, ol.w);
        const int off = (int)(pp < a.M ? ((unsigned)pp * (unsigned)a.ldo + (unsigned)c0) * 2u : kOob);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, oh), roh, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, ol), rol, off, 0, 0);
      }
  };

  // rows stream through two register sets: while a row's 72 MFMAs run, the next row (of this tile or the next) is in flight
  Row ra, rb;
  int tile = blockIdx.x * NW + wid;
  if (tile >= a.ntiles) return;
  Geo g = geo_of(tile);
  load_row(ra, g, -1);
  for (;;) {  // two tiles per trip: three rows each, the register sets alternating a b a | b a b
    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    load_row(rb, g, 0);
    mul_row(acc, ra, g, -1);
    load_row(ra, g, 1);
    mul_row(acc, rb, g, 0);
    const int t2 = tile + stride_t;
    const bool more2 = t2 < a.ntiles;
    Geo g2 = geo_of(more2 ? t2 : tile);
    if (more2) load_row(rb, g2, -1);
    mul_row(acc, ra, g, 1);
    epilogue(acc, tile);
    if (!more2) break;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    load_row(ra, g2, 0);
    mul_row(acc, rb, g2, -1);
    load_row(rb, g2, 1);
    mul_row(acc, ra, g2, 0);
    tile = t2 + stride_t;
    const bool more3 = tile < a.ntiles;
    g = geo_of(more3 ? tile : t2);
    if (more3) load_row(ra, g, -1);
    mul_row(acc, rb, g2, 1);
    epilogue(acc, t2);
    if (!more3) break;
  }
}

}  // namespace

extern "C" int avt_conv33_x3_supported(int cin, int cout) { return (cin == 64 && cout == 64) ? 1 : 0; }

extern "C" int avt_conv33_x3(const void* x_hi, const void* x_lo, const void* wfrag, const float* coef, void* out_hi, void* out_lo,
                             int batch, int t, int h, int w, int ldi, int ldo, int relu, int plane_dtype, void* stream) {
  AVT_REQUIRE(x_hi && x_lo && wfrag && coef && out_hi && out_lo, "avt_conv33_x3: NULL pointer");
  AVT_REQUIRE(batch > 0 && t > 0 && h > 0 && w > 0 && ldi >= 64 && ldo >= 64 && ldi % 8 == 0 && ldo % 8 == 0,
              "avt_conv33_x3: bad sizes (64 -> 64 channels; leading dimensions in multiples of 8)");
  AVT_REQUIRE(avt::aligned16(x_hi) && avt::aligned16(x_lo) && avt::aligned16(wfrag) && avt::aligned16(coef) && avt::aligned16(out_hi) &&
                  avt::aligned16(out_lo),
              "avt_conv33_x3: pointers must be 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_conv33_x3: bad plane_dtype");
  const int64_t m = (int64_t)batch * t * h * w;
  AVT_REQUIRE(m * ldi * 2 < (1ll << 32) - 64 && m * ldo * 2 < (1ll << 32) - 64, "avt_conv33_x3: tensor too large for 32-bit offsets");
  C33Args a;
  a.xh = static_cast<const uint16_t*>(x_hi);
  a.xl = static_cast<const uint16_t*>(x_lo);
  a.oh = static_cast<uint16_t*>(out_hi);
  a.ol = static_cast<uint16_t*>(out_lo);
  a.wf = static_cast<const i32x4*>(wfrag);
  a.coef = coef;
  a.M = (int)m;
  a.H = h;
  a.W = w;
  a.ldi = ldi;
  a.ldo = ldo;
  a.relu = relu;
  a.ntiles = (int)((m + 31) / 32);
  a.x_bytes = (unsigned)(m * ldi * 2);
  a.o_bytes = (unsigned)(m * ldo * 2);
  constexpr int lds_bytes = NFR * 1024 + 128 * 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto kern = plane_dtype == AVT_X3_F16 ? conv33_x3_kernel<true> : conv33_x3_kernel<false>;
  static const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv33_x3_kernel<true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  static const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv33_x3_kernel<false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    avt::set_error("avt_conv33_x3: hipFuncSetAttribute(%d B LDS) failed", lds_bytes);
    return AVT_ERR_LAUNCH;
  }
  int grid = (a.ntiles + NW - 1) / NW;
  if (grid > 256) grid = 256;  // persistent: one workgroup per CU (147 KB of weights each)
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds_bytes, st, a);
  return avt::check_launch("avt_conv33_x3");
}
