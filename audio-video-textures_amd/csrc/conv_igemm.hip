// conv3d_igemm — implicit-GEMM 3D convolution on the gfx950 matrix cores, for the SlowFast encoder.
// Replaces the cuDNN/MIOpen conv3d + BatchNorm + ReLU (+ residual add) sequence the reference runs through
// the third-party SlowFast model for EVERY clip window (contrastive_video_textures/models/models.py:335, 399)
// — ~all the FLOPs of the hot path (100 GFLOP per clip per encoder).
//
//   out[m, n] = act( sum_k A[m, k] * Wt[n, k] + bias[n] (+ res[m, n]) )
//     m = (b, to, ho, wo) output position, n = output channel,
//     k = ((dt*KH + dh)*KW + dw)*Cin + c   — Cin innermost, i.e. activations are NDHWC (channels-last-3d) bf16
//     Wt = BN-folded weights packed [Cout, K] bf16 (K contiguous), bias = folded BN shift, fp32.
//
// Both operands have K contiguous per row, so a 16-byte chunk of A is 8 consecutive input channels of ONE tap:
// the "im2col" is only an address computation — per-row base offset + per-K-chunk tap offset from a small table
// (copied to LDS once per workgroup) + a per-row separable bitmask of in-bounds taps.  Operands are fetched with
// SRSRC buffer loads: padding taps, rows past M and the K tail get an offset beyond the descriptor's extent and the
// hardware range check returns zeros, so the K loop has no branch and no exec masking.  Rows are decoded with
// magic-number division (1x1x1 stride-1 layers skip the decode).
//
// The MFMA is issued with the weights as the first operand (D = Wt * A^T), so each lane ends up holding 4
// CONSECUTIVE CHANNELS of one output position: the epilogue packs them with v_cvt_pk_bf16_f32, stages the tile
// through LDS, and writes / reads (residual, requested before the staging) global memory in 16-byte row-contiguous
// chunks.  BN, ReLU, residual add and the channel-slice write of the lateral-fusion concat are fused here, so every
// activation tensor crosses HBM once per consumer.
//
// Tiles (256 threads = 4 waves, v_mfma_f32_32x32x16_bf16, fp32 accumulate), <BM, BN, WTM = wave-tile rows>:
//   <128,128, 64>: 2x2 waves, wave tile  64(m) x 64(n)   — wide layers
//   <256, 64, 64>: 4x1 waves, wave tile  64(m) x 64(n)   — Cout = 64
//   <256, 32, 64>: 4x1 waves, wave tile  64(m) x 32(n)   — few output channels
// LDS rows are padded to 144 B (conflict-free ds_read_b128, see sim_gemm.hip).  Roofline: MFMA for the wide
// 3x3 / temporal layers, HBM for the 1x1x1 (+ residual) layers (bytes = activations in + out (+ residual)).
// Tried and measured slower (profiles/r01/probe_glds_ab.log, probe_earlyres_ab.log): LDS-DMA (global_load_lds)
// staging with a source-side swizzle and two LDS stages; requesting the residual before the K loop (+32 VGPRs).
#include <stdlib.h>

#include <type_traits>

#include "avt_common.h"
#include "conv_args.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#ifdef AVT_CONV_STAMP
// diagnostic build only (`make stamp` -> libavt_hip_stamp.so, never the shipped library): the hooks live in tools/diag
#include "../../tools/diag/conv_stamp.h"
#else
#define STAMP_BEGIN()
#define STAMP(i)
#define STAMP_FINE(i)
#define STAMP_END()
#endif

// waves per SIMD to keep resident: bounds the register allocation (guide §6 G1)
#define AVT_CONV_MIN_WAVES(BM, BN, WTM) ((WTM) == 128 ? 2 : ((BM) == 128 ? 3 : ((BN) == 64 ? 2 : 4)))

template <int BM, int BN, int WTM, bool TABLDS>
__global__ __launch_bounds__(256, AVT_CONV_MIN_WAVES(BM, BN, WTM)) void conv_igemm_kernel(ConvArgs a) {
  constexpr int WAVES_M = BM / WTM;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = BN / WAVES_N;  // wave tile width in n
  constexpr int NT = WN / 32, MT = WTM / 32;
  constexpr int AU = BM / 32, BU = BN / 32;  // 16-byte chunks per thread per K-step
  constexpr int A_BYTES = BM * LSTR;
  constexpr int STAGE = (BM + BN) * LSTR;  // one K-step of both operands
  constexpr int ESTR = BN * 2 + 16;        // epilogue staging row stride (bytes)
  // the epilogue stages EROWS rows at a time: the whole tile when it fits in the workgroup's LDS, else one wave-row
  constexpr int EPASS = (BM * ESTR <= STAGE + (TABLDS ? kMaxTabSteps * 64 : 0)) ? 1 : WAVES_M;
  constexpr int EROWS = BM / EPASS;
  constexpr int CPR = BN / 8;  // 16-byte chunks per output row
  constexpr int EU = (EROWS * CPR) / 256;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  // XCD-aware tile order (see sim_gemm.hip): consecutive tiles of one XCD share the activation rows
  const int bid = blockIdx.x;
  const int qd = a.nblk / 8, rm = a.nblk % 8, xc = bid % 8;
  const int swz = (xc < rm ? xc * (qd + 1) : rm * (qd + 1) + (xc - rm) * qd) + bid / 8;
  const int tm = swz / a.tiles_n, tn = swz % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;
  const int r0 = tid >> 3, c16 = tid & 7;

  // ---- per-row state of the A gather: base element offset and the bitmask of in-bounds taps
  int rowoff[AU];
  unsigned rowmask[AU];
#pragma unroll
  for (int u = 0; u < AU; ++u) {
    const int m = m0 + r0 + 32 * u;
    rowoff[u] = 0;
    rowmask[u] = 0u;
    if (m < a.M && a.pointwise) {
      // 1x1x1, stride 1, no padding: output position == input position, its single tap is always in bounds
      rowoff[u] = m * a.ldi;
      rowmask[u] = 0x010101u;
    } else if (m < a.M) {
      const int t1 = (int)fastdiv((uint32_t)m, a.dWo), wo = m - t1 * a.Wo;
      const int t2 = (int)fastdiv((uint32_t)t1, a.dHo), ho = t1 - t2 * a.Ho;
      const int b = (int)fastdiv((uint32_t)t2, a.dTo), to = t2 - b * a.To;
      const int ti0 = to * a.st - a.pt, hi0 = ho * a.sh - a.ph, wi0 = wo * a.sw - a.pw;
      rowoff[u] = (((b * a.T + ti0) * a.H + hi0) * a.W + wi0) * a.ldi;
      // in-bounds taps are separable: bits 0-7 = dt, 8-15 = dh, 16-23 = dw
      unsigned mask = 0u;
      for (int dt = 0; dt < a.KT; ++dt) mask |= ((unsigned)(ti0 + dt) < (unsigned)a.T ? 1u : 0u) << dt;
      for (int dh = 0; dh < a.KH; ++dh) mask |= ((unsigned)(hi0 + dh) < (unsigned)a.H ? 1u : 0u) << (8 + dh);
      for (int dw = 0; dw < a.KW; ++dw) mask |= ((unsigned)(wi0 + dw) < (unsigned)a.W ? 1u : 0u) << (16 + dw);
      rowmask[u] = mask;
    }
  }
  // rows of the weight tile this thread stages
  int wrow[BU];
  unsigned wsel[BU];  // all-ones where the weight row exists
#pragma unroll
  for (int u = 0; u < BU; ++u) {
    const int n = n0 + r0 + 32 * u;
    wrow[u] = n < a.Cout ? n * a.K : 0;
    wsel[u] = n < a.Cout ? 0xFFFFFFFFu : 0u;
  }

  f32x16 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[MT], wf[NT];
      const int koff = ks * 32 + lh * 16;
#pragma unroll
      for (int j = 0; j < MT; ++j)
        af[j] = *reinterpret_cast<const bf16x8*>(lds + (wm * WTM + j * 32 + lr) * LSTR + koff);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        wf[i] = *reinterpret_cast<const bf16x8*>(lds + A_BYTES + (wn * WN + i * 32 + lr) * LSTR + koff);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);  // D[n][m]
    }
  };

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
  int2* ltab = reinterpret_cast<int2*>(lds + STAGE);  // [nk*8] behind the operand slabs
  if constexpr (TABLDS) {
    for (int i = tid; i < a.nk * 8; i += 256) ltab[i] = a.ktab[i];
  }
  i32x4 ra[AU], rb[BU];
  auto gload = [&](int kt) {
    // {offset, tap bits}; chunks past K carry bit 31, which no row mask has -> never "ok".  A global read of the
    // table here would put a dependent ~500-cycle L2 round trip in front of every K-step's loads: it lives in LDS.
    const int2 e = TABLDS ? ltab[kt * 8 + c16] : a.ktab[kt * 8 + c16];
    const unsigned ebits = (unsigned)e.y;
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const unsigned sel = ((rowmask[u] & ebits) == ebits) ? 0xFFFFFFFFu : 0u;
      const unsigned off = (((unsigned)(rowoff[u] + e.x) * 2u) & sel) | (kOob & ~sel);
      ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)off, 0, 0);
    }
    const unsigned ksel = ~(unsigned)(e.y >> 31);  // all-ones for chunks inside K
    const unsigned kc2 = (unsigned)((kt * 8 + c16) * 16);
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const unsigned sel = ksel & wsel[u];
      const unsigned off = (((unsigned)wrow[u] * 2u + kc2) & sel) | (kOob & ~sel);
      rb[u] = __builtin_amdgcn_raw_buffer_load_b128(rwt, (int)off, 0, 0);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u = 0; u < AU; ++u) *reinterpret_cast<i32x4*>(lds + (r0 + 32 * u) * LSTR + c16 * 16) = ra[u];
#pragma unroll
    for (int u = 0; u < BU; ++u) *reinterpret_cast<i32x4*>(lds + A_BYTES + (r0 + 32 * u) * LSTR + c16 * 16) = rb[u];
  };
  if constexpr (TABLDS) __syncthreads();
  STAMP_BEGIN();
  gload(0);
  lstore();
  __syncthreads();
  STAMP(0);  // prologue: first slab
  for (int kt = 0; kt < a.nk; ++kt) {
    if (kt + 1 < a.nk) gload(kt + 1);  // next slab's latency hides under this slab's MFMAs
    STAMP(1);  // issue of the next slab's loads
    compute();
    STAMP(2);  // fragment reads + MFMAs
    __syncthreads();
    STAMP(3);  // barrier after compute (includes the vmcnt(0) the compiler puts in front of it)
    if (kt + 1 < a.nk) {
      lstore();
      STAMP(4);  // wait for the loads + ds_write
      __syncthreads();
      STAMP(5);  // barrier after the stores
    }
  }

  // ---- epilogue, EPASS passes of EROWS rows
  const bool has_res = a.res != nullptr;
#pragma unroll
  for (int p = 0; p < EPASS; ++p) {
    // the residual chunks this thread will need are requested before the accumulators are staged through LDS, so
    // their HBM latency overlaps the staging instead of serialising one round trip per chunk
    uint4 rres[EU];
    if (has_res) {
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int c = tid + 256 * u;
        const int m = m0 + p * EROWS + c / CPR, n = n0 + (c % CPR) * 8;
        rres[u] = (m < a.M && n < a.Cout) ? *reinterpret_cast<const uint4*>(a.res + (int64_t)m * a.ldr + n)
                                          : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    // phase 1: (+bias [, relu]) -> bf16 -> LDS staging tile [m][n]
    // D layout: column (lane & 31) = m, rows (reg&3) + 8*(reg>>2) + 4*(lane>>5) = n -> regs 4g..4g+3 are 4 consecutive n
    if (EPASS == 1 || wm == p) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn * WN + i * 32 + 8 * g + 4 * lh;  // tile-local first channel of this group
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a.bias && n0 + nl < a.Cout) bv = *reinterpret_cast<const float4*>(a.bias + n0 + nl);
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            float v0 = acc[i][j][4 * g + 0] + bv.x, v1 = acc[i][j][4 * g + 1] + bv.y;
            float v2 = acc[i][j][4 * g + 2] + bv.z, v3 = acc[i][j][4 * g + 3] + bv.w;
            if (a.relu && !has_res) {
              v0 = fmaxf(v0, 0.f);
              v1 = fmaxf(v1, 0.f);
              v2 = fmaxf(v2, 0.f);
              v3 = fmaxf(v3, 0.f);
            }
            uint2 pk;
            pk.x = avt::pack_bf16x2(v0, v1);
            pk.y = avt::pack_bf16x2(v2, v3);
            const int ml = (EPASS == 1 ? wm * WTM : 0) + j * 32 + lr;
            *reinterpret_cast<uint2*>(lds + ml * ESTR + nl * 2) = pk;
          }
        }
    }
    __syncthreads();
    // phase 2: 16-byte row-contiguous chunks: (+residual, relu) -> global
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int c = tid + 256 * u;
      const int row = c / CPR, cc = c % CPR;
      const int m = m0 + p * EROWS + row, n = n0 + cc * 8;
      if (m < a.M && n < a.Cout) {
        uint4 v = *reinterpret_cast<const uint4*>(lds + row * ESTR + cc * 16);
        if (has_res) {
          uint32_t* pv = reinterpret_cast<uint32_t*>(&v);
          const uint32_t* pr = reinterpret_cast<const uint32_t*>(&rres[u]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x0 = avt::bf16x2_lo(pv[e]) + avt::bf16x2_lo(pr[e]);
            float x1 = avt::bf16x2_hi(pv[e]) + avt::bf16x2_hi(pr[e]);
            if (a.relu) {
              x0 = fmaxf(x0, 0.f);
              x1 = fmaxf(x1, 0.f);
            }
            pv[e] = avt::pack_bf16x2(x0, x1);
          }
        }
        *reinterpret_cast<uint4*>(a.out + (int64_t)out_row(a, m) * a.ldo + n) = v;
      }
    }
    if (p + 1 < EPASS) __syncthreads();  // the staging tile is reused by the next wave-row
  }
  STAMP_END();
}

template <int BM, int BN, int WTM, bool TABLDS = true>
int launch(ConvArgs& a, hipStream_t st) {
  if (TABLDS && a.nk > kMaxTabSteps) return launch<BM, BN, WTM, false>(a, st);  // table stays in global memory
  const int tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  a.nblk = tiles_m * a.tiles_n;
  constexpr int lds_main = (BM + BN) * LSTR + (TABLDS ? kMaxTabSteps * 8 * 8 : 0);
  constexpr int epass = (BM * (BN * 2 + 16) <= lds_main) ? 1 : BM / WTM;
  constexpr int lds_epi = (BM / epass) * (BN * 2 + 16);
  constexpr int lds_bytes = lds_main > lds_epi ? lds_main : lds_epi;
  if (lds_bytes > 64 * 1024) {  // above the default dynamic-LDS limit: opt in once per kernel
    static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WTM, TABLDS>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
      avt::set_error("avt_conv3d_igemm_bf16: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
      return AVT_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WTM, TABLDS>), dim3((unsigned)a.nblk), dim3(256), lds_bytes, st, a);
  return avt::check_launch("avt_conv3d_igemm_bf16");
}

// ---------------------------------------------------------------------------------------------------------------
// XL tile: 256(m) x 256(n), 512 threads = 8 waves as 2(m) x 4(n), wave tile 128 x 64 — for the GEMM-like layers
// (Cout >= 256, long K).  Half the operand bytes per MFMA of the 128x128 tile (the vector-L1 / LDS-write path is
// what bounds those layers), and no VGPR staging or ds_write at all: both operand slabs go
// global -> LDS by LDS-DMA (global_load_lds, 16 B per lane).
//
// Pipeline unit = HALF a K-step (32 of K: A 256 rows x 64 B + B 256 rows x 64 B = 32 KB, 4 DMA instructions per
// thread), ring of 4 units (128 KB), ONE raw s_barrier per unit.  While unit i is multiplied, units i+1 and i+2 are
// landed or in flight and unit i+3 is being issued — two DMA instructions between the MFMA groups of each 16-wide
// k-slice, so the vector-L1 address pipe (64 B/clk, what a DMA burst would otherwise serialise in front of the MFMAs:
// measured 1,130 cycles per K-step, tools/probe_stamps.sh) works under the matrix pipe.  A unit is retired by a
// counted s_waitcnt vmcnt(8) (its successors stay in flight; never 0 inside the loop) followed by the barrier, which
// also proves every wave is done with unit i-1, whose slot unit i+3 then overwrites.
// A DMA wave-instruction fills 1 KB of LDS lane-linearly = 16 rows x 64 B; rows are unpadded, so the 16-byte slot of
// logical chunk c of row r is c ^ ((r >> 2) & 3): conflict-free for the ds_read_b128 lane groups of a 32-row fragment,
// applied on the SOURCE side of the DMA (each lane fetches the chunk that belongs in its fixed destination slot).
// Padding taps / tails read 16 zero bytes (behind the tap table); taps are decoded arithmetically (no table read in
// the loop: an LDS read there makes the compiler drain vmcnt(0)).
constexpr int XT = 512;   // 8 waves
constexpr int XRING = 4;  // pipeline units in the LDS ring
#ifndef XL_DECODE_IN_B
#define XL_DECODE_IN_B 1  // where the scalar tap decode runs: 0 = the read phase (A), 1 = under the MFMAs (B); measured equal
#endif

// Tile shapes <XBM, XBN, WM = waves along m>: <256,256,2> (wave tile 128x64) for Cout >= 256; <256,128,4> and <512,64,8>
// (wave tile 64x64) for the 128- and 64-channel layers.  The unit is (XBM + XBN) x 64 bytes in every shape.
template <int XBM, int XBN, int WM>
__global__ __launch_bounds__(XT, 2) void conv_xl_kernel(ConvArgs a) {
  constexpr int WN = 8 / WM, WTM = XBM / WM, WTN = XBN / WN, MT = WTM / 32, NT = WTN / 32;
  constexpr int XUNIT = (XBM + XBN) * 64;  // one 32-wide K slice of both operands
  constexpr int AIW = XBM / 128;           // A-operand DMA instructions per wave and unit (16 rows each)
  constexpr bool BHALF = XBN < 128;        // fewer B rows than 8 waves x 16: only waves 0 .. XBN/16-1 stage B
  constexpr int BIW = BHALF ? 1 : XBN / 128;
  constexpr int NMF = 2 * NT * MT;         // MFMAs of a phase B
  constexpr int ESTR = XBN * 2 + 16;
  constexpr int EPASS = (XBM * ESTR <= XRING * XUNIT) ? 1 : 2;  // epilogue staging passes through the ring's LDS
  constexpr int EROWS = XBM / EPASS;
  constexpr int CPR = XBN / 8;
  constexpr int EU = (EROWS * CPR) / XT;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0 && EROWS * ESTR <= XRING * XUNIT, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int bid = blockIdx.x;
  const int qd = a.nblk / 8, rm = a.nblk % 8, xc = bid % 8;
  const int swz = (xc < rm ? xc * (qd + 1) : rm * (qd + 1) + (xc - rm) * qd) + bid / 8;
  const int tm = swz / a.tiles_n, tn = swz % a.tiles_n;
  const int m0 = tm * XBM, n0 = tn * XBN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int grp = wid >> 2;  // the two waves of a SIMD (w, w+4) are in different phase groups
  const int lr = lane & 31, lh = lane >> 5;
  // staging role: DMA instruction u (0/1) of wave wid fills rows u*128 + wid*16 + (lane >> 2), slot lane & 3
  const int srow = wid * 16 + (lane >> 2);
  const int c4 = (lane & 3) ^ ((lane >> 4) & 3);  // logical chunk (of the unit's 4) that belongs in this lane's slot

  int rowoff[AIW];
  unsigned rowmask[AIW];
#pragma unroll
  for (int u = 0; u < AIW; ++u) {
    const int m = m0 + u * 128 + srow;
    rowoff[u] = 0;
    rowmask[u] = 0u;
    if (m < a.M && a.pointwise) {
      rowoff[u] = m * a.ldi;
      rowmask[u] = 0x010101u;
    } else if (m < a.M) {
      const int t1 = (int)fastdiv((uint32_t)m, a.dWo), wo = m - t1 * a.Wo;
      const int t2 = (int)fastdiv((uint32_t)t1, a.dHo), ho = t1 - t2 * a.Ho;
      const int b = (int)fastdiv((uint32_t)t2, a.dTo), to = t2 - b * a.To;
      const int ti0 = to * a.st - a.pt, hi0 = ho * a.sh - a.ph, wi0 = wo * a.sw - a.pw;
      rowoff[u] = (((b * a.T + ti0) * a.H + hi0) * a.W + wi0) * a.ldi;
      unsigned mask = 0u;
      for (int dt = 0; dt < a.KT; ++dt) mask |= ((unsigned)(ti0 + dt) < (unsigned)a.T ? 1u : 0u) << dt;
      for (int dh = 0; dh < a.KH; ++dh) mask |= ((unsigned)(hi0 + dh) < (unsigned)a.H ? 1u : 0u) << (8 + dh);
      for (int dw = 0; dw < a.KW; ++dw) mask |= ((unsigned)(wi0 + dw) < (unsigned)a.W ? 1u : 0u) << (16 + dw);
      rowmask[u] = mask;
    }
  }
  int wrow[BIW];
#pragma unroll
  for (int u = 0; u < BIW; ++u) {
    const int n = n0 + u * 128 + srow;
    wrow[u] = (n < a.Cout && u * 128 + srow < XBN) ? n * a.K : -1;
  }

  f32x16 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... lds): padding taps, rows past M / Cout and the K tail
  // get an offset beyond the descriptor's extent and the hardware range check writes zeros — a 32-bit select per
  // instruction, no 64-bit address arithmetic.  The tap decode of a unit is wave-uniform (Cin % 32 == 0: the unit's 4
  // chunks share a tap), so it runs on the scalar unit; a lane adds its chunk.  Phase A is issue-bound (one wave,
  // ~140 instructions against the other group's 16 MFMAs), so every VALU instruction here is on the critical path.
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
  const int nu = (a.K + 31) / 32;  // pipeline units
  const int swid = __builtin_amdgcn_readfirstlane(wid);  // wave-uniform in an SGPR: LDS destinations stay scalar
  unsigned rowb[AIW], wrowb[BIW];  // byte offsets of this lane's rows (+ its chunk within a unit), OOB when absent
#pragma unroll
  for (int u = 0; u < AIW; ++u) rowb[u] = (unsigned)(rowoff[u] + c4 * 8) * 2u;
#pragma unroll
  for (int u = 0; u < BIW; ++u) wrowb[u] = wrow[u] >= 0 ? (unsigned)(wrow[u] + c4 * 8) * 2u : kOob;
  const bool stage_b = !BHALF || swid < XBN / 16;  // wave-uniform: this wave owns B rows

  // tap decode of unit i: wave-uniform, all on the scalar unit (it is computed one unit ahead, inside the MFMA phase,
  // where scalar issue slots are free; moving the per-lane offset selects there too measured slower: VALU work
  // competes with the MFMA issue, profiles/r01/probe_xl_layers_v7.log)
  // K is walked TAPS INNERMOST (unit i = tap i % ntaps of the 32-channel chunk i / ntaps; the sum over K does not care):
  // a tap-major walk re-reads every input row once per tap, a whole tap phase (32 tiles x 128+ KB per XCD) apart — more
  // than the 4 MB L2 holds, so 26 % of the L2 requests of the [3,1,1] layers missed (profiles/r01/conv_l2_1024.log) and
  // the DMA ring waited on HBM latency.  Taps innermost, the re-reads of a chunk follow each other within a few units.
  unsigned exb_n = 0, ey_n = 0, kb_n = 0;
  int cc_n = 0;
  auto decode = [&](int i) {
    int tap, cc;  // 32-channel chunk of the input, tap
    if (a.tapinner) {
      cc = (int)fastdiv((uint32_t)i, a.dNT);
      tap = i - cc * (int)a.dNT.d;
    } else {
      tap = (int)fastdiv((uint32_t)(i * 4), a.dCpt);
      cc = i - tap * ((int)a.dCpt.d >> 2);
    }
    const int dt = (int)fastdiv((uint32_t)tap, a.dKHW), r2 = tap - dt * (int)a.dKHW.d;
    const int dh = (int)fastdiv((uint32_t)r2, a.dKW), dw = r2 - dh * (int)a.dKW.d;
    exb_n = (unsigned)(((dt * a.H + dh) * a.W + dw) * a.ldi + cc * 32) * 2u;
    ey_n = (1u << dt) | (1u << (8 + dh)) | (1u << (16 + dw));
    kb_n = (unsigned)(tap * (int)a.dCpt.d * 8 + cc * 32) * 2u;  // byte offset of the unit inside a weight row
    cc_n = cc;
  };
  unsigned offa_c[AIW], offb_c[BIW];  // DMA offsets of the unit about to be issued (computed in phase A, issued in phase B)
  auto offsets = [&](int i, unsigned exb, unsigned ey) {
    const bool kin = (cc_n * 4 + c4) < (int)a.dCpt.d && i < nu;  // chunk inside the input channels (K tail / trailing units)
    const unsigned kb = kb_n;
#pragma unroll
    for (int u = 0; u < AIW; ++u) offa_c[u] = (kin && ((rowmask[u] & ey) == ey)) ? rowb[u] + exb : kOob;
#pragma unroll
    for (int u = 0; u < BIW; ++u) offb_c[u] = (kin && wrowb[u] != kOob) ? wrowb[u] + kb : kOob;
  };
  auto issue = [&](int i) {  // this wave's DMA instructions of unit i
    char* st = lds + (i & (XRING - 1)) * XUNIT;
#pragma unroll
    for (int u = 0; u < AIW; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(st + (u * 128 + swid * 16) * 64),
                                               16, (int)offa_c[u], 0, 0, 0);
    if (stage_b) {
#pragma unroll
      for (int u = 0; u < BIW; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rwt, (__attribute__((address_space(3))) void*)(st + XBM * 64 + (u * 128 + swid * 16) * 64), 16, (int)offb_c[u], 0, 0,
            0);
    }
  };
  // counted waits: "all but my youngest `units` units' DMAs have landed"
  auto wait_units = [&](auto units) {
    constexpr int U = decltype(units)::value;
    if (stage_b)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U * (AIW + BIW)) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U * AIW) : "memory");
  };
  const int xa = (lr >> 2) & 3;  // swizzle key of this lane's fragment rows (tile offsets are multiples of 16)
  STAMP_BEGIN();
  for (int i = 0; i < 3; ++i) {  // units past the end of K are all-zero fills (stage_*: kin false): uniform counting
    decode(i);
    offsets(i, exb_n, ey_n);
    issue(i);
  }
#if XL_DECODE_IN_B
  decode(3);
#endif
  wait_units(std::integral_constant<int, 2>{});  // this wave's part of unit 0 (units 1, 2 stay in flight)
  __builtin_amdgcn_s_barrier();
  STAMP(0);  // prologue: first three units issued, unit 0 complete
  // Two wave groups (wm = 0 / 1: the two waves of every SIMD) run ONE PHASE APART: a unit is phase A (12 fragment
  // reads into registers, the 4 DMA instructions of unit i+3, address arithmetic) then phase B (16 MFMAs); while one
  // group multiplies, the other reads, issues and computes addresses on the same SIMD.  Every phase ends in a barrier;
  // the late group takes one extra barrier first, the early group one extra at the end.
  // Hazards: a wave ends phase A(j) with vmcnt(8) — its part of unit j+1 has landed, one or two phases before anyone
  // reads it — and lgkmcnt(0) — its reads of unit j are retired, so the DMA of unit j+4 (next phase A of either group
  // at the earliest) may overwrite the slot.
  if (grp == 1) __builtin_amdgcn_s_barrier();
  for (int i = 0; i < nu; ++i) {
    const char* st = lds + (i & (XRING - 1)) * XUNIT;
    bf16x8 af[2][MT], wf[2][NT];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int koff = ((ks * 2 + lh) ^ xa) * 16;
#pragma unroll
      for (int j = 0; j < MT; ++j)
        af[ks][j] = *reinterpret_cast<const bf16x8*>(st + (wm * WTM + j * 32 + lr) * 64 + koff);
#pragma unroll
      for (int n = 0; n < NT; ++n)
        wf[ks][n] = *reinterpret_cast<const bf16x8*>(st + XBM * 64 + (wn * WTN + n * 32 + lr) * 64 + koff);
    }
    STAMP_FINE(4);  // fragment-read issue
#if !XL_DECODE_IN_B
    decode(i + 3);  // scalar tap decode of the unit issued in the coming phase B (phase A has the slack: stamps)
#endif
    offsets(i + 3, exb_n, ey_n);  // per-lane selects of that unit
    STAMP_FINE(6);  // offset selects (slot 6 is re-used: the epilogue share is lost in this mode)
    // this wave's part of unit i+1 has landed when only unit i+2's 4 DMAs are outstanding (unit i+3 is issued in B)
    wait_units(std::integral_constant<int, 1>{});
    STAMP_FINE(0);  // vmcnt wait (slot 0 re-used)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(1);  // phase A: reads + DMA issue + waits
    __builtin_amdgcn_sched_barrier(0);  // the phases are the schedule: nothing moves across their barriers
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3);  // barrier after A
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[n][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][n], af[ks][j], acc[n][j], 0, 0, 0);  // D[n][m]
    issue(i + 3);  // the DMA of unit i+3 rides under this phase's MFMAs
#if XL_DECODE_IN_B
    decode(i + 4);  // ... and so does the scalar tap decode for the next phase A
#endif
    constexpr int DSTEP = NMF / (AIW + BIW) > 0 ? NMF / (AIW + BIW) : 1;
#pragma unroll
    for (int g = 0; g < NMF; ++g) {  // one MFMA, a few scalar instructions, and every DSTEP-th time one DMA instruction
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x004, XL_DECODE_IN_B ? (80 / NMF) : 2, 0);
      if (g % DSTEP == DSTEP / 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    STAMP(2);  // phase B: MFMAs
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    STAMP(5);  // barrier after B
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing zero-fill units
  __syncthreads();  // all MFMA operand reads done: the ring becomes the epilogue's staging tile

  const bool has_res = a.res != nullptr;
#pragma unroll
  for (int p = 0; p < EPASS; ++p) {
    uint4 rres[EU];
    if (has_res) {
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int c = tid + XT * u;
        const int m = m0 + p * EROWS + c / CPR, n = n0 + (c % CPR) * 8;
        rres[u] = (m < a.M && n < a.Cout) ? *reinterpret_cast<const uint4*>(a.res + (int64_t)m * a.ldr + n)
                                          : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    if ((wm * WTM) / EROWS == p) {  // this wave's rows belong to this pass
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn * WTN + i * 32 + 8 * g + 4 * lh;
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a.bias && n0 + nl < a.Cout) bv = *reinterpret_cast<const float4*>(a.bias + n0 + nl);
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            float v0 = acc[i][j][4 * g + 0] + bv.x, v1 = acc[i][j][4 * g + 1] + bv.y;
            float v2 = acc[i][j][4 * g + 2] + bv.z, v3 = acc[i][j][4 * g + 3] + bv.w;
            if (a.relu && !has_res) {
              v0 = fmaxf(v0, 0.f);
              v1 = fmaxf(v1, 0.f);
              v2 = fmaxf(v2, 0.f);
              v3 = fmaxf(v3, 0.f);
            }
            uint2 pk;
            pk.x = avt::pack_bf16x2(v0, v1);
            pk.y = avt::pack_bf16x2(v2, v3);
            *reinterpret_cast<uint2*>(lds + (wm * WTM - p * EROWS + j * 32 + lr) * ESTR + nl * 2) = pk;
          }
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int c = tid + XT * u;
      const int row = c / CPR, cc = c % CPR;
      const int m = m0 + p * EROWS + row, n = n0 + cc * 8;
      if (m < a.M && n < a.Cout) {
        uint4 v = *reinterpret_cast<const uint4*>(lds + row * ESTR + cc * 16);
        if (has_res) {
          uint32_t* pv = reinterpret_cast<uint32_t*>(&v);
          const uint32_t* pr = reinterpret_cast<const uint32_t*>(&rres[u]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x0 = avt::bf16x2_lo(pv[e]) + avt::bf16x2_lo(pr[e]);
            float x1 = avt::bf16x2_hi(pv[e]) + avt::bf16x2_hi(pr[e]);
            if (a.relu) {
              x0 = fmaxf(x0, 0.f);
              x1 = fmaxf(x1, 0.f);
            }
            pv[e] = avt::pack_bf16x2(x0, x1);
          }
        }
        *reinterpret_cast<uint4*>(a.out + (int64_t)out_row(a, m) * a.ldo + n) = v;
      }
    }
    if (p + 1 < EPASS) __syncthreads();
  }
  STAMP_END();
}

// ---------------------------------------------------------------------------------------------------------------
// XB tile: the XL tile with the WEIGHT operand bypassing the LDS.  What caps the XL tile is the price of its LDS-DMA
// pieces — 4 per wave per 16 MFMAs (profiles/r01/mfma_peak.log: that mix sustains 1.33 PFLOP/s, 8 reads + 2 pieces
// 1.62).  Here only the activations go through the DMA ring (2 pieces per wave and unit, 16 KB units); the weights are
// pre-packed on the host in MFMA-fragment order along the K walk, so a wave fetches its four fragments of a unit with
// four fully coalesced 1 KB loads straight into registers, three units ahead (a ring of four register sets; the loop is
// unrolled by four so every index is static).  Same two-phase wave groups, same epilogue.
template <bool ISSUE_A>  // the DMA pieces of unit i+3 are issued in the read phase (true) or under the MFMAs
__global__ __launch_bounds__(XT, 2) void conv_xb_kernel(ConvArgs a) {
  constexpr int XBM = 256, XBN = 256, WN = 4, WTM = 128, WTN = 64, MT = 4, NT = 2;
  constexpr int XUNIT = XBM * 64;  // one 32-wide K slice of the activations
  constexpr int XR = 4;
  constexpr int AIW = 2, BLD = 2 * NT, UOPS = AIW + BLD;  // vector-memory operations per wave and unit
  constexpr int NMF = 2 * NT * MT;
  constexpr int ESTR = XBN * 2 + 16;
  constexpr int EPASS = 2, EROWS = XBM / EPASS, CPR = XBN / 8, EU = (EROWS * CPR) / XT;
  extern __shared__ __attribute__((aligned(16))) char lds[];  // max(XR * XUNIT, EROWS * ESTR) bytes

  const int bid = blockIdx.x;
  const int qd = a.nblk / 8, rm = a.nblk % 8, xc = bid % 8;
  const int swz = (xc < rm ? xc * (qd + 1) : rm * (qd + 1) + (xc - rm) * qd) + bid / 8;
  const int tm = swz / a.tiles_n, tn = swz % a.tiles_n;
  const int m0 = tm * XBM, n0 = tn * XBN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int grp = wid >> 2;
  const int lr = lane & 31, lh = lane >> 5;
  const int srow = wid * 16 + (lane >> 2);
  const int c4 = (lane & 3) ^ ((lane >> 4) & 3);

  unsigned rowb[AIW], rowmask[AIW];
#pragma unroll
  for (int u = 0; u < AIW; ++u) {
    const int m = m0 + u * 128 + srow;
    int ro = 0;
    unsigned mask = 0u;
    if (m < a.M && a.pointwise) {
      ro = m * a.ldi;
      mask = 0x010101u;
    } else if (m < a.M) {
      const int t1 = (int)fastdiv((uint32_t)m, a.dWo), wo = m - t1 * a.Wo;
      const int t2 = (int)fastdiv((uint32_t)t1, a.dHo), ho = t1 - t2 * a.Ho;
      const int b = (int)fastdiv((uint32_t)t2, a.dTo), to = t2 - b * a.To;
      const int ti0 = to * a.st - a.pt, hi0 = ho * a.sh - a.ph, wi0 = wo * a.sw - a.pw;
      ro = (((b * a.T + ti0) * a.H + hi0) * a.W + wi0) * a.ldi;
      for (int dt = 0; dt < a.KT; ++dt) mask |= ((unsigned)(ti0 + dt) < (unsigned)a.T ? 1u : 0u) << dt;
      for (int dh = 0; dh < a.KH; ++dh) mask |= ((unsigned)(hi0 + dh) < (unsigned)a.H ? 1u : 0u) << (8 + dh);
      for (int dw = 0; dw < a.KW; ++dw) mask |= ((unsigned)(wi0 + dw) < (unsigned)a.W ? 1u : 0u) << (16 + dw);
    }
    rowb[u] = (unsigned)(ro + c4 * 8) * 2u;
    rowmask[u] = mask;
  }

  f32x16 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwf = __builtin_amdgcn_make_buffer_rsrc((void*)a.wfrag, 0, a.wf_bytes, 0x00020000);
  const int nu = (a.K + 31) / 32;
  const int swid = __builtin_amdgcn_readfirstlane(wid);
  // this wave's weight fragments of unit i: tile (n0/32 + 2 wn + n), k-slice ks -> byte ((tile * nup + i) * 2 + ks) * 1024
  const unsigned wfb = (unsigned)(((n0 >> 5) + wn * NT) * a.nup) * 2048u + (unsigned)lane * 16u;
  const unsigned wft = (unsigned)a.nup * 2048u;  // next 32-row tile

  unsigned exb_n = 0, ey_n = 0;
  int cc_n = 0;
  auto decode = [&](int i) {  // taps innermost (the host packs the weight fragments along the same walk)
    const int cc = (int)fastdiv((uint32_t)i, a.dNT), tap = i - cc * (int)a.dNT.d;
    const int dt = (int)fastdiv((uint32_t)tap, a.dKHW), r2 = tap - dt * (int)a.dKHW.d;
    const int dh = (int)fastdiv((uint32_t)r2, a.dKW), dw = r2 - dh * (int)a.dKW.d;
    exb_n = (unsigned)(((dt * a.H + dh) * a.W + dw) * a.ldi + cc * 32) * 2u;
    ey_n = (1u << dt) | (1u << (8 + dh)) | (1u << (16 + dw));
    cc_n = cc;
  };
  unsigned offa_c[AIW];
  auto offsets = [&](int i, unsigned exb, unsigned ey) {
    const bool kin = (cc_n * 4 + c4) < (int)a.dCpt.d && i < nu;
#pragma unroll
    for (int u = 0; u < AIW; ++u) offa_c[u] = (kin && ((rowmask[u] & ey) == ey)) ? rowb[u] + exb : kOob;
  };
  auto issue = [&](int slot) {
    char* st = lds + slot * XUNIT;
#pragma unroll
    for (int u = 0; u < AIW; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(st + (u * 128 + swid * 16) * 64),
                                               16, (int)offa_c[u], 0, 0, 0);
  };
  bf16x8 wq[XR][2][NT];  // weight fragments of units i .. i+3
  auto loadb = [&](auto sc, int i) {
    constexpr int S = decltype(sc)::value;
    const unsigned ub = wfb + (unsigned)i * 2048u;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        wq[S][ks][n] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rwf, (int)(ub + n * wft + ks * 1024u), 0, 0));
  };
  const int xa = (lr >> 2) & 3;
  STAMP_BEGIN();

  // One unit with every ring index static.  Phase A: fragment reads of the activations, offsets of unit i+3, waits;
  // phase B: 16 MFMAs with the DMA pieces and weight-fragment loads of unit i+3 and the scalar decode in their shadow.
  auto body = [&](auto sc, int i) {
    constexpr int S = decltype(sc)::value;
    const char* st = lds + S * XUNIT;
    bf16x8 af[2][MT];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int koff = ((ks * 2 + lh) ^ xa) * 16;
#pragma unroll
      for (int j = 0; j < MT; ++j) af[ks][j] = *reinterpret_cast<const bf16x8*>(st + (wm * WTM + j * 32 + lr) * 64 + koff);
    }
    offsets(i + 3, exb_n, ey_n);
    if constexpr (ISSUE_A) {
      // the two DMA pieces of unit i+3 are issued HERE (a piece costs 60-185 issue cycles: in the read phase they overlap
      // the counted wait instead of stretching the MFMA phase); unit i+2's operations + these two stay outstanding
      issue((S + 3) & (XR - 1));
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UOPS + AIW) : "memory");
    } else {
      // my part of unit i+1 (pieces and fragments) is done when only unit i+2's operations are outstanding
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UOPS) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(1);  // phase A: reads, offsets, waits
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3);  // barrier after A
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[n][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[S][ks][n], af[ks][j], acc[n][j], 0, 0, 0);  // D[n][m]
    if constexpr (!ISSUE_A) issue((S + 3) & (XR - 1));
    loadb(std::integral_constant<int, (S + 3) & (XR - 1)>{}, i + 3);
    decode(i + 4);
#pragma unroll
    for (int g = 0; g < NMF; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);
      if (g % 2 == 1 && g < 2 * (ISSUE_A ? BLD : UOPS)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    STAMP(2);  // phase B: MFMAs + DMA pieces + fragment loads + decode
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    STAMP(5);  // barrier after B
  };

  decode(0);
  offsets(0, exb_n, ey_n);
  issue(0);
  loadb(std::integral_constant<int, 0>{}, 0);
  decode(1);
  offsets(1, exb_n, ey_n);
  issue(1);
  loadb(std::integral_constant<int, 1>{}, 1);
  decode(2);
  offsets(2, exb_n, ey_n);
  issue(2);
  loadb(std::integral_constant<int, 2>{}, 2);
  decode(3);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * UOPS) : "memory");  // my part of unit 0
  __builtin_amdgcn_s_barrier();
  STAMP(0);  // prologue
  if (grp == 1) __builtin_amdgcn_s_barrier();
  for (int i = 0; i < nu; i += 4) {  // (a K that is not a multiple of 128 multiplies up to three all-zero units)
    body(std::integral_constant<int, 0>{}, i);
    body(std::integral_constant<int, 1>{}, i + 1);
    body(std::integral_constant<int, 2>{}, i + 2);
    body(std::integral_constant<int, 3>{}, i + 3);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const bool has_res = a.res != nullptr;
#pragma unroll 1
  for (int p = 0; p < EPASS; ++p) {
    uint4 rres[EU];
    if (has_res) {
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int c = tid + XT * u;
        const int m = m0 + p * EROWS + c / CPR, n = n0 + (c % CPR) * 8;
        rres[u] = (m < a.M && n < a.Cout) ? *reinterpret_cast<const uint4*>(a.res + (int64_t)m * a.ldr + n)
                                          : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    if (wm == p) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn * WTN + i * 32 + 8 * g + 4 * lh;
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a.bias && n0 + nl < a.Cout) bv = *reinterpret_cast<const float4*>(a.bias + n0 + nl);
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            float v0 = acc[i][j][4 * g + 0] + bv.x, v1 = acc[i][j][4 * g + 1] + bv.y;
            float v2 = acc[i][j][4 * g + 2] + bv.z, v3 = acc[i][j][4 * g + 3] + bv.w;
            if (a.relu && !has_res) {
              v0 = fmaxf(v0, 0.f);
              v1 = fmaxf(v1, 0.f);
              v2 = fmaxf(v2, 0.f);
              v3 = fmaxf(v3, 0.f);
            }
            uint2 pk;
            pk.x = avt::pack_bf16x2(v0, v1);
            pk.y = avt::pack_bf16x2(v2, v3);
            *reinterpret_cast<uint2*>(lds + (j * 32 + lr) * ESTR + nl * 2) = pk;
          }
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int c = tid + XT * u;
      const int row = c / CPR, cc = c % CPR;
      const int m = m0 + p * EROWS + row, n = n0 + cc * 8;
      if (m < a.M && n < a.Cout) {
        uint4 v = *reinterpret_cast<const uint4*>(lds + row * ESTR + cc * 16);
        if (has_res) {
          uint32_t* pv = reinterpret_cast<uint32_t*>(&v);
          const uint32_t* pr = reinterpret_cast<const uint32_t*>(&rres[u]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x0 = avt::bf16x2_lo(pv[e]) + avt::bf16x2_lo(pr[e]);
            float x1 = avt::bf16x2_hi(pv[e]) + avt::bf16x2_hi(pr[e]);
            if (a.relu) {
              x0 = fmaxf(x0, 0.f);
              x1 = fmaxf(x1, 0.f);
            }
            pv[e] = avt::pack_bf16x2(x0, x1);
          }
        }
        *reinterpret_cast<uint4*>(a.out + (int64_t)out_row(a, m) * a.ldo + n) = v;
      }
    }
    __syncthreads();
  }
  STAMP_END();
}

int launch_xb(ConvArgs& a, hipStream_t st) {
  const int tiles_m = (a.M + 255) / 256;
  a.tiles_n = (a.Cout + 255) / 256;
  a.nblk = tiles_m * a.tiles_n;
  a.dCpt = make_fastdiv((uint32_t)(a.K / (a.KT * a.KH * a.KW) / 8));
  a.dKHW = make_fastdiv((uint32_t)(a.KH * a.KW));
  a.dKW = make_fastdiv((uint32_t)a.KW);
  a.dNT = make_fastdiv((uint32_t)(a.KT * a.KH * a.KW));
  a.tapinner = 1;
  // (conv_xb_kernel<true>: the activation DMA issued in the fragment-read phase, +1-5 % per layer, profiles/r01/probe_ab_chain.log)
  constexpr int lds_bytes = 128 * (256 * 2 + 16) > 4 * 256 * 64 ? 128 * (256 * 2 + 16) : 4 * 256 * 64;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xb_kernel<true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("avt_conv3d_igemm_bf16: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((conv_xb_kernel<true>), dim3((unsigned)a.nblk), dim3(XT), lds_bytes, st, a);
  return avt::check_launch("avt_conv3d_igemm_bf16");
}


template <int XBM, int XBN, int WM>
int launch_xl(ConvArgs& a, hipStream_t st) {
  const int tiles_m = (a.M + XBM - 1) / XBM;
  a.tiles_n = (a.Cout + XBN - 1) / XBN;
  a.nblk = tiles_m * a.tiles_n;
  a.dCpt = make_fastdiv((uint32_t)(a.K / (a.KT * a.KH * a.KW) / 8));
  a.dKHW = make_fastdiv((uint32_t)(a.KH * a.KW));
  a.dKW = make_fastdiv((uint32_t)a.KW);
  a.dNT = make_fastdiv((uint32_t)(a.KT * a.KH * a.KW));
  a.tapinner = a.KT * a.KH * a.KW > 1 ? 1 : 0;  // taps innermost: a chunk's per-tap re-reads meet in L2 (+3-5 %, round 1)
  constexpr int lds_bytes = XRING * (XBM + XBN) * 64;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xl_kernel<XBM, XBN, WM>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("avt_conv3d_igemm_bf16: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((conv_xl_kernel<XBM, XBN, WM>), dim3((unsigned)a.nblk), dim3(XT), lds_bytes, st, a);
  return avt::check_launch("avt_conv3d_igemm_bf16");
}

}  // namespace

extern "C" int avt_conv3d_ktab(int cin, int kt, int kh, int kw, int h, int w, int ldi, int32_t* ktab, int n_entries) {
  AVT_REQUIRE(ktab && cin > 0 && cin % 8 == 0, "avt_conv3d_ktab: Cin must be a positive multiple of 8");
  AVT_REQUIRE(kt > 0 && kh > 0 && kw > 0 && kt <= 8 && kh <= 8 && kw <= 8, "avt_conv3d_ktab: kernel extents must be 1..8");
  const int K = kt * kh * kw * cin;
  const int nk = (K + BK - 1) / BK;
  AVT_REQUIRE(n_entries == nk * 8 + 2, "avt_conv3d_ktab: n_entries must be %d", nk * 8 + 2);
  ktab[2 * (nk * 8)] = ktab[2 * (nk * 8) + 1] = ktab[2 * (nk * 8) + 2] = ktab[2 * (nk * 8) + 3] = 0;  // the zero chunk
  const int cpt = cin / 8;
  for (int kc = 0; kc < nk * 8; ++kc) {
    if (kc * 8 < K) {
      const int tap = kc / cpt, c8 = kc % cpt;
      const int dt = tap / (kh * kw), dh = (tap / kw) % kh, dw = tap % kw;
      ktab[2 * kc] = ((dt * h + dh) * w + dw) * ldi + c8 * 8;
      ktab[2 * kc + 1] = (1 << dt) | (1 << (8 + dh)) | (1 << (16 + dw));
    } else {
      ktab[2 * kc] = 0;
      ktab[2 * kc + 1] = -1;  // bit 31 (and every other bit) set: matches no row mask
    }
  }
  return AVT_OK;
}

extern "C" int avt_conv3d_igemm_bf16(const void* in, const void* wt, const float* bias, const void* res, void* out,
                                     const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt,
                                     int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho,
                                     int wo, int ldi, int ldo, int ldr, int relu, void* stream) {
  return avt_conv3d_igemm_rows_bf16(in, wt, bias, res, out, ktab, batch, t, h, w, cin, cout, kt, kh, kw, st, sh, sw, pt, ph,
                                    pw, to, ho, wo, ldi, ldo, ldr, relu, 1, 0, 0, stream);
}

extern "C" int avt_conv3d_igemm_rows_bf16(const void* in, const void* wt, const float* bias, const void* res, void* out,
                                          const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt,
                                          int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho,
                                          int wo, int ldi, int ldo, int ldr, int relu, int out_row_stride, int out_h,
                                          int out_w, void* stream) {
  return avt_conv3d_igemm_wfrag_bf16(in, wt, bias, res, out, ktab, batch, t, h, w, cin, cout, kt, kh, kw, st, sh, sw, pt, ph,
                                     pw, to, ho, wo, ldi, ldo, ldr, relu, out_row_stride, out_h, out_w, nullptr, 0, stream);
}

extern "C" int avt_conv3d_igemm_wfrag_supported(int cin, int cout, int kt, int kh, int kw) {
  const int taps = kt * kh * kw;
  constexpr int min_nk = 16;  // shortest K loop (64-wide steps) the XB tile is used for
  return (cout >= 256 && (taps * cin + 63) / 64 >= min_nk && cin % 32 == 0) ? 1 : 0;
}

extern "C" int avt_conv3d_igemm_wfrag_bf16(const void* in, const void* wt, const float* bias, const void* res, void* out,
                                           const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt,
                                           int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho,
                                           int wo, int ldi, int ldo, int ldr, int relu, int out_row_stride, int out_h,
                                           int out_w, const void* wfrag, int nup, void* stream) {
  ConvArgs a;
  const int rc = conv_args_fill(a, "avt_conv3d_igemm_bf16", in, wt, bias, res, out, ktab, batch, t, h, w, cin, cout, kt, kh, kw,
                                st, sh, sw, pt, ph, pw, to, ho, wo, ldi, ldo, ldr, relu, out_row_stride, out_h, out_w);
  if (rc != AVT_OK) return rc;
  a.wfrag = static_cast<const uint16_t*>(wfrag);
  a.nup = nup;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (wfrag) {
    const int nu = (a.K + 31) / 32;
    AVT_REQUIRE(avt_conv3d_igemm_wfrag_supported(cin, cout, kt, kh, kw) && avt::aligned16(wfrag) && nup >= ((nu + 3) / 4) * 4 + 3,
                "avt_conv3d_igemm_wfrag_bf16: fragment-order weights need Cout >= 256, a long K, Cin %% 32 == 0 and nup >= %d units",
                ((nu + 3) / 4) * 4 + 3);
    const int64_t wfb = (int64_t)((cout + 255) / 256) * 8 * nup * 2048;
    AVT_REQUIRE(wfb < (1ll << 31), "avt_conv3d_igemm_wfrag_bf16: fragment array too large for 32-bit offsets");
    a.wf_bytes = (unsigned)wfb;
    return launch_xb(a, s);
  }
  // GEMM-like layers: the 256x256 LDS-DMA tile for K loops of 16 or more 64-wide steps (shorter ones — the 1x1x1 + residual
  // layers — are faster on the 128x128 tile).  A unit of 32 K must not straddle two taps: one tap, or Cin % 32 == 0.
  if (cout >= 256 && a.nk >= 16 && (cin % 32 == 0 || kt * kh * kw == 1)) return launch_xl<256, 256, 2>(a, s);
  if (cout <= 32) return launch<256, 32, 64>(a, s);
  if (cout <= 64) return launch<256, 64, 64>(a, s);
  // (a 256x128 tile, 128x64 per wave, measured 7 % slower end to end: 2 waves/SIMD hide less of the gather, profiles/r01/probe_big_ab.log)
  return launch<128, 128, 64>(a, s);
}
