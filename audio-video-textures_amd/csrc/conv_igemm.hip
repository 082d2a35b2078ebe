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
// the "im2col" is only an address computation (per-row base offset + per-K-chunk tap offset from a small table,
// validity from a per-row bitmask of in-bounds taps) — nothing is materialised.  The MFMA is issued with the
// weights as the first operand (D = Wt * A^T), so each lane ends up holding 4 CONSECUTIVE CHANNELS of one output
// position: the epilogue packs them to bf16 (8 B), stages the tile through LDS, and writes / reads (residual)
// global memory in 16-byte row-contiguous chunks.  BN, ReLU, residual add and the channel-slice write of the
// lateral-fusion concat are all fused here, so every activation tensor crosses HBM once per consumer.
//
// Tiles (256 threads = 4 waves, v_mfma_f32_32x32x16_bf16, fp32 accumulate):
//   <128,128>: 2x2 waves, wave tile 64(m) x 64(n)      — wide layers
//   <256, 64>: 4x1 waves, wave tile 64(m) x 64(n)      — Cout = 64
//   <256, 32>: 4x1 waves, wave tile 64(m) x 32(n)      — fast-pathway layers with few channels
// LDS rows are padded to 144 B (conflict-free ds_read_b128, see sim_gemm.hip).  Roofline: MFMA for the wide
// 3x3 layers, HBM for the 1x1x1 / few-channel layers (bytes = activations in + out (+ residual)).
#include <stdlib.h>

#include "avt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFFF0u;  // byte offset beyond every buffer: the bounds check returns zeros
constexpr int kMaxTabSteps = 128;        // tap table kept in LDS up to this many K-steps (8 KB)

constexpr int LSTR = 144;  // LDS row stride (128 data bytes = 64 bf16 of K, + 16 pad)
constexpr int BK = 64;

// unsigned division by a launch-time constant without the ~40-instruction runtime divide: the row decode of a
// short-K layer (1x1x1, few channels) would otherwise cost more than its whole K loop.
struct FastDiv {
  uint32_t d, magic, shift;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{d, 0u, 0u};
  if (d > 1) {
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.magic = (uint32_t)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
    f.shift = s - 1;
  }
  return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv& f) {
  if (f.d == 1) return n;
  const uint32_t t = __umulhi(n, f.magic);
  return (t + ((n - t) >> 1)) >> f.shift;
}

struct ConvArgs {
  const uint16_t* in;
  const uint16_t* wt;
  const float* bias;
  const uint16_t* res;
  uint16_t* out;
  const int2* ktab;  // [nk*8] {element offset of the chunk's tap+channel relative to the row base, tap bits or -1}
  int T, H, W;       // input extent
  int To, Ho, Wo;
  int Cout, K;
  int KT, KH, KW, st, sh, sw, pt, ph, pw;
  int ldi, ldo, ldr;
  int relu;
  int M;  // output positions (B*To*Ho*Wo)
  int nk;
  int tiles_n, nblk;
  FastDiv dWo, dHo, dTo;
  unsigned in_bytes, wt_bytes;  // extents for the buffer descriptors (hardware range check = free zero fill)
  int pointwise;                // 1x1x1 / stride 1 / pad 0: rows need no decode
};

// waves per SIMD to keep: the register budget the allocator may use follows from it (guide §6 G1)
#define AVT_CONV_MIN_WAVES(BM, BN) ((BM) == 128 ? 3 : ((BN) == 64 ? 2 : 4))

// GLDS = true: operand slabs go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write),
// two LDS stages, ONE barrier per K-step.  The LDS image must then be lane-linear (1 KiB per wave-instruction =
// 8 rows x 128 B, unpadded), so the bank-conflict fix is an XOR swizzle applied on the SOURCE side: the lane that
// fills slot p of row r fetches K-chunk p ^ (r & 7), and fragment reads use the same XOR (guide rule 21).
// Out-of-bounds / K-tail chunks are fetched from 16 zero bytes kept behind the tap table.
// EARLYRES: short-K layers with a residual (1x1x1 + skip connection, HBM-bound) request the residual chunks BEFORE the
// K loop, so a workgroup exposes one memory latency instead of two; costs 32 more VGPRs, hence its own instantiation.
template <int BM, int BN, bool GLDS, bool TABLDS, bool EARLYRES = false>
__global__ __launch_bounds__(256, EARLYRES ? 2 : AVT_CONV_MIN_WAVES(BM, BN)) void conv_igemm_kernel(ConvArgs a) {
  constexpr int RSTR = GLDS ? 128 : LSTR;  // LDS row stride of an operand slab
  constexpr int WAVES_M = BM / 64;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = BN / WAVES_N;  // wave tile width in n
  constexpr int NT = WN / 32, MT = 2;
  constexpr int AU = BM / 32, BU = BN / 32;  // 16-byte chunks per thread per K-step
  constexpr int A_BYTES = BM * RSTR;
  constexpr int STAGE = (BM + BN) * RSTR;  // one K-step of both operands
  constexpr int ESTR = BN * 2 + 16;        // epilogue staging row stride (bytes)
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int bid = blockIdx.x;
  const int qd = a.nblk / 8, rm = a.nblk % 8, xc = bid % 8;
  const int swz = (xc < rm ? xc * (qd + 1) : rm * (qd + 1) + (xc - rm) * qd) + bid / 8;
  const int tm = swz / a.tiles_n, tn = swz % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;
  const int r0 = tid >> 3;
  // K-chunk this thread stages: GLDS fills LDS slot (tid & 7) of its rows with chunk slot ^ (row & 7); every row a
  // thread touches has the same (row & 7) = (tid >> 3) & 7, so the chunk index is a per-thread constant either way
  const int c16 = GLDS ? ((tid & 7) ^ ((tid >> 3) & 7)) : (tid & 7);

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
#pragma unroll
  for (int u = 0; u < BU; ++u) {
    const int n = n0 + r0 + 32 * u;
    wrow[u] = n < a.Cout ? n * a.K : -1;
  }

  f32x16 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  constexpr int CPR = BN / 8;  // 16-byte chunks per output row
  constexpr int EU = (BM * CPR) / 256;
  const bool has_res = a.res != nullptr;
  uint4 rres[EU];
  auto prefetch_res = [&]() {
    if (has_res) {
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int c = tid + 256 * u;
        const int m = m0 + c / CPR, n = n0 + (c % CPR) * 8;
        rres[u] = (m < a.M && n < a.Cout) ? *reinterpret_cast<const uint4*>(a.res + (int64_t)m * a.ldr + n)
                                          : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  if constexpr (EARLYRES) prefetch_res();

  auto compute = [&](const char* st) {
    const int xa = GLDS ? (lr & 7) : 0;  // swizzle key of this lane's fragment rows ((row & 7) = lr & 7)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[MT], wf[NT];
      const int koff = GLDS ? (((ks * 2 + lh) ^ xa) * 16) : (ks * 32 + lh * 16);
#pragma unroll
      for (int j = 0; j < MT; ++j)
        af[j] = *reinterpret_cast<const bf16x8*>(st + (wm * 64 + j * 32 + lr) * RSTR + koff);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        wf[i] = *reinterpret_cast<const bf16x8*>(st + A_BYTES + (wn * WN + i * 32 + lr) * RSTR + koff);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);  // D[n][m]
    }
  };

  if constexpr (GLDS) {
    const uint16_t* zeros = reinterpret_cast<const uint16_t*>(a.ktab + a.nk * 8);  // 16 zero bytes (avt_conv3d_ktab)
    auto stage = [&](int kt, char* st) {
      const int2 e = a.ktab[kt * 8 + c16];
      const bool kin = e.y >= 0;
      const int kc = (kt * 8 + c16) * 8;
#pragma unroll
      for (int u = 0; u < AU; ++u) {
        const bool ok = kin && ((rowmask[u] & (unsigned)e.y) == (unsigned)e.y);
        const uint16_t* src = ok ? a.in + (rowoff[u] + e.x) : zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(st + (32 * u + 8 * wid) * 128), 16, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < BU; ++u) {
        const uint16_t* src = (kin && wrow[u] >= 0) ? a.wt + (wrow[u] + kc) : zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(st + A_BYTES + (32 * u + 8 * wid) * 128),
                                         16, 0, 0);
      }
    };
    stage(0, lds);
    __syncthreads();  // (emits vmcnt(0) while LDS-DMA is in flight)
    for (int kt = 0; kt < a.nk; ++kt) {
      char* cur = lds + (kt & 1) * STAGE;
      if (kt + 1 < a.nk) stage(kt + 1, lds + ((kt + 1) & 1) * STAGE);  // lands while this slab is multiplied
      compute(cur);
      __syncthreads();
    }
  } else {
    // Buffer loads (SRSRC + 32-bit byte offset): padding taps, rows past M and the K tail get an offset beyond the
    // descriptor's extent and the hardware range check returns zeros.  The selects are written as bit arithmetic so
    // the K loop has no branch and no exec masking, and the tap table is read from LDS (filled once per workgroup):
    // a global read of it would put a dependent ~500-cycle L2 round trip in front of every K-step's loads.
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
    int2* ltab = reinterpret_cast<int2*>(lds + STAGE);  // [nk*8] behind the operand slabs (when it fits)
    if constexpr (TABLDS) {
      for (int i = tid; i < a.nk * 8; i += 256) ltab[i] = a.ktab[i];
    }
    unsigned wsel[BU];  // all-ones where the weight row exists
#pragma unroll
    for (int u = 0; u < BU; ++u) wsel[u] = wrow[u] >= 0 ? 0xFFFFFFFFu : 0u;
    i32x4 ra[AU], rb[BU];
    auto gload = [&](int kt) {
      // {offset, tap bits}; chunks past K carry bit 31, which no row mask has -> never "ok"
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
    gload(0);
    lstore();
    __syncthreads();
    for (int kt = 0; kt < a.nk; ++kt) {
      if (kt + 1 < a.nk) gload(kt + 1);
      compute(lds);
      __syncthreads();
      if (kt + 1 < a.nk) {
        lstore();
        __syncthreads();
      }
    }
  }

  // ---- epilogue.  The residual chunks this thread will need are requested before the accumulators are staged
  // through LDS (or, EARLYRES, before the K loop), so their HBM latency is not serialised behind the staging.
  if constexpr (!EARLYRES) prefetch_res();
  // phase 1: (+bias [, relu]) -> bf16 -> LDS staging tile [m][n]
  // D layout: column (lane & 31) = m, rows (reg&3) + 8*(reg>>2) + 4*(lane>>5) = n  -> regs 4g..4g+3 are 4 consecutive n
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
        const int ml = wm * 64 + j * 32 + lr;
        *reinterpret_cast<uint2*>(lds + ml * ESTR + nl * 2) = pk;
      }
    }
  __syncthreads();
  // phase 2: 16-byte row-contiguous chunks: (+residual, relu) -> global
#pragma unroll
  for (int u = 0; u < EU; ++u) {
    const int c = tid + 256 * u;
    const int row = c / CPR, cc = c % CPR;
    const int m = m0 + row, n = n0 + cc * 8;
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
      *reinterpret_cast<uint4*>(a.out + (int64_t)m * a.ldo + n) = v;
    }
  }
}

template <int BM, int BN, bool GLDS, bool TABLDS = true, bool EARLYRES = false>
int launch(ConvArgs& a, hipStream_t st) {
  if (!GLDS && TABLDS && a.nk > kMaxTabSteps) return launch<BM, BN, false, false>(a, st);  // table stays in global memory
  if constexpr (!GLDS && TABLDS && !EARLYRES) {
    static const bool early = []() {
      // measured (profiles/r01/probe_earlyres_ab.log): 3158 vs 3358 clips/s — the 32 extra VGPRs cost a resident
      // workgroup per SIMD, which hurts more than the second latency exposure: OFF unless AVT_CONV_EARLYRES=1
      const char* e = getenv("AVT_CONV_EARLYRES");
      return e ? atoi(e) != 0 : false;
    }();
    if (early && a.res && a.nk <= 2) return launch<BM, BN, false, true, true>(a, st);
  }
  const int tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  a.nblk = tiles_m * a.tiles_n;
  constexpr int lds_main = GLDS ? 2 * (BM + BN) * 128 : (BM + BN) * LSTR + (TABLDS ? kMaxTabSteps * 8 * 8 : 0);
  constexpr int lds_epi = BM * (BN * 2 + 16);
  constexpr int lds_bytes = lds_main > lds_epi ? lds_main : lds_epi;
  if (lds_bytes > 64 * 1024) {  // above the default dynamic-LDS limit: opt in once per kernel
    static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, GLDS, TABLDS, EARLYRES>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
      avt::set_error("avt_conv3d_igemm_bf16: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
      return AVT_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, GLDS, TABLDS, EARLYRES>), dim3((unsigned)a.nblk), dim3(256), lds_bytes, st, a);
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
  AVT_REQUIRE(in && wt && out && ktab, "avt_conv3d_igemm_bf16: NULL pointer");
  AVT_REQUIRE(cin > 0 && cin % 8 == 0 && cout > 0 && cout % 8 == 0, "avt_conv3d_igemm_bf16: Cin/Cout must be multiples of 8");
  AVT_REQUIRE(kt >= 1 && kh >= 1 && kw >= 1 && kt <= 8 && kh <= 8 && kw <= 8, "avt_conv3d_igemm_bf16: kernel extents must be 1..8");
  AVT_REQUIRE(ldi % 8 == 0 && ldo % 8 == 0 && (!res || ldr % 8 == 0) && ldi >= cin && ldo >= cout,
              "avt_conv3d_igemm_bf16: leading dimensions must be multiples of 8 and cover the channels");
  AVT_REQUIRE(avt::aligned16(in) && avt::aligned16(wt) && avt::aligned16(out) && (!res || avt::aligned16(res)) &&
                  (!bias || avt::aligned16(bias)),
              "avt_conv3d_igemm_bf16: pointers must be 16-byte aligned");
  ConvArgs a;
  a.in = static_cast<const uint16_t*>(in);
  a.wt = static_cast<const uint16_t*>(wt);
  a.bias = bias;
  a.res = static_cast<const uint16_t*>(res);
  a.out = static_cast<uint16_t*>(out);
  a.ktab = reinterpret_cast<const int2*>(ktab);
  a.T = t;
  a.H = h;
  a.W = w;
  // output extent: 0 = the symmetric-padding formula; a smaller explicit extent crops the far edge
  // (used by the stem, whose pixel-pair form needs padding 2 on the left and 1 on the right)
  const int fto = (t + 2 * pt - kt) / st + 1, fho = (h + 2 * ph - kh) / sh + 1, fwo = (w + 2 * pw - kw) / sw + 1;
  a.To = to > 0 ? to : fto;
  a.Ho = ho > 0 ? ho : fho;
  a.Wo = wo > 0 ? wo : fwo;
  AVT_REQUIRE(a.To > 0 && a.Ho > 0 && a.Wo > 0 && batch > 0 && a.To <= fto && a.Ho <= fho && a.Wo <= fwo,
              "avt_conv3d_igemm_bf16: bad output extent %dx%dx%d (max %dx%dx%d)", a.To, a.Ho, a.Wo, fto, fho, fwo);
  a.Cout = cout;
  a.K = kt * kh * kw * cin;
  a.KT = kt;
  a.KH = kh;
  a.KW = kw;
  a.st = st;
  a.sh = sh;
  a.sw = sw;
  a.pt = pt;
  a.ph = ph;
  a.pw = pw;
  a.ldi = ldi;
  a.ldo = ldo;
  a.ldr = ldr;
  a.relu = relu;
  const int64_t M = (int64_t)batch * a.To * a.Ho * a.Wo;
  AVT_REQUIRE(M < (1ll << 31) && (int64_t)batch * t * h * w * ldi < (1ll << 31) - 64 && M * (int64_t)ldo < (1ll << 62) &&
                  (int64_t)cout * a.K < (1ll << 31) - 64,
              "avt_conv3d_igemm_bf16: tensor too large for 32-bit offsets");
  a.in_bytes = (unsigned)((int64_t)batch * t * h * w * ldi * 2);
  a.wt_bytes = (unsigned)((int64_t)cout * a.K * 2);
  a.M = (int)M;
  a.nk = (a.K + BK - 1) / BK;
  a.pointwise = (kt == 1 && kh == 1 && kw == 1 && st == 1 && sh == 1 && sw == 1 && pt == 0 && ph == 0 && pw == 0 &&
                 a.To == t && a.Ho == h && a.Wo == w) ? 1 : 0;
  a.dWo = make_fastdiv((uint32_t)a.Wo);
  a.dHo = make_fastdiv((uint32_t)a.Ho);
  a.dTo = make_fastdiv((uint32_t)a.To);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // LDS-DMA staging per tile shape: bit 0 = <128,128>, bit 1 = <256,64>, bit 2 = <256,32> (AVT_CONV_GLDS).
  // Measured on MI355X, fused SlowFast, 64 clips (profiles/r01/probe_glds_ab.log): off 3009 clips/s, bit0 2767,
  // bits0-1 2703, all 2532 -> OFF by default.  This single-stage-ahead form halves the resident workgroups
  // (2 x 32 KB stages) and its unpadded image reads 2-way conflicted; register staging at 3 waves/SIMD wins.
  static const int glds = []() {
    const char* e = getenv("AVT_CONV_GLDS");
    return e ? atoi(e) : 0;
  }();
  if (cout <= 32) return (glds & 4) ? launch<256, 32, true>(a, s) : launch<256, 32, false>(a, s);
  if (cout <= 64) return (glds & 2) ? launch<256, 64, true>(a, s) : launch<256, 64, false>(a, s);
  return (glds & 1) ? launch<128, 128, true>(a, s) : launch<128, 128, false>(a, s);
}
