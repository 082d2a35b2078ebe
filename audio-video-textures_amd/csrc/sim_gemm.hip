// sim_gemm_nt — out[i,j] = <q_i, t_j> / temp on the gfx950 matrix cores.
// Replaces torch.bmm(q, t) + `output /= temp` of the reference operator
// (contrastive_video_textures/models/models.py:416-417; audio :439, :455-457),
// for every query row at once instead of one GEMV per query per chunk.
//
// Both operands are row-major with K contiguous ("NT"), which is exactly the
// MFMA A/B fragment order: lane l of a wave holds 16 contiguous bytes of row
// (l & 31) at k-offset 16*(l >> 5) bytes, for A and B alike.
//
//   AVT_SIM_F32    v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 fmaf chain.
//                  A lane loads 4 consecutive k (one ds_read_b128); MFMA step s
//                  uses element s of both halves, so inside a group of 8 the
//                  chain visits k = 0,4,1,5,2,6,3,7.  That order IS the
//                  canonical definition (oracle/avt_oracle.c) and makes the
//                  whole matrix bit-identical to the CPU oracle.
//   AVT_SIM_BF16   v_mfma_f32_32x32x16_bf16 on the bf16(hi) tables.
//   AVT_SIM_BF16X3 three MFMAs per k-step: hi*hi + hi*lo + lo*hi (~2^-16 rel).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each
// 64x64 = 2x2 MFMA tiles, 64 accumulator VGPRs).  Per K-step every operand
// plane is a [128 rows][128 data bytes] slab staged global -> registers -> LDS
// (the next slab's loads are issued before the MFMAs of the current one), LDS
// rows padded to 144 B so each 16-lane ds_read_b128 group (16 distinct rows at
// one k-offset) touches all 64 banks once.  Roofline: MFMA (2*nq*nt*d flop).
#include "avt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128;
constexpr int ROWB = 128;        // data bytes per row per K-step
constexpr int LSTR = 144;        // LDS row stride in bytes (ROWB + 16 pad)
constexpr int PLANE = 128 * LSTR;  // bytes of one operand plane in LDS

template <int MODE>
struct Cfg {
  static constexpr int ELEM = (MODE == AVT_SIM_F32) ? 4 : 2;
  static constexpr int BK = ROWB / ELEM;                       // elements per K-step
  static constexpr int NPL = (MODE == AVT_SIM_BF16X3) ? 2 : 1;  // planes per operand
};

struct Args {
  const char* q[2];  // plane 0 (f32 / bf16 hi), plane 1 (bf16 lo)
  const char* t[2];
  int64_t nq, nt;
  int d;
  float temp;
  float* out;
  int64_t ldo;
  int tiles_n;
  int nblk;
};

// one 16-byte chunk of a slab: rows beyond `nrows` and k beyond d read as zero
template <int ELEM, bool ALIGNED>
__device__ __forceinline__ uint4 load_chunk(const char* base, int64_t row, int64_t nrows, int d, int k0, int col16) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (row >= nrows) return v;
  constexpr int EPC = 16 / ELEM;  // elements per chunk
  const int k = k0 + col16 * EPC;
  const char* p = base + ((int64_t)row * d + k) * ELEM;
  if (ALIGNED) {
    if (k < d) v = *reinterpret_cast<const uint4*>(p);  // d % EPC == 0: chunk is all in or all out
  } else {
    if (ELEM == 4) {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < d) w[e] = reinterpret_cast<const uint32_t*>(p)[e];
      v = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      uint16_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (k + e < d) h[e] = reinterpret_cast<const uint16_t*>(p)[e];
      v = make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16), h[4] | ((uint32_t)h[5] << 16),
                     h[6] | ((uint32_t)h[7] << 16));
    }
  }
  return v;
}

// TN = columns (target rows) per tile: 128, or 64 (AVT_SIM_F32_TN=64, f32 mode) — half the accumulators (96 + 32 registers
// instead of 96 + 64) let a FOURTH workgroup share a CU, and 4096 x 4096 is then 2048 tiles over 1024 slots = two full
// rounds instead of 1024 over 768 = one round and a third; an output element's k order, hence its bits, does not depend on
// the tile shape.  Measured, f32 mode, fraction of the 157 TFLOP/s peak (128 | 64): N = 2048: 0.59 | 0.60, 4096: 0.69 | 0.66,
// 16384: 0.82 | 0.79 — the narrower tile loses more per tile than the even rounds give back, so 128 stays the default; the
// gap between 4096^2 and 16384^2 is the ramp and the tail of a 0.7 ms launch, not the K loop.
template <int MODE, bool ALIGNED, int TN>
__global__ __launch_bounds__(256) void sim_gemm_kernel(Args a) {
  using C = Cfg<MODE>;
  constexpr int NPL = C::NPL;
  constexpr int NN = TN / 64;         // 32-column MFMA blocks per wave
  constexpr int BU = TN / 32;         // B chunks per thread and plane
  constexpr int BPLANE = TN * LSTR;   // bytes of one B plane
  constexpr int BBASE = NPL * PLANE;  // B planes follow the A planes
  // LDS (dynamic, 2*NPL planes): [A planes][B planes], each 128 rows x 144 B
  extern __shared__ __attribute__((aligned(16))) char lds[];

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so
  // give each XCD a contiguous run of tiles (they share the same Q panel and
  // walk neighbouring T panels -> L2 hits).  Bijective for any nblk.
  const int bid = blockIdx.x;
  const int qd = a.nblk / 8, rm = a.nblk % 8, x = bid % 8;
  const int swz = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + bid / 8;
  const int tm = swz / a.tiles_n, tn = swz % a.tiles_n;
  const int64_t row0 = (int64_t)tm * BM, col0 = (int64_t)tn * TN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[2][NN];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  // staging registers: per plane 4 chunks per thread (1024 chunks / 256 threads)
  uint4 ra[NPL][4], rb[NPL][BU];
  const int nk = (a.d + C::BK - 1) / C::BK;

  // one staging load (piece p of the slab kt: plane, chunk, operand)
  constexpr int PPP = 4 + BU;  // pieces per plane: 4 chunks of A, BU of B
  auto gpiece = [&](int kt, int p) {
    const int k0 = kt * C::BK;
    const int pl = p / PPP, r_ = p % PPP;
    if (r_ < 4) {
      const int c = tid + r_ * 256, r = c >> 3, c16 = c & 7;
      ra[pl][r_] = load_chunk<C::ELEM, ALIGNED>(a.q[pl], row0 + r, a.nq, a.d, k0, c16);
    } else {
      const int u = r_ - 4, c = tid + u * 256, r = c >> 3, c16 = c & 7;
      rb[pl][u] = load_chunk<C::ELEM, ALIGNED>(a.t[pl], col0 + r, a.nt, a.d, k0, c16);
    }
  };
  constexpr int NPIECE = PPP * NPL;
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) gpiece(kt, p);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = tid + u * 256, r = c >> 3, c16 = c & 7;
        *reinterpret_cast<uint4*>(lds + p * PLANE + r * LSTR + c16 * 16) = ra[p][u];
      }
#pragma unroll
      for (int u = 0; u < BU; ++u) {
        const int c = tid + u * 256, r = c >> 3, c16 = c & 7;
        *reinterpret_cast<uint4*>(lds + BBASE + p * BPLANE + r * LSTR + c16 * 16) = rb[p][u];
      }
    }
  };

  // this lane's fragment row addresses (bytes) inside a plane
  const int arow = (wr * 64 + lr) * LSTR + lh * 16;
  const int brow = (wc * (TN / 2) + lr) * LSTR + lh * 16;

  // f32 mode: the next slab's staging loads are issued between the MFMA groups of this slab, not as a burst in front of
  // them (a burst stalls the wave on the CU's vector-memory issue path — 31 % of a workgroup's time in the conv tiles'
  // stamps, profiles/r02/probe_stamps_x3_wide_v1.log): +4.5 % (0.649 -> 0.677 of the f32 peak).  The bf16 modes keep the
  // burst: interleaved, the plain bf16 mode measured 30 % slower (its 4 MFMAs per k-slice leave no room between them).
  auto compute = [&](bool more, int ktn) {
    int pc = 0;
    if (MODE != AVT_SIM_F32 && more) gload(ktn);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // 4 sub-steps of 32 bytes of K per row
      if (MODE == AVT_SIM_F32) {
        float4 fa[2], fb[NN];
#pragma unroll
        for (int m = 0; m < 2; ++m) fa[m] = *reinterpret_cast<const float4*>(lds + arow + m * 32 * LSTR + ks * 32);
#pragma unroll
        for (int n = 0; n < NN; ++n) fb[n] = *reinterpret_cast<const float4*>(lds + BBASE + brow + n * 32 * LSTR + ks * 32);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NN; ++n) {
              const float av = s == 0 ? fa[m].x : s == 1 ? fa[m].y : s == 2 ? fa[m].z : fa[m].w;
              const float bv = s == 0 ? fb[n].x : s == 1 ? fb[n].y : s == 2 ? fb[n].z : fb[n].w;
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][n], 0, 0, 0);
            }
        if (more) {  // the slab's pieces over its first two k-slices: all out within the first half of the MFMAs
          constexpr int HALF = (NPIECE + 1) / 2;
#pragma unroll
          for (int e = 0; e < HALF; ++e)
            if (ks < 2 && ks * HALF + e < NPIECE) gpiece(ktn, ks * HALF + e);
        }
      } else {
        bf16x8 ah[2], bh[NN], al[2], bl[NN];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          ah[m] = *reinterpret_cast<const bf16x8*>(lds + arow + m * 32 * LSTR + ks * 32);
          if (MODE == AVT_SIM_BF16X3) al[m] = *reinterpret_cast<const bf16x8*>(lds + PLANE + arow + m * 32 * LSTR + ks * 32);
        }
#pragma unroll
        for (int n = 0; n < NN; ++n) {
          bh[n] = *reinterpret_cast<const bf16x8*>(lds + BBASE + brow + n * 32 * LSTR + ks * 32);
          if (MODE == AVT_SIM_BF16X3) bl[n] = *reinterpret_cast<const bf16x8*>(lds + BBASE + BPLANE + brow + n * 32 * LSTR + ks * 32);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            if (MODE == AVT_SIM_BF16X3) {
              // small cross terms first, then the dominant hi*hi
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
            }
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
          }
      }
    }
    (void)pc;
  };

  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    compute(kt + 1 < nk, kt + 1);  // the next slab's loads ride between this slab's MFMA groups
    __syncthreads();
    if (kt + 1 < nk) {
      lstore();
      __syncthreads();
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      const int64_t col = col0 + wc * (TN / 2) + n * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < a.nq && col < a.nt) a.out[row * a.ldo + col] = __fdiv_rn(acc[m][n][r], a.temp);
      }
    }
}

// ---- f32 mode, four workgroups per CU -------------------------------------------------------------------------------------
// The 128 x 128 f32 tile above needs 160 registers (3 waves per SIMD): 4096 x 4096 is 1024 tiles over 768 resident slots =
// one round and a third, and the launch pays two rounds (0.62-0.69 of the f32 peak against 0.82 at 16384^2, where the rounds
// are many).  This form halves the K-step (64-byte rows: 16 staging registers instead of 32), fetches with buffer loads
// (32-bit offsets; the hardware range check returns the zeros of rows beyond n and of the k tail: no 64-bit address
// arithmetic, no bounds branches) and fits 128 registers: four workgroups per CU, 1024 slots, ONE round at 4096^2 and two at
// config 4's 2048 x 16384 shard.  The k order of every output element — groups of 8 ascending, 0,4,1,5,2,6,3,7 inside a
// group — is the one of the tile above, so the matrix stays bit-identical to the oracle.
constexpr int LSTR2 = 80;             // 64 data bytes + 16 pad: ds_read_b128 of 16 consecutive rows touches 64 banks once
constexpr int PLANE2 = 128 * LSTR2;
constexpr unsigned kOobS = 0xFFFFFFF0u;

__global__ __launch_bounds__(256, 4) void sim_f32_v2_kernel(Args a, unsigned q_bytes, unsigned t_bytes) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int bid = blockIdx.x;
  const int qd = a.nblk / 8, rm = a.nblk % 8, x = bid % 8;
  const int swz = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + bid / 8;
  const int tm = swz / a.tiles_n, tn = swz % a.tiles_n;
  const int row0 = tm * 128, col0 = tn * 128;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)a.q[0], 0, q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)a.t[0], 0, t_bytes, 0x00020000);
  // this thread's two chunks of each operand: rows r, r + 64 of the tile, 16-byte chunk c4 of the 64-byte row
  const int r = tid >> 2, c4 = tid & 3;
  unsigned qo[2], to_[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int64_t qr = (int64_t)row0 + r + 64 * u, tr = (int64_t)col0 + r + 64 * u;
    qo[u] = qr < a.nq ? (unsigned)((qr * a.d + c4 * 4) * 4) : kOobS;
    to_[u] = tr < a.nt ? (unsigned)((tr * a.d + c4 * 4) * 4) : kOobS;
  }
  const int kc = c4 * 4;  // first k of this thread's chunk inside a K-step
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 ra[2], rb[2];
  auto gpiece = [&](int kt, int p) {
    const int k0 = kt * 16;
    const bool kin = k0 + kc < a.d;  // d % 4 == 0: the chunk is all in or all out
    const int u = p & 1;
    if (p < 2) ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rq, (int)((kin && qo[u] != kOobS) ? qo[u] + (unsigned)k0 * 4u : kOobS), 0, 0);
    else rb[u] = __builtin_amdgcn_raw_buffer_load_b128(rt, (int)((kin && to_[u] != kOobS) ? to_[u] + (unsigned)k0 * 4u : kOobS), 0, 0);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<i32x4*>(lds + (r + 64 * u) * LSTR2 + c4 * 16) = ra[u];
      *reinterpret_cast<i32x4*>(lds + PLANE2 + (r + 64 * u) * LSTR2 + c4 * 16) = rb[u];
    }
  };
  const int arow = (wr * 64 + lr) * LSTR2 + lh * 16;
  const int brow = PLANE2 + (wc * 64 + lr) * LSTR2 + lh * 16;
  const int nk = (a.d + 15) / 16;
#pragma unroll
  for (int p = 0; p < 4; ++p) gpiece(0, p);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float4 fa[2], fb[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) fa[m] = *reinterpret_cast<const float4*>(lds + arow + m * 32 * LSTR2 + ks * 32);
#pragma unroll
      for (int n = 0; n < 2; ++n) fb[n] = *reinterpret_cast<const float4*>(lds + brow + n * 32 * LSTR2 + ks * 32);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const float av = s_ == 0 ? fa[m].x : s_ == 1 ? fa[m].y : s_ == 2 ? fa[m].z : fa[m].w;
            const float bv = s_ == 0 ? fb[n].x : s_ == 1 ? fb[n].y : s_ == 2 ? fb[n].z : fb[n].w;
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][n], 0, 0, 0);
          }
      if (more && ks == 0) {  // the next slab's four loads ride behind the first k-slice's MFMAs
#pragma unroll
        for (int p = 0; p < 4; ++p) gpiece(kt + 1, p);
      }
    }
    __syncthreads();
    if (more) {
      lstore();
      __syncthreads();
    }
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int64_t col = (int64_t)col0 + wc * 64 + n * 32 + lr;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int64_t row = (int64_t)row0 + wr * 64 + m * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
        if (row < a.nq && col < a.nt) a.out[row * a.ldo + col] = __fdiv_rn(acc[m][n][rr], a.temp);
      }
    }
}

int launch_f32_v2(Args& a, hipStream_t st) {
  const int64_t tiles_m = (a.nq + 127) / 128, tiles_n = (a.nt + 127) / 128;
  AVT_REQUIRE(tiles_m * tiles_n < (1ll << 31), "avt_sim_gemm_nt: grid too large");
  a.tiles_n = (int)tiles_n;
  a.nblk = (int)(tiles_m * tiles_n);
  hipLaunchKernelGGL(sim_f32_v2_kernel, dim3((unsigned)a.nblk), dim3(256), 2 * PLANE2, st, a, (unsigned)(a.nq * a.d * 4),
                     (unsigned)(a.nt * a.d * 4));
  return avt::check_launch("avt_sim_gemm_nt");
}

template <int MODE, int TN>
int launch(Args& a, bool aligned, hipStream_t st) {
  const int64_t tiles_m = (a.nq + BM - 1) / BM, tiles_n = (a.nt + TN - 1) / TN;
  AVT_REQUIRE(tiles_m * tiles_n < (1ll << 31), "avt_sim_gemm_nt: grid too large");
  a.tiles_n = (int)tiles_n;
  a.nblk = (int)(tiles_m * tiles_n);
  const dim3 grid((unsigned)a.nblk), block(256);
  constexpr int lds_bytes = Cfg<MODE>::NPL * (PLANE + TN * LSTR);
  auto kern = aligned ? sim_gemm_kernel<MODE, true, TN> : sim_gemm_kernel<MODE, false, TN>;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       lds_bytes);
    if (e != hipSuccess) {
      avt::set_error("avt_sim_gemm_nt: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
      return AVT_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, a);
  return avt::check_launch("avt_sim_gemm_nt");
}

}  // namespace

extern "C" int avt_sim_gemm_nt(const void* q, const void* q_lo, const void* t, const void* t_lo, int64_t nq,
                               int64_t nt, int d, float temp, int precision, float* out, int64_t ldo, void* stream) {
  AVT_REQUIRE(nq >= 0 && nt >= 0 && d > 0, "avt_sim_gemm_nt: bad sizes nq=%lld nt=%lld d=%d", (long long)nq,
              (long long)nt, d);
  if (nq == 0 || nt == 0) return AVT_OK;  // empty matrix (pointers may be NULL)
  AVT_REQUIRE(q && t && out, "avt_sim_gemm_nt: NULL q/t/out");
  AVT_REQUIRE(ldo >= nt, "avt_sim_gemm_nt: ldo < nt");
  AVT_REQUIRE(temp != 0.0f, "avt_sim_gemm_nt: temp == 0");
  AVT_REQUIRE(precision == AVT_SIM_BF16 || precision == AVT_SIM_BF16X3 || precision == AVT_SIM_F32,
              "avt_sim_gemm_nt: unknown precision %d", precision);
  AVT_REQUIRE(precision != AVT_SIM_BF16X3 || (q_lo && t_lo), "avt_sim_gemm_nt: bf16x3 needs q_lo and t_lo");
  if (nq == 0 || nt == 0) return AVT_OK;
  Args a;
  a.q[0] = static_cast<const char*>(q);
  a.q[1] = static_cast<const char*>(q_lo);
  a.t[0] = static_cast<const char*>(t);
  a.t[1] = static_cast<const char*>(t_lo);
  a.nq = nq;
  a.nt = nt;
  a.d = d;
  a.temp = temp;
  a.out = out;
  a.ldo = ldo;
  const int epc = precision == AVT_SIM_F32 ? 4 : 8;
  bool aligned = (d % epc == 0) && avt::aligned16(q) && avt::aligned16(t);
  if (precision == AVT_SIM_BF16X3) aligned = aligned && avt::aligned16(q_lo) && avt::aligned16(t_lo);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (precision) {
    case AVT_SIM_F32: {
      if (aligned && nq * (int64_t)d * 4 < (1ll << 32) - 64 && nt * (int64_t)d * 4 < (1ll << 32) - 64) return launch_f32_v2(a, st);
      return launch<AVT_SIM_F32, 128>(a, aligned, st);  // (unaligned / very large operands: the round-2 tile)
    }
    case AVT_SIM_BF16: return launch<AVT_SIM_BF16, 128>(a, aligned, st);
    default: return launch<AVT_SIM_BF16X3, 128>(a, aligned, st);
  }
}
