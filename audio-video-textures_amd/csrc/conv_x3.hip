// conv3d_igemm_x3 — the CONTRACT-GRADE form of the encoder convolution: split-bf16 ("bf16x3") implicit GEMM.
//
// The reference's encoders compute in fp32 (contrastive_video_textures/models/models.py:335, 399) and its contract on
// this path is "similarity within 1e-3" = 1e-4 on a cosine: plain bf16 activations (2^-9 per element, every layer) are
// two orders of magnitude away from that.  The f32-input MFMA runs at 1/16 of the bf16 rate, so this kernel keeps the
// bf16 matrix pipe and splits every operand in two bf16 planes instead:
//     x = hi + lo,  hi = bf16(x),  lo = bf16(x - hi)            (|x - hi - lo| <= 2^-16 |x|; fp16 planes: 2^-22)
//     sum_k a_k w_k  ~=  sum_k ( wh*ah + wh*al + wl*ah )         (the dropped wl*al term is <= 2^-16 |a w|; fp16: 2^-22)
// three v_mfma_f32_32x32x16_bf16 per (n, m, k) sub-tile into ONE fp32 accumulator: 1/3 of the bf16 MFMA rate, 5x the
// f32 MFMA rate, and ~1e-5 relative per product instead of 4e-3.
//
// Activations travel between layers as TWO bf16 planes of identical geometry (NDHWC rows [M, ld]): the producing
// layer's epilogue splits each fp32 result once, so the K loop of every consumer is the plain bf16 gather of
// conv_igemm.hip done twice (same tap table, same row masks, same SRSRC zero fill) — no conversion in the loop, and a
// 3x3 consumer does not re-split its input nine times.  Bias, residual add (hi + lo of the residual planes, summed in
// fp32), ReLU and the concat-slice write are fused in the epilogue like in the bf16 kernel; the tile is staged through
// LDS in fp32 so the split happens on the final value (one rounding to the pair, not two).
//
// Tiles (256 threads = 4 waves): <128,128,64> wide layers, <128,64,64> Cout <= 64, <128,32,32> Cout <= 32; BK = 64;
// LDS per workgroup 2 x (BM + BN) x 144 B (two planes per operand) + the tap table: 82 / 63 / 54 KB.  Per K-step a wave
// of the wide tile reads 8 fragments per 16-wide k-slice for 12 MFMAs (the bf16 tile: 4 for 4), so this form is much
// closer to MFMA-bound than the bf16 one.  Roofline: MFMA at 1/3 of the bf16 peak (833 TFLOP/s algorithmic).
// Measured and not kept (profiles/r02/probe_x3_wide_pipelined_tile_slower.log, probe_stamps_x3_wide_pipelined.log): a
// software-pipelined form of the wide tile — 32-wide steps in two swizzled LDS buffers, the loads of step kt+2 and the
// ds_writes of step kt+1 interleaved with the MFMA triples of step kt, one barrier per step — was correct but 12 % SLOWER
// on every long-K layer (293 vs 333 TFLOP/s), with the loads issued as a burst or interleaved alike: its MFMA phase
// stretches to 4.4x the matrix time (16 fragment reads exposed at every 24-MFMA step, ds_writes in the MFMA stream).
// Also measured and not kept (profiles/r02/probe_x3_paired_antiphase_tile_slower.log): two tiles per 512-thread workgroup,
// the second group of four waves half a K-step behind the first so that every barrier is shared and one group is in its
// MFMA phase while the other waits for loads / writes LDS (the two independent workgroups of a CU drift INTO phase).
// Correct (42 parity tests) but 5-9 % slower on every long-K layer: one wave per SIMD cannot keep the matrix pipe full by
// itself, so enforcing the alternation trades the free-running form's fine-grained sharing of the pipe for barrier bubbles.
#include <stdlib.h>

#include <type_traits>

#include "avt_common.h"
#include "conv_args.h"
#include "split_planes.h"

#ifdef AVT_CONV_STAMP
// diagnostic build only (`make stamp` -> libavt_hip_stamp.so, never the shipped library): the hooks live in tools/diag
#define AVT_STAMP_FN avt_debug_stamps_x3
namespace {
#include "../../tools/diag/conv_stamp.h"
}
// phase-skip diagnostic of the same build (1 skip the LDS stage, 2 skip the MFMAs, 4 skip the global loads, 8 skip the fragment reads):
// what the kernel's time really hangs on is what it gets faster without (tools/probe_conv_phases.sh)
// — a COMPILE-TIME switch (-DAVT_DBG_CONST=n, one library per setting): read at run time, the branches around the MFMAs
// alone cost 35 %.
#ifndef AVT_DBG_CONST
#define AVT_DBG_CONST 0
#endif
#define DBG_SKIP(bit) (((AVT_DBG_CONST) & (bit)) != 0)
// -DAVT_STAMP_EPI (STAMP_EXTRA of `make stamp`): the XL tile's slots 1..4 time the EPILOGUE's sub-phases instead of the K loop's
// (1 first slab staged + barrier, 2 residual requests + staging of the next slab, 3 LDS reads / split / stores, 4 barrier)
#ifdef AVT_STAMP_EPI
#define KSTAMP(i) STAMP(0)
#define EPI_STAMP(i) STAMP(i)
#else
#define KSTAMP(i) STAMP(i)
#define EPI_STAMP(i)
#endif
#else
#define STAMP_BEGIN()
#define STAMP(i)
#define KSTAMP(i)
#define EPI_STAMP(i)
#define STAMP_END()
#define DBG_SKIP(bit) false
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool F16>
__device__ __forceinline__ f32x16 mfma(i32x4 w, i32x4 x, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}

// IO32: the TRAINING form (train_ops.conv3d: forward and stride-1 dgrad of the SlowFast convolutions, train.py:114-141) —
// activations are read as fp32 NDHWC rows and split into the two planes in registers on their way to the LDS (same number
// of 16-byte loads as two planes; ~4 VALU operations per element next to 48 MFMAs per K-step), the result is written as
// fp32: a drop-in for an fp32 convolution on channels-last tensors, no plane-pair tensors in the autograd graph.
// BST (IO32 only): the backward-statistics epilogue of conv_args.h (a separate instance: its 40 extra live registers stay out of
// the forward / plain input-gradient launches)
template <int BM, int BN, int WTM, bool F16, bool IO32 = false, bool BST = false>
__global__ __launch_bounds__(256, 2) void conv_x3_kernel(ConvArgs a) {
  constexpr int WAVES_M = BM / WTM;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = BN / WAVES_N;
  constexpr int NT = WN / 32, MT = WTM / 32;
  constexpr int AU = BM / 32, BU = BN / 32;  // 16-byte chunks per thread, plane and K-step
  constexpr int A_BYTES = BM * LSTR, B_BYTES = BN * LSTR;
  constexpr int A_LO = A_BYTES, B_HI = 2 * A_BYTES, B_LO = 2 * A_BYTES + B_BYTES;
  constexpr int STAGE = 2 * (A_BYTES + B_BYTES);
  constexpr int ESTR = BN * 4 + 16;  // fp32 epilogue staging row stride (bytes)
  constexpr int CPR = BN / 8;        // 8-channel chunks per output row
  constexpr int EU = (BM * CPR) / 256;
  static_assert(NT >= 1 && MT >= 1 && BM * ESTR <= STAGE + kMaxTabSteps * 64 && EU >= 1, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int swz = avt::xcd_contiguous(blockIdx.x, a.nblk);  // consecutive tiles of one XCD share activation rows
  const int tm = swz / a.tiles_n, tn = swz % a.tiles_n;
  const int n0 = tn * BN;
  int m0, m_end;  // (IO32 with BatchNorm statistics: the tiles are laid per group and end with their group; else 0 .. M)
  if constexpr (IO32) tile_rows(a, tm, BM, m0, m_end);
  else { m0 = tm * BM; m_end = a.M; }
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;
  const int r0 = tid >> 3, c16 = tid & 7;

  int rowoff[AU];
  unsigned rowmask[AU];
#pragma unroll
  for (int u = 0; u < AU; ++u) {
    const int m = m0 + r0 + 32 * u;
    rowoff[u] = 0;
    rowmask[u] = 0u;
    if (m < m_end && a.pointwise) {
      rowoff[u] = m * a.ldi;
      rowmask[u] = 0x010101u;
    } else if (m < m_end) {
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
  int wrow[BU];
  unsigned wsel[BU];
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

  // Fragments are double-buffered in registers: the eight ds_read_b128 of k-slice ks + 1 are issued BEFORE the twelve
  // MFMAs of slice ks, so the matrix pipe never waits on an LDS round trip inside a K-step (with one register set the
  // compiler re-used 24 registers and parked an s_waitcnt lgkmcnt(0) in front of every other MFMA group: the pipe idled
  // ~half of each slice, profiles/r02/conv_x3_pmc_1024_256.log).
  struct Frags {
    i32x4 ah[MT], al[MT], wh[NT], wl[NT];
  };
  auto fload = [&](Frags& f, int ks) {
    if (DBG_SKIP(8)) {  // diagnostic: MFMAs without their LDS fragment reads (operands that depend on ks, so nothing folds)
#pragma unroll
      for (int j = 0; j < MT; ++j) f.ah[j] = f.al[j] = i32x4{ks, lane, j, 0x3c003c00};
#pragma unroll
      for (int i = 0; i < NT; ++i) f.wh[i] = f.wl[i] = i32x4{ks, lane, i, 0x3c003c00};
      return;
    }
    const int koff = ks * 32 + lh * 16;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int o = (wm * WTM + j * 32 + lr) * LSTR + koff;
      f.ah[j] = *reinterpret_cast<const i32x4*>(lds + o);
      f.al[j] = *reinterpret_cast<const i32x4*>(lds + A_LO + o);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int o = (wn * WN + i * 32 + lr) * LSTR + koff;
      f.wh[i] = *reinterpret_cast<const i32x4*>(lds + B_HI + o);
      f.wl[i] = *reinterpret_cast<const i32x4*>(lds + B_LO + o);
    }
  };
  // The next slab's 2 (AU + BU) buffer loads are issued ONE AT A TIME between the MFMA triples of this slab: issued as a
  // burst in front of the MFMAs they stall the wave for ~1800 cycles per K-step on the CU's vector-memory issue path
  // (64 B/clk: 64 KB per workgroup and K-step; in-kernel stamps, profiles/r02/probe_stamps_x3_wide_v1.log: 31 % of a
  // workgroup's time, MFMAs 38 %) — interleaved, that wait hides under the matrix pipe.
  // (two pieces per triple: every load is out within the first half of the slab's MFMAs, so the last one has the second
  // half to land before the stores need it — issued evenly, the wait for the late pieces was 24 % of a workgroup's time)
  constexpr int NPIECE = 2 * (AU + BU), NGROUP = 4 * NT * MT, PPG = 2 * ((NPIECE + NGROUP - 1) / NGROUP);
  // `more` (this is not the tile's last slab: the next slab's loads ride along) is a COMPILE-TIME constant — the K loop below is
  // peeled.  As a run-time flag every load sat in a branch of its own, and the compiler, which cannot know that the branch around
  // the ds_writes consuming the loads is taken exactly when the branches issuing them are, put `s_waitcnt vmcnt(0)` in front of
  // every load pair of the fp32 (IO32) form (its second offset is computed in the load's destination register): four serialized
  // L2 round trips per slab inside the MFMA sequence (round 5, tools/diag/isa_loop_summary.py)
  auto fmul = [&](const Frags& f, int g0, auto more_, auto&& piece) {
    constexpr bool more = decltype(more_)::value;
    // small terms first, the leading product last: D[n][m] += wl*ah + wh*al + wh*ah
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        if (!DBG_SKIP(2)) {
          acc[i][j] = mfma<F16>(f.wl[i], f.ah[j], acc[i][j]);
          acc[i][j] = mfma<F16>(f.wh[i], f.al[j], acc[i][j]);
          acc[i][j] = mfma<F16>(f.wh[i], f.ah[j], acc[i][j]);
        }
        if (more && !DBG_SKIP(4)) {
#pragma unroll
          for (int e = 0; e < PPG; ++e) {
            const int pc = (g0 + i * MT + j) * PPG + e;
            if (pc < NPIECE) piece(pc);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  auto compute = [&](auto more, auto&& prep, auto&& piece) {
    // (sched_barrier: without it the machine scheduler sinks every read back to just before its first use)
    Frags f0, f1;
    fload(f0, 0);
    fload(f1, 1);
    if constexpr (decltype(more)::value) prep();  // the next slab's offsets: its table read joins the fragment reads, its VALU the first MFMAs
    __builtin_amdgcn_sched_barrier(0);
    fmul(f0, 0, more, piece);
    fload(f0, 2);
    __builtin_amdgcn_sched_barrier(0);
    fmul(f1, NT * MT, more, piece);
    fload(f1, 3);
    __builtin_amdgcn_sched_barrier(0);
    fmul(f0, 2 * NT * MT, more, piece);
    fmul(f1, 3 * NT * MT, more, piece);
  };

  const __amdgpu_buffer_rsrc_t rih = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ril = __builtin_amdgcn_make_buffer_rsrc((void*)a.in_lo, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt_lo, 0, a.wt_bytes, 0x00020000);
  int2* ltab = reinterpret_cast<int2*>(lds + STAGE);
  const bool tab_lds = a.nk <= kMaxTabSteps;  // uniform
  if (tab_lds)
    for (int i = tid; i < a.nk * 8; i += 256) ltab[i] = a.ktab[i];
  i32x4 rah[AU], ral[AU], rbh[BU], rbl[BU];
  unsigned aoffs[AU], boffs[BU];
  // (TL: the table is in the LDS — a compile-time constant per instance of the K loop below: read through `tab_lds ? ltab : a.ktab`
  //  the compiler selects the POINTER and emits a flat load, whose use waits for every fragment read issued before it)
  auto gprep = [&](auto tl_, int kt) {  // byte offsets of the slab's chunks (out of range = zero fill)
    int2 e;
    if constexpr (decltype(tl_)::value) e = ltab[kt * 8 + c16];
    else e = a.ktab[kt * 8 + c16];
    const unsigned ebits = (unsigned)e.y;
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const unsigned sel = ((rowmask[u] & ebits) == ebits) ? 0xFFFFFFFFu : 0u;
      aoffs[u] = (((unsigned)(rowoff[u] + e.x) * (IO32 ? 4u : 2u)) & sel) | (kOob & ~sel);
    }
    const unsigned ksel = ~(unsigned)(e.y >> 31);
    const unsigned kc2 = (unsigned)((kt * 8 + c16) * 16);
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const unsigned sel = ksel & wsel[u];
      boffs[u] = (((unsigned)wrow[u] * 2u + kc2) & sel) | (kOob & ~sel);
    }
  };
  auto gpiece = [&](int p) {  // p (a constant after unrolling): piece 2u + plane of A, then of B
    if (p < 2 * AU) {
      const int u = p >> 1;
      if constexpr (IO32) {  // eight consecutive fp32 channels = two 16-byte loads of the SAME tensor
        const unsigned o2 = aoffs[u] == kOob ? kOob : aoffs[u] + 16u;
        if (p & 1) ral[u] = __builtin_amdgcn_raw_buffer_load_b128(rih, (int)o2, 0, 0);
        else rah[u] = __builtin_amdgcn_raw_buffer_load_b128(rih, (int)aoffs[u], 0, 0);
      } else {
        if (p & 1) ral[u] = __builtin_amdgcn_raw_buffer_load_b128(ril, (int)aoffs[u], 0, 0);
        else rah[u] = __builtin_amdgcn_raw_buffer_load_b128(rih, (int)aoffs[u], 0, 0);
      }
    } else {
      const int u = (p - 2 * AU) >> 1;
      if (p & 1) rbl[u] = __builtin_amdgcn_raw_buffer_load_b128(rwl, (int)boffs[u], 0, 0);
      else rbh[u] = __builtin_amdgcn_raw_buffer_load_b128(rwh, (int)boffs[u], 0, 0);
    }
  };
  auto gload = [&](int kt) {
    gprep(std::false_type{}, kt);
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) gpiece(p);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const int o = (r0 + 32 * u) * LSTR + c16 * 16;
      if constexpr (IO32) {
        const float* fa = reinterpret_cast<const float*>(&rah[u]);
        const float* fb = reinterpret_cast<const float*>(&ral[u]);
        uint4 h, l;
        const float f8[8] = {fa[0], fa[1], fa[2], fa[3], fb[0], fb[1], fb[2], fb[3]};
        avt::split8<F16>(f8, h, l);
        *reinterpret_cast<uint4*>(lds + o) = h;
        *reinterpret_cast<uint4*>(lds + A_LO + o) = l;
      } else {
        *reinterpret_cast<i32x4*>(lds + o) = rah[u];
        *reinterpret_cast<i32x4*>(lds + A_LO + o) = ral[u];
      }
    }
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const int o = (r0 + 32 * u) * LSTR + c16 * 16;
      *reinterpret_cast<i32x4*>(lds + B_HI + o) = rbh[u];
      *reinterpret_cast<i32x4*>(lds + B_LO + o) = rbl[u];
    }
  };
  // the tile's bias / scale, once, into LDS behind everything else (the epilogue used to fetch them from global memory one
  // accumulator group at a time: a chain of L2 round trips per tile, each wait also sitting out earlier stores)
  float* cfl = reinterpret_cast<float*>(lds + a.cf_ofs);
  for (int i = tid; i < 2 * BN; i += 256) {
    const int n = n0 + (i < BN ? i : i - BN);
    float v = i < BN ? 0.0f : 1.0f;
    if (n < a.Cout) {
      if (i < BN) { if (a.bias) v = a.bias[n]; } else if (a.wscale) v = a.wscale[n];
    }
    cfl[i] = v;
  }
  __syncthreads();
  STAMP_BEGIN();
  gload(0);
  lstore();
  __syncthreads();
  STAMP(0);  // prologue
  auto kloop = [&](auto tl) {  // every slab but the last: no branch in the body
    for (int kt = 0; kt + 1 < a.nk; ++kt) {
      STAMP(1);
      compute(std::true_type{}, [&]() { gprep(tl, kt + 1); }, gpiece);
      STAMP(2);  // fragment reads + MFMAs + the next slab's loads, one per MFMA triple
      __syncthreads();
      STAMP(3);  // barrier after compute
      if (!DBG_SKIP(1)) lstore();
      STAMP(4);  // wait for the loads + ds_write
      __syncthreads();
      STAMP(5);  // barrier after the stores
    }
  };
  if (tab_lds) kloop(std::true_type{});
  else kloop(std::false_type{});
  STAMP(1);
  compute(std::false_type{}, []() {}, gpiece);  // the last slab
  STAMP(2);
  __syncthreads();
  STAMP(3);

  // ---- epilogue: residual planes requested first, tile staged in fp32, split on the final value
  const bool has_res = !IO32 && a.res != nullptr;  // (IO32: `res` is an fp32 tensor added in the store loop below)
  uint4 rrh[EU], rrl[EU];
  if (has_res) {
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int c = tid + 256 * u;
      const int m = m0 + c / CPR, n = n0 + (c % CPR) * 8;
      const bool ok = m < m_end && n < a.Cout;
      rrh[u] = ok ? *reinterpret_cast<const uint4*>(a.res + (int64_t)m * a.ldr + n) : make_uint4(0u, 0u, 0u, 0u);
      rrl[u] = ok ? *reinterpret_cast<const uint4*>(a.res_lo + (int64_t)m * a.ldr + n) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  if constexpr (IO32) {
    if (a.res) {  // the fp32 `add` operand (train_ops.conv3d_fork), requested here in one go: fetched chunk by chunk in the store loop,
                  // every chunk waited (vmcnt(0)) for its own two loads AND the previous chunk's stores — eight round trips per tile
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int c = tid + 256 * u;
        const int m = m0 + c / CPR, n = n0 + (c % CPR) * 8;
        const bool ok = m < m_end && n < a.Cout;
        const float* rf = reinterpret_cast<const float*>(a.res) + (int64_t)m * a.ldr + n;
        rrh[u] = ok ? *reinterpret_cast<const uint4*>(rf) : make_uint4(0u, 0u, 0u, 0u);
        rrl[u] = ok ? *reinterpret_cast<const uint4*>(rf + 4) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
  // backward statistics of the BatchNorm whose output gradient this launch computes (conv_args.h): its input's octets for the rows
  // this thread stores, the mask nibbles and the octet's coefficients — requested here, used in the store loop
  float4 bxa[BST ? EU : 1], bxb[BST ? EU : 1];
  unsigned bbits[BST ? EU : 1];
  BstCoef bk;
  if constexpr (BST) {
    {
      const int n = n0 + (tid % CPR) * 8;
      bst_load(a, tm / a.stat_tpg, n < a.Cout ? n : 0, bk);
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int c = tid + 256 * u;
        const int m = m0 + c / CPR;
        const bool ok = m < m_end && n < a.Cout;
        const int64_t o = ok ? (int64_t)m * a.ldo + n : 0;
        bxa[u] = *reinterpret_cast<const float4*>(a.bst_x + o);
        bxb[u] = *reinterpret_cast<const float4*>(a.bst_x + o + 4);
        bbits[u] = a.bst_mask ? *reinterpret_cast<const uint16_t*>(a.bst_mask + (o >> 2)) : 0u;
        bbits[u] = (bbits[u] & 0xFu) | ((bbits[u] >> 4) & 0xF0u);
      }
    }
  }
  // D layout: column (lane & 31) = m, rows (reg&3) + 8*(reg>>2) + 4*(lane>>5) = n -> regs 4g..4g+3 are 4 consecutive n
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = wn * WN + i * 32 + 8 * g + 4 * lh;
      const float4 bv = *reinterpret_cast<const float4*>(cfl + nl);
      const float4 sv = *reinterpret_cast<const float4*>(cfl + BN + nl);  // exact powers of two
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        float4 v;
        v.x = acc[i][j][4 * g + 0] * sv.x + bv.x;
        v.y = acc[i][j][4 * g + 1] * sv.y + bv.y;
        v.z = acc[i][j][4 * g + 2] * sv.z + bv.z;
        v.w = acc[i][j][4 * g + 3] * sv.w + bv.w;
        const int ml = wm * WTM + j * 32 + lr;
        *reinterpret_cast<float4*>(lds + ml * ESTR + nl * 4) = v;
      }
    }
  __syncthreads();
  // BatchNorm statistics of the rows this thread stores (its 8 channels are the same in every row: 256 % CPR == 0)
  float st_s[IO32 ? 8 : 1], st_q[IO32 ? 8 : 1];
  if constexpr (IO32) {
#pragma unroll
    for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.0f;
  }
#pragma unroll
  for (int u = 0; u < EU; ++u) {
    const int c = tid + 256 * u;
    const int row = c / CPR, cc = c % CPR;
    const int m = m0 + row, n = n0 + cc * 8;
    if (m < m_end && n < a.Cout) {
      const float4 v0 = *reinterpret_cast<const float4*>(lds + row * ESTR + cc * 32);
      const float4 v1 = *reinterpret_cast<const float4*>(lds + row * ESTR + cc * 32 + 16);
      float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if (has_res) {
        const uint32_t* ph = reinterpret_cast<const uint32_t*>(&rrh[u]);
        const uint32_t* pl = reinterpret_cast<const uint32_t*>(&rrl[u]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const avt::f32x2 r = avt::join2<F16>(ph[e], pl[e]);
          x[2 * e] += r.x;
          x[2 * e + 1] += r.y;
        }
      }
      if (a.relu == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = avt::relu_keep_nan(x[e]);
      } else if (a.relu == 2) {  // LeakyReLU(0.1): the SuperSloMo UNets (models/slowmo.py:69-71, 132-134, 195-207)
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.1f * x[e]);
      }
      const int64_t o = (int64_t)out_row(a, m) * a.ldo + n;
      if constexpr (IO32) {
        if (a.res) {  // out = conv + add: a gradient that reaches the same tensor by another path, summed here instead of
                      // in a separate pass (train_ops.conv3d_fork)
          const float4 r0_ = __builtin_bit_cast(float4, rrh[u]), r1_ = __builtin_bit_cast(float4, rrl[u]);
          x[0] += r0_.x; x[1] += r0_.y; x[2] += r0_.z; x[3] += r0_.w;
          x[4] += r1_.x; x[5] += r1_.y; x[6] += r1_.z; x[7] += r1_.w;
        }
        if constexpr (BST) {  // backward statistics: x becomes g = mask * dz; sums of g and g * xhat
          const float xv[8] = {bxa[u].x, bxa[u].y, bxa[u].z, bxa[u].w, bxb[u].x, bxb[u].y, bxb[u].z, bxb[u].w};
          bst_apply(a, bk, xv, bbits[u], x, st_s, st_q);
        } else if (a.stat_part) {  // (uniform) forward statistics of the values as stored: what the BatchNorm's own pass would read back
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            st_s[e] += x[e];
            st_q[e] += x[e] * x[e];
          }
        }
        float* of = reinterpret_cast<float*>(a.out) + o;
        // (whole 128-byte lines of a tensor far larger than the caches: non-temporal stores, +0.8 % on the training step)
        typedef float f32x4n __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f32x4n{x[0], x[1], x[2], x[3]}, reinterpret_cast<f32x4n*>(of));
        __builtin_nontemporal_store(f32x4n{x[4], x[5], x[6], x[7]}, reinterpret_cast<f32x4n*>(of + 4));
      } else {
        uint4 oh, ol;
        avt::split8<F16>(x, oh, ol);
        *reinterpret_cast<uint4*>(a.out + o) = oh;
        *reinterpret_cast<uint4*>(a.out_lo + o) = ol;
      }
    }
  }
  if constexpr (IO32) {
    if (a.stat_part) {  // per-thread sums (fp32 over <= 8 rows) -> LDS [thread-row][sum | squares][column] -> fp64 fold -> the tile's row
      __syncthreads();  // (the staging rows have been read)
      float* red = reinterpret_cast<float*>(lds);
      const int tr = tid / CPR, cc = tid % CPR;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(tr * 2 + 0) * BN + cc * 8 + e] = st_s[e];
        red[(tr * 2 + 1) * BN + cc * 8 + e] = st_q[e];
      }
      __syncthreads();
      stat_fold_store<BN, 256>(a, red, 256 / CPR, tm, n0, tid);
    }
  }
  STAMP_END();
}

// ---- the XL tile: 256 x 256 outputs per workgroup, 8 waves, for the long-K layers with Cout % 256 == 0 ----------------------
// What bounds the 128 x 128 tile on those layers is the operand traffic from L2 into the LDS, not the matrix pipe: the
// phase-skip build (profiles/r02/probe_conv_x3_phases.log; 1024 -> 256 [3,1,1] at 64 clips) runs 349 us with the MFMAs
// compiled out — 64 KB per workgroup and 64-wide K-step, 4.8 GB per launch at 13.8 TB/s — against 250-270 us of matrix time,
// and the two do not hide under each other.  A 256 x 256 tile moves HALF the bytes per flop (the activation rows are not
// fetched once per 128-wide column block, the weight rows not once per 128-row block), which puts the load pipeline (~175 us)
// under the matrix time.  Structure: 32-wide K-steps, TWO LDS stages of 64 KB (rows of 64 bytes, unpadded, 16-byte chunks
// XOR-swizzled by (row >> 2) & 3: conflict-free ds_read_b128 / ds_write_b128), ONE barrier per step — the next step's
// operands arrive by LDS-DMA (buffer_load ... lds, hardware zero fill, the swizzle applied on the source side), one piece
// per MFMA triple, straight into the other stage — no staging registers, no ds_write phase, which leaves room for the
// fragments of BOTH k-slices of a step in registers; 8 waves as
// 2 (M) x 4 (N), 128 x 64 outputs per wave in 128 accumulator registers, 12 fragment reads per 24 MFMAs.  The fp32 epilogue
// staging does not fit for 256 x 256 at once: four passes of one 64-column slab each (the waves of that column block write,
// every thread splits and stores).
// IO32 (the training form, avt_conv3d_igemm_x3_f32 on the long-K layers now that a rank's items are one batch): the activations
// are fp32 rows — they cannot go to the LDS by DMA, so each thread fetches its two 8-channel chunks of the NEXT step into
// registers (four 16-byte loads) under this step's MFMAs and splits them into the two planes on the way to the LDS after them;
// the weight planes still arrive by LDS-DMA; ONE fragment set instead of two pays for the staging registers; fp32 epilogue
// with the `add` operand.
// OUT32 without IO32 (round 5): plane inputs by LDS-DMA, fp32 rows out, every result divided by a.odiv — the bf16x3 similarity
// (avt_gemm_nt_x3_f32out: Q_hat / T_hat planes are "activations" / "weights" of a pointwise layer with K = D).
template <bool F16, bool IO32 = false, bool OUT32 = IO32>
__global__ __launch_bounds__(512, 2) void conv_x3_xl_kernel(ConvArgs a) {
  constexpr int BM = 256, BN = 256, NTHR = 512, KB = 32;  // K-step in elements
  constexpr int MT = 4, NT = 2;                            // 32 x 32 sub-tiles of a wave's 128 x 64
  constexpr int AU = BM / 128, BU = BN / 128;              // 16-byte chunks per thread, plane and K-step
  constexpr int PL = BM * 64;                              // bytes of one operand plane in a stage (BM == BN)
  constexpr int STG = 4 * PL;                              // A hi | A lo | B hi | B lo
  constexpr int ESTR = BN * 4 + 16;                        // epilogue staging row stride: a slab row = all 256 columns in fp32
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int swz = avt::xcd_contiguous(blockIdx.x, a.nblk);
  const int tm = swz / a.tiles_n, tn = swz % a.tiles_n;
  const int n0 = tn * BN;
  int m0, m_end;  // (IO32 with BatchNorm statistics: tiles laid per group; see conv_args.h)
  if constexpr (IO32) tile_rows(a, tm, BM, m0, m_end);
  else { m0 = tm * BM; m_end = a.M; }
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int lr = lane & 31, lh = lane >> 5;
  // LDS-DMA staging: a wave-instruction lands 1 KB = 16 rows x 64 B at consecutive LDS addresses (lane * 16), so a lane's
  // slot inside its row is fixed (tid & 3) and the swizzle is applied on the SOURCE side: it fetches chunk slot ^ ((row >> 2) & 3)
  const int r0 = tid >> 2, c4 = (tid & 3) ^ ((r0 >> 2) & 3);

  int rowoff[AU];
  unsigned rowmask[AU];
#pragma unroll
  for (int u = 0; u < AU; ++u) {
    const int m = m0 + r0 + 128 * u;
    rowoff[u] = 0;
    rowmask[u] = 0u;
    if (m < m_end && a.pointwise) {
      rowoff[u] = m * a.ldi;
      rowmask[u] = 0x010101u;
    } else if (m < m_end) {
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
  int wrow[BU];
  unsigned wsel[BU];
#pragma unroll
  for (int u = 0; u < BU; ++u) {
    const int n = n0 + r0 + 128 * u;
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
  const __amdgpu_buffer_rsrc_t rih = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ril = __builtin_amdgcn_make_buffer_rsrc((void*)a.in_lo, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt_lo, 0, a.wt_bytes, 0x00020000);
  int2* ltab = reinterpret_cast<int2*>(lds + 2 * STG);
  // (the tap table always lives in the LDS here — the launcher routes longer K loops to the 128 x 128 tile: a table read
  //  that may be global or LDS compiles to a FLAT load, whose use waits for every fragment read issued before it)
  for (int i = tid; i < a.nk * 8; i += NTHR) ltab[i] = a.ktab[i];
  const int nk32 = (a.K + KB - 1) / KB;  // the tap table has one entry per 8-channel chunk: 4 per 32-wide step

  unsigned aoffs[AU], boffs[BU];
  auto gtab = [&](int kt) -> int2 { return ltab[kt * 4 + c4]; };
  auto gcalc = [&](int kt, int2 e) {
    const unsigned ebits = (unsigned)e.y;
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const unsigned sel = ((rowmask[u] & ebits) == ebits) ? 0xFFFFFFFFu : 0u;
      aoffs[u] = (((unsigned)(rowoff[u] + e.x) * (IO32 ? 4u : 2u)) & sel) | (kOob & ~sel);
    }
    const unsigned ksel = ~(unsigned)(e.y >> 31);
    const unsigned kc2 = (unsigned)((kt * 4 + c4) * 16);
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const unsigned sel = ksel & wsel[u];
      boffs[u] = (((unsigned)wrow[u] * 2u + kc2) & sel) | (kOob & ~sel);
      // K-blocked weight planes [K / 32][Cout][32]: the 16 rows of a wave-instruction are 1 KB of consecutive bytes (8 whole
      // cache lines) instead of 16 half lines 2 K bytes apart — measured +5 % on the long-K layers (profiles/r03/probe_xl_traffic.log)
      if (a.wblk) boffs[u] = (((unsigned)(kt * a.Cout + n0 + r0 + 128 * u) * 64u + (unsigned)c4 * 16u) & sel) | (kOob & ~sel);
    }
  };
  auto gprep = [&](int kt) { gcalc(kt, gtab(kt)); };
  constexpr int NPIECE = 2 * (AU + BU);  // 8 LDS-DMA pieces per wave and step
  // piece p of the step that lands in stage `st`: the hardware writes lane l's 16 bytes at (wave's 16-row block) + l * 16;
  // out-of-range offsets (padding taps, rows / channels beyond the tensor) arrive as zeros
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int wrow0 = wid * 16 * 64;  // byte offset of this wave's 16-row block inside a plane
  auto gpiece = [&](int p, char* st) {
    if (p < 2 * AU) {
      const int u = p >> 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds((p & 1) ? ril : rih, (lds_ptr)(st + (p & 1) * PL + wrow0 + u * 128 * 64), 16,
                                               (int)aoffs[u], 0, 0, 0);
    } else {
      const int u = (p - 2 * AU) >> 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds((p & 1) ? rwl : rwh, (lds_ptr)(st + (2 + (p & 1)) * PL + wrow0 + u * 128 * 64), 16,
                                               (int)boffs[u], 0, 0, 0);
    }
  };
  // IO32: this thread's two 8-channel chunks of the step (fp32: two 16-byte loads each) -> planes -> the slot the DMA would fill
  i32x4 ra[IO32 ? AU : 1][2];
  auto aload = [&]() {
    if constexpr (IO32) {
#pragma unroll
      for (int u = 0; u < AU; ++u) {
        const unsigned o2 = aoffs[u] == kOob ? kOob : aoffs[u] + 16u;
        ra[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rih, (int)aoffs[u], 0, 0);
        ra[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rih, (int)o2, 0, 0);
      }
    }
  };
  auto astore = [&](char* st) {
    if constexpr (IO32) {
#pragma unroll
      for (int u = 0; u < AU; ++u) {
        const float* fa = reinterpret_cast<const float*>(&ra[u][0]);
        const float* fb = reinterpret_cast<const float*>(&ra[u][1]);
        uint4 h, l;
        const float f8[8] = {fa[0], fa[1], fa[2], fa[3], fb[0], fb[1], fb[2], fb[3]};
        avt::split8<F16>(f8, h, l);
        const int o = (r0 + 128 * u) * 64 + (tid & 3) * 16;
        *reinterpret_cast<uint4*>(st + o) = h;
        *reinterpret_cast<uint4*>(st + PL + o) = l;
      }
    }
  };
  const int fsw = (lr >> 2) & 3;  // the swizzle of this lane's fragment rows (tile bases are multiples of 32)
  struct Frags {
    i32x4 ah[MT], al[MT], wh[NT], wl[NT];
  };
  // (reads are issued in the order the MFMAs first need them — wh0 wl0 ah0 al0, the other activation fragments, then the
  //  second weight pair — so that the first triple of a step starts after three reads, not after all twelve: in-kernel
  //  stamps put 16 % of a workgroup's time into the fragments' arrival at the top of each step)
  auto fload = [&](Frags& f, const char* st, int ks) {
    const int slot = ((ks * 2 + lh) ^ fsw) * 16;
    auto rd_w = [&](int i) {
      const int o = (wn * 64 + i * 32 + lr) * 64 + slot;
      f.wh[i] = *reinterpret_cast<const i32x4*>(st + 2 * PL + o);
      f.wl[i] = *reinterpret_cast<const i32x4*>(st + 3 * PL + o);
    };
    rd_w(0);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int o = (wm * 128 + j * 32 + lr) * 64 + slot;
      f.ah[j] = *reinterpret_cast<const i32x4*>(st + o);
      f.al[j] = *reinterpret_cast<const i32x4*>(st + PL + o);
    }
#pragma unroll
    for (int i = 1; i < NT; ++i) rd_w(i);
  };
  auto fmul = [&](const Frags& f, int g0, bool more, char* nst) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        acc[i][j] = mfma<F16>(f.wl[i], f.ah[j], acc[i][j]);
        acc[i][j] = mfma<F16>(f.wh[i], f.al[j], acc[i][j]);
        acc[i][j] = mfma<F16>(f.wh[i], f.ah[j], acc[i][j]);
        const int pc = g0 + i * MT + j + (IO32 ? 2 * AU : 0);  // (IO32: the weight pieces only)
        if (more && pc < NPIECE) gpiece(pc, nst);  // the next step's DMA, one piece per MFMA triple of the first k-slice
        __builtin_amdgcn_sched_barrier(0);
      }
  };

  // the tile's 256 bias / scale values, once, into LDS behind the stages and the table (see the epilogue)
  float* cfl = reinterpret_cast<float*>(lds + a.cf_ofs);
  {
    const int n = n0 + (tid & 255);
    float v = tid < 256 ? 0.0f : 1.0f;
    if (n < a.Cout) {
      if (tid < 256) { if (a.bias) v = a.bias[n]; } else if (a.wscale) v = a.wscale[n];
    }
    cfl[tid] = v;
  }
  __syncthreads();
  STAMP_BEGIN();
  gprep(0);
  if constexpr (IO32) {
    aload();
#pragma unroll
    for (int p = 2 * AU; p < NPIECE; ++p) gpiece(p, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    astore(lds);
  } else {
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) gpiece(p, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  STAMP(0);  // prologue
  if constexpr (IO32) {
    for (int kt = 0; kt < nk32; ++kt) {
      const bool more = kt + 1 < nk32;
      char* cur = lds + (kt & 1) * STG;
      char* nst = lds + ((kt + 1) & 1) * STG;
      int2 e = make_int2(0, 0);
      if (more) e = gtab(kt + 1);
      Frags f;
      fload(f, cur, 0);
      if (more) {
        gcalc(kt + 1, e);
        aload();  // in flight under this step's MFMAs
      }
      __builtin_amdgcn_sched_barrier(0);
      fmul(f, 0, more, nst);
      fload(f, cur, 1);
      __builtin_amdgcn_sched_barrier(0);
      fmul(f, NPIECE, false, nst);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (more) astore(nst);
      __syncthreads();
    }
  } else {
    // The loop rotated by one k-slice: what follows a barrier is the reads of the new stage's FIRST slice and then the MFMAs
    // of the previous stage's SECOND slice (operands already in registers) — the two waves of a SIMD arrive at the barrier
    // together, so without this both sit out the LDS latency of the first fragments at the top of every step (16 % of a
    // workgroup's time in the in-kernel stamps, profiles/r03/stamps_xl.log).  The DMA of stage k + 1 (into the buffer of
    // stage k - 1, whose reads all waves completed before barrier k) is issued under those deferred MFMAs and has the first
    // slice of stage k to land before it is waited for.
    Frags f0, f1;
    fload(f0, lds, 0);
    if (nk32 > 1) gprep(1);
    fload(f1, lds, 1);
    __builtin_amdgcn_sched_barrier(0);
    fmul(f0, 0, nk32 > 1, lds + STG);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 1; kt < nk32; ++kt) {
      const bool more = kt + 1 < nk32;
      char* cur = lds + (kt & 1) * STG;
      char* nst = lds + ((kt + 1) & 1) * STG;
      int2 e = make_int2(0, 0);
      if (more) e = gtab(kt + 1);  // (before the fragment reads: its use must not wait for them — LDS returns in order)
      fload(f0, cur, 0);
      if (more) gcalc(kt + 1, e);
      __builtin_amdgcn_sched_barrier(0);
      fmul(f1, 0, more, nst);  // second slice of stage kt - 1 + the DMA of stage kt + 1
      KSTAMP(1);
      fload(f1, cur, 1);
      __builtin_amdgcn_sched_barrier(0);
      fmul(f0, NPIECE, false, nst);  // first slice of stage kt
      KSTAMP(2);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // own DMA pieces landed, own reads of stage kt done
      KSTAMP(3);
      __syncthreads();
      KSTAMP(4);
    }
    fmul(f1, NPIECE, false, lds);
  }

  // ---- epilogue: one 64-ROW slab (all 256 columns) per pass, staged in fp32 in one of TWO buffers (the operand stages + the
  // table area are free now): while every thread splits and stores slab p, the FOUR waves that own rows 64 (p + 1) .. of the
  // tile write their accumulators for slab p + 1 — one barrier per pass.  (Round 4: the slabs used to be 64 COLUMNS wide, owned by
  // the two waves wid = wn and wn + 4 — which share a SIMD — so a pass waited for ~1000 VALU instructions of staging on one SIMD
  // with the other three idle at the barrier: 7.6 k cycles per pass, 30 k per tile, profiles/r04/probe_xl_stamps_epilogue.log.
  // A row slab is staged by the waves wid = 4 wm .. 4 wm + 3, one per SIMD, half of their accumulators each, and a slab row is
  // 512 contiguous bytes per plane in the output instead of 128.)
  constexpr int SR = 64;                       // rows per slab
  constexpr int CPR = BN / 8;                  // 32 eight-channel chunks per slab row
  constexpr int EU = (SR * CPR) / NTHR;        // 4 chunks per thread and pass
  constexpr int EBUF = SR * ESTR;
  static_assert(2 * EBUF <= 2 * STG + kMaxTabSteps * 64, "two staging buffers fit the operand stages + the table area");
  const bool has_res = !OUT32 && a.res != nullptr;  // (IO32: `res` is an fp32 tensor added in the store loop)
  auto stage_slab = [&](auto half, int pass) {  // rows 64 pass .. + 63 = row tiles j = 2 half, 2 half + 1 of the waves wm = pass >> 1
    constexpr int H = decltype(half)::value;
    char* eb = lds + (pass & 1) * EBUF;
    // D layout: column (lane & 31) = m, rows (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) = n
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = wn * 64 + i * 32 + 8 * g + 4 * lh;  // column inside the tile
        // (from LDS, filled in the prologue.  Fetched from global memory here, each (i, g) pair waited for its own two loads —
        //  eight L2 round trips in a row, and the first wait (vmcnt counts stores too) also sat out the previous slab's stores)
        const float4 bv = *reinterpret_cast<const float4*>(cfl + nl);
        const float4 sv = *reinterpret_cast<const float4*>(cfl + 256 + nl);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * H + jj;
          float4 v;
          v.x = acc[i][j][4 * g + 0] * sv.x + bv.x;
          v.y = acc[i][j][4 * g + 1] * sv.y + bv.y;
          v.z = acc[i][j][4 * g + 2] * sv.z + bv.z;
          v.w = acc[i][j][4 * g + 3] * sv.w + bv.w;
          const int rl = jj * 32 + lr;  // row inside the slab
          *reinterpret_cast<float4*>(eb + rl * ESTR + nl * 4) = v;
        }
      }
  };
  KSTAMP(5);  // (the last k-slice's MFMAs)
  if (wm == 0) stage_slab(std::integral_constant<int, 0>{}, 0);
  __syncthreads();
  EPI_STAMP(1);
  // BatchNorm statistics of the rows this thread stores over the four passes (its 8 channels are the same in every row)
  float st_s[IO32 ? 8 : 1], st_q[IO32 ? 8 : 1];
  if constexpr (IO32) {
#pragma unroll
    for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.0f;
  }
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const char* eb = lds + (pass & 1) * EBUF;
    uint4 rrh[EU], rrl[EU];
    if (has_res) {
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int c = tid + NTHR * u;
        const int m = m0 + pass * SR + c / CPR, n = n0 + (c % CPR) * 8;
        const bool ok = m < m_end && n < a.Cout;
        rrh[u] = ok ? *reinterpret_cast<const uint4*>(a.res + (int64_t)m * a.ldr + n) : make_uint4(0u, 0u, 0u, 0u);
        rrl[u] = ok ? *reinterpret_cast<const uint4*>(a.res_lo + (int64_t)m * a.ldr + n) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    if constexpr (IO32) {
      if (a.res) {  // the fp32 `add` operand of this slab, requested in one go (see the 128-wide tile's epilogue)
#pragma unroll
        for (int u = 0; u < EU; ++u) {
          const int c = tid + NTHR * u;
          const int m = m0 + pass * SR + c / CPR, n = n0 + (c % CPR) * 8;
          const bool ok = m < m_end && n < a.Cout;
          const float* rf = reinterpret_cast<const float*>(a.res) + (int64_t)m * a.ldr + n;
          rrh[u] = ok ? *reinterpret_cast<const uint4*>(rf) : make_uint4(0u, 0u, 0u, 0u);
          rrl[u] = ok ? *reinterpret_cast<const uint4*>(rf + 4) : make_uint4(0u, 0u, 0u, 0u);
        }
      }
    }
    if (pass + 1 < 4 && wm == ((pass + 1) >> 1)) {  // into the other buffer (its last readers passed the barrier below)
      if ((pass + 1) & 1) stage_slab(std::integral_constant<int, 1>{}, pass + 1);
      else stage_slab(std::integral_constant<int, 0>{}, pass + 1);
    }
    EPI_STAMP(2);
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int c = tid + NTHR * u;
      const int row = c / CPR, cc = c % CPR;
      const int m = m0 + pass * SR + row, n = n0 + cc * 8;
      if (m < m_end && n < a.Cout) {
        const float4 v0 = *reinterpret_cast<const float4*>(eb + row * ESTR + cc * 32);
        const float4 v1 = *reinterpret_cast<const float4*>(eb + row * ESTR + cc * 32 + 16);
        float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (has_res) {
          const uint32_t* ph = reinterpret_cast<const uint32_t*>(&rrh[u]);
          const uint32_t* pl = reinterpret_cast<const uint32_t*>(&rrl[u]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const avt::f32x2 r = avt::join2<F16>(ph[e], pl[e]);
            x[2 * e] += r.x;
            x[2 * e + 1] += r.y;
          }
        }
        if (a.relu == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = avt::relu_keep_nan(x[e]);
        } else if (a.relu == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.1f * x[e]);
        }
        const int64_t o = (int64_t)out_row(a, m) * a.ldo + n;
        if constexpr (OUT32) {
          if (a.odiv != 0.0f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = __fdiv_rn(x[e], a.odiv);
          }
          if (IO32 && a.res) {  // out = conv + add (train_ops.conv3d_fork): fp32 rows
            const float4 q0 = __builtin_bit_cast(float4, rrh[u]), q1 = __builtin_bit_cast(float4, rrl[u]);
            x[0] += q0.x; x[1] += q0.y; x[2] += q0.z; x[3] += q0.w;
            x[4] += q1.x; x[5] += q1.y; x[6] += q1.z; x[7] += q1.w;
          }
          if constexpr (IO32) {
            if (a.stat_part) {  // (uniform)
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                st_s[e] += x[e];
                st_q[e] += x[e] * x[e];
              }
            }
          }
          float* of = reinterpret_cast<float*>(a.out) + o;
          typedef float f32x4n __attribute__((ext_vector_type(4)));
          __builtin_nontemporal_store(f32x4n{x[0], x[1], x[2], x[3]}, reinterpret_cast<f32x4n*>(of));
          __builtin_nontemporal_store(f32x4n{x[4], x[5], x[6], x[7]}, reinterpret_cast<f32x4n*>(of + 4));
        } else {
          uint4 oh, ol;
          avt::split8<F16>(x, oh, ol);
          *reinterpret_cast<uint4*>(a.out + o) = oh;  // (non-temporal stores here measured equal)
          *reinterpret_cast<uint4*>(a.out_lo + o) = ol;
        }
      }
    }
    EPI_STAMP(3);
    __syncthreads();  // slab pass + 1 is staged; buffer pass & 1 is free for slab pass + 2
    EPI_STAMP(4);
  }
  if constexpr (IO32) {
    if (a.stat_part) {  // 16 thread-rows x [sum | squares] x 256 columns in the (free) staging area -> fp64 fold -> the tile's row
      float* red = reinterpret_cast<float*>(lds);
      const int tr = tid / CPR, cc = tid % CPR;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(tr * 2 + 0) * BN + cc * 8 + e] = st_s[e];
        red[(tr * 2 + 1) * BN + cc * 8 + e] = st_q[e];
      }
      __syncthreads();
      stat_fold_store<BN, NTHR>(a, red, NTHR / CPR, tm, n0, tid);
    }
  }
  STAMP_END();
}

template <bool F16, bool IO32 = false, bool OUT32 = IO32>
int launch_x3_xl(ConvArgs& a, hipStream_t st) {
  if (a.stat_part) a.stat_tpg = (a.stat_mg + 255) / 256;
  const int tiles_m = a.stat_part ? a.stat_groups * a.stat_tpg : (a.M + 255) / 256;
  a.tiles_n = (a.Cout + 255) / 256;
  a.nblk = tiles_m * a.tiles_n;
  constexpr int lds_max = 2 * 4 * 256 * 64 + kMaxTabSteps * 64 + 2 * 256 * 4;
  const int lds_bytes = lds_max;  // (the epilogue's second staging buffer reaches into the table area; then bias | scale)
  a.cf_ofs = lds_max - 2 * 256 * 4;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_xl_kernel<F16, IO32, OUT32>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  if (e != hipSuccess) {
    avt::set_error("avt_conv3d_igemm_x3: hipFuncSetAttribute(%d B LDS): %s", lds_max, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((conv_x3_xl_kernel<F16, IO32, OUT32>), dim3((unsigned)a.nblk), dim3(512), lds_bytes, st, a);
  return avt::check_launch("avt_conv3d_igemm_x3");
}

template <int BM, int BN, int WTM, bool F16, bool IO32 = false, bool BST = false>
int launch_x3(ConvArgs& a, hipStream_t st) {
  if (a.stat_part) a.stat_tpg = (a.stat_mg + BM - 1) / BM;
  const int tiles_m = a.stat_part ? a.stat_groups * a.stat_tpg : (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  a.nblk = tiles_m * a.tiles_n;
  // operand slabs + the tap table of THIS layer (not the 128-step maximum): the wide tile then needs 73.7 + <= 6 KB for
  // every K loop of up to 96 steps, so two workgroups (2 waves per SIMD) share a CU's 160 KB
  constexpr int epi_bytes = BM * (BN * 4 + 16);
  constexpr int opnd_max = 2 * (BM + BN) * LSTR + kMaxTabSteps * 64;
  constexpr int lds_max = (opnd_max > epi_bytes ? opnd_max : epi_bytes) + 2 * BN * 4;
  const int tab_bytes = (a.nk <= kMaxTabSteps ? a.nk : 0) * 64;
  int lds_bytes = 2 * (BM + BN) * LSTR + tab_bytes;
  if (lds_bytes < epi_bytes) lds_bytes = epi_bytes;
  a.cf_ofs = lds_bytes;       // the tile's bias | scale floats behind the operand slabs / the table / the epilogue staging
  lds_bytes += 2 * BN * 4;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_kernel<BM, BN, WTM, F16, IO32, BST>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  if (e != hipSuccess) {
    avt::set_error("avt_conv3d_igemm_x3: hipFuncSetAttribute(%d B LDS): %s", lds_max, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((conv_x3_kernel<BM, BN, WTM, F16, IO32, BST>), dim3((unsigned)a.nblk), dim3(256), lds_bytes, st, a);
  return avt::check_launch("avt_conv3d_igemm_x3");
}

}  // namespace

// which tile avt_conv3d_igemm_x3 launches (bench.py names its roofline rows by it)
extern "C" int avt_conv3d_igemm_x3_xl_picked(int cout, int k, int m) {
  return (cout % 256 == 0 && k >= 256 && k <= kMaxTabSteps * 64 && m >= 256 * 64) ? 1 : 0;
}

static int igemm_x3_impl(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo,
                         const float* bias, const void* res_hi, const void* res_lo, void* out_hi, void* out_lo,
                         const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt, int kh,
                         int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho, int wo, int ldi,
                         int ldo, int ldr, int relu, int out_row_stride, int out_h, int out_w, int plane_dtype,
                         const float* wscale, void* stream, int wblk) {
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_conv3d_igemm_x3: plane_dtype must be 0 (bf16) or 1 (fp16)");
  AVT_REQUIRE(!wscale || avt::aligned16(wscale), "avt_conv3d_igemm_x3: wscale must be 16-byte aligned");
  AVT_REQUIRE(in_lo && wt_lo && out_lo && (!res_hi == !res_lo), "avt_conv3d_igemm_x3: every tensor needs both planes");
  AVT_REQUIRE(avt::aligned16(in_lo) && avt::aligned16(wt_lo) && avt::aligned16(out_lo) && (!res_lo || avt::aligned16(res_lo)),
              "avt_conv3d_igemm_x3: pointers must be 16-byte aligned");
  ConvArgs a;
  const int rc = conv_args_fill(a, "avt_conv3d_igemm_x3", in_hi, wt_hi, bias, res_hi, out_hi, ktab, batch, t, h, w, cin, cout,
                                kt, kh, kw, st, sh, sw, pt, ph, pw, to, ho, wo, ldi, ldo, ldr, relu, out_row_stride, out_h,
                                out_w);
  if (rc != AVT_OK) return rc;
  a.in_lo = static_cast<const uint16_t*>(in_lo);
  a.wt_lo = static_cast<const uint16_t*>(wt_lo);
  a.res_lo = static_cast<const uint16_t*>(res_lo);
  a.out_lo = static_cast<uint16_t*>(out_lo);
  a.wscale = wscale;
  a.wfrag = nullptr;
  a.nup = 0;
  a.wblk = wblk;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // long-K layers whose output channels fill 256-wide tiles: the XL tile
  AVT_REQUIRE(!wblk || (avt_conv3d_igemm_x3_xl_picked(cout, a.K, a.M) && a.K % 32 == 0),
              "avt_conv3d_igemm_x3_wblk: K-blocked weights are the 256 x 256 tile's (avt_conv3d_igemm_x3_xl_picked, K %% 32 == 0)");
  if (avt_conv3d_igemm_x3_xl_picked(cout, a.K, a.M))
    return plane_dtype == AVT_X3_F16 ? launch_x3_xl<true>(a, s) : launch_x3_xl<false>(a, s);
  if (plane_dtype == AVT_X3_F16) {
    if (cout <= 32) return launch_x3<128, 32, 32, true>(a, s);
    if (cout <= 64) return launch_x3<128, 64, 64, true>(a, s);
    return launch_x3<128, 128, 64, true>(a, s);
  }
  if (cout <= 32) return launch_x3<128, 32, 32, false>(a, s);
  if (cout <= 64) return launch_x3<128, 64, 64, false>(a, s);
  return launch_x3<128, 128, 64, false>(a, s);
}

extern "C" int avt_conv3d_igemm_x3(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo,
                                   const float* bias, const void* res_hi, const void* res_lo, void* out_hi, void* out_lo,
                                   const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt, int kh,
                                   int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho, int wo, int ldi,
                                   int ldo, int ldr, int relu, int out_row_stride, int out_h, int out_w, int plane_dtype,
                                   const float* wscale, void* stream) {
  return igemm_x3_impl(in_hi, in_lo, wt_hi, wt_lo, bias, res_hi, res_lo, out_hi, out_lo, ktab, batch, t, h, w, cin, cout, kt, kh, kw,
                       st, sh, sw, pt, ph, pw, to, ho, wo, ldi, ldo, ldr, relu, out_row_stride, out_h, out_w, plane_dtype, wscale,
                       stream, 0);
}

// the same convolution with the weight planes in K-blocked order [K / 32][cout][32] (see include/avt.h): only where
// avt_conv3d_igemm_x3_xl_picked(cout, K, M) holds and K % 32 == 0
extern "C" int avt_conv3d_igemm_x3_wblk(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo,
                                        const float* bias, const void* res_hi, const void* res_lo, void* out_hi, void* out_lo,
                                        const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt, int kh,
                                        int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho, int wo, int ldi,
                                        int ldo, int ldr, int relu, int out_row_stride, int out_h, int out_w, int plane_dtype,
                                        const float* wscale, void* stream) {
  return igemm_x3_impl(in_hi, in_lo, wt_hi, wt_lo, bias, res_hi, res_lo, out_hi, out_lo, ktab, batch, t, h, w, cin, cout, kt, kh, kw,
                       st, sh, sw, pt, ph, pw, to, ho, wo, ldi, ldo, ldr, relu, out_row_stride, out_h, out_w, plane_dtype, wscale,
                       stream, 1);
}

// out[m][n] = (sum_k A[m][k] * B[n][k]) / divisor with A, B as plane pairs and an fp32 result: the 256 x 256 LDS-DMA tile as a plain
// NT GEMM (see include/avt.h) — the bf16x3 similarity Q_hat T_hat^T / temp.  ktab = avt_conv3d_ktab(k, 1, 1, 1, 1, m, lda).
extern "C" int avt_gemm_nt_x3_f32out(const void* a_hi, const void* a_lo, int lda, const void* b_hi, const void* b_lo, float* out, int64_t ldo,
                                     int m, int n, int k, float divisor, const int32_t* ktab, int plane_dtype, void* stream) {
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_gemm_nt_x3_f32out: plane_dtype must be 0 (bf16) or 1 (fp16)");
  AVT_REQUIRE(a_lo && b_lo && avt::aligned16(a_lo) && avt::aligned16(b_lo) && divisor != 0.0f, "avt_gemm_nt_x3_f32out: both planes, divisor != 0");
  AVT_REQUIRE(m > 0 && n % 256 == 0 && k % 32 == 0 && k >= 256 && k <= kMaxTabSteps * 64 && ldo >= n && ldo % 4 == 0 && ldo < (1ll << 31),
              "avt_gemm_nt_x3_f32out: n %% 256 == 0, 256 <= k <= %d in multiples of 32 (got m %d n %d k %d)", kMaxTabSteps * 64, m, n, k);
  ConvArgs a;
  const int rc = conv_args_fill(a, "avt_gemm_nt_x3_f32out", a_hi, b_hi, nullptr, nullptr, out, ktab, 1, 1, 1, m, k, n, 1, 1, 1, 1, 1, 1, 0, 0, 0,
                                0, 0, 0, lda, 8 * (int)((ldo + 7) / 8), 0, 0, 1, 0, 0);
  if (rc != AVT_OK) return rc;
  a.ldo = (int)ldo;
  a.in_lo = static_cast<const uint16_t*>(a_lo);
  a.wt_lo = static_cast<const uint16_t*>(b_lo);
  a.res_lo = nullptr;
  a.out_lo = nullptr;
  a.wscale = nullptr;
  a.wfrag = nullptr;
  a.nup = 0;
  a.wblk = 0;
  a.odiv = divisor;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return plane_dtype == AVT_X3_F16 ? launch_x3_xl<true, false, true>(a, s) : launch_x3_xl<false, false, true>(a, s);
}

// fp32 rows in, fp32 rows out, split-plane arithmetic in between (the IO32 form of the kernel): see include/avt.h
// the IO32 tile that avt_conv3d_igemm_x3_f32 launches for (cout, K, M): 256 = the XL tile, else 128 (rows per tile)
// ... and 64 (round 6): a wide layer (128-column tiles) at a batch whose 128-row tiles would not give every CU its two workgroups —
// config 5 at ONE item per rank: 15 target clips are 368 tiles at res4 and 184 at res5 on 256 CUs, 170-230 TFLOP/s where the same
// layers reach 300 at 8 items (profiles/r06/train_layers_one_item.log).  Rows of 64 double the workgroups (the operand the kernel
// pays VALU for — the fp32 activation rows it splits — is fetched as often as before; the weight planes twice as often, from L2).
static int g_io32_small = 1;  // avt_conv_x3_set_small_tile
static int io32_tile_rows(int cout, int k, int64_t m) {
  if (k % 32 == 0 && m < (1ll << 31) && avt_conv3d_igemm_x3_xl_picked(cout, k, (int)m) && m >= 256 * 256) return 256;
  if (g_io32_small && cout > 64 && ((m + 127) / 128) * ((cout + 127) / 128) < 512) return 64;
  return 128;
}

// 1 (default): the 64-row tile where io32_tile_rows picks it; 0: never (A/Bs, tests of the 128-row form at small sizes) -> the old value
extern "C" int avt_conv_x3_set_small_tile(int on) {
  const int was = g_io32_small;
  g_io32_small = on ? 1 : 0;
  return was;
}

static int igemm_x3_f32_impl(const float* in, const void* wt_hi, const void* wt_lo, const float* wscale, const float* add,
                             float* out, const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt, int kh,
                             int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho, int wo, int ldi, int ldo, int lda,
                             int out_row_stride, int out_h, int out_w, int plane_dtype, void* stream, double* stat_part = nullptr,
                             int stat_groups = 0, int stat_c = 0, const ConvArgs* bst = nullptr) {
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_conv3d_igemm_x3_f32: plane_dtype must be 0 (bf16) or 1 (fp16)");
  AVT_REQUIRE(wt_lo && avt::aligned16(wt_lo) && (!wscale || avt::aligned16(wscale)), "avt_conv3d_igemm_x3_f32: weight planes / wscale NULL or unaligned");
  AVT_REQUIRE((int64_t)batch * t * h * w * ldi < (1ll << 30) - 64, "avt_conv3d_igemm_x3_f32: input too large for 32-bit byte offsets");
  ConvArgs a;
  const int rc = conv_args_fill(a, "avt_conv3d_igemm_x3_f32", in, wt_hi, nullptr, add, out, ktab, batch, t, h, w, cin, cout, kt, kh, kw,
                                st, sh, sw, pt, ph, pw, to, ho, wo, ldi, ldo, add ? lda : 0, 0, out_row_stride, out_h, out_w);
  if (rc != AVT_OK) return rc;
  a.in_bytes *= 2u;  // fp32 elements
  a.in_lo = nullptr;
  a.wt_lo = static_cast<const uint16_t*>(wt_lo);
  a.res_lo = nullptr;
  a.out_lo = nullptr;
  a.wscale = wscale;
  a.wfrag = nullptr;
  a.nup = 0;
  a.wblk = 0;
  if (stat_part) {
    AVT_REQUIRE(stat_groups >= 1 && a.M % stat_groups == 0 && stat_c >= 8 && cout % stat_c == 0 && a.oH == 0 && (!add || bst) &&
                    avt::aligned16(stat_part),
                "avt_conv3d_igemm_x3_f32_stats: %d rows in %d groups, %d channels of %d columns, no row remap, no add operand", a.M,
                stat_groups, stat_c, cout);
    // pixel-grouped form (stat_c < cout: column n is channel n % stat_c): stat_fold_store folds the pixel copies of a channel
    // INSIDE one N tile and writes the channel's slot with a plain store — two N tiles of one M tile would overwrite each
    // other's partial sums, so every copy of a channel must lie in the launch's single N tile
    {
      const int bn_launch = (!bst && a.oH == 0 && io32_tile_rows(cout, a.K, a.M) == 256) ? 256 : (cout <= 32 ? 32 : (cout <= 64 ? 64 : 128));
      AVT_REQUIRE(stat_c == cout || cout <= bn_launch,
                  "avt_conv3d_igemm_x3_f32_stats: %d pixel copies of %d channels span more than one %d-column tile", cout / stat_c, stat_c,
                  bn_launch);
    }
    if (bst) {  // backward statistics: the BatchNorm input has the output's geometry, contiguous rows
      AVT_REQUIRE(bst->bst_x && bst->bst_mean && bst->bst_invstd && bst->bst_gamma && ldo == cout && avt::aligned16(bst->bst_x) &&
                      (!bst->bst_relu || bst->bst_mask || bst->bst_beta),
                  "avt_conv3d_igemm_x3_f32_bwdstats: BatchNorm input / statistics / scale missing, or the output rows are not contiguous");
      a.bst_x = bst->bst_x; a.bst_mean = bst->bst_mean; a.bst_invstd = bst->bst_invstd; a.bst_gamma = bst->bst_gamma;
      a.bst_beta = bst->bst_beta; a.bst_mask = bst->bst_mask; a.bst_relu = bst->bst_relu;
    }
    a.stat_part = stat_part;
    a.stat_groups = stat_groups;
    a.stat_mg = a.M / stat_groups;
    a.stat_c = stat_c;
    AVT_REQUIRE((int64_t)stat_groups * ((a.stat_mg + 127) / 128) < (1ll << 24), "avt_conv3d_igemm_x3_f32_stats: too many tiles");
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (bst) {  // backward statistics: the 128-row tiles only (bf16 planes: gradients)
    AVT_REQUIRE(plane_dtype == AVT_X3_BF16 && io32_tile_rows(cout, a.K, a.M) != 256,
                "avt_conv3d_igemm_x3_f32_bwdstats: bf16 planes, layers of the 128- / 64-row tiles (avt_conv3d_igemm_x3_f32_bwdstats_rows)");
    if (cout <= 32) return launch_x3<128, 32, 32, false, true, true>(a, s);
    if (cout <= 64) return launch_x3<128, 64, 64, false, true, true>(a, s);
    if (io32_tile_rows(cout, a.K, a.M) == 64) return launch_x3<64, 128, 32, false, true, true>(a, s);
    return launch_x3<128, 128, 64, false, true, true>(a, s);
  }
  const bool small = io32_tile_rows(cout, a.K, a.M) == 64;
  // long-K layers at a batch that fills 256 x 256 tiles (a rank's items as one batch): the XL tile's IO32 form
  if (a.oH == 0 && io32_tile_rows(cout, a.K, a.M) == 256)
    return plane_dtype == AVT_X3_F16 ? launch_x3_xl<true, true>(a, s) : launch_x3_xl<false, true>(a, s);
  if (plane_dtype == AVT_X3_F16) {
    if (cout <= 32) return launch_x3<128, 32, 32, true, true>(a, s);
    if (cout <= 64) return launch_x3<128, 64, 64, true, true>(a, s);
    if (small) return launch_x3<64, 128, 32, true, true>(a, s);
    return launch_x3<128, 128, 64, true, true>(a, s);
  }
  if (cout <= 32) return launch_x3<128, 32, 32, false, true>(a, s);
  if (cout <= 64) return launch_x3<128, 64, 64, false, true>(a, s);
  if (small) return launch_x3<64, 128, 32, false, true>(a, s);
  return launch_x3<128, 128, 64, false, true>(a, s);
}

extern "C" int avt_conv3d_igemm_x3_f32(const float* in, const void* wt_hi, const void* wt_lo, const float* wscale, const float* add,
                                       float* out, const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt, int kh,
                                       int kw, int st, int sh, int sw, int pt, int ph, int pw, int ldi, int ldo, int lda, int plane_dtype,
                                       void* stream) {
  return igemm_x3_f32_impl(in, wt_hi, wt_lo, wscale, add, out, ktab, batch, t, h, w, cin, cout, kt, kh, kw, st, sh, sw, pt, ph, pw, 0, 0, 0,
                           ldi, ldo, lda, 1, 0, 0, plane_dtype, stream);
}

// ... leaving the train-mode BatchNorm statistics of its output behind (see include/avt.h): per-tile sums in `stat_part`, the rows as
// `groups` slabs with statistics of their own; avt_conv3d_igemm_x3_f32_stat_rows = rows of partials per group it writes
extern "C" int avt_conv3d_igemm_x3_f32_stat_rows(int cout, int k, int64_t m, int groups) {
  if (groups < 1 || m <= 0 || m % groups) return -1;
  const int bm = io32_tile_rows(cout, k, m);
  return (int)((m / groups + bm - 1) / bm);
}

extern "C" int avt_conv3d_igemm_x3_f32_stats(const float* in, const void* wt_hi, const void* wt_lo, const float* wscale, float* out,
                                             const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt, int kh, int kw,
                                             int st, int sh, int sw, int pt, int ph, int pw, int ldi, int ldo, int plane_dtype,
                                             void* stat_part, int groups, int stat_c, void* stream) {
  AVT_REQUIRE(stat_part, "avt_conv3d_igemm_x3_f32_stats: NULL stat_part");
  return igemm_x3_f32_impl(in, wt_hi, wt_lo, wscale, nullptr, out, ktab, batch, t, h, w, cin, cout, kt, kh, kw, st, sh, sw, pt, ph, pw, 0, 0, 0,
                           ldi, ldo, 0, 1, 0, 0, plane_dtype, stream, static_cast<double*>(stat_part), groups, stat_c);
}

// rows of partials per group avt_conv3d_igemm_x3_f32_bwdstats writes, or -1 where it does not apply (the 256 x 256 tile's layers)
extern "C" int avt_conv3d_igemm_x3_f32_bwdstats_rows(int cout, int k, int64_t m, int groups) {
  if (groups < 1 || m <= 0 || m % groups) return -1;
  const int bm = io32_tile_rows(cout, k, m);
  if (bm == 256) return -1;
  return (int)((m / groups + bm - 1) / bm);
}

// The stride-1 INPUT GRADIENT whose result is the output gradient of a train-mode BatchNorm (+ ReLU): see include/avt.h
extern "C" int avt_conv3d_igemm_x3_f32_bwdstats(const float* in, const void* wt_hi, const void* wt_lo, const float* wscale, const float* add,
                                                float* out, const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt,
                                                int kh, int kw, int pt, int ph, int pw, int ldi, int ldo, int lda, int plane_dtype,
                                                const float* bn_x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                                const float* bn_beta, const void* bn_mask, int relu, void* stat_part, int groups, int stat_c,
                                                void* stream) {
  AVT_REQUIRE(stat_part, "avt_conv3d_igemm_x3_f32_bwdstats: NULL stat_part");
  ConvArgs b = {};
  b.bst_x = bn_x; b.bst_mean = bn_mean; b.bst_invstd = bn_invstd; b.bst_gamma = bn_gamma; b.bst_beta = bn_beta;
  b.bst_mask = static_cast<const uint8_t*>(bn_mask); b.bst_relu = relu;
  return igemm_x3_f32_impl(in, wt_hi, wt_lo, wscale, add, out, ktab, batch, t, h, w, cin, cout, kt, kh, kw, 1, 1, 1, pt, ph, pw, 0, 0, 0, ldi,
                           ldo, lda, 1, 0, 0, plane_dtype, stream, static_cast<double*>(stat_part), groups, stat_c, &b);
}

// ... with an explicit output extent (any padding on the far side) and the output-row remap of avt_conv3d_igemm_x3: one class of
// a strided transposed convolution (train_ops._dgrad_strided); no `add` operand
extern "C" int avt_conv3d_igemm_x3_f32_ex(const float* in, const void* wt_hi, const void* wt_lo, const float* wscale, float* out,
                                          const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt, int kh, int kw,
                                          int pt, int ph, int pw, int to, int ho, int wo, int ldi, int ldo, int out_row_stride, int out_h,
                                          int out_w, int plane_dtype, void* stream) {
  return igemm_x3_f32_impl(in, wt_hi, wt_lo, wscale, nullptr, out, ktab, batch, t, h, w, cin, cout, kt, kh, kw, 1, 1, 1, pt, ph, pw, to, ho,
                           wo, ldi, ldo, 0, out_row_stride, out_h, out_w, plane_dtype, stream);
}
