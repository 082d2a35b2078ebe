// conv3d_wgrad_x3 — the WEIGHT gradient of the training convolutions on the split-plane MFMA arithmetic (the third piece
// of the training step next to conv_x3's IO32 forward / input-gradient form; contrastive_video_textures/train.py:114-141
// runs it through autograd -> MIOpen, whose fp32 bwd_weight kernels were 37 % of the step, profiles/r02/
// train_fp32_default_kernels.log):
//     dW[co][tap][ci] = sum over output positions m of  dY[m][co] * X[in(m, tap)][ci]
// a GEMM whose REDUCTION runs over the rows of both operands ([positions, channels] fp32 NDHWC rows): both arrive
// transposed for the matrix pipe.  Nothing is transposed on the way in: a 64-position slab of either operand is read with
// coalesced 16-byte loads (32 lanes = 512 contiguous bytes of a row), split into bf16 hi / lo planes with the hardware
// pair conversion (v_cvt_pk_bf16_f32: 3 VALU operations per element, already packed) and written to the LDS as it lies,
// [position][channel], 256-byte rows with the chunk swizzle of the CDNA4 guide; the MFMA operands are then fetched with
// gfx950's TRANSPOSING read, ds_read_b64_tr_b16 (a 4-position x 16-channel block per 16 lanes, delivered channel-major):
// two of them make the 8 reduction values a lane owes for its channel row.  (The first version transposed on the way in —
// a lane per position, one ds_write_b16 per element — and lost to MIOpen: its global loads touched 64 cache lines per
// instruction; profiles/r02/train_wgrad_v1_uncoalesced.log.)  bf16 planes, not fp16: dY needs fp32's exponent range.
// Per product: dyh*xh + dyh*xl + dyl*xh into one fp32 accumulator (2^-16), like conv_x3.
// The x operand's axis is (tap, ci) FLATTENED (E = taps * Cin, dW is [co][E]): im2col rows that are never materialised — a
// float4 of x is fetched from the row its tap points at (per-slab table of 64 positions x taps offsets, padding = -1) —
// so an 8-channel 3x3 layer fills 72 of a tile's 128 rows and reads dY once, not 8 of 128 nine times over.
// Work split: one workgroup = (chunk of slabs, 128 of the longer axis, BN of the shorter one); partial tiles are added into
// dW with hardware fp32 atomics (dW is zeroed first; the order is not deterministic, as MIOpen's is not).  Strides,
// padding and ragged edges live in the table; input channels in multiples of 4 (a float4 per tap), output channels of 8; at most
// 28 taps (3x3x3, 7x1x1) with two workgroups per CU, up to 49 ([1,7,7]: the SlowFast stems, whose 3 input channels travel as
// 4 — the stems' weight gradient was the last MIOpen kernel of weight in the step, 77 ms of 730) with one; a frame-tap slice
// of a longer filter (the fast stem's [5,7,7] = five [1,7,7] slices) is a sub-problem: explicit output extent, any pt, ldw.
#include <stdlib.h>

#include <type_traits>

#include "avt_common.h"
#include "conv_args.h"

// phase-skip diagnostic (tools/probe_wgrad_phases.py against a library built with -DAVT_WGRAD_DBG_CONST=n: 1 skip the LDS
// stage, 2 the MFMAs, 4 the global loads): compile-time, the shipped library carries no switch
#ifndef AVT_WGRAD_DBG_CONST
#define AVT_WGRAD_DBG_CONST 0
#endif
#define WG_SKIP(bit) (((AVT_WGRAD_DBG_CONST) & (bit)) != 0)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxTaps = 28;   // taps the position table holds with two workgroups per CU (3x3x3 = 27; the 7x1x1 lateral ones = 7)
constexpr int kMaxTapsBig = 49;  // ... with one workgroup per CU ([1,7,7]: the stems)

struct WgArgs {
  const float* dy;  // [M, ldy]
  const float* x;   // [B*T*H*W, ldx]
  float* dw;        // [Cout][E], E = taps * Cin: the x operand's axis is (tap, ci) flattened — im2col rows, never materialised
  int B, T, H, W, To, Ho, Wo, KT, KH, KW, st, sh, sw, pt, ph, pw;
  int Cin, Cout, ldx, ldy, M, taps, E;
  int tapcap;  // taps the position table has room for (its row stride)
  int ldw;     // row stride of dW (elements): E, or more when dW is a slice of a longer filter
  int r_tiles, s_tiles;  // tiles of the 128-wide / BN-wide operand axis
  int nslab, slabs_per_chunk, nchunk;
  FastDiv dWo, dHo, dTo, dKW, dKH, dCin;
  int pointwise;               // 1x1x1 / stride 1 / no padding: input row = output position
  unsigned x_bytes, dy_bytes;  // the pipelined tile's buffer loads (tensors below kRowOob bytes, or the phase-serial tile runs)
};

__device__ __forceinline__ f32x16 mfma(i32x4 a, i32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kPlane = 64 * 256;  // one plane of one operand: [64 positions][128 channels x 2 B]
// BUF forms (tensors below 4 GB): the position tables hold BYTE offsets of rows, kRowOob for "no row"; a thread adds its channel's
// bytes and loads through a buffer resource, whose bounds check returns zeros — see the pipelined tile below
constexpr unsigned kRowOob = 0xFFF00000u;  // (+ a channel offset of < 1 MB stays beyond every buffer and does not wrap)

// byte offset of 16-byte chunk `ch` (8 channels) of position row `row`: 256-byte rows, chunks XOR-swizzled so that the row
// writes and the transposed reads are both conflict-free (cdna_hip_programming.md T10, image (b))
__device__ __forceinline__ int swz(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// four consecutive channels of one position -> 8 bytes of the hi plane + 8 bytes of the lo plane
__device__ __forceinline__ void put4(char* lds, int hi_base, int off, float4 v) {
  const uint32_t h01 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v.x, v.y}, bf16x2));
  const uint32_t h23 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v.z, v.w}, bf16x2));
  const float r0 = v.x - __builtin_bit_cast(float, h01 << 16), r1 = v.y - __builtin_bit_cast(float, h01 & 0xFFFF0000u);
  const float r2 = v.z - __builtin_bit_cast(float, h23 << 16), r3 = v.w - __builtin_bit_cast(float, h23 & 0xFFFF0000u);
  const uint32_t l01 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r0, r1}, bf16x2));
  const uint32_t l23 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r2, r3}, bf16x2));
  *reinterpret_cast<uint2*>(lds + hi_base + off) = make_uint2(h01, h23);
  *reinterpret_cast<uint2*>(lds + hi_base + kPlane + off) = make_uint2(l01, l23);
}

// SWAP = false: the 128-wide axis is x's (tap, ci), the BN-wide one dy's co; true: the other way round (Cout > taps * Cin).
template <int BN, bool SWAP, bool BUF>
__global__ __launch_bounds__(256, 2) void wgrad_x3_kernel(WgArgs a) {
  constexpr int BM = 128;
  constexpr int WTM = BN == 32 ? 32 : 64;
  constexpr int WAVES_M = BM / WTM, WAVES_N = 4 / WAVES_M, WN = BN / WAVES_N;
  constexpr int NT = WN / 32, MT = WTM / 32;
  constexpr int RQ = 8, SQ = BN / 16;  // float4 loads per thread and slab
  constexpr int XQ = SWAP ? SQ : RQ;   // ... of which belong to the x operand
  constexpr int XW = SWAP ? BN : BM;   // width of the x operand's tile
  constexpr int R_HI = 0, S_HI = 2 * kPlane, YTAB = 4 * kPlane, XTAB = YTAB + 2 * 64 * 4;
  static_assert(NT >= 1 && MT >= 1, "tile");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int* const ytab = reinterpret_cast<int*>(lds + YTAB);  // [2][64]: element offset of a position's dy row, -1 = none
  int* const xtab = reinterpret_cast<int*>(lds + XTAB);  // [2][64][tapcap]: ... of its x row under every tap, -1 = padding
  int* const ttab = xtab + 2 * 64 * a.tapcap;            // [taps]: dt | dh << 8 | dw << 16, once per workgroup (two divisions per tap and
                                                         // slab otherwise: the tile is bound by its vector instructions)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;

  // work item: s-tile fastest (neighbours share the 128-wide operand's slabs), then r-tile, chunk — and neighbours in this
  // list run on ONE XCD (blocks are dealt round-robin over the 8 XCDs, each with its own L2): all tiles of a chunk then
  // meet their 64-position slabs of dY and X in that L2.  Dealt naively the kernel was bound by its global loads (phase-skip
  // diagnostic, profiles/r02/probe_wgrad_phases.log: without them 2.1x faster, without the MFMAs 1.15x).
  int b_ = avt::xcd_contiguous(blockIdx.x, gridDim.x);
  const int ts = b_ % a.s_tiles; b_ /= a.s_tiles;
  const int tr = b_ % a.r_tiles;
  const int chunk = b_ / a.r_tiles;
  const int r0 = tr * BM, s0 = ts * BN;
  const int slab0 = chunk * a.slabs_per_chunk;
  const int slab1 = min(a.nslab, slab0 + a.slabs_per_chunk);
  const int x0 = SWAP ? s0 : r0, y0 = SWAP ? r0 : s0;  // first (tap, ci) index / first co of this tile

  // the x operand's loads: thread-constant (tap, ci) of each float4
  int xtap[XQ], xci[XQ];
#pragma unroll
  for (int q = 0; q < XQ; ++q) {
    const int idx = q * 256 + tid, e = x0 + 4 * (idx % (XW / 4));
    xtap[q] = e < a.E ? (int)fastdiv((uint32_t)e, a.dCin) : -1;
    xci[q] = e - xtap[q] * a.Cin;
  }

  f32x16 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // table entries: element offsets, -1 = none; BUF: byte offsets, kRowOob = none
  const int none = BUF ? (int)kRowOob : -1;
  auto enc = [](int elem) { return BUF ? (int)(4u * (unsigned)elem) : elem; };
  auto decode = [&](int slab, int buf) {  // thread -> position tid & 63, taps (tid >> 6), + 4, + 8, ...
    const int p = tid & 63, m = slab * 64 + p;
    const bool ok = m < a.M;
    if (BUF && a.pointwise) {  // 1x1x1, stride 1, no padding: output position m reads input row m
      if (tid < 64) {
        ytab[buf * 64 + p] = ok ? enc(m * a.ldy) : none;
        xtab[(buf * 64 + p) * a.tapcap] = ok ? enc(m * a.ldx) : none;
      }
      return;
    }
    const int q1 = (int)fastdiv((uint32_t)(ok ? m : 0), a.dWo), wo = (ok ? m : 0) - q1 * a.Wo;
    const int q2 = (int)fastdiv((uint32_t)q1, a.dHo), ho = q1 - q2 * a.Ho;
    const int bb = (int)fastdiv((uint32_t)q2, a.dTo), to = q2 - bb * a.To;
    if (tid < 64) ytab[buf * 64 + p] = ok ? enc(m * a.ldy) : none;
    for (int tap = tid >> 6; tap < a.taps; tap += 4) {
      const int pk = ttab[tap], dt = pk & 255, dh = (pk >> 8) & 255, dw_ = pk >> 16;
      const int ti = to * a.st - a.pt + dt, hi = ho * a.sh - a.ph + dh, wi = wo * a.sw - a.pw + dw_;
      const bool in = ok && (unsigned)ti < (unsigned)a.T && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      xtab[(buf * 64 + p) * a.tapcap + tap] = in ? enc((((bb * a.T + ti) * a.H + hi) * a.W + wi) * a.ldx) : none;
    }
  };
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  float4 rq[RQ], sq[SQ];
  unsigned rmask = 0u, smask = 0u;  // which pieces are real (the others are zeroed when they are staged, not when they are loaded)
  // Table reads first, all of them, then the 16-byte loads, all unconditional (an out-of-range piece reads the tensor's
  // first bytes and is zeroed afterwards): written as `ok ? load : 0` per piece, every load sat behind its own LDS round trip
  // and exec-mask branch, and the load phase was half of the kernel's time (profiles/r02/probe_wgrad_phases.log).
  auto gload = [&](int buf) {
    int rrow[RQ], srow[SQ];
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int idx = q * 256 + tid, p = idx >> 5;
      if constexpr (SWAP) rrow[q] = ytab[buf * 64 + p];
      else rrow[q] = xtab[(buf * 64 + p) * a.tapcap + (xtap[q] < 0 ? 0 : xtap[q])];
    }
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
      const int idx = q * 256 + tid, p = idx / (BN / 4);
      if constexpr (SWAP) srow[q] = xtab[(buf * 64 + p) * a.tapcap + (xtap[q] < 0 ? 0 : xtap[q])];
      else srow[q] = ytab[buf * 64 + p];
    }
    if constexpr (BUF) {  // one add + one select per piece; out-of-range pieces come back as zeros (rmask / smask stay all-ones)
      rmask = smask = 0xFFFFFFFFu;
#pragma unroll
      for (int q = 0; q < RQ; ++q) {
        const int idx = q * 256 + tid, c = r0 + 4 * (idx & 31);
        const bool col = SWAP ? c < a.Cout : xtap[q] >= 0;
        const unsigned off = col ? (unsigned)rrow[q] + 4u * (unsigned)(SWAP ? c : xci[q]) : kRowOob;
        rq[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(SWAP ? rsy : rsx, (int)off, 0, 0));
      }
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        const int idx = q * 256 + tid, c = s0 + 4 * (idx % (BN / 4));
        const bool col = SWAP ? xtap[q] >= 0 : c < a.Cout;
        const unsigned off = col ? (unsigned)srow[q] + 4u * (unsigned)(SWAP ? xci[q] : c) : kRowOob;
        sq[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(SWAP ? rsx : rsy, (int)off, 0, 0));
      }
      return;
    }
    rmask = smask = 0u;
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int idx = q * 256 + tid;
      int64_t off;
      bool ok;
      if constexpr (SWAP) {
        const int c = r0 + 4 * (idx & 31);
        ok = rrow[q] >= 0 && c < a.Cout;
        off = (int64_t)rrow[q] + c;
      } else {
        ok = rrow[q] >= 0 && xtap[q] >= 0;
        off = (int64_t)rrow[q] + xci[q];
      }
      rmask |= ok ? 1u << q : 0u;
      rq[q] = *reinterpret_cast<const float4*>((SWAP ? a.dy : a.x) + (ok ? off : 0));
    }
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
      const int idx = q * 256 + tid;
      int64_t off;
      bool ok;
      if constexpr (SWAP) {
        ok = srow[q] >= 0 && xtap[q] >= 0;
        off = (int64_t)srow[q] + xci[q];
      } else {
        const int c = s0 + 4 * (idx % (BN / 4));
        ok = srow[q] >= 0 && c < a.Cout;
        off = (int64_t)srow[q] + c;
      }
      smask |= ok ? 1u << q : 0u;
      sq[q] = *reinterpret_cast<const float4*>((SWAP ? a.x : a.dy) + (ok ? off : 0));
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int idx = q * 256 + tid, p = idx >> 5, cq = idx & 31;
      put4(lds, R_HI, swz(p, cq >> 1) + 8 * (cq & 1), (BUF || ((rmask >> q) & 1u)) ? rq[q] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
      const int idx = q * 256 + tid, p = idx / (BN / 4), cq = idx % (BN / 4);
      put4(lds, S_HI, swz(p, cq >> 1) + 8 * (cq & 1), (BUF || ((smask >> q) & 1u)) ? sq[q] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
  };
  // transposed-read addresses at k-slice 0 (T10: lane 4q + p of a 16-lane group supplies row q, channels 4p .. 4p + 3 of
  // the block; groups 0 / 1 = channels 0-15 / 16-31 of the 32-row operand block, groups 2 / 3 the same for positions + 8).
  // A k-slice is 16 positions = 4096 bytes further: + 16 rows leaves the swizzle's (row & 3) and (row >> 2) & 3 alone.
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  int raddr[MT][2], saddr[NT][2];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
      raddr[j][h2] = R_HI + swz(8 * (g >> 1) + 4 * h2 + tq, (wm * WTM + j * 32 + 16 * (g & 1)) / 8 + (tp >> 1)) + 8 * (tp & 1);
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
      saddr[i][h2] = S_HI + swz(8 * (g >> 1) + 4 * h2 + tq, (wn * WN + i * 32 + 16 * (g & 1)) / 8 + (tp >> 1)) + 8 * (tp & 1);
  auto frag = [&](int a0, int a1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a0)));
    const uint2 v = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a1)));
    return i32x4{(int)u.x, (int)u.y, (int)v.x, (int)v.y};
  };
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      i32x4 rh[MT], rl[MT], sh_[NT], sl[NT];
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        rh[j] = frag(raddr[j][0] + 4096 * ks, raddr[j][1] + 4096 * ks);
        rl[j] = frag(raddr[j][0] + kPlane + 4096 * ks, raddr[j][1] + kPlane + 4096 * ks);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        sh_[i] = frag(saddr[i][0] + 4096 * ks, saddr[i][1] + 4096 * ks);
        sl[i] = frag(saddr[i][0] + kPlane + 4096 * ks, saddr[i][1] + kPlane + 4096 * ks);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          acc[i][j] = mfma(sl[i], rh[j], acc[i][j]);  // small terms first
          acc[i][j] = mfma(sh_[i], rl[j], acc[i][j]);
          acc[i][j] = mfma(sh_[i], rh[j], acc[i][j]);
        }
    }
  };

  for (int tap = tid; tap < a.taps; tap += 256) {
    const int t1 = (int)fastdiv((uint32_t)tap, a.dKW), dw_ = tap - t1 * a.KW;
    const int dt = (int)fastdiv((uint32_t)t1, a.dKH), dh = t1 - dt * a.KH;
    ttab[tap] = dt | (dh << 8) | (dw_ << 16);
  }
  __syncthreads();
  if (slab0 < slab1) {
    decode(slab0, slab0 & 1);
    __syncthreads();
    gload(slab0 & 1);
  }
  for (int s = slab0; s < slab1; ++s) {
    if (s + 1 < slab1) decode(s + 1, (s + 1) & 1);
    __syncthreads();  // the previous slab's fragments have been read; the next slab's table is written
    if (!WG_SKIP(1)) lstore();
    __syncthreads();
    if (s + 1 < slab1 && !WG_SKIP(4)) gload((s + 1) & 1);  // in flight under the MFMAs
    if (!WG_SKIP(2)) compute();
  }

  // D layout: column (lane & 31) = index on the 128-wide axis, rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = index on the other.
  // dW is [co][e]: consecutive lanes must walk e for the atomics to coalesce.  Not swapped, they do.  Swapped, lanes walk co
  // (rows 4 E bytes apart): the tile goes through the LDS first.
  if constexpr (!SWAP) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int e = r0 + wm * WTM + j * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = s0 + wn * WN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (e < a.E && co < a.Cout) unsafeAtomicAdd(a.dw + (int64_t)co * a.ldw + e, acc[i][j][r]);
        }
      }
  } else {
    constexpr int ES = BN + 1;  // floats per staged row (co), padded: a wave's 32 co rows hit 32 banks
    float* const stage = reinterpret_cast<float*>(lds);
    __syncthreads();  // the last slab's fragments have been read
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int col = wm * WTM + j * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[col * ES + wn * WN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc[i][j][r];
      }
    __syncthreads();
    for (int idx = tid; idx < BM * BN; idx += 256) {
      const int col = idx / BN, el = idx % BN;
      const int co = r0 + col, e = s0 + el;
      if (e < a.E && co < a.Cout) unsafeAtomicAdd(a.dw + (int64_t)co * a.ldw + e, stage[col * ES + el]);
    }
  }
  (void)y0;
}

extern int g_wgrad_buf;
template <int BN, bool SWAP, bool BUF>
int launch_b(WgArgs& a, int s_count, hipStream_t st);

// The split over positions.  Every workgroup of a launch walks the same number of slabs and `slots` of them run at a time (two per CU
// for the phase-serial tile, one for the pipelined one), so the launch lasts ceil(grid / slots) rounds of slabs_per_chunk slabs.  Rounds
// 3-5 took "enough chunks to fill the chip twice" = ceil(2 * slots / tiles) — which makes the grid tiles * chunks >= 2 * slots, and
// usually a few workgroups MORE: 1025, 1026, 1044 workgroups on 512 slots, 528 and 576 on 256 (res2 b, res3 b, res4 b, res4 a, res5 a
// at 8 items of config 5) are THREE rounds, the third all but empty — a third of those launches' time (round 6, found on paper:
// profiles/r06/wgrad_grid_rounds_model.txt).  Here: the chunk count at or below that figure with the least rounds * slabs; ties keep
// the MORE chunks (two rounds of half the length suffer half as much from a CU that another stream's kernel holds).
int g_wgrad_rounds = 1;  // (avt_wgrad_x3_set_xl(7) / (8): the round-aware choice off / on — A/Bs)
static int pick_chunks(int tiles, int nslab, int slots, int min_slabs, bool even) {
  int chunks = (2 * slots + tiles - 1) / tiles;
  const int most = nslab / min_slabs > 1 ? nslab / min_slabs : 1;
  if (chunks > most) chunks = most;
  if (chunks < 1) chunks = 1;
  if (!g_wgrad_rounds) return chunks;
  long best_cost = -1;
  int best = chunks;
  for (int c = chunks; c >= 1; --c) {
    int spc = (nslab + c - 1) / c;
    if (even) spc += spc & 1;
    const int nc = (nslab + spc - 1) / spc;
    const long grid = (long)tiles * nc;
    const long cost = ((grid + slots - 1) / slots) * spc;
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = c;
    }
    if (grid <= slots) break;  // (one round already: fewer chunks only lengthen it)
  }
  return best;
}

template <int BN, bool SWAP>
int launch(WgArgs& a, int s_count, hipStream_t st) {
  const bool fits32 = g_wgrad_buf && a.x_bytes != 0u && a.dy_bytes != 0u && a.ldx < (1 << 17) && a.ldy < (1 << 17);
  return fits32 ? launch_b<BN, SWAP, true>(a, s_count, st) : launch_b<BN, SWAP, false>(a, s_count, st);
}

template <int BN, bool SWAP, bool BUF>
int launch_b(WgArgs& a, int s_count, hipStream_t st) {
  a.s_tiles = (s_count + BN - 1) / BN;
  const int tiles = a.r_tiles * a.s_tiles;
  // split over positions: enough workgroups to fill the chip (256 CUs x 2), but at least 16 slabs each — a workgroup's
  // epilogue is 128 x BN atomics, and with 3 slabs apiece the pointwise layers spent their time there
  // (profiles/r02/probe_wgrad_v2.log: 16 TFLOP/s on 256 -> 1024 channels)
  const int chunks = pick_chunks(tiles, a.nslab, 512, 16, false);
  a.slabs_per_chunk = (a.nslab + chunks - 1) / chunks;
  a.nchunk = (a.nslab + a.slabs_per_chunk - 1) / a.slabs_per_chunk;
  constexpr int lds_small = 4 * kPlane + 2 * 64 * 4 + 2 * 64 * kMaxTaps * 4 + 64 * 4;  // (+ the tap table)
  constexpr int lds_big = 4 * kPlane + 2 * 64 * 4 + 2 * 64 * kMaxTapsBig * 4 + 64 * 4;
  static_assert(128 * (BN + 1) * 4 <= lds_small, "the swapped epilogue stages its tile over the operand planes (and tables)");
  static_assert(lds_small <= 80 * 1024, "two workgroups per CU");
  const int lds_bytes = a.tapcap > kMaxTaps ? lds_big : lds_small;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_kernel<BN, SWAP, BUF>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
  if (e != hipSuccess) {
    avt::set_error("avt_conv3d_wgrad_x3_f32: hipFuncSetAttribute(%d B LDS): %s", lds_big, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  const int64_t grid = (int64_t)tiles * a.nchunk;
  AVT_REQUIRE(grid < (1ll << 31), "avt_conv3d_wgrad_x3_f32: grid too large");
  hipLaunchKernelGGL((wgrad_x3_kernel<BN, SWAP, BUF>), dim3((unsigned)grid), dim3(256), lds_bytes, st, a);
  return avt::check_launch("avt_conv3d_wgrad_x3_f32");
}


// ---- the 256 x 128 tile (round 5) ---------------------------------------------------------------------------
// What bounds the 128-wide tile is not one thing: without its global loads, its LDS stage or its MFMAs it ran 1.7-1.8 x faster
// each (profiles/r03/probe_wgrad_phases_b120.log) — three phases that follow each other inside a workgroup (load -> barrier -> split
// + ds_write -> barrier -> fragments + MFMAs) and overlap only through the CU's second workgroup.  Here they overlap inside ONE
// workgroup of 8 waves: 32-position steps in TWO LDS stages (48 KB each); in step s a wave issues the global loads of step s + 2
// (into the register set that step s's data left free), multiplies step s out of its stage, and between its two k-slices splits
// and writes step s + 1's data — loaded during step s - 1 — into the other stage; one barrier per step, the position tables in a
// ring of four decoded three steps ahead.  256 x 128 outputs per workgroup (8 waves as 4 x 2, 64 x 64 each, the 128-wide tile's
// wave): 48 KB per 6.3 MFLOP instead of 64 KB.  Layers with >= 256 on the longer axis and >= 128 on the shorter, <= 28 taps.
// Its loads are BUFFER loads: the position tables hold byte offsets of rows, kRowOob for "no row" (padding taps, positions past the
// tensor or past the chunk) — a thread adds its channel's bytes and the hardware bounds check returns zeros: no per-piece mask
// registers, no selects in front of the ds_writes, 32-bit address arithmetic (the tile is VALU-bound: ~315 vector instructions per
// wave and 32-position step against 24 MFMAs, counted in the ISA).
constexpr int XP = 32;                 // positions per step
constexpr int XPL = XP * 256;          // one 128-channel plane of a stage: [32 positions][256 B]
constexpr int XSTAGE = 6 * XPL;        // R0 hi | R0 lo | R1 hi | R1 lo | S hi | S lo
constexpr int XTABS = 2 * XSTAGE;      // ytab [4][32], then xtab [4][32][kMaxTaps]

__device__ __forceinline__ void put4x(char* base, int off, float4 v) {  // put4 with this tile's plane pitch
  const uint32_t h01 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v.x, v.y}, bf16x2));
  const uint32_t h23 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v.z, v.w}, bf16x2));
  const float r0 = v.x - __builtin_bit_cast(float, h01 << 16), r1 = v.y - __builtin_bit_cast(float, h01 & 0xFFFF0000u);
  const float r2 = v.z - __builtin_bit_cast(float, h23 << 16), r3 = v.w - __builtin_bit_cast(float, h23 & 0xFFFF0000u);
  const uint32_t l01 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r0, r1}, bf16x2));
  const uint32_t l23 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r2, r3}, bf16x2));
  *reinterpret_cast<uint2*>(base + off) = make_uint2(h01, h23);
  *reinterpret_cast<uint2*>(base + XPL + off) = make_uint2(l01, l23);
}

// SW = width of the S axis' tile: 128 (waves as 4 (R) x 2 (S), 64 x 64 outputs each), 64 (4 x 2, 64 x 32) or 32 (8 x 1, 32 x 32).  The narrow
// forms are for the layers the 128-wide tile ran LATENCY-bound (64 -> 64 [1,3,3]: 4 us per 64-position slab of which 0.3 us are MFMAs;
// the 8- to 32-channel fast-pathway layers at 0.9 TB/s): the same pipeline, the S planes of a stage only partly used.
template <bool SWAP, int SW>
__global__ __launch_bounds__(512, 1) void wgrad_x3_xl_kernel(WgArgs a) {
  constexpr int RQ = 4;                        // float4 loads per thread and step on the R axis: 256 channels x 32 positions over 512 threads
  constexpr int SQT = SW / 4;                  // S quads per position (32 / 16 / 8) ...
  constexpr int SPP = 512 / SQT;               // ... positions one pass of the 512 threads covers (16 / 32 / 64)
  constexpr int SQ = SPP >= XP ? 1 : XP / SPP; // ... passes per step (2 / 1 / 1)
  constexpr int WS = SW >= 64 ? 2 : 1, WR = 8 / WS;       // waves along S / R
  constexpr int RWV = 256 / WR, SWV = SW / WS;            // a wave's outputs: 64 x 64, 64 x 32, 32 x 32
  constexpr int NI = SWV / 32, NJ = RWV / 32;             // 32 x 32 blocks of a wave along S / R
  extern __shared__ __attribute__((aligned(16))) char lds[];
  unsigned* const ytab = reinterpret_cast<unsigned*>(lds + XTABS);   // [4][32] byte offsets of dy rows
  unsigned* const xtab = ytab + 4 * XP;                              // [4][32][tapcap] ... of x rows under every tap
  int* const ttab = reinterpret_cast<int*>(xtab + 4 * XP * kMaxTaps); // [taps]: dt | dh << 8 | dw << 16 (see the 128-wide tile)
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WS, wn = wid % WS;
  const int lr = lane & 31, lh = lane >> 5;

  int b_ = avt::xcd_contiguous(blockIdx.x, gridDim.x);
  const int ts = b_ % a.s_tiles; b_ /= a.s_tiles;
  const int tr = b_ % a.r_tiles;
  const int chunk = b_ / a.r_tiles;
  const int r0 = tr * 256, s0 = ts * SW;
  const int st0 = chunk * a.slabs_per_chunk;                         // (steps of 32 positions here)
  const int st1 = min(a.nslab, st0 + a.slabs_per_chunk);
  const int x0 = SWAP ? s0 : r0;

  // this thread's pieces: R quad (tid & 63) of position (q * 8 + (tid >> 6)), S quad (tid & 31) of position (q * 16 + (tid >> 5));
  // the x operand's (tap, ci) is the same for all of a thread's pieces
  const int rcq = tid & 63, scq = tid % SQT, sp0 = tid / SQT;       // (S: position sp0 + SPP * pass; beyond the step for SW = 32's upper half)
  const int xe = x0 + 4 * (SWAP ? scq : rcq);
  const int xtap = xe < a.E ? (int)fastdiv((uint32_t)xe, a.dCin) : -1;
  const int xci = xe - xtap * a.Cin;
  const int ych = (SWAP ? r0 + 4 * rcq : s0 + 4 * scq);              // the dy operand's first channel
  const bool y_ok = ych < a.Cout;
  // bytes this thread adds to a row's offset on either axis; a column past the operand's end never loads (kRowOob, below)
  const unsigned rcb = 4u * (unsigned)(SWAP ? ych : xci), scb = 4u * (unsigned)(SWAP ? xci : ych);
  const bool r_ok = SWAP ? y_ok : xtap >= 0, s_ok = (SWAP ? xtap >= 0 : y_ok) && (SPP <= XP || sp0 < XP);

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto decode = [&](int step) {  // thread -> position tid & 31, taps (tid >> 5), + 16
    const int slot = step & 3, p = tid & 31, m = step * XP + p;
    const bool ok = m < a.M && step < st1;  // (a step past the chunk's end belongs to the next chunk's workgroup: zeros here)
    if (a.pointwise) {  // 1x1x1, stride 1, no padding: output position m reads input row m (half of the step's launches)
      if (tid < XP) {
        ytab[slot * XP + p] = ok ? 4u * (unsigned)(m * a.ldy) : kRowOob;
        xtab[(slot * XP + p) * a.tapcap] = ok ? 4u * (unsigned)(m * a.ldx) : kRowOob;
      }
      return;
    }
    const int q1 = (int)fastdiv((uint32_t)(ok ? m : 0), a.dWo), wo = (ok ? m : 0) - q1 * a.Wo;
    const int q2 = (int)fastdiv((uint32_t)q1, a.dHo), ho = q1 - q2 * a.Ho;
    const int bb = (int)fastdiv((uint32_t)q2, a.dTo), to = q2 - bb * a.To;
    if (tid < XP) ytab[slot * XP + p] = ok ? 4u * (unsigned)(m * a.ldy) : kRowOob;
    for (int tap = tid >> 5; tap < a.taps; tap += 16) {
      const int pk = ttab[tap], dt = pk & 255, dh = (pk >> 8) & 255, dw_ = pk >> 16;
      const int ti = to * a.st - a.pt + dt, hi = ho * a.sh - a.ph + dh, wi = wo * a.sw - a.pw + dw_;
      const bool in = ok && (unsigned)ti < (unsigned)a.T && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      xtab[(slot * XP + p) * a.tapcap + tap] = in ? 4u * (unsigned)((((bb * a.T + ti) * a.H + hi) * a.W + wi) * a.ldx) : kRowOob;
    }
  };

  i32x4 rq[2][RQ], sq[2][SQ];
  auto gload = [&](auto set_, int step) {  // all table reads first, then the loads, unconditional (see the 128-wide tile)
    constexpr int SET = decltype(set_)::value;
    const int slot = step & 3;
    unsigned rrow[RQ], srow[SQ];
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int p = q * 8 + (tid >> 6);
      rrow[q] = SWAP ? ytab[slot * XP + p] : xtab[(slot * XP + p) * a.tapcap + (xtap < 0 ? 0 : xtap)];
    }
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
      const int p = (q * SPP + sp0) & (XP - 1);
      srow[q] = SWAP ? xtab[(slot * XP + p) * a.tapcap + (xtap < 0 ? 0 : xtap)] : ytab[slot * XP + p];
    }
#pragma unroll
    for (int q = 0; q < RQ; ++q)
      rq[SET][q] = __builtin_amdgcn_raw_buffer_load_b128(SWAP ? rsy : rsx, (int)(r_ok ? rrow[q] + rcb : kRowOob), 0, 0);
#pragma unroll
    for (int q = 0; q < SQ; ++q)
      sq[SET][q] = __builtin_amdgcn_raw_buffer_load_b128(SWAP ? rsx : rsy, (int)(s_ok ? srow[q] + scb : kRowOob), 0, 0);
  };
  const int rsub = (rcq >> 5) * 2 * XPL, rc = rcq & 31;
  auto lstore = [&](auto set_, char* stg) {
    constexpr int SET = decltype(set_)::value;
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int p = q * 8 + (tid >> 6);
      put4x(stg + rsub, swz(p, rc >> 1) + 8 * (rc & 1), __builtin_bit_cast(float4, rq[SET][q]));
    }
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
      const int p = q * SPP + sp0;
      if (SPP <= XP || p < XP) put4x(stg + 4 * XPL, swz(p, scq >> 1) + 8 * (scq & 1), __builtin_bit_cast(float4, sq[SET][q]));
    }
  };
  // transposed-read addresses inside a stage (k-slice 0; + 4096 per 16 positions): see the 128-wide tile
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  int raddr0_[NJ][2], saddr0_[NI][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int rofs = wm * RWV + j * 32;  // first R channel of the block: plane pair rofs >> 7, channel rofs & 127 inside it
      raddr0_[j][h2] = (rofs >> 7) * 2 * XPL + swz(8 * (g >> 1) + 4 * h2 + tq, ((rofs & 127) + 16 * (g & 1)) / 8 + (tp >> 1)) + 8 * (tp & 1);
    }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
      saddr0_[i][h2] = 4 * XPL + swz(8 * (g >> 1) + 4 * h2 + tq, (wn * SWV + i * 32 + 16 * (g & 1)) / 8 + (tp >> 1)) + 8 * (tp & 1);
  struct Frags {
    i32x4 rh[NJ], rl[NJ], sh[NI], sl[NI];
  };
  // (the second stage's addresses in registers of their own: stage base + plane + k-slice exceeds the 16-bit offset field of a
  //  ds_read, and the compiler spent one v_add per transposing read — 32 per step — on them)
  int raddr1[NJ][2], saddr1[NI][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) raddr1[j][h2] = raddr0_[j][h2] + XSTAGE;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) saddr1[i][h2] = saddr0_[i][h2] + XSTAGE;
  auto fload = [&](Frags& f, auto par_, int ks) {
    constexpr int P = decltype(par_)::value;
    const char* const stg = lds;
    auto& raddr = *(P ? &raddr1 : &raddr0_);
    auto& saddr = *(P ? &saddr1 : &saddr0_);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto frag = [&](int a0, int a1) {
      const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(stg + a0)));
      const uint2 v = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(stg + a1)));
      return i32x4{(int)u.x, (int)u.y, (int)v.x, (int)v.y};
    };
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f.rh[j] = frag(raddr[j][0] + 4096 * ks, raddr[j][1] + 4096 * ks);
      f.rl[j] = frag(raddr[j][0] + XPL + 4096 * ks, raddr[j][1] + XPL + 4096 * ks);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f.sh[i] = frag(saddr[i][0] + 4096 * ks, saddr[i][1] + 4096 * ks);
      f.sl[i] = frag(saddr[i][0] + XPL + 4096 * ks, saddr[i][1] + XPL + 4096 * ks);
    }
  };
  auto fmul = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[i][j] = mfma(f.sl[i], f.rh[j], acc[i][j]);  // small terms first
        acc[i][j] = mfma(f.sh[i], f.rl[j], acc[i][j]);
        acc[i][j] = mfma(f.sh[i], f.rh[j], acc[i][j]);
      }
  };
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;
  auto step = [&](auto par_, int s) {  // par = (s - st0) & 1: stage and register set of step s
    constexpr int P = decltype(par_)::value;
    char* nxt = lds + (P ^ 1) * XSTAGE;
    __syncthreads();  // stage P is complete; the other stage's readers (step s - 1) are done; the table of step s + 2 is decoded
    Frags f0, f1;
    fload(f0, par_, 0);
    fload(f1, par_, 1);
    __builtin_amdgcn_sched_barrier(0);
    fmul(f0);  // (issued first: the table reads, address arithmetic and the split + ds_write below run under the matrix pipe)
    // NO branch around the loads, the decode or the stores (steps past the chunk's end are masked to zeros instead): a load
    // under a condition makes the compiler count the loads in flight for the path that skipped it, and the s_waitcnt in front of
    // the ds_writes of step s + 1's data then ALSO waits for the loads of step s + 2 issued a moment ago — the two-step
    // prefetch distance was one step minus the second k-slice's MFMAs (vmcnt(5..0) where vmcnt(11..6) was meant)
    gload(std::integral_constant<int, P>{}, s + 2);
    decode(s + 3);
    __builtin_amdgcn_sched_barrier(0);
    fmul(f1);
    lstore(std::integral_constant<int, P ^ 1>{}, nxt);  // step s + 1's data, requested during step s - 1
  };

  if (tid < a.taps) {
    const int t1 = (int)fastdiv((uint32_t)tid, a.dKW), dw_ = tid - t1 * a.KW;
    const int dt = (int)fastdiv((uint32_t)t1, a.dKH), dh = t1 - dt * a.KH;
    ttab[tid] = dt | (dh << 8) | (dw_ << 16);
  }
  __syncthreads();
  decode(st0);
  decode(st0 + 1);
  decode(st0 + 2);
  __syncthreads();
  gload(S0{}, st0);
  gload(S1{}, st0 + 1);
  lstore(S0{}, lds);
  for (int s = st0; s < st1; s += 2) {  // in pairs: an odd chunk's last pair multiplies one step of zeros
    step(S0{}, s);
    step(S1{}, s + 1);
  }

  // D layout: column (lane & 31) = index on the R axis, rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = index on the S axis
  if constexpr (!SWAP) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int e = r0 + wm * RWV + j * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = s0 + wn * SWV + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (e < a.E && co < a.Cout) unsafeAtomicAdd(a.dw + (int64_t)co * a.ldw + e, acc[i][j][r]);
        }
      }
  } else {  // lanes would walk co (rows 4 E bytes apart): through the LDS, one half of the R axis (128 co) at a time
    constexpr int ES = SW + 1;
    float* const stage = reinterpret_cast<float*>(lds);
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int rofs = wm * RWV + j * 32;
        if ((rofs >> 7) == half) {
          const int col = (rofs & 127) + lr;
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[col * ES + wn * SWV + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc[i][j][r];
        }
      }
      __syncthreads();
      for (int idx = tid; idx < 128 * SW; idx += 512) {
        const int col = idx / SW, el = idx % SW;
        const int co = r0 + half * 128 + col, e = s0 + el;
        if (e < a.E && co < a.Cout) unsafeAtomicAdd(a.dw + (int64_t)co * a.ldw + e, stage[col * ES + el]);
      }
    }
  }
}

template <bool SWAP, int SW>
int launch_xl(WgArgs& a, int r_count, int s_count, hipStream_t st) {
  a.r_tiles = (r_count + 255) / 256;
  a.s_tiles = (s_count + SW - 1) / SW;
  const int tiles = a.r_tiles * a.s_tiles;
  a.nslab = (a.M + XP - 1) / XP;  // steps of 32 positions
  // one workgroup per CU; at least 32 steps each (the epilogue is 256 x 128 atomics)
  const int chunks = pick_chunks(tiles, a.nslab, 256, 32, true);
  a.slabs_per_chunk = (a.nslab + chunks - 1) / chunks;
  a.slabs_per_chunk += a.slabs_per_chunk & 1;  // (the kernel walks its steps in pairs)
  a.nchunk = (a.nslab + a.slabs_per_chunk - 1) / a.slabs_per_chunk;
  constexpr int lds_bytes = XTABS + 4 * XP * 4 + 4 * XP * kMaxTaps * 4 + 64 * 4;  // (+ the tap table)
  static_assert(128 * 129 * 4 <= XTABS && lds_bytes <= 160 * 1024, "the swapped epilogue's staging and the whole layout fit");
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_xl_kernel<SWAP, SW>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("avt_conv3d_wgrad_x3_f32: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  const int64_t grid = (int64_t)tiles * a.nchunk;
  AVT_REQUIRE(grid < (1ll << 31), "avt_conv3d_wgrad_x3_f32: grid too large");
  hipLaunchKernelGGL((wgrad_x3_xl_kernel<SWAP, SW>), dim3((unsigned)grid), dim3(512), lds_bytes, st, a);
  return avt::check_launch("avt_conv3d_wgrad_x3_f32");
}

// which tile avt_conv3d_wgrad_x3_f32 launches for (longer axis, shorter axis, taps, positions): 1 = the 256 x 128 tile
int g_wgrad_xl = 1;  // (avt_wgrad_x3_set_xl: A/B switch for tools and tests; the shipped default is on)
int g_wgrad_buf = 1;     // (avt_wgrad_x3_set_xl(5) / (6): the phase-serial tile's buffer-load form off / on — off is what tensors of 4 GB and more get)
int g_wgrad_narrow = 1;  // (avt_wgrad_x3_set_xl(3) / (4): the 64- and 32-wide pipelined forms on / off under mode 1)
// g_wgrad_xl: 1 = where it measured faster; 2 = every layer, at the S width that fits (tests, probes); 0 = never
// -> the S width of the pipelined tile to launch (128 / 64 / 32), or 0 for the 128-wide phase-serial tile
static int wgrad_xl_picked(bool swap, int r_count, int s_count, int taps, int spatial_taps, int m, bool fits32) {
  if (!(taps <= kMaxTaps && m >= 64 * XP && fits32) || g_wgrad_xl == 0) return 0;
  const int sw = s_count > 64 ? 128 : (s_count > 32 ? 64 : 32);
  if (g_wgrad_xl >= 2) return sw;
  // mode 1, by profiles/r05/probe_wgrad_xl.log (its last table, both tiles on buffer loads): the pipelined tile wins on the
  // pointwise and temporal-tap layers (+3 ... +31 %) and is equal or behind on the [1,3,3] ones (0 ... -2 %; 64 -> 64 at 56 x 56:
  // -22 %) and on the short-axis-heavy 128 x 384 layer (-7 %); its 32-wide form loses on the 8-channel layers: tests / probes only
  if (spatial_taps != 1) return 0;
  if (sw == 128) return r_count >= 512 ? 128 : 0;
  return (g_wgrad_narrow && sw == 64 && r_count >= 224) ? 64 : 0;
}

template <bool SWAP>
int dispatch(WgArgs& a, int r_count, int s_count, hipStream_t s) {
  const int xl = wgrad_xl_picked(SWAP, r_count, s_count, a.taps, a.KH * a.KW, a.M, a.x_bytes != 0u && a.dy_bytes != 0u && a.ldx < (1 << 17) && a.ldy < (1 << 17));
  if (xl == 128) return launch_xl<SWAP, 128>(a, r_count, s_count, s);
  if (xl == 64) return launch_xl<SWAP, 64>(a, r_count, s_count, s);
  if (xl == 32) return launch_xl<SWAP, 32>(a, r_count, s_count, s);
  a.r_tiles = (r_count + 127) / 128;
  if (s_count <= 32) return launch<32, SWAP>(a, s_count, s);
  if (s_count <= 64) return launch<64, SWAP>(a, s_count, s);
  return launch<128, SWAP>(a, s_count, s);
}

}  // namespace

// The general entry: explicit output extent (to, ho, wo; 0 = the symmetric-padding formula), any pt / ph / pw (a slice of a
// longer filter is the same convolution with a shifted padding), dW rows ldw elements apart, zeroed here or by the caller.
extern "C" int avt_conv3d_wgrad_x3_sub_f32(const float* dy, const float* x, float* dw, int batch, int t, int h, int w, int cin, int cout,
                                           int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho, int wo,
                                           int ldx, int ldy, int ldw, int zero_dw, void* stream) {
  AVT_REQUIRE(dy && x && dw, "avt_conv3d_wgrad_x3_f32: NULL pointer");
  AVT_REQUIRE(cin > 0 && cin % 4 == 0 && cout > 0 && cout % 8 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= cin && ldy >= cout,
              "avt_conv3d_wgrad_x3_f32: input channels in multiples of 4, output channels of 8, row strides in multiples of 4 covering them");
  AVT_REQUIRE(kt >= 1 && kh >= 1 && kw >= 1 && kt * kh * kw <= kMaxTapsBig && st >= 1 && sh >= 1 && sw >= 1,
              "avt_conv3d_wgrad_x3_f32: 1..%d kernel taps, strides >= 1", kMaxTapsBig);
  AVT_REQUIRE(avt::aligned16(dy) && avt::aligned16(x) && avt::aligned16(dw), "avt_conv3d_wgrad_x3_f32: pointers must be 16-byte aligned");
  WgArgs a = {};
  a.dy = dy; a.x = x; a.dw = dw;
  a.B = batch; a.T = t; a.H = h; a.W = w;
  a.To = to > 0 ? to : (t + 2 * pt - kt) / st + 1;
  a.Ho = ho > 0 ? ho : (h + 2 * ph - kh) / sh + 1;
  a.Wo = wo > 0 ? wo : (w + 2 * pw - kw) / sw + 1;
  AVT_REQUIRE(batch > 0 && a.To > 0 && a.Ho > 0 && a.Wo > 0, "avt_conv3d_wgrad_x3_f32: empty output");
  const int64_t M = (int64_t)batch * a.To * a.Ho * a.Wo;
  AVT_REQUIRE(M < (1ll << 31) - 64 && M * ldy < (1ll << 31) && (int64_t)batch * t * h * w * ldx < (1ll << 31),
              "avt_conv3d_wgrad_x3_f32: tensors too large for 32-bit element offsets");
  a.KT = kt; a.KH = kh; a.KW = kw; a.st = st; a.sh = sh; a.sw = sw; a.pt = pt; a.ph = ph; a.pw = pw;
  a.Cin = cin; a.Cout = cout; a.ldx = ldx; a.ldy = ldy; a.M = (int)M; a.taps = kt * kh * kw;
  a.E = a.taps * cin;
  a.tapcap = a.taps <= kMaxTaps ? kMaxTaps : kMaxTapsBig;
  a.ldw = ldw > 0 ? ldw : a.E;
  AVT_REQUIRE(a.ldw >= a.E, "avt_conv3d_wgrad_x3_f32: ldw (%d) < taps * cin (%d)", a.ldw, a.E);
  a.dCin = make_fastdiv((uint32_t)cin);
  a.pointwise = (a.taps == 1 && st == 1 && sh == 1 && sw == 1 && pt == 0 && ph == 0 && pw == 0 && a.To == t && a.Ho == h && a.Wo == w) ? 1 : 0;
  {
    const int64_t xb = (int64_t)batch * t * h * w * ldx * 4, yb = M * ldy * 4;
    a.x_bytes = xb < (int64_t)kRowOob ? (unsigned)xb : 0u;  // (0: too large for the pipelined tile's 32-bit byte offsets)
    a.dy_bytes = yb < (int64_t)kRowOob ? (unsigned)yb : 0u;
  }
  a.nslab = (int)((M + 63) / 64);
  a.dWo = make_fastdiv((uint32_t)a.Wo); a.dHo = make_fastdiv((uint32_t)a.Ho); a.dTo = make_fastdiv((uint32_t)a.To);
  a.dKW = make_fastdiv((uint32_t)kw); a.dKH = make_fastdiv((uint32_t)kh);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (zero_dw) {
    AVT_REQUIRE(a.ldw == a.E, "avt_conv3d_wgrad_x3_f32: a slice of a longer dW is zeroed by the caller (zero_dw = 0)");
    if (hipMemsetAsync(dw, 0, sizeof(float) * (size_t)cout * a.E, s) != hipSuccess) {
      avt::set_error("avt_conv3d_wgrad_x3_f32: hipMemsetAsync failed");
      return AVT_ERR_LAUNCH;
    }
  }
  // the longer axis — x's (tap, ci) or dy's co — takes the 128-wide side of the tile
  if (cout > a.E) return dispatch<true>(a, cout, a.E, s);
  return dispatch<false>(a, a.E, cout, s);
}

// 1 (default): the layers the 256 x 128 pipelined tile measured faster on take it; 2: every layer it can take; 0: none (A/Bs, tests)
extern "C" int avt_wgrad_x3_set_xl(int on) {
  const int was = g_wgrad_xl;
  if (on == 3 || on == 4) {  // (probes: mode 1 with / without the narrow pipelined forms)
    g_wgrad_xl = 1;
    g_wgrad_narrow = on == 3 ? 1 : 0;
    return was;
  }
  if (on == 5 || on == 6) {  // (tests: the phase-serial tile without / with buffer loads; the tile mode stays)
    g_wgrad_buf = on == 6 ? 1 : 0;
    return was;
  }
  if (on == 7 || on == 8) {  // (A/Bs: the round-aware split over positions off / on; the tile mode stays)
    g_wgrad_rounds = on == 8 ? 1 : 0;
    return was;
  }
  g_wgrad_xl = on < 0 ? 0 : (on > 2 ? 2 : on);
  return was;
}

extern "C" int avt_conv3d_wgrad_x3_f32(const float* dy, const float* x, float* dw, int batch, int t, int h, int w, int cin, int cout,
                                       int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int ldx, int ldy,
                                       void* stream) {
  AVT_REQUIRE(cin % 8 == 0 && kt * kh * kw <= kMaxTaps, "avt_conv3d_wgrad_x3_f32: channel counts in multiples of 8, at most %d taps "
              "(avt_conv3d_wgrad_x3_sub_f32 takes 4-channel inputs and [1,7,7] filters)", kMaxTaps);
  return avt_conv3d_wgrad_x3_sub_f32(dy, x, dw, batch, t, h, w, cin, cout, kt, kh, kw, st, sh, sw, pt, ph, pw, 0, 0, 0, ldx, ldy, 0, 1,
                                     stream);
}
