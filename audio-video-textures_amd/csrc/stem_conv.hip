// stem_conv — the SlowFast stem convolutions (pixel-pair form) with the input patch resident in LDS.
// Same arithmetic as conv3d_igemm on the packed stem weights of fused_slowfast.stem_conv (Conv3d(3, C, [kt,7,7],
// stride [1,2,2], pad [kt//2,3,3]) + BN + ReLU of the third-party SlowFast model the reference calls at
// contrastive_video_textures/models/models.py:335, 399), for the production shape (224^2 clips).
//
// Why a second kernel: as an implicit GEMM the stem gathers every input element once per kernel tap — 16x (slow,
// [1,7,4] pair taps) to 28x (time-grouped fast stem, [8,7,4]) its own size through L2/L1 and the LDS write port —
// and that gather, not the MFMA work, bounds conv3d_igemm on these two layers (18 TB/s of L2 reads at 23 % MFMA
// issue).  Here a workgroup owns 4 full output rows of one (clip, output frame): per input frame it stages the
// 13 input rows those outputs touch ONCE (24 KB), and the MFMA operand of every (row tap dh, pair tap dp) is read
// straight from that patch: with v_mfma_f32_16x16x32_bf16 one instruction's K = 32 is exactly the four pair taps x 8
// channels of one (dt, dh), and a lane's operand is the 16-byte chunk at patch[(2r + dh), wo + dp] — 16 consecutive
// output columns x 4 taps read overlapping consecutive chunks (conflict-free ds_read_b128, broadcast on overlap).
// No im2col staging, no tap table, 5-8x fewer bytes gathered per output.
//
// Layout: in  [B, T, H, PW, 8] bf16 (the channels-last clip [B,T,H,W,4] read as pixel pairs, PW = W/2),
//         wt  [Cout/32, KT, 7, 2, 4, 16, 8] bf16 = the LDS image of each 32-channel group's frame-tap slab (BN folded;
//             Cout = frames-per-group x channels for the time-grouped form), see include/avt.h,
//         out [B, To, Ho, Wo, Cout] bf16, Ho = H/2, Wo = PW; temporal stride st, temporal pad pt.
// One wave per output row; the weights are the first MFMA operand (D = Wt * A^T) with the channel order permuted in
// the LDS image so that a lane ends up with 8 consecutive channels of one position -> one 16-byte store, and the
// four lane groups of an instruction complete a 64-byte row.  Input frames outside [0, T) are skipped, not zero-
// filled (the time-grouped form has 8 frame taps of which 2-3 are padding at the clip ends).
// Roofline: MFMA (structured zeros of the pixel-pair / time-group forms included in the issued work).
#include <stdlib.h>

#include "avt_common.h"
#include "split_planes.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned kOob = 0xFFFFFFF0u;

// PL = 0: bf16 (one plane per tensor).  PL = 1 / 2: the contract-grade split-plane arithmetic (conv_x3.hip) with bf16 /
// fp16 planes: the patch and the weight slab are staged for both planes and every product is three MFMAs.
template <int PL>
__device__ __forceinline__ f32x4 mfma16(i32x4 w, i32x4 x, f32x4 c) {
  if constexpr (PL == 2)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}

struct StemArgs {
  const uint16_t* in;
  const uint16_t* wt;
  const float* bias;
  uint16_t* out;
  int T, H, PW;
  int To, Ho;
  int KT, st, pt;
  int Cout;
  int relu;
  unsigned in_bytes, wt_bytes;
  // fused max-pool (POOL): out is the pooled tensor [B, To*tgroup, Ho/2, Wo/2, .] with row stride ldo (elements)
  int tgroup, ldo;
  int ncg;  // 32-channel groups
  int swz;  // XCD-aware work order (always on)
  // split-plane form (PL > 0): the low-order planes, same geometry, and the fp16 planes' per-channel weight scale
  const uint16_t* in_lo;
  const uint16_t* wt_lo;
  uint16_t* out_lo;
  const float* wscale;
  // training forward (avt_stem_conv_x3_f32): `out` is an fp32 NDHWC tensor [B, To * tgroup, Ho, Wo, Cout / tgroup] — the
  // time-grouped channels go back to their frames on the way out
  int out_f32;
  // frame table (avt_stem_conv_x3 with frame_idx): `in` holds n_table distinct frames [n_table, H, PW, 8] and input frame t of clip b
  // is table frame fidx[b * T + t] — clip windows that overlap (W / S of their frames each) and the fast pathway's repeated
  // frames are then packed ONCE instead of once per (window, slot).  nullptr = the dense clip tensor [B, T, H, PW, 8]
  const int32_t* fidx;
  // merged frame taps (avt_stem_conv_x3_merged): the fast pathway samples 32 slots from a W-frame window with
  // linspace(0, W-1, 32).long() — for W = 20 only 5 of the 8 slots an output-frame group meets are DISTINCT source frames.
  // Conv is linear in the weights, so the taps that read one source frame are summed on the host: group `to` walks ktm merged
  // taps, tap j reads table frame mtab[(b * To + to) * ktm + j] (-1 = no further tap) with the weight slab (to * ktm + j) of the
  // image, and bit n of mact[to * ktm + j] says whether tile n's rows of that slab are non-zero.  nullptr = one tap per slot.
  const int32_t* mtab;
  const int32_t* mact;
  int ktm;
  // frame-major tiles of the time-grouped form (non-pooled split-plane entries): tile n of a 32-channel group holds output
  // frames 2n, 2n + 1 (8 channels each), so it meets frame taps 2n .. 2n + kt0 only (kt0 = the convolution's own frame taps)
  // and the other (tap, tile) pairs — structural zeros of the block-Toeplitz weights, 4 of 16 for [5,7,7] — are skipped.
  // 0 = the classic image (a lane's two accumulators are 8 consecutive channels)
  int fm_kt0;
};

constexpr int RB = 4;   // conv rows a workgroup owns in the plain form (pooled: RBP = 8, plus one recomputed row above them)
constexpr int RBP = 8;
constexpr int NT = 2;             // 16-channel tiles per workgroup (32 output channels; blockIdx.y walks the rest)
constexpr int KF = 7 * 4 * 8;     // K per input frame: 7 row taps x 4 pair taps x 8 (pixel-in-pair, channel)
constexpr int BCH = NT * 16 * 28;  // 16-byte weight chunks per frame

__device__ __forceinline__ uint32_t max2(uint32_t x, uint32_t y) {  // packed bf16x2 max
  const uint32_t lo = (avt::bf16x2_lo(y) > avt::bf16x2_lo(x)) ? (y & 0xffffu) : (x & 0xffffu);
  const uint32_t hi = (avt::bf16x2_hi(y) > avt::bf16x2_hi(x)) ? (y & 0xffff0000u) : (x & 0xffff0000u);
  return lo | hi;
}

// POOL: the workgroup computes conv rows 8hg-1 .. 8hg+7 (the first is recomputed by its upper neighbour's lower edge)
// and writes the four MaxPool3d((1,3,3),(1,2,2),(0,1,1)) rows 4hg .. 4hg+3 they complete — the conv output never
// reaches HBM.  Needs ReLU (values >= 0, so the pool's padding can be 0).
// Work split: the R x MT (conv row, 16-position tile) units are dealt out evenly to the waves (plain: 4 rows x 7 tiles
// on 4 waves = one row each; pooled: 9 x 7 = 63 units on 8 waves) — the kernel is MFMA-bound, and one wave per row
// with 5 or 9 rows leaves one SIMD with twice the work of the others (measured 2.4x slower than stem + pool).
constexpr int RBPX = 4;  // (split-plane instantiations never pool; kept for the shared index arithmetic)
template <int MT, bool POOL, int PL = 0>
__global__ __launch_bounds__(POOL ? 512 : 256, POOL ? 4 : (PL ? 2 : 3)) void stem_kernel(StemArgs a) {
  constexpr int OWNP = PL ? RBPX : RBP;
  constexpr int R = POOL ? OWNP + 1 : RB;  // conv rows computed
  constexpr int OWN = POOL ? OWNP : RB;    // conv rows owned
  constexpr int NWV = POOL ? 8 : 4;
  constexpr int NTHR = NWV * 64;
  constexpr int TPW = (R * MT + NWV - 1) / NWV;  // (row, tile) units per wave
  constexpr int PROWS = 2 * R + 5;  // input rows 2*r0-3 .. 2*(r0+R-1)+3
  constexpr int BU = (BCH + NTHR - 1) / NTHR;
  constexpr int WO = MT * 16;
  constexpr int PWP = WO + 4;  // patch row: pairs -2 .. WO+1
  constexpr int PCH = PROWS * PWP;
  constexpr int PU = (PCH + NTHR - 1) / NTHR;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* lp = lds;             // patch [PROWS][PWP] x 16 B
  char* lb = lds + PCH * 16;  // weights [7 dh][NT][4 dp][16 rows] x 16 B
  char* lpl = lb + BCH * 16;  // (PL) the low-order planes of both
  char* lbl = lpl + PCH * 16;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = POOL ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;  // pooled: the unit offsets below stay in SGPRs
  const int l15 = lane & 15, q = lane >> 4;
  const int hgroups = a.Ho / OWN;
  // XCD-aware order: workgroup ids round-robin over the 8 XCDs, so give every XCD a CONTIGUOUS range of the work list
  // (channel group fastest, then row group, output frame, clip): the workgroups that share input — the channel groups of
  // one patch, the row groups above / below (5 halo rows of 13), the neighbouring output frames of the time-grouped stem
  // (4 of 8 frame taps) — then run on one XCD at about the same time and meet in its L2.  Without it the fast stem
  // fetched 5.8 GB per 128 clips for 2.5 GB algorithmic (PMC) and ran at the HBM roof instead of the MFMA one.
  const int nblk = gridDim.x, xc = blockIdx.x % 8, qd = nblk / 8, rmd = nblk % 8;
  int bid = a.swz ? (xc < rmd ? xc * (qd + 1) : rmd * (qd + 1) + (xc - rmd) * qd) + blockIdx.x / 8 : (int)blockIdx.x;
  const int cgi = bid % a.ncg;
  bid /= a.ncg;
  const int hg = bid % hgroups;
  bid /= hgroups;
  const int to = bid % a.To, b = bid / a.To;
  const int ho0 = hg * OWN - (POOL ? 1 : 0);  // first conv row computed (may be -1: its result is unused)
  const int n_base = cgi * (NT * 16);

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rinl = __builtin_amdgcn_make_buffer_rsrc((void*)(PL ? a.in_lo : a.in), 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwtl = __builtin_amdgcn_make_buffer_rsrc((void*)(PL ? a.wt_lo : a.wt), 0, a.wt_bytes, 0x00020000);

  // per-thread patch chunks: byte offset inside a frame, or out of bounds (padding rows / pairs)
  unsigned poff[PU];
#pragma unroll
  for (int u = 0; u < PU; ++u) {
    const int c = tid + NTHR * u;
    const int j = c / PWP, col = c - j * PWP;
    const int hi = 2 * ho0 - 3 + j, wi = col - 2;
    const bool ok = c < PCH && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.PW;
    poff[u] = ok ? (unsigned)((hi * a.PW + wi) * 16) : kOob;
  }
  // weights arrive in the LDS image order [group of 32 channels][frame tap][7 dh][NT][4 dp][16 rows] x 16 B (host-side
  // repack, see include/avt.h): staging them is a linear copy — coalesced global reads, conflict-free ds_write_b128.
  // Row 4q'+i of tile nt holds channel 8q' + 4nt + i of the group, so a lane's two accumulators are 8 consecutive channels.
  const unsigned gbase = a.mtab ? (unsigned)((cgi * a.To + to) * a.ktm) * (unsigned)(BCH * 16) : (unsigned)(cgi * a.KT) * (unsigned)(BCH * 16);
  const int32_t* mrow = a.mtab ? a.mtab + (int64_t)(b * a.To + to) * a.ktm : nullptr;

  f32x4 acc[TPW][NT];
  const int lane_off = (l15 + q) * 16;
  int ubase[TPW];  // this wave's units (conv row 0 .. R-1, tile): patch byte offset of tap (0, 0)
  auto unit_row = [&](int i) {
    const int u = w * TPW + i;
    return (u < R * MT ? u : R * MT - 1) / MT;  // (a surplus unit repeats the last one; its result is not written)
  };
  auto unit_mt = [&](int i) {
    const int u = w * TPW + i;
    return (u < R * MT ? u : R * MT - 1) % MT;
  };
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    ubase[i] = ((2 * unit_row(i)) * PWP + unit_mt(i) * 16) * 16;
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // frame taps that fall inside the clip
  const int t0 = to * a.st - a.pt;
  int dt_lo = t0 < 0 ? -t0 : 0;
  int dt_hi = (a.T - t0) < a.KT ? (a.T - t0) : a.KT;
  if (mrow) {  // merged taps: 0 .. the first -1 (uniform)
    dt_lo = 0;
    dt_hi = 0;
    while (dt_hi < a.ktm && mrow[dt_hi] >= 0) ++dt_hi;
  }

  constexpr int NPL = PL ? 2 : 1;
  i32x4 rp[NPL][PU], rb[NPL][BU];
  auto gload = [&](int dt) {
    const int fr = b * a.T + t0 + dt;
    const unsigned fbase = (unsigned)(((mrow ? mrow[dt] : (a.fidx ? a.fidx[fr] : fr)) * a.H) * a.PW) * 16u;
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const unsigned off = poff[u] == kOob ? kOob : fbase + poff[u];
      rp[0][u] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)off, 0, 0);
      if constexpr (PL != 0) rp[1][u] = __builtin_amdgcn_raw_buffer_load_b128(rinl, (int)off, 0, 0);
    }
    const unsigned kb = gbase + (unsigned)dt * (unsigned)(BCH * 16);
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const int e = tid + NTHR * u;
      const int off = (int)(e < BCH ? kb + (unsigned)e * 16u : kOob);
      rb[0][u] = __builtin_amdgcn_raw_buffer_load_b128(rwt, off, 0, 0);
      if constexpr (PL != 0) rb[1][u] = __builtin_amdgcn_raw_buffer_load_b128(rwtl, off, 0, 0);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int c = tid + NTHR * u;
      if (c < PCH) {
        *reinterpret_cast<i32x4*>(lp + c * 16) = rp[0][u];
        if constexpr (PL != 0) *reinterpret_cast<i32x4*>(lpl + c * 16) = rp[1][u];
      }
    }
#pragma unroll
    for (int u = 0; u < BU; ++u)
      if (tid + NTHR * u < BCH) {
        *reinterpret_cast<i32x4*>(lb + (tid + NTHR * u) * 16) = rb[0][u];
        if constexpr (PL != 0) *reinterpret_cast<i32x4*>(lbl + (tid + NTHR * u) * 16) = rb[1][u];
      }
  };

  // plain: the next frame's patch is fetched into registers under this frame's MFMAs.  pooled: no register prefetch
  // (its 32 VGPRs would cost the second resident workgroup, which is what hides the fetch there; the slow stem it is
  // used for has a single frame tap anyway)
  if (!POOL && dt_lo < dt_hi) gload(dt_lo);
  for (int dt = dt_lo; dt < dt_hi; ++dt) {
    if (POOL) gload(dt);
    lstore();
    __syncthreads();
    if (!POOL && dt + 1 < dt_hi) gload(dt + 1);  // in flight under this frame's MFMAs
    bool act[NT];  // uniform
#pragma unroll
    for (int n = 0; n < NT; ++n)
      act[n] = mrow ? ((a.mact[to * a.ktm + dt] >> n) & 1) != 0 : (POOL || PL == 0 || !a.fm_kt0 || (dt >= 2 * n && dt <= 2 * n + a.fm_kt0));
#pragma unroll
    for (int dh = 0; dh < 7; ++dh) {
      i32x4 bf[NT], bfl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int o = (((dh * NT + n) * 4 + q) * 16 + l15) * 16;
        bf[n] = *reinterpret_cast<const i32x4*>(lb + o);
        if constexpr (PL != 0) bfl[n] = *reinterpret_cast<const i32x4*>(lbl + o);
      }
      const int prow = ((2 * w + dh) * PWP + l15 + q) * 16;  // plain: unit i = tile i of row w
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int o = POOL ? lane_off + ubase[i] + dh * (PWP * 16) : prow + i * 256;
        const i32x4 af = *reinterpret_cast<const i32x4*>(lp + o);
        if constexpr (PL != 0) {
          const i32x4 afl = *reinterpret_cast<const i32x4*>(lpl + o);
#pragma unroll
          for (int n = 0; n < NT; ++n) {  // small terms first: wl*ah + wh*al + wh*ah
            if (!act[n]) continue;
            acc[i][n] = mfma16<PL>(bfl[n], af, acc[i][n]);
            acc[i][n] = mfma16<PL>(bf[n], afl, acc[i][n]);
            acc[i][n] = mfma16<PL>(bf[n], af, acc[i][n]);
          }
        } else {
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[i][n] = mfma16<0>(bf[n], af, acc[i][n]);  // D[channel][position]
        }
      }
    }
    __syncthreads();
  }

  // epilogue: lane = position (tile*16 + l15) of conv row ho0 + urow, channels n_base + 8q .. +7
  const int c0 = n_base + 8 * q;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (a.bias && c0 < a.Cout) {
    b0 = *reinterpret_cast<const float4*>(a.bias + c0);
    b1 = *reinterpret_cast<const float4*>(a.bias + c0 + 4);
  }
  float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0;
  if (PL != 0 && a.wscale && c0 < a.Cout) {
    s0 = *reinterpret_cast<const float4*>(a.wscale + c0);
    s1 = *reinterpret_cast<const float4*>(a.wscale + c0 + 4);
  }
  auto values = [&](int m, float* v) {
    v[0] = acc[m][0][0] * s0.x + b0.x; v[1] = acc[m][0][1] * s0.y + b0.y; v[2] = acc[m][0][2] * s0.z + b0.z; v[3] = acc[m][0][3] * s0.w + b0.w;
    v[4] = acc[m][1][0] * s1.x + b1.x; v[5] = acc[m][1][1] * s1.y + b1.y; v[6] = acc[m][1][2] * s1.z + b1.z; v[7] = acc[m][1][3] * s1.w + b1.w;
    if (a.relu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = PL ? avt::relu_keep_nan(v[i]) : fmaxf(v[i], 0.f);
    }
  };
  auto packed = [&](int m) {
    float v[8];
    values(m, v);
    uint4 pk;
    pk.x = avt::pack_bf16x2(v[0], v[1]);
    pk.y = avt::pack_bf16x2(v[2], v[3]);
    pk.z = avt::pack_bf16x2(v[4], v[5]);
    pk.w = avt::pack_bf16x2(v[6], v[7]);
    return pk;
  };
  if constexpr (!POOL && PL != 0) {
    if (a.fm_kt0) {  // uniform.  Frame-major tiles: this lane's rows 4q .. 4q + 3 of tile n are channels n_base + 16n + 4q ..
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int c = n_base + 16 * n + 4 * q;
        if (c >= a.Cout) continue;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + c);
        if (a.wscale) sv = *reinterpret_cast<const float4*>(a.wscale + c);
        const int cf = a.Cout / a.tgroup, j = c / cf, cin_f = c - j * cf;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          if (w * TPW + i >= R * MT) continue;
          float v[4] = {acc[i][n][0] * sv.x + bv.x, acc[i][n][1] * sv.y + bv.y, acc[i][n][2] * sv.z + bv.z, acc[i][n][3] * sv.w + bv.w};
          if (a.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = avt::relu_keep_nan(v[e]);
          }
          const int64_t pos = (int64_t)((b * a.To + to) * a.Ho + ho0 + unit_row(i)) * WO + unit_mt(i) * 16 + l15;
          if (a.out_f32) {
            const int64_t of = ((int64_t)(((b * a.To + to) * a.tgroup + j) * a.Ho + ho0 + unit_row(i)) * WO + unit_mt(i) * 16 + l15) * cf + cin_f;
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + of) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 oh, ol;
            avt::split2<PL == 2>(v[0], v[1], oh.x, ol.x);
            avt::split2<PL == 2>(v[2], v[3], oh.y, ol.y);
            *reinterpret_cast<uint2*>(a.out + pos * a.Cout + c) = oh;
            *reinterpret_cast<uint2*>(a.out_lo + pos * a.Cout + c) = ol;
          }
        }
      }
      return;
    }
  }
  if constexpr (!POOL) {
    if (c0 < a.Cout) {
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int64_t o = ((int64_t)((b * a.To + to) * a.Ho + ho0 + unit_row(i)) * WO + unit_mt(i) * 16 + l15) * a.Cout + c0;
        if (w * TPW + i < R * MT) {
          if constexpr (PL != 0) {
            float v[8];
            values(i, v);
            if (a.out_f32) {  // uniform
              const int cf = a.Cout / a.tgroup, j = c0 / cf, cin_f = c0 - j * cf;
              const int64_t of = ((int64_t)(((b * a.To + to) * a.tgroup + j) * a.Ho + ho0 + unit_row(i)) * WO + unit_mt(i) * 16 + l15) * cf + cin_f;
              float* d = reinterpret_cast<float*>(a.out) + of;
              *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
              *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
              continue;
            }
            uint4 oh, ol;
            avt::split2<PL == 2>(v[0], v[1], oh.x, ol.x);
            avt::split2<PL == 2>(v[2], v[3], oh.y, ol.y);
            avt::split2<PL == 2>(v[4], v[5], oh.z, ol.z);
            avt::split2<PL == 2>(v[6], v[7], oh.w, ol.w);
            *reinterpret_cast<uint4*>(a.out + o) = oh;
            *reinterpret_cast<uint4*>(a.out_lo + o) = ol;
          } else {
            *reinterpret_cast<uint4*>(a.out + o) = packed(i);
          }
        }
      }
    }
  } else if constexpr (PL == 0) {
    // conv tile [R rows][WO][32 channels] bf16 in LDS (over the patch: every wave is past its last read of it)
    constexpr int TROW = WO * 64;  // bytes per conv row
#pragma unroll
    for (int i = 0; i < TPW; ++i)
      *reinterpret_cast<uint4*>(lds + unit_row(i) * TROW + (unit_mt(i) * 16 + l15) * 64 + q * 16) = packed(i);  // (a repeat rewrites the same bytes)
    __syncthreads();
    constexpr int WP = WO / 2;
    const int Hp = a.Ho / 2;
    const int cf = a.Cout / a.tgroup;  // channels per output frame (time-grouped form: Cout = tgroup frames x cf)
    for (int i = tid; i < (OWN / 2) * WP * 4; i += NTHR) {
      const int cc = i & 3, pw_ = (i >> 2) % WP, pl = (i >> 2) / WP;  // 8-channel chunk, pooled column, pooled row of the group
      const int ch = n_base + cc * 8;
      if (ch >= a.Cout) continue;
      uint4 mx = make_uint4(0u, 0u, 0u, 0u);  // ReLU output >= 0 == bf16 +0
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        const int tr = 2 * pl + dh;  // tile row; conv row = ho0 + tr
        if (ho0 + tr < 0) continue;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          const int wc = 2 * pw_ - 1 + dw;
          if (wc < 0) continue;
          const uint4 v = *reinterpret_cast<const uint4*>(lds + tr * TROW + wc * 64 + cc * 16);
          mx.x = max2(mx.x, v.x);
          mx.y = max2(mx.y, v.y);
          mx.z = max2(mx.z, v.z);
          mx.w = max2(mx.w, v.w);
        }
      }
      const int j = ch / cf, cin_f = ch - j * cf;  // output frame within the time group, channel within the frame
      const int64_t pos = ((int64_t)((b * a.To + to) * a.tgroup + j) * Hp + (OWN / 2) * hg + pl) * WP + pw_;
      *reinterpret_cast<uint4*>(a.out + pos * a.ldo + cin_f) = mx;
    }
  } else {
    static_assert(PL == 0 || !POOL, "the pooled form exists in bf16 only (the split-plane one measured slower than stem + pool)");
  }
}

template <int MT, bool POOL, int PL = 0>
int launch(const StemArgs& a, int batch, hipStream_t st, const char* what) {
  constexpr int OWNP = PL ? RBPX : RBP;
  constexpr int R = POOL ? OWNP + 1 : RB;
  constexpr int patch = ((2 * R + 5) * (MT * 16 + 4) + BCH) * 16 * (PL ? 2 : 1), tile = POOL ? R * MT * 16 * (PL ? 128 : 64) : 0;
  constexpr int lds_bytes = patch > tile ? patch : tile;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_kernel<MT, POOL, PL>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("%s: hipFuncSetAttribute(%d B LDS): %s", what, lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  StemArgs b = a;
  b.ncg = (a.Cout + NT * 16 - 1) / (NT * 16);
  b.swz = 1;
  const dim3 grid((unsigned)(batch * a.To * (a.Ho / (POOL ? OWNP : RB)) * b.ncg));
  hipLaunchKernelGGL((stem_kernel<MT, POOL, PL>), grid, dim3(POOL ? 512 : 256), lds_bytes, st, b);
  return avt::check_launch(what);
}

int fill(StemArgs& a, const char* what, const void* in, const void* wt, const float* bias, void* out, int batch, int t,
         int h, int pw, int cout, int kt, int st, int pt, int relu, bool dense_input = true) {
  AVT_REQUIRE(in && wt && out, "%s: NULL pointer", what);
  AVT_REQUIRE(batch > 0 && t > 0 && kt > 0 && st > 0 && pt >= 0 && pt < kt, "%s: bad sizes", what);
  AVT_REQUIRE(avt_stem_conv_supported(h, pw, cout),
              "%s: unsupported shape h=%d pairs=%d cout=%d (rows/2 %% 4 == 0, 112 or 32 pairs; use "
              "avt_conv3d_igemm_bf16)", what, h, pw, cout);
  AVT_REQUIRE(avt::aligned16(in) && avt::aligned16(wt) && avt::aligned16(out) && (!bias || avt::aligned16(bias)),
              "%s: pointers must be 16-byte aligned", what);
  a.in = static_cast<const uint16_t*>(in);
  a.wt = static_cast<const uint16_t*>(wt);
  a.bias = bias;
  a.out = static_cast<uint16_t*>(out);
  a.T = t;
  a.H = h;
  a.PW = pw;
  a.To = (t + 2 * pt - kt) / st + 1;
  a.Ho = h / 2;
  a.KT = kt;
  a.st = st;
  a.pt = pt;
  a.Cout = cout;
  a.relu = relu;
  a.tgroup = 1;
  a.ldo = cout;
  a.in_lo = a.wt_lo = nullptr;
  a.out_lo = nullptr;
  a.wscale = nullptr;
  a.out_f32 = 0;
  a.fm_kt0 = 0;
  a.fidx = nullptr;
  a.mtab = a.mact = nullptr;
  a.ktm = 0;
  AVT_REQUIRE(a.To > 0, "%s: no output frames", what);
  const int64_t in_b = (int64_t)batch * t * h * pw * 16, wt_b = (int64_t)cout * kt * KF * 2;
  // (dense_input = false: `in` is a frame table whose own size the caller checks — the clips it stands for may exceed 4 GB)
  AVT_REQUIRE((!dense_input || in_b < (1ll << 32) - 64) && wt_b < (1ll << 31) && (int64_t)batch * a.To * a.Ho * pw < (1ll << 31),
              "%s: tensor too large for 32-bit offsets", what);
  a.in_bytes = dense_input ? (unsigned)in_b : 0u;
  a.wt_bytes = (unsigned)wt_b;
  return AVT_OK;
}

}  // namespace

extern "C" int avt_stem_conv_supported(int h, int pw, int cout) {
  return (h % 2 == 0 && (h / 2) % RB == 0 && (pw == 112 || pw == 32) && cout % 32 == 0) ? 1 : 0;
}

extern "C" int avt_stem_conv_bf16(const void* in, const void* wt, const float* bias, void* out, int batch, int t, int h,
                                  int pw, int cout, int kt, int st, int pt, int relu, void* stream) {
  StemArgs a;
  const int rc = fill(a, "avt_stem_conv_bf16", in, wt, bias, out, batch, t, h, pw, cout, kt, st, pt, relu);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pw == 112 ? launch<7, false>(a, batch, s, "avt_stem_conv_bf16") : launch<2, false>(a, batch, s, "avt_stem_conv_bf16");
}

extern "C" int avt_stem_conv_pool_bf16(const void* in, const void* wt, const float* bias, void* out, int batch, int t,
                                       int h, int pw, int cout, int kt, int st, int pt, int tgroup, int ldo,
                                       void* stream) {
  StemArgs a;
  const int rc = fill(a, "avt_stem_conv_pool_bf16", in, wt, bias, out, batch, t, h, pw, cout, kt, st, pt, 1);
  if (rc) return rc;
  AVT_REQUIRE(tgroup >= 1 && cout % tgroup == 0 && (cout / tgroup) % 8 == 0 && ldo % 8 == 0 && ldo >= cout / tgroup,
              "avt_stem_conv_pool_bf16: tgroup must split the channels into multiples of 8; ldo >= channels per frame");
  AVT_REQUIRE((h / 2) % RBP == 0, "avt_stem_conv_pool_bf16: conv rows (%d) must be a multiple of %d", h / 2, RBP);
  a.tgroup = tgroup;
  a.ldo = ldo;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pw == 112 ? launch<7, true>(a, batch, s, "avt_stem_conv_pool_bf16")
                   : launch<2, true>(a, batch, s, "avt_stem_conv_pool_bf16");
}

// frames_per_tile (the time-grouped form, st = frames per 32-channel group > 1): 0 = the classic weight image; 2 = the
// frame-major image (a 16-channel tile = 2 output frames of 8 channels) whose structurally-zero (frame tap, tile) pairs are skipped
static int fm_arg(StemArgs& a, const char* what, int frames_per_tile, int cout, int kt, int st) {
  AVT_REQUIRE(frames_per_tile == 0 || (frames_per_tile == 2 && st == 4 && cout == 32 && kt > st - 1),
              "%s: frames_per_tile is 0, or 2 for the 4-frame x 8-channel time-grouped form (st 4, cout 32)", what);
  a.fm_kt0 = frames_per_tile ? kt - st + 1 : 0;
  a.tgroup = frames_per_tile ? st : a.tgroup;
  return AVT_OK;
}

extern "C" int avt_stem_conv_x3(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo, const float* bias,
                                const float* wscale, void* out_hi, void* out_lo, int batch, int t, int h, int pw, int cout, int kt,
                                int st, int pt, int relu, int plane_dtype, int frames_per_tile, const int32_t* frame_idx,
                                int n_table_frames, void* stream) {
  StemArgs a;
  int rc = fill(a, "avt_stem_conv_x3", in_hi, wt_hi, bias, out_hi, batch, t, h, pw, cout, kt, st, pt, relu, frame_idx == nullptr);
  if (rc) return rc;
  if (frame_idx) {  // the input is a table of distinct frames (entries of frame_idx must lie in [0, n_table_frames): the
    // buffer descriptor's range check turns an index beyond it into zeros, never into a fault)
    AVT_REQUIRE(n_table_frames > 0 && (int64_t)n_table_frames * h * pw * 16 < (1ll << 32) - 64,
                "avt_stem_conv_x3: frame table of %d frames is empty or too large for 32-bit offsets", n_table_frames);
    a.fidx = frame_idx;
    a.in_bytes = (unsigned)((int64_t)n_table_frames * h * pw * 16);
  }
  rc = fm_arg(a, "avt_stem_conv_x3", frames_per_tile, cout, kt, st);
  if (rc) return rc;
  AVT_REQUIRE(in_lo && wt_lo && out_lo && avt::aligned16(in_lo) && avt::aligned16(wt_lo) && avt::aligned16(out_lo) &&
                  (!wscale || avt::aligned16(wscale)),
              "avt_stem_conv_x3: every tensor needs both planes, 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_stem_conv_x3: bad plane_dtype");
  a.in_lo = static_cast<const uint16_t*>(in_lo);
  a.wt_lo = static_cast<const uint16_t*>(wt_lo);
  a.out_lo = static_cast<uint16_t*>(out_lo);
  a.wscale = wscale;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (plane_dtype == AVT_X3_F16)
    return pw == 112 ? launch<7, false, 2>(a, batch, s, "avt_stem_conv_x3") : launch<2, false, 2>(a, batch, s, "avt_stem_conv_x3");
  return pw == 112 ? launch<7, false, 1>(a, batch, s, "avt_stem_conv_x3") : launch<2, false, 1>(a, batch, s, "avt_stem_conv_x3");
}

// the time-grouped fast stem over a frame table with the taps of one source frame merged (see StemArgs::mtab, include/avt.h)
extern "C" int avt_stem_conv_x3_merged(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo, const float* bias,
                                       const float* wscale, void* out_hi, void* out_lo, int batch, int t, int h, int pw, int cout,
                                       int kt, int st, int pt, int relu, int plane_dtype, const int32_t* tap_frames,
                                       const int32_t* tap_tiles, int ktm, int n_table_frames, void* stream) {
  StemArgs a;
  int rc = fill(a, "avt_stem_conv_x3_merged", in_hi, wt_hi, bias, out_hi, batch, t, h, pw, cout, kt, st, pt, relu, false);
  if (rc) return rc;
  rc = fm_arg(a, "avt_stem_conv_x3_merged", 2, cout, kt, st);  // the frame-major 4-frame x 8-channel form only
  if (rc) return rc;
  AVT_REQUIRE(tap_frames && tap_tiles && ktm > 0 && ktm <= kt, "avt_stem_conv_x3_merged: tap tables / 0 < ktm <= kt");
  AVT_REQUIRE(n_table_frames > 0 && (int64_t)n_table_frames * h * pw * 16 < (1ll << 32) - 64,
              "avt_stem_conv_x3_merged: frame table of %d frames is empty or too large for 32-bit offsets", n_table_frames);
  AVT_REQUIRE(in_lo && wt_lo && out_lo && avt::aligned16(in_lo) && avt::aligned16(wt_lo) && avt::aligned16(out_lo) &&
                  (!wscale || avt::aligned16(wscale)),
              "avt_stem_conv_x3_merged: every tensor needs both planes, 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_stem_conv_x3_merged: bad plane_dtype");
  a.in_bytes = (unsigned)((int64_t)n_table_frames * h * pw * 16);
  a.wt_bytes = (unsigned)((int64_t)cout * a.To * ktm * KF * 2);  // the merged image: [cout / 32][To * ktm] slabs
  a.mtab = tap_frames;
  a.mact = tap_tiles;
  a.ktm = ktm;
  a.in_lo = static_cast<const uint16_t*>(in_lo);
  a.wt_lo = static_cast<const uint16_t*>(wt_lo);
  a.out_lo = static_cast<uint16_t*>(out_lo);
  a.wscale = wscale;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (plane_dtype == AVT_X3_F16)
    return pw == 112 ? launch<7, false, 2>(a, batch, s, "avt_stem_conv_x3_merged") : launch<2, false, 2>(a, batch, s, "avt_stem_conv_x3_merged");
  return pw == 112 ? launch<7, false, 1>(a, batch, s, "avt_stem_conv_x3_merged") : launch<2, false, 1>(a, batch, s, "avt_stem_conv_x3_merged");
}

// the training forward (train_ops._StemX3): fp32 NDHWC out, no bias, no ReLU (BatchNorm follows in train mode); see include/avt.h
extern "C" int avt_stem_conv_x3_f32(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo, const float* wscale,
                                    float* out, int batch, int t, int h, int pw, int cout, int kt, int st, int pt, int tgroup,
                                    int plane_dtype, int frames_per_tile, void* stream) {
  StemArgs a;
  int rc = fill(a, "avt_stem_conv_x3_f32", in_hi, wt_hi, nullptr, out, batch, t, h, pw, cout, kt, st, pt, 0);
  if (rc) return rc;
  rc = fm_arg(a, "avt_stem_conv_x3_f32", frames_per_tile, cout, kt, st);
  if (rc) return rc;
  AVT_REQUIRE(!frames_per_tile || tgroup == st, "avt_stem_conv_x3_f32: the frame-major form has tgroup == st");
  AVT_REQUIRE(in_lo && wt_lo && avt::aligned16(in_lo) && avt::aligned16(wt_lo) && (!wscale || avt::aligned16(wscale)),
              "avt_stem_conv_x3_f32: every operand needs both planes, 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_stem_conv_x3_f32: bad plane_dtype");
  AVT_REQUIRE(tgroup >= 1 && cout % tgroup == 0 && (cout / tgroup) % 8 == 0,
              "avt_stem_conv_x3_f32: tgroup must split the channels into multiples of 8");
  a.in_lo = static_cast<const uint16_t*>(in_lo);
  a.wt_lo = static_cast<const uint16_t*>(wt_lo);
  a.out_lo = nullptr;
  a.wscale = wscale;
  a.tgroup = tgroup;
  a.out_f32 = 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (plane_dtype == AVT_X3_F16)
    return pw == 112 ? launch<7, false, 2>(a, batch, s, "avt_stem_conv_x3_f32") : launch<2, false, 2>(a, batch, s, "avt_stem_conv_x3_f32");
  return pw == 112 ? launch<7, false, 1>(a, batch, s, "avt_stem_conv_x3_f32") : launch<2, false, 1>(a, batch, s, "avt_stem_conv_x3_f32");
}

