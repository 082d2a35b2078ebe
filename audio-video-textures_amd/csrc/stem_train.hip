// stem_train — the TRAINING side of the SlowFast stems (Conv3d(3, C, [kt,7,7], stride [1,2,2], pad [kt//2,3,3]); the
// reference trains them through autograd -> MIOpen, contrastive_video_textures/train.py:114-141):
//   avt_clip_planes_f32   the clip [B,3,T,H,W] fp32 (any strides) -> the pixel-pair planes [B,T,H,W/2,8] hi / lo that the
//                         patch-resident stem kernels read (8 = 2 pixels x 4 channels, the 4th zero);
//   avt_stem_wgrad_x3     the stems' WEIGHT gradient in the pixel-pair form, patch-resident like the forward (stem_conv.hip).
// (The forward of the training step is avt_stem_conv_x3_f32 in stem_conv.hip: the inference kernel with fp32 output.)
//
// Why a kernel of its own: as a generic weight-gradient GEMM (wgrad_x3.hip, one [1,7,7] frame-tap slice per launch) the fast
// stem gathered every input pixel once per tap — 49 float4 gathers per output position and slice, 4.7 GB through L1/L2 per
// slice at 15 clips — and ran 9.5 ms of a 95 ms item (profiles/r03/train_layers_before_stem.log).  Here a workgroup stages
// the 13 input rows x 116 pixel pairs that 4 output rows touch ONCE per input frame (both bf16 planes, 55 KB) together with
// the dY rows of the kt output frames that frame feeds, and reads both MFMA operands from the LDS with the transposing read
// (ds_read_b64_tr_b16: positions are the REDUCTION axis of a weight gradient, channels are contiguous in memory):
//     dWp[co][dt][dh][dp][e] = sum over (b, to, ho, wo) of dY[b, to, ho, wo, co] * Xp[b, to + dt - pt, 2 ho + dh - 3, wo + dp - 2, e]
// with Xp the pair layout (e = pixel-in-pair * 4 + channel), dp = 0..3 the pair taps (8 column taps, of which the first is the
// structural zero of the pixel-pair form: its gradient is computed and never read).  v_mfma_f32_16x16x32_bf16: M = 16 =
// 2 pair taps x 8, K = 32 consecutive output columns of one row, N = 16 = 16 output channels (a slice of the slow stem's 64)
// or, for the fast stem's 8 channels, 2 FRAME TAPS x 8: one x fragment meets the dY rows of two output frames in the same
// instruction (5 frame taps = 3 pair steps, the sixth tap's rows are zeros).  Three products per term (xl*dyh + xh*dyl +
// xh*dyh, bf16 planes: gradients need fp32's exponent range) into fp32: 2^-16 per product, as wgrad_x3.
// Work split: one workgroup = 8 waves = (pair-tap half) x (32-column block); a wave keeps its accumulator tiles (3 x 7 for
// the fast stem) across ALL the units (clip, input frame, 4-row group) it is dealt, the partial sums of the workgroup meet
// in the LDS (ds_add_f32) and leave as coalesced fp32 atomics once per workgroup.  An x fragment (patch row j, column block)
// is read once per step and used by every (output row r, row tap dh) with 2 r + dh = j.
// Roofline: MFMA (3 x the bf16 work; the padded columns 112 -> 128 and the sixth frame tap are issued work, not algorithmic).
#include "avt_common.h"
#include "split_planes.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ---- clip -> pixel-pair planes -------------------------------------------------------------------------------------------
struct CpArgs {
  const float* in;
  uint16_t* hi;
  uint16_t* lo;
  int64_t sb, sc, st, sh, sw;  // element strides of the [B, 3, T, H, W] view
  int T, H, W;
  int64_t npix;
};

template <bool F16>
__global__ __launch_bounds__(256) void clip_planes_kernel(CpArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.npix) return;
  const int w = (int)(p % a.W);
  int64_t r = p / a.W;
  const int h = (int)(r % a.H);
  r /= a.H;
  const int t = (int)(r % a.T);
  const int64_t b = r / a.T;
  const float* s = a.in + b * a.sb + t * a.st + h * a.sh + w * a.sw;
  const float c0 = s[0], c1 = s[a.sc], c2 = s[2 * a.sc];
  uint2 vh, vl;
  avt::split2<F16>(c0, c1, vh.x, vl.x);
  avt::split2<F16>(c2, 0.0f, vh.y, vl.y);
  *reinterpret_cast<uint2*>(a.hi + p * 4) = vh;
  *reinterpret_cast<uint2*>(a.lo + p * 4) = vl;
}

// ---- the stems' weight gradient -------------------------------------------------------------------------------------------
constexpr int RB = 4;              // output rows per unit
constexpr int PROWS = 2 * RB + 5;  // input rows they touch
constexpr int KBLK = 4;            // 32-column blocks per output row (<= 128 columns)
constexpr int PWP = 32 * KBLK + 4;  // patch row in pairs: -2 .. 129 (columns beyond the clip are zeros)
constexpr int NTHR = 512;

struct SwArgs {
  const uint16_t* x_hi;  // [B, T, H, PW, 8] bf16 planes
  const uint16_t* x_lo;
  const float* dy;  // [B, To, Ho, Wo, Cout] fp32
  float* dw;        // [Cout][KT][7][4][8] fp32, zeroed by the caller: atomics
  int B, T, H, PW, To, Ho, Wo, Cout, pt;
  int hgroups, nunit;
  unsigned x_bytes;
};

__device__ __forceinline__ f32x4 mfma(i32x4 a, i32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ i32x4 tr_frag(const char* lds, int a0, int a1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a0)));
  const uint2 v = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a1)));
  return i32x4{(int)u.x, (int)u.y, (int)v.x, (int)v.y};
}

// KT frame taps; CO = output channels a workgroup owns per position row in the LDS (8: the fast stem; 16: a 16-channel slice
// of the slow stem's 64, blockIdx.y walks the slices)
template <int KT, int CO>
__global__ __launch_bounds__(NTHR, 2) void stem_wgrad_kernel(SwArgs a) {
  constexpr int FP = CO == 8 ? 2 : 1;           // frame taps per MFMA (N = FP x CO = 16)
  constexpr int NP = (KT + FP - 1) / FP;        // steps over the frame taps
  constexpr int PPL = PROWS * PWP * 16;         // one patch plane
  constexpr int YROW = 32 * KBLK * CO * 2;      // one dY row of one plane: [128 positions][CO] bf16
  constexpr int YPL = NP * FP * RB * YROW;      // one dY plane: [NP * FP frame taps][RB] rows (taps >= KT stay zeros)
  constexpr int LDS_BYTES = 2 * PPL + 2 * YPL;
  constexpr int RED = 16 * NP * 7 * 32 * 4;     // the flush buffer [16 n][NP][7][32] fp32 (over the staging area)
  static_assert(RED <= LDS_BYTES, "the flush buffer fits the staging area");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const lp = lds;             // patch hi | patch lo
  char* const ly = lds + 2 * PPL;   // dY hi | dY lo

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int at = w & 1, kb = w >> 1;  // pair taps 2 at, 2 at + 1; output columns 32 kb .. 32 kb + 31
  const int co0 = blockIdx.y * 16;
  const int nkb = (a.Wo + 31) / 32;

  // everything starts as zeros: the patch columns beyond the clip and the dY positions beyond Wo are never written
  for (int i = tid; i < LDS_BYTES / 16; i += NTHR) *reinterpret_cast<uint4*>(lds + i * 16) = make_uint4(0u, 0u, 0u, 0u);

  f32x4 acc[NP][7];
#pragma unroll
  for (int d = 0; d < NP; ++d)
#pragma unroll
    for (int h = 0; h < 7; ++h) acc[d][h] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read addresses (ds_read_b64_tr_b16: lane 4q + p of a 16-lane group supplies 8 bytes = channels 4p .. 4p + 3 of
  // position row q; lane i of the group receives channel i of the four positions).  Group g = lane >> 4 owns positions
  // 8g .. 8g + 7 of the 32-position block: two reads (positions + 0 .. 3, + 4 .. 7) make the 8 reduction values of a lane.
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int pos0 = 32 * kb + 8 * g + tq;                  // first-half position this lane addresses (second: + 4)
  const int xa = (pos0 + 2 * at) * 16 + 8 * tp;           // inside a patch row: pairs pos + 2 at, pos + 2 at + 1 = 32 bytes
  // inside a dY row: channel chunk tp of the 16 columns = (frame tap tp >> 1 of the pair, channels 4 (tp & 1) ..) for CO = 8
  const int ya = CO == 8 ? (tp >> 1) * (RB * YROW) + pos0 * 16 + 8 * (tp & 1) : pos0 * 32 + 8 * tp;

  const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)a.x_hi, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)a.x_lo, 0, a.x_bytes, 0x00020000);
  const int pcols = a.PW + 4;                   // chunks of a row that can hold pixels (pairs -2 .. PW + 1)

  for (int unit = blockIdx.x; unit < a.nunit; unit += gridDim.x) {
    int u = unit;
    const int hg = u % a.hgroups;
    u /= a.hgroups;
    const int ti = u % a.T, b = u / a.T;
    const int ho0 = hg * RB;
    __syncthreads();  // the previous unit's fragments have been read (first unit: the zero fill is complete)
    // patch: input rows 2 ho0 - 3 .. + 12, pairs -2 .. PW + 1 (out-of-range chunks arrive as zeros: buffer loads)
    const unsigned fbase = (unsigned)(((b * a.T + ti) * a.H) * a.PW) * 16u;
    for (int c = tid; c < PROWS * pcols; c += NTHR) {
      const int j = c / pcols, col = c - j * pcols;
      const int hi = 2 * ho0 - 3 + j, wi = col - 2;
      const bool ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.PW;
      const int off = ok ? (int)(fbase + (unsigned)((hi * a.PW + wi) * 16)) : (int)0xFFFFFFF0u;
      const i32x4 vh = __builtin_amdgcn_raw_buffer_load_b128(rxh, off, 0, 0);
      const i32x4 vl = __builtin_amdgcn_raw_buffer_load_b128(rxl, off, 0, 0);
      *reinterpret_cast<i32x4*>(lp + (j * PWP + col) * 16) = vh;
      *reinterpret_cast<i32x4*>(lp + PPL + (j * PWP + col) * 16) = vl;
    }
    // dY: for frame tap dt the output frame to = ti + pt - dt; rows ho0 .. ho0 + 3; fp32 -> bf16 planes [position][CO]
    constexpr int CQ = CO / 4;  // float4 per position
    for (int c = tid; c < KT * RB * a.Wo * CQ; c += NTHR) {
      const int cq = c % CQ;
      int r = c / CQ;
      const int wo = r % a.Wo;
      r /= a.Wo;
      const int row = r % RB, dt = r / RB;
      const int to = ti + a.pt - dt;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)to < (unsigned)a.To && co0 + 4 * cq < a.Cout)
        v = *reinterpret_cast<const float4*>(a.dy + ((int64_t)((b * a.To + to) * a.Ho + ho0 + row) * a.Wo + wo) * a.Cout + co0 + 4 * cq);
      uint2 vh, vl;
      avt::split2<false>(v.x, v.y, vh.x, vl.x);
      avt::split2<false>(v.z, v.w, vh.y, vl.y);
      const int o = (dt * RB + row) * YROW + wo * (CO * 2) + 8 * cq;
      *reinterpret_cast<uint2*>(ly + o) = vh;
      *reinterpret_cast<uint2*>(ly + YPL + o) = vl;
    }
    __syncthreads();
    if (kb < nkb) {
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const int to_a = ti + a.pt - pp * FP, to_b = to_a - (FP - 1);
        if ((unsigned)to_a >= (unsigned)a.To && (unsigned)to_b >= (unsigned)a.To) continue;  // uniform: both frames outside
        i32x4 yh[RB], yl[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int a0 = 2 * PPL + (pp * FP * RB + r) * YROW + ya, a1 = a0 + 4 * (CO * 2);
          yh[r] = tr_frag(lds, a0, a1);
          yl[r] = tr_frag(lds, a0 + YPL, a1 + YPL);
        }
#pragma unroll
        for (int j = 0; j < PROWS; ++j) {
          const int o = j * (PWP * 16) + xa;
          const i32x4 xh = tr_frag(lds, o, o + 64), xl = tr_frag(lds, PPL + o, PPL + o + 64);
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            const int dh = j - 2 * r;
            if (dh < 0 || dh >= 7) continue;  // compile-time
            acc[pp][dh] = mfma(xl, yh[r], acc[pp][dh]);  // small terms first
            acc[pp][dh] = mfma(xh, yl[r], acc[pp][dh]);
            acc[pp][dh] = mfma(xh, yh[r], acc[pp][dh]);
          }
        }
      }
    }
  }

  // flush: D[m][n], lane (n = lane & 15, q = lane >> 4): m = 4q + i = (pair tap within the half) * 8 + e;
  // n = co (CO = 16) or (frame tap of the pair) * 8 + co (CO = 8)
  __syncthreads();
  float* const red = reinterpret_cast<float*>(lds);
  for (int i = tid; i < RED / 4; i += NTHR) red[i] = 0.0f;
  __syncthreads();
  if (kb < nkb) {
    const int n = lane & 15, q = lane >> 4;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
      for (int dh = 0; dh < 7; ++dh)
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(&red[((n * NP + pp) * 7 + dh) * 32 + at * 16 + 4 * q + i], acc[pp][dh][i]);
  }
  __syncthreads();
  constexpr int PER_CO = KT * 7 * 32;
  for (int i = tid; i < CO * PER_CO; i += NTHR) {  // dW[co][dt][dh][32]: consecutive threads walk consecutive floats
    const int co = i / PER_CO, rest = i - co * PER_CO;
    const int dt = rest / (7 * 32), tail = rest - dt * (7 * 32);
    const int n = CO == 8 ? (dt % FP) * 8 + co : co, pp = dt / FP;
    if (co0 + co < a.Cout) unsafeAtomicAdd(a.dw + (int64_t)(co0 + co) * PER_CO + rest, red[(n * NP + pp) * (7 * 32) + tail]);
  }
}

// ---- the stems' MaxPool3d((1,3,3),(1,2,2),(0,1,1)) in the training step ---------------------------------------------------------
// fp32 NDHWC rows.  Forward: the max of the window and WHICH tap held it (4 bits per element, first maximum in row-major tap
// order, as torch's kernel picks it; a NaN wins, as there).  Backward as a GATHER: an input position sums the dy of the (at most
// four) windows that cover it and whose recorded tap is this position — no atomics, fixed order.  Both passes HBM-bound: the
// forward reads x once (neighbouring windows' re-reads meet in L1 / L2), the backward writes dx once.
struct MpArgs {
  const float* x;
  float* y;
  uint8_t* tap;  // [bt, Ho, Wo, C / 2]: two 4-bit taps per byte
  const float* dy;
  float* dx;
  int bt, H, W, C, Ho, Wo;
  int64_t ld;  // floats between consecutive rows of the forward's y / the backward's dy (C, or a concatenation buffer's width)
};

__global__ __launch_bounds__(256) void maxpool_train_fwd_kernel(MpArgs a) {
  const unsigned cpr = (unsigned)a.C >> 2;  // float4 chunks per row
  const unsigned total = (unsigned)a.bt * a.Ho * a.Wo * cpr;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned p = i / cpr;
    const unsigned cc = i - p * cpr;
    const unsigned q = p / (unsigned)a.Wo;
    const int wo = (int)(p - q * a.Wo);
    const unsigned b = q / (unsigned)a.Ho;
    const int ho = (int)(q - b * a.Ho);
    const float* frame = a.x + (int64_t)b * a.H * a.W * a.C + cc * 4;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned t[4] = {0u, 0u, 0u, 0u};
    bool first = true;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int hi = 2 * ho - 1 + dh;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int wi = 2 * wo - 1 + dw;
        if ((unsigned)hi >= (unsigned)a.H || (unsigned)wi >= (unsigned)a.W) continue;  // padding: never the maximum
        const float4 v = *reinterpret_cast<const float4*>(frame + (int64_t)(hi * a.W + wi) * a.C);
        const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (first || x[e] > m[e] || x[e] != x[e]) {  // strictly greater: the FIRST maximum keeps the gradient; a NaN takes it
            m[e] = x[e];
            t[e] = (unsigned)(dh * 3 + dw);
          }
        first = false;
      }
    }
    *reinterpret_cast<float4*>(a.y + (int64_t)p * a.ld + cc * 4) = make_float4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<uint16_t*>(a.tap + ((int64_t)p * a.C + cc * 4) / 2) = (uint16_t)(t[0] | (t[1] << 4) | (t[2] << 8) | (t[3] << 12));
  }
}

__global__ __launch_bounds__(256) void maxpool_train_bwd_kernel(MpArgs a) {
  const unsigned cpr = (unsigned)a.C >> 2;
  const unsigned total = (unsigned)a.bt * a.H * a.W * cpr;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned p = i / cpr;
    const unsigned cc = i - p * cpr;
    const unsigned q = p / (unsigned)a.W;
    const int w = (int)(p - q * a.W);
    const unsigned b = q / (unsigned)a.H;
    const int h = (int)(q - b * a.H);
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    // windows (ho, wo) with 2 ho - 1 + dh == h: ho = (h + 1 - dh) / 2 for the dh of h + 1's parity
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int hh = h + 1 - dh;
      if (hh < 0 || (hh & 1) || (hh >> 1) >= a.Ho) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int ww = w + 1 - dw;
        if (ww < 0 || (ww & 1) || (ww >> 1) >= a.Wo) continue;
        const int64_t row = (int64_t)(b * a.Ho + (hh >> 1)) * a.Wo + (ww >> 1);
        const unsigned t = *reinterpret_cast<const uint16_t*>(a.tap + (row * a.C + cc * 4) / 2);
        const float4 d = *reinterpret_cast<const float4*>(a.dy + row * a.ld + cc * 4);
        const unsigned me = (unsigned)(dh * 3 + dw);
        g[0] += ((t & 15u) == me) ? d.x : 0.f;
        g[1] += (((t >> 4) & 15u) == me) ? d.y : 0.f;
        g[2] += (((t >> 8) & 15u) == me) ? d.z : 0.f;
        g[3] += ((t >> 12) == me) ? d.w : 0.f;
      }
    }
    *reinterpret_cast<float4*>(a.dx + (int64_t)p * a.C + cc * 4) = make_float4(g[0], g[1], g[2], g[3]);
  }
}

// ---- weight planes of the training convolutions ------------------------------------------------------------------------------
// The optimizer changes every weight every step, so every step re-splits them into the planes conv_x3 reads.  As torch ops that
// was ~35 tiny launches per convolution (abs / amax / log2 / floor / pow / mul / casts / flip / permute-copies): 7 600 launches
// and 42 ms of a 415 ms step once a rank's items had become one pass (profiles/r03/rocprof_train_kernel_stats_before_planes.csv).
// Two kernels instead: the forward's planes straight from the channels-last weight (its memory IS [cout][taps][cin]), and the
// input gradient's transposed, tap-selected planes W'[ci][tap a][co] = W[co][taps[a]][ci] through a 32 x 32 LDS transpose.
template <bool F16>
__device__ __forceinline__ void weight_planes_row(const float* __restrict__ w, int K, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                  float* __restrict__ wscale, int r, float* red) {
  const float* row = w + (int64_t)r * K;
  float sc = 1.0f;
  if (wscale) {  // fp16 planes: the row scaled by a power of two into [2^9, 2^10) (undone on the accumulator: wscale = 1 / scale)
    float mx = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(row[k]));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
      __syncthreads();
    }
    mx = fmaxf(red[0], 1e-30f);
    int e;
    (void)frexpf(mx, &e);  // mx = m * 2^e, m in [0.5, 1): floor(log2(mx)) = e - 1
    sc = ldexpf(1.0f, 10 - e);
    if (threadIdx.x == 0) wscale[r] = ldexpf(1.0f, e - 10);
  }
  for (int k = 2 * threadIdx.x; k < K; k += 512) {
    uint32_t h, l;
    avt::split2<F16>(row[k] * sc, row[k + 1] * sc, h, l);
    *reinterpret_cast<uint32_t*>(hi + (int64_t)r * K + k) = h;
    *reinterpret_cast<uint32_t*>(lo + (int64_t)r * K + k) = l;
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void weight_planes_kernel(const float* __restrict__ w, int K, uint16_t* __restrict__ hi,
                                                            uint16_t* __restrict__ lo, float* __restrict__ wscale) {
  __shared__ float red[256];
  weight_planes_row<F16>(w, K, hi, lo, wscale, (int)blockIdx.x, red);
}

struct WtArgs {
  const float* w;  // [cout][taps][cin]
  uint16_t* hi;    // [cin][nsel][cout]
  uint16_t* lo;
  int cout, taps, cin, nsel;
  int sel[32];     // source tap of output tap a
};

// (tile: 32 x 33 floats)
__device__ __forceinline__ void weight_planes_t_block(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                      int cout, int taps, int cin, int nsel, int src, int bx, int by, int tap,
                                                      float (*tile)[33]) {
  const int co0 = by * 32, ci0 = bx * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int j = ty; j < 32; j += 8) {
    const int co = co0 + j, ci = ci0 + tx;
    tile[j][tx] = (co < cout && ci < cin) ? w[((int64_t)co * taps + src) * cin + ci] : 0.0f;
  }
  __syncthreads();
  // out row ci0 + i, columns co0 .. co0 + 31 as 16 pairs: thread (pair p = tx & 15, row i = ty + 8 * (tx >> 4) + 16 * h)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int i = ty + 8 * (tx >> 4) + 16 * h, p = tx & 15;
    const int ci = ci0 + i, co = co0 + 2 * p;
    if (ci < cin && co < cout) {
      uint32_t hh, ll;
      avt::split2<false>(tile[2 * p][i], tile[2 * p + 1][i], hh, ll);
      const int64_t o = ((int64_t)ci * nsel + tap) * cout + co;
      *reinterpret_cast<uint32_t*>(hi + o) = hh;
      *reinterpret_cast<uint32_t*>(lo + o) = ll;
    }
  }
}

__global__ __launch_bounds__(256) void weight_planes_t_kernel(WtArgs a) {
  __shared__ float tile[32][33];
  weight_planes_t_block(a.w, a.hi, a.lo, a.cout, a.taps, a.cin, a.nsel, a.sel[blockIdx.z], (int)blockIdx.x, (int)blockIdx.y,
                        (int)blockIdx.z, tile);
}

template <int KT, int CO>
int launch_wgrad(SwArgs& a, hipStream_t st) {
  constexpr int FP = CO == 8 ? 2 : 1, NP = (KT + FP - 1) / FP;
  constexpr int lds_bytes = 2 * PROWS * PWP * 16 + 2 * NP * FP * RB * 32 * KBLK * CO * 2;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad_kernel<KT, CO>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("avt_stem_wgrad_x3: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  const int slices = (a.Cout + 15) / 16;
  constexpr int wgs = 256;  // persistent workgroups (one per CU: up to 150 KB of LDS)
  int gx = wgs / slices;
  if (gx < 1) gx = 1;
  if (gx > a.nunit) gx = a.nunit;
  hipLaunchKernelGGL((stem_wgrad_kernel<KT, CO>), dim3((unsigned)gx, (unsigned)slices), dim3(NTHR), lds_bytes, st, a);
  return avt::check_launch("avt_stem_wgrad_x3");
}

}  // namespace

extern "C" int avt_clip_planes_f32(const float* in, int batch, int t, int h, int w, int64_t sb, int64_t sc, int64_t st, int64_t sh,
                                   int64_t sw, void* out_hi, void* out_lo, int plane_dtype, void* stream) {
  AVT_REQUIRE(in && out_hi && out_lo, "avt_clip_planes_f32: NULL pointer");
  AVT_REQUIRE(batch > 0 && t > 0 && h > 0 && w > 0, "avt_clip_planes_f32: bad sizes");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_clip_planes_f32: plane_dtype must be 0 (bf16) or 1 (fp16)");
  AVT_REQUIRE(avt::aligned16(out_hi) && avt::aligned16(out_lo), "avt_clip_planes_f32: planes must be 16-byte aligned");
  CpArgs a;
  a.in = in;
  a.hi = static_cast<uint16_t*>(out_hi);
  a.lo = static_cast<uint16_t*>(out_lo);
  a.sb = sb; a.sc = sc; a.st = st; a.sh = sh; a.sw = sw;
  a.T = t; a.H = h; a.W = w;
  a.npix = (int64_t)batch * t * h * w;
  AVT_REQUIRE(a.npix < (1ll << 31) * 256, "avt_clip_planes_f32: too many pixels");
  const dim3 grid((unsigned)((a.npix + 255) / 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (plane_dtype == AVT_X3_F16) hipLaunchKernelGGL(clip_planes_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(clip_planes_kernel<false>, grid, dim3(256), 0, s, a);
  return avt::check_launch("avt_clip_planes_f32");
}

extern "C" int avt_stem_wgrad_x3_supported(int h, int pw, int cout, int kt) {
  return (h % 2 == 0 && (h / 2) % RB == 0 && pw >= 8 && pw <= 32 * KBLK && (kt == 1 || kt == 5) && cout % 8 == 0 &&
          (cout == 8 || cout % 16 == 0)) ? 1 : 0;
}

extern "C" int avt_stem_wgrad_x3(const void* x_hi, const void* x_lo, const float* dy, float* dw, int batch, int t, int h, int pw,
                                 int cout, int kt, int pt, void* stream) {
  AVT_REQUIRE(x_hi && x_lo && dy && dw, "avt_stem_wgrad_x3: NULL pointer");
  AVT_REQUIRE(batch > 0 && t > 0 && pt >= 0 && pt < kt, "avt_stem_wgrad_x3: bad sizes");
  AVT_REQUIRE(avt_stem_wgrad_x3_supported(h, pw, cout, kt),
              "avt_stem_wgrad_x3: unsupported shape h=%d pairs=%d cout=%d kt=%d (rows/2 %% 4 == 0, 8..128 pairs, kt 1 or 5, "
              "cout 8 or a multiple of 16; use avt_conv3d_wgrad_x3_sub_f32)", h, pw, cout, kt);
  AVT_REQUIRE(avt::aligned16(x_hi) && avt::aligned16(x_lo) && avt::aligned16(dy) && avt::aligned16(dw),
              "avt_stem_wgrad_x3: pointers must be 16-byte aligned");
  SwArgs a;
  a.x_hi = static_cast<const uint16_t*>(x_hi);
  a.x_lo = static_cast<const uint16_t*>(x_lo);
  a.dy = dy;
  a.dw = dw;
  a.B = batch; a.T = t; a.H = h; a.PW = pw;
  a.To = t + 2 * pt - kt + 1;
  a.Ho = h / 2; a.Wo = pw;
  a.Cout = cout; a.pt = pt;
  AVT_REQUIRE(a.To > 0, "avt_stem_wgrad_x3: no output frames");
  const int64_t xb = (int64_t)batch * t * h * pw * 16;
  AVT_REQUIRE(xb < (1ll << 32) - 64 && (int64_t)batch * a.To * a.Ho * a.Wo * cout < (1ll << 40), "avt_stem_wgrad_x3: tensor too large");
  a.x_bytes = (unsigned)xb;
  a.hgroups = a.Ho / RB;
  a.nunit = batch * t * a.hgroups;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (kt == 5) return cout == 8 ? launch_wgrad<5, 8>(a, s) : launch_wgrad<5, 16>(a, s);
  return cout == 8 ? launch_wgrad<1, 8>(a, s) : launch_wgrad<1, 16>(a, s);
}

// MaxPool3d((1,3,3),(1,2,2),(0,1,1)) of the training step on fp32 NDHWC rows (see include/avt.h)
extern "C" int avt_maxpool_train_fwd(const float* x, float* y, void* tap, int bt, int h, int w, int c, int64_t ldy, void* stream) {
  AVT_REQUIRE(x && y && tap && bt > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "avt_maxpool_train_fwd: NULL pointer / bad sizes (c %% 4 == 0)");
  AVT_REQUIRE(avt::aligned16(x) && avt::aligned16(y) && avt::aligned16(tap), "avt_maxpool_train_fwd: pointers must be 16-byte aligned");
  MpArgs a = {};
  a.x = x; a.y = y; a.tap = static_cast<uint8_t*>(tap);
  a.bt = bt; a.H = h; a.W = w; a.C = c;
  a.Ho = (h - 1) / 2 + 1; a.Wo = (w - 1) / 2 + 1;
  AVT_REQUIRE(ldy == 0 || (ldy >= c && ldy % 4 == 0), "avt_maxpool_train_fwd: ldy = %lld must be 0 (contiguous) or a multiple of 4 >= c", (long long)ldy);
  a.ld = ldy ? ldy : c;
  const int64_t total = (int64_t)bt * a.Ho * a.Wo * (c / 4);
  AVT_REQUIRE((int64_t)bt * h * w * (c / 4) < (1ll << 32), "avt_maxpool_train_fwd: more than 2^32 chunks");
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(maxpool_train_fwd_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  return avt::check_launch("avt_maxpool_train_fwd");
}

extern "C" int avt_maxpool_train_bwd(const float* dy, const void* tap, float* dx, int bt, int h, int w, int c, int64_t ld_dy, void* stream) {
  AVT_REQUIRE(dy && dx && tap && bt > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "avt_maxpool_train_bwd: NULL pointer / bad sizes (c %% 4 == 0)");
  AVT_REQUIRE(avt::aligned16(dy) && avt::aligned16(dx) && avt::aligned16(tap), "avt_maxpool_train_bwd: pointers must be 16-byte aligned");
  MpArgs a = {};
  a.dy = dy; a.dx = dx; a.tap = const_cast<uint8_t*>(static_cast<const uint8_t*>(tap));
  a.bt = bt; a.H = h; a.W = w; a.C = c;
  a.Ho = (h - 1) / 2 + 1; a.Wo = (w - 1) / 2 + 1;
  AVT_REQUIRE(ld_dy == 0 || (ld_dy >= c && ld_dy % 4 == 0), "avt_maxpool_train_bwd: ld_dy = %lld must be 0 (contiguous) or a multiple of 4 >= c", (long long)ld_dy);
  a.ld = ld_dy ? ld_dy : c;
  const int64_t total = (int64_t)bt * h * w * (c / 4);
  AVT_REQUIRE(total < (1ll << 32), "avt_maxpool_train_bwd: more than 2^32 chunks");
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(maxpool_train_bwd_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  return avt::check_launch("avt_maxpool_train_bwd");
}

// Weight planes of a training convolution (see include/avt.h)
// ... with the rows GATHERED through an index map (round 6): row r, column k = w[map[r * K + k]], or 0 where the map holds -1 — the
// pixel-grouped (block-Toeplitz) forms of the few-channel layers' weights (train_ops._grouped_planes), which torch assembled with a flip,
// a cat, an index, a permute and a copy per weight and step (~700 launches of a config-5 step) in front of weight_planes_kernel
template <bool F16>
__device__ __forceinline__ void weight_planes_gather_row(const float* __restrict__ w, const int32_t* __restrict__ map, int K,
                                                         uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, float* __restrict__ wscale,
                                                         int r, float* red) {
  const int32_t* mrow = map + (int64_t)r * K;
  auto at = [&](int k) { const int32_t i = mrow[k]; return i >= 0 ? w[i] : 0.0f; };
  float sc = 1.0f;
  if (wscale) {
    float mx = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(at(k)));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
      __syncthreads();
    }
    mx = fmaxf(red[0], 1e-30f);
    int e;
    (void)frexpf(mx, &e);
    sc = ldexpf(1.0f, 10 - e);
    if (threadIdx.x == 0) wscale[r] = ldexpf(1.0f, e - 10);
  }
  for (int k = 2 * threadIdx.x; k < K; k += 512) {
    uint32_t h, l;
    avt::split2<F16>(at(k) * sc, at(k + 1) * sc, h, l);
    *reinterpret_cast<uint32_t*>(hi + (int64_t)r * K + k) = h;
    *reinterpret_cast<uint32_t*>(lo + (int64_t)r * K + k) = l;
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void weight_planes_gather_kernel(const float* __restrict__ w, const int32_t* __restrict__ map, int K,
                                                                   uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                                   float* __restrict__ wscale) {
  __shared__ float red[256];
  weight_planes_gather_row<F16>(w, map, K, hi, lo, wscale, (int)blockIdx.x, red);
}

// Every weight's planes of a training step in ONE launch (round 6): the optimizer has changed all of them, and re-made one by one
// they are ~450 launches of 5 us in front of the convolutions that read them — 2.3 ms of a 49 ms config-5 step at one item per rank
// (profiles/r06/one_item_graph_timeline.txt).  A job is one launch of the kernels above (include/avt.h AvtPlaneJob); block b of the
// grid is block b - blk0 of job blk2job[b]: the same device functions, the same bits.
struct PlaneJob {
  const float* w;
  uint16_t* hi;
  uint16_t* lo;
  float* wscale;        // fp16 row planes / gathered rows: 1 / scale per row; NULL: bf16 planes
  const int32_t* map;   // gathered rows
  int kind;             // 0 rows, 1 transposed + tap-selected, 2 gathered rows
  int f16;              // rows / gathered rows: fp16 planes (with wscale) or bf16 planes
  int rows, K;          // rows / gathered rows
  int cout, taps, cin, nsel;  // transposed: w [cout][taps][cin] -> [cin][nsel][cout]
  int gx, gy;           // transposed: tiles along cin, cout (blocks = gx * gy * nsel)
  int blk0;             // the job's first block
  int pad_;
  int sel[32];
};
static_assert(sizeof(PlaneJob) == 216, "AvtPlaneJob layout (include/avt.h)");

__global__ __launch_bounds__(256) void weight_planes_multi_kernel(const PlaneJob* __restrict__ jobs, const int32_t* __restrict__ blk2job) {
  __shared__ float sm[32 * 33];
  const PlaneJob& j = jobs[blk2job[blockIdx.x]];
  const int b = (int)blockIdx.x - j.blk0;
  if (j.kind == 0) {
    if (j.f16) weight_planes_row<true>(j.w, j.K, j.hi, j.lo, j.wscale, b, sm);
    else weight_planes_row<false>(j.w, j.K, j.hi, j.lo, nullptr, b, sm);
  } else if (j.kind == 1) {
    const int bx = b % j.gx, t1 = b / j.gx, by = t1 % j.gy, tap = t1 / j.gy;
    weight_planes_t_block(j.w, j.hi, j.lo, j.cout, j.taps, j.cin, j.nsel, j.sel[tap], bx, by, tap, reinterpret_cast<float(*)[33]>(sm));
  } else {
    if (j.f16) weight_planes_gather_row<true>(j.w, j.map, j.K, j.hi, j.lo, j.wscale, b, sm);
    else weight_planes_gather_row<false>(j.w, j.map, j.K, j.hi, j.lo, nullptr, b, sm);
  }
}

extern "C" int avt_weight_planes_job_bytes(void) { return (int)sizeof(PlaneJob); }

// jobs: DEVICE array of AvtPlaneJob, blk2job: DEVICE int32 [nblocks]; the host has validated the jobs (train_ops builds them from the
// arguments its one-by-one launches passed the checks of avt_weight_planes_*_f32 with)
extern "C" int avt_weight_planes_multi(const void* jobs, const int32_t* blk2job, int nblocks, void* stream) {
  AVT_REQUIRE(jobs && blk2job && nblocks > 0, "avt_weight_planes_multi: NULL pointer / no blocks");
  AVT_REQUIRE(reinterpret_cast<uintptr_t>(jobs) % 8 == 0, "avt_weight_planes_multi: the job table must be 8-byte aligned");
  hipLaunchKernelGGL(weight_planes_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const PlaneJob*>(jobs), blk2job);
  return avt::check_launch("avt_weight_planes_multi");
}

extern "C" int avt_weight_planes_gather_f32(const float* w, const int32_t* map, int rows, int k, void* hi, void* lo, float* wscale,
                                            int plane_dtype, void* stream) {
  AVT_REQUIRE(w && map && hi && lo && rows > 0 && k > 0 && k % 2 == 0, "avt_weight_planes_gather_f32: NULL pointer / bad sizes (k even)");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_weight_planes_gather_f32: bad plane_dtype");
  AVT_REQUIRE((plane_dtype == AVT_X3_F16) == (wscale != nullptr),
              "avt_weight_planes_gather_f32: fp16 planes are row-scaled (wscale), bf16 planes are not");
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint16_t *h = static_cast<uint16_t*>(hi), *l = static_cast<uint16_t*>(lo);
  if (plane_dtype == AVT_X3_F16)
    hipLaunchKernelGGL(weight_planes_gather_kernel<true>, dim3((unsigned)rows), dim3(256), 0, s, w, map, k, h, l, wscale);
  else
    hipLaunchKernelGGL(weight_planes_gather_kernel<false>, dim3((unsigned)rows), dim3(256), 0, s, w, map, k, h, l, wscale);
  return avt::check_launch("avt_weight_planes_gather_f32");
}

extern "C" int avt_weight_planes_f32(const float* w, int cout, int k, void* hi, void* lo, float* wscale, int plane_dtype, void* stream) {
  AVT_REQUIRE(w && hi && lo && cout > 0 && k > 0 && k % 2 == 0, "avt_weight_planes_f32: NULL pointer / bad sizes (k even)");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_weight_planes_f32: bad plane_dtype");
  AVT_REQUIRE((plane_dtype == AVT_X3_F16) == (wscale != nullptr), "avt_weight_planes_f32: fp16 planes are row-scaled (wscale), bf16 planes are not");
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto h = static_cast<uint16_t*>(hi), l = static_cast<uint16_t*>(lo);
  if (plane_dtype == AVT_X3_F16) hipLaunchKernelGGL(weight_planes_kernel<true>, dim3((unsigned)cout), dim3(256), 0, s, w, k, h, l, wscale);
  else hipLaunchKernelGGL(weight_planes_kernel<false>, dim3((unsigned)cout), dim3(256), 0, s, w, k, h, l, wscale);
  return avt::check_launch("avt_weight_planes_f32");
}

extern "C" int avt_weight_planes_t_f32(const float* w, int cout, int taps, int cin, const int32_t* sel, int nsel, void* hi, void* lo,
                                       void* stream) {
  AVT_REQUIRE(w && hi && lo && sel && cout > 0 && cout % 2 == 0 && cin > 0 && taps > 0 && nsel > 0 && nsel <= 32,
              "avt_weight_planes_t_f32: NULL pointer / bad sizes (cout even, 1..32 selected taps)");
  WtArgs a;
  a.w = w; a.hi = static_cast<uint16_t*>(hi); a.lo = static_cast<uint16_t*>(lo);
  a.cout = cout; a.taps = taps; a.cin = cin; a.nsel = nsel;
  for (int i = 0; i < nsel; ++i) {
    AVT_REQUIRE(sel[i] >= 0 && sel[i] < taps, "avt_weight_planes_t_f32: tap %d outside the filter", sel[i]);
    a.sel[i] = sel[i];
  }
  const dim3 grid((unsigned)((cin + 31) / 32), (unsigned)((cout + 31) / 32), (unsigned)nsel);
  hipLaunchKernelGGL(weight_planes_t_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return avt::check_launch("avt_weight_planes_t_f32");
}
