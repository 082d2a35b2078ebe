// bottleneck_fused — one whole residual bottleneck of the SlowFast FAST pathway in a single kernel:
//     out = relu( c( relu( b( relu( a(x) ) ) ) ) + x )
//     a: Conv3d [3,1,1] C -> Cm  (temporal),  b: Conv3d [1,3,3] Cm -> Cm  (spatial),  c: Conv3d [1,1,1] Cm -> C
// with the BatchNorms folded (the blocks of the third-party SlowFast model the reference runs per clip window,
// contrastive_video_textures/models/models.py:335, 399; identity-shortcut, stride-1 blocks of the beta = 1/8 pathway:
// C = 32 / 64 / 128, Cm = 8 / 16 / 32, 32 frames).
//
// Why: as three launches these layers are 12 % of the encoder's FLOPs but 26 % of its time — a and b run at 2-3 TB/s
// on few-channel tensors, and every intermediate crosses HBM.  Fused, the block's HBM traffic is x once (+ a 2-row
// halo per strip) and out once; the 8/16-channel intermediates never leave LDS and the residual is read from the
// staged x.  The kernel is HBM-bound by construction (MFMA work: ~1,500 of ~6,800 cycles per step).
//
// Work decomposition: a workgroup (8 waves) owns one (clip, strip of HT rows, chunk of TC frames) and walks the
// frames in order.  LDS holds a ring of 3 frames of the x strip (HT+2 rows, LDS-DMA with hardware zero fill for rows /
// frames outside the tensor = the convolutions' zero padding), the a-output strip with a zero border column on each
// side, and the b-output strip.  Per frame: [a] 16x16x32 MFMAs over the 3 ring frames -> relu -> bf16 -> LDS (rows
// outside the image forced to zero: b pads with zeros, not with a(0)); barrier; DMA of frame t+2 into the slot of
// t-1; [b] 9 taps as 5 tap-pair MFMAs read straight from the a strip at shifted addresses; barrier; [c] + bias +
// residual (from the ring) -> relu -> 16-byte stores.  Weights arrive pre-packed in MFMA fragment order (1 KB per
// fragment, lane-linear) and live in registers; the weights are the first MFMA operand, so a lane ends with 4
// consecutive channels of one position, and c's channel order is permuted in the packing so two tiles give 8
// consecutive channels (one 16-byte store), as in stem_conv.hip.
// Cm = 8 is padded to 16 on the host (zero filters / zero taps): the structured zeros cost MFMA issue only.  For
// Cm = 32 the a / b weight fragments (42 KB) live in LDS and b takes one tap per MFMA k-step.
#include <stdlib.h>

#include "avt_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFFF0u;

struct BArgs {
  const uint16_t* x;
  uint16_t* out;
  const i32x4* wa;  // [3][C/32][64 lanes]
  const i32x4* wb;  // [5][64]
  const i32x4* wc;  // [C/16][64]
  const i32x4* wsc;  // first-block form: shortcut conv fragments [C/16][64]
  const float* ba;  // [16]
  const float* bb;  // [16]
  const float* bc;  // [C]
  int T, H;
  int strips, tchunks, TC;
  int swz;  // XCD-contiguous work order (always on since round 2)
  unsigned x_bytes;
};

// CMP = bottleneck width as packed: 16 (Cm = 8 / 16: weights in registers, b's taps in pairs) or 32 (Cm = 32: the
// a / b weight fragments live in LDS, one tap per MFMA k-step)
// CIN = input channels: C (identity shortcut: out = relu(c(..) + x)) or 8 (first block of res2: x has 8 channels, the
// three frame taps of a are ONE MFMA k-step, and the shortcut is a 1x1x1 conv of x accumulated into c's MFMA tile)
// ST = 2: first block of res3 / res4 (CIN = C/2 >= 32): b has spatial stride 2 (the strip walks OUTPUT rows, a is
// computed on the 2*HT+1 input rows they touch) and the 1x1x1 shortcut samples x at the even positions.
// NW = waves per workgroup (2-4 per SIMD: the stages are latency- / issue-bound, not MFMA-bound)
template <int C, int W, int HT, int CMP, int CIN = C, int ST = 1, int NW = 8>
__global__ __launch_bounds__(NW * 64, 1) void bottleneck_kernel(BArgs a) {
  constexpr bool SC = CIN != C;   // shortcut conv instead of the identity
  constexpr bool FIRST = CIN == 8;  // ... with all three frame taps of a in one MFMA k-step
  constexpr int WO = W / ST;        // output row length
  constexpr int KS = FIRST ? 1 : CIN / 32;  // k-steps of the shortcut conv
  constexpr int RX = ST * HT + (ST == 1 ? 2 : 1);  // x / a-output rows of a strip (input rows its outputs touch)
  constexpr int PX = RX * W;            // positions of the x strip
  constexpr int REC = CIN * 2;          // bytes per x position
  constexpr int CH = CIN / 8;           // 16-byte chunks per x position
  constexpr int PPI = 1024 / REC;       // positions per DMA wave-instruction
  constexpr int NDMA = (PX + PPI - 1) / PPI;  // DMA wave-instructions per frame
  constexpr int XFRAME = NDMA * 1024;   // LDS bytes per ring frame (whole instructions)
  constexpr int AW = W + 2;             // a-output row length (zero border columns)
  constexpr int APOS = RX * AW;
  constexpr int MTA = (PX + 15) / 16;   // a-stage M-tiles
  constexpr int PB = HT * WO;           // b / c positions
  constexpr int MTB = (PB + 15) / 16;
  constexpr int AREC = CMP * 2;         // bytes per position of the a / b strips
  constexpr int NTA = CMP / 16;         // N-tiles of a and b
  constexpr int NB = CMP == 16 ? 5 : 9;  // k-steps of b: tap pairs (16 channels) or single taps (32 channels)
  constexpr bool WLDS = CMP > 16;
  constexpr int ABYTES = (APOS + 32) * AREC;  // a strip + room for the last partial tile's stores
  constexpr int BBYTES = MTB * 16 * AREC;
  constexpr int KA = FIRST ? 1 : CIN / 32;  // k-steps per frame tap of a (FIRST: one k-step holds all three taps)
  constexpr int NFA = (FIRST ? 1 : 3) * KA * NTA, NFB = NB * NTA;  // weight fragments of a and b
  constexpr int NTC = C / 16;           // N-tiles of c
  constexpr int CIT = (MTB + NW - 1) / NW;   // c-stage tiles per wave
  constexpr int NST = CIT * (NTC / 2);       // ... = output store instructions per wave per frame
  static_assert(CH == 1 || CH == 4 || CH == 8 || CH == 16, "x records of 16 / 64 / 128 / 256 bytes");
  static_assert(!FIRST || CMP == 16, "8-channel first-block form: width <= 16");
  static_assert(ST == 1 || (SC && !FIRST && W % 2 == 0), "strided form: first block with a shortcut conv");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* xr = lds;                       // [3][XFRAME]
  char* ao = lds + 3 * XFRAME;          // [APOS (+pad)][CMP channels]
  char* bo = ao + ABYTES;               // [MTB*16][CMP channels]
  char* wl = bo + BBYTES;               // WLDS: [NFA + NFB] fragments of 1 KB

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  int bid = a.swz ? avt::xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int tch = bid % a.tchunks;
  bid /= a.tchunks;
  const int strip = bid % a.strips, b = bid / a.strips;
  const int h0 = strip * HT, t0 = tch * a.TC;  // h0: first OUTPUT row of the strip
  const int hin0 = ST * h0 - 1;                // ... and the input row of x-strip row 0
  const int t1 = (t0 + a.TC < a.T) ? t0 + a.TC : a.T;

  // ---- weights: MFMA fragments in registers (or, for the wide form, a's and b's in LDS)
  bf16x8 wa_r[WLDS ? 1 : NFA], wb_r[WLDS ? 1 : NFB], wc[NTC], wsc[SC ? NTC * KS : 1];
  if constexpr (WLDS) {
    for (int f = wid; f < NFA + NFB; f += NW)
      *reinterpret_cast<i32x4*>(wl + f * 1024 + lane * 16) = f < NFA ? a.wa[f * 64 + lane] : a.wb[(f - NFA) * 64 + lane];
  } else {
#pragma unroll
    for (int f = 0; f < NFA; ++f) wa_r[f] = __builtin_bit_cast(bf16x8, a.wa[f * 64 + lane]);
#pragma unroll
    for (int f = 0; f < NFB; ++f) wb_r[f] = __builtin_bit_cast(bf16x8, a.wb[f * 64 + lane]);
  }
  auto WA = [&](int dt, int k, int n) {
    const int f = (dt * KA + k) * NTA + n;
    if constexpr (WLDS) return *reinterpret_cast<const bf16x8*>(wl + f * 1024 + lane * 16);
    else return wa_r[f];
  };
  auto WB = [&](int j, int n) {
    const int f = j * NTA + n;
    if constexpr (WLDS) return *reinterpret_cast<const bf16x8*>(wl + (NFA + f) * 1024 + lane * 16);
    else return wb_r[f];
  };
#pragma unroll
  for (int n = 0; n < NTC; ++n) wc[n] = __builtin_bit_cast(bf16x8, a.wc[n * 64 + lane]);
  if constexpr (SC) {
#pragma unroll
    for (int n = 0; n < NTC * KS; ++n) wsc[n] = __builtin_bit_cast(bf16x8, a.wsc[n * 64 + lane]);
  }
  float4 bav[NTA], bbv[NTA];
#pragma unroll
  for (int n = 0; n < NTA; ++n) {
    bav[n] = *reinterpret_cast<const float4*>(a.ba + 16 * n + 4 * q);
    bbv[n] = *reinterpret_cast<const float4*>(a.bb + 16 * n + 4 * q);
  }
  float4 bcv[NTC / 2][2];
#pragma unroll
  for (int np = 0; np < NTC / 2; ++np) {
    bcv[np][0] = *reinterpret_cast<const float4*>(a.bc + 32 * np + 8 * q);
    bcv[np][1] = *reinterpret_cast<const float4*>(a.bc + 32 * np + 8 * q + 4);
  }

  // ---- zero the a strip once: its border columns (and tile padding) stay zero for the whole walk
  for (int i = tid * 16; i < ABYTES; i += NW * 64 * 16) *reinterpret_cast<i32x4*>(ao + i) = i32x4{0, 0, 0, 0};

  // ---- x ring DMA: instruction d of a frame fills positions d*PPI + lane/CH, slot lane%CH (chunk = slot ^ swizzle)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  auto swz = [&](int p) { return (CH == 4 ? (p >> 2) : CH == 8 ? (p >> 1) : p) & (CH - 1); };  // CH == 1: 0
  constexpr int NDW = (NDMA + NW - 1) / NW;  // DMA instructions per wave per frame
  unsigned poff[NDW];                  // byte offset of this lane's chunk inside a frame, or OOB
#pragma unroll
  for (int u = 0; u < NDW; ++u) {
    const int d = wid + NW * u;
    const int p = d * PPI + lane / CH, slot = lane % CH;
    const int r = p / W, w = p - r * W;
    const int chunk = slot ^ swz(p);
    const int h = hin0 + r;
    const bool ok = d < NDMA && p < PX && (unsigned)h < (unsigned)a.H;
    poff[u] = ok ? (unsigned)(((h * W + w) * CIN + chunk * 8) * 2) : kOob;
  }
  auto dma_frame = [&](int tt) {  // frame tt of the clip -> ring slot tt mod 3 (zeros when tt is outside the clip)
    const int slot = ((tt % 3) + 3) % 3;
    const bool tin = (unsigned)tt < (unsigned)a.T;
    const unsigned fbase = (unsigned)((b * a.T + tt) * a.H) * (unsigned)(W * REC);
#pragma unroll
    for (int u = 0; u < NDW; ++u) {
      const int d = wid + NW * u;
      if (d < NDMA) {
        const unsigned off = (tin && poff[u] != kOob) ? fbase + poff[u] : kOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xr + slot * XFRAME + d * 1024),
                                                 16, (int)off, 0, 0, 0);
      }
    }
  };
  auto xoff = [&](int p, int chunk) {  // byte offset, inside a ring frame, of a 16-byte chunk of x position p
    return p * REC + ((chunk ^ swz(p)) * 16);
  };

  // ---- per-lane addresses of every tile this wave owns, computed ONCE: the stages are instruction-issue bound
  // (8 waves x ~900 instructions per frame before this table, ~350 after), not MFMA- or LDS-bound
  constexpr int AIT = (MTA + NW - 1) / NW;  // a-stage tiles per wave
  int a_rd[AIT][KA], a_st[AIT];
  unsigned a_in = 0;  // bit it: the tile's row is inside the image (else b's zero padding)
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int p = (wid + NW * it) * 16 + l15;
    const int pc = p < PX ? p : PX - 1;  // partial last tile: read a valid position, store behind the strip
#pragma unroll
    for (int k = 0; k < KA; ++k) a_rd[it][k] = FIRST ? pc * REC : xoff(pc, k * 4 + q);
    const int r = p / W, w = p - r * W;
    a_st[it] = (p < PX ? r * AW + w + 1 : APOS + (p - PX)) * AREC + q * 8;
    a_in |= ((unsigned)(hin0 + r) < (unsigned)a.H ? 1u : 0u) << it;
  }
  int tapoff[NB];  // b: byte offset of this lane's operand chunk for k-step j, relative to tap (0,0) of its position
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if constexpr (CMP == 16) {  // k-groups 0,1 = tap 2j, 2,3 = tap 2j+1 (tap 9 has zero weights: any finite data)
      const int tap = (q >> 1) ? (2 * j + 1 < 9 ? 2 * j + 1 : 8) : 2 * j;
      tapoff[j] = ((tap / 3) * AW + tap % 3) * AREC + (q & 1) * 16;
    } else {
      tapoff[j] = ((j / 3) * AW + j % 3) * AREC + q * 16;
    }
  }
  const int cchunk = (CMP == 16 ? (q & 1) : q) * 16;  // c: this lane's k-group inside a b-strip record
  int b_rd[CIT], c_res[CIT][SC ? KS : NTC / 2], c_out[CIT];
  unsigned c_ok = 0;
  const int Ho = a.H / ST;
#pragma unroll
  for (int it = 0; it < CIT; ++it) {
    const int mt = wid + NW * it;
    const int p = mt * 16 + l15;
    const int pc = p < PB ? p : PB - 1;
    const int r = pc / WO, w = pc - r * WO;
    b_rd[it] = (ST * r * AW + ST * w) * AREC;  // tap (0,0) of this output position in the a strip
    const int px = (ST * r + 1) * W + ST * w;   // x-strip position of the output's centre tap (residual / shortcut operand)
    if constexpr (SC) {
#pragma unroll
      for (int k = 0; k < KS; ++k) c_res[it][k] = FIRST ? px * REC : xoff(px, k * 4 + q);
    } else {
#pragma unroll
      for (int np = 0; np < NTC / 2; ++np) c_res[it][np] = xoff(px, (32 * np + 8 * q) / 8);
    }
    c_out[it] = ((h0 + r) * WO + w) * C + 8 * q;
    c_ok |= ((mt < MTB && p < PB && h0 + r < Ho) ? 1u : 0u) << it;
  }

  dma_frame(t0 - 1);
  dma_frame(t0);
  dma_frame(t0 + 1);
  for (int t = t0; t < t1; ++t) {
    // frame t+1's DMA was issued BEFORE the previous frame's output stores: wait for it, not for the stores
    if (t == t0)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    __syncthreads();  // frames t-1, t, t+1 are in the ring; every wave is done with the previous frame's c stage
    const char* xm_ = xr + ((((t - 1) % 3) + 3) % 3) * XFRAME;
    const char* x0_ = xr + (t % 3) * XFRAME;
    const char* xp_ = xr + ((t + 1) % 3) * XFRAME;
    const char* xq_ = q == 0 ? xm_ : (q == 2 ? xp_ : x0_);  // FIRST: this lane's frame tap
    (void)xq_;
    // ---- [a] temporal conv over the ring -> relu -> a strip
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      if (wid + NW * it < MTA) {  // wave-uniform
        f32x4 acc[NTA];
#pragma unroll
        for (int n = 0; n < NTA; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (FIRST) {  // k-groups 0,1,2 = frames t-1, t, t+1 (8 channels each); group 3 has zero weights
          const bf16x8 xq = *reinterpret_cast<const bf16x8*>(xq_ + a_rd[it][0]);
#pragma unroll
          for (int n = 0; n < NTA; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WA(0, 0, n), xq, acc[n], 0, 0, 0);
        } else {
#pragma unroll
          for (int k = 0; k < KA; ++k) {
            const bf16x8 xm = *reinterpret_cast<const bf16x8*>(xm_ + a_rd[it][k]);
            const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(x0_ + a_rd[it][k]);
            const bf16x8 xp = *reinterpret_cast<const bf16x8*>(xp_ + a_rd[it][k]);
#pragma unroll
            for (int n = 0; n < NTA; ++n) {
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WA(0, k, n), xm, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WA(1, k, n), x0, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WA(2, k, n), xp, acc[n], 0, 0, 0);
            }
          }
        }
        const bool rowin = (a_in >> it) & 1u;  // rows outside the image are b's ZERO padding, not a(0)
#pragma unroll
        for (int n = 0; n < NTA; ++n) {
          uint2 pk;
          pk.x = rowin ? avt::pack_bf16x2(fmaxf(acc[n][0] + bav[n].x, 0.f), fmaxf(acc[n][1] + bav[n].y, 0.f)) : 0u;
          pk.y = rowin ? avt::pack_bf16x2(fmaxf(acc[n][2] + bav[n].z, 0.f), fmaxf(acc[n][3] + bav[n].w, 0.f)) : 0u;
          *reinterpret_cast<uint2*>(ao + a_st[it] + n * 32) = pk;
        }
      }
    }
    __syncthreads();  // a strip complete; ring slot of frame t-1 is free
    if (t + 2 <= t1) dma_frame(t + 2);  // lands under the b and c stages (frame t1 itself is the last halo needed)
    // ---- [b] 3x3 spatial conv, operands straight from the a strip at tap-shifted addresses
#pragma unroll
    for (int it = 0; it < CIT; ++it) {
      if (wid + NW * it < MTB) {
        f32x4 acc[NTA];
#pragma unroll
        for (int n = 0; n < NTA; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(ao + b_rd[it] + tapoff[j]);
#pragma unroll
          for (int n = 0; n < NTA; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WB(j, n), af, acc[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < NTA; ++n) {
          uint2 pk;
          pk.x = avt::pack_bf16x2(fmaxf(acc[n][0] + bbv[n].x, 0.f), fmaxf(acc[n][1] + bbv[n].y, 0.f));
          pk.y = avt::pack_bf16x2(fmaxf(acc[n][2] + bbv[n].z, 0.f), fmaxf(acc[n][3] + bbv[n].w, 0.f));
          *reinterpret_cast<uint2*>(bo + ((wid + NW * it) * 16 + l15) * AREC + n * 32 + q * 8) = pk;
        }
      }
    }
    __syncthreads();  // b strip complete
    // ---- [c] pointwise conv + bias + residual (x of frame t, from the ring) -> relu -> global
    uint16_t* oframe = a.out + (int64_t)((b * a.T + t) * Ho) * (WO * C);
#pragma unroll
    for (int it = 0; it < CIT; ++it) {  // the same trip count in every wave: the stores are counted by s_waitcnt
      const int mt = wid + NW * it;
      const int pr = (mt < MTB ? mt : MTB - 1) * 16 + l15;
      const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bo + pr * AREC + cchunk);  // CMP = 16: k >= 16 has zero weights
      bf16x8 xs[SC ? KS : 1];  // shortcut conv operand: x(t) at the output's centre tap
      if constexpr (SC) {
#pragma unroll
        for (int k = 0; k < KS; ++k) xs[k] = *reinterpret_cast<const bf16x8*>(x0_ + c_res[it][k]);
      }
#pragma unroll
      for (int np = 0; np < NTC / 2; ++np) {
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[2 * np], bf, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[2 * np + 1], bf, c1, 0, 0, 0);
        const float4 b0 = bcv[np][0], b1 = bcv[np][1];
        uint4 rs = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (SC) {  // shortcut = 1x1x1 conv of x(t) accumulated into the same tile; bc holds bc + b_shortcut
#pragma unroll
          for (int k = 0; k < KS; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsc[(2 * np) * KS + k], xs[k], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsc[(2 * np + 1) * KS + k], xs[k], c1, 0, 0, 0);
          }
        } else {
          rs = *reinterpret_cast<const uint4*>(x0_ + c_res[it][np]);  // identity: the residual from the staged x
        }
        uint4 o;
        o.x = avt::pack_bf16x2(fmaxf(c0[0] + b0.x + avt::bf16x2_lo(rs.x), 0.f), fmaxf(c0[1] + b0.y + avt::bf16x2_hi(rs.x), 0.f));
        o.y = avt::pack_bf16x2(fmaxf(c0[2] + b0.z + avt::bf16x2_lo(rs.y), 0.f), fmaxf(c0[3] + b0.w + avt::bf16x2_hi(rs.y), 0.f));
        o.z = avt::pack_bf16x2(fmaxf(c1[0] + b1.x + avt::bf16x2_lo(rs.z), 0.f), fmaxf(c1[1] + b1.y + avt::bf16x2_hi(rs.z), 0.f));
        o.w = avt::pack_bf16x2(fmaxf(c1[2] + b1.z + avt::bf16x2_lo(rs.w), 0.f), fmaxf(c1[3] + b1.w + avt::bf16x2_hi(rs.w), 0.f));
        if ((c_ok >> it) & 1u) *reinterpret_cast<uint4*>(oframe + c_out[it] + 32 * np) = o;
      }
    }
  }
}

template <int C, int W, int HT, int CMP, int CIN = C, int ST = 1, int NW = 8>
int launch(BArgs& a, int batch, int h, hipStream_t st) {
  constexpr int RX = ST * HT + (ST == 1 ? 2 : 1), PX = RX * W, PPI = 1024 / (CIN * 2), NDMA = (PX + PPI - 1) / PPI;
  constexpr int MTB = (HT * (W / ST) + 15) / 16, AW = W + 2, NTA = CMP / 16, NB = CMP == 16 ? 5 : 9;
  constexpr int wfrags = CMP > 16 ? (3 * (CIN / 32) * NTA + NB * NTA) : 0;
  constexpr int lds_bytes = 3 * NDMA * 1024 + (RX * AW + 32) * CMP * 2 + MTB * 16 * CMP * 2 + wfrags * 1024;
  static_assert(lds_bytes <= 160 * 1024, "strip does not fit the LDS");
  a.strips = (h / ST + HT - 1) / HT;
  a.swz = 1;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_kernel<C, W, HT, CMP, CIN, ST, NW>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("avt_bottleneck_fused_bf16: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((bottleneck_kernel<C, W, HT, CMP, CIN, ST, NW>), dim3((unsigned)(batch * a.strips * a.tchunks)), dim3(NW * 64),
                     lds_bytes, st, a);
  return avt::check_launch("avt_bottleneck_fused_bf16");
}

}  // namespace

extern "C" int avt_bottleneck_fused_supported(int c, int w) {
  return ((c == 32 && (w == 56 || w == 12)) || (c == 64 && (w == 28 || w == 10)) || (c == 128 && (w == 14 || w == 6))) ? 1 : 0;
}

static int run_bottleneck(const char* what, const void* x, void* out, const void* wa, const float* ba, const void* wb,
                          const float* bb, const void* wc, const void* wsc, const float* bc, int batch, int t, int h, int w,
                          int cin, int c, int tchunk, void* stream) {
  AVT_REQUIRE(x && out && wa && ba && wb && bb && wc && bc && (cin == c || wsc), "%s: NULL pointer", what);
  AVT_REQUIRE(batch > 0 && t > 0 && h > 0 && tchunk > 0, "%s: bad sizes", what);
  AVT_REQUIRE(x != out, "%s: in-place is not supported (neighbouring strips read x)", what);
  AVT_REQUIRE(avt::aligned16(x) && avt::aligned16(out) && avt::aligned16(wa) && avt::aligned16(wb) && avt::aligned16(wc) &&
                  avt::aligned16(ba) && avt::aligned16(bb) && avt::aligned16(bc) && (!wsc || avt::aligned16(wsc)),
              "%s: pointers must be 16-byte aligned", what);
  const int64_t xb = (int64_t)batch * t * h * w * cin * 2, ob = (int64_t)batch * t * h * w * c * 2;
  AVT_REQUIRE(xb < (1ll << 32) - 64 && ob < (1ll << 33), "%s: tensor too large for 32-bit offsets", what);
  BArgs a;
  a.x = static_cast<const uint16_t*>(x);
  a.out = static_cast<uint16_t*>(out);
  a.wa = static_cast<const i32x4*>(wa);
  a.wb = static_cast<const i32x4*>(wb);
  a.wc = static_cast<const i32x4*>(wc);
  a.wsc = static_cast<const i32x4*>(wsc);
  a.ba = ba;
  a.bb = bb;
  a.bc = bc;
  a.T = t;
  a.H = h;
  a.TC = tchunk < t ? tchunk : t;
  a.tchunks = (t + a.TC - 1) / a.TC;
  a.x_bytes = (unsigned)xb;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // 16 waves (4 per SIMD, <= 128 VGPRs) hide the stages' latencies better than 8 (+10 % on the C = 32 form) where the
  // kernel fits them without spilling (C <= 64)
#define AVT_BN_LAUNCH(...) launch<__VA_ARGS__, 16>(a, batch, h, s)
  if (cin == 8) {
    if (w == 56) return AVT_BN_LAUNCH(32, 56, 8, 16, 8, 1);
    return launch<32, 12, 5, 16, 8>(a, batch, h, s);
  }
  if (cin != c) {  // strided first blocks (w = input width)
    if (c == 64 && w == 56) return AVT_BN_LAUNCH(64, 56, 4, 16, 32, 2);
    if (c == 128 && w == 28) return launch<128, 28, 4, 32, 64, 2>(a, batch, h, s);  // wide form: 16 waves would spill
    if (c == 64 && w == 12) return launch<64, 12, 3, 16, 32, 2>(a, batch, h, s);
    return launch<128, 8, 2, 32, 64, 2>(a, batch, h, s);
  }
  if (c == 32 && w == 56) return AVT_BN_LAUNCH(32, 56, 8, 16, 32, 1);
  if (c == 64 && w == 28) return AVT_BN_LAUNCH(64, 28, 7, 16, 64, 1);
  if (c == 128 && w == 14) return launch<128, 14, 7, 32>(a, batch, h, s);
#undef AVT_BN_LAUNCH
  if (c == 32 && w == 12) return launch<32, 12, 5, 16>(a, batch, h, s);  // small shapes for the tests: ragged strips,
  if (c == 64 && w == 10) return launch<64, 10, 4, 16>(a, batch, h, s);  // partial tiles, partial DMA instructions
  return launch<128, 6, 3, 32>(a, batch, h, s);
}

extern "C" int avt_bottleneck_fused_bf16(const void* x, void* out, const void* wa, const float* ba, const void* wb,
                                         const float* bb, const void* wc, const float* bc, int batch, int t, int h, int w,
                                         int c, int tchunk, void* stream) {
  AVT_REQUIRE(avt_bottleneck_fused_supported(c, w),
              "avt_bottleneck_fused_bf16: unsupported shape C=%d W=%d (fast-pathway identity blocks: 32x56, 64x28, 128x14)", c, w);
  return run_bottleneck("avt_bottleneck_fused_bf16", x, out, wa, ba, wb, bb, wc, nullptr, bc, batch, t, h, w, c, c, tchunk,
                        stream);
}

extern "C" int avt_bottleneck_first_supported(int cin, int c, int w) {
  return ((cin == 8 && c == 32 && (w == 56 || w == 12)) || (cin == 32 && c == 64 && (w == 56 || w == 12)) ||
          (cin == 64 && c == 128 && (w == 28 || w == 8))) ? 1 : 0;
}

extern "C" int avt_bottleneck_first_bf16(const void* x, void* out, const void* wa, const float* ba, const void* wb,
                                         const float* bb, const void* wc, const void* wsc, const float* bc, int batch, int t,
                                         int h, int w, int cin, int c, int tchunk, void* stream) {
  AVT_REQUIRE(avt_bottleneck_first_supported(cin, c, w),
              "avt_bottleneck_first_bf16: unsupported shape Cin=%d C=%d W=%d (first fast-pathway blocks: 8->32 W 56; strided 32->64 W 56, 64->128 W 28)",
              cin, c, w);
  return run_bottleneck("avt_bottleneck_first_bf16", x, out, wa, ba, wb, bb, wc, wsc, bc, batch, t, h, w, cin, c, tchunk,
                        stream);
}
