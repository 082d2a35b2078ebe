// Shared argument block and host-side validation of the implicit-GEMM convolution kernels
// (conv_igemm.hip: bf16 tiles; conv_x3.hip: split-bf16 "x3" tiles with fp32-grade activations).
#pragma once
#include "avt_common.h"

namespace {

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
  const uint32_t t = __umulhi(n, f.magic);
  const uint32_t q = (t + ((n - t) >> 1)) >> f.shift;
  return f.d == 1 ? n : q;  // a select, not a branch: keeps the caller's loop body one scheduling region
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
  FastDiv dCpt, dKHW, dKW;  // tap decode without the table (XL kernel): chunks per tap, KH*KW, KW
  FastDiv dNT;              // taps (XL kernel, taps-innermost K walk)
  int tapinner;
  unsigned in_bytes, wt_bytes;  // extents for the buffer descriptors (hardware range check = free zero fill)
  int pointwise;                // 1x1x1 / stride 1 / pad 0: rows need no decode
  const uint16_t* wfrag;        // XB kernel: weights in MFMA-fragment order [Cout/32][nup][2][64][8] (walk order of K), or NULL
  int nup;                      // ... units per 32-row tile in that array (>= units walked + 3)
  int wblk;                     // x3 XL tile: weight planes in K-blocked order [K / 32][Cout][32] (avt_conv3d_igemm_x3_wblk)
  unsigned wf_bytes;
  int ors, oH, oW;              // output row remap: position (f, ho, wo) -> row (f * oH + ors * ho) * oW + ors * wo (ors = 1: none)
  // split-bf16 ("x3") kernels: the low-order planes, same geometry as in / wt / res / out (conv_x3.hip)
  const uint16_t* in_lo;
  const uint16_t* wt_lo;
  const uint16_t* res_lo;
  uint16_t* out_lo;
  const float* wscale;  // per-output-channel power-of-two factor applied to the accumulator (fp16 planes: weights are stored
                        // pre-scaled into the fp16 normal range), or NULL
  float odiv;           // x3 256 x 256 tile with fp32 output from plane inputs (avt_gemm_nt_x3_f32out): every result is DIVIDED by this
                        // (the similarity's temperature: out = <q, t> / temp with one correctly rounded division), 0 = off
  int cf_ofs;           // x3 kernels: LDS byte offset of the tile's [bias BN | scale BN] floats (set by the launchers; filled in the
                        // prologue, read by the epilogue from LDS instead of one global load per accumulator group)
  // BatchNorm statistics on the epilogue (round 5; IO32 tiles, avt_conv3d_igemm_x3_f32_stats): the layer's output feeds a train-mode
  // BatchNorm, whose statistics pass would re-read it.  The rows are `stat_groups` slabs of `stat_mg` rows with statistics of their
  // own (bn_train.hip GROUPS); the M tiles are laid PER GROUP (stat_tpg tiles each, the last one short) so that no tile straddles
  // two groups, and every tile leaves the per-channel sum / sum of squares of its rows in `stat_part` in the layout
  // bn_fwd_finalize_kernel sums over (one row of partials per tile).  stat_part = NULL: off, tiles walk the rows 0 .. M.
  double* stat_part;
  int stat_groups, stat_mg, stat_tpg;
  int stat_c;           // BatchNorm channels: Cout, or Cout / g for the pixel-grouped form (columns n and n + stat_c are one channel)
  // BatchNorm BACKWARD statistics on the epilogue of an input-gradient launch (round 5; avt_conv3d_igemm_x3_f32_bwdstats): the
  // launch's output dz is the gradient of a train-mode BatchNorm (+ ReLU)'s OUTPUT.  The epilogue reads that BatchNorm's input
  // rows (bst_x, same geometry as the output), rebuilds the ReLU mask — from x with the forward's own expression, or from the
  // 4-bits-per-chunk mask the forward saved when there was a shortcut — stores g = mask * dz (what the BatchNorm's apply pass
  // wants), and leaves the per-tile sums of g and g * xhat in stat_part: the BatchNorm's backward statistics pass (dy and x read
  // once more) is gone.  bst_x = NULL: off (stat_part then means the FORWARD statistics above).
  const float* bst_x;
  const float* bst_mean;    // [groups][C]
  const float* bst_invstd;  // [groups][C]
  const float* bst_gamma;   // [C]
  const float* bst_beta;    // [C]: mask recomputed from x ((x - mean) * (invstd * gamma) + beta > 0); unused with bst_mask
  const uint8_t* bst_mask;  // [M * C / 4] the forward's saved mask bits, or NULL
  int bst_relu;             // 0: no activation (g = dz)
};

// The eight per-channel coefficients of one thread (its channel octet is the same in every row it stores) and the masking /
// accumulation of one row's octet for the backward statistics.
struct BstCoef {
  float mu[8], sc[8], be[8];  // (invstd itself multiplies the folded sum once per channel: stat_fold_store)
};
__device__ __forceinline__ void bst_load(const ConvArgs& a, int group, int n, BstCoef& k) {
  const int C = a.stat_c;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = (n + e) % C;  // (pixel-grouped form: column n is channel n % C)
    k.mu[e] = a.bst_mean[(size_t)group * C + c];
    k.sc[e] = a.bst_invstd[(size_t)group * C + c] * a.bst_gamma[c];   // the forward's scale: the same product, the same rounding
    k.be[e] = a.bst_beta ? a.bst_beta[c] : 0.0f;
  }
}
// x[8] = dz of (row m, columns n ..): -> g in place; s / q += g, g * (x - mean).  xv = the BatchNorm input's octet, bits = its two mask nibbles
// (The launch-uniform choices — no activation / saved mask bits / recomputed mask — are folded into the OPERANDS of one data-flow select
//  per element.  Written as `if (relu) keep = mask ? bit : recomputed; g = keep ? x : 0`, hipcc 7.2 lowered the three-way uniform
//  branch around a divergent select to "g = 0; if (keep) {}" — every gradient zeroed; found by tools/experimental/debug_bwdstats.py.)
__device__ __forceinline__ void bst_apply(const ConvArgs& a, const BstCoef& k, const float* xv, unsigned bits, float* x, float* s, float* q) {
  const bool by_bits = a.bst_relu != 0 && a.bst_mask != nullptr, by_x = a.bst_relu != 0 && a.bst_mask == nullptr;
  const unsigned mbits = by_bits ? bits : 0xFFu;  // not masked by bits: all ones
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float xc = xv[e] - k.mu[e];
    const float act = by_x ? xc * k.sc[e] + k.be[e] : 1.0f;  // not masked by the recomputed activation: positive
    const float sel = (((mbits >> e) & 1u) != 0u && act > 0.0f) ? 1.0f : 0.0f;
    const float g = sel != 0.0f ? x[e] : 0.0f;
    x[e] = g;
    s[e] += g;
    q[e] += g * xc;
  }
}

// this tile's first row and the end of the rows it may touch (its group's end when the tiles are laid per group)
__device__ __forceinline__ void tile_rows(const ConvArgs& a, int tm, int bm, int& m0, int& m_end) {
  if (a.stat_part) {
    const int g = tm / a.stat_tpg, t = tm - g * a.stat_tpg;
    m0 = g * a.stat_mg + t * bm;
    m_end = (g + 1) * a.stat_mg;
  } else {
    m0 = tm * bm;
    m_end = a.M;
  }
}

// BatchNorm partials of one tile: `red` = [nrow][2][BN] floats in LDS (per thread-row sums / sums of squares of the tile's
// columns); thread i < 2 * nch folds the thread-rows (and, pixel-grouped form, the columns of one channel) in fp64 and writes the
// slot bn_train.hip's channel_sums reads: row (tm * unit + quad / 256), slot (quad % nq) * 8 + e (+ 4 for the squares)
template <int BN, int NTHR>
__device__ __forceinline__ void stat_fold_store(const ConvArgs& a, const float* red, int nrow, int tm, int n0, int tid) {
  const int C = a.stat_c;
  int ncols = a.Cout - n0;
  ncols = ncols < BN ? ncols : BN;
  const int nch = ncols < C ? ncols : C;
  const int q4 = C / 4, nq = q4 < 256 ? q4 : 256, unit = q4 > 256 ? q4 / 256 : 1;
  for (int i = tid; i < 2 * nch; i += NTHR) {
    const int qi = i / nch, ch = i - qi * nch;
    double acc = 0.0;
    for (int col = ch; col < ncols; col += C)
      for (int r = 0; r < nrow; ++r) acc += (double)red[(r * 2 + qi) * BN + col];
    const int c = (n0 + ch) % C, quad = c >> 2, e = c & 3;
    if (a.bst_x && qi == 1) acc *= (double)a.bst_invstd[(size_t)(tm / a.stat_tpg) * C + c];  // sum of g * (x - mean) -> of g * xhat
    a.stat_part[((size_t)tm * unit + (quad >> 8)) * nq * 8 + (size_t)(quad % nq) * 8 + e + 4 * qi] = acc;
  }
}



// Row of the output buffer an output position lands in.  ors > 1: the layer writes into (a channel slice of) a buffer
// laid out over a `ors`x finer grid — a stride-2 [1,3,3] conv writing behind the channels of ITS OWN input's rows, so
// that the block's c conv and strided shortcut conv become one GEMM over K = [x | b-output] (fused_slowfast._Block).
__device__ __forceinline__ int out_row(const ConvArgs& a, int m) {
  if (a.oH == 0) return m;
  const int t1 = (int)fastdiv((uint32_t)m, a.dWo), wo = m - t1 * a.Wo;
  const int t2 = (int)fastdiv((uint32_t)t1, a.dHo), ho = t1 - t2 * a.Ho;
  return (t2 * a.oH + ho * a.ors) * a.oW + wo * a.ors;
}


// Validates the C-ABI arguments and fills the launch-independent part of ConvArgs (one bf16 plane per tensor; the x3
// entry adds its low-order planes afterwards).  `who` names the entry point in error messages.
inline int conv_args_fill(ConvArgs& a, const char* who, const void* in, const void* wt, const float* bias,
                          const void* res, void* out, const int32_t* ktab, int batch, int t, int h, int w, int cin,
                          int cout, int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho,
                          int wo, int ldi, int ldo, int ldr, int relu, int out_row_stride, int out_h, int out_w) {
  AVT_REQUIRE(in && wt && out && ktab, "%s: NULL pointer", who);
  AVT_REQUIRE(cin > 0 && cin % 8 == 0 && cout > 0 && cout % 8 == 0, "%s: Cin/Cout must be multiples of 8", who);
  AVT_REQUIRE(kt >= 1 && kh >= 1 && kw >= 1 && kt <= 8 && kh <= 8 && kw <= 8, "%s: kernel extents must be 1..8", who);
  AVT_REQUIRE(ldi % 8 == 0 && ldo % 8 == 0 && (!res || ldr % 8 == 0) && ldi >= cin && ldo >= cout,
              "%s: leading dimensions must be multiples of 8 and cover the channels", who);
  AVT_REQUIRE(avt::aligned16(in) && avt::aligned16(wt) && avt::aligned16(out) && (!res || avt::aligned16(res)) &&
                  (!bias || avt::aligned16(bias)),
              "%s: pointers must be 16-byte aligned", who);
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
  // (an explicit extent may also reach past the formula's by up to k - 1 positions: padding on the far side only — every tap is
  //  bounds-checked against the input by the kernels; avt_conv3d_igemm_x3_f32_ex)
  AVT_REQUIRE(a.To > 0 && a.Ho > 0 && a.Wo > 0 && batch > 0 && a.To <= fto + (kt - 1) / st && a.Ho <= fho + (kh - 1) / sh &&
                  a.Wo <= fwo + (kw - 1) / sw,
              "%s: bad output extent %dx%dx%d (max %dx%dx%d)", who, a.To, a.Ho, a.Wo, fto + (kt - 1) / st, fho + (kh - 1) / sh,
              fwo + (kw - 1) / sw);
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
              "%s: tensor too large for 32-bit offsets", who);
  a.in_bytes = (unsigned)((int64_t)batch * t * h * w * ldi * 2);
  a.wt_bytes = (unsigned)((int64_t)cout * a.K * 2);
  a.M = (int)M;
  a.nk = (a.K + BK - 1) / BK;
  a.pointwise = (kt == 1 && kh == 1 && kw == 1 && st == 1 && sh == 1 && sw == 1 && pt == 0 && ph == 0 && pw == 0 &&
                 a.To == t && a.Ho == h && a.Wo == w) ? 1 : 0;
  a.dWo = make_fastdiv((uint32_t)a.Wo);
  a.dHo = make_fastdiv((uint32_t)a.Ho);
  a.dTo = make_fastdiv((uint32_t)a.To);
  // (out_row_stride 1 with an out_h x out_w grid: the frame's rows land at the top-left of a larger frame — one temporal class
  //  of a strided transposed convolution, train_ops._dgrad_strided)
  const bool remap = out_row_stride > 1 || (out_h > 0 && out_w > 0);
  AVT_REQUIRE(out_row_stride >= 1 && (!remap || (!res && out_h >= out_row_stride * (a.Ho - 1) + 1 &&
                                                 out_w >= out_row_stride * (a.Wo - 1) + 1)),
              "%s: the remapped rows need an out_h x out_w grid that holds %d x (%d x %d), no residual", who,
              out_row_stride, a.Ho, a.Wo);
  AVT_REQUIRE(!remap || (int64_t)batch * a.To * out_h * out_w < (1ll << 31),
              "%s: output buffer too large for 32-bit rows", who);
  a.ors = out_row_stride;
  a.oH = remap ? out_h : 0;
  a.oW = remap ? out_w : 0;
  a.stat_part = nullptr;
  a.odiv = 0.0f;
  a.stat_groups = a.stat_mg = a.stat_tpg = a.stat_c = 0;
  a.bst_x = a.bst_mean = a.bst_invstd = a.bst_gamma = a.bst_beta = nullptr;
  a.bst_mask = nullptr;
  a.bst_relu = 0;
  return AVT_OK;
}

}  // namespace
