// row_transition / row_topk — per-row transition select for gfx950.
// Replaces the CPU torch row post-process of the reference stitch loop
// (contrastive_video_textures/validate.py:524-572) and the target ordering of
// validate.py:369-378, for all query rows in one launch.
//
// HBM-bound (nq*nt*4 B read once); one 256-thread workgroup per row.  The row's
// p values live in LDS (rows up to 16384 entries) so the four passes
// (sum -> p,max -> exp-sum,survivor-sum -> ordered compaction) touch HBM once.
// Reductions follow the canonical rounding of oracle/avt_oracle.c: sums in
// fp64 rounded to fp32 once, every other step one correctly rounded fp32 op
// (__fdiv_rn/__fmul_rn/... so the compiler cannot contract or reassociate).
#include <math.h>

#include "avt_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxLds = 16384;  // row entries cached in LDS (64 KB)

__device__ __forceinline__ int64_t target_len(int64_t q, int64_t n_seg) {
  const int64_t pos = q + 1 < n_seg - 1 ? q + 1 : n_seg - 1;
  return pos == q ? n_seg : n_seg - 1;
}
// position in the reference's [pos] + others order -> segment id (validate.py:369-378)
__device__ __forceinline__ int64_t target_seg(int64_t q, int64_t n_seg, int64_t position) {
  const int64_t pos = q + 1 < n_seg - 1 ? q + 1 : n_seg - 1;
  if (position == 0) return pos;
  const int64_t lo = q < pos ? q : pos, hi = q < pos ? pos : q;
  int64_t id = position - 1;
  if (id >= lo) ++id;
  if (hi != lo && id >= hi) ++id;
  return id;
}

struct BlockRed {
  double d[kThreads / 64];
  float f[kThreads / 64];
  int i[kThreads / 64];
};

__device__ __forceinline__ double block_sum(double v, BlockRed& r) {
  v = avt::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) r.d[threadIdx.x >> 6] = v;
  __syncthreads();
  return (r.d[0] + r.d[1]) + (r.d[2] + r.d[3]);
}
__device__ __forceinline__ float block_max(float v, BlockRed& r) {
  v = avt::wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) r.f[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(r.f[0], r.f[1]), fmaxf(r.f[2], r.f[3]));
}

struct TArgs {
  const float* sim;
  const float* sim_a;
  const int64_t* q_ids;
  int64_t nq, nt, ld, ld_a, n_seg;
  float af, bf, threshold;
  int cap;
  int32_t* surv_idx;
  int32_t* surv_seg;
  float* surv_p;
  int32_t* surv_cnt;
  float* stats;
};

template <bool CACHE>
__global__ __launch_bounds__(kThreads) void row_transition_kernel(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pbuf[];  // CACHE: L floats
  __shared__ BlockRed red;
  __shared__ int s_base;
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x;
  const float* x = a.sim + r * a.ld;
  const float* xa = a.sim_a ? a.sim_a + r * a.ld_a : nullptr;
  const bool perm = a.q_ids != nullptr;
  const int64_t q = perm ? a.q_ids[r] : -1;
  const int64_t L = perm ? target_len(q, a.n_seg) : a.nt;
  auto col = [&](int64_t i) { return perm ? target_seg(q, a.n_seg, i) : i; };

  // pass 1: row sums (validate.py:524, :526)
  double s = 0.0, sa = 0.0;
  for (int64_t i = tid; i < L; i += kThreads) {
    const int64_t c = col(i);
    s += (double)x[c];
    if (xa) sa += (double)xa[c];
  }
  s = block_sum(s, red);
  if (xa) sa = block_sum(sa, red);
  const float sf = (float)s, saf = (float)sa;

  auto pval = [&](int64_t i) {
    const int64_t c = col(i);
    float v = __fdiv_rn(x[c], sf);
    if (xa) {  // validate.py:527  alpha*p + (1-alpha)*p_a, two roundings + one add
      const float va = __fdiv_rn(xa[c], saf);
      v = __fadd_rn(__fmul_rn(a.af, v), __fmul_rn(a.bf, va));
    }
    return v;
  };
  auto getp = [&](int64_t i) { return CACHE ? pbuf[i] : pval(i); };

  // pass 2: p and its max
  float mx = -INFINITY;
  for (int64_t i = tid; i < L; i += kThreads) {
    const float v = pval(i);
    if (CACHE) pbuf[i] = v;
    mx = fmaxf(mx, v);
  }
  mx = block_max(mx, red);  // (the barriers inside also publish pbuf)
  const float p0 = getp(0);

  // pass 3: CE denominator (validate.py:531) and the survivors' sum (validate.py:554-558)
  const float cut = __fsub_rn(mx, __fmul_rn(a.threshold, mx));
  double se = 0.0, s2 = 0.0;
  for (int64_t i = tid; i < L; i += kThreads) {
    const float v = getp(i);
    se += (double)__expf(v - mx);  // CE is a reporting value (validate.py:531): fast fp32 exp, fp64 sum
    if (!(v < cut)) s2 += (double)v;
  }
  se = block_sum(se, red);
  s2 = block_sum(s2, red);
  const float s2f = (float)s2;

  // pass 4: ordered compaction of nonzero(p) (validate.py:558-568)
  if (tid == 0) s_base = 0;
  __syncthreads();
  double el = 0.0;
  const int lane = tid & 63, wid = tid >> 6;
  for (int64_t i0 = 0; i0 < L; i0 += kThreads) {
    const int64_t i = i0 + tid;
    float pn = 0.0f;
    bool keep = false;
    if (i < L) {
      const float v = getp(i);
      if (!(v < cut) && v != 0.0f) {
        pn = __fdiv_rn(v, s2f);
        keep = pn != 0.0f;
      }
    }
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) red.i[wid] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += red.i[w];
    const int total = red.i[0] + red.i[1] + red.i[2] + red.i[3];
    const int base = s_base;
    if (keep) {
      const int slot = base + woff + before;
      if (slot < a.cap) {
        const int64_t o = r * (int64_t)a.cap + slot;
        if (a.surv_idx) a.surv_idx[o] = (int32_t)i;
        if (a.surv_seg) a.surv_seg[o] = (int32_t)col(i);
        if (a.surv_p) a.surv_p[o] = pn;
      }
      el += log((double)pn);
    }
    __syncthreads();
    if (tid == 0) s_base = base + total;
    __syncthreads();
  }
  el = block_sum(el, red);
  if (tid == 0) {
    const int cnt = s_base;
    if (a.surv_cnt) a.surv_cnt[r] = cnt;
    if (a.stats) {
      a.stats[r * 4 + 0] = sf;
      a.stats[r * 4 + 1] = mx;
      a.stats[r * 4 + 2] = (float)((double)mx + log(se) - (double)p0);
      a.stats[r * 4 + 3] = cnt ? (float)fabs(el / cnt) : 0.0f;
    }
  }
}

// ---- fast path: rows of up to 256*ITEMS entries live in registers ---------------------------------------------
// Same arithmetic, same order of survivors; the row is read from HBM exactly once (coalesced: thread t owns
// positions t, t+256, ...), every reduction is one shuffle tree + one LDS hop, and the ordered compaction needs a
// single table of ITEMS x 4 ballot counts instead of one barrier round per 256 positions.
// NTHR = 256: rows up to 4096 entries; NTHR = 1024 (16 waves): rows up to 16384 entries — config 4's N = 16384, where the
// LDS-cached form below spent its time in 64 barrier rounds of compaction per row (0.4 TB/s).
template <int ITEMS, int NTHR = kThreads>
__global__ __launch_bounds__(NTHR) void row_transition_reg_kernel(TArgs a) {
  constexpr int NWV = NTHR / 64;
  __shared__ double red_d[NWV], red2[NWV];
  __shared__ float red_f[NWV];
  __shared__ int cnt_tab[ITEMS * NWV + 1];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // block reductions over NWV waves; fp64 partials are added in wave order, pairwise for NWV = 4 (the order the 256-thread
  // form has always used)
  auto bsum = [&](double v) {
    v = avt::wave_sum(v);
    __syncthreads();
    if (lane == 0) red_d[wid] = v;
    __syncthreads();
    if constexpr (NWV == 4) return (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; w += 2) t += red_d[w] + red_d[w + 1];
    return t;
  };
  auto bmax = [&](float v) {
    v = avt::wave_max(v);
    __syncthreads();
    if (lane == 0) red_f[wid] = v;
    __syncthreads();
    float t = red_f[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) t = fmaxf(t, red_f[w]);
    return t;
  };
  const float* x = a.sim + (int64_t)r * a.ld;
  const float* xa = a.sim_a ? a.sim_a + (int64_t)r * a.ld_a : nullptr;
  const bool perm = a.q_ids != nullptr;
  const int n_seg = (int)a.n_seg;
  const int q = perm ? (int)a.q_ids[r] : -1;
  const int pos = q + 1 < n_seg - 1 ? q + 1 : n_seg - 1;
  const int lo = q < pos ? q : pos, hi = q < pos ? pos : q;
  const int L = perm ? (pos == q ? n_seg : n_seg - 1) : (int)a.nt;
  auto col = [&](int i) {
    if (!perm) return i;
    if (i == 0) return pos;
    int id = i - 1;
    if (id >= lo) ++id;
    if (hi != lo && id >= hi) ++id;
    return id;
  };
  float v[ITEMS], va[ITEMS];
  double s = 0.0, sa = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = tid + k * NTHR;
    v[k] = 0.0f;
    va[k] = 0.0f;
    if (i < L) {
      const int c = col(i);
      v[k] = x[c];
      s += (double)v[k];
      if (xa) {
        va[k] = xa[c];
        sa += (double)va[k];
      }
    }
  }
  s = bsum(s);
  if (xa) sa = bsum(sa);
  const float sf = (float)s, saf = (float)sa;
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = tid + k * NTHR;
    float p = __fdiv_rn(v[k], sf);
    if (xa) p = __fadd_rn(__fmul_rn(a.af, p), __fmul_rn(a.bf, __fdiv_rn(va[k], saf)));
    v[k] = p;
    if (i < L) mx = fmaxf(mx, p);
  }
  mx = bmax(mx);
  const float cut = __fsub_rn(mx, __fmul_rn(a.threshold, mx));
  double se = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = tid + k * NTHR;
    if (i < L) {
      se += (double)__expf(v[k] - mx);  // CE is a reporting value: fast exp, fp64 sum
      if (!(v[k] < cut)) s2 += (double)v[k];
    }
  }
  // two sums in one LDS hop
  se = avt::wave_sum(se);
  s2 = avt::wave_sum(s2);
  __syncthreads();
  if (lane == 0) {
    red_d[wid] = se;
    red2[wid] = s2;
  }
  __syncthreads();
  if constexpr (NWV == 4) {
    se = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
    s2 = (red2[0] + red2[1]) + (red2[2] + red2[3]);
  } else {
    se = s2 = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; w += 2) {
      se += red_d[w] + red_d[w + 1];
      s2 += red2[w] + red2[w + 1];
    }
  }
  const float s2f = (float)s2;
  const float p0 = __shfl(v[0], 0, 64);  // position 0 lives in thread 0 (wave 0); broadcast below through LDS
  // survivors: ballot per (round k, wave) -> exclusive offsets from one table
  unsigned long long bal[ITEMS];
  float pn[ITEMS];
  double el = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = tid + k * NTHR;
    pn[k] = 0.0f;
    bool keep = false;
    if (i < L && !(v[k] < cut) && v[k] != 0.0f) {
      pn[k] = __fdiv_rn(v[k], s2f);
      keep = pn[k] != 0.0f;
    }
    bal[k] = __ballot(keep);
    el += keep ? (double)__logf(pn[k]) : 0.0;  // entropy is a reporting value: fast fp32 log, no divergent fp64 libm
    if (lane == 0) cnt_tab[k * NWV + wid] = __popcll(bal[k]);
  }
  __shared__ float s_p0;
  if (tid == 0) s_p0 = p0;
  __syncthreads();  // counts (and p0) are in LDS
  {                 // exclusive scan of the ITEMS * NWV counts (position order: round k, then wave) by wave 0; barriers stay uniform
    constexpr int n = ITEMS * NWV, PER = (n + 63) / 64;  // PER consecutive entries per lane
    int c[PER], incl = 0;
    if (wid == 0) {
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        c[e] = lane * PER + e < n ? cnt_tab[lane * PER + e] : 0;
        incl += c[e];
      }
      const int mine = incl;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      incl -= mine;  // exclusive over lanes
    }
    __syncthreads();
    if (wid == 0) {
      int run = incl;
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        if (lane * PER + e < n) cnt_tab[lane * PER + e] = run;
        run += c[e];
      }
      if (lane == 63) cnt_tab[n] = run;
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    if ((bal[k] >> lane) & 1ull) {
      const int slot = cnt_tab[k * NWV + wid] + __popcll(bal[k] & ((1ull << lane) - 1ull));
      if (slot < a.cap) {
        const int i = tid + k * NTHR;
        const int64_t o = (int64_t)r * a.cap + slot;
        if (a.surv_idx) a.surv_idx[o] = i;
        if (a.surv_seg) a.surv_seg[o] = col(i);
        if (a.surv_p) a.surv_p[o] = pn[k];
      }
    }
  }
  const int cnt = cnt_tab[ITEMS * NWV];
  const float p_first = s_p0;
  el = bsum(el);
  if (tid == 0) {
    if (a.surv_cnt) a.surv_cnt[r] = cnt;
    if (a.stats) {
      a.stats[r * 4 + 0] = sf;
      a.stats[r * 4 + 1] = mx;
      a.stats[r * 4 + 2] = (float)((double)mx + log(se) - (double)p_first);
      a.stats[r * 4 + 3] = cnt ? (float)fabs(el / cnt) : 0.0f;
    }
  }
}

// ---- top-k ------------------------------------------------------------------
struct KArgs {
  const float* sim;
  const int64_t* self_col;
  int64_t nq, nt, ld;
  int k;
  int32_t* top_idx;
  float* top_val;
};

__global__ __launch_bounds__(kThreads) void row_topk_kernel(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) float row[];  // nt floats
  __shared__ float s_v[kThreads / 64];
  __shared__ int s_i[kThreads / 64];
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* x = a.sim + r * a.ld;
  const int64_t self = a.self_col ? a.self_col[r] : -1;
  // taken / excluded entries are marked NaN so that a genuine -inf score can still be picked once
  for (int64_t j = tid; j < a.nt; j += kThreads) row[j] = (j == self) ? __builtin_nanf("") : x[j];
  __syncthreads();
  for (int s = 0; s < a.k; ++s) {
    float bv = -INFINITY;
    int bi = -1;
    for (int64_t j = tid; j < a.nt; j += kThreads) {
      const float v = row[j];
      if (v == v && (bi < 0 || v > bv)) {  // strided ascending j: first max kept -> lowest column on ties
        bv = v;
        bi = (int)j;
      }
    }
    // reduce (value desc, index asc); bi < 0 means "nothing left"
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      s_v[wid] = bv;
      s_i[wid] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      float fv = s_v[0];
      int fi = s_i[0];
      for (int w = 1; w < kThreads / 64; ++w)
        if (s_i[w] >= 0 && (fi < 0 || s_v[w] > fv || (s_v[w] == fv && s_i[w] < fi))) {
          fv = s_v[w];
          fi = s_i[w];
        }
      a.top_idx[r * (int64_t)a.k + s] = fi;
      a.top_val[r * (int64_t)a.k + s] = fi >= 0 ? fv : -INFINITY;
      if (fi >= 0) row[fi] = __builtin_nanf("");
    }
    __syncthreads();
  }
}

// Register-resident top-k: thread t holds positions t, t + NTHR, ... (the row is read from HBM once, coalesced); a round is a
// per-thread scan of its ITEMS registers + one shuffle tree + one LDS hop, the winner's owner marks its register taken.
// Same order as the LDS form: value descending, lowest column on ties; excluded / taken entries are NaN.
template <int ITEMS, int NTHR>
__global__ __launch_bounds__(NTHR) void row_topk_reg_kernel(KArgs a) {
  constexpr int NWV = NTHR / 64;
  __shared__ float s_v[NWV];
  __shared__ int s_i[NWV];
  __shared__ int s_win;
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* x = a.sim + r * a.ld;
  const int self = a.self_col ? (int)a.self_col[r] : -1;
  const int nt = (int)a.nt;
  float v[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int j = tid + k * NTHR;
    v[k] = (j < nt && j != self) ? x[j] : __builtin_nanf("");
  }
  for (int s = 0; s < a.k; ++s) {
    float bv = -INFINITY;
    int bi = -1;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {  // ascending j: the first maximum is kept -> lowest column on ties
      const float u = v[k];
      if (u == u && (bi < 0 || u > bv)) {
        bv = u;
        bi = tid + k * NTHR;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      s_v[wid] = bv;
      s_i[wid] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      float fv = s_v[0];
      int fi = s_i[0];
      for (int w = 1; w < NWV; ++w)
        if (s_i[w] >= 0 && (fi < 0 || s_v[w] > fv || (s_v[w] == fv && s_i[w] < fi))) {
          fv = s_v[w];
          fi = s_i[w];
        }
      a.top_idx[r * (int64_t)a.k + s] = fi;
      a.top_val[r * (int64_t)a.k + s] = fi >= 0 ? fv : -INFINITY;
      s_win = fi;
    }
    __syncthreads();
    const int win = s_win;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
      if (tid + k * NTHR == win) v[k] = __builtin_nanf("");
    // (the next round's first barrier orders the s_win read above before its rewrite)
  }
}

}  // namespace

extern "C" int avt_row_transition(const float* sim, int64_t nq, int64_t nt, int64_t ld, const int64_t* q_ids,
                                  int64_t n_seg, const float* sim_a, int64_t ld_a, float alpha, float threshold,
                                  int cap, int32_t* surv_idx, int32_t* surv_seg, float* surv_p, int32_t* surv_cnt,
                                  float* stats, void* stream) {
  AVT_REQUIRE(nq >= 0 && nt > 0 && ld >= nt, "avt_row_transition: bad sizes nq=%lld nt=%lld ld=%lld", (long long)nq,
              (long long)nt, (long long)ld);
  if (nq == 0) return AVT_OK;
  AVT_REQUIRE(sim, "avt_row_transition: sim is NULL");
  AVT_REQUIRE(!q_ids || (n_seg == nt && n_seg >= 2), "avt_row_transition: with q_ids, nt must equal n_seg >= 2");
  AVT_REQUIRE(!sim_a || ld_a >= nt, "avt_row_transition: ld_a < nt");
  AVT_REQUIRE(cap >= 0 && (cap == 0 || surv_idx || surv_seg || surv_p), "avt_row_transition: cap/outputs mismatch");
  if (nq == 0) return AVT_OK;
  TArgs a;
  a.sim = sim;
  a.sim_a = sim_a;
  a.q_ids = q_ids;
  a.nq = nq;
  a.nt = nt;
  a.ld = ld;
  a.ld_a = ld_a;
  a.n_seg = n_seg;
  a.af = alpha;
  a.bf = (float)(1.0 - (double)alpha);
  a.threshold = threshold;
  a.cap = cap;
  a.surv_idx = surv_idx;
  a.surv_seg = surv_seg;
  a.surv_p = surv_p;
  a.surv_cnt = surv_cnt;
  a.stats = stats;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)nq), block(kThreads);
  if (nt <= 8 * kThreads)
    hipLaunchKernelGGL(row_transition_reg_kernel<8>, grid, block, 0, st, a);
  else if (nt <= 16 * kThreads)
    hipLaunchKernelGGL(row_transition_reg_kernel<16>, grid, block, 0, st, a);
  else if (nt <= 8 * 1024)
    hipLaunchKernelGGL((row_transition_reg_kernel<8, 1024>), grid, dim3(1024), 0, st, a);
  else if (nt <= 16 * 1024)
    hipLaunchKernelGGL((row_transition_reg_kernel<16, 1024>), grid, dim3(1024), 0, st, a);
  else if (nt <= kMaxLds)
    hipLaunchKernelGGL(row_transition_kernel<true>, grid, block, (size_t)nt * sizeof(float), st, a);
  else
    hipLaunchKernelGGL(row_transition_kernel<false>, grid, block, 0, st, a);
  return avt::check_launch("avt_row_transition");
}

extern "C" int avt_row_topk(const float* sim, int64_t nq, int64_t nt, int64_t ld, const int64_t* self_col, int k,
                            int32_t* top_idx, float* top_val, void* stream) {
  AVT_REQUIRE(nq >= 0 && nt > 0 && ld >= nt && k > 0, "avt_row_topk: bad sizes");
  if (nq == 0) return AVT_OK;
  AVT_REQUIRE(sim && top_idx && top_val, "avt_row_topk: NULL pointer");
  if (nt > kMaxLds) {
    avt::set_error("avt_row_topk: nt=%lld exceeds the LDS-resident row limit %d", (long long)nt, kMaxLds);
    return AVT_ERR_UNSUPPORTED;
  }
  if (nq == 0) return AVT_OK;
  KArgs a{sim, self_col, nq, nt, ld, k, top_idx, top_val};
  hipStream_t st_ = static_cast<hipStream_t>(stream);
  if (nt <= 16 * 256) {  // register-resident forms up to 16384-wide rows, the LDS-resident form beyond
    hipLaunchKernelGGL((row_topk_reg_kernel<16, 256>), dim3((unsigned)nq), dim3(256), 0, st_, a);
    return avt::check_launch("avt_row_topk");
  }
  if (nt <= 16 * 1024) {
    hipLaunchKernelGGL((row_topk_reg_kernel<16, 1024>), dim3((unsigned)nq), dim3(1024), 0, st_, a);
    return avt::check_launch("avt_row_topk");
  }
  hipLaunchKernelGGL(row_topk_kernel, dim3((unsigned)nq), dim3(kThreads), (size_t)nt * sizeof(float),
                     static_cast<hipStream_t>(stream), a);
  return avt::check_launch("avt_row_topk");
}
