/*
 * avt_oracle.c — CPU restatement of the reference arithmetic of the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under audio-video-textures_amd/ may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, and only as the checker / the timed CPU
 * baseline.  The product path is the HIP library behind include/avt.h.
 *
 * Reference lines restated (all under /root/reference/contrastive_video_textures):
 *   avt_oracle_l2norm_rows     models/models.py:347-351, 408-412, 433-436
 *                              (torch.cat + F.normalize, eps 1e-12)
 *   avt_oracle_sim_*           models/models.py:416-417, 439, 455-457
 *                              (torch.bmm + `/= temp`)
 *   avt_oracle_row_transition  validate.py:369-378 (target order) and
 *                              validate.py:524-572 (row post-process)
 *   avt_oracle_softmax_ce_*    train.py:129-135 (nn.CrossEntropyLoss, label 0)
 *
 * Parity pin: tests/test_oracle_golden.py checks these functions against the
 * fixtures in tests/golden/ that tools/gen_golden.py produced by importing and
 * running the reference's own Python in the build container.
 *
 * Canonical rounding.  The reference runs fp32 torch ops whose reduction
 * order is a library detail (MKL / ATen vectorised sums), so two platforms of
 * the reference itself disagree in the last bit.  To make "bit-exact stitch
 * indices" a property that holds by construction, the build fixes ONE order
 * for every reduction and implements it identically here and on the GPU:
 *   - row sums / sums of squares: accumulate in fp64, round to fp32 once;
 *   - dot products (AVT_SIM_F32): the fp32 fmaf chain of the f32-input MFMA,
 *     k visited as 0,4,1,5,2,6,3,7 inside each group of 8, starting from +0;
 *   - every other step is a single correctly-rounded fp32 operation.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void avt_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#endif
}

int avt_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- bf16 helpers ------------------------------------------------------ */
static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* ---- l2norm ------------------------------------------------------------ */
void avt_oracle_l2norm_rows(const float* x0, int d0, const float* x1, int d1,
                            int64_t n, float eps, float* y_f32, uint16_t* y_hi,
                            uint16_t* y_lo) {
  const int d = d0 + d1;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    const float* a = x0 + r * (int64_t)d0;
    const float* b = x1 ? x1 + r * (int64_t)d1 : NULL;
    double ss = 0.0;
    for (int k = 0; k < d0; ++k) ss += (double)a[k] * (double)a[k];
    for (int k = 0; k < d1; ++k) ss += (double)b[k] * (double)b[k];
    float nrm = sqrtf((float)ss);
    float den = nrm > eps ? nrm : eps;
    for (int k = 0; k < d; ++k) {
      float v = (k < d0 ? a[k] : b[k - d0]) / den;
      if (y_f32) y_f32[r * (int64_t)d + k] = v;
      if (y_hi || y_lo) {
        uint16_t hi = f32_to_bf16_rne(v);
        if (y_hi) y_hi[r * (int64_t)d + k] = hi;
        if (y_lo) y_lo[r * (int64_t)d + k] = f32_to_bf16_rne(v - bf16_to_f32(hi));
      }
    }
  }
}

/* ---- similarity, canonical f32 fmaf chain ------------------------------- */
/* plain (slow) statement of the definition; used to check the blocked one */
void avt_oracle_sim_f32_naive(const float* q, const float* t, int64_t nq,
                              int64_t nt, int d, float temp, float* out,
                              int64_t ldo) {
  const int dp = (d + 7) & ~7;
  for (int64_t i = 0; i < nq; ++i)
    for (int64_t j = 0; j < nt; ++j) {
      float acc = 0.0f;
      for (int g = 0; g < dp; g += 8)
        for (int s = 0; s < 4; ++s) {
          int k0 = g + s, k1 = g + 4 + s;
          float a0 = k0 < d ? q[i * d + k0] : 0.0f, b0 = k0 < d ? t[j * d + k0] : 0.0f;
          float a1 = k1 < d ? q[i * d + k1] : 0.0f, b1 = k1 < d ? t[j * d + k1] : 0.0f;
          acc = fmaf(a0, b0, acc);
          acc = fmaf(a1, b1, acc);
        }
      out[i * ldo + j] = acc / temp;
    }
}

#define OT_I 64 /* rows per cache tile   */
#define OT_J 64 /* columns per cache tile */

void avt_oracle_sim_f32(const float* q, const float* t, int64_t nq, int64_t nt,
                        int d, float temp, float* out, int64_t ldo) {
  const int dp = (d + 7) & ~7;
  int* ord = (int*)malloc(sizeof(int) * dp);
  for (int g = 0; g < dp; g += 8)
    for (int s = 0; s < 4; ++s) {
      ord[g + 2 * s] = g + s;
      ord[g + 2 * s + 1] = g + 4 + s;
    }
  /* Q reordered along k: qr[i][kk] = q[i][ord[kk]] */
  float* qr = (float*)aligned_alloc(64, sizeof(float) * (size_t)nq * dp);
  /* T in 8-column panels, k-major inside a panel: tp[p][kk][8] */
  const int64_t npan = (nt + 7) / 8;
  float* tp = (float*)aligned_alloc(64, sizeof(float) * (size_t)npan * dp * 8);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nq; ++i)
    for (int kk = 0; kk < dp; ++kk) {
      int k = ord[kk];
      qr[i * dp + kk] = k < d ? q[i * (int64_t)d + k] : 0.0f;
    }
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < npan; ++p)
    for (int kk = 0; kk < dp; ++kk) {
      int k = ord[kk];
      for (int c = 0; c < 8; ++c) {
        int64_t j = p * 8 + c;
        tp[(p * dp + kk) * 8 + c] = (j < nt && k < d) ? t[j * (int64_t)d + k] : 0.0f;
      }
    }
  const int64_t ti = (nq + OT_I - 1) / OT_I, tj = (nt + OT_J - 1) / OT_J;
  const __m256 vtemp = _mm256_set1_ps(temp);
#pragma omp parallel for schedule(dynamic) collapse(2)
  for (int64_t bi = 0; bi < ti; ++bi)
    for (int64_t bj = 0; bj < tj; ++bj) {
      const int64_t i1 = (bi + 1) * OT_I < nq ? (bi + 1) * OT_I : nq;
      const int64_t p0 = bj * (OT_J / 8);
      const int64_t p1 = (p0 + OT_J / 8) < npan ? (p0 + OT_J / 8) : npan;
      for (int64_t p = p0; p < p1; p += 2) {
        const int two = (p + 1 < p1);
        const float* pa = tp + p * dp * 8;
        const float* pb = two ? pa + (size_t)dp * 8 : pa;
        for (int64_t i = bi * OT_I; i < i1; i += 4) {
          const int nr = (int)((i1 - i) < 4 ? (i1 - i) : 4);
          const float* r0 = qr + i * dp;
          const float* r1 = qr + (nr > 1 ? i + 1 : i) * dp;
          const float* r2 = qr + (nr > 2 ? i + 2 : i) * dp;
          const float* r3 = qr + (nr > 3 ? i + 3 : i) * dp;
          __m256 a00 = _mm256_setzero_ps(), a01 = a00, a10 = a00, a11 = a00;
          __m256 a20 = a00, a21 = a00, a30 = a00, a31 = a00;
          for (int kk = 0; kk < dp; ++kk) {
            const __m256 b0 = _mm256_load_ps(pa + kk * 8);
            const __m256 b1 = _mm256_load_ps(pb + kk * 8);
            __m256 s;
            s = _mm256_broadcast_ss(r0 + kk);
            a00 = _mm256_fmadd_ps(s, b0, a00); a01 = _mm256_fmadd_ps(s, b1, a01);
            s = _mm256_broadcast_ss(r1 + kk);
            a10 = _mm256_fmadd_ps(s, b0, a10); a11 = _mm256_fmadd_ps(s, b1, a11);
            s = _mm256_broadcast_ss(r2 + kk);
            a20 = _mm256_fmadd_ps(s, b0, a20); a21 = _mm256_fmadd_ps(s, b1, a21);
            s = _mm256_broadcast_ss(r3 + kk);
            a30 = _mm256_fmadd_ps(s, b0, a30); a31 = _mm256_fmadd_ps(s, b1, a31);
          }
          float buf[4][16];
          _mm256_storeu_ps(buf[0], _mm256_div_ps(a00, vtemp)); _mm256_storeu_ps(buf[0] + 8, _mm256_div_ps(a01, vtemp));
          _mm256_storeu_ps(buf[1], _mm256_div_ps(a10, vtemp)); _mm256_storeu_ps(buf[1] + 8, _mm256_div_ps(a11, vtemp));
          _mm256_storeu_ps(buf[2], _mm256_div_ps(a20, vtemp)); _mm256_storeu_ps(buf[2] + 8, _mm256_div_ps(a21, vtemp));
          _mm256_storeu_ps(buf[3], _mm256_div_ps(a30, vtemp)); _mm256_storeu_ps(buf[3] + 8, _mm256_div_ps(a31, vtemp));
          for (int r = 0; r < nr; ++r)
            for (int c = 0; c < (two ? 16 : 8); ++c) {
              int64_t j = p * 8 + c;
              if (j < nt) out[(i + r) * ldo + j] = buf[r][c];
            }
        }
      }
    }
  free(ord);
  free(qr);
  free(tp);
}

/* bf16 paths: what the bf16 / bf16x3 MFMA modes approximate.  The MFMA's
 * internal summation order is not architecturally fixed, so these are compared
 * with a tolerance (tests state it), never bit-for-bit.  Accumulate in fp64. */
void avt_oracle_sim_bf16(const uint16_t* qh, const uint16_t* ql,
                         const uint16_t* th, const uint16_t* tl, int64_t nq,
                         int64_t nt, int d, float temp, int x3, float* out,
                         int64_t ldo) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nq; ++i)
    for (int64_t j = 0; j < nt; ++j) {
      double acc = 0.0;
      for (int k = 0; k < d; ++k) {
        double ah = bf16_to_f32(qh[i * (int64_t)d + k]), bh = bf16_to_f32(th[j * (int64_t)d + k]);
        acc += ah * bh;
        if (x3) {
          double al = bf16_to_f32(ql[i * (int64_t)d + k]), bl = bf16_to_f32(tl[j * (int64_t)d + k]);
          acc += ah * bl + al * bh;
        }
      }
      out[i * ldo + j] = (float)acc / temp;
    }
}

/* ---- target order (validate.py:369-378) -------------------------------- */
/* position -> segment id for query q among n_seg segments; returns row length */
static inline int64_t target_len(int64_t q, int64_t n_seg) {
  int64_t pos = q + 1 < n_seg - 1 ? q + 1 : n_seg - 1;
  return pos == q ? n_seg : n_seg - 1;
}
static inline int64_t target_seg(int64_t q, int64_t n_seg, int64_t position) {
  int64_t pos = q + 1 < n_seg - 1 ? q + 1 : n_seg - 1;
  if (position == 0) return pos;
  int64_t lo = q < pos ? q : pos, hi = q < pos ? pos : q;
  int64_t id = position - 1;
  if (id >= lo) ++id;
  if (hi != lo && id >= hi) ++id;
  return id;
}
void avt_oracle_target_order(int64_t q, int64_t n_seg, int64_t* ids, int64_t* len) {
  int64_t L = target_len(q, n_seg);
  for (int64_t p = 0; p < L; ++p) ids[p] = target_seg(q, n_seg, p);
  *len = L;
}

/* ---- row post-process (validate.py:524-572) ---------------------------- */
void avt_oracle_row_transition(const float* sim, int64_t nq, int64_t nt,
                               int64_t ld, const int64_t* q_ids, int64_t n_seg,
                               const float* sim_a, int64_t ld_a, float alpha,
                               float threshold, int cap, int32_t* surv_idx,
                               int32_t* surv_seg, float* surv_p,
                               int32_t* surv_cnt, float* stats) {
  const float af = alpha, bf = (float)(1.0 - (double)alpha);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nq; ++r) {
    const float* x = sim + r * ld;
    const float* xa = sim_a ? sim_a + r * ld_a : NULL;
    const int64_t q = q_ids ? q_ids[r] : -1;
    const int64_t L = q_ids ? target_len(q, n_seg) : nt;
    float* p = (float*)malloc(sizeof(float) * (size_t)L);
    /* validate.py:524  output /= output.sum() */
    double s = 0.0, sa = 0.0;
    for (int64_t i = 0; i < L; ++i) {
      int64_t c = q_ids ? target_seg(q, n_seg, i) : i;
      s += (double)x[c];
      if (xa) sa += (double)xa[c];
    }
    const float sf = (float)s, saf = (float)sa;
    for (int64_t i = 0; i < L; ++i) {
      int64_t c = q_ids ? target_seg(q, n_seg, i) : i;
      float v = x[c] / sf;
      if (xa) { /* validate.py:526-527 */
        float va = xa[c] / saf;
        float m0 = af * v, m1 = bf * va;
        v = m0 + m1;
      }
      p[i] = v;
    }
    /* validate.py:531 CrossEntropyLoss(output, label 0) — reporting only */
    float mx = p[0];
    for (int64_t i = 1; i < L; ++i) mx = p[i] > mx ? p[i] : mx;
    double se = 0.0;
    for (int64_t i = 0; i < L; ++i) se += exp((double)p[i] - (double)mx);
    const float ce = (float)((double)mx + log(se) - (double)p[0]);
    /* validate.py:554 */
    const float tm = threshold * mx;
    const float cut = mx - tm;
    double s2 = 0.0;
    for (int64_t i = 0; i < L; ++i) {
      if (p[i] < cut) p[i] = 0.0f;
      s2 += (double)p[i];
    }
    const float s2f = (float)s2;
    /* validate.py:558-568 */
    int32_t cnt = 0;
    double el = 0.0;
    for (int64_t i = 0; i < L; ++i) {
      if (p[i] != 0.0f) {
        float pn = p[i] / s2f;
        if (pn != 0.0f) { /* nonzero() is evaluated after the renormalise */
          if (cnt < cap) {
            if (surv_idx) surv_idx[r * (int64_t)cap + cnt] = (int32_t)i;
            if (surv_seg) surv_seg[r * (int64_t)cap + cnt] = (int32_t)(q_ids ? target_seg(q, n_seg, i) : i);
            if (surv_p) surv_p[r * (int64_t)cap + cnt] = pn;
          }
          el += log((double)pn);
          ++cnt;
        }
      }
    }
    if (surv_cnt) surv_cnt[r] = cnt;
    if (stats) {
      stats[r * 4 + 0] = sf;
      stats[r * 4 + 1] = mx;
      stats[r * 4 + 2] = ce;
      stats[r * 4 + 3] = cnt ? (float)fabs(el / cnt) : 0.0f;
    }
    free(p);
  }
}

/* ---- top-k -------------------------------------------------------------- */
void avt_oracle_row_topk(const float* sim, int64_t nq, int64_t nt, int64_t ld,
                         const int64_t* self_col, int k, int32_t* top_idx,
                         float* top_val) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nq; ++r) {
    const float* x = sim + r * ld;
    uint8_t* used = (uint8_t*)calloc((size_t)nt, 1);
    if (self_col && self_col[r] >= 0 && self_col[r] < nt) used[self_col[r]] = 1;
    for (int s = 0; s < k; ++s) {
      int64_t best = -1;
      for (int64_t j = 0; j < nt; ++j)
        if (!used[j] && (best < 0 || x[j] > x[best])) best = j;
      if (best >= 0) used[best] = 1;
      top_idx[r * (int64_t)k + s] = (int32_t)best;
      top_val[r * (int64_t)k + s] = best >= 0 ? x[best] : -INFINITY;
    }
    free(used);
  }
}

/* ---- softmax cross-entropy (train.py:129-135) -------------------------- */
void avt_oracle_softmax_ce_fwd(const float* logits, int64_t b, int64_t c,
                               const int64_t* label, float* loss, float* prob) {
  for (int64_t r = 0; r < b; ++r) {
    const float* x = logits + r * c;
    float mx = x[0];
    for (int64_t j = 1; j < c; ++j) mx = x[j] > mx ? x[j] : mx;
    double se = 0.0;
    for (int64_t j = 0; j < c; ++j) se += exp((double)x[j] - (double)mx);
    const int64_t y = label ? label[r] : 0;
    if (loss) loss[r] = (float)((double)mx + log(se) - (double)x[y]);
    if (prob)
      for (int64_t j = 0; j < c; ++j) prob[r * c + j] = (float)(exp((double)x[j] - (double)mx) / se);
  }
}
void avt_oracle_softmax_ce_bwd(const float* prob, const int64_t* label,
                               int64_t b, int64_t c, float scale, float* dlogits) {
  for (int64_t r = 0; r < b; ++r) {
    const int64_t y = label ? label[r] : 0;
    for (int64_t j = 0; j < c; ++j)
      dlogits[r * c + j] = scale * (prob[r * c + j] - (j == y ? 1.0f : 0.0f));
  }
}
