// classic — the two remaining matrix steps of the classic (Schoedl-style) video-texture baseline, BASELINE config 1
// (SURVEY.md §8f-4), after pairwise.hip's D1:
//   diag_filter  D2[i, j] = sum_k w[k] * D1[i + k, j + k]   the diagonal binomial filter the reference applies as a conv2d
//                with a diagonal [fs, fs] kernel (baselines/classic_video_textures/computeD2.py:21-52): fs taps, not fs^2.
//   q_learning   the future-cost iteration (q_learning.py:27-68): sweep until mean((new - old)^2) <= tol:
//                    mins[j] = min_{k != j} old[j, k];   new[i, :] = D3[i, :] + alpha * mins   (rows n-1 .. 1; row 0 stays)
//                ~700 sweeps at alpha = 0.997; the reference pays two tensor copies and a masked gather per ROW per sweep.
//                Here ONE workgroup keeps the whole matrix in LDS (n <= 200: 160 KB) and the base matrix D3 in registers and
//                runs every sweep without leaving the kernel: wave-per-row minima, barrier, update + fp64 residual, barrier.
//                Same fp32 arithmetic as the reference (one rounded multiply, one rounded add per element).
#include "avt_common.h"

namespace {

__global__ __launch_bounds__(256) void diag_filter_kernel(const float* __restrict__ d1, int n, const float* __restrict__ w, int fs,
                                                          float* __restrict__ out, int m) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= m * m) return;
  const int i = idx / m, j = idx - i * m;
  float s = 0.0f;
  for (int k = 0; k < fs; ++k) s = __fmaf_rn(w[k], d1[(int64_t)(i + k) * n + j + k], s);
  out[idx] = s;
}

constexpr int QT = 1024, QE = 40;  // threads, matrix elements per thread (n * n <= QT * QE = 40960)

__global__ __launch_bounds__(QT) void q_learning_kernel(const float* __restrict__ d3, int n, float alpha, float tol, int max_iter,
                                                        float* __restrict__ out, int* __restrict__ iters) {
  extern __shared__ float cur[];  // [n * n] the running matrix
  __shared__ float mins[256];
  __shared__ double red[QT / 64];
  __shared__ int stop;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nn = n * n;
  float base[QE];
#pragma unroll
  for (int k = 0; k < QE; ++k) {
    const int e = tid + k * QT;
    base[k] = e < nn ? d3[e] : 0.0f;
    if (e < nn) cur[e] = base[k];
  }
  if (tid == 0) stop = 0;
  __syncthreads();
  int it = 0;
  while (true) {
    // off-diagonal row minima of the matrix as it stands (the reference's D3_old)
    for (int j = wid; j < n; j += QT / 64) {
      float m = INFINITY;
      for (int k = lane; k < n; k += 64)
        if (k != j) m = fminf(m, cur[j * n + k]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
      if (lane == 0) mins[j] = m;
    }
    __syncthreads();
    double sq = 0.0;
#pragma unroll
    for (int k = 0; k < QE; ++k) {
      const int e = tid + k * QT;
      if (e < nn && e >= n) {  // rows 1 .. n-1
        const int j = e % n;
        const float v = __fadd_rn(base[k], __fmul_rn(alpha, mins[j]));
        const float d = __fsub_rn(v, cur[e]);
        sq += (double)__fmul_rn(d, d);
        cur[e] = v;
      }
    }
    sq = avt::wave_sum(sq);
    if (lane == 0) red[wid] = sq;
    __syncthreads();
    ++it;
    if (tid == 0) {
      double s = 0.0;
      for (int w = 0; w < QT / 64; ++w) s += red[w];
      const float eps = (float)(s / (double)nn);
      stop = (!(eps > tol) || it >= max_iter) ? 1 : 0;
    }
    __syncthreads();
    if (stop) break;
  }
#pragma unroll
  for (int k = 0; k < QE; ++k) {
    const int e = tid + k * QT;
    if (e < nn) out[e] = cur[e];
  }
  if (tid == 0 && iters) *iters = it;
}

}  // namespace

extern "C" int avt_diag_filter_f32(const float* d1, int n, const float* w, int fs, float* out, void* stream) {
  AVT_REQUIRE(d1 && w && out, "avt_diag_filter_f32: NULL pointer");
  AVT_REQUIRE(n > 0 && fs > 0 && fs <= n, "avt_diag_filter_f32: need 0 < fs <= n");
  const int m = n - fs + 1;
  AVT_REQUIRE((int64_t)m * m < (1ll << 31), "avt_diag_filter_f32: matrix too large");
  hipLaunchKernelGGL(diag_filter_kernel, dim3((unsigned)((m * m + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     d1, n, w, fs, out, m);
  return avt::check_launch("avt_diag_filter_f32");
}

extern "C" int avt_q_learning_supported(int n) { return (n >= 2 && n <= 200) ? 1 : 0; }

extern "C" int avt_q_learning_f32(const float* d3, int n, float alpha, float tol, int max_iter, float* out, int* iters,
                                  void* stream) {
  AVT_REQUIRE(d3 && out, "avt_q_learning_f32: NULL pointer");
  AVT_REQUIRE(avt_q_learning_supported(n), "avt_q_learning_f32: the matrix must be 2..200 rows square (LDS-resident), got %d", n);
  AVT_REQUIRE(max_iter > 0, "avt_q_learning_f32: max_iter must be positive");
  const int lds_bytes = n * n * 4;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(q_learning_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 200 * 200 * 4);
  if (e != hipSuccess) {
    avt::set_error("avt_q_learning_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(q_learning_kernel, dim3(1), dim3(QT), lds_bytes, static_cast<hipStream_t>(stream), d3, n, alpha, tol, max_iter,
                     out, iters);
  return avt::check_launch("avt_q_learning_f32");
}
