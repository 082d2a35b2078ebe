// pairwise_l2 — D1[i, j] = || f_i - f_j ||_2 over flattened frames, the distance matrix of the classic (Schoedl-style)
// video-texture baseline (baselines/classic_video_textures/computeD1.py:47-96: the reference materialises
// [bs, bs, H*W*C] difference tensors tile by tile on the GPU / CPU).  BASELINE config 1: 200 frames of 128x128x3.
//
// Canonical rounding: the squared differences are exact in fp64 for fp32 inputs; each thread
// accumulates a strided subset in fp64, the 256 partial sums are reduced in a fixed tree, one sqrt in fp64, one
// rounding to fp32 — independent of tiling and launch shape, and within 1 ulp (fp32) of any correctly rounded
// evaluation.  HBM/L2-bound: a workgroup owns row i and 8 columns, streams x_i once against the 8 x_j rows
// (9 row reads per 8 distances).
#include "avt_common.h"

namespace {

constexpr int kT = 256, kCols = 8;

__global__ __launch_bounds__(kT) void pairwise_l2_kernel(const float* __restrict__ x, int n, int64_t d, float* __restrict__ out) {
  __shared__ double red[kCols][kT / 64];
  const int i = blockIdx.y, j0 = blockIdx.x * kCols;
  const float* xi = x + (int64_t)i * d;
  double acc[kCols];
#pragma unroll
  for (int c = 0; c < kCols; ++c) acc[c] = 0.0;
  for (int64_t k = threadIdx.x; k < d; k += kT) {
    const double a = (double)xi[k];
#pragma unroll
    for (int c = 0; c < kCols; ++c) {
      if (j0 + c < n) {
        const double df = a - (double)x[(int64_t)(j0 + c) * d + k];
        acc[c] = fma(df, df, acc[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < kCols; ++c) {
    const double s = avt::wave_sum(acc[c]);
    if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < kCols && j0 + (int)threadIdx.x < n) {
    const double s = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    out[(int64_t)i * n + j0 + threadIdx.x] = (float)sqrt(s);
  }
}

}  // namespace

extern "C" int avt_pairwise_l2_f32(const float* x, int n, int64_t d, float* out, void* stream) {
  AVT_REQUIRE(n >= 0 && d > 0, "avt_pairwise_l2_f32: bad sizes");
  if (n == 0) return AVT_OK;
  AVT_REQUIRE(x && out, "avt_pairwise_l2_f32: NULL pointer");
  AVT_REQUIRE(n <= 65535, "avt_pairwise_l2_f32: at most 65535 rows");
  hipLaunchKernelGGL(pairwise_l2_kernel, dim3((unsigned)((n + kCols - 1) / kCols), (unsigned)n), dim3(kT), 0,
                     static_cast<hipStream_t>(stream), x, n, d, out);
  return avt::check_launch("avt_pairwise_l2_f32");
}
