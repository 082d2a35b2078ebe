// softmax_ce — InfoNCE cross-entropy over the [B, 1+negs] logits for gfx950.
// Replaces nn.CrossEntropyLoss of the reference training step
// (contrastive_video_textures/train.py:129-135; positives at column 0) and its
// autograd backward.  Latency-bound (B*C*4 bytes); one 64-lane wave per row,
// DPP-shuffle max/sum, fp64 exp like the oracle so losses agree to rounding.
#include <math.h>

#include "avt_common.h"

namespace {

__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float* __restrict__ logits, int64_t b, int64_t c,
                                                              const int64_t* __restrict__ label,
                                                              float* __restrict__ loss, float* __restrict__ prob) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= b) return;
  const float* x = logits + r * c;
  float mx = -INFINITY;
  for (int64_t j = lane; j < c; j += 64) mx = fmaxf(mx, x[j]);
  mx = avt::wave_max(mx);
  double se = 0.0;
  for (int64_t j = lane; j < c; j += 64) se += exp((double)x[j] - (double)mx);
  se = avt::wave_sum(se);
  const int64_t y = label ? label[r] : 0;
  if (loss && lane == 0) loss[r] = (float)((double)mx + log(se) - (double)x[y]);
  if (prob)
    for (int64_t j = lane; j < c; j += 64) prob[r * c + j] = (float)(exp((double)x[j] - (double)mx) / se);
}

__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const float* __restrict__ prob,
                                                              const int64_t* __restrict__ label, int64_t b, int64_t c,
                                                              float scale, float* __restrict__ dlogits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= b * c) return;
  const int64_t r = i / c, j = i - r * c;
  const int64_t y = label ? label[r] : 0;
  dlogits[i] = __fmul_rn(scale, __fsub_rn(prob[i], j == y ? 1.0f : 0.0f));
}

}  // namespace

extern "C" int avt_softmax_ce_fwd(const float* logits, int64_t b, int64_t c, const int64_t* label, float* loss,
                                  float* prob, void* stream) {
  AVT_REQUIRE(b >= 0 && c > 0, "avt_softmax_ce_fwd: bad sizes");
  if (b == 0) return AVT_OK;
  AVT_REQUIRE(logits && (loss || prob), "avt_softmax_ce_fwd: NULL pointer");
  hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3((unsigned)((b + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), logits, b, c, label, loss, prob);
  return avt::check_launch("avt_softmax_ce_fwd");
}

extern "C" int avt_softmax_ce_bwd(const float* prob, const int64_t* label, int64_t b, int64_t c, float scale,
                                  float* dlogits, void* stream) {
  AVT_REQUIRE(b >= 0 && c > 0, "avt_softmax_ce_bwd: bad sizes");
  if (b == 0) return AVT_OK;
  AVT_REQUIRE(prob && dlogits, "avt_softmax_ce_bwd: NULL pointer");
  hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3((unsigned)((b * c + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), prob, label, b, c, scale, dlogits);
  return avt::check_launch("avt_softmax_ce_bwd");
}
