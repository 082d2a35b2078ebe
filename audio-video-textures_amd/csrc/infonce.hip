// infonce_fwd / infonce_bwd — the training branch of the operator as two fused kernels for gfx950.
// Replaces F.normalize(q,1), F.normalize(t,2), torch.bmm and `/= temp` of the reference
// (contrastive_video_textures/models/models.py:351, 412-417, training branch :385-417) and their autograd
// backward: logits[b,j] = <q_b, t_bj> / (max(|q_b|,eps) max(|t_bj|,eps)) / temp.
//
// Latency-bound (B x (1+negs) x D = 8 x 15 x 2304 in the reference recipe): one 256-thread workgroup per batch
// row keeps q_b in registers, streams the n target rows once, and finishes all 2n+1 reductions with wave shuffles.
// The backward recomputes unit vectors from the saved inverse norms instead of storing them:
//   dq = (sum_j a_j t^_j  -  q^ sum_j a_j c_j) / |q|,   dt_j = a_j (q^ - c_j t^_j) / |t_j|,   a_j = g_j / temp,
// with c_j = <q^, t^_j> = logits_j * temp.
#include "avt_common.h"

namespace {

constexpr int kT = 256;
constexpr int kMaxPer = 64;  // elements of a row per thread kept in registers (D <= 16384)

__device__ __forceinline__ float block_sum_f(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <int PER>
__global__ __launch_bounds__(kT) void infonce_fwd_kernel(const float* __restrict__ q, const float* __restrict__ t, int n,
                                                          int d, float temp, float eps, float* __restrict__ logits,
                                                          float* __restrict__ inv_q, float* __restrict__ inv_t) {
  __shared__ float sh[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* qb = q + (int64_t)b * d;
  float qr[PER];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + k * kT;
    qr[k] = i < d ? qb[i] : 0.f;
    ss += qr[k] * qr[k];
  }
  ss = block_sum_f(ss, sh);
  const float iq = 1.0f / fmaxf(sqrtf(ss), eps);
  if (tid == 0) inv_q[b] = iq;
  for (int j = 0; j < n; ++j) {
    const float* tj = t + ((int64_t)b * n + j) * d;
    float st = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * kT;
      const float v = i < d ? tj[i] : 0.f;
      st += v * v;
      dot += v * qr[k];
    }
    st = block_sum_f(st, sh);
    dot = block_sum_f(dot, sh);
    if (tid == 0) {
      const float it = 1.0f / fmaxf(sqrtf(st), eps);
      inv_t[(int64_t)b * n + j] = it;
      logits[(int64_t)b * n + j] = dot * iq * it / temp;
    }
  }
}

template <int PER>
__global__ __launch_bounds__(kT) void infonce_bwd_kernel(const float* __restrict__ q, const float* __restrict__ t,
                                                          const float* __restrict__ logits, const float* __restrict__ g,
                                                          const float* __restrict__ inv_q, const float* __restrict__ inv_t,
                                                          int n, int d, float temp, float* __restrict__ dq,
                                                          float* __restrict__ dt) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float iq = inv_q[b];
  const float* qb = q + (int64_t)b * d;
  float qh[PER], acc[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + k * kT;
    qh[k] = i < d ? qb[i] * iq : 0.f;
    acc[k] = 0.f;
  }
  float sac = 0.f;  // sum_j a_j c_j
  for (int j = 0; j < n; ++j) {
    const int64_t bj = (int64_t)b * n + j;
    const float a = g[bj] / temp, c = logits[bj] * temp, it = inv_t[bj];
    sac += a * c;
    const float* tj = t + bj * d;
    float* dtj = dt + bj * d;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * kT;
      if (i < d) {
        const float th = tj[i] * it;
        acc[k] += a * th;
        dtj[i] = a * (qh[k] - c * th) * it;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + k * kT;
    if (i < d) dq[(int64_t)b * d + i] = (acc[k] - qh[k] * sac) * iq;
  }
}

template <typename F>
int dispatch_per(int d, F&& f) {
  const int per = (d + kT - 1) / kT;
  if (per <= 4) return f(std::integral_constant<int, 4>{});
  if (per <= 16) return f(std::integral_constant<int, 16>{});
  if (per <= kMaxPer) return f(std::integral_constant<int, kMaxPer>{});
  avt::set_error("infonce: d=%d exceeds the register-resident limit %d", d, kMaxPer * kT);
  return AVT_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int avt_infonce_fwd(const float* q, const float* t, int64_t b, int n, int d, float temp, float eps,
                               float* logits, float* inv_q, float* inv_t, void* stream) {
  AVT_REQUIRE(b >= 0 && n > 0 && d > 0 && temp != 0.0f, "avt_infonce_fwd: bad sizes");
  if (b == 0) return AVT_OK;
  AVT_REQUIRE(q && t && logits && inv_q && inv_t, "avt_infonce_fwd: NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rc = dispatch_per(d, [&](auto per) {
    hipLaunchKernelGGL((infonce_fwd_kernel<decltype(per)::value>), dim3((unsigned)b), dim3(kT), 0, st, q, t, n, d, temp, eps,
                       logits, inv_q, inv_t);
    return AVT_OK;
  });
  return rc ? rc : avt::check_launch("avt_infonce_fwd");
}

extern "C" int avt_infonce_bwd(const float* q, const float* t, const float* logits, const float* dlogits,
                               const float* inv_q, const float* inv_t, int64_t b, int n, int d, float temp, float* dq,
                               float* dt, void* stream) {
  AVT_REQUIRE(b >= 0 && n > 0 && d > 0 && temp != 0.0f, "avt_infonce_bwd: bad sizes");
  if (b == 0) return AVT_OK;
  AVT_REQUIRE(q && t && logits && dlogits && inv_q && inv_t && dq && dt, "avt_infonce_bwd: NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rc = dispatch_per(d, [&](auto per) {
    hipLaunchKernelGGL((infonce_bwd_kernel<decltype(per)::value>), dim3((unsigned)b), dim3(kT), 0, st, q, t, logits, dlogits,
                       inv_q, inv_t, n, d, temp, dq, dt);
    return AVT_OK;
  });
  return rc ? rc : avt::check_launch("avt_infonce_bwd");
}
