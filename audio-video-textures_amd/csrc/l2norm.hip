// l2norm_rows — row-wise y = [x0|x1] / max(||[x0|x1]||, eps) for gfx950.
// Replaces torch.cat + F.normalize of the reference operator
// (contrastive_video_textures/models/models.py:347-351, 408-412, 433-436).
//
// HBM-bound: one 64-lane wave owns one row, reads it once with 16-byte loads,
// keeps it in registers (rows up to 3072 elements; longer rows are re-read,
// from L2), reduces the sum of squares in fp64 by DPP shuffles, and writes the
// fp32 row plus the bf16 hi/lo split the MFMA similarity kernels consume.
// Algorithmic bytes per row: d*4 read + d*(4 [+2 +2]) written.
#include "avt_common.h"

namespace {

constexpr int kWaves = 4;          // rows per 256-thread block
constexpr int kCache = 12;         // float4 per lane kept in registers (3072 elements/row)

struct Src {
  const float* x0;
  const float* x1;
  int d0, d1;
};

__device__ __forceinline__ float4 load4(const Src& s, int64_t row, int k) {
  // k is a multiple of 4 and d0 % 4 == 0, so a float4 never straddles x0|x1
  if (k < s.d0) return *reinterpret_cast<const float4*>(s.x0 + row * s.d0 + k);
  return *reinterpret_cast<const float4*>(s.x1 + row * s.d1 + (k - s.d0));
}
__device__ __forceinline__ float load1(const Src& s, int64_t row, int k) {
  return k < s.d0 ? s.x0[row * s.d0 + k] : s.x1[row * s.d1 + (k - s.d0)];
}

__device__ __forceinline__ void store_split(float v, uint16_t& hi, uint16_t& lo) {
  const uint32_t h = avt::pack_bf16x2(v, 0.0f);  // hardware RNE, same rounding as the oracle's software one
  hi = (uint16_t)(h & 0xffffu);
  lo = (uint16_t)(avt::pack_bf16x2(v - avt::bf16x2_lo(h), 0.0f) & 0xffffu);
}

template <bool CACHED>
__global__ __launch_bounds__(256) void l2norm_vec4(Src s, int64_t n, float eps, float* __restrict__ y,
                                                    uint16_t* __restrict__ yh, uint16_t* __restrict__ yl) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (row >= n) return;
  const int d = s.d0 + s.d1;
  const int nv = d >> 2;  // float4 per row
  float4 c[kCache];
  double ss = 0.0;
  if (CACHED) {
#pragma unroll
    for (int u = 0; u < kCache; ++u) {
      const int v = lane + u * 64;
      c[u] = v < nv ? load4(s, row, v * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kCache; ++u)
      ss += (double)c[u].x * c[u].x + (double)c[u].y * c[u].y + (double)c[u].z * c[u].z + (double)c[u].w * c[u].w;
  } else {
    for (int v = lane; v < nv; v += 64) {
      const float4 a = load4(s, row, v * 4);
      ss += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
    }
  }
  ss = avt::wave_sum(ss);
  const float nrm = __builtin_sqrtf((float)ss);
  const float den = nrm > eps ? nrm : eps;
  auto emit = [&](int v, float4 a) {
    float4 o;
    o.x = __fdiv_rn(a.x, den);
    o.y = __fdiv_rn(a.y, den);
    o.z = __fdiv_rn(a.z, den);
    o.w = __fdiv_rn(a.w, den);
    const int64_t off = row * d + (int64_t)v * 4;
    if (y) *reinterpret_cast<float4*>(y + off) = o;
    if (yh || yl) {
      ushort4 h, l;
      store_split(o.x, h.x, l.x);
      store_split(o.y, h.y, l.y);
      store_split(o.z, h.z, l.z);
      store_split(o.w, h.w, l.w);
      if (yh) *reinterpret_cast<ushort4*>(yh + off) = h;
      if (yl) *reinterpret_cast<ushort4*>(yl + off) = l;
    }
  };
  if (CACHED) {
#pragma unroll
    for (int u = 0; u < kCache; ++u) {
      const int v = lane + u * 64;
      if (v < nv) emit(v, c[u]);
    }
  } else {
    for (int v = lane; v < nv; v += 64) emit(v, load4(s, row, v * 4));
  }
}

// any d0/d1, any alignment
__global__ __launch_bounds__(256) void l2norm_scalar(Src s, int64_t n, float eps, float* __restrict__ y,
                                                      uint16_t* __restrict__ yh, uint16_t* __restrict__ yl) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (row >= n) return;
  const int d = s.d0 + s.d1;
  double ss = 0.0;
  for (int k = lane; k < d; k += 64) {
    const float a = load1(s, row, k);
    ss += (double)a * a;
  }
  ss = avt::wave_sum(ss);
  const float nrm = __builtin_sqrtf((float)ss);
  const float den = nrm > eps ? nrm : eps;
  for (int k = lane; k < d; k += 64) {
    const float o = __fdiv_rn(load1(s, row, k), den);
    const int64_t off = row * d + k;
    if (y) y[off] = o;
    if (yh || yl) {
      uint16_t h, l;
      store_split(o, h, l);
      if (yh) yh[off] = h;
      if (yl) yl[off] = l;
    }
  }
}

}  // namespace

extern "C" int avt_l2norm_rows(const float* x0, int d0, const float* x1, int d1, int64_t n, float eps, float* y_f32,
                               void* y_hi, void* y_lo, void* stream) {
  AVT_REQUIRE(n >= 0 && d0 > 0 && d1 >= 0, "avt_l2norm_rows: bad sizes n=%lld d0=%d d1=%d", (long long)n, d0, d1);
  if (n == 0) return AVT_OK;  // empty table: nothing to do (pointers may be NULL)
  AVT_REQUIRE(x0, "avt_l2norm_rows: x0 is NULL");
  AVT_REQUIRE((x1 != nullptr) == (d1 > 0), "avt_l2norm_rows: x1/d1 mismatch");
  AVT_REQUIRE(y_f32 || y_hi || y_lo, "avt_l2norm_rows: no output requested");
  Src s{x0, x1, d0, d1};
  const int d = d0 + d1;
  const bool vec = (d0 % 4 == 0) && (d1 % 4 == 0) && avt::aligned16(x0) && (!x1 || avt::aligned16(x1)) &&
                   (!y_f32 || avt::aligned16(y_f32)) && (!y_hi || avt::aligned16(y_hi)) &&
                   (!y_lo || avt::aligned16(y_lo));
  const dim3 grid((unsigned)((n + kWaves - 1) / kWaves)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint16_t* yh = static_cast<uint16_t*>(y_hi);
  uint16_t* yl = static_cast<uint16_t*>(y_lo);
  if (vec && d <= kCache * 64 * 4)
    hipLaunchKernelGGL(l2norm_vec4<true>, grid, block, 0, st, s, n, eps, y_f32, yh, yl);
  else if (vec)
    hipLaunchKernelGGL(l2norm_vec4<false>, grid, block, 0, st, s, n, eps, y_f32, yh, yl);
  else
    hipLaunchKernelGGL(l2norm_scalar, grid, block, 0, st, s, n, eps, y_f32, yh, yl);
  return avt::check_launch("avt_l2norm_rows");
}
