// maxpool_hw3s2 — MaxPool3d((1,3,3), stride (1,2,2), padding (0,1,1)) on NDHWC bf16 rows, for the SlowFast
// stems (the reference reaches it through the third-party SlowFast model,
// contrastive_video_textures/models/models.py:335, 399).
// maxpool_hw2s2 — MaxPool2d(2, stride 2), floor mode, of VGGish
// (contrastive_video_textures/models/audio_models/vggish.py:15-33) on NHWC bf16 rows.  HBM-bound: each thread owns 8 channels (16 bytes) of
// one output position, reads its <= 9 taps with 16-byte loads and writes one 16-byte chunk — optionally into a
// channel slice of a wider row buffer (ldo), which is how the slow stem lands in the lateral-fusion concat.
#include "avt_common.h"
#include "split_planes.h"

namespace {

// bf16 pairs -> order-preserving signed 16-bit keys (negative values get their magnitude bits flipped; an involution),
// so the 3x3 max is v_pk_max_i16 on two channels at once instead of unpack / compare / select per channel: the kernel
// was VALU-bound (~700 VALU instructions per 16-byte output with the 64-bit index divisions), not HBM-bound.
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t order_key(uint32_t x) {
  uint32_t m = (x >> 15) & 0x00010001u;
  m = (m << 15) - m;  // 0x7fff in every negative half
  return x ^ m;
}
__device__ __forceinline__ uint32_t kmax(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}

// Flat 32-bit index over (frame, ho, wo, 8-channel chunk): three 32-bit divisions instead of three 64-bit ones.
template <int KS, int PAD>
__global__ __launch_bounds__(256) void maxpool_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                       int bt, int H, int W, int C, int ldi, int ldo, int Ho, int Wo, int tgroup) {
  const unsigned cpr = (unsigned)C >> 3;
  const unsigned cg = (unsigned)(C / tgroup);
  const unsigned total = (unsigned)bt * Ho * Wo * cpr;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned p = i / cpr;
    const unsigned cc = i - p * cpr;
    const unsigned q = p / (unsigned)Wo;
    const int wo = (int)(p - q * Wo);
    const unsigned b = q / (unsigned)Ho;
    const int ho = (int)(q - b * Ho);
    const uint16_t* frame = in + (int64_t)b * H * W * ldi + cc * 8;
    // taps outside the image are clamped onto the nearest valid tap (a duplicate does not change a max): nine
    // unconditional 16-byte loads in flight instead of a branch per tap
    uint4 v[KS * KS];
#pragma unroll
    for (int dh = 0; dh < KS; ++dh) {
      int hi = 2 * ho - PAD + dh;
      hi = hi < 0 ? 0 : (hi > H - 1 ? H - 1 : hi);
#pragma unroll
      for (int dw = 0; dw < KS; ++dw) {
        int wi = 2 * wo - PAD + dw;
        wi = wi < 0 ? 0 : (wi > W - 1 ? W - 1 : wi);
        v[dh * KS + dw] = *reinterpret_cast<const uint4*>(frame + (int64_t)(hi * W + wi) * ldi);
      }
    }
    uint4 m = {order_key(v[0].x), order_key(v[0].y), order_key(v[0].z), order_key(v[0].w)};
#pragma unroll
    for (int k = 1; k < KS * KS; ++k) {
      m.x = kmax(m.x, order_key(v[k].x));
      m.y = kmax(m.y, order_key(v[k].y));
      m.z = kmax(m.z, order_key(v[k].z));
      m.w = kmax(m.w, order_key(v[k].w));
    }
    m.x = order_key(m.x);
    m.y = order_key(m.y);
    m.z = order_key(m.z);
    m.w = order_key(m.w);
    // tgroup > 1: the C channels are `tgroup` consecutive frames of C/tgroup channels each (the time-grouped stem):
    // un-group while writing, so the result is plain NDHWC with bt*tgroup frames
    const unsigned j = (cc * 8) / cg, c0 = (cc * 8) - j * cg;
    *reinterpret_cast<uint4*>(out + ((((int64_t)b * tgroup + j) * Ho + ho) * Wo + wo) * ldo + c0) = m;
  }
}

}  // namespace

extern "C" int avt_maxpool_hw3s2_ndhwc_bf16(const void* in, void* out, int bt, int h, int w, int c, int ldi, int ldo,
                                            int tgroup, void* stream) {
  AVT_REQUIRE(in && out, "avt_maxpool_hw3s2_ndhwc_bf16: NULL pointer");
  AVT_REQUIRE(bt > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0 && ldi >= c,
              "avt_maxpool_hw3s2_ndhwc_bf16: channels / leading dimensions must be multiples of 8");
  AVT_REQUIRE(avt::aligned16(in) && avt::aligned16(out), "avt_maxpool_hw3s2_ndhwc_bf16: pointers must be 16-byte aligned");
  AVT_REQUIRE(tgroup >= 1 && c % tgroup == 0 && (c / tgroup) % 8 == 0 && ldo >= c / tgroup,
              "avt_maxpool_hw3s2_ndhwc_bf16: tgroup must split the channels into multiples of 8");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)bt * ho * wo * (c / 8);
  AVT_REQUIRE(total < (1ll << 31), "avt_maxpool_hw3s2_ndhwc_bf16: more than 2^31 output chunks");
  const int64_t blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
  hipLaunchKernelGGL((maxpool_kernel<3, 1>), dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint16_t*>(in), static_cast<uint16_t*>(out), bt, h, w, c, ldi, ldo, ho, wo, tgroup);
  return avt::check_launch("avt_maxpool_hw3s2_ndhwc_bf16");
}

// mean over the positions of a clip: the global average pool of the SlowFast head (AdaptiveAvgPool3d(1) per pathway
// before the concat, models/models.py:576-580 head surgery) on NDHWC bf16 rows -> fp32, written straight into a
// column slice of the [B, 2304] embedding table.  HBM-bound; deterministic: a workgroup owns (clip, 64 channels),
// 32 row groups accumulate in fp32 and are reduced through LDS in a fixed order.
__global__ __launch_bounds__(256) void mean_positions_kernel(const uint16_t* __restrict__ in, int P, int C, int ldi,
                                                             float* __restrict__ out, int ldo) {
  __shared__ float red[32][65];
  const int b = blockIdx.x, c0 = blockIdx.y * 64;
  const int cc = threadIdx.x & 7, rg = threadIdx.x >> 3;  // 8-channel chunk, row group
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int ch = c0 + cc * 8;
  if (ch < C) {
    const uint16_t* base = in + (int64_t)b * P * ldi + ch;
    for (int r = rg; r < P; r += 32) {
      const uint4 v = *reinterpret_cast<const uint4*>(base + (int64_t)r * ldi);
      acc[0] += avt::bf16x2_lo(v.x);
      acc[1] += avt::bf16x2_hi(v.x);
      acc[2] += avt::bf16x2_lo(v.y);
      acc[3] += avt::bf16x2_hi(v.y);
      acc[4] += avt::bf16x2_lo(v.z);
      acc[5] += avt::bf16x2_hi(v.z);
      acc[6] += avt::bf16x2_lo(v.w);
      acc[7] += avt::bf16x2_hi(v.w);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rg][cc * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
    float s = 0.f;
    for (int g = 0; g < 32; ++g) s += red[g][threadIdx.x];
    out[(int64_t)b * ldo + c0 + threadIdx.x] = s / (float)P;
  }
}

extern "C" int avt_mean_positions_bf16(const void* in, int batch, int p, int c, int ldi, float* out, int ldo, void* stream) {
  AVT_REQUIRE(in && out, "avt_mean_positions_bf16: NULL pointer");
  AVT_REQUIRE(batch > 0 && p > 0 && c > 0 && c % 8 == 0 && ldi % 8 == 0 && ldi >= c && ldo >= c,
              "avt_mean_positions_bf16: channels / input stride must be multiples of 8, strides must cover the channels");
  AVT_REQUIRE(avt::aligned16(in), "avt_mean_positions_bf16: input must be 16-byte aligned");
  hipLaunchKernelGGL(mean_positions_kernel, dim3((unsigned)batch, (unsigned)((c + 63) / 64)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint16_t*>(in), p, c, ldi, out, ldo);
  return avt::check_launch("avt_mean_positions_bf16");
}

extern "C" int avt_maxpool_hw2s2_ndhwc_bf16(const void* in, void* out, int bt, int h, int w, int c, int ldi, int ldo,
                                            void* stream) {
  AVT_REQUIRE(in && out, "avt_maxpool_hw2s2_ndhwc_bf16: NULL pointer");
  AVT_REQUIRE(bt > 0 && h >= 2 && w >= 2 && c > 0 && c % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0 && ldi >= c && ldo >= c,
              "avt_maxpool_hw2s2_ndhwc_bf16: h, w >= 2; channels / leading dimensions must be multiples of 8");
  AVT_REQUIRE(avt::aligned16(in) && avt::aligned16(out), "avt_maxpool_hw2s2_ndhwc_bf16: pointers must be 16-byte aligned");
  const int ho = h / 2, wo = w / 2;  // floor mode: an odd last row / column is dropped
  const int64_t total = (int64_t)bt * ho * wo * (c / 8);
  AVT_REQUIRE(total < (1ll << 31), "avt_maxpool_hw2s2_ndhwc_bf16: more than 2^31 output chunks");
  const int64_t blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
  hipLaunchKernelGGL((maxpool_kernel<2, 0>), dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint16_t*>(in), static_cast<uint16_t*>(out), bt, h, w, c, ldi, ldo, ho, wo, 1);
  return avt::check_launch("avt_maxpool_hw2s2_ndhwc_bf16");
}

// ---- split-plane ("x3") forms for the contract-grade encoder: every tensor is a (hi, lo) pair of 16-bit planes ------
// The max is taken on the fp32 values hi + lo and split again (value-preserving: hi + lo is exact in fp32); the mean
// sums them in fp32 in the same fixed order as the bf16 kernel.
namespace {

// KS = 3, PAD = 1: MaxPool3d((1,3,3),(1,2,2),(0,1,1)), the SlowFast stems;  KS = 2, PAD = 0: MaxPool2d(2, 2) floor mode, VGGish
template <bool F16, int KS = 3, int PAD = 1>
__global__ __launch_bounds__(256) void maxpool3_x3_kernel(const uint16_t* __restrict__ in_hi, const uint16_t* __restrict__ in_lo,
                                                           uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo, int bt,
                                                           int H, int W, int C, int ldi, int ldo, int Ho, int Wo, int tgroup,
                                                           const int32_t* __restrict__ fidx = nullptr) {
  // fidx (round 4): output frame b pools INPUT frame fidx[b] — the slow stem runs once per distinct source frame and the pool
  // hands every (window, slot) its frame (overlapping windows share about half of their slow-pathway frames)
  const unsigned cpr = (unsigned)C >> 3;
  const unsigned cg = (unsigned)(C / tgroup);
  const unsigned total = (unsigned)bt * Ho * Wo * cpr;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned p = i / cpr;
    const unsigned cc = i - p * cpr;
    const unsigned q = p / (unsigned)Wo;
    const int wo = (int)(p - q * Wo);
    const unsigned b = q / (unsigned)Ho;
    const int ho = (int)(q - b * Ho);
    const int64_t frame = (int64_t)(fidx ? (unsigned)fidx[b] : b) * H * W * ldi + cc * 8;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int dh = 0; dh < KS; ++dh) {
      int hi = 2 * ho - PAD + dh;
      hi = hi < 0 ? 0 : (hi > H - 1 ? H - 1 : hi);  // clamped taps duplicate a valid one: a max does not care
#pragma unroll
      for (int dw = 0; dw < KS; ++dw) {
        int wi = 2 * wo - PAD + dw;
        wi = wi < 0 ? 0 : (wi > W - 1 ? W - 1 : wi);
        const int64_t o = frame + (int64_t)(hi * W + wi) * ldi;
        const uint4 vh = *reinterpret_cast<const uint4*>(in_hi + o);
        const uint4 vl = *reinterpret_cast<const uint4*>(in_lo + o);
        const uint32_t* ph = reinterpret_cast<const uint32_t*>(&vh);
        const uint32_t* pl = reinterpret_cast<const uint32_t*>(&vl);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const avt::f32x2 v = avt::join2<F16>(ph[e], pl[e]);
          // (x > m ? x : m keeps a NaN of x; fmaxf would drop it — a poisoned activation must reach the embedding)
          m[2 * e] = (v.x > m[2 * e] || v.x != v.x) ? v.x : m[2 * e];
          m[2 * e + 1] = (v.y > m[2 * e + 1] || v.y != v.y) ? v.y : m[2 * e + 1];
        }
      }
    }
    uint4 oh, ol;
    avt::split8<F16>(m, oh, ol);
    const unsigned j = (cc * 8) / cg, c0 = (cc * 8) - j * cg;
    const int64_t o = ((((int64_t)b * tgroup + j) * Ho + ho) * Wo + wo) * ldo + c0;
    *reinterpret_cast<uint4*>(out_hi + o) = oh;
    *reinterpret_cast<uint4*>(out_lo + o) = ol;
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void mean_positions_x3_kernel(const uint16_t* __restrict__ in_hi,
                                                                 const uint16_t* __restrict__ in_lo, int P, int C, int ldi,
                                                                 float* __restrict__ out, int ldo) {
  __shared__ float red[32][65];
  const int b = blockIdx.x, c0 = blockIdx.y * 64;
  const int cc = threadIdx.x & 7, rg = threadIdx.x >> 3;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int ch = c0 + cc * 8;
  if (ch < C) {
    const int64_t base = (int64_t)b * P * ldi + ch;
    for (int r = rg; r < P; r += 32) {
      const uint4 vh = *reinterpret_cast<const uint4*>(in_hi + base + (int64_t)r * ldi);
      const uint4 vl = *reinterpret_cast<const uint4*>(in_lo + base + (int64_t)r * ldi);
      const uint32_t* ph = reinterpret_cast<const uint32_t*>(&vh);
      const uint32_t* pl = reinterpret_cast<const uint32_t*>(&vl);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const avt::f32x2 v = avt::join2<F16>(ph[e], pl[e]);
        acc[2 * e] += v.x;
        acc[2 * e + 1] += v.y;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rg][cc * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
    float s = 0.f;
    for (int g = 0; g < 32; ++g) s += red[g][threadIdx.x];
    out[(int64_t)b * ldo + c0 + threadIdx.x] = s / (float)P;
  }
}

}  // namespace

extern "C" int avt_maxpool_hw3s2_ndhwc_x3(const void* in_hi, const void* in_lo, void* out_hi, void* out_lo, int bt, int h,
                                          int w, int c, int ldi, int ldo, int tgroup, int plane_dtype, const int32_t* frame_idx,
                                          void* stream) {
  AVT_REQUIRE(in_hi && in_lo && out_hi && out_lo, "avt_maxpool_hw3s2_ndhwc_x3: NULL pointer");
  AVT_REQUIRE(!frame_idx || tgroup == 1, "avt_maxpool_hw3s2_ndhwc_x3: frame_idx goes with tgroup == 1");
  AVT_REQUIRE(bt > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0 && ldi >= c,
              "avt_maxpool_hw3s2_ndhwc_x3: channels / leading dimensions must be multiples of 8");
  AVT_REQUIRE(avt::aligned16(in_hi) && avt::aligned16(in_lo) && avt::aligned16(out_hi) && avt::aligned16(out_lo),
              "avt_maxpool_hw3s2_ndhwc_x3: pointers must be 16-byte aligned");
  AVT_REQUIRE(tgroup >= 1 && c % tgroup == 0 && (c / tgroup) % 8 == 0 && ldo >= c / tgroup,
              "avt_maxpool_hw3s2_ndhwc_x3: tgroup must split the channels into multiples of 8");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_maxpool_hw3s2_ndhwc_x3: bad plane_dtype");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)bt * ho * wo * (c / 8);
  AVT_REQUIRE(total < (1ll << 31), "avt_maxpool_hw3s2_ndhwc_x3: more than 2^31 output chunks");
  const int64_t blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
  auto ih = static_cast<const uint16_t*>(in_hi), il = static_cast<const uint16_t*>(in_lo);
  auto oh = static_cast<uint16_t*>(out_hi), ol = static_cast<uint16_t*>(out_lo);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (plane_dtype == AVT_X3_F16)
    hipLaunchKernelGGL((maxpool3_x3_kernel<true>), dim3(grid), dim3(256), 0, st, ih, il, oh, ol, bt, h, w, c, ldi, ldo, ho, wo, tgroup, frame_idx);
  else
    hipLaunchKernelGGL((maxpool3_x3_kernel<false>), dim3(grid), dim3(256), 0, st, ih, il, oh, ol, bt, h, w, c, ldi, ldo, ho, wo, tgroup, frame_idx);
  return avt::check_launch("avt_maxpool_hw3s2_ndhwc_x3");
}

extern "C" int avt_maxpool_hw2s2_ndhwc_x3(const void* in_hi, const void* in_lo, void* out_hi, void* out_lo, int bt, int h, int w,
                                          int c, int ldi, int ldo, int plane_dtype, void* stream) {
  AVT_REQUIRE(in_hi && in_lo && out_hi && out_lo, "avt_maxpool_hw2s2_ndhwc_x3: NULL pointer");
  AVT_REQUIRE(bt > 0 && h >= 2 && w >= 2 && c > 0 && c % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0 && ldi >= c && ldo >= c,
              "avt_maxpool_hw2s2_ndhwc_x3: h, w >= 2; channels / leading dimensions must be multiples of 8");
  AVT_REQUIRE(avt::aligned16(in_hi) && avt::aligned16(in_lo) && avt::aligned16(out_hi) && avt::aligned16(out_lo),
              "avt_maxpool_hw2s2_ndhwc_x3: pointers must be 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_maxpool_hw2s2_ndhwc_x3: bad plane_dtype");
  const int ho = h / 2, wo = w / 2;  // floor mode: an odd last row / column is dropped
  const int64_t total = (int64_t)bt * ho * wo * (c / 8);
  AVT_REQUIRE(total < (1ll << 31), "avt_maxpool_hw2s2_ndhwc_x3: more than 2^31 output chunks");
  const int64_t blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
  auto ih = static_cast<const uint16_t*>(in_hi), il = static_cast<const uint16_t*>(in_lo);
  auto oh = static_cast<uint16_t*>(out_hi), ol = static_cast<uint16_t*>(out_lo);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (plane_dtype == AVT_X3_F16)
    hipLaunchKernelGGL((maxpool3_x3_kernel<true, 2, 0>), dim3(grid), dim3(256), 0, st, ih, il, oh, ol, bt, h, w, c, ldi, ldo, ho, wo, 1, nullptr);
  else
    hipLaunchKernelGGL((maxpool3_x3_kernel<false, 2, 0>), dim3(grid), dim3(256), 0, st, ih, il, oh, ol, bt, h, w, c, ldi, ldo, ho, wo, 1, nullptr);
  return avt::check_launch("avt_maxpool_hw2s2_ndhwc_x3");
}

extern "C" int avt_mean_positions_x3(const void* in_hi, const void* in_lo, int batch, int p, int c, int ldi, float* out, int ldo,
                                     int plane_dtype, void* stream) {
  AVT_REQUIRE(in_hi && in_lo && out, "avt_mean_positions_x3: NULL pointer");
  AVT_REQUIRE(batch > 0 && p > 0 && c > 0 && c % 8 == 0 && ldi % 8 == 0 && ldi >= c && ldo >= c,
              "avt_mean_positions_x3: channels / input stride must be multiples of 8, strides must cover the channels");
  AVT_REQUIRE(avt::aligned16(in_hi) && avt::aligned16(in_lo), "avt_mean_positions_x3: input must be 16-byte aligned");
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_mean_positions_x3: bad plane_dtype");
  const dim3 grid((unsigned)batch, (unsigned)((c + 63) / 64));
  auto ih = static_cast<const uint16_t*>(in_hi), il = static_cast<const uint16_t*>(in_lo);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (plane_dtype == AVT_X3_F16)
    hipLaunchKernelGGL((mean_positions_x3_kernel<true>), grid, dim3(256), 0, st, ih, il, p, c, ldi, out, ldo);
  else
    hipLaunchKernelGGL((mean_positions_x3_kernel<false>), grid, dim3(256), 0, st, ih, il, p, c, ldi, out, ldo);
  return avt::check_launch("avt_mean_positions_x3");
}
