// interp — the non-convolution passes of the SuperSloMo frame interpolation the reference runs at every jump of the
// stitched video (contrastive_video_textures/interpolate.py:75-147 `interpolate.forward`, models/slowmo.py:10-135 the
// UNet's pooling / upsampling, :211-284 `backWarp`; called from validate.py:588-611).  The two UNets' convolutions run on
// the split-plane implicit-GEMM kernel (conv_x3.hip, LeakyReLU(0.1) epilogue); everything between them is here, each
// pass HBM-bound over NHWC rows:
//   pack_pair     two uint8 RGB frames -> ToTensor + Normalize(mean, 1) fp32 images and flowComp's 8-channel input planes
//   avgpool2      F.avg_pool2d(x, 2) on plane pairs                                    (slowmo.py:67)
//   upsample2     F.interpolate(scale_factor=2, bilinear, align_corners=False) on plane pairs, into a channel slice of the
//                 concat buffer the next convolution reads                             (slowmo.py:130-134)
//   mid_input     per intermediate time t: F_t_0 / F_t_1 from the two flows, both back-warps, and the 20 (padded to 24)
//                 input channels of ArbTimeFlowIntrp                                   (interpolate.py:108-120)
//   final         refined flows, visibility sigmoid, the two refined back-warps, the blend, un-normalise and the
//                 ToPILImage conversion (x * 255 truncated to uint8)                   (interpolate.py:122-135)
// Arithmetic follows the reference's fp32 operation order (no fused multiply-add where torch rounds twice), so the only
// difference to the fp32 pipeline is the convolutions' 2^-22 split.  grid_sample: bilinear, zero padding,
// align_corners=False (what `grid_sample(img, grid)` means in the torch this image ships).
#include "avt_common.h"
#include "split_planes.h"

namespace {

constexpr int kT = 256;

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }

struct Rgb {
  float r, g, b;
};

// backWarp.forward (slowmo.py:251-284) + grid_sample for ONE output pixel: the grid is pixel + flow, normalised as the
// reference does (2 (x / W - 0.5)), un-normalised as torch does (((g + 1) W - 1) / 2), sampled with zero padding.
__device__ __forceinline__ Rgb backwarp(const float4* __restrict__ img, int H, int W, int px, int py, float u, float v) {
  const float x = add((float)px, u), y = add((float)py, v);
  const float gx = mul(2.0f, sub(__fdiv_rn(x, (float)W), 0.5f)), gy = mul(2.0f, sub(__fdiv_rn(y, (float)H), 0.5f));
  const float ix = __fdiv_rn(sub(mul(add(gx, 1.0f), (float)W), 1.0f), 2.0f);
  const float iy = __fdiv_rn(sub(mul(add(gy, 1.0f), (float)H), 1.0f), 2.0f);
  const float fx = floorf(ix), fy = floorf(iy);
  const float wx1 = sub(ix, fx), wy1 = sub(iy, fy);            // ix - ix_nw
  const float wx0 = sub(add(fx, 1.0f), ix), wy0 = sub(add(fy, 1.0f), iy);  // ix_se - ix
  const float nw = mul(wx0, wy0), ne = mul(wx1, wy0), sw = mul(wx0, wy1), se = mul(wx1, wy1);
  // (a NaN / huge flow makes every corner invalid: zeros, as torch's bounds test does)
  const bool okx0 = fx >= 0.0f && fx <= (float)(W - 1), okx1 = fx + 1.0f >= 0.0f && fx + 1.0f <= (float)(W - 1);
  const bool oky0 = fy >= 0.0f && fy <= (float)(H - 1), oky1 = fy + 1.0f >= 0.0f && fy + 1.0f <= (float)(H - 1);
  const int x0 = okx0 ? (int)fx : 0, x1 = okx1 ? (int)fx + 1 : 0, y0 = oky0 ? (int)fy : 0, y1 = oky1 ? (int)fy + 1 : 0;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 a = okx0 && oky0 ? img[y0 * W + x0] : z;
  const float4 b = okx1 && oky0 ? img[y0 * W + x1] : z;
  const float4 c = okx0 && oky1 ? img[y1 * W + x0] : z;
  const float4 d = okx1 && oky1 ? img[y1 * W + x1] : z;
  Rgb o;
  o.r = add(add(add(mul(a.x, nw), mul(b.x, ne)), mul(c.x, sw)), mul(d.x, se));
  o.g = add(add(add(mul(a.y, nw), mul(b.y, ne)), mul(c.y, sw)), mul(d.y, se));
  o.b = add(add(add(mul(a.z, nw), mul(b.z, ne)), mul(c.z, sw)), mul(d.z, se));
  return o;
}

template <bool F16>
__device__ __forceinline__ void store8(uint16_t* hi, uint16_t* lo, int64_t off, const float* x) {
  uint4 oh, ol;
  avt::split2<F16>(x[0], x[1], oh.x, ol.x);
  avt::split2<F16>(x[2], x[3], oh.y, ol.y);
  avt::split2<F16>(x[4], x[5], oh.z, ol.z);
  avt::split2<F16>(x[6], x[7], oh.w, ol.w);
  *reinterpret_cast<uint4*>(hi + off) = oh;
  *reinterpret_cast<uint4*>(lo + off) = ol;
}
template <bool F16>
__device__ __forceinline__ void load8(const uint16_t* hi, const uint16_t* lo, int64_t off, float* x) {
  const uint4 h = *reinterpret_cast<const uint4*>(hi + off), l = *reinterpret_cast<const uint4*>(lo + off);
  const uint32_t* ph = reinterpret_cast<const uint32_t*>(&h);
  const uint32_t* pl = reinterpret_cast<const uint32_t*>(&l);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const avt::f32x2 r = avt::join2<F16>(ph[e], pl[e]);
    x[2 * e] = r.x;
    x[2 * e + 1] = r.y;
  }
}

template <bool F16>
__global__ __launch_bounds__(kT) void pack_pair_kernel(const uint8_t* __restrict__ f0, const uint8_t* __restrict__ f1, int hw,
                                                       float m0, float m1, float m2, float4* __restrict__ img,
                                                       uint16_t* __restrict__ xh, uint16_t* __restrict__ xl) {
  const int p = blockIdx.x * kT + threadIdx.x;
  if (p >= hw) return;
  // transforms.ToTensor: uint8 -> float / 255; Normalize(mean, std = 1): (x - mean) / 1     (interpolate.py:51-61)
  float x[8];
  x[0] = sub(__fdiv_rn((float)f0[3 * p + 0], 255.0f), m0);
  x[1] = sub(__fdiv_rn((float)f0[3 * p + 1], 255.0f), m1);
  x[2] = sub(__fdiv_rn((float)f0[3 * p + 2], 255.0f), m2);
  x[3] = sub(__fdiv_rn((float)f1[3 * p + 0], 255.0f), m0);
  x[4] = sub(__fdiv_rn((float)f1[3 * p + 1], 255.0f), m1);
  x[5] = sub(__fdiv_rn((float)f1[3 * p + 2], 255.0f), m2);
  x[6] = x[7] = 0.0f;
  img[p] = make_float4(x[0], x[1], x[2], 0.0f);
  img[hw + p] = make_float4(x[3], x[4], x[5], 0.0f);
  store8<F16>(xh, xl, (int64_t)p * 8, x);
}

template <bool F16>
__global__ __launch_bounds__(kT) void avgpool2_kernel(const uint16_t* __restrict__ ih, const uint16_t* __restrict__ il, int b, int H,
                                                      int W, int C, int ldi, uint16_t* __restrict__ oh, uint16_t* __restrict__ ol,
                                                      int ldo) {
  const unsigned cpr = (unsigned)C >> 3, Ho = (unsigned)H >> 1, Wo = (unsigned)W >> 1;
  const unsigned total = (unsigned)b * Ho * Wo * cpr;
  for (unsigned i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    const unsigned p = i / cpr, cc = i - p * cpr;
    const unsigned q = p / Wo, wo = p - q * Wo;
    const unsigned n = q / Ho, ho = q - n * Ho;
    const int64_t r0 = ((int64_t)(n * H + 2 * ho) * W + 2 * wo) * ldi + cc * 8;
    float a[8], c[8], d[8], e[8], o[8];
    load8<F16>(ih, il, r0, a);
    load8<F16>(ih, il, r0 + ldi, c);
    load8<F16>(ih, il, r0 + (int64_t)W * ldi, d);
    load8<F16>(ih, il, r0 + (int64_t)W * ldi + ldi, e);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = mul(add(add(add(a[k], c[k]), d[k]), e[k]), 0.25f);
    store8<F16>(oh, ol, (int64_t)p * ldo + cc * 8, o);
  }
}

// source index / weight of torch's upsample_bilinear2d at scale 2, align_corners=False
__device__ __forceinline__ void up_src(int d, int n, int& i0, int& i1, float& l0, float& l1) {
  float s = sub(mul(add((float)d, 0.5f), 0.5f), 0.5f);
  s = s < 0.0f ? 0.0f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = sub(s, (float)i0);
  l0 = sub(1.0f, l1);
}

template <bool F16>
__global__ __launch_bounds__(kT) void upsample2_kernel(const uint16_t* __restrict__ ih, const uint16_t* __restrict__ il, int b, int h,
                                                       int w, int C, int ldi, uint16_t* __restrict__ oh, uint16_t* __restrict__ ol,
                                                       int ldo) {
  const unsigned cpr = (unsigned)C >> 3, Ho = 2u * h, Wo = 2u * w;
  const unsigned total = (unsigned)b * Ho * Wo * cpr;
  for (unsigned i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    const unsigned p = i / cpr, cc = i - p * cpr;
    const unsigned q = p / Wo, wo = p - q * Wo;
    const unsigned n = q / Ho, ho = q - n * Ho;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    up_src((int)ho, h, y0, y1, ly0, ly1);
    up_src((int)wo, w, x0, x1, lx0, lx1);
    const int64_t base = (int64_t)n * h * w;
    float a[8], c[8], d[8], e[8], o[8];
    load8<F16>(ih, il, (base + y0 * w + x0) * ldi + cc * 8, a);
    load8<F16>(ih, il, (base + y0 * w + x1) * ldi + cc * 8, c);
    load8<F16>(ih, il, (base + y1 * w + x0) * ldi + cc * 8, d);
    load8<F16>(ih, il, (base + y1 * w + x1) * ldi + cc * 8, e);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      o[k] = add(mul(ly0, add(mul(lx0, a[k]), mul(lx1, c[k]))), mul(ly1, add(mul(lx0, d[k]), mul(lx1, e[k]))));
    store8<F16>(oh, ol, (int64_t)p * ldo + cc * 8, o);
  }
}

struct MidCoef {
  float c[16][4];  // per intermediate frame: -t(1-t), t^2, (1-t)^2, -t(1-t)   (interpolate.py:105-107), rounded from double
};

template <bool F16>
__global__ __launch_bounds__(kT) void mid_input_kernel(const float4* __restrict__ img, const uint16_t* __restrict__ fh,
                                                       const uint16_t* __restrict__ fl, int H, int W, int nt, MidCoef co,
                                                       uint16_t* __restrict__ xh, uint16_t* __restrict__ xl, float4* __restrict__ ft) {
  const int hw = H * W;
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= nt * hw) return;
  const int k = i / hw, p = i - k * hw;
  const int py = p / W, px = p - py * W;
  float f[8];
  load8<F16>(fh, fl, (int64_t)p * 8, f);  // flowOut: F_0_1 = [0:2], F_1_0 = [2:4]
  const float c0 = co.c[k][0], c1 = co.c[k][1], c2 = co.c[k][2], c3 = co.c[k][3];
  const float t0x = add(mul(c0, f[0]), mul(c1, f[2])), t0y = add(mul(c0, f[1]), mul(c1, f[3]));  // F_t_0
  const float t1x = add(mul(c2, f[0]), mul(c3, f[2])), t1y = add(mul(c2, f[1]), mul(c3, f[3]));  // F_t_1
  const Rgb g0 = backwarp(img, H, W, px, py, t0x, t0y);
  const Rgb g1 = backwarp(img + hw, H, W, px, py, t1x, t1y);
  const float4 a = img[p], b = img[hw + p];
  // cat(I0, I1, F_0_1, F_1_0, F_t_1, F_t_0, g_I1_F_t_1, g_I0_F_t_0)   (interpolate.py:117-118), padded 20 -> 24
  float x[24] = {a.x, a.y, a.z, b.x, b.y, b.z, f[0], f[1], f[2], f[3], t1x, t1y, t0x, t0y, g1.r, g1.g,
                 g1.b, g0.r, g0.g, g0.b, 0.f, 0.f, 0.f, 0.f};
  const int64_t row = (int64_t)i * 24;
  store8<F16>(xh, xl, row, x);
  store8<F16>(xh, xl, row + 8, x + 8);
  store8<F16>(xh, xl, row + 16, x + 16);
  ft[i] = make_float4(t0x, t0y, t1x, t1y);
}

struct FinCoef {
  float w0[16], w1[16];  // 1 - t, t
};

template <bool F16>
__global__ __launch_bounds__(kT) void final_kernel(const float4* __restrict__ img, const float4* __restrict__ ft,
                                                   const uint16_t* __restrict__ oh, const uint16_t* __restrict__ ol, int H, int W, int nt,
                                                   FinCoef co, float m0, float m1, float m2, uint8_t* __restrict__ out) {
  const int hw = H * W;
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= nt * hw) return;
  const int k = i / hw, p = i - k * hw;
  const int py = p / W, px = p - py * W;
  float o[8];
  load8<F16>(oh, ol, (int64_t)i * 8, o);  // intrpOut: dF_t_0 [0:2], dF_t_1 [2:4], visibility logit [4]
  const float4 t = ft[i];
  const float f0x = add(o[0], t.x), f0y = add(o[1], t.y), f1x = add(o[2], t.z), f1y = add(o[3], t.w);
  const float v0 = __fdiv_rn(1.0f, add(1.0f, expf(-o[4])));  // torch.sigmoid
  const float v1 = sub(1.0f, v0);
  const Rgb g0 = backwarp(img, H, W, px, py, f0x, f0y);
  const Rgb g1 = backwarp(img + hw, H, W, px, py, f1x, f1y);
  const float a0 = mul(co.w0[k], v0), a1 = mul(co.w1[k], v1);
  const float den = add(a0, a1);
  // Ft_p = (w0 V0 g0 + w1 V1 g1) / (w0 V0 + w1 V1); revNormalize: x - (-mean); ToPILImage: mul(255).byte()
  const float r = __fdiv_rn(add(mul(a0, g0.r), mul(a1, g1.r)), den);
  const float g = __fdiv_rn(add(mul(a0, g0.g), mul(a1, g1.g)), den);
  const float b = __fdiv_rn(add(mul(a0, g0.b), mul(a1, g1.b)), den);
  out[3 * (int64_t)i + 0] = (uint8_t)(int)mul(sub(r, -m0), 255.0f);
  out[3 * (int64_t)i + 1] = (uint8_t)(int)mul(sub(g, -m1), 255.0f);
  out[3 * (int64_t)i + 2] = (uint8_t)(int)mul(sub(b, -m2), 255.0f);
}

inline unsigned blocks_for(int64_t n) {
  int64_t b = (n + kT - 1) / kT;
  return (unsigned)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

}  // namespace

#define AVT_PLANES(pd, who) AVT_REQUIRE((pd) == 0 || (pd) == 1, who ": plane_dtype must be 0 (bf16) or 1 (fp16)")

extern "C" int avt_interp_pack_pair_u8(const uint8_t* frame0, const uint8_t* frame1, int height, int width, const float* mean3,
                                       float* img, void* x_hi, void* x_lo, int plane_dtype, void* stream) {
  AVT_REQUIRE(frame0 && frame1 && mean3 && img && x_hi && x_lo, "avt_interp_pack_pair_u8: NULL pointer");
  AVT_REQUIRE(height > 0 && width > 0 && (int64_t)height * width <= (1 << 24), "avt_interp_pack_pair_u8: bad frame size %d x %d", height, width);
  AVT_REQUIRE(avt::aligned16(img) && avt::aligned16(x_hi) && avt::aligned16(x_lo), "avt_interp_pack_pair_u8: outputs must be 16-byte aligned");
  AVT_PLANES(plane_dtype, "avt_interp_pack_pair_u8");
  const int hw = height * width;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto* xh = static_cast<uint16_t*>(x_hi);
  auto* xl = static_cast<uint16_t*>(x_lo);
  if (plane_dtype)
    hipLaunchKernelGGL(pack_pair_kernel<true>, dim3(blocks_for(hw)), dim3(kT), 0, st, frame0, frame1, hw, mean3[0], mean3[1], mean3[2],
                       reinterpret_cast<float4*>(img), xh, xl);
  else
    hipLaunchKernelGGL(pack_pair_kernel<false>, dim3(blocks_for(hw)), dim3(kT), 0, st, frame0, frame1, hw, mean3[0], mean3[1], mean3[2],
                       reinterpret_cast<float4*>(img), xh, xl);
  return avt::check_launch("avt_interp_pack_pair_u8");
}

extern "C" int avt_avgpool2_x3(const void* in_hi, const void* in_lo, int batch, int h, int w, int c, int ldi, void* out_hi, void* out_lo,
                               int ldo, int plane_dtype, void* stream) {
  AVT_REQUIRE(in_hi && in_lo && out_hi && out_lo, "avt_avgpool2_x3: NULL pointer");
  AVT_REQUIRE(batch > 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0 && c > 0 && c % 8 == 0 && ldi >= c && ldo >= c && ldi % 8 == 0 && ldo % 8 == 0,
              "avt_avgpool2_x3: even extents and channel counts / row strides in multiples of 8 (got %dx%dx%dx%d, ld %d -> %d)", batch, h, w, c, ldi, ldo);
  AVT_REQUIRE((int64_t)batch * h * w * (c / 8) < (1ll << 31), "avt_avgpool2_x3: too many elements");
  AVT_REQUIRE(avt::aligned16(in_hi) && avt::aligned16(in_lo) && avt::aligned16(out_hi) && avt::aligned16(out_lo), "avt_avgpool2_x3: 16-byte alignment");
  AVT_PLANES(plane_dtype, "avt_avgpool2_x3");
  const int64_t total = (int64_t)batch * (h / 2) * (w / 2) * (c / 8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto ih = static_cast<const uint16_t*>(in_hi), il = static_cast<const uint16_t*>(in_lo);
  auto oh = static_cast<uint16_t*>(out_hi), ol = static_cast<uint16_t*>(out_lo);
  if (plane_dtype)
    hipLaunchKernelGGL(avgpool2_kernel<true>, dim3(blocks_for(total)), dim3(kT), 0, st, ih, il, batch, h, w, c, ldi, oh, ol, ldo);
  else
    hipLaunchKernelGGL(avgpool2_kernel<false>, dim3(blocks_for(total)), dim3(kT), 0, st, ih, il, batch, h, w, c, ldi, oh, ol, ldo);
  return avt::check_launch("avt_avgpool2_x3");
}

extern "C" int avt_upsample2_bilinear_x3(const void* in_hi, const void* in_lo, int batch, int h, int w, int c, int ldi, void* out_hi,
                                         void* out_lo, int ldo, int plane_dtype, void* stream) {
  AVT_REQUIRE(in_hi && in_lo && out_hi && out_lo, "avt_upsample2_bilinear_x3: NULL pointer");
  AVT_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && ldi >= c && ldo >= c && ldi % 8 == 0 && ldo % 8 == 0,
              "avt_upsample2_bilinear_x3: channel counts / row strides in multiples of 8 (got %dx%dx%dx%d, ld %d -> %d)", batch, h, w, c, ldi, ldo);
  AVT_REQUIRE((int64_t)batch * h * w * 4 * (c / 8) < (1ll << 31), "avt_upsample2_bilinear_x3: too many elements");
  AVT_REQUIRE(avt::aligned16(in_hi) && avt::aligned16(in_lo) && avt::aligned16(out_hi) && avt::aligned16(out_lo), "avt_upsample2_bilinear_x3: 16-byte alignment");
  AVT_PLANES(plane_dtype, "avt_upsample2_bilinear_x3");
  const int64_t total = (int64_t)batch * h * w * 4 * (c / 8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto ih = static_cast<const uint16_t*>(in_hi), il = static_cast<const uint16_t*>(in_lo);
  auto oh = static_cast<uint16_t*>(out_hi), ol = static_cast<uint16_t*>(out_lo);
  if (plane_dtype)
    hipLaunchKernelGGL(upsample2_kernel<true>, dim3(blocks_for(total)), dim3(kT), 0, st, ih, il, batch, h, w, c, ldi, oh, ol, ldo);
  else
    hipLaunchKernelGGL(upsample2_kernel<false>, dim3(blocks_for(total)), dim3(kT), 0, st, ih, il, batch, h, w, c, ldi, oh, ol, ldo);
  return avt::check_launch("avt_upsample2_bilinear_x3");
}

extern "C" int avt_interp_mid_input(const float* img, const void* flow_hi, const void* flow_lo, int height, int width, int sf,
                                    void* x_hi, void* x_lo, float* ft, int plane_dtype, void* stream) {
  AVT_REQUIRE(img && flow_hi && flow_lo && x_hi && x_lo && ft, "avt_interp_mid_input: NULL pointer");
  AVT_REQUIRE(height > 0 && width > 0 && sf >= 2 && sf <= 17 && (int64_t)height * width * (sf - 1) <= (1 << 24),
              "avt_interp_mid_input: frame %d x %d, slomo factor %d (2..17)", height, width, sf);
  AVT_REQUIRE(avt::aligned16(img) && avt::aligned16(flow_hi) && avt::aligned16(flow_lo) && avt::aligned16(x_hi) && avt::aligned16(x_lo) && avt::aligned16(ft),
              "avt_interp_mid_input: 16-byte alignment");
  AVT_PLANES(plane_dtype, "avt_interp_mid_input");
  MidCoef co = {};
  for (int i = 1; i < sf; ++i) {  // interpolate.py:103-107, in double like the Python floats, rounded once
    const double t = (double)i / (double)sf, temp = -t * (1.0 - t);
    co.c[i - 1][0] = (float)temp;
    co.c[i - 1][1] = (float)(t * t);
    co.c[i - 1][2] = (float)((1.0 - t) * (1.0 - t));
    co.c[i - 1][3] = (float)temp;
  }
  const int nt = sf - 1;
  const int64_t total = (int64_t)nt * height * width;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto fh = static_cast<const uint16_t*>(flow_hi), fl = static_cast<const uint16_t*>(flow_lo);
  auto xh = static_cast<uint16_t*>(x_hi), xl = static_cast<uint16_t*>(x_lo);
  if (plane_dtype)
    hipLaunchKernelGGL(mid_input_kernel<true>, dim3(blocks_for(total)), dim3(kT), 0, st, reinterpret_cast<const float4*>(img), fh, fl, height,
                       width, nt, co, xh, xl, reinterpret_cast<float4*>(ft));
  else
    hipLaunchKernelGGL(mid_input_kernel<false>, dim3(blocks_for(total)), dim3(kT), 0, st, reinterpret_cast<const float4*>(img), fh, fl, height,
                       width, nt, co, xh, xl, reinterpret_cast<float4*>(ft));
  return avt::check_launch("avt_interp_mid_input");
}

extern "C" int avt_interp_final_u8(const float* img, const float* ft, const void* o_hi, const void* o_lo, int height, int width, int sf,
                                   const float* mean3, uint8_t* out, int plane_dtype, void* stream) {
  AVT_REQUIRE(img && ft && o_hi && o_lo && mean3 && out, "avt_interp_final_u8: NULL pointer");
  AVT_REQUIRE(height > 0 && width > 0 && sf >= 2 && sf <= 17 && (int64_t)height * width * (sf - 1) <= (1 << 24),
              "avt_interp_final_u8: frame %d x %d, slomo factor %d (2..17)", height, width, sf);
  AVT_REQUIRE(avt::aligned16(img) && avt::aligned16(ft) && avt::aligned16(o_hi) && avt::aligned16(o_lo), "avt_interp_final_u8: 16-byte alignment");
  AVT_PLANES(plane_dtype, "avt_interp_final_u8");
  FinCoef co = {};
  for (int i = 1; i < sf; ++i) {  // wCoeff = [1 - t, t]   (interpolate.py:130)
    const double t = (double)i / (double)sf;
    co.w0[i - 1] = (float)(1.0 - t);
    co.w1[i - 1] = (float)t;
  }
  const int nt = sf - 1;
  const int64_t total = (int64_t)nt * height * width;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto oh = static_cast<const uint16_t*>(o_hi), ol = static_cast<const uint16_t*>(o_lo);
  if (plane_dtype)
    hipLaunchKernelGGL(final_kernel<true>, dim3(blocks_for(total)), dim3(kT), 0, st, reinterpret_cast<const float4*>(img),
                       reinterpret_cast<const float4*>(ft), oh, ol, height, width, nt, co, mean3[0], mean3[1], mean3[2], out);
  else
    hipLaunchKernelGGL(final_kernel<false>, dim3(blocks_for(total)), dim3(kT), 0, st, reinterpret_cast<const float4*>(img),
                       reinterpret_cast<const float4*>(ft), oh, ol, height, width, nt, co, mean3[0], mean3[1], mean3[2], out);
  return avt::check_launch("avt_interp_final_u8");
}
