// clip_pack — uint8 video frames -> SlowFast pathway tensors, for gfx950.
// Replaces the per-window Python preprocessing of the reference
// (contrastive_video_textures/models/models.py:364-383, validate.py:120-125 and
// :333-344, dataset/dataset.py:68-73 and :145-154): /255, RGB->BGR,
// (x-0.45)/0.225, temporal resample of the W-frame window to 32 (fast) and 8
// (slow) frames by linspace().long(), per-plane bilinear resize
// (align_corners=False) to out_hw x out_hw.
//
// HBM-write-bound: every clip costs 3*(8+32)*out_hw^2 output elements
// (12.0 MB in bf16 at 224) but only W source frames of H*W*3 bytes, and
// neighbouring windows share frames (W/S of them each) while the fast pathway
// repeats frames when W < 32.  So the kernel is organised by SOURCE frame: a
// workgroup resizes a strip of one (frame, channel) plane once, keeps it in
// registers, and stores it to every (window, slot) that samples the frame
// (CSR plan built on the host by avt_clip_pack_plan).  Each lane owns 8
// consecutive output pixels of a row = one 16-byte bf16 store per destination.
#include <math.h>
#include <string.h>

#include "avt_common.h"
#include "split_planes.h"

namespace {

struct PArgs {
  const uint8_t* frames;
  int n_frames, H, W;
  const int32_t* dst_off;
  const int32_t* dst_slot;
  int hw;
  float mean, std;
  int bgr;
  void* slow;
  void* fast;
  void* slow_lo;  // split-plane packing only
  void* fast_lo;
  int cpr;    // 8-pixel chunks per output row
  int rpb;    // output rows per workgroup
  int tiles;  // row strips per plane
  float scale_h, scale_w;
};

template <typename T>
struct Out;
template <>
struct Out<float> {
  static __device__ __forceinline__ float cvt(float v) { return v; }
};
template <>
struct Out<uint16_t> {
  static __device__ __forceinline__ uint16_t cvt(float v) { return (uint16_t)(avt::pack_bf16x2(v, 0.0f) & 0xffffu); }
};

// torch area_pixel_compute_source_index(scale, dst, align_corners=False, cubic=False)
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
  // one fused multiply-add, as the reference's device build of this expression contracts it
  // (the fraction of s is the bilinear weight, so its last bit is worth ~1e-5 in the output)
  float s = __fmaf_rn(scale, (float)dst + 0.5f, -0.5f);
  s = s < 0.0f ? 0.0f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void clip_pack_kernel(PArgs a) {
  __shared__ float lut[256];
  // normalisation LUT in the reference's op order: (v/255 - mean)/std, each step one fp32 rounding
  lut[threadIdx.x] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)threadIdx.x, 255.0f), a.mean), a.std);
  __syncthreads();

  int b = blockIdx.x;
  const int tile = b % a.tiles;
  b /= a.tiles;
  const int c = b % 3;
  const int f = b / 3;
  const int e0 = a.dst_off[f], e1 = a.dst_off[f + 1];
  if (e0 == e1) return;  // frame not sampled by any window

  const int tid = threadIdx.x;
  const int ry = tid / a.cpr, cx = tid - ry * a.cpr;
  const int y = tile * a.rpb + ry;
  if (ry >= a.rpb || y >= a.hw) return;
  const int x0 = cx * 8;

  int y0, y1;
  float ly;
  src_index(a.scale_h, y, a.H, y0, y1, ly);
  const float hy = 1.0f - ly;
  const int sc = a.bgr ? 2 - c : c;
  const uint8_t* r0 = a.frames + ((int64_t)f * a.H + y0) * a.W * 3 + sc;
  const uint8_t* r1 = a.frames + ((int64_t)f * a.H + y1) * a.W * 3 + sc;

  T v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int x = x0 + j;
    float o = 0.0f;
    if (x < a.hw) {
      int xa, xb;
      float lx;
      src_index(a.scale_w, x, a.W, xa, xb, lx);
      const float hx = 1.0f - lx;
      const float p00 = lut[r0[xa * 3]], p01 = lut[r0[xb * 3]];
      const float p10 = lut[r1[xa * 3]], p11 = lut[r1[xb * 3]];
      o = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
    }
    v[j] = Out<T>::cvt(o);
  }

  const int64_t plane = (int64_t)a.hw * a.hw;
  const int64_t inplane = (int64_t)y * a.hw + x0;
  for (int e = e0; e < e1; ++e) {
    const int ds = a.dst_slot[e];
    const int n = ds / AVT_SLOTS, slot = ds - n * AVT_SLOTS;
    T* dst = slot < AVT_SLOW_T
                 ? static_cast<T*>(a.slow) + (((int64_t)n * 3 + c) * AVT_SLOW_T + slot) * plane
                 : static_cast<T*>(a.fast) + (((int64_t)n * 3 + c) * AVT_FAST_T + (slot - AVT_SLOW_T)) * plane;
    dst += inplane;
    if (VEC) {
      if (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(v);
      } else {
        reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(v)[0];
        reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(v)[1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (x0 + j < a.hw) dst[j] = v[j];
    }
  }
}

// Gather form for TRAINING batches (config 5: a query, its positive and the sampled negatives are scattered windows with
// little frame reuse): organised by DESTINATION — one workgroup per (window, slot, channel, row strip) — with the window
// starts read from a DEVICE array, so a step needs no host-built plan and no host round trip between the device-side
// negative sampling (negsample.hip) and the packing.  Same arithmetic as clip_pack_kernel, value for value.
struct GArgs {
  const uint8_t* frames;
  int n_frames, H, W;
  const int32_t* win_start;  // [n_win] device
  int n_win, win_len;
  int hw;
  float mean, std;
  int bgr;
  void* slow;
  void* fast;
  int cpr, rpb, tiles;
  float scale_h, scale_w;
  int fast_idx[AVT_FAST_T], slow_idx[AVT_SLOW_T];
};

template <typename T>
__global__ __launch_bounds__(256) void clip_pack_gather_kernel(GArgs a) {
  __shared__ float lut[256];
  lut[threadIdx.x] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)threadIdx.x, 255.0f), a.mean), a.std);
  __syncthreads();
  int b = blockIdx.x;
  const int tile = b % a.tiles;
  b /= a.tiles;
  const int c = b % 3;
  b /= 3;
  const int slot = b % AVT_SLOTS;
  const int n = b / AVT_SLOTS;
  int f = a.win_start[n] + (slot < AVT_SLOW_T ? a.slow_idx[slot] : a.fast_idx[slot - AVT_SLOW_T]);
  f = f < 0 ? 0 : (f > a.n_frames - 1 ? a.n_frames - 1 : f);  // a bad start cannot fault (range of the ids: checked once by dataset.DeviceSegmentBatcher)
  const int tid = threadIdx.x;
  const int ry = tid / a.cpr, cx = tid - ry * a.cpr;
  const int y = tile * a.rpb + ry;
  if (ry >= a.rpb || y >= a.hw) return;
  const int x0 = cx * 8;
  int y0, y1;
  float ly;
  src_index(a.scale_h, y, a.H, y0, y1, ly);
  const float hy = 1.0f - ly;
  const int sc = a.bgr ? 2 - c : c;
  const uint8_t* r0 = a.frames + ((int64_t)f * a.H + y0) * a.W * 3 + sc;
  const uint8_t* r1 = a.frames + ((int64_t)f * a.H + y1) * a.W * 3 + sc;
  const int64_t plane = (int64_t)a.hw * a.hw;
  T* dst = slot < AVT_SLOW_T
               ? static_cast<T*>(a.slow) + (((int64_t)n * 3 + c) * AVT_SLOW_T + slot) * plane
               : static_cast<T*>(a.fast) + (((int64_t)n * 3 + c) * AVT_FAST_T + (slot - AVT_SLOW_T)) * plane;
  dst += (int64_t)y * a.hw + x0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int x = x0 + j;
    if (x < a.hw) {
      int xa, xb;
      float lx;
      src_index(a.scale_w, x, a.W, xa, xb, lx);
      const float hx = 1.0f - lx;
      const float p00 = lut[r0[xa * 3]], p01 = lut[r0[xb * 3]];
      const float p10 = lut[r1[xa * 3]], p11 = lut[r1[xb * 3]];
      dst[j] = Out<T>::cvt(hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11));
    }
  }
}

// Channels-last variant for the MFMA stem: slow [n,8,hw,hw,4], fast [n,32,hw,hw,4] bf16 (NDHWC, C padded
// 3 -> 4 with a zero).  A workgroup resizes a strip of one source frame for all three channels; each lane owns
// 4 consecutive output pixels = 32 contiguous bytes per destination.
// PL = 0: one bf16 plane (the fast bf16 encoder); PL = 1 / 2: (hi, lo) bf16 / fp16 planes for the split-plane ("x3")
// contract-grade encoder (a.slow_lo / a.fast_lo = the low-order planes, same geometry).
template <int PL>
__global__ __launch_bounds__(256) void clip_pack_nhwc4_kernel(PArgs a) {
  __shared__ float lut[256];
  lut[threadIdx.x] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)threadIdx.x, 255.0f), a.mean), a.std);
  __syncthreads();
  const int tile = blockIdx.x % a.tiles;
  const int f = blockIdx.x / a.tiles;
  const int e0 = a.dst_off[f], e1 = a.dst_off[f + 1];
  if (e0 == e1) return;
  const int tid = threadIdx.x;
  const int ry = tid / a.cpr, cx = tid - ry * a.cpr;
  const int y = tile * a.rpb + ry;
  if (ry >= a.rpb || y >= a.hw) return;
  const int x0 = cx * 4;
  int y0, y1;
  float ly;
  src_index(a.scale_h, y, a.H, y0, y1, ly);
  const float hy = 1.0f - ly;
  const uint8_t* r0 = a.frames + ((int64_t)f * a.H + y0) * a.W * 3;
  const uint8_t* r1 = a.frames + ((int64_t)f * a.H + y1) * a.W * 3;
  uint16_t v[16], vlo[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int x = x0 + j;
    int xa = 0, xb = 0;
    float lx = 0.0f;
    if (x < a.hw) src_index(a.scale_w, x, a.W, xa, xb, lx);
    const float hx = 1.0f - lx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int sc = a.bgr ? 2 - c : c;
      const float p00 = lut[r0[xa * 3 + sc]], p01 = lut[r0[xb * 3 + sc]];
      const float p10 = lut[r1[xa * 3 + sc]], p11 = lut[r1[xb * 3 + sc]];
      const float o = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
      if constexpr (PL == 0) {
        v[j * 4 + c] = (x < a.hw) ? (uint16_t)(avt::pack_bf16x2(o, 0.0f) & 0xffffu) : (uint16_t)0;
      } else {
        uint32_t h2, l2;
        avt::split2<PL == 2>(o, 0.0f, h2, l2);
        v[j * 4 + c] = (x < a.hw) ? (uint16_t)(h2 & 0xffffu) : (uint16_t)0;
        vlo[j * 4 + c] = (x < a.hw) ? (uint16_t)(l2 & 0xffffu) : (uint16_t)0;
      }
    }
    v[j * 4 + 3] = 0;
    vlo[j * 4 + 3] = 0;
  }
  const int64_t plane = (int64_t)a.hw * a.hw * 4;
  const int64_t inplane = ((int64_t)y * a.hw + x0) * 4;
  const bool full = x0 + 4 <= a.hw;
  for (int e = e0; e < e1; ++e) {
    const int ds = a.dst_slot[e];
    const int n = ds / AVT_SLOTS, slot = ds - n * AVT_SLOTS;
    uint16_t* dst = slot < AVT_SLOW_T
                        ? static_cast<uint16_t*>(a.slow) + ((int64_t)n * AVT_SLOW_T + slot) * plane
                        : static_cast<uint16_t*>(a.fast) + ((int64_t)n * AVT_FAST_T + (slot - AVT_SLOW_T)) * plane;
    dst += inplane;
    if (full) {
      reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(v)[0];
      reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(v)[1];
    } else {
      for (int j = 0; j < 16; ++j)
        if (x0 + j / 4 < a.hw) dst[j] = v[j];
    }
    if constexpr (PL != 0) {
      uint16_t* dl = slot < AVT_SLOW_T
                         ? static_cast<uint16_t*>(a.slow_lo) + ((int64_t)n * AVT_SLOW_T + slot) * plane
                         : static_cast<uint16_t*>(a.fast_lo) + ((int64_t)n * AVT_FAST_T + (slot - AVT_SLOW_T)) * plane;
      dl += inplane;
      if (full) {
        reinterpret_cast<uint4*>(dl)[0] = reinterpret_cast<const uint4*>(vlo)[0];
        reinterpret_cast<uint4*>(dl)[1] = reinterpret_cast<const uint4*>(vlo)[1];
      } else {
        for (int j = 0; j < 16; ++j)
          if (x0 + j / 4 < a.hw) dl[j] = vlo[j];
      }
    }
  }
}

// torch.linspace(start=0, end, steps) in fp32 (ATen RangeFactories: symmetric halves), then .long()
void linspace_long(float end, int steps, int32_t* out) {
  if (steps == 1) {
    out[0] = 0;
    return;
  }
  const float step = end / (float)(steps - 1);
  const int half = steps / 2;
  for (int i = 0; i < steps; ++i) {
    const float v = i < half ? step * (float)i : end - step * (float)(steps - 1 - i);
    out[i] = (int32_t)v;
  }
}

}  // namespace

extern "C" int avt_clip_sample_table(int win_len, int32_t* fast_idx, int32_t* slow_idx) {
  AVT_REQUIRE(win_len > 0 && fast_idx && slow_idx, "avt_clip_sample_table: bad arguments");
  linspace_long((float)(win_len - 1), AVT_FAST_T, fast_idx);
  int32_t pick[AVT_SLOW_T];
  linspace_long((float)(AVT_FAST_T - 1), AVT_SLOW_T, pick);
  for (int i = 0; i < AVT_SLOW_T; ++i) slow_idx[i] = fast_idx[pick[i]];
  return AVT_OK;
}

extern "C" int avt_clip_pack_plan(const int32_t* win_start, int n_win, int win_len, int n_frames, int32_t* dst_off,
                                  int32_t* dst_slot) {
  AVT_REQUIRE(win_start && dst_off && dst_slot, "avt_clip_pack_plan: NULL pointer");
  AVT_REQUIRE(n_win >= 0 && win_len > 0 && n_frames > 0, "avt_clip_pack_plan: bad sizes");
  AVT_REQUIRE((int64_t)n_win * AVT_SLOTS < (1ll << 31), "avt_clip_pack_plan: too many windows");
  int32_t fast_idx[AVT_FAST_T], slow_idx[AVT_SLOW_T];
  avt_clip_sample_table(win_len, fast_idx, slow_idx);
  for (int f = 0; f <= n_frames; ++f) dst_off[f] = 0;
  for (int n = 0; n < n_win; ++n) {
    AVT_REQUIRE(win_start[n] >= 0 && (int64_t)win_start[n] + win_len <= n_frames,
                "avt_clip_pack_plan: window %d [%d,%d) leaves [0,%d)", n, win_start[n], win_start[n] + win_len,
                n_frames);
    for (int s = 0; s < AVT_SLOW_T; ++s) dst_off[win_start[n] + slow_idx[s] + 1]++;
    for (int s = 0; s < AVT_FAST_T; ++s) dst_off[win_start[n] + fast_idx[s] + 1]++;
  }
  for (int f = 0; f < n_frames; ++f) dst_off[f + 1] += dst_off[f];
  // second pass fills the lists (cursor kept in a scratch copy at the tail end of dst_off's values)
  int32_t* cur = new int32_t[n_frames];
  memcpy(cur, dst_off, sizeof(int32_t) * (size_t)n_frames);
  for (int n = 0; n < n_win; ++n) {
    for (int s = 0; s < AVT_SLOW_T; ++s) dst_slot[cur[win_start[n] + slow_idx[s]]++] = n * AVT_SLOTS + s;
    for (int s = 0; s < AVT_FAST_T; ++s) dst_slot[cur[win_start[n] + fast_idx[s]]++] = n * AVT_SLOTS + AVT_SLOW_T + s;
  }
  delete[] cur;
  return AVT_OK;
}

extern "C" int avt_clip_pack_u8(const uint8_t* frames, int n_frames, int height, int width, const int32_t* dst_off,
                                const int32_t* dst_slot, int n_win, int out_hw, float mean, float std, int bgr,
                                void* slow, void* fast, int out_dtype, void* stream) {
  AVT_REQUIRE(n_frames > 0 && height > 0 && width > 0 && out_hw > 0 && n_win >= 0, "avt_clip_pack_u8: bad sizes");
  if (n_win == 0) return AVT_OK;
  AVT_REQUIRE(frames && dst_off && dst_slot && slow && fast, "avt_clip_pack_u8: NULL pointer");
  AVT_REQUIRE(out_hw <= 2048, "avt_clip_pack_u8: out_hw > 2048");
  AVT_REQUIRE(std != 0.0f, "avt_clip_pack_u8: std == 0");
  AVT_REQUIRE(out_dtype == AVT_DT_F32 || out_dtype == AVT_DT_BF16, "avt_clip_pack_u8: unknown out_dtype %d", out_dtype);
  if (n_win == 0) return AVT_OK;
  PArgs a;
  a.frames = frames;
  a.n_frames = n_frames;
  a.H = height;
  a.W = width;
  a.dst_off = dst_off;
  a.dst_slot = dst_slot;
  a.hw = out_hw;
  a.mean = mean;
  a.std = std;
  a.bgr = bgr;
  a.slow = slow;
  a.fast = fast;
  a.cpr = (out_hw + 7) / 8;
  a.rpb = 256 / a.cpr;
  a.tiles = (out_hw + a.rpb - 1) / a.rpb;
  a.scale_h = (float)height / (float)out_hw;
  a.scale_w = (float)width / (float)out_hw;
  const int64_t nblk = (int64_t)a.tiles * 3 * n_frames;
  AVT_REQUIRE(nblk < (1ll << 31), "avt_clip_pack_u8: grid too large");
  const bool vec = (out_hw % 8 == 0) && avt::aligned16(slow) && avt::aligned16(fast);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)nblk), block(256);
  if (out_dtype == AVT_DT_BF16) {
    if (vec)
      hipLaunchKernelGGL((clip_pack_kernel<uint16_t, true>), grid, block, 0, st, a);
    else
      hipLaunchKernelGGL((clip_pack_kernel<uint16_t, false>), grid, block, 0, st, a);
  } else {
    if (vec)
      hipLaunchKernelGGL((clip_pack_kernel<float, true>), grid, block, 0, st, a);
    else
      hipLaunchKernelGGL((clip_pack_kernel<float, false>), grid, block, 0, st, a);
  }
  return avt::check_launch("avt_clip_pack_u8");
}

static int clip_pack_ndhwc4_impl(const uint8_t* frames, int n_frames, int height, int width, const int32_t* dst_off,
                                 const int32_t* dst_slot, int n_win, int out_hw, float mean, float std, int bgr, void* slow,
                                 void* fast, void* slow_lo, void* fast_lo, int planes, void* stream) {
  AVT_REQUIRE(n_frames > 0 && height > 0 && width > 0 && out_hw > 0 && n_win >= 0, "avt_clip_pack_u8_ndhwc4: bad sizes");
  if (n_win == 0) return AVT_OK;
  AVT_REQUIRE(frames && dst_off && dst_slot && slow && fast, "avt_clip_pack_u8_ndhwc4: NULL pointer");
  AVT_REQUIRE(out_hw <= 1020 && out_hw % 2 == 0, "avt_clip_pack_u8_ndhwc4: out_hw must be even and <= 1020");
  AVT_REQUIRE(std != 0.0f, "avt_clip_pack_u8_ndhwc4: std == 0");
  AVT_REQUIRE(avt::aligned16(slow) && avt::aligned16(fast), "avt_clip_pack_u8_ndhwc4: outputs must be 16-byte aligned");
  if (n_win == 0) return AVT_OK;
  PArgs a;
  a.frames = frames;
  a.n_frames = n_frames;
  a.H = height;
  a.W = width;
  a.dst_off = dst_off;
  a.dst_slot = dst_slot;
  a.hw = out_hw;
  a.mean = mean;
  a.std = std;
  a.bgr = bgr;
  a.slow = slow;
  a.fast = fast;
  a.slow_lo = slow_lo;
  a.fast_lo = fast_lo;
  a.cpr = (out_hw + 3) / 4;
  a.rpb = 256 / a.cpr;
  a.tiles = (out_hw + a.rpb - 1) / a.rpb;
  a.scale_h = (float)height / (float)out_hw;
  a.scale_w = (float)width / (float)out_hw;
  const int64_t nblk = (int64_t)a.tiles * n_frames;
  AVT_REQUIRE(nblk < (1ll << 31), "avt_clip_pack_u8_ndhwc4: grid too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (planes == 0)
    hipLaunchKernelGGL(clip_pack_nhwc4_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, st, a);
  else if (planes == 1)
    hipLaunchKernelGGL(clip_pack_nhwc4_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(clip_pack_nhwc4_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, st, a);
  return avt::check_launch("avt_clip_pack_u8_ndhwc4");
}

extern "C" int avt_clip_pack_u8_ndhwc4(const uint8_t* frames, int n_frames, int height, int width,
                                       const int32_t* dst_off, const int32_t* dst_slot, int n_win, int out_hw, float mean,
                                       float std, int bgr, void* slow, void* fast, void* stream) {
  return clip_pack_ndhwc4_impl(frames, n_frames, height, width, dst_off, dst_slot, n_win, out_hw, mean, std, bgr, slow, fast,
                               nullptr, nullptr, 0, stream);
}

extern "C" int avt_clip_pack_u8_ndhwc4_x3(const uint8_t* frames, int n_frames, int height, int width,
                                          const int32_t* dst_off, const int32_t* dst_slot, int n_win, int out_hw, float mean,
                                          float std, int bgr, void* slow_hi, void* slow_lo, void* fast_hi, void* fast_lo,
                                          int plane_dtype, void* stream) {
  AVT_REQUIRE(plane_dtype == AVT_X3_BF16 || plane_dtype == AVT_X3_F16, "avt_clip_pack_u8_ndhwc4_x3: bad plane_dtype");
  AVT_REQUIRE(n_win == 0 || (slow_lo && fast_lo && avt::aligned16(slow_lo) && avt::aligned16(fast_lo)),
              "avt_clip_pack_u8_ndhwc4_x3: the low-order planes must be 16-byte aligned device buffers");
  return clip_pack_ndhwc4_impl(frames, n_frames, height, width, dst_off, dst_slot, n_win, out_hw, mean, std, bgr, slow_hi,
                               fast_hi, slow_lo, fast_lo, plane_dtype == AVT_X3_F16 ? 2 : 1, stream);
}

extern "C" int avt_clip_pack_gather_u8(const uint8_t* frames, int n_frames, int height, int width, const int32_t* win_start,
                                       int n_win, int win_len, int out_hw, float mean, float std, int bgr, void* slow,
                                       void* fast, int out_dtype, void* stream) {
  AVT_REQUIRE(n_frames > 0 && height > 0 && width > 0 && out_hw > 0 && n_win >= 0 && win_len > 0 && win_len <= n_frames,
              "avt_clip_pack_gather_u8: bad sizes");
  if (n_win == 0) return AVT_OK;
  AVT_REQUIRE(frames && win_start && slow && fast, "avt_clip_pack_gather_u8: NULL pointer");
  AVT_REQUIRE(out_hw <= 2048 && std != 0.0f, "avt_clip_pack_gather_u8: out_hw > 2048 or std == 0");
  AVT_REQUIRE(out_dtype == AVT_DT_F32 || out_dtype == AVT_DT_BF16, "avt_clip_pack_gather_u8: unknown out_dtype %d", out_dtype);
  GArgs a;
  a.frames = frames;
  a.n_frames = n_frames;
  a.H = height;
  a.W = width;
  a.win_start = win_start;
  a.n_win = n_win;
  a.win_len = win_len;
  a.hw = out_hw;
  a.mean = mean;
  a.std = std;
  a.bgr = bgr;
  a.slow = slow;
  a.fast = fast;
  a.cpr = (out_hw + 7) / 8;
  a.rpb = 256 / a.cpr;
  a.tiles = (out_hw + a.rpb - 1) / a.rpb;
  a.scale_h = (float)height / (float)out_hw;
  a.scale_w = (float)width / (float)out_hw;
  avt_clip_sample_table(win_len, a.fast_idx, a.slow_idx);
  const int64_t nblk = (int64_t)a.tiles * 3 * AVT_SLOTS * n_win;
  AVT_REQUIRE(nblk < (1ll << 31), "avt_clip_pack_gather_u8: grid too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (out_dtype == AVT_DT_BF16)
    hipLaunchKernelGGL(clip_pack_gather_kernel<uint16_t>, dim3((unsigned)nblk), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(clip_pack_gather_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, st, a);
  return avt::check_launch("avt_clip_pack_gather_u8");
}
