// conv33_c64 — Conv3d [1,3,3] 64 -> 64 (+ BN + ReLU) of the SLOW-pathway res2 blocks with the input strip resident
// in LDS (blocks of the third-party SlowFast model the reference runs per clip window,
// contrastive_video_textures/models/models.py:335, 399).
//
// Why: as an implicit GEMM this layer sits at 0.45-0.5 PFLOP/s and 1.8 TB/s — neither roof — whatever the tile
// (256x64 register-staged and 512x64 LDS-DMA measure the same, profiles/r01/probe_xls_layers.log): nine gathers of the
// same input per output and a K loop of 9 short steps.  Here a workgroup (14 waves, one 16-position tile each) owns a
// strip of 4 rows of one clip and walks its frames: the 72 weight fragments stay in LDS for the whole walk, the input
// strip (6 rows, LDS-DMA with hardware zero fill, double-buffered: frame t+1 lands under frame t's MFMAs) is read at
// tap-shifted addresses — every input byte crosses HBM / L2 once (+ the 2-row halo) — and the K loop is 18 MFMA
// k-steps on register/LDS operands.  HBM-bound by construction.
#include <stdlib.h>

#include "avt_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFFF0u;
constexpr int CCH = 64;
constexpr int NFB = 9 * 2 * 4;  // weight fragments: [tap][k half][N-tile]

struct C33Args {
  const uint16_t* in;   // [B, T, H, W, 64]
  uint16_t* out;        // [B, T, H, W, ldo]
  const i32x4* wb;      // [NFB][64 lanes]
  const float* bias;    // [64]
  int T, H, strips, ldo, relu;
  int swz;  // XCD-contiguous work order (always on)
  unsigned in_bytes;
};

template <int W, int HT>
__global__ __launch_bounds__(((HT * W + 15) / 16) * 64, 1) void c33_kernel(C33Args a) {
  constexpr int RS = W + 1;                 // strip row stride in records (shared zero border)
  constexpr int SPOS = (HT + 2) * RS + 1;   // records per strip
  constexpr int SBYTES = SPOS * 128;
  constexpr int IPR = W / 8;                // DMA instructions per strip row
  constexpr int NDMA = (HT + 2) * IPR;
  constexpr int PB = HT * W, NWV = (PB + 15) / 16;  // one 16-position tile per wave
  constexpr int NDW = (NDMA + NWV - 1) / NWV;
  static_assert(W % 8 == 0 && NWV <= 16, "strip rows are staged 8 positions per DMA instruction; <= 16 waves");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* wbl = lds;               // [NFB] fragments of 1 KB
  char* st0 = wbl + NFB * 1024;  // [2][SBYTES]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int bid = a.swz ? avt::xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int strip = bid % a.strips, b = bid / a.strips;
  const int h0 = strip * HT;

  for (int f = wid; f < NFB; f += NWV) *reinterpret_cast<i32x4*>(wbl + f * 1024 + lane * 16) = a.wb[f * 64 + lane];
  for (int i = tid * 16; i < 2 * SBYTES; i += NWV * 64 * 16) *reinterpret_cast<i32x4*>(st0 + i) = i32x4{0, 0, 0, 0};
  // bias of this lane's 8 consecutive channels per N-tile pair (rows are permuted in the packing, see include/avt.h)
  float4 bv[2][2];
#pragma unroll
  for (int np = 0; np < 2; ++np) {
    bv[np][0] = *reinterpret_cast<const float4*>(a.bias + 32 * np + 8 * q);
    bv[np][1] = *reinterpret_cast<const float4*>(a.bias + 32 * np + 8 * q + 4);
  }
  __syncthreads();  // the zeroing must not race the first DMA

  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  unsigned poff[NDW];
  int pdst[NDW];
#pragma unroll
  for (int u = 0; u < NDW; ++u) {
    const int d = wid + NWV * u;
    const int r = d / IPR, j = d - r * IPR;
    const int w = 8 * j + (lane >> 3), slot = lane & 7;
    const int sp = r * RS + 1 + w;
    const int chunk = slot ^ ((sp >> 1) & 7);
    const int h = h0 - 1 + r;
    const bool ok = d < NDMA && (unsigned)h < (unsigned)a.H;
    poff[u] = ok ? (unsigned)(((h * W + w) * CCH + chunk * 8) * 2) : kOob;
    pdst[u] = (r * RS + 1 + 8 * j) * 128;
  }
  auto dma_strip = [&](int t, char* dst) {
    const unsigned fbase = (unsigned)((b * a.T + t) * a.H) * (unsigned)(W * CCH * 2);
#pragma unroll
    for (int u = 0; u < NDW; ++u) {
      if (wid + NWV * u < NDMA) {
        const unsigned off = poff[u] != kOob ? fbase + poff[u] : kOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rm, (__attribute__((address_space(3))) void*)(dst + __builtin_amdgcn_readfirstlane(pdst[u])), 16, (int)off, 0, 0, 0);
      }
    }
  };

  // this wave's tile
  const int p = wid * 16 + l15;
  const int pc = p < PB ? p : PB - 1;
  const int r = pc / W, w = pc - r * W;
  const int sbase = r * RS + w;  // strip record of tap (0,0): tap (dh, dw) -> + dh*RS + dw
  const int gpos = (h0 + r) * W + w;
  const bool ok = p < PB && h0 + r < a.H;
  const bool any_ok = __ballot(ok) != 0ull;  // wave-uniform
  int toff[9];  // byte offset of this lane's first chunk (k half 0) per tap, swizzle applied
  int tsw[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int sp = sbase + (tap / 3) * RS + tap % 3;
    toff[tap] = sp * 128;
    tsw[tap] = (sp >> 1) & 7;
  }

  dma_strip(0, st0);
  for (int t = 0; t < a.T; ++t) {
    const char* cur = st0 + (t & 1) * SBYTES;
    // the strip's DMA was issued before the previous frame's 2 output stores: wait for it, not for them
    if (t == 0 || !any_ok)  // (a wave whose tile is entirely outside the image issues no stores to count)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __syncthreads();  // frame t's strip is complete; every wave is done with frame t-1's (the other buffer)
    if (t + 1 < a.T) dma_strip(t + 1, st0 + ((t + 1) & 1) * SBYTES);
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(cur + toff[tap] + (((kh * 4 + q) ^ tsw[tap]) * 16));
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wbl + ((tap * 2 + kh) * 4 + n) * 1024 + lane * 16);
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[n], 0, 0, 0);
        }
      }
    }
    uint16_t* orow = a.out + ((int64_t)((b * a.T + t) * a.H) * W + gpos) * a.ldo + 8 * q;
#pragma unroll
    for (int np = 0; np < 2; ++np) {  // tiles 2np, 2np+1 -> channels 32np + 8q .. +7
      float v[8] = {acc[2 * np][0] + bv[np][0].x,     acc[2 * np][1] + bv[np][0].y,     acc[2 * np][2] + bv[np][0].z,
                    acc[2 * np][3] + bv[np][0].w,     acc[2 * np + 1][0] + bv[np][1].x, acc[2 * np + 1][1] + bv[np][1].y,
                    acc[2 * np + 1][2] + bv[np][1].z, acc[2 * np + 1][3] + bv[np][1].w};
      if (a.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
      }
      uint4 o;
      o.x = avt::pack_bf16x2(v[0], v[1]);
      o.y = avt::pack_bf16x2(v[2], v[3]);
      o.z = avt::pack_bf16x2(v[4], v[5]);
      o.w = avt::pack_bf16x2(v[6], v[7]);
      if (ok) *reinterpret_cast<uint4*>(orow + 32 * np) = o;
    }
  }
}

template <int W, int HT>
int launch(C33Args& a, int batch, int h, hipStream_t st) {
  constexpr int NWV = (HT * W + 15) / 16;
  constexpr int lds_bytes = NFB * 1024 + 2 * (((HT + 2) * (W + 1) + 1) * 128);
  static_assert(lds_bytes <= 160 * 1024, "strip does not fit the LDS");
  a.strips = (h + HT - 1) / HT;
  constexpr int swz = 1;  // XCD-contiguous work order
  a.swz = swz;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c33_kernel<W, HT>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("avt_conv33_c64_bf16: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((c33_kernel<W, HT>), dim3((unsigned)(batch * a.strips)), dim3(NWV * 64), lds_bytes, st, a);
  return avt::check_launch("avt_conv33_c64_bf16");
}

}  // namespace

extern "C" int avt_conv33_c64_supported(int cin, int cout, int w) { return (cin == 64 && cout == 64 && (w == 56 || w == 16)) ? 1 : 0; }

extern "C" int avt_conv33_c64_bf16(const void* in, const void* wb, const float* bias, void* out, int batch, int t, int h, int w,
                                   int ldo, int relu, void* stream) {
  AVT_REQUIRE(in && wb && bias && out, "avt_conv33_c64_bf16: NULL pointer");
  AVT_REQUIRE(batch > 0 && t > 0 && h > 0 && ldo >= 64 && ldo % 8 == 0, "avt_conv33_c64_bf16: bad sizes");
  AVT_REQUIRE(avt_conv33_c64_supported(64, 64, w), "avt_conv33_c64_bf16: unsupported width %d (56)", w);
  AVT_REQUIRE(avt::aligned16(in) && avt::aligned16(wb) && avt::aligned16(bias) && avt::aligned16(out),
              "avt_conv33_c64_bf16: pointers must be 16-byte aligned");
  const int64_t ib = (int64_t)batch * t * h * w * 64 * 2;
  AVT_REQUIRE(ib < (1ll << 32) - 64, "avt_conv33_c64_bf16: tensor too large for 32-bit offsets");
  C33Args a;
  a.in = static_cast<const uint16_t*>(in);
  a.out = static_cast<uint16_t*>(out);
  a.wb = static_cast<const i32x4*>(wb);
  a.bias = bias;
  a.T = t;
  a.H = h;
  a.ldo = ldo;
  a.relu = relu;
  a.in_bytes = (unsigned)ib;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return w == 56 ? launch<56, 4>(a, batch, h, s) : launch<16, 3>(a, batch, h, s);
}
