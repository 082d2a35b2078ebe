// bc_fused — the spatial conv and the expanding conv of a SLOW-pathway res2 bottleneck in one kernel:
//     out = relu( c( relu( b(m) ) ) + x ),   b: Conv3d [1,3,3] 64 -> 64,   c: Conv3d [1,1,1] 64 -> 256,
// BatchNorms folded (blocks of the third-party SlowFast model the reference runs per clip window,
// contrastive_video_textures/models/models.py:335, 399; m = the block's a-output, x = its input / shortcut).
//
// Why: as two launches b runs at 1.8 TB/s / 0.5 PFLOP/s (neither roof) and its 64-channel output makes a round
// trip through HBM; fused, the pair moves m once (+ a 2-row halo per 3-row strip), x once and out once — c's own
// traffic — and b's MFMAs ride under it.  HBM-bound by construction (~6,000 MFMA cycles per SIMD against ~21,000
// cycles of HBM time per strip and frame).
//
// A workgroup (7 waves) owns one (clip, strip of HT rows) and walks the frames.  LDS: b's 72 weight fragments
// (72 KB, resident for the whole walk), two buffers of the m strip (HT+2 rows by LDS-DMA, hardware zero fill for the
// rows outside the image; rows are W+1 records apart so one zero record serves as right border of a row and left
// border of the next), a 2 KB scratch tile per wave.  Per frame ONE barrier: wait for the strip's DMA, start the
// next frame's, then every wave takes 16-position tiles: [b] 9 taps x 2 k-steps of v_mfma_f32_16x16x32_bf16 straight
// from the strip at tap-shifted addresses (16-byte chunks XOR-swizzled by position: unpadded 128-byte records,
// conflict-free fragment reads) -> bias, ReLU, bf16 -> the wave's scratch tile; [c] the scratch tile is the operand
// of 16 output tiles (c's 32 weight fragments live in registers), + bias + residual (16-byte global loads) -> ReLU
// -> 16-byte stores.  Weight rows are permuted in the packing so a lane ends with 8 consecutive channels.
#include <stdlib.h>

#include "avt_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFFF0u;
constexpr int NWV = 7;        // waves per workgroup
constexpr int CMID = 64, COUT = 256;
constexpr int NFB = 9 * 2 * 4;  // b's weight fragments: [tap][k half][N-tile]
constexpr int NFC = 16 * 2;     // c's: [N-tile][k half]

struct BCArgs {
  const uint16_t* m;    // [B, T, H, W, 64]
  const uint16_t* res;  // [B, T, H, W, 256]
  uint16_t* out;        // [B, T, H, W, 256]
  const i32x4* wb;      // [NFB][64 lanes]
  const i32x4* wc;      // [NFC][64]
  const float* bb;      // [64]
  const float* bc;      // [256]
  int T, H, strips;
  int ldr, ldo;  // row strides (elements) of res and out: channel slices of wider buffers are allowed
  unsigned m_bytes;
};

template <int W, int HT>
__global__ __launch_bounds__(NWV * 64, 1) void bc_kernel(BCArgs a) {
  constexpr int RS = W + 1;                 // strip row stride in records (shared zero border)
  constexpr int SPOS = (HT + 2) * RS + 1;   // records per strip
  constexpr int SBYTES = SPOS * 128;
  constexpr int IPR = W / 8;                // DMA instructions per strip row
  constexpr int NDMA = (HT + 2) * IPR;
  constexpr int NDW = (NDMA + NWV - 1) / NWV;
  constexpr int PB = HT * W, MTB = (PB + 15) / 16;
  constexpr int TIT = (MTB + NWV - 1) / NWV;  // tiles per wave
  static_assert(W % 8 == 0, "strip rows are staged 8 positions per DMA instruction");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* wbl = lds;                          // [NFB] fragments of 1 KB
  char* st0 = wbl + NFB * 1024;             // [2][SBYTES]
  char* scr = st0 + 2 * SBYTES;             // [NWV][2 KB]
  float* bcl = reinterpret_cast<float*>(scr + NWV * 2048);  // [256] c's bias

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int strip = blockIdx.x % a.strips, b = blockIdx.x / a.strips;
  const int h0 = strip * HT;

  // ---- one-time staging: b's weights and c's bias to LDS, c's weights to registers, zero strips (their borders stay)
  for (int f = wid; f < NFB; f += NWV) *reinterpret_cast<i32x4*>(wbl + f * 1024 + lane * 16) = a.wb[f * 64 + lane];
  for (int i = tid; i < COUT; i += NWV * 64) bcl[i] = a.bc[i];
  for (int i = tid * 16; i < 2 * SBYTES; i += NWV * 64 * 16) *reinterpret_cast<i32x4*>(st0 + i) = i32x4{0, 0, 0, 0};
  bf16x8 wc[NFC];
#pragma unroll
  for (int f = 0; f < NFC; ++f) wc[f] = __builtin_bit_cast(bf16x8, a.wc[f * 64 + lane]);
  float4 bbv[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) bbv[n] = *reinterpret_cast<const float4*>(a.bb + 16 * n + 4 * q);
  __syncthreads();  // the zeroing must not race the first DMA

  // ---- strip DMA: instruction d = (row, j): 8 positions of a row; lane = (position lane>>3, slot lane&7)
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)a.m, 0, a.m_bytes, 0x00020000);
  unsigned poff[NDW];
  int pdst[NDW];
#pragma unroll
  for (int u = 0; u < NDW; ++u) {
    const int d = wid + NWV * u;
    const int r = d / IPR, j = d - r * IPR;
    const int w = 8 * j + (lane >> 3), slot = lane & 7;
    const int sp = r * RS + 1 + w;  // record index of this lane's position in the strip
    const int chunk = slot ^ ((sp >> 1) & 7);
    const int h = h0 - 1 + r;
    const bool ok = d < NDMA && (unsigned)h < (unsigned)a.H;
    poff[u] = ok ? (unsigned)(((h * W + w) * CMID + chunk * 8) * 2) : kOob;
    pdst[u] = (r * RS + 1 + 8 * j) * 128;  // wave-uniform destination of the instruction
  }
  auto dma_strip = [&](int t, char* dst) {
    const unsigned fbase = (unsigned)((b * a.T + t) * a.H) * (unsigned)(W * CMID * 2);
#pragma unroll
    for (int u = 0; u < NDW; ++u) {
      if (wid + NWV * u < NDMA) {
        const unsigned off = poff[u] != kOob ? fbase + poff[u] : kOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rm, (__attribute__((address_space(3))) void*)(dst + __builtin_amdgcn_readfirstlane(pdst[u])), 16, (int)off, 0, 0, 0);
      }
    }
  };

  // ---- per-tile addresses (frame-independent)
  int sbase[TIT], gpos[TIT];  // strip record index of tap (0,0); position index inside a frame
  unsigned tok = 0;
#pragma unroll
  for (int it = 0; it < TIT; ++it) {
    const int mt = wid + NWV * it;
    const int p = mt * 16 + l15;
    const int pc = p < PB ? p : PB - 1;
    const int r = pc / W, w = pc - r * W;
    sbase[it] = r * RS + w;  // tap (dh, dw) -> + dh*RS + dw  (record 0 of a row = the zero border)
    gpos[it] = (h0 + r) * W + w;
    tok |= ((mt < MTB && p < PB && h0 + r < a.H) ? 1u : 0u) << it;
  }
  char* myscr = scr + wid * 2048;
  // b's output tile in the wave's scratch: position l15, 128-byte records, chunk XOR-swizzled by the position
  const int sst = l15 * 128 + (q & 1) * 8;   // + ((2*nt + (q >> 1)) ^ (l15 & 7)) * 16
  const int sld = l15 * 128;                 // + ((4*kk + q) ^ (l15 & 7)) * 16

  dma_strip(0, st0);
  for (int t = 0; t < a.T; ++t) {
    char* cur = st0 + (t & 1) * SBYTES;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // frame t's strip is complete; every wave is done with frame t-1's (the other buffer)
    if (t + 1 < a.T) dma_strip(t + 1, st0 + ((t + 1) & 1) * SBYTES);
    const int64_t fp = (int64_t)((b * a.T + t) * a.H) * W;  // first position of the frame
#pragma unroll
    for (int it = 0; it < TIT; ++it) {
      if (wid + NWV * it < MTB) {  // wave-uniform
        // ---- [b] 3x3 conv of this tile: 9 taps x 2 k-steps, 4 N-tiles
        f32x4 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int sp = sbase[it] + (tap / 3) * RS + tap % 3;
          const char* rec = cur + sp * 128;
          const int sw = (sp >> 1) & 7;
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(rec + (((kh * 4 + q) ^ sw) * 16));
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wbl + ((tap * 2 + kh) * 4 + n) * 1024 + lane * 16);
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[n], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          uint2 pk;
          pk.x = avt::pack_bf16x2(fmaxf(acc[n][0] + bbv[n].x, 0.f), fmaxf(acc[n][1] + bbv[n].y, 0.f));
          pk.y = avt::pack_bf16x2(fmaxf(acc[n][2] + bbv[n].z, 0.f), fmaxf(acc[n][3] + bbv[n].w, 0.f));
          *reinterpret_cast<uint2*>(myscr + sst + (((2 * n + (q >> 1)) ^ (l15 & 7)) * 16)) = pk;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-local hand-off through the scratch tile
        // ---- [c] 64 -> 256 on the tile, + bias + residual -> relu -> global
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(myscr + sld + ((q ^ (l15 & 7)) * 16));
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(myscr + sld + (((4 + q) ^ (l15 & 7)) * 16));
        const bool ok = (tok >> it) & 1u;
#pragma unroll
        for (int np = 0; np < 8; ++np) {
          f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[(2 * np) * 2], a0, c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[(2 * np) * 2 + 1], a1, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[(2 * np + 1) * 2], a0, c1, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[(2 * np + 1) * 2 + 1], a1, c1, 0, 0, 0);
          const float4 b0 = *reinterpret_cast<const float4*>(bcl + 32 * np + 8 * q);
          const float4 b1 = *reinterpret_cast<const float4*>(bcl + 32 * np + 8 * q + 4);
          uint4 rs = make_uint4(0u, 0u, 0u, 0u);
          if (ok) rs = *reinterpret_cast<const uint4*>(a.res + (fp + gpos[it]) * a.ldr + 8 * q + 32 * np);
          uint4 o;
          o.x = avt::pack_bf16x2(fmaxf(c0[0] + b0.x + avt::bf16x2_lo(rs.x), 0.f), fmaxf(c0[1] + b0.y + avt::bf16x2_hi(rs.x), 0.f));
          o.y = avt::pack_bf16x2(fmaxf(c0[2] + b0.z + avt::bf16x2_lo(rs.y), 0.f), fmaxf(c0[3] + b0.w + avt::bf16x2_hi(rs.y), 0.f));
          o.z = avt::pack_bf16x2(fmaxf(c1[0] + b1.x + avt::bf16x2_lo(rs.z), 0.f), fmaxf(c1[1] + b1.y + avt::bf16x2_hi(rs.z), 0.f));
          o.w = avt::pack_bf16x2(fmaxf(c1[2] + b1.z + avt::bf16x2_lo(rs.w), 0.f), fmaxf(c1[3] + b1.w + avt::bf16x2_hi(rs.w), 0.f));
          if (ok) *reinterpret_cast<uint4*>(a.out + (fp + gpos[it]) * a.ldo + 8 * q + 32 * np) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the scratch reads are done before the next tile overwrites it
      }
    }
  }
}

template <int W, int HT>
int launch(BCArgs& a, int batch, int h, hipStream_t st) {
  constexpr int lds_bytes = NFB * 1024 + 2 * (((HT + 2) * (W + 1) + 1) * 128) + NWV * 2048 + COUT * 4;
  static_assert(lds_bytes <= 160 * 1024, "strip does not fit the LDS");
  a.strips = (h + HT - 1) / HT;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bc_kernel<W, HT>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    avt::set_error("avt_bc_fused_bf16: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((bc_kernel<W, HT>), dim3((unsigned)(batch * a.strips)), dim3(NWV * 64), lds_bytes, st, a);
  return avt::check_launch("avt_bc_fused_bf16");
}

}  // namespace

extern "C" int avt_bc_fused_supported(int cm, int c, int w) { return (cm == 64 && c == 256 && (w == 56 || w == 16)) ? 1 : 0; }

extern "C" int avt_bc_fused_bf16(const void* m, const void* res, void* out, const void* wb, const float* bb, const void* wc,
                                 const float* bc, int batch, int t, int h, int w, int cm, int c, int ldr, int ldo,
                                 void* stream) {
  AVT_REQUIRE(m && res && out && wb && bb && wc && bc, "avt_bc_fused_bf16: NULL pointer");
  AVT_REQUIRE(batch > 0 && t > 0 && h > 0 && ldr >= c && ldo >= c && ldr % 8 == 0 && ldo % 8 == 0, "avt_bc_fused_bf16: bad sizes");
  AVT_REQUIRE(avt_bc_fused_supported(cm, c, w), "avt_bc_fused_bf16: unsupported shape Cm=%d C=%d W=%d (slow res2: 64, 256, 56)",
              cm, c, w);
  AVT_REQUIRE(avt::aligned16(m) && avt::aligned16(res) && avt::aligned16(out) && avt::aligned16(wb) && avt::aligned16(wc) &&
                  avt::aligned16(bb) && avt::aligned16(bc),
              "avt_bc_fused_bf16: pointers must be 16-byte aligned");
  const int64_t mb = (int64_t)batch * t * h * w * cm * 2;
  AVT_REQUIRE(mb < (1ll << 32) - 64, "avt_bc_fused_bf16: tensor too large for 32-bit offsets");
  BCArgs a;
  a.m = static_cast<const uint16_t*>(m);
  a.res = static_cast<const uint16_t*>(res);
  a.out = static_cast<uint16_t*>(out);
  a.wb = static_cast<const i32x4*>(wb);
  a.wc = static_cast<const i32x4*>(wc);
  a.bb = bb;
  a.bc = bc;
  a.T = t;
  a.H = h;
  a.ldr = ldr;
  a.ldo = ldo;
  a.m_bytes = (unsigned)mb;
  hipStream_t s = static_cast<hipStream_t>(stream);
  static const int ht = []() {
    const char* e = getenv("AVT_BC_HT");
    return e ? atoi(e) : 2;
  }();
  if (w == 56) return ht == 3 ? launch<56, 3>(a, batch, h, s) : launch<56, 2>(a, batch, h, s);
  return launch<16, 3>(a, batch, h, s);
}
