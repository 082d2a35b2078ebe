// What slows the matrix pipes down when a long-K tile is resident?  8 waves per CU (two per SIMD), each iteration = 16
// independent v_mfma_f32_32x32x16_bf16 (random operands) plus, per mode: (1) nothing, (2) 12 ds_read_b128 into a rotating
// register set (the XL tile's fragment reads), (3) 4 LDS-DMA instructions from an L2-resident window (its operand
// stream), (4) both.  Reports TFLOP/s per mode.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_mix.hip -o tools/mfma_mix.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NRD, int NDMA, int NGL, bool S16 = false>
__global__ __launch_bounds__(512, 2) void mix_kernel(const unsigned* seed, const char* src, unsigned bytes, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned s0 = seed[(tid + blockIdx.x * 512) & 1023];
  union { bf16x8 v; unsigned u[4]; i32x4 i; } a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      s0 = s0 * 1664525u + 1013904223u;
      a[i].u[j] = (s0 & 0x807f807fu) | 0x3f003f00u;
      s0 = s0 * 1664525u + 1013904223u;
      b[i].u[j] = (s0 & 0x807f807fu) | 0x3f003f00u;
    }
  for (int i = tid; i < 32768; i += 512) reinterpret_cast<unsigned*>(lds)[i] = (s0 + i) & 0x3f7f3f7fu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  unsigned base = (unsigned)((blockIdx.x * 8 + wid) * 16 * 2048 + (lane >> 2) * 2048 + (lane & 3) * 16);
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  i32x4 sink = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    i32x4 fr[NRD > 0 ? NRD : 1], gl[NGL > 0 ? NGL : 1];
#pragma unroll
    for (int u = 0; u < NRD; ++u) fr[u] = *reinterpret_cast<const i32x4*>(lds + ((it * NRD + u) & 63) * 1024 + lane * 16);
#pragma unroll
    for (int u = 0; u < NDMA; ++u) {
      const unsigned off = (base + (unsigned)(it * NDMA + u) * 64u) % (bytes - 4096u);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 65536 + ((u & 3) * 8 + wid) * 1024), 16,
                                               (int)(off & ~15u), 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NGL; ++u)  // packed fragments: 1 KB contiguous per wave instruction
      gl[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((((unsigned)(it * NGL + u) * 8u + wid) * 1024u + lane * 16u) % (bytes - 4096u)), 0, 0);
    if (S16) {  // the same FLOPs as 32 MFMAs of the 16x16x32 shape (16 independent 16x16 accumulators)
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        f32x4 t = {acc[i & 7][(i >> 3) * 4 + 0], acc[i & 7][(i >> 3) * 4 + 1], acc[i & 7][(i >> 3) * 4 + 2], acc[i & 7][(i >> 3) * 4 + 3]};
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3].v, b[(i >> 2) & 3].v, t, 0, 0, 0);
        acc[i & 7][(i >> 3) * 4 + 0] = t[0];
        acc[i & 7][(i >> 3) * 4 + 1] = t[1];
        acc[i & 7][(i >> 3) * 4 + 2] = t[2];
        acc[i & 7][(i >> 3) * 4 + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3].v, b[(i >> 2) & 3].v, acc[i & 7], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NRD; ++u) sink ^= fr[u];
#pragma unroll
    for (int u = 0; u < NGL; ++u) sink ^= gl[u];
    if (NDMA > 0 && NGL == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  float sum = (float)(sink[0] ^ sink[1] ^ sink[2] ^ sink[3]);
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) sum += acc[i][r];
  if (sum == 123.456f) out[lane] = sum;
}

template <int NRD, int NDMA, int NGL, bool S16 = false>
void run(const unsigned* seed, const char* buf, float* out, int iters, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<NRD, NDMA, NGL, S16>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipLaunchKernelGGL((mix_kernel<NRD, NDMA, NGL, S16>), dim3(256), dim3(512), 128 * 1024, 0, seed, buf, 16u << 20, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((mix_kernel<NRD, NDMA, NGL, S16>), dim3(256), dim3(512), 128 * 1024, 0, seed, buf, 16u << 20, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 8 * iters * 16 * 2.0 * 32 * 32 * 16;
  printf("%-58s %.2f ms -> %.0f TFLOP/s\n", what, ms, flops / ms / 1e9);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned* seed;
  char* buf;
  float* out;
  hipMalloc(&seed, 4096);
  hipMalloc(&out, 4096);
  hipMalloc(&buf, 64 << 20);
  hipMemset(buf, 1, 64 << 20);
  unsigned h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 12345u + 7919u * i;
  hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
  run<0, 0, 0>(seed, buf, out, iters, "16 MFMA per iteration, 2 waves per SIMD");
  run<12, 0, 0>(seed, buf, out, iters, "  + 12 ds_read_b128 per wave and iteration");
  run<0, 4, 0>(seed, buf, out, iters, "  + 4 LDS-DMA instructions (1 KB each) per wave and iteration");
  run<12, 4, 0>(seed, buf, out, iters, "  + both (the XL tile's mix)");
  run<8, 2, 4>(seed, buf, out, iters, "  8 ds_read + 2 LDS-DMA + 4 packed global loads (B from registers)");
  run<8, 2, 0>(seed, buf, out, iters, "  8 ds_read + 2 LDS-DMA");
  run<6, 4, 0>(seed, buf, out, iters, "  6 ds_read + 4 LDS-DMA (128x128 wave tiles)");
  run<0, 0, 0, true>(seed, buf, out, iters, "32 MFMA 16x16x32 per iteration (same FLOPs), 2 waves per SIMD");
  run<12, 4, 0, true>(seed, buf, out, iters, "  + 12 ds_read + 4 LDS-DMA (the XL mix on the 16x16x32 shape)");
  run<8, 2, 0, true>(seed, buf, out, iters, "  + 8 ds_read + 2 LDS-DMA (the XB mix without its register loads) on 16x16x32");
  run<8, 2, 4, true>(seed, buf, out, iters, "  + 8 ds_read + 2 LDS-DMA + 4 packed loads on 16x16x32");
  return 0;
}
