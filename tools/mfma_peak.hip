// MFMA-only microbenchmark: what the matrix pipes sustain with no memory traffic at all (power / clock behaviour included).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o gpurun_out/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_loop(const unsigned* seed, float* out, int iters, int waves_mask, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned s0 = seed[(threadIdx.x + blockIdx.x * 256) & 1023];
  union { bf16x8 v; unsigned u[4]; } a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      s0 = s0 * 1664525u + 1013904223u;
      a[i].u[j] = seed[0] ? ((s0 & 0x807f807fu) | 0x3f003f00u) : 0u;  // random mantissas around +-0.5..1, or all zeros
      s0 = s0 * 1664525u + 1013904223u;
      b[i].u[j] = seed[0] ? ((s0 & 0x807f807fu) | 0x3f003f00u) : 0u;
    }
  f32x16 acc[16];
  for (int i = 0; i < 16; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc[i * 4 + j], 0, 0, 0);
  }
  float sum = 0.f;
  for (int i = 0; i < 16; ++i)
    for (int r = 0; r < 16; ++r) sum += acc[i][r];
  if (sum == 123.456f) out[lane] = sum;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = __builtin_amdgcn_s_memtime() - t0;      // shader-clock ticks?
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;  // constant 100 MHz
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned* seed;
  float* out;
  hipMalloc(&seed, 4096);
  hipMalloc(&out, 4096);
  unsigned long long* clk;
  hipMalloc(&clk, 64);
  unsigned h[1024];
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 1024; ++i) h[i] = mode ? 12345u + 7919u * i : 0u;
    hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
    for (int wgs = 256; wgs <= 512; wgs *= 2) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, seed, out, 200, 0, clk);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, seed, out, iters, 0, clk);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)wgs * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
      unsigned long long hc[2];
      hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
      printf("%s operands, %d workgroups x 4 waves, %d x 16 MFMA 32x32x16 bf16 per wave: %.2f ms -> %.0f TFLOP/s; s_memtime %.1f ticks/iter, "
             "s_memrealtime %.2f ticks/iter -> s_memtime runs at %.0f MHz; 512 MFMA cycles/iter -> core clock %.0f MHz\n",
             mode ? "random" : "zero", wgs, iters, ms, flops / ms / 1e9, (double)hc[0] / iters, (double)hc[1] / iters,
             100.0 * hc[0] / hc[1], 512.0 * iters / (hc[1] / 100.0) / (wgs / 256));
    }
  }
  return 0;
}
