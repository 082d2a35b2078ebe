// Operand-path microbenchmark: bytes per clock per CU that (a) LDS-DMA (buffer_load_dwordx4 ... lds) and (b) plain 16-byte
// global loads into registers sustain from an L2-resident (or HBM-resident) buffer, 8 waves per CU, no MFMA.
// Build: hipcc -O3 --offload-arch=gfx950 tools/dma_peak.hip -o tools/dma_peak.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool DMA>
__global__ __launch_bounds__(512) void stream_kernel(const char* src, unsigned bytes, int iters, int inflight, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  // each instruction: 16 rows of 64 B (row stride 2 KB), like an operand slab of the conv tiles
  unsigned base = (unsigned)((blockIdx.x * 8 + wid) * 16 * 2048 + (lane >> 2) * 2048 + (lane & 3) * 16);
  i32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned off = (base + (unsigned)(it * 8 + u) * 64u) % (bytes - 4096u);
      if (DMA) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + ((u & 7) * 8 + wid) * 1024), 16,
                                                 (int)(off & ~15u), 0, 0, 0);
      } else {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off & ~15u), 0, 0);
        acc += v;
      }
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678) sink[tid] = 1;
  if (DMA && lds[tid] == 77 && inflight == -1) sink[tid] = 2;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  char* buf;
  int* sink;
  const size_t big = 1ull << 30;
  hipMalloc(&buf, big);
  hipMalloc(&sink, 4096);
  hipMemset(buf, 1, big);
  for (int mode = 0; mode < 2; ++mode)
    for (unsigned bytes : {16u << 20, 1u << 30}) {  // 16 MB: L2 / MALL resident; 1 GB: HBM
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      auto launch = [&](int n) {
        if (mode == 0)
          hipLaunchKernelGGL(stream_kernel<true>, dim3(256), dim3(512), 64 * 1024, 0, buf, bytes, n, 0, sink);
        else
          hipLaunchKernelGGL(stream_kernel<false>, dim3(256), dim3(512), 0, 0, buf, bytes, n, 0, sink);
      };
      launch(50);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch(iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double total = 256.0 * 8 * iters * 8 * 1024;
      printf("%s, %4u MB window: %.3f ms, %.2f TB/s = %.1f B/clk/CU at 2.4 GHz\n", mode == 0 ? "LDS-DMA (buffer_load ... lds)" : "register loads (b128)      ",
             bytes >> 20, ms, total / ms / 1e9, total / ms / 1e6 / 256 / 2.4e3);
    }
  return 0;
}
