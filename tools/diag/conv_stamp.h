// Diagnostic hooks of csrc/conv_igemm.hip — compiled ONLY into libavt_hip_stamp.so (`make -C audio-video-textures_amd/csrc stamp`),
// never into the shipped libavt_hip.so.  Cycles per K-loop segment, summed over workgroups (wave 0, lane 0), read back
// through avt_debug_stamps(); tools/probe_stamps.sh prints them.  -DAVT_CONV_STAMP_FINE adds the sub-phase stamps.
#pragma once
__device__ unsigned long long g_stamp[10];  // [0..6] segments, [7] workgroups, [8] s_memtime ticks, [9] s_memrealtime ticks (100 MHz)
#define STAMP_BEGIN()                                               \
  unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0};            \
  unsigned long long last_ = __builtin_amdgcn_s_memtime();          \
  const unsigned long long t0_ = last_, r0_ = __builtin_amdgcn_s_memrealtime()
#ifdef AVT_STAMP_OFF  // the phase-skip diagnostic (tools/probe_conv_phases.sh) wants the hooks' build without their cost
#define STAMP(i) (void)seg_, (void)last_
#else
#define STAMP(i)                                                          \
  do {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                    \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
    __builtin_amdgcn_sched_barrier(0);                                    \
    seg_[i] += now_ - last_;                                              \
    last_ = now_;                                                         \
  } while (0)
#endif
#ifdef AVT_CONV_STAMP_FINE
#define STAMP_FINE(i) STAMP(i)
#else
#define STAMP_FINE(i)
#endif
#ifdef AVT_STAMP_OFF
#define STAMP_END() (void)t0_, (void)r0_
#else
#define STAMP_END()                                                          \
  do {                                                                       \
    STAMP(6); /* epilogue */                                                 \
    if (threadIdx.x == 0) {                                                  \
      for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&g_stamp[i_], seg_[i_]);      \
      atomicAdd(&g_stamp[7], 1ull);                                          \
      atomicAdd(&g_stamp[8], __builtin_amdgcn_s_memtime() - t0_);            \
      atomicAdd(&g_stamp[9], __builtin_amdgcn_s_memrealtime() - r0_);        \
    }                                                                        \
  } while (0)
#endif

#ifndef AVT_STAMP_FN
#define AVT_STAMP_FN avt_debug_stamps
#endif
extern "C" int AVT_STAMP_FN(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 10);
  if (reset) {
    unsigned long long z[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z));
  }
  return 0;
}
