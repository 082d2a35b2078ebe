// Shared helpers of the gfx950 kernels behind include/avt.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/avt.h"

namespace avt {

void set_error(const char* fmt, ...);

#define AVT_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      avt::set_error(__VA_ARGS__);        \
      return AVT_ERR_ARG;                 \
    }                                     \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  return AVT_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// fp32 -> bf16 bits, round-to-nearest-even, NaN kept a NaN
// (same arithmetic as oracle/avt_oracle.c f32_to_bf16_rne).
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
  return __uint_as_float((uint32_t)h << 16);
}
// two fp32 -> packed bf16x2 (low half = a) in ONE instruction: v_cvt_pk_bf16_f32, round-to-nearest-even, NaN stays
// NaN — the same rounding as f32_to_bf16_rne (which costs ~6 VALU per value and dominated the HBM-bound epilogues).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16x2_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16x2_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2).  Map them so that every XCD walks a
// CONTIGUOUS range of the work list: neighbours in the list — which share halo rows / frames or operand chunks — then run
// on one XCD at about the same time and meet in its L2 (the stem kernels' HBM fetch fell 3x with this, PMC).
__device__ __forceinline__ int xcd_contiguous(int bid, int nblk) {
  const int xc = bid % 8, qd = nblk / 8, rm = nblk % 8;
  return (xc < rm ? xc * (qd + 1) : rm * (qd + 1) + (xc - rm) * qd) + bid / 8;
}
inline int env_int_flag(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// 64-lane wave reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace avt
