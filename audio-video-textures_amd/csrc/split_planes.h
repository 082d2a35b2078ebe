// Split-plane ("x3") number format shared by the contract-grade encoder kernels (conv_x3.hip, pool.hip, clip_pack.hip):
// an fp32 value travels as TWO 16-bit planes, x = hi + lo.
#pragma once
#include "avt_common.h"

namespace avt {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// fp32 pair -> packed (hi, lo) planes, both round-to-nearest-even; x - hi is exact in fp32.
// F16 = false: bf16 planes (8 + 8 significant bits, |x - hi - lo| <= 2^-16 |x| at every magnitude; 2^-17 observed).
// F16 = true : fp16 planes (11 + 11 bits, <= 2^-22 |x| while lo stays a normal fp16, i.e. |x| >= 2^-3; below that the
//              ABSOLUTE error is <= 2^-24) — FINITE values are clamped to the fp16 range first (65504); a NaN stays a NaN
//              and an infinity stays that infinity (hi = +-inf, lo = 0), so a poisoned or diverged activation still
//              reaches the embedding / the loss instead of turning into a plausible finite number (fminf / fmaxf alone
//              would map a NaN to -65504).
template <bool F16>
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  if constexpr (F16) {
    const bool fin0 = __builtin_fabsf(x0) <= 3.402823466e38f, fin1 = __builtin_fabsf(x1) <= 3.402823466e38f;  // false: inf, NaN
    const f32x2 v = {fin0 ? fminf(fmaxf(x0, -65504.0f), 65504.0f) : x0, fin1 ? fminf(fmaxf(x1, -65504.0f), 65504.0f) : x1};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f32x2 r = {fin0 ? v.x - hf.x : 0.0f, fin1 ? v.y - hf.y : 0.0f};
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
  } else {
    hi = avt::pack_bf16x2(x0, x1);
    lo = avt::pack_bf16x2(x0 - avt::bf16x2_lo(hi), x1 - avt::bf16x2_hi(hi));
  }
}
// Four / two pairs at once (an epilogue's 16-byte / 8-byte store per plane).  fp16 planes: ONE wave-wide test — is any of the values
// outside the fp16 range or not a number? — decides between split2's careful form and the plain one (convert, convert back,
// subtract, convert: 6 VALU operations per pair instead of ~18).  For values inside the range the clamps and selects of split2 are
// identities, so the planes are the same bits; the test is a v_cmp per value whose result is already a wave mask (ORed in the scalar
// unit) and the branch is uniform.  The x3 epilogues are VALU-bound (profiles/r04/xl_epilogue_row_slabs_ab.log): this is ~45 % of
// their instructions.  Used where it measured faster (both conv_x3 tiles, the identity blocks of bneck_x3, the stems' max-pool:
// profiles/r04/split8_ab.log); pw_x3 / conv33_x3 / the stems were equal or slower with it and keep split2.
template <bool F16, int NP>
__device__ __forceinline__ void split_pairs(const float* x, uint32_t* hi, uint32_t* lo) {
  if constexpr (F16) {
    unsigned long long odd = 0ull;
#pragma unroll
    for (int e = 0; e < 2 * NP; ++e) odd |= __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(x[e]) <= 65504.0f));  // (a NaN is "odd" too)
    if (odd == 0ull) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const f32x2 v = {x[2 * p], x[2 * p + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        const f32x2 hf = __builtin_convertvector(h, f32x2);
        const f32x2 r = {v.x - hf.x, v.y - hf.y};
        hi[p] = __builtin_bit_cast(uint32_t, h);
        lo[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
      }
      return;
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) split2<F16>(x[2 * p], x[2 * p + 1], hi[p], lo[p]);
}
template <bool F16>
__device__ __forceinline__ void split8(const float* x, uint4& hi, uint4& lo) {
  split_pairs<F16, 4>(x, reinterpret_cast<uint32_t*>(&hi), reinterpret_cast<uint32_t*>(&lo));
}
template <bool F16>
__device__ __forceinline__ void split4(const float* x, uint2& hi, uint2& lo) {
  split_pairs<F16, 2>(x, reinterpret_cast<uint32_t*>(&hi), reinterpret_cast<uint32_t*>(&lo));
}
// ReLU that keeps a NaN a NaN, like torch.relu (fmaxf(NaN, 0) = 0 under IEEE maxNum would turn a poisoned activation into a
// plausible zero; an infinity times a weight's low plane of either sign arrives here as a NaN too)
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.0f ? 0.0f : v; }
// packed (hi, lo) planes -> the fp32 pair they stand for
template <bool F16>
__device__ __forceinline__ f32x2 join2(uint32_t hi, uint32_t lo) {
  if constexpr (F16) {
    const f32x2 h = __builtin_convertvector(__builtin_bit_cast(f16x2, hi), f32x2);
    const f32x2 l = __builtin_convertvector(__builtin_bit_cast(f16x2, lo), f32x2);
    return f32x2{h.x + l.x, h.y + l.y};
  } else {
    return f32x2{avt::bf16x2_lo(hi) + avt::bf16x2_lo(lo), avt::bf16x2_hi(hi) + avt::bf16x2_hi(lo)};
  }
}

// join2 for fp16 planes as ONE mixed-precision FMA per value: fp32(hi half) * 1.0 + fp32(lo half), rounded once — the bits of the two
// conversions and the fp32 add of join2 (both addends are exact in fp32), 2 VALU operations per pair instead of 5.  Used where it
// measured faster (bneck_x3's 14-wide identity block: -10 %); the max-pool was 3 % slower with it, the other kernels equal
// (profiles/r04/split8_ab.log)
__device__ __forceinline__ f32x2 join2_mix_f16(uint32_t hi, uint32_t lo) {
  f32x2 r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r.x) : "v"(hi), "v"(lo));
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r.y) : "v"(hi), "v"(lo));
  return r;
}

}  // namespace avt
