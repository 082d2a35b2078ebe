// negative_sample_mt19937 — the training dataset's negative sampling ON THE DEVICE, stream-for-stream equal to the
// reference's NumPy calls (contrastive_video_textures/dataset/dataset.py:128-139, 181-190):
//     neg = np.random.choice(others, n_negs, replace=False)        # others = arange(len+1) without idx, idx+1
//     hard = [idx-4 .. idx-1, idx+2 .. idx+5] clipped to [0, len];  neg[:len(hard)] = hard
// np.random.choice(a, size, replace=False) of the legacy RandomState is permutation(len(a))[:size]; permutation is a
// Fisher-Yates shuffle of arange walking i = n-1 .. 1 with j = random_interval(i): 32-bit MT19937 draws masked to the
// smallest 2^k - 1 >= i, redrawn while > i (numpy/random/mtrand.pyx, _shuffle_raw; distributions.c, random_interval).
// The MT19937 state (624 words + position) lives in device memory and is advanced exactly as NumPy would advance it, so
// a host that uploads np.random.get_state() and later downloads the state stays in step with the reference's stream
// (fixture G9 pins it).  The generator is sequential: one wave per launch, lane 0 walks the stream, the permutation
// buffer lives in LDS; the other lanes help with the state regeneration ("twist").  ~0.1 ms per item — beside a
// 2-second training step; what matters is that sample -> pack -> encode needs no host round trip.
#include "avt_common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int kMaxPop = 12288;  // permutation entries kept in LDS (48 KB): videos of up to ~12k segments

// regenerate the 624-word state in place (genrand's "twist"), in the three dependency-free phases of the recurrence
__device__ void mt_twist(uint32_t* mt, int lane) {
  auto mix = [](uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
  };
  // phase 1: i in [0, 227): needs old mt[i+1], old mt[i+397]: chunks of 64 are independent (i+1 may be in the next chunk
  // and must be read OLD: read before write across the wave)
  for (int base = 0; base < MT_N - MT_M; base += 64) {
    const int i = base + lane;
    uint32_t v = 0;
    const bool on = i < MT_N - MT_M;
    if (on) v = mt[i + MT_M] ^ mix(mt[i], mt[i + 1]);
    __builtin_amdgcn_wave_barrier();
    if (on) mt[i] = v;
    __builtin_amdgcn_wave_barrier();
  }
  // phase 2: i in [227, 623): needs old mt[i+1] and NEW mt[i-227]; chunks of 64 <= 227 keep the new values ready
  for (int base = MT_N - MT_M; base < MT_N - 1; base += 64) {
    const int i = base + lane;
    uint32_t v = 0;
    const bool on = i < MT_N - 1;
    if (on) v = mt[i + (MT_M - MT_N)] ^ mix(mt[i], mt[i + 1]);
    __builtin_amdgcn_wave_barrier();
    if (on) mt[i] = v;
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) mt[MT_N - 1] = mt[MT_M - 1] ^ mix(mt[MT_N - 1], mt[0]);
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64) void negative_sample_kernel(uint32_t* state, const int64_t* idx, int batch, int n_len,
                                                             int n_negs, int32_t* neg_out) {
  __shared__ uint32_t mt[MT_N];
  __shared__ int32_t perm[kMaxPop];
  __shared__ int s_pos;
  const int lane = threadIdx.x;
  for (int i = lane; i < MT_N; i += 64) mt[i] = state[i];
  if (lane == 0) s_pos = (int)state[MT_N];
  __syncthreads();
  const int pop = n_len + 1 - 2;  // arange(len + 1) without idx and idx + 1
  for (int b = 0; b < batch; ++b) {
    const int q = (int)idx[b];
    for (int i = lane; i < pop; i += 64) perm[i] = i;
    __syncthreads();
    // Fisher-Yates, i = pop-1 .. 1 (lane 0 draws; every lane joins the twists)
    int i = pop - 1;
    while (i >= 1) {  // uniform control flow: all lanes follow lane 0's progress through LDS
      int pos = s_pos;
      if (pos >= MT_N) {
        mt_twist(mt, lane);
        pos = 0;
      }
      __syncthreads();
      if (lane == 0) {
        // consume draws until the state runs out or the shuffle is done
        while (i >= 1 && pos < MT_N) {
          uint32_t mask = (uint32_t)i;
          mask |= mask >> 1;
          mask |= mask >> 2;
          mask |= mask >> 4;
          mask |= mask >> 8;
          mask |= mask >> 16;
          uint32_t y = mt[pos++];
          y ^= y >> 11;
          y ^= (y << 7) & 0x9d2c5680u;
          y ^= (y << 15) & 0xefc60000u;
          y ^= y >> 18;
          const uint32_t v = y & mask;
          if (v <= (uint32_t)i) {  // accepted: swap; rejected draws are simply consumed
            const int32_t t = perm[i];
            perm[i] = perm[v];
            perm[v] = t;
            --i;
          }
        }
        s_pos = pos;
        perm[kMaxPop - 1] = i;  // publish progress (slot unused: pop < kMaxPop)
      }
      __syncthreads();
      i = perm[kMaxPop - 1];
      __syncthreads();
    }
    // neg = others[perm[:n_negs]], then the hard negatives overwrite the head (dataset.py:183-190)
    for (int k = lane; k < n_negs; k += 64) {
      int v = perm[k];  // position among the others (ascending segment ids without q, q+1)
      if (v >= q) v += 2;
      neg_out[(int64_t)b * n_negs + k] = v;
    }
    __syncthreads();
    if (lane == 0) {
      const int hard[8] = {q - 4, q - 3, q - 2, q - 1, q + 2, q + 3, q + 4, q + 5};
      int k = 0;
      for (int h = 0; h < 8; ++h)
        if (hard[h] >= 0 && hard[h] <= n_len) {
          if (k < n_negs) neg_out[(int64_t)b * n_negs + k] = hard[h];
          ++k;
        }
    }
    __syncthreads();
  }
  for (int i = lane; i < MT_N; i += 64) state[i] = mt[i];
  if (lane == 0) state[MT_N] = (uint32_t)s_pos;
}

}  // namespace

extern "C" int avt_negative_sample_mt19937(uint32_t* mt_state, const int64_t* idx, int batch, int n_len, int n_negs,
                                           int32_t* neg_out, void* stream) {
  AVT_REQUIRE(mt_state && idx && neg_out, "avt_negative_sample_mt19937: NULL pointer");
  AVT_REQUIRE(batch >= 0 && n_len >= 3 && n_negs >= 1 && n_negs <= n_len - 1,
              "avt_negative_sample_mt19937: need batch >= 0, len >= 3, 1 <= n_negs <= len - 1 (cannot draw %d of %d)", n_negs,
              n_len - 1);
  AVT_REQUIRE(n_len - 1 < kMaxPop, "avt_negative_sample_mt19937: more than %d candidate segments", kMaxPop - 1);
  if (batch == 0) return AVT_OK;
  hipLaunchKernelGGL(negative_sample_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), mt_state, idx, batch,
                     n_len, n_negs, neg_out);
  return avt::check_launch("avt_negative_sample_mt19937");
}
