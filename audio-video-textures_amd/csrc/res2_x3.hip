// res2_x3 — one whole IDENTITY bottleneck of the SlowFast SLOW pathway's res2 stage in a single launch, contract-grade
// split-plane arithmetic (conv_x3.hip's number format: every tensor is two 16-bit planes x = hi + lo, a product is three MFMA
// passes wl*xh + wh*xl + wh*xh into one fp32 accumulator):
//     out = relu( c( relu( b( relu( a(x) ) ) ) ) + x )      a: 1x1x1 C -> CM,  b: [1,3,3] CM -> CM,  c: 1x1x1 CM -> C,  BN folded
// with C = 256, CM = 64 at 56 x 56 (blocks of the third-party SlowFast model the reference runs per clip window,
// contrastive_video_textures/models/models.py:335, 399).  Round 5 ran such a block as three launches (pw_x3 a, conv33_x3 b,
// pw_x3 / pw_chain_x3 c): 3 KB of HBM traffic per position against 2 KB for x in + out once, at 4.3-5 TB/s — the streaming 40 % of
// the encoder step that VERDICT r5 names.  Why it was not one kernel before: the three weight sets are 272 KB as plane pairs and do
// not fit the LDS beside anything (the fast pathway's bneck_x3 keeps its 30-110 KB resident).
//
// Form: the activations stay put, the WEIGHTS stream.
//   * Positions are FLAT (frame-major, row-major): a workgroup of 8 waves owns a contiguous range of 128-position steps; wave w owns
//     the 16 positions [128 s + 16 w, + 16) of step s in every phase — a 16 x 16 x 32 MFMA column block (pw_x3's operand forms).
//   * Iteration i runs phase A (the a conv) of step i, then phases B and C of step i - 1: A's relu'd, split output lands in an LDS
//     RING of 384 positions x 64 channels x 2 planes (96 KB, 16-byte chunks XOR-swizzled by position), so the 3 x 3 taps of phase
//     B read finished neighbours on both sides (a row + 1 = 57 positions) — no halo recompute.
//   * Phase B (b conv): the taps' operands are ring reads at shifted slots; a neighbour outside the frame (row / column edge, the
//     frame before / after in the flat order) reads a 16-byte ZERO slot instead (address select: the zero padding).
//   * Phase C (c conv + bias + residual + ReLU): phase B's accumulator layout IS the operand layout of the next GEMM (output rows
//     permuted in the packing, pw_chain's trick) — b's output never leaves registers.  The RESIDUAL is the x tile phase A used one
//     iteration earlier, still in registers (the operand chunk of k-step M, lane group q IS the residual of output channels
//     32 M + 8 q ..): x crosses HBM exactly once.  As phase C releases the tile k-step by k-step, the tile of step i + 1 is
//     requested into the freed registers, 3 to 17 chunks before phase A uses it.
//   * The 136 weight-fragment pairs (2 KB each: hi + lo plane of a 16-row x 32-k MFMA operand) arrive in 17 CHUNKS of 8 pairs by
//     LDS-DMA (buffer_load ... lds) into a three-deep 48 KB rotation, two chunks ahead of their use; one workgroup barrier per
//     chunk.  They come from L2 (272 KB per 128 positions, the same every step).
// History of the form (profiles/r06/README.md): v1 (8 waves x 32 positions, residual re-read from L2 / Infinity Cache) 5.15 ms per
// 249-clip launch = the three launches' time; v2 (4 waves x 32 positions, one per SIMD with 448 registers, residual in registers) 4.95 ms
// — its phase-skip builds (R2_DBG) showed 3.3 ms of compute SERIALIZED in the single wave of a SIMD (MFMA, the epilogues' VALU and the
// LDS reads do not overlap without a second wave) and 1.7 ms of memory time exposed; this form keeps two waves per SIMD by halving
// the tile (two x tiles live = 128 registers).
// Counted waits: every wave issues the SAME sequence of vector-memory operations per chunk (loads and stores of out-of-range
// positions carry an out-of-bounds offset), so "chunk c has landed" is s_waitcnt vmcnt(N(c)) with N(c) = the operations issued
// after its DMA pieces (vmcnt retires in order on gfx9-family parts; a scratch spill would only make a wait longer).
// Roofline: HBM (2 * C * 4 B per position).
#include <stdlib.h>

#include <type_traits>

#include "avt_common.h"
#include "split_planes.h"

namespace {

// phase-skip diagnostic (tools/r06_runs/gpu_r06_res2_phases.sh builds one library per -DR2_DBG=mask: 1 no stores, 2 no x loads after
// the preamble, 4 no weight DMA after the preamble, 8 no barriers / counted waits, 16 no MFMAs, 32 plain epilogues (no split));
// the shipped library is built with 0
#ifndef R2_DBG
#define R2_DBG 0
#endif
#define R2_SKIP(bit) (((R2_DBG) & (bit)) != 0)
// cycle stamps (diagnostic builds only: -DR2_STAMP; tools/r06_runs/gpu_r06_res2_stamps.sh): where a wave's cycles go, wave 0 of every
// workgroup, summed: [0] chunk-top waits (own DMA pieces, LDS drain, barrier), [1] phase A chunks, [2] A epilogue, [3] phase B taps,
// [4] B epilogue, [5] phase C MFMAs, [6] phase C epilogues (+ stores, next tile's loads), [7] workgroups, [8] memtime, [9] realtime
#ifdef R2_STAMP
__device__ unsigned long long g_r2_stamp[10];
#define R2_ST_BEGIN()                                        \
  unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0};     \
  unsigned long long last_ = __builtin_amdgcn_s_memtime();   \
  const unsigned long long t0_ = last_, r0_ = __builtin_amdgcn_s_memrealtime()
#define R2_ST(i)                                                  \
  do {                                                            \
    __builtin_amdgcn_sched_barrier(0);                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                            \
    seg_[i] += now_ - last_;                                      \
    last_ = now_;                                                 \
  } while (0)
#define R2_ST_END()                                                             \
  do {                                                                          \
    if (threadIdx.x == 0) {                                                     \
      for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&g_r2_stamp[i_], seg_[i_]);      \
      atomicAdd(&g_r2_stamp[7], 1ull);                                          \
      atomicAdd(&g_r2_stamp[8], __builtin_amdgcn_s_memtime() - t0_);            \
      atomicAdd(&g_r2_stamp[9], __builtin_amdgcn_s_memrealtime() - r0_);        \
    }                                                                           \
  } while (0)
#else
#define R2_ST_BEGIN()
#define R2_ST(i)
#define R2_ST_END()
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFFF0u;

template <bool F16>
__device__ __forceinline__ f32x4 mfma(i32x4 w, i32x4 x, f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}
// one split-plane product, small terms first (conv_x3.hip's order)
template <bool F16>
__device__ __forceinline__ f32x4 mfma3(i32x4 wh, i32x4 wl, i32x4 xh, i32x4 xl, f32x4 c) {
  if (R2_SKIP(16)) {  // diagnostic: the operands stay live (one VALU op each), no matrix instruction
    c[0] += __builtin_bit_cast(float, wh[0] ^ wl[1] ^ xh[2] ^ xl[3]);
    return c;
  }
  c = mfma<F16>(wl, xh, c);
  c = mfma<F16>(wh, xl, c);
  return mfma<F16>(wh, xh, c);
}

constexpr int C = 256, CM = 64;
constexpr int NWV = 8, TP = 16, STEP = NWV * TP, RING = 384;
constexpr int KA = C / 32;                       // k-steps of a (and chunks of 8 channels x 4 lane groups of an x tile)
constexpr int CH_PAIRS = 8, CH_BYTES = CH_PAIRS * 2048, NBUF = 3;
#ifndef R2_AHEAD
#define R2_AHEAD 2  // (2, 3, 4 measured equal; 2 is the deepest without a spilled register)
#endif
constexpr int AHEAD = R2_AHEAD;                  // fragment pairs read ahead of the MFMAs that use them
#ifndef R2_FENCE
#define R2_FENCE 1
#endif
#if R2_FENCE
#define R2_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define R2_SCHED_FENCE()
#endif
constexpr int PIECES = CH_BYTES / 1024 / NWV;    // 1 KB DMA pieces per wave and chunk
constexpr int NCH_A = KA * (CM / 16) / CH_PAIRS; // 4 chunks: 2 k-steps x 4 n-tiles each
constexpr int NCH_B = 9;                         // one tap each: 2 k-steps x 4 n-tiles
constexpr int NCH_C = (C / 16) * 2 / CH_PAIRS;   // 4 chunks: 4 n-tiles x 2 k-steps each
constexpr int NCH = NCH_A + NCH_B + NCH_C;       // 17
constexpr int RING_PLANE = RING * CM * 2;        // 49152 B
constexpr int ZERO_OFF = 2 * RING_PLANE;         // 16 zero bytes (+ padding to 1 KB)
constexpr int WB_OFF = ZERO_OFF + 1024;
constexpr int CF_OFF = WB_OFF + NBUF * CH_BYTES;
constexpr int NCOEF = 4 * CM + 2 * C;
constexpr int LDS_BYTES = CF_OFF + NCOEF * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(NCH_A == 4 && NCH_C == 4 && PIECES == 2, "chunk layout");

// "other" vector-memory operations a wave issues during chunk c (after the chunk-top DMA pieces): every output-channel group of 32 of
// phase C (two per chunk) stores 2 planes and requests one k-step x 2 planes of the x tile after next; phases A and B issue none
constexpr int other_ops(int c) {
  c = ((c % NCH) + NCH) % NCH;
  return c < NCH_A + NCH_B ? 0 : 8;
}
constexpr int wait_count(int c) { return other_ops(c - 2) + PIECES + other_ops(c - 1); }
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool F16>
__device__ __forceinline__ void r2_split(const float* v, uint4& h, uint4& l) {
  if (R2_SKIP(32)) {  // diagnostic: no plane split
    h = uint4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    l = uint4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
    return;
  }
  avt::split8<F16>(v, h, l);
}

struct R2Args {
  const uint16_t* xh;
  const uint16_t* xl;
  uint16_t* oh;
  uint16_t* ol;
  const void* wf;      // [NCH][8 pairs][2 planes][64 lanes][16 B]
  const float* coef;   // [sa CM | ba CM | sb CM | bb CM | sc C | bc C]
  int P, H, HW;        // positions; rows per frame; H * W
  int ldi, ldo;        // row pitch of x / out in elements
  int nsteps, spw;     // 128-position steps; steps per workgroup
  unsigned x_bytes, o_bytes, w_bytes;
};

template <int W, bool F16>
__global__ __launch_bounds__(NWV * 64) void res2_x3_kernel(R2Args a) {
  // iteration i: phase A of step i, then phases B / C of step i - 1.  B / C of step i - 1 read ring slots of positions
  // [STEP (i - 1) - (W + 1), STEP i + W]; phase A of step i + 1 writes [STEP (i + 1), STEP (i + 2)) while B / C of step i still need
  // [STEP i - (W + 1), ...): the live span is 2 STEP + W + 1 positions
  static_assert(W + 1 <= STEP && 2 * STEP + W + 1 < RING, "the ring covers two steps and a halo");
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lp = lane & 15, q = lane >> 4;
  float* cf = reinterpret_cast<float*>(lds + CF_OFF);
  for (int i = tid; i < NCOEF; i += NWV * 64) cf[i] = a.coef[i];
  if (tid < 64) *reinterpret_cast<i32x4*>(lds + ZERO_OFF + tid * 16) = i32x4{0, 0, 0, 0};

  const int s0 = (int)blockIdx.x * a.spw;
  const int s1 = s0 + a.spw < a.nsteps ? s0 + a.spw : a.nsteps;
  if (s0 >= s1) return;

  const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)a.xh, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)a.xl, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t roh = __builtin_amdgcn_make_buffer_rsrc((void*)a.oh, 0, a.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rol = __builtin_amdgcn_make_buffer_rsrc((void*)a.ol, 0, a.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwf = __builtin_amdgcn_make_buffer_rsrc((void*)a.wf, 0, a.w_bytes, 0x00020000);

  // weight chunk `cc` (0 .. NCH-1, the same stream every step) -> rotation buffer `buf`: this wave's 1 KB pieces
  auto dma_chunk = [&](int cc, int buf) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int piece = wid * PIECES + j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rwf, (lds_ptr)(lds + WB_OFF + buf * CH_BYTES + piece * 1024), 16,
                                               cc * CH_BYTES + piece * 1024 + lane * 16, 0, 0, 0);
    }
  };

  // ---- the x tiles: xt[set][k][plane] = this lane's 8 channels 32 k + 8 q .. of its position.  Set (j & 1) holds the tile of local
  // iteration j: phase A's operand there, and — the SAME registers, the same layout — phase C's residual of output channels
  // 32 k + 8 q .. one iteration later; as phase C releases it k-step by k-step, the tile of iteration j + 2 is requested into the
  // freed registers
  i32x4 xt[2][KA][2];

  auto x_off = [&](int p) -> unsigned {  // byte offset of this lane's chunk of k-step 0 of position p
    return (p >= 0 && p < a.P) ? ((unsigned)p * (unsigned)a.ldi + (unsigned)(q * 8)) * 2u : kOob;
  };
  const int p_lane = TP * wid + lp;
  auto load_tile = [&](auto set_c, int k, unsigned base) {
    constexpr int SET = decltype(set_c)::value;
    const int off = (int)((base != kOob) ? base + (unsigned)(k * 64) : kOob);
    xt[SET][k][0] = __builtin_amdgcn_raw_buffer_load_b128(rxh, off, 0, 0);
    xt[SET][k][1] = __builtin_amdgcn_raw_buffer_load_b128(rxl, off, 0, 0);
  };

  // ---- preamble: the tile of iteration 1 (older than everything that is waited for), chunks 0 and 1, then the tile of iteration 0
  // (the order the wait counts assume: 2 K loads stay younger than chunk 0's pieces, like a phase C's operations)
  int gc = 0;  // chunks consumed so far: chunk gc lives in rotation buffer gc % NBUF
  {
    const unsigned b1 = x_off(STEP * s0 + p_lane);
#pragma unroll
    for (int k = 0; k < KA; ++k) load_tile(std::integral_constant<int, 1>{}, k, b1);
  }
  asm volatile("" ::: "memory");
  dma_chunk(0, 0);
  dma_chunk(1, 1);
  asm volatile("" ::: "memory");
  {
    const unsigned b0 = x_off(STEP * (s0 - 1) + p_lane);
#pragma unroll
    for (int k = 0; k < KA; ++k) load_tile(std::integral_constant<int, 0>{}, k, b0);
  }
  asm volatile("" ::: "memory");

  R2_ST_BEGIN();
  // top of chunk CC: its pieces have landed (every wave's: barrier), the buffer of chunk gc - 1 is free -> request chunk gc + 2
  auto chunk_top = [&](auto cc_c, auto seg_c) -> const char* {
    constexpr int CC = decltype(cc_c)::value;
    R2_ST(decltype(seg_c)::value);  // what ran since the last stamp belongs to the caller's segment
    if (!R2_SKIP(8)) {
      wait_vm<wait_count(CC)>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    R2_ST(0);
    asm volatile("" ::: "memory");
    if (!R2_SKIP(4)) dma_chunk((CC + 2) % NCH, (gc + 2) % NBUF);
    asm volatile("" ::: "memory");
    const char* cur = lds + WB_OFF + (gc % NBUF) * CH_BYTES;
    ++gc;
    return cur;
  };
  auto WP = [&](const char* cur, int pair, int plane, int lofs) {
    return *reinterpret_cast<const i32x4*>(cur + (pair * 2 + plane) * 1024 + lofs);
  };

  // this lane's phase-B/C position inside its frame, carried from iteration to iteration (a runtime modulo per step kept its
  // magic number in a spilled register)
  int qf;
  {
    const int pb0 = STEP * (s0 - 2) + p_lane;
    qf = pb0 % a.HW;
    if (qf < 0) qf += a.HW;
  }

  // local iteration j (step i = s0 - 1 + j): phase A of step i on tile set SET = j & 1, phases B / C of step i - 1 with tile set
  // SET ^ 1 as the residual
  int pa_next = STEP * (s0 - 1) + p_lane;  // (carried: recomputing it from the lane index per iteration cost a spilled register)
  auto iteration = [&](auto set_c, int i) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value, OLD = SET ^ 1;
    const bool store_ok = i - 1 >= s0 && i - 1 < s1;
    const int pa = pa_next;                   // phase A's position (before / after the tensor: its ring slot is never read)
    pa_next += STEP;
    const int pb = pa - STEP;                 // phase B / C's position
    const unsigned ra = (unsigned)(pa + 8 * RING) % (unsigned)RING;
    const unsigned rb = (unsigned)(pb + 8 * RING) % (unsigned)RING;
    const bool pb_in = pb >= 0 && pb < a.P;
    const int qq = qf;
    const int y = qq / W, xc = qq - y * W;
    qf += STEP;
    while (qf >= a.HW) qf -= a.HW;

    // ================= phase A: a = relu(sa * (Wa x) + ba) -> ring
    {
      f32x4 acc[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto a_chunk = [&](auto jc_c) {
        constexpr int JC = decltype(jc_c)::value;
        const char* cur = chunk_top(std::integral_constant<int, JC>{}, std::integral_constant<int, (JC == 0 ? 6 : 1)>{});
        int lofs = lane * 16;
        asm volatile("" : "+v"(lofs));
        // the chunk's fragment pairs are read AHEAD of their MFMAs: with every wave of the CU reading, an LDS read returns after
        // ~200 cycles — read-then-use per pair made the kernel LDS-latency-bound (v3a: 136 pairs x ~200 cycles per iteration)
        i32x4 wf_[CH_PAIRS][2];
#pragma unroll
        for (int pr = 0; pr < AHEAD; ++pr) { wf_[pr][0] = WP(cur, pr, 0, lofs); wf_[pr][1] = WP(cur, pr, 1, lofs); }
#pragma unroll
        for (int pr = 0; pr < CH_PAIRS; ++pr) {  // pair (kk, nt) at 4 kk + nt: k-step 2 JC + kk, n-tile nt
          if (pr + AHEAD < CH_PAIRS) { wf_[pr + AHEAD][0] = WP(cur, pr + AHEAD, 0, lofs); wf_[pr + AHEAD][1] = WP(cur, pr + AHEAD, 1, lofs); }
          R2_SCHED_FENCE();
          const int k = JC * 2 + pr / 4, n = pr % 4;
          acc[n] = mfma3<F16>(wf_[pr][0], wf_[pr][1], xt[SET][k][0], xt[SET][k][1], acc[n]);
          R2_SCHED_FENCE();
        }
      };
      a_chunk(std::integral_constant<int, 0>{});
      a_chunk(std::integral_constant<int, 1>{});
      a_chunk(std::integral_constant<int, 2>{});
      a_chunk(std::integral_constant<int, 3>{});
      R2_ST(1);
      const unsigned sw = (ra >> 1) & 7u;
#pragma unroll
      for (int m = 0; m < 2; ++m) {  // n-tiles 2 m, 2 m + 1 give this lane channels 32 m + 8 q .. + 7 (the packing's row order)
        const int c0 = 32 * m + 8 * q;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = avt::relu_keep_nan(acc[2 * m + (e >> 2)][e & 3] * cf[c0 + e] + cf[CM + c0 + e]);
        uint4 h, l;
        r2_split<F16>(v, h, l);
        const unsigned chunk = (unsigned)(4 * m + q);
        const unsigned o = ra * 128u + ((chunk ^ sw) * 16u);
        *reinterpret_cast<uint4*>(lds + o) = h;
        *reinterpret_cast<uint4*>(lds + RING_PLANE + o) = l;
      }
    }

    // ================= phase B (step i - 1): b = relu(sb * (Wb * taps(a)) + bb), operands from the ring
    i32x4 zb[2][2];  // phase C's operand: k-step m of b's output, [plane]
    {
      f32x4 acc[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto b_tap = [&](auto t_c) {
        constexpr int T = decltype(t_c)::value;
        constexpr int dy = T / 3 - 1, dx = T % 3 - 1;
        const char* cur = chunk_top(std::integral_constant<int, NCH_A + T>{}, std::integral_constant<int, (T == 0 ? 2 : 3)>{});
        // the neighbour's ring slot, or the zero slot when it lies outside the frame
        const bool ok = pb_in && (unsigned)(y + dy) < (unsigned)a.H && (unsigned)(xc + dx) < (unsigned)W;
        int nb = (int)rb + dy * W + dx;
        nb = nb < 0 ? nb + RING : (nb >= RING ? nb - RING : nb);
        const unsigned nsw = ((unsigned)nb >> 1) & 7u;
        int lofs = lane * 16;
        asm volatile("" : "+v"(lofs));
        i32x4 fx[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const unsigned o = ok ? (unsigned)nb * 128u + ((((unsigned)(4 * k + q)) ^ nsw) * 16u) : (unsigned)ZERO_OFF;
          fx[k][0] = *reinterpret_cast<const i32x4*>(lds + o);
          fx[k][1] = *reinterpret_cast<const i32x4*>(lds + (ok ? RING_PLANE : 0) + o);
        }
        i32x4 wf_[CH_PAIRS][2];
#pragma unroll
        for (int pr = 0; pr < AHEAD; ++pr) { wf_[pr][0] = WP(cur, pr, 0, lofs); wf_[pr][1] = WP(cur, pr, 1, lofs); }
#pragma unroll
        for (int pr = 0; pr < CH_PAIRS; ++pr) {  // pair (kk, nt) at 4 kk + nt
          if (pr + AHEAD < CH_PAIRS) { wf_[pr + AHEAD][0] = WP(cur, pr + AHEAD, 0, lofs); wf_[pr + AHEAD][1] = WP(cur, pr + AHEAD, 1, lofs); }
          R2_SCHED_FENCE();
          const int k = pr / 4, n = pr % 4;
          acc[n] = mfma3<F16>(wf_[pr][0], wf_[pr][1], fx[k][0], fx[k][1], acc[n]);
          R2_SCHED_FENCE();
        }
      };
      b_tap(std::integral_constant<int, 0>{});
      b_tap(std::integral_constant<int, 1>{});
      b_tap(std::integral_constant<int, 2>{});
      b_tap(std::integral_constant<int, 3>{});
      b_tap(std::integral_constant<int, 4>{});
      b_tap(std::integral_constant<int, 5>{});
      b_tap(std::integral_constant<int, 6>{});
      b_tap(std::integral_constant<int, 7>{});
      b_tap(std::integral_constant<int, 8>{});
      R2_ST(3);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int c0 = 32 * m + 8 * q;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = avt::relu_keep_nan(acc[2 * m + (e >> 2)][e & 3] * cf[2 * CM + c0 + e] + cf[3 * CM + c0 + e]);
        uint4 h, l;
        r2_split<F16>(v, h, l);
        zb[m][0] = __builtin_bit_cast(i32x4, h);
        zb[m][1] = __builtin_bit_cast(i32x4, l);
      }
    }

    // ================= phase C (step i - 1): out = relu(sc * (Wc b) + bc + x); x = the tile phase A used one iteration ago, still in
    // registers; the tile of step i + 1 is requested into them as they are released
    {
      const unsigned out_base = (pb_in && store_ok) ? ((unsigned)pb * (unsigned)a.ldo + (unsigned)(q * 8)) * 2u : kOob;
      const unsigned xn = (i + 1 <= s1 && !R2_SKIP(2)) ? x_off(pa + STEP) : kOob;  // (past the last iteration: the loads still issue)
      auto c_chunk = [&](auto jc_c) {
        constexpr int JC = decltype(jc_c)::value;
        const char* cur = chunk_top(std::integral_constant<int, NCH_A + NCH_B + JC>{}, std::integral_constant<int, (JC == 0 ? 4 : 6)>{});
        int lofs = lane * 16;
        asm volatile("" : "+v"(lofs));
        // all four n-tiles' MFMAs first (fragments read ahead), then the two epilogues: the epilogue VALU of this wave runs under the
        // other wave's MFMAs
        f32x4 acs[4];
        {
          i32x4 wf_[CH_PAIRS][2];
#pragma unroll
          for (int pr = 0; pr < AHEAD; ++pr) { wf_[pr][0] = WP(cur, pr, 0, lofs); wf_[pr][1] = WP(cur, pr, 1, lofs); }
#pragma unroll
          for (int pr = 0; pr < CH_PAIRS; ++pr) {  // pair (nn, kk) at 2 nn + kk
            if (pr + AHEAD < CH_PAIRS) { wf_[pr + AHEAD][0] = WP(cur, pr + AHEAD, 0, lofs); wf_[pr + AHEAD][1] = WP(cur, pr + AHEAD, 1, lofs); }
            R2_SCHED_FENCE();
            const int nn = pr / 2, k = pr % 2;
            if (k == 0) acs[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
            acs[nn] = mfma3<F16>(wf_[pr][0], wf_[pr][1], zb[k][0], zb[k][1], acs[nn]);
            R2_SCHED_FENCE();
          }
        }
        R2_ST(5);
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {  // output channels 32 M + 8 q .. + 7: n-tiles 2 M, 2 M + 1 = the chunk's n-tiles 2 mm, 2 mm + 1
          const int M = JC * 2 + mm;
          const f32x4 ac[2] = {acs[2 * mm], acs[2 * mm + 1]};
          const int c0 = 32 * M + 8 * q;
          float v[8];
          const i32x4 rh = xt[OLD][M][0], rl = xt[OLD][M][1];  // channels 32 M + 8 q .. of this position: the residual
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            avt::f32x2 r;  // (fp16 planes: hi + lo as ONE mixed-precision FMA per value, the bits of join2's two conversions + add)
            if constexpr (F16) r = avt::join2_mix_f16((uint32_t)rh[e], (uint32_t)rl[e]);
            else r = avt::join2<F16>((uint32_t)rh[e], (uint32_t)rl[e]);
            v[2 * e] = ac[(2 * e) >> 2][(2 * e) & 3] * cf[4 * CM + c0 + 2 * e] + cf[4 * CM + C + c0 + 2 * e] + r.x;
            v[2 * e + 1] = ac[(2 * e + 1) >> 2][(2 * e + 1) & 3] * cf[4 * CM + c0 + 2 * e + 1] + cf[4 * CM + C + c0 + 2 * e + 1] + r.y;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = avt::relu_keep_nan(v[e]);
          uint4 oh, ol;
          r2_split<F16>(v, oh, ol);
          const int off = (int)((out_base != kOob && !R2_SKIP(1)) ? out_base + (unsigned)(M * 64) : kOob);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, oh), roh, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, ol), rol, off, 0, 0);
          load_tile(std::integral_constant<int, OLD>{}, M, xn);  // k-step M of the tile of step i + 1 into the registers just released
        }
      };
      c_chunk(std::integral_constant<int, 0>{});
      c_chunk(std::integral_constant<int, 1>{});
      c_chunk(std::integral_constant<int, 2>{});
      c_chunk(std::integral_constant<int, 3>{});
    }
  };

  // iterations i = s0 - 1 .. s1: phase A runs one step ahead of B / C (step s0 - 1 gives step s0 its upper halo; B / C of steps
  // s0 - 2 and s0 - 1 compute on unwritten ring slots and store nothing)
  for (int i = s0 - 1; i <= s1;) {
    iteration(std::integral_constant<int, 0>{}, i);
    if (++i > s1) break;
    iteration(std::integral_constant<int, 1>{}, i);
    ++i;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the two chunks requested past the end land before the LDS is released
  R2_ST(6);
  R2_ST_END();
}

template <int W, bool F16>
int launch(R2Args& a, hipStream_t st) {
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(res2_x3_kernel<W, F16>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) {
    avt::set_error("avt_res2_x3: hipFuncSetAttribute(%d B LDS): %s", LDS_BYTES, hipGetErrorString(e));
    return AVT_ERR_LAUNCH;
  }
  int grid = a.nsteps < 256 ? a.nsteps : 256;  // persistent: one workgroup per CU (147 KB of LDS each)
  a.spw = (a.nsteps + grid - 1) / grid;
  grid = (a.nsteps + a.spw - 1) / a.spw;
  hipLaunchKernelGGL((res2_x3_kernel<W, F16>), dim3((unsigned)grid), dim3(NWV * 64), LDS_BYTES, st, a);
  return avt::check_launch("avt_res2_x3");
}

}  // namespace

#ifdef R2_STAMP
extern "C" int avt_debug_stamps_res2(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_r2_stamp), sizeof(unsigned long long) * 10);
  if (reset) {
    unsigned long long z[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_r2_stamp), z, sizeof(z));
  }
  return 0;
}
#endif

extern "C" int avt_res2_x3_supported(int c, int cm, int w) { return (c == C && cm == CM && (w == 56 || w == 12)) ? 1 : 0; }

// bytes of the packed weight stream avt_res2_x3 reads (fused_slowfast.pack_res2_x3)
extern "C" int avt_res2_x3_wfrag_bytes(void) { return NCH * CH_BYTES; }

extern "C" int avt_res2_x3(const void* x_hi, const void* x_lo, void* out_hi, void* out_lo, const void* wfrag, const float* coef, int batch,
                           int t, int h, int w, int ldi, int ldo, int plane_dtype, void* stream) {
  AVT_REQUIRE(x_hi && x_lo && out_hi && out_lo && wfrag && coef, "avt_res2_x3: NULL pointer");
  AVT_REQUIRE(avt_res2_x3_supported(C, CM, w), "avt_res2_x3: unsupported width %d (56; 12 for tests)", w);
  AVT_REQUIRE(batch > 0 && t > 0 && h > 0 && ldi >= C && ldo >= C && ldi % 8 == 0 && ldo % 8 == 0,
              "avt_res2_x3: bad sizes (256 channels; row pitches in multiples of 8)");
  AVT_REQUIRE(x_hi != out_hi && x_lo != out_lo, "avt_res2_x3: in-place is not supported (the residual is re-read after neighbours stored)");
  AVT_REQUIRE(avt::aligned16(x_hi) && avt::aligned16(x_lo) && avt::aligned16(out_hi) && avt::aligned16(out_lo) && avt::aligned16(wfrag) &&
                  avt::aligned16(coef),
              "avt_res2_x3: pointers must be 16-byte aligned");
  // (the bf16 split costs 32 more registers than the fp16 one: the instance spills inside its load sequences; bf16x3 keeps the three launches)
  AVT_REQUIRE(plane_dtype == AVT_X3_F16, "avt_res2_x3: fp16 planes only (plane_dtype %d)", plane_dtype);
  const int64_t m = (int64_t)batch * t * h * w;
  AVT_REQUIRE(m * ldi * 2 < (1ll << 32) - 64 && m * ldo * 2 < (1ll << 32) - 64 && m < (1ll << 31) - 4 * STEP,
              "avt_res2_x3: tensor too large for 32-bit offsets");
  R2Args a;
  a.xh = static_cast<const uint16_t*>(x_hi);
  a.xl = static_cast<const uint16_t*>(x_lo);
  a.oh = static_cast<uint16_t*>(out_hi);
  a.ol = static_cast<uint16_t*>(out_lo);
  a.wf = wfrag;
  a.coef = coef;
  a.P = (int)m;
  a.H = h;
  a.HW = h * w;
  a.ldi = ldi;
  a.ldo = ldo;
  a.nsteps = (int)((m + STEP - 1) / STEP);
  a.x_bytes = (unsigned)(m * ldi * 2);
  a.o_bytes = (unsigned)(m * ldo * 2);
  a.w_bytes = (unsigned)(NCH * CH_BYTES);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return w == 56 ? launch<56, true>(a, s) : launch<12, true>(a, s);
}
