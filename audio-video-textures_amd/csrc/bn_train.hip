// bn_train — train-mode BatchNorm3d fused with what surrounds it in the SlowFast blocks, forward and backward, for
// the contrastive TRAINING step (BASELINE config 5; contrastive_video_textures/train.py:114-141 runs the third-party
// SlowFast in train mode: per-replica batch statistics, models/models.py:385-417):
//     y = act( (x - mean_c) * invstd_c * gamma_c + beta_c  [+ r] )          act = ReLU or identity, r = the block's shortcut
// In the steady-state step of the MIOpen path a third of the device time is NOT convolution (profiles/r02/
// train_fp32_steady_state_kernels.log): MIOpenBatchNormFwdTrainSpatial 11.7 %, MIOpenBatchNormBwdSpatial 11.2 %, the
// residual adds / ReLUs / their backward ~8 % as separate elementwise passes.  Here the normalise, the shortcut add and the
// ReLU are ONE pass over the rows in each direction, next to one statistics pass: forward = read x twice (+ r once), write
// y once; backward = read dy, x (and y when there was a shortcut) twice, write dx (+ dr) once — without a shortcut the ReLU mask
// is RECOMPUTED from x with the forward's own expression (same operations in the same order, -ffp-contract=off: the same bits),
// which takes the forward output out of both backward passes (a third of their bytes).  HBM-bound; channels-last rows [M, C] (the physical layout
// of a channels_last_3d tensor), C a power of two >= 8 (every BatchNorm of SlowFast-8x8-R50).
// Statistics accumulate in fp64 with NO atomics (hipcc lowers atomicAdd(double) to a compare-and-swap loop, which under a
// few thousand workgroups per address cost 6x the whole pass — measured, profiles/r02/train_fused_bn_atomics.log): per-thread
// registers -> a halving tree in LDS -> one row of partials per workgroup -> a finalize launch where one wavefront per channel
// sums the rows in a fixed order.  The result is bitwise reproducible run to run, and the mean / variance are the correctly
// rounded ones.  The finalize launch also turns the sums into the per-channel coefficients the apply pass needs, so that
// pass reads two floats per channel instead of redoing the fp64 arithmetic in every thread.
// A thread walks the rows with a stride that is a multiple of the row's float4 chunks, so it always sees the SAME four
// channels and keeps their partial sums / scale / shift in registers.
// GROUPS (round 3): the rows are `groups` equal, consecutive slabs, each with its OWN batch statistics — the items of a batch as
// the reference's DataParallel replicas see them (per-replica BatchNorm, main.py:420) in ONE launch over the whole batch:
// blockIdx.y walks the groups in the streaming passes, a finalize wavefront walks them in order for its channel (so the running
// statistics receive the groups' updates in item order, as a loop over the items would apply them).
#include "avt_common.h"

namespace {

constexpr int kT = 256;

// streamed once: non-temporal loads in every pass and non-temporal stores of the backward's outputs: +2.4 % on the training
// step (profiles/r03/train_bn_nt_ab.log)
typedef float f32x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4(const float* p, int64_t i) {
  const f32x4n v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(p) + i);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stg4(float* p, int64_t i, float4 v) {
  __builtin_nontemporal_store(f32x4n{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4n*>(p) + i);
}
constexpr int kMaxBlocks = 1024;  // 4 workgroups per CU: enough 16-byte loads in flight for HBM, few enough rows of partials

struct BnArgs {
  const float* x;
  const float* res;     // forward: shortcut added before the activation, or NULL
  const float* dy;      // backward
  const float* y;       // backward: the forward output (ReLU mask), or NULL when there was no ReLU
  uint8_t* mask;        // the ReLU mask as 4 bits per float4 chunk (one byte per chunk): written by the forward's apply pass when
                        // not NULL, read by the backward's passes INSTEAD of y (a sixteenth of its bytes)
  float* out;           // forward: y; backward: dx
  float* dres;          // backward: gradient of the shortcut (= masked dy), or NULL
  double* part;         // [blocks][nq * 8] per-workgroup partial sums (slot = quad-in-workgroup * 8 + {p0[4], p1[4]})
  float* coef;          // [2C] forward: scale, shift; backward: mean(dz), mean(dz * xhat)
  const float* gamma;
  const float* beta;
  const float* mean;    // backward: saved mean / invstd
  const float* invstd;
  float* save_mean;     // forward outputs
  float* save_invstd;
  float* running_mean;  // forward: updated in place when not NULL
  float* running_var;
  long long* tracked;   // forward: nn.BatchNorm's num_batches_tracked, incremented here when not NULL (no launch of its own)
  float* dgamma;        // backward outputs
  float* dbeta;
  int64_t M;            // rows PER GROUP
  int groups;           // slabs of M rows with statistics of their own (1: plain BatchNorm)
  int C;
  int relu;
  float eps, momentum;
  int64_t nchunk;       // M * C / 4
  int64_t stride;       // total threads (a multiple of C / 4)
  int blocks;
  int nq;               // quads a workgroup sees: min(C / 4, kT)
  int unit;             // workgroups per row when a row is wider than one (C / 4 > kT), else 1
  int64_t ld4;          // leading dimension, in float4 chunks, of the ONE tensor that may be a channel slice of wider rows: the forward's
                        // out, the backward's dy (a concatenation buffer's slice: no torch.cat copy, no .contiguous() of its gradient);
                        // C / 4 when contiguous
};

// this thread's four channels (fixed for the whole walk)
__device__ __forceinline__ int my_quad(const BnArgs& a) {
  const int64_t i0 = (int64_t)blockIdx.x * kT + threadIdx.x;
  return (int)(i0 % (a.C / 4));
}

// the walk over the strided tensor (a.ld4): this thread's first chunk and its step — chunk i of the contiguous walk is (row i / (C/4),
// quad i % (C/4)); the stride is a whole number of rows, so the quad stays and the row advances by stride / (C/4)
__device__ __forceinline__ int64_t strided_first(const BnArgs& a) {
  const int64_t i0 = (int64_t)blockIdx.x * kT + threadIdx.x;
  const int q4 = a.C / 4;
  return ((int64_t)blockIdx.y * a.M + i0 / q4) * a.ld4 + i0 % q4;
}
__device__ __forceinline__ int64_t strided_step(const BnArgs& a) { return a.stride / (a.C / 4) * a.ld4; }

// per-thread fp64 partials -> halving tree in LDS (threads t and t + w share their quad for every power of two w >= nq)
// -> this workgroup's row of partials.  Fixed order: deterministic.
__device__ __forceinline__ void block_reduce_store(const BnArgs& a, const double* p0, const double* p1) {
  __shared__ double acc[8][kT];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc[e][threadIdx.x] = p0[e];
    acc[4 + e][threadIdx.x] = p1[e];
  }
  __syncthreads();
  for (int w = kT / 2; w >= a.nq; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k][threadIdx.x] += acc[k][threadIdx.x + w];
    }
    __syncthreads();
  }
  double* row = a.part + (size_t)blockIdx.x * a.nq * 8;
  for (int i = threadIdx.x; i < a.nq * 8; i += kT) row[i] = acc[i & 7][i >> 3];
}

// One wavefront per channel: sum that channel's two partials over the workgroups that saw it, in a fixed order — for up to
// kGroupBatch groups AT ONCE (a lane keeps one pair of sums per group and walks the groups' rows of partials together: with
// 8 groups walked one after the other the finalize launches were 64 us each, 25 ms of a 415 ms step).  Valid in lane 0.
constexpr int kGroupBatch = 8;
__device__ __forceinline__ void channel_sums(const BnArgs& a, int g0, int ng, int c, double* s0, double* s1) {
  const int quad = c >> 2, e = c & 3, lane = threadIdx.x & 63;
  const int slot = (quad % a.nq) * 8 + e;
  const int first = a.unit > 1 ? quad / kT : 0;  // wide rows: workgroup b holds quads (b % unit) * kT ...
  double t0[kGroupBatch], t1[kGroupBatch];
#pragma unroll
  for (int j = 0; j < kGroupBatch; ++j) t0[j] = t1[j] = 0.0;
  const size_t gstride = (size_t)a.blocks * a.nq * 8;
  for (int b = first + lane * a.unit; b < a.blocks; b += 64 * a.unit) {
    const double* row = a.part + ((size_t)g0 * a.blocks + b) * a.nq * 8 + slot;
#pragma unroll
    for (int j = 0; j < kGroupBatch; ++j)
      if (j < ng) {  // uniform
        t0[j] += row[j * gstride];
        t1[j] += row[j * gstride + 4];
      }
  }
#pragma unroll
  for (int j = 0; j < kGroupBatch; ++j) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      t0[j] += __shfl_down(t0[j], off, 64);
      t1[j] += __shfl_down(t1[j], off, 64);
    }
    s0[j] = t0[j];
    s1[j] = t1[j];
  }
}

// this workgroup's group: its slab of the activation (in float4 chunks) and its rows of partials
__device__ __forceinline__ int64_t slab(const BnArgs& a) { return (int64_t)blockIdx.y * a.nchunk; }

__global__ __launch_bounds__(kT) void bn_fwd_stats_kernel(BnArgs a) {
  a.x += 4 * slab(a);
  a.part += (size_t)blockIdx.y * a.blocks * a.nq * 8;
  double p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
  // (two rows of the walk per trip: both loads are issued before either is summed — same order of additions, twice the bytes
  //  in flight per thread)
  auto add = [&](const float4 v) {
    p0[0] += v.x; p1[0] += (double)v.x * v.x;
    p0[1] += v.y; p1[1] += (double)v.y * v.y;
    p0[2] += v.z; p1[2] += (double)v.z * v.z;
    p0[3] += v.w; p1[3] += (double)v.w * v.w;
  };
  int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  for (; i + a.stride < a.nchunk; i += 2 * a.stride) {
    const float4 v = ldg4(a.x, i);
    const float4 u = ldg4(a.x, i + a.stride);
    add(v);
    add(u);
  }
  if (i < a.nchunk) add(ldg4(a.x, i));
  block_reduce_store(a, p0, p1);
}

__global__ __launch_bounds__(kT) void bn_fwd_finalize_kernel(BnArgs a) {  // grid C / 4, a wavefront per channel
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int g0 = 0; g0 < a.groups; g0 += kGroupBatch) {
    const int ng = a.groups - g0 < kGroupBatch ? a.groups - g0 : kGroupBatch;
    double s0[kGroupBatch], s1[kGroupBatch];
    channel_sums(a, g0, ng, c, s0, s1);
    if ((threadIdx.x & 63) != 0) continue;
    for (int j = 0; j < ng; ++j) {
      const int g = g0 + j;
      // running statistics / num_batches_tracked: GROUP 0's update only.  The groups stand for the reference's DataParallel
      // replicas (main.py:420), and DataParallel keeps the buffer updates of the replica on device 0 alone (the other replicas'
      // buffers are broadcast copies that are dropped after the forward): one momentum step and +1 tracked per forward
      if (a.tracked && c == 0 && g == 0) *a.tracked += 1;
      const double mean = s0[j] / (double)a.M;
      double var = s1[j] / (double)a.M - mean * mean;  // biased variance (normalisation)
      var = var > 0.0 ? var : 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
      const float sc = invstd * a.gamma[c];
      float* coef = a.coef + (size_t)g * 2 * a.C;
      coef[c] = sc;
      coef[a.C + c] = a.beta[c] - (float)mean * sc;
      a.save_mean[(size_t)g * a.C + c] = (float)mean;
      a.save_invstd[(size_t)g * a.C + c] = invstd;
      if (a.running_mean && g == 0) {  // torch: running = (1 - m) * running + m * batch, running_var with the UNBIASED batch variance
        const double unb = a.M > 1 ? var * (double)a.M / (double)(a.M - 1) : var;
        a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
        a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * (float)unb;
      }
    }
  }
}

// Pre-reduction of a producer's partial rows (avt_bn_train_fwd_pre): a convolution leaves one row per M tile — thousands per group on
// the early layers — and the finalize launch (one wavefront per channel walking ALL rows, a few workgroups in all) doubled in time
// on them (profiles/r05: 19 -> 40 us per BatchNorm).  Here every workgroup sums kPreRows consecutive rows of one group (same `unit`
// phase) into one row of a second array, in a fixed order; the finalize launch then walks rows / kPreRows rows.
constexpr int kPreRows = 64;
struct PreArgs {
  const double* src;   // [groups][rows * unit][len]
  double* dst;         // [groups][ceil(rows / kPreRows) * unit][len]
  int rows, unit, len; // len = nq * 8 doubles per row
};
__global__ __launch_bounds__(kT) void bn_pre_reduce_kernel(PreArgs a) {
  __shared__ double sh[kT];
  const int chunk = blockIdx.x, u = blockIdx.y % a.unit, g = blockIdx.y / a.unit;
  const int nchunk = (a.rows + kPreRows - 1) / kPreRows;
  const int cols = a.len < kT ? a.len : kT, slices = kT / cols;  // (len is a power of two: 64 .. 2048)
  const int r0 = chunk * kPreRows, r1 = r0 + kPreRows < a.rows ? r0 + kPreRows : a.rows;
  const double* src = a.src + ((size_t)g * a.rows * a.unit + u) * a.len;
  double* dst = a.dst + (((size_t)g * nchunk + chunk) * a.unit + u) * a.len;
  const int sl = threadIdx.x / cols, c = threadIdx.x % cols;
  const int col = blockIdx.z * cols + c;  // (grid.z = len / cols column blocks: a launch is one round of loads deep, not len / kT)
  // kPreDepth rows' loads are issued before the first is added — in the same order as a one-by-one walk, so the sums keep their
  // bits: the walk was a chain of 32-64 dependent L2 round trips (24 us per launch on a config-5 step of ONE item, 243 launches a
  // step: profiles/r06/rocprof_train_one_item_kernel_stats.csv)
  constexpr int kPreDepth = 16;
  const size_t rstride = (size_t)a.unit * a.len;
  double acc = 0.0;
  int r = r0 + sl;
  for (; r + (kPreDepth - 1) * slices < r1; r += kPreDepth * slices) {
    double v[kPreDepth];
#pragma unroll
    for (int k = 0; k < kPreDepth; ++k) v[k] = src[(size_t)(r + k * slices) * rstride + col];
#pragma unroll
    for (int k = 0; k < kPreDepth; ++k) acc += v[k];
  }
  for (; r < r1; r += slices) acc += src[(size_t)r * rstride + col];
  if (slices > 1) {
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0) {
      for (int k = 1; k < slices; ++k) acc += sh[k * cols + c];
      dst[col] = acc;
    }
  } else {
    dst[col] = acc;
  }
}

__global__ __launch_bounds__(kT) void bn_fwd_apply_kernel(BnArgs a) {
  const int quad = my_quad(a);
  a.x += 4 * slab(a);
  if (a.res) a.res += 4 * slab(a);
  a.coef += (size_t)blockIdx.y * 2 * a.C;
  a.save_mean += (size_t)blockIdx.y * a.C;
  if (a.mask) a.mask += slab(a);
  int64_t io = strided_first(a);
  const int64_t so = strided_step(a);
  // y = (x - mean) * (invstd * gamma) + beta, the subtraction FIRST as stock BatchNorm does it: the folded form
  // x * sc + (beta - mean * sc) cancels two large terms when |mean| >> std (error ~ 2^-24 |mean| / std of the result)
  const float4 sc = reinterpret_cast<const float4*>(a.coef)[quad];
  const float4 mu = reinterpret_cast<const float4*>(a.save_mean)[quad];
  const float4 be = reinterpret_cast<const float4*>(a.beta)[quad];
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < a.nchunk; i += a.stride, io += so) {
    const float4 v = ldg4(a.x, i);
    float4 o = make_float4((v.x - mu.x) * sc.x + be.x, (v.y - mu.y) * sc.y + be.y, (v.z - mu.z) * sc.z + be.z,
                           (v.w - mu.w) * sc.w + be.w);
    if (a.res) {
      const float4 r = ldg4(a.res, i);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    if (a.relu) {
      if (a.mask) a.mask[i] = (uint8_t)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    reinterpret_cast<float4*>(a.out)[io] = o;  // (a non-temporal store here measured equal: the next convolution reads y at once)
  }
}

__global__ __launch_bounds__(kT) void bn_bwd_stats_kernel(BnArgs a) {
  const int quad = my_quad(a);
  a.x += 4 * slab(a);
  if (a.y) a.y += 4 * slab(a);
  if (a.mask) a.mask += slab(a);
  a.mean += (size_t)blockIdx.y * a.C;
  a.invstd += (size_t)blockIdx.y * a.C;
  a.part += (size_t)blockIdx.y * a.blocks * a.nq * 8;
  int64_t id = strided_first(a);
  const int64_t sd = strided_step(a);
  const float4 mu = reinterpret_cast<const float4*>(a.mean)[quad];
  const float4 is = reinterpret_cast<const float4*>(a.invstd)[quad];
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), be = sc;
  if (!a.y && a.relu) {  // the forward's scale = invstd * gamma (bn_fwd_finalize_kernel) and shift
    const float4 gm = reinterpret_cast<const float4*>(a.gamma)[quad];
    sc = make_float4(is.x * gm.x, is.y * gm.y, is.z * gm.z, is.w * gm.w);
    be = reinterpret_cast<const float4*>(a.beta)[quad];
  }
  double p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < a.nchunk; i += a.stride, id += sd) {
    float4 g = ldg4(a.dy, id);
    const float4 v = ldg4(a.x, i);
    if (a.mask) {  // ReLU backward: the gradient passes where the forward output was positive — from the saved bits,
      const unsigned mk = a.mask[i];
      g.x = (mk & 1u) ? g.x : 0.f; g.y = (mk & 2u) ? g.y : 0.f; g.z = (mk & 4u) ? g.z : 0.f; g.w = (mk & 8u) ? g.w : 0.f;
    } else if (a.y) {  // ... from the output itself,
      const float4 yv = reinterpret_cast<const float4*>(a.y)[i];
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    } else if (a.relu) {  // ... recomputed: bn_fwd_apply_kernel's expression, no shortcut
      g.x = (v.x - mu.x) * sc.x + be.x > 0.f ? g.x : 0.f; g.y = (v.y - mu.y) * sc.y + be.y > 0.f ? g.y : 0.f;
      g.z = (v.z - mu.z) * sc.z + be.z > 0.f ? g.z : 0.f; g.w = (v.w - mu.w) * sc.w + be.w > 0.f ? g.w : 0.f;
    }
    p0[0] += g.x; p1[0] += (double)g.x * ((v.x - mu.x) * is.x);
    p0[1] += g.y; p1[1] += (double)g.y * ((v.y - mu.y) * is.y);
    p0[2] += g.z; p1[2] += (double)g.z * ((v.z - mu.z) * is.z);
    p0[3] += g.w; p1[3] += (double)g.w * ((v.w - mu.w) * is.w);
  }
  block_reduce_store(a, p0, p1);
}

__global__ __launch_bounds__(kT) void bn_bwd_finalize_kernel(BnArgs a) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  double g0s = 0.0, g1s = 0.0;  // dbeta / dgamma: over ALL groups (the parameters are shared), summed in group order
  for (int g0 = 0; g0 < a.groups; g0 += kGroupBatch) {
    const int ng = a.groups - g0 < kGroupBatch ? a.groups - g0 : kGroupBatch;
    double s0[kGroupBatch], s1[kGroupBatch];
    channel_sums(a, g0, ng, c, s0, s1);
    if ((threadIdx.x & 63) != 0) continue;
    for (int j = 0; j < ng; ++j) {
      float* coef = a.coef + (size_t)(g0 + j) * 2 * a.C;
      coef[c] = (float)(s0[j] / (double)a.M);        // mean of dz within the group
      coef[a.C + c] = (float)(s1[j] / (double)a.M);  // mean of dz * xhat
      g0s += s0[j];
      g1s += s1[j];
    }
  }
  if ((threadIdx.x & 63) != 0) return;
  a.dbeta[c] = (float)g0s;
  a.dgamma[c] = (float)g1s;
}

__global__ __launch_bounds__(kT) void bn_bwd_apply_kernel(BnArgs a) {
  const int quad = my_quad(a);
  int64_t id = strided_first(a);
  const int64_t sd = strided_step(a);
  a.x += 4 * slab(a);
  a.out += 4 * slab(a);
  if (a.y) a.y += 4 * slab(a);
  if (a.mask) a.mask += slab(a);
  if (a.dres) a.dres += 4 * slab(a);
  a.mean += (size_t)blockIdx.y * a.C;
  a.invstd += (size_t)blockIdx.y * a.C;
  a.coef += (size_t)blockIdx.y * 2 * a.C;
  const float4 mu = reinterpret_cast<const float4*>(a.mean)[quad];
  const float4 is = reinterpret_cast<const float4*>(a.invstd)[quad];
  const float4 gm = reinterpret_cast<const float4*>(a.gamma)[quad];
  const float4 k0 = reinterpret_cast<const float4*>(a.coef)[quad];
  const float4 k1 = reinterpret_cast<const float4*>(a.coef + a.C)[quad];
  const float4 gs = make_float4(gm.x * is.x, gm.y * is.y, gm.z * is.z, gm.w * is.w);
  const float4 sc = make_float4(is.x * gm.x, is.y * gm.y, is.z * gm.z, is.w * gm.w);  // the forward's scale (same product)
  float4 be = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!a.y && a.relu) be = reinterpret_cast<const float4*>(a.beta)[quad];
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < a.nchunk; i += a.stride, id += sd) {
    float4 g = ldg4(a.dy, id);
    const float4 v = ldg4(a.x, i);
    if (a.mask) {
      const unsigned mk = a.mask[i];
      g.x = (mk & 1u) ? g.x : 0.f; g.y = (mk & 2u) ? g.y : 0.f; g.z = (mk & 4u) ? g.z : 0.f; g.w = (mk & 8u) ? g.w : 0.f;
    } else if (a.y) {
      const float4 yv = reinterpret_cast<const float4*>(a.y)[i];
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    } else if (a.relu) {
      g.x = (v.x - mu.x) * sc.x + be.x > 0.f ? g.x : 0.f; g.y = (v.y - mu.y) * sc.y + be.y > 0.f ? g.y : 0.f;
      g.z = (v.z - mu.z) * sc.z + be.z > 0.f ? g.z : 0.f; g.w = (v.w - mu.w) * sc.w + be.w > 0.f ? g.w : 0.f;
    }
    if (a.dres) stg4(a.dres, i, g);
    float4 o;
    o.x = gs.x * (g.x - k0.x - (v.x - mu.x) * is.x * k1.x);
    o.y = gs.y * (g.y - k0.y - (v.y - mu.y) * is.y * k1.y);
    o.z = gs.z * (g.z - k0.z - (v.z - mu.z) * is.z * k1.z);
    o.w = gs.w * (g.w - k0.w - (v.w - mu.w) * is.w * k1.w);
    stg4(a.out, i, o);
  }
}

bool shape_ok(int64_t m, int c) { return m > 0 && c >= 8 && (c & (c - 1)) == 0 && c <= 4096; }

void layout(BnArgs& a, int64_t m, int c) {
  a.M = m;
  a.C = c;
  a.nchunk = m * (int64_t)(c / 4);
  const int q = c / 4;
  a.nq = q < kT ? q : kT;
  a.unit = q > kT ? q / kT : 1;
  int64_t blocks = (a.nchunk + kT - 1) / kT;
  // (with groups the grid is blocks x groups: the same 1024-2048 workgroups in all, so that the finalize pass — one wavefront
  //  per channel over every row of partials — does not grow with the number of groups)
  const int64_t most = a.groups > 1 ? (2 * kMaxBlocks / a.groups > 128 ? 2 * kMaxBlocks / a.groups : 128) : kMaxBlocks;
  if (blocks > most) blocks = most;
  // every workgroup leaves a row of 2 C doubles for the finalize pass: keep those rows a few per cent of the tensor itself
  // (with 1024 of them a [23520, 1024] activation's partials were a third of its bytes: 3.4 TB/s backward instead of 5.2)
  // — but not below two workgroups per CU, which the streaming itself needs
  const int64_t floor_ = a.groups > 1 ? 128 : 512;
  const int64_t cap = m / 64 + 1 > floor_ ? m / 64 + 1 : floor_;
  if (blocks > cap) blocks = cap;
  blocks = ((blocks + a.unit - 1) / a.unit) * a.unit;  // whole rows per sweep, so a thread keeps its four channels
  a.blocks = (int)blocks;
  a.stride = blocks * kT;
}

size_t ws_bytes(const BnArgs& a) {
  return (size_t)a.groups * (((size_t)a.blocks * a.nq * 8) * sizeof(double) + (size_t)2 * a.C * sizeof(float));
}

int geometry(BnArgs& a, const char* who, int64_t m, int c, int groups, void* ws, size_t ws_size) {
  AVT_REQUIRE(shape_ok(m, c), "%s: rows > 0 and a power-of-two channel count in 8..4096 (got %lld x %d)", who, (long long)m, c);
  AVT_REQUIRE(groups >= 1 && groups <= 65535 && m % groups == 0, "%s: %lld rows do not split into %d groups", who, (long long)m, groups);
  a.groups = groups;
  layout(a, m / groups, c);
  AVT_REQUIRE(a.stride % (c / 4) == 0, "%s: internal: stride %lld not a multiple of %d chunks", who, (long long)a.stride, c / 4);
  AVT_REQUIRE(ws && avt::aligned16(ws) && ws_size >= ws_bytes(a), "%s: workspace of %zu bytes needed (avt_bn_train_ws_bytes), got %zu",
              who, ws_bytes(a), ws_size);
  a.part = static_cast<double*>(ws);
  a.coef = reinterpret_cast<float*>(a.part + (size_t)a.groups * a.blocks * a.nq * 8);
  return AVT_OK;
}

}  // namespace

extern "C" int64_t avt_bn_train_ws_bytes(int64_t m, int c, int groups) {
  if (!shape_ok(m, c) || groups < 1 || m % groups) return -1;
  BnArgs a = {};
  a.groups = groups;
  layout(a, m / groups, c);
  return (int64_t)ws_bytes(a);
}

// pre_rows > 0: the statistics pass is NOT run — the producing convolution's epilogue left `pre_rows` rows of partial sums per group
// at the head of the workspace (avt_conv3d_igemm_x3_f32_stats / avt_pw_x3_f32_stats, in channel_sums' layout); the finalize
// launch sums those instead.  Workspace: groups * (pre_rows * min(C / 4, 256) * 8 doubles) + groups * 2 C floats
// (avt_bn_train_ws_bytes_pre).
static int bn_train_fwd_impl(const float* x, const float* res, float* y, int64_t m, int c, const float* gamma, const float* beta,
                             float eps, float momentum, int relu, int groups, void* ws, int64_t ws_size, float* save_mean,
                             float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                             void* relu_mask, int64_t ldy, int pre_rows, void* stream);

// rows (per group) the finalize launch walks for pre_rows producer rows: pre-reduced by kPreRows when there are many
static int pre_rows_final(int pre_rows, int unit) {
  const int r = pre_rows / unit;
  return r > 2 * kPreRows ? ((r + kPreRows - 1) / kPreRows) * unit : 0;  // 0: no pre-reduction
}

extern "C" int64_t avt_bn_train_ws_bytes_pre(int c, int groups, int pre_rows) {
  if (!shape_ok(1, c) || groups < 1 || pre_rows < 1) return -1;
  const int q = c / 4, nq = q < kT ? q : kT, unit = q > kT ? q / kT : 1;
  if (pre_rows % unit) return -1;
  const int64_t rows = (int64_t)pre_rows + pre_rows_final(pre_rows, unit);
  return (int64_t)groups * (rows * nq * 8 * (int64_t)sizeof(double) + (int64_t)2 * c * (int64_t)sizeof(float));
}

extern "C" int avt_bn_train_fwd(const float* x, const float* res, float* y, int64_t m, int c, const float* gamma, const float* beta,
                                float eps, float momentum, int relu, int groups, void* ws, int64_t ws_size, float* save_mean,
                                float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                void* relu_mask, int64_t ldy, void* stream) {
  return bn_train_fwd_impl(x, res, y, m, c, gamma, beta, eps, momentum, relu, groups, ws, ws_size, save_mean, save_invstd, running_mean,
                           running_var, num_batches_tracked, relu_mask, ldy, 0, stream);
}

extern "C" int avt_bn_train_fwd_pre(const float* x, const float* res, float* y, int64_t m, int c, const float* gamma, const float* beta,
                                    float eps, float momentum, int relu, int groups, void* ws, int64_t ws_size, float* save_mean,
                                    float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                    void* relu_mask, int64_t ldy, int pre_rows, void* stream) {
  AVT_REQUIRE(pre_rows > 0, "avt_bn_train_fwd_pre: pre_rows must be positive");
  return bn_train_fwd_impl(x, res, y, m, c, gamma, beta, eps, momentum, relu, groups, ws, ws_size, save_mean, save_invstd, running_mean,
                           running_var, num_batches_tracked, relu_mask, ldy, pre_rows, stream);
}

static int bn_train_fwd_impl(const float* x, const float* res, float* y, int64_t m, int c, const float* gamma, const float* beta,
                             float eps, float momentum, int relu, int groups, void* ws, int64_t ws_size, float* save_mean,
                             float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                             void* relu_mask, int64_t ldy, int pre_rows, void* stream) {
  AVT_REQUIRE(x && y && gamma && beta && save_mean && save_invstd && (!running_mean == !running_var), "avt_bn_train_fwd: NULL pointer");
  AVT_REQUIRE(avt::aligned16(x) && avt::aligned16(y) && (!res || avt::aligned16(res)), "avt_bn_train_fwd: rows must be 16-byte aligned");
  AVT_REQUIRE(avt::aligned16(beta) && avt::aligned16(save_mean), "avt_bn_train_fwd: beta / save_mean must be 16-byte aligned");
  BnArgs a = {};
  int rc;
  if (pre_rows > 0) {  // the streaming geometry as usual (the apply pass), the partials' geometry from the producer
    AVT_REQUIRE(shape_ok(m, c) && groups >= 1 && groups <= 65535 && m % groups == 0, "avt_bn_train_fwd_pre: bad shape");
    a.groups = groups;
    layout(a, m / groups, c);
    const int64_t need = avt_bn_train_ws_bytes_pre(c, groups, pre_rows);
    AVT_REQUIRE(ws && avt::aligned16(ws) && ws_size >= need && pre_rows % a.unit == 0,
                "avt_bn_train_fwd_pre: workspace of %lld bytes needed (avt_bn_train_ws_bytes_pre), got %lld", (long long)need, (long long)ws_size);
    a.part = static_cast<double*>(ws);
    a.coef = reinterpret_cast<float*>(a.part + (size_t)groups * (pre_rows + pre_rows_final(pre_rows, a.unit)) * a.nq * 8);
    rc = AVT_OK;
  } else {
    rc = geometry(a, "avt_bn_train_fwd", m, c, groups, ws, (size_t)(ws_size < 0 ? 0 : ws_size));
  }
  if (rc) return rc;
  a.x = x; a.res = res; a.out = y; a.gamma = gamma; a.beta = beta; a.eps = eps; a.momentum = momentum; a.relu = relu;
  a.save_mean = save_mean; a.save_invstd = save_invstd; a.running_mean = running_mean; a.running_var = running_var;
  a.tracked = reinterpret_cast<long long*>(num_batches_tracked);
  a.mask = relu ? static_cast<uint8_t*>(relu_mask) : nullptr;
  AVT_REQUIRE(ldy == 0 || (ldy >= c && ldy % 4 == 0), "avt_bn_train_fwd: ldy = %lld must be 0 (contiguous) or a multiple of 4 >= c", (long long)ldy);
  a.ld4 = (ldy ? ldy : c) / 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (pre_rows > 0) {
    BnArgs f = a;
    f.blocks = pre_rows;  // rows of partials per group, as the producer wrote them
    const int fin = pre_rows_final(pre_rows, a.unit);
    if (fin) {  // many rows: kPreRows of them into one first
      PreArgs p;
      p.src = a.part;
      p.dst = a.part + (size_t)groups * pre_rows * a.nq * 8;
      p.rows = pre_rows / a.unit;
      p.unit = a.unit;
      p.len = a.nq * 8;
      hipLaunchKernelGGL(bn_pre_reduce_kernel, dim3(fin / a.unit, a.unit * groups, p.len > kT ? p.len / kT : 1), dim3(kT), 0, st, p);
      f.part = p.dst;
      f.blocks = fin;
    }
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(c / 4), dim3(kT), 0, st, f);
  } else {
    hipLaunchKernelGGL(bn_fwd_stats_kernel, dim3(a.blocks, groups), dim3(kT), 0, st, a);
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(c / 4), dim3(kT), 0, st, a);
  }
  hipLaunchKernelGGL(bn_fwd_apply_kernel, dim3(a.blocks, groups), dim3(kT), 0, st, a);
  return avt::check_launch("avt_bn_train_fwd");
}

// avt_bn_train_bwd without its statistics pass: `g` is the ALREADY MASKED output gradient and `ws` holds pre_rows rows per group of
// the partial sums of g and g * xhat, both left by the launch that produced g (avt_conv3d_igemm_x3_f32_bwdstats).  The shortcut's
// gradient of a BatchNorm with a shortcut IS g: the caller aliases it, nothing is written for it here.
extern "C" int avt_bn_train_bwd_pre(const float* g, const float* x, int64_t m, int c, const float* gamma, const float* save_mean,
                                    const float* save_invstd, int groups, void* ws, int64_t ws_size, int pre_rows, float* dx, float* dgamma,
                                    float* dbeta, void* stream) {
  AVT_REQUIRE(g && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && pre_rows > 0, "avt_bn_train_bwd_pre: NULL pointer / no rows");
  AVT_REQUIRE(avt::aligned16(g) && avt::aligned16(x) && avt::aligned16(dx) && avt::aligned16(gamma) && avt::aligned16(save_mean) &&
                  avt::aligned16(save_invstd),
              "avt_bn_train_bwd_pre: rows and per-channel vectors must be 16-byte aligned");
  AVT_REQUIRE(shape_ok(m, c) && groups >= 1 && groups <= 65535 && m % groups == 0, "avt_bn_train_bwd_pre: bad shape");
  BnArgs a = {};
  a.groups = groups;
  layout(a, m / groups, c);
  const int64_t need = avt_bn_train_ws_bytes_pre(c, groups, pre_rows);
  AVT_REQUIRE(ws && avt::aligned16(ws) && ws_size >= need && pre_rows % a.unit == 0,
              "avt_bn_train_bwd_pre: workspace of %lld bytes needed (avt_bn_train_ws_bytes_pre), got %lld", (long long)need, (long long)ws_size);
  a.part = static_cast<double*>(ws);
  const int fin = pre_rows_final(pre_rows, a.unit);
  a.coef = reinterpret_cast<float*>(a.part + (size_t)groups * (pre_rows + fin) * a.nq * 8);
  a.dy = g; a.y = nullptr; a.x = x; a.gamma = gamma; a.beta = nullptr; a.mean = save_mean; a.invstd = save_invstd;
  a.relu = 0;  // (g is masked already)
  a.mask = nullptr;
  a.out = dx; a.dres = nullptr; a.dgamma = dgamma; a.dbeta = dbeta;
  a.ld4 = c / 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  BnArgs f = a;
  f.blocks = pre_rows;
  if (fin) {
    PreArgs p;
    p.src = a.part;
    p.dst = a.part + (size_t)groups * pre_rows * a.nq * 8;
    p.rows = pre_rows / a.unit;
    p.unit = a.unit;
    p.len = a.nq * 8;
    hipLaunchKernelGGL(bn_pre_reduce_kernel, dim3(fin / a.unit, a.unit * groups, p.len > kT ? p.len / kT : 1), dim3(kT), 0, st, p);
    f.part = p.dst;
    f.blocks = fin;
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c / 4), dim3(kT), 0, st, f);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(a.blocks, groups), dim3(kT), 0, st, a);
  return avt::check_launch("avt_bn_train_bwd_pre");
}

extern "C" int avt_bn_train_bwd(const float* dy, const float* y, const float* x, int64_t m, int c, const float* gamma, const float* beta,
                                const float* save_mean, const float* save_invstd, int relu, int groups, const void* relu_mask, void* ws,
                                int64_t ws_size, float* dx, float* dres, float* dgamma, float* dbeta, int64_t ld_dy, void* stream) {
  AVT_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta, "avt_bn_train_bwd: NULL pointer");
  AVT_REQUIRE(!relu || y || relu_mask || (beta && avt::aligned16(beta) && !dres),
              "avt_bn_train_bwd: a ReLU needs the forward's mask, its output y, or (no shortcut) beta to recompute the mask from x");
  AVT_REQUIRE(avt::aligned16(dy) && avt::aligned16(x) && avt::aligned16(dx) && (!y || avt::aligned16(y)) && (!dres || avt::aligned16(dres)) &&
                  avt::aligned16(gamma) && avt::aligned16(save_mean) && avt::aligned16(save_invstd),
              "avt_bn_train_bwd: rows and per-channel vectors must be 16-byte aligned");
  BnArgs a = {};
  const int rc = geometry(a, "avt_bn_train_bwd", m, c, groups, ws, (size_t)(ws_size < 0 ? 0 : ws_size));
  if (rc) return rc;
  a.dy = dy; a.y = relu ? y : nullptr; a.x = x; a.gamma = gamma; a.beta = beta; a.mean = save_mean; a.invstd = save_invstd;
  a.relu = relu;
  a.mask = relu ? const_cast<uint8_t*>(static_cast<const uint8_t*>(relu_mask)) : nullptr;
  a.out = dx; a.dres = dres; a.dgamma = dgamma; a.dbeta = dbeta;
  AVT_REQUIRE(ld_dy == 0 || (ld_dy >= c && ld_dy % 4 == 0), "avt_bn_train_bwd: ld_dy = %lld must be 0 (contiguous) or a multiple of 4 >= c", (long long)ld_dy);
  a.ld4 = (ld_dy ? ld_dy : c) / 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(a.blocks, groups), dim3(kT), 0, st, a);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c / 4), dim3(kT), 0, st, a);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(a.blocks, groups), dim3(kT), 0, st, a);
  return avt::check_launch("avt_bn_train_bwd");
}
