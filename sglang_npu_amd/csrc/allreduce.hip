// P2P all-reduce over IPC-mapped peer buffers (xGMI) for small / medium TP messages.
//
// Role in the reference: sgl-kernel/csrc/allreduce/custom_all_reduce_hip.cuh:261-350,498-568
// (one-shot / two-shot kernels over hipIpc buffers with Signal-flag barriers), driven by
// python/sglang/srt/distributed/device_communicators/custom_all_reduce.py:326-410 from
// GroupCoordinator.all_reduce (parallel_state.py:519-524).
//
// MI355X-first design ("Distributed comm backend"): the 8 GPUs of a node are fully
// connected by 7 xGMI links each, so instead of a ring every rank talks to all 7 peers at once:
//   one-shot  (<= 256 KiB): copy-in, barrier, every rank sums all peers' buffers;
//   two-shot  (larger):     copy-in, barrier, rank r reduces slice r from all peers (reduce-scatter by
//                           pull), barrier, every rank gathers the 8 reduced slices (all-gather by pull).
// Buffers are double-buffered by call parity, so no trailing barrier is needed: a half is rewritten only
// after the next call's first barrier proved that every peer left the previous call.  The parity is PER BLOCK (block b's
// own call count) and a block's barrier only proves that the peers' blocks b left the previous call, so this holds as long
// as a given byte of the staging area is always handled by the same block index.  Each kernel FAMILY therefore has its own
// staging area, flags and counters (round 3): 0 = plain all-reduce (vector i <-> block (i / 256) % 64 whatever
// the size), 1 = all-reduce fused with add + RMSNorm (row b <-> block b; the row length H is pinned per communicator by
// the launcher), 2 = QuickReduce (one regime per communicator).
// The sum order is rank 0..W-1 on every rank, so all ranks produce bit-identical results (and integer-
// valued payloads are exact, the property test_custom_allreduce.py:118-146 relies on).
// The call counter lives in device memory (kernel arguments are frozen under HIP-graph replay).
// Every spin is bounded and a timeout FAILS CLOSED: the block that gave up raises a sticky status word in
// host-mapped pinned memory (so the host polls it without a device sync), every block that sees it fills
// its part of the output with all-ones bytes (NaN in bf16 / fp16 / fp32) instead of summing buffers that
// were never synchronised, and every later call on that communicator does the same without spinning.
// GroupCoordinator.all_reduce checks the word before each launch and raises (distributed.py).
#include <string.h>

#include "common.h"
#include "partials.h"

namespace sglm {
namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 64;
constexpr int kThreads = 256;
constexpr unsigned kSpinLimit = 1u << 27;

struct ArComm {
  int rank, world;
  size_t max_bytes;      // payload capacity per half
  char* base;            // own allocation: [signals | counter | timeout | pad][data half 0][data half 1]
  size_t data_off, half_bytes;
  char* peer[kMaxRanks]; // mapped bases of every rank (own entry = base)
  bool opened[kMaxRanks];
  uint32_t* status_host; // pinned, device-mapped: 0 = healthy, 1 = a barrier timed out (sticky)
  uint32_t* status_dev;
  int64_t norm_hidden;   // row length the fused all-reduce + norm family is bound to (0 = not yet used), see kFamNorm
};

struct ArArgs {
  char* peer[kMaxRanks];
  size_t data_off, half_bytes;
  int rank, world;
  uint32_t* status;      // sticky failure word in this rank's own device buffer (read on every call: must be cheap)
  uint32_t* status_host; // its mirror in host-mapped pinned memory (written once, when a wait gives up)
  unsigned spin_limit;
  int family;            // staging area / flags / counters of this kernel family (kFam*); data_off already points into it
  int test_delay;        // test hook (sgl_mi355_ar_set_test_delay): 0 in production
};

// header: signals uint32 [kFamilies][2 slots][kMaxRanks][kMaxBlocks] at offset 0; counters uint32 [kFamilies][kMaxBlocks]
// after them; then the sticky status word.  Data: [kFamilies][2 halves][half_bytes] from kHeaderBytes on.
constexpr int kFamilies = 3;
enum { kFamPlain = 0, kFamNorm = 1, kFamQuick = 2 };
__device__ __forceinline__ uint32_t* sig_ptr(char* base, int family, int slot, int from, int block) {
  return reinterpret_cast<uint32_t*>(base) + ((family * 2 + slot) * kMaxRanks + from) * kMaxBlocks + block;
}
constexpr size_t kSigBytes = (size_t)kFamilies * 2 * kMaxRanks * kMaxBlocks * 4;
constexpr size_t kCounterOff = kSigBytes;
constexpr size_t kStatusOff = kCounterOff + (size_t)kFamilies * kMaxBlocks * 4;
constexpr size_t kHeaderBytes = 16384;
static_assert(kStatusOff + 64 <= kHeaderBytes, "header layout");
__device__ __forceinline__ uint32_t* counter_ptr(const ArArgs& a) {
  return reinterpret_cast<uint32_t*>(a.peer[a.rank] + kCounterOff) + a.family * kMaxBlocks + blockIdx.x;
}

// Test hook for the staging-store / flag ordering (round 4's defect, see block_barrier): the LAST wave of every workgroup idles
// `test_delay` x ~3.4 us (s_sleep 127 = 8128 cycles) in front of its phase-A staging stores, so that on every call those stores
// are the last thing issued before the workgroup barrier -- in flight while wave 0 is ready to publish the flags the moment the
// barrier opens.  What was a rare interleaving (late stores behind the split-K slab loads of the PARTIALS form) becomes the
// order of every workgroup of every call (tests/test_custom_allreduce_gpu.py).  One scalar compare when the hook is off.
__device__ __forceinline__ void test_delay_last_wave(const ArArgs& a) {
  if (a.test_delay > 0 && (int)(threadIdx.x >> 6) == (int)(blockDim.x >> 6) - 1)
    for (int i = 0; i < a.test_delay; ++i) __builtin_amdgcn_s_sleep(127);
}

// Returns false when this rank's communicator is (or has just become) failed: the caller must not read peer data.
__device__ __forceinline__ bool block_barrier(const ArArgs& a, int slot, uint32_t val) {
  __shared__ int failed;
  if (threadIdx.x == 0) failed = 0;
  // Every thread's stores into the staging area must be ACKNOWLEDGED before one of the first `world` threads publishes the
  // flag: the workgroup barrier below orders execution, not the completion of other waves' stores, and a fence by the
  // signalling thread covers what has reached the cache, not what another wave still has in flight.  (Round 4: the
  // 8-ranks-in-one-process test saw one stale 16-byte vector of a peer's row once in several full runs -- the PARTIALS
  // form of the fused norm, whose phase-A stores are issued late, behind the split-K slab loads.)
#ifndef SGLM_AR_NO_STAGING_WAIT  // (defined only in the variant build that shows the regression test catching the defect)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __syncthreads();
  if (threadIdx.x < a.world) {
    const int t = threadIdx.x;
    __threadfence_system();
    __hip_atomic_store(sig_ptr(a.peer[t], a.family, slot, a.rank, blockIdx.x), val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t* mine = sig_ptr(a.peer[a.rank], a.family, slot, t, blockIdx.x);
    unsigned spins = 0;
    while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < val) {
      // another block of this rank (or an earlier call) already gave up: do not wait out the full limit again
      if (++spins > a.spin_limit ||
          ((spins & 1023u) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
        __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.status_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        failed = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return failed == 0;
}

// all-ones bytes: NaN in every supported dtype
__device__ __forceinline__ void poison(uint4* out, int64_t n16, int64_t tid, int64_t nthr) {
  const uint4 bad{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  for (int64_t i = tid; i < n16; i += nthr) out[i] = bad;
}

template <int DTYPE>
struct Acc8 {
  using H = Half16<DTYPE>;
  static __device__ __forceinline__ void add(float* f, const uint4& v) {
    const typename H::x8 x = __builtin_bit_cast(typename H::x8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] += H::to_f32(x[j]);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    typename H::x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = H::from_f32(f[j]);
    return __builtin_bit_cast(uint4, o);
  }
};
struct AccF32 {
  static __device__ __forceinline__ void add(float* f, const uint4& v) {
    const f32x4 x = __builtin_bit_cast(f32x4, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] += x[j];
  }
  static __device__ __forceinline__ uint4 pack(const float* f) { return __builtin_bit_cast(uint4, f32x4{f[0], f[1], f[2], f[3]}); }
};

// A = accumulator policy; n16 = payload size in 16-byte vectors
template <typename A>
__global__ __launch_bounds__(kThreads) void all_reduce_kernel(ArArgs a, const uint4* __restrict__ inp, uint4* __restrict__ out,
                                                              int64_t n16, int two_shot) {
  uint32_t* counter = counter_ptr(a);
  const uint32_t call = *counter + 1;  // 1, 2, 3, ... ; every rank runs the same sequence of calls
  const int half = call & 1;
  const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t nthr = (int64_t)gridDim.x * kThreads;
  char* my_data = a.peer[a.rank] + a.data_off + half * a.half_bytes;
  // result area of the two-shot variant sits behind the payload area of the same half
  const size_t res_off = a.half_bytes / 2;

  // a communicator that has timed out once stays failed: no spinning, poisoned output
  bool ok = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0;
  if (ok) {
    // phase A: publish my input
    test_delay_last_wave(a);
    for (int64_t i = tid; i < n16; i += nthr) reinterpret_cast<uint4*>(my_data)[i] = inp[i];
    ok = block_barrier(a, 0, call);
  }
  if (!ok) {
    poison(out, n16, tid, nthr);
    if (threadIdx.x == 0) *counter = call;
    return;
  }

  if (!two_shot) {
    for (int64_t i = tid; i < n16; i += nthr) {
      float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int p = 0; p < a.world; ++p)
        A::add(f, reinterpret_cast<const uint4*>(a.peer[p] + a.data_off + half * a.half_bytes)[i]);
      out[i] = A::pack(f);
    }
  } else {
    // phase B: reduce my slice from every peer
    const int64_t per = (n16 + a.world - 1) / a.world;
    const int64_t lo = (int64_t)a.rank * per, hi = (lo + per) < n16 ? (lo + per) : n16;
    // The flag barriers pair block b of this rank with block b of every peer, so a block may only read
    // what the SAME block index wrote elsewhere: element i is always handled by global thread i % nthr
    // (phase A wrote it that way, phase C reads it that way), so walk the slice with that phase.
    const int64_t first = lo + (((tid - lo) % nthr) + nthr) % nthr;
    for (int64_t i = first; i < hi; i += nthr) {
      float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int p = 0; p < a.world; ++p)
        A::add(f, reinterpret_cast<const uint4*>(a.peer[p] + a.data_off + half * a.half_bytes)[i]);
      reinterpret_cast<uint4*>(my_data + res_off)[i] = A::pack(f);
    }
    if (!block_barrier(a, 1, call)) {
      poison(out, n16, tid, nthr);
      if (threadIdx.x == 0) *counter = call;
      return;
    }
    // phase C: gather every rank's reduced slice
    for (int64_t i = tid; i < n16; i += nthr) {
      const int owner = (int)(i / per);
      out[i] = reinterpret_cast<const uint4*>(a.peer[owner] + a.data_off + half * a.half_bytes + res_off)[i];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *counter = call;
}

// ------------------------------------------------------------------------------------------
// QuickReduce-class all-reduce: two-shot with BLOCK-SCALED INTEGER TRANSPORT for prefill-size messages.
//
// Role in the reference: sgl-kernel/csrc/allreduce/quick_all_reduce.cuh (AllReduceTwoshot over CodecQ8 / CodecQ6 /
// CodecQ4, :50-480) behind device_communicators/quick_all_reduce.py:56-260 -- ROCm only, opt-in through
// ROCM_QUICK_REDUCE_QUANTIZATION, for messages the custom all-reduce does not take (its 16 MiB cap) or is slower on.
// Codec semantics restated from there (quick_all_reduce.cuh:71-135 Q4, :210-290 Q6, :346-440 Q8): blocks of 32
// consecutive values share one half-precision scale; with R = 2^(bits-1),
//     dec = half(-absmax / R)            (the NEGATIVE scale of the reference: +absmax maps to -R, exactly representable)
//     enc = 1 / (dec + eps)              (eps = the smallest positive half, 2^-24)
//     q   = clamp(rint(x * enc), -R, R - 1) + R      (unsigned, `bits` wide)
//     x'  = (q - R) * dec
// The reference does this arithmetic in packed half / bfloat16; here it is fp32 with the scale rounded to half (what is
// transported), which only makes the rounding decisions more faithful.  The wire layout is this file's own (all ranks
// run this code): per transmitted region the packed integers of unit u (8 values, `bits` bytes) at u * bits, then the
// scales, one half per 4 units.
//
// Protocol = the two-shot all-reduce above on the QUANTISED image, chunked by the host so that any message size runs
// through the fixed IPC staging area (the reference's tiles, quick_all_reduce.cuh kTileSize):
//   A  quantise my chunk into my IPC buffer;                                              flag barrier 0
//   B  owner r: dequantise slice r of every rank (rank order, fp32 sum), re-quantise, store in its result area;   barrier 1
//   C  everyone: dequantise the W reduced slices into `out`.
// All ranks decode the same bytes in phase C, so every rank returns bit-identical results; the result differs from the
// exact sum by at most two quantisation steps per element (test_quick_allreduce.py:162-163 allows atol 1.25 W, rtol 0.5 W).
template <int BITS>
struct QPack {  // 8 unsigned `BITS`-wide values <-> BITS bytes
  static __device__ __forceinline__ void store(char* p, const uint32_t* q) {
    if constexpr (BITS == 8) {
      *reinterpret_cast<uint2*>(p) = uint2{q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24), q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24)};
    } else if constexpr (BITS == 4) {
      uint32_t w = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) w |= q[j] << (4 * j);
      *reinterpret_cast<uint32_t*>(p) = w;
    } else {
      uint64_t w = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) w |= (uint64_t)q[j] << (6 * j);
      uint16_t* d = reinterpret_cast<uint16_t*>(p);
      d[0] = (uint16_t)w; d[1] = (uint16_t)(w >> 16); d[2] = (uint16_t)(w >> 32);
    }
  }
  static __device__ __forceinline__ void load(const char* p, uint32_t* q) {
    if constexpr (BITS == 8) {
      const uint2 w = *reinterpret_cast<const uint2*>(p);
#pragma unroll
      for (int j = 0; j < 4; ++j) { q[j] = (w.x >> (8 * j)) & 0xffu; q[4 + j] = (w.y >> (8 * j)) & 0xffu; }
    } else if constexpr (BITS == 4) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = (w >> (4 * j)) & 0xfu;
    } else {
      const uint16_t* d = reinterpret_cast<const uint16_t*>(p);
      const uint64_t w = (uint64_t)d[0] | ((uint64_t)d[1] << 16) | ((uint64_t)d[2] << 32);
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = (uint32_t)(w >> (6 * j)) & 0x3fu;
    }
  }
};

// quantise the 8 values of this thread; the 4 threads of a 32-value block are 4 consecutive lanes (aligned)
template <int BITS>
__device__ __forceinline__ void q_encode(const float* f, char* qdst, _Float16* sdst, bool leader) {
  constexpr float R = (float)(1 << (BITS - 1));
  float am = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(f[j]));
  am = fmaxf(am, __shfl_xor(am, 1));
  am = fmaxf(am, __shfl_xor(am, 2));
  const _Float16 dech = (_Float16)(am * (-1.0f / R));
  const float dec = (float)dech;
  const float enc = 1.0f / (dec + 5.9604644775390625e-8f);
  uint32_t q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = (uint32_t)(int)(fminf(fmaxf(rintf(f[j] * enc), -R), R - 1.0f) + R);
  QPack<BITS>::store(qdst, q);
  if (leader) *sdst = dech;
}
template <int BITS>
__device__ __forceinline__ void q_decode_add(float* f, const char* qsrc, const _Float16* ssrc) {
  constexpr int R = 1 << (BITS - 1);
  uint32_t q[8];
  QPack<BITS>::load(qsrc, q);
  const float dec = (float)*ssrc;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] += (float)((int)q[j] - R) * dec;
}

template <int DTYPE, int BITS>
__global__ __launch_bounds__(kThreads) void quick_reduce_kernel(ArArgs a, const uint4* __restrict__ inp, uint4* __restrict__ out,
                                                                int64_t n_units) {
  using A = Acc8<DTYPE>;
  uint32_t* counter = counter_ptr(a);
  const uint32_t call = *counter + 1;
  const int half = call & 1;
  const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t nthr = (int64_t)gridDim.x * kThreads;
  const size_t area = a.data_off + half * a.half_bytes;  // this call's quantised-input area (per rank)
  const size_t res_off = a.half_bytes / 2;               // the owner's re-quantised slice, same indexing
  const size_t sc_off = ((size_t)n_units * BITS + 15) & ~(size_t)15;
  char* mine = a.peer[a.rank] + area;

  bool ok = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0;
  if (ok) {
    // phase A: my chunk, quantised (n_units % 4 == 0 and nthr % 4 == 0: a block's 4 threads move together)
    test_delay_last_wave(a);
    for (int64_t u = tid; u < n_units; u += nthr) {
      float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      A::add(f, inp[u]);
      q_encode<BITS>(f, mine + u * BITS, reinterpret_cast<_Float16*>(mine + sc_off) + (u >> 2), (u & 3) == 0);
    }
    ok = block_barrier(a, 0, call);
  }
  if (!ok) {
    poison(out, n_units, tid, nthr);
    if (threadIdx.x == 0) *counter = call;
    return;
  }
  // phase B: my slice of every rank -> fp32 sum in rank order -> quantised again into my result area.  As in the
  // exact kernel, unit u is always handled by global thread u % nthr (the flag barriers pair equal block indices).
  int64_t per = (n_units + a.world - 1) / a.world;
  per = (per + 3) & ~(int64_t)3;
  const int64_t lo = (int64_t)a.rank * per < n_units ? (int64_t)a.rank * per : n_units;
  const int64_t hi = (lo + per) < n_units ? (lo + per) : n_units;
  const int64_t first = lo + (((tid - lo) % nthr) + nthr) % nthr;
  for (int64_t u = first; u < hi; u += nthr) {
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = 0; p < a.world; ++p) {
      const char* src = a.peer[p] + area;
      q_decode_add<BITS>(f, src + u * BITS, reinterpret_cast<const _Float16*>(src + sc_off) + (u >> 2));
    }
    q_encode<BITS>(f, mine + res_off + u * BITS, reinterpret_cast<_Float16*>(mine + res_off + sc_off) + (u >> 2), (u & 3) == 0);
  }
  if (!block_barrier(a, 1, call)) {
    poison(out, n_units, tid, nthr);
    if (threadIdx.x == 0) *counter = call;
    return;
  }
  // phase C: all reduced slices
  for (int64_t u = tid; u < n_units; u += nthr) {
    const int owner = (int)(u / per);
    const char* src = a.peer[owner] + area + res_off;
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    q_decode_add<BITS>(f, src + u * BITS, reinterpret_cast<const _Float16*>(src + sc_off) + (u >> 2));
    out[u] = A::pack(f);
  }
  __syncthreads();
  if (threadIdx.x == 0) *counter = call;
}

// ------------------------------------------------------------------------------------------
// All-reduce + residual add + RMSNorm (+ per-token FP8 quant) in ONE kernel: the consumer of a row-parallel GEMM under
// tensor parallelism (o_proj / down_proj -> the next norm).  Upstream seam: RowParallelLinear.forward(...,
// can_fuse_mlp_allreduce=True) skips its collective (layers/linear.py:1285-1303) and
// RMSNorm.forward_with_allreduce_fusion (layers/layernorm.py:191-216, called from layers/communicator.py:190-199,425-441)
// performs it -- today only through flashinfer on sm100.  At Llama-3-70B TP = 8 a decode step has 160 such pairs on
// [64, 8192] bf16 (1 MiB): a separate collective launch plus a norm launch each would be ~10 us of pure latency.
//
// Protocol = the two-shot all-reduce above with the reduce-scatter cut by COLUMNS and the all-gather fused into the
// norm: block b owns row b (rows b + gridDim.x, ... for longer inputs) in every phase, on every rank:
//   A  copy my row into my IPC buffer;                                     flag barrier 0
//   B  rank r sums columns [r H/W, (r+1) H/W) of the row over all peers (rank order, fp32), rounds to the 16-bit
//      dtype -- exactly the all-reduce's output value -- into its result area; flag barrier 1
//   C  gather the W reduced column slices of the row, add the residual (fp32 sum, residual updated with its rounding),
//      RMSNorm on the unrounded sum, optional per-token FP8 quant -- the arithmetic of rmsnorm_kernel (elementwise.hip).
// Small inputs (one-shot: T H 2 B <= 256 KiB) skip phase B: phase C sums the peers' full rows itself.
// Bit-identical to custom_all_reduce followed by fused_add_rmsnorm / rmsnorm_quant_fp8 on every rank.
__device__ __forceinline__ float ar_block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  __syncthreads();
  return r;
}
__device__ __forceinline__ float ar_block_max(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  const int nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r = fmaxf(r, red[i]);
  __syncthreads();
  return r;
}

// kNormThreads threads x kNormVPT 8-element vectors per row: the launcher picks the SAME pair as launch_rmsnorm
// (elementwise.hip) would for the shape, so that the sum of squares is reduced in the same order and the result is
// bit-identical to the unfused norm.
// PARTIALS: the rank's addend is a split-K GEMM in flight (PartialSrc); a separate instantiation so that the plain form
// keeps its register footprint (the 8-ranks-on-one-GPU protocol test needs all 8 x 64 workgroups resident at once).
template <int DTYPE, int kNormVPT, int kNormThreads, bool PARTIALS>
__global__ __launch_bounds__(kNormThreads) void ar_add_rmsnorm_kernel(
    ArArgs a, const typename Half16<DTYPE>::T* inp /* may alias out */, typename Half16<DTYPE>::T* __restrict__ residual,
    const typename Half16<DTYPE>::T* __restrict__ weight, typename Half16<DTYPE>::T* out,
    uint8_t* __restrict__ out_q, float* __restrict__ out_s, int T, int H, float eps, int one_shot, PartialSrc ps) {
  using Hh = Half16<DTYPE>;
  using x8 = typename Hh::x8;
  __shared__ float red[kNormThreads / 64];
  uint32_t* counter = counter_ptr(a);
  const uint32_t call = *counter + 1;
  const int half = call & 1;
  char* my_data = a.peer[a.rank] + a.data_off + half * a.half_bytes;
  const size_t res_off = a.half_bytes / 2;
  const int nv = H >> 3;            // vectors per row
  const int nvs = nv / a.world;     // vectors per column slice
  const int tid = threadIdx.x;

  auto fail = [&]() {  // a peer never arrived: NaN rows instead of a norm over unsynchronised buffers
    for (int row = blockIdx.x; row < T; row += gridDim.x)
      for (int vi = tid; vi < nv; vi += kNormThreads) {
        const uint4 bad{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (out) reinterpret_cast<uint4*>(out + (int64_t)row * H)[vi] = bad;
        if (out_q) reinterpret_cast<uint2*>(out_q + (int64_t)row * H)[vi] = uint2{0x7f7f7f7fu, 0x7f7f7f7fu};  // e4m3 NaN
        if (out_s && vi == 0) out_s[row] = __builtin_nanf("");
      }
    if (tid == 0) *counter = call;
  };

  bool ok = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0;
  if (ok) {
    // phase A (ps.partials: my rows are still a split-K GEMM -- its epilogue runs here, rounded to the 16-bit dtype exactly as
    // fp8_gemm_finalize_kernel would have, so the sum over ranks sees the same addends)
    test_delay_last_wave(a);
    for (int row = blockIdx.x; row < T; row += gridDim.x)
      for (int vi = tid; vi < nv; vi += kNormThreads)
        if constexpr (PARTIALS)
          reinterpret_cast<uint4*>(my_data + (int64_t)row * H * 2)[vi] = __builtin_bit_cast(uint4, gemm_row8<DTYPE>(ps, row, 8 * vi));
        else
          reinterpret_cast<uint4*>(my_data + (int64_t)row * H * 2)[vi] = reinterpret_cast<const uint4*>(inp + (int64_t)row * H)[vi];
    ok = block_barrier(a, 0, call);
  }
  if (!ok) {
    fail();
    return;
  }
  if (!one_shot) {
    // phase B: my column slice of my rows, summed in rank order, rounded like the all-reduce's output
    for (int row = blockIdx.x; row < T; row += gridDim.x)
      for (int vs = tid; vs < nvs; vs += kNormThreads) {
        const int vi = a.rank * nvs + vs;
        float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = 0; p < a.world; ++p)
          Acc8<DTYPE>::add(f, reinterpret_cast<const uint4*>(a.peer[p] + a.data_off + half * a.half_bytes + (int64_t)row * H * 2)[vi]);
        reinterpret_cast<uint4*>(my_data + res_off + (int64_t)row * H * 2)[vi] = Acc8<DTYPE>::pack(f);
      }
    if (!block_barrier(a, 1, call)) {
      fail();
      return;
    }
  }
  // phase C: gather + add + norm (+ quant), one row at a time
  for (int row = blockIdx.x; row < T; row += gridDim.x) {
    float v[kNormVPT][8];
    x8 wq[kNormVPT];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < kNormVPT; ++i) {
      const int vi = tid + kNormThreads * i;
      if (vi < nv) {
        wq[i] = reinterpret_cast<const x8*>(weight)[vi];
        const x8 rv = reinterpret_cast<const x8*>(residual + (int64_t)row * H)[vi];
        x8 xv;
        if (one_shot) {
          float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int p = 0; p < a.world; ++p)
            Acc8<DTYPE>::add(f, reinterpret_cast<const uint4*>(a.peer[p] + a.data_off + half * a.half_bytes + (int64_t)row * H * 2)[vi]);
          xv = __builtin_bit_cast(x8, Acc8<DTYPE>::pack(f));
        } else {
          const int owner = vi / nvs;
          xv = __builtin_bit_cast(x8, reinterpret_cast<const uint4*>(a.peer[owner] + a.data_off + half * a.half_bytes + res_off +
                                                                    (int64_t)row * H * 2)[vi]);
        }
        x8 nr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[i][j] = Hh::to_f32(xv[j]) + Hh::to_f32(rv[j]);  // the norm continues on the unrounded fp32 sum
          nr[j] = Hh::from_f32(v[i][j]);
          ss += v[i][j] * v[i][j];
        }
        reinterpret_cast<x8*>(residual + (int64_t)row * H)[vi] = nr;
      }
    }
    ss = ar_block_sum(ss, red);
    const float inv = 1.0f / sqrtf(ss / (float)H + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < kNormVPT; ++i) {
      const int vi = tid + kNormThreads * i;
      if (vi < nv) {
        x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o[j] = Hh::from_f32(v[i][j] * inv * Hh::to_f32(wq[i][j]));
          v[i][j] = Hh::to_f32(o[j]);
          amax = fmaxf(amax, fabsf(v[i][j]));
        }
        if (out) reinterpret_cast<x8*>(out + (int64_t)row * H)[vi] = o;
      }
    }
    if (out_q) {
      amax = ar_block_max(amax, red);
      const float scale = amax / 448.0f;
      if (tid == 0) out_s[row] = scale;
      const float sinv = scale == 0.f ? 0.f : 1.0f / scale;
#pragma unroll
      for (int i = 0; i < kNormVPT; ++i) {
        const int vi = tid + kNormThreads * i;
        if (vi < nv) {
          float f[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(v[i][j] * sinv, -448.0f), 448.0f);
          int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
          int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
          hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
          reinterpret_cast<uint2*>(out_q + (int64_t)row * H)[vi] = uint2{(unsigned)lo, (unsigned)hi};
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) *counter = call;
}

}  // namespace
}  // namespace sglm

using namespace sglm;

extern "C" int sgl_mi355_ar_create(int rank, int world_size, int64_t max_bytes, void** comm_out) {
  SGLM_CHECK_ARG(comm_out != nullptr, "ar_create: null output");
  SGLM_CHECK_ARG(world_size >= 1 && world_size <= kMaxRanks && rank >= 0 && rank < world_size,
                 "ar_create: bad rank/world (%d/%d), world must be <= %d", rank, world_size, kMaxRanks);
  SGLM_CHECK_ARG(max_bytes > 0 && max_bytes % 16 == 0 && max_bytes <= (1ll << 30), "ar_create: bad max_bytes");
  ArComm* c = new ArComm();
  c->rank = rank; c->world = world_size; c->max_bytes = (size_t)max_bytes; c->norm_hidden = 0;
  c->half_bytes = 2 * (size_t)max_bytes;  // payload + two-shot result area
  c->data_off = kHeaderBytes;
  const size_t total = kHeaderBytes + (size_t)kFamilies * 2 * c->half_bytes;
  void* p = nullptr;
  // uncached (fine-grained) memory: flags and payload are read by peer GPUs while kernels run
  hipError_t e = hipExtMallocWithFlags(&p, total, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    delete c;
    return check_hip(e, "hipExtMallocWithFlags(hipDeviceMallocUncached)");
  }
  e = hipMemset(p, 0, total);
  if (e != hipSuccess) {
    (void)hipFree(p);
    delete c;
    return check_hip(e, "hipMemset");
  }
  c->base = (char*)p;
  void* st = nullptr;
  e = hipHostMalloc(&st, 64, hipHostMallocMapped);
  if (e == hipSuccess) {
    *(volatile uint32_t*)st = 0;
    void* st_dev = nullptr;
    e = hipHostGetDevicePointer(&st_dev, st, 0);
    c->status_host = (uint32_t*)st;
    c->status_dev = (uint32_t*)st_dev;
  }
  if (e != hipSuccess) {
    if (st) (void)hipHostFree(st);
    (void)hipFree(p);
    delete c;
    return check_hip(e, "hipHostMalloc(mapped status word)");
  }
  for (int i = 0; i < kMaxRanks; ++i) { c->peer[i] = nullptr; c->opened[i] = false; }
  c->peer[rank] = c->base;
  *comm_out = c;
  return 0;
}

extern "C" int sgl_mi355_ar_get_ipc_handle(void* comm, void* handle_out /* 64 bytes */) {
  SGLM_CHECK_ARG(comm && handle_out, "ar_get_ipc_handle: null argument");
  ArComm* c = (ArComm*)comm;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
  return check_hip(hipIpcGetMemHandle((hipIpcMemHandle_t*)handle_out, c->base), "hipIpcGetMemHandle");
}

extern "C" int sgl_mi355_ar_open_peers(void* comm, const void* all_handles /* world x 64 bytes */) {
  SGLM_CHECK_ARG(comm && all_handles, "ar_open_peers: null argument");
  ArComm* c = (ArComm*)comm;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)all_handles + 64 * r, 64);
    void* p = nullptr;
    SGLM_CHECK_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    c->peer[r] = (char*)p;
    c->opened[r] = true;
  }
  return 0;
}

extern "C" int sgl_mi355_ar_set_peers_local(void* comm, void* const* comms /* world communicators of THIS process */) {
  SGLM_CHECK_ARG(comm && comms, "ar_set_peers_local: null argument");
  ArComm* c = (ArComm*)comm;
  for (int r = 0; r < c->world; ++r) {
    SGLM_CHECK_ARG(comms[r] != nullptr, "ar_set_peers_local: communicator %d is null", r);
    const ArComm* o = (const ArComm*)comms[r];
    SGLM_CHECK_ARG(o->rank == r && o->world == c->world && o->max_bytes == c->max_bytes,
                   "ar_set_peers_local: communicator %d does not belong to this group", r);
    c->peer[r] = o->base;
  }
  return 0;
}

static int g_test_delay = 0;
// Test hook: see test_delay_last_wave.  iters x ~3.4 us of idling per call and workgroup; 0 (the default) switches it off.
extern "C" int sgl_mi355_ar_set_test_delay(int64_t iters) {
  SGLM_CHECK_ARG(iters >= 0 && iters <= 4096, "ar_set_test_delay: 0 .. 4096");
  g_test_delay = (int)iters;
  return 0;
}

static unsigned g_spin_limit = kSpinLimit;
extern "C" int sgl_mi355_ar_set_spin_limit(int64_t spins) {
  SGLM_CHECK_ARG(spins > 0 && spins <= (int64_t)kSpinLimit, "ar_set_spin_limit: 1 .. %u", kSpinLimit);
  g_spin_limit = (unsigned)spins;
  return 0;
}

// Forget the row length the fused all-reduce + norm staging area is bound to (see kFamNorm).  Only while NO call of this
// communicator is in flight on ANY rank (e.g. after a group barrier that follows a device synchronisation): the start-up
// self-check uses a row length of its own before the model's first call.
extern "C" int sgl_mi355_ar_rebind_norm(void* comm) {
  SGLM_CHECK_ARG(comm, "ar_rebind_norm: null communicator");
  ((ArComm*)comm)->norm_hidden = 0;
  return 0;
}

extern "C" int sgl_mi355_ar_all_reduce(void* comm, const void* inp, void* out, int64_t nbytes, int dtype /* 0 bf16, 1 fp16, 2 fp32 */,
                                        void* stream) {
  SGLM_CHECK_ARG(comm, "ar_all_reduce: null communicator");
  ArComm* c = (ArComm*)comm;
  SGLM_CHECK_ARG(nbytes >= 0 && nbytes % 16 == 0, "ar_all_reduce: size (%ld B) must be a multiple of 16 bytes", (long)nbytes);
  SGLM_CHECK_ARG((size_t)nbytes <= c->max_bytes, "ar_all_reduce: %ld B exceeds the registered capacity %ld B", (long)nbytes, (long)c->max_bytes);
  SGLM_CHECK_ARG(dtype >= 0 && dtype <= 2, "ar_all_reduce: bad dtype %d", dtype);
  if (nbytes == 0) return 0;
  SGLM_CHECK_ARG(inp && out, "ar_all_reduce: null tensor pointer");
  for (int r = 0; r < c->world; ++r) SGLM_CHECK_ARG(c->peer[r] != nullptr, "ar_all_reduce: peer %d not opened", r);
  ArArgs a{};
  for (int r = 0; r < c->world; ++r) a.peer[r] = c->peer[r];
  a.family = kFamPlain;
  a.data_off = c->data_off + (size_t)kFamPlain * 2 * c->half_bytes; a.half_bytes = c->half_bytes; a.rank = c->rank; a.world = c->world;
  a.status = reinterpret_cast<uint32_t*>(c->base + kStatusOff); a.status_host = c->status_dev; a.spin_limit = g_spin_limit; a.test_delay = g_test_delay;
  const int64_t n16 = nbytes / 16;
  const int two_shot = nbytes > 256 * 1024 && c->world > 1;
  int blocks = (int)((n16 + kThreads - 1) / kThreads);
  blocks = blocks < 1 ? 1 : (blocks > kMaxBlocks ? kMaxBlocks : blocks);
  hipStream_t s = as_stream(stream);
  if (dtype == 0)
    hipLaunchKernelGGL((all_reduce_kernel<Acc8<SGL_MI355_BF16>>), dim3(blocks), dim3(kThreads), 0, s, a, (const uint4*)inp, (uint4*)out, n16, two_shot);
  else if (dtype == 1)
    hipLaunchKernelGGL((all_reduce_kernel<Acc8<SGL_MI355_FP16>>), dim3(blocks), dim3(kThreads), 0, s, a, (const uint4*)inp, (uint4*)out, n16, two_shot);
  else
    hipLaunchKernelGGL((all_reduce_kernel<AccF32>), dim3(blocks), dim3(kThreads), 0, s, a, (const uint4*)inp, (uint4*)out, n16, two_shot);
  return check_hip(hipGetLastError(), "all_reduce_kernel launch");
}

// QuickReduce entry point (qr_all_reduce of the reference, _custom_ops.py / quick_all_reduce.cu:60-110): any size that is
// a multiple of 64 bytes (whole 32-value blocks), fp16 / bf16; regime = QuickReduceRegime (quick_all_reduce.py:47-52):
// 0 FP (exact two-shot, chunked), 1 INT8, 2 INT6, 3 INT4.  The message runs through the communicator's staging area in
// chunks, one kernel launch per chunk on `stream`.
extern "C" int sgl_mi355_ar_quick_all_reduce(void* comm, const void* inp, void* out, int64_t nbytes, int dtype /* 0 bf16, 1 fp16 */,
                                              int regime, void* stream) {
  SGLM_CHECK_ARG(comm, "ar_quick_all_reduce: null communicator");
  ArComm* c = (ArComm*)comm;
  SGLM_CHECK_ARG(nbytes >= 0 && nbytes % 64 == 0, "ar_quick_all_reduce: size (%ld B) must be a multiple of 64 bytes", (long)nbytes);
  SGLM_CHECK_ARG(dtype == 0 || dtype == 1, "ar_quick_all_reduce: dtype must be bf16 (0) or fp16 (1), got %d", dtype);
  SGLM_CHECK_ARG(regime >= 0 && regime <= 3, "ar_quick_all_reduce: regime must be 0 (FP), 1 (INT8), 2 (INT6) or 3 (INT4), got %d", regime);
  if (nbytes == 0) return 0;
  SGLM_CHECK_ARG(inp && out, "ar_quick_all_reduce: null tensor pointer");
  for (int r = 0; r < c->world; ++r) SGLM_CHECK_ARG(c->peer[r] != nullptr, "ar_quick_all_reduce: peer %d not opened", r);
  ArArgs a{};
  for (int r = 0; r < c->world; ++r) a.peer[r] = c->peer[r];
  a.family = kFamQuick;
  a.data_off = c->data_off + (size_t)kFamQuick * 2 * c->half_bytes; a.half_bytes = c->half_bytes; a.rank = c->rank; a.world = c->world;
  a.status = reinterpret_cast<uint32_t*>(c->base + kStatusOff); a.status_host = c->status_dev; a.spin_limit = g_spin_limit; a.test_delay = g_test_delay;
  hipStream_t s = as_stream(stream);
  const int bits = regime == 0 ? 16 : regime == 1 ? 8 : regime == 2 ? 6 : 4;
  // units of 8 values (16 B of input) per chunk: packed integers + one half per 4 units must fit the area of max_bytes
  int64_t cap_units = regime == 0 ? (int64_t)(c->max_bytes / 16) : (int64_t)((c->max_bytes - 64) * 2 / (2 * bits + 1));
  cap_units &= ~(int64_t)(4 * kMaxRanks - 1);  // whole 32-value blocks, and slices that are whole blocks
  SGLM_CHECK_ARG(cap_units > 0, "ar_quick_all_reduce: the staging area (%ld B) is too small", (long)c->max_bytes);
  const int64_t total_units = nbytes / 16;
  for (int64_t u0 = 0; u0 < total_units; u0 += cap_units) {
    const int64_t n = (total_units - u0) < cap_units ? (total_units - u0) : cap_units;
    const uint4* ip = (const uint4*)inp + u0;
    uint4* op = (uint4*)out + u0;
    int blocks = (int)((n + kThreads - 1) / kThreads);
    blocks = blocks < 1 ? 1 : (blocks > kMaxBlocks ? kMaxBlocks : blocks);
#define QR_GO(DT)                                                                                                    \
    do {                                                                                                             \
      if (regime == 0) hipLaunchKernelGGL((all_reduce_kernel<Acc8<DT>>), dim3(blocks), dim3(kThreads), 0, s, a, ip, op, n, c->world > 1 ? 1 : 0); \
      else if (regime == 1) hipLaunchKernelGGL((quick_reduce_kernel<DT, 8>), dim3(blocks), dim3(kThreads), 0, s, a, ip, op, n);  \
      else if (regime == 2) hipLaunchKernelGGL((quick_reduce_kernel<DT, 6>), dim3(blocks), dim3(kThreads), 0, s, a, ip, op, n);  \
      else hipLaunchKernelGGL((quick_reduce_kernel<DT, 4>), dim3(blocks), dim3(kThreads), 0, s, a, ip, op, n);                   \
    } while (0)
    if (dtype == 0) QR_GO(SGL_MI355_BF16); else QR_GO(SGL_MI355_FP16);
#undef QR_GO
    int rc = check_hip(hipGetLastError(), "quick_reduce_kernel launch");
    if (rc) return rc;
  }
  return 0;
}

static int ar_fused_add_rmsnorm_impl(void* comm, const void* inp, const PartialSrc& ps, void* residual, const void* weight,
                                     void* out, void* out_q, float* out_s, int64_t num_tokens, int64_t hidden, float eps,
                                     int dtype, void* stream) {
  SGLM_CHECK_ARG(comm, "ar_fused_add_rmsnorm: null communicator");
  ArComm* c = (ArComm*)comm;
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "ar_fused_add_rmsnorm: dtype must be bfloat16 or float16");
  SGLM_CHECK_ARG(num_tokens >= 0 && hidden > 0 && hidden % (8 * c->world) == 0 && hidden <= 16384,
                 "ar_fused_add_rmsnorm: hidden (%ld) must be a multiple of 8 x world (%d) and at most 16384", (long)hidden,
                 c->world);
  const int64_t nbytes = num_tokens * hidden * 2;
  SGLM_CHECK_ARG((size_t)nbytes <= c->max_bytes, "ar_fused_add_rmsnorm: %ld B exceeds the registered capacity %ld B", (long)nbytes,
                 (long)c->max_bytes);
  if (num_tokens == 0) return 0;
  // row b <-> block b only holds for ONE row length: a byte of the staging area must always belong to the same block
  // index (see the file header); a communicator is bound to the first H it is used with
  if (c->norm_hidden == 0) c->norm_hidden = hidden;
  if (c->norm_hidden != hidden) {
    set_error("ar_fused_add_rmsnorm: this communicator's fused-norm staging area is bound to H=%ld (got %ld): use the plain "
              "all-reduce followed by fused_add_rmsnorm for other row lengths", (long)c->norm_hidden, (long)hidden);
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  SGLM_CHECK_ARG((inp || ps.partials) && residual && weight && (out || out_q), "ar_fused_add_rmsnorm: null tensor pointer");
  SGLM_CHECK_ARG(!out_q || out_s, "ar_fused_add_rmsnorm: out_q needs out_s");
  for (int r = 0; r < c->world; ++r) SGLM_CHECK_ARG(c->peer[r] != nullptr, "ar_fused_add_rmsnorm: peer %d not opened", r);
  ArArgs a{};
  for (int r = 0; r < c->world; ++r) a.peer[r] = c->peer[r];
  a.family = kFamNorm;
  a.data_off = c->data_off + (size_t)kFamNorm * 2 * c->half_bytes; a.half_bytes = c->half_bytes; a.rank = c->rank; a.world = c->world;
  a.status = reinterpret_cast<uint32_t*>(c->base + kStatusOff); a.status_host = c->status_dev; a.spin_limit = g_spin_limit; a.test_delay = g_test_delay;
  const int one_shot = nbytes <= 256 * 1024 || c->world == 1;
  const int blocks = (int)(num_tokens < kMaxBlocks ? num_tokens : kMaxBlocks);
  hipStream_t s = as_stream(stream);
  // threads / vectors per thread exactly as launch_rmsnorm (elementwise.hip): few long rows take 512 / 1024 threads
  const int nv = (int)(hidden >> 3);
  const bool wide = num_tokens <= 2048 && nv >= 512 && nv <= 2048;
  const int vpt256 = (nv + 255) / 256;
#define ARN(D, V, NT_)                                                                                                    \
  do {                                                                                                                    \
    if (ps.partials != nullptr)                                                                                           \
      hipLaunchKernelGGL((ar_add_rmsnorm_kernel<D, V, NT_, true>), dim3(blocks), dim3(NT_), 0, s, a, (const Half16<D>::T*)inp, \
                         (Half16<D>::T*)residual, (const Half16<D>::T*)weight, (Half16<D>::T*)out, (uint8_t*)out_q, out_s, \
                         (int)num_tokens, (int)hidden, eps, one_shot, ps);                                               \
    else                                                                                                                  \
      hipLaunchKernelGGL((ar_add_rmsnorm_kernel<D, V, NT_, false>), dim3(blocks), dim3(NT_), 0, s, a, (const Half16<D>::T*)inp, \
                         (Half16<D>::T*)residual, (const Half16<D>::T*)weight, (Half16<D>::T*)out, (uint8_t*)out_q, out_s, \
                         (int)num_tokens, (int)hidden, eps, one_shot, ps);                                               \
  } while (0)
#define ARN_D(D)                                   \
  do {                                             \
    if (wide) {                                    \
      if (nv <= 512) ARN(D, 1, 512);               \
      else if (nv <= 1024) ARN(D, 1, 1024);        \
      else ARN(D, 2, 1024);                        \
    } else if (vpt256 <= 1) ARN(D, 1, 256);        \
    else if (vpt256 <= 2) ARN(D, 2, 256);          \
    else if (vpt256 <= 4) ARN(D, 4, 256);          \
    else ARN(D, 8, 256);                           \
  } while (0)
  if (dtype == SGL_MI355_BF16) ARN_D(SGL_MI355_BF16);
  else ARN_D(SGL_MI355_FP16);
#undef ARN_D
#undef ARN
  return check_hip(hipGetLastError(), "ar_add_rmsnorm_kernel launch");
}

extern "C" int sgl_mi355_ar_fused_add_rmsnorm(void* comm, const void* inp, void* residual, const void* weight, void* out,
                                              void* out_q, float* out_s, int64_t num_tokens, int64_t hidden, float eps,
                                              int dtype, void* stream) {
  SGLM_CHECK_ARG(inp != nullptr || num_tokens == 0, "ar_fused_add_rmsnorm: null input");
  return ar_fused_add_rmsnorm_impl(comm, inp, PartialSrc{}, residual, weight, out, out_q, out_s, num_tokens, hidden, eps, dtype,
                                   stream);
}

// The same with this rank's addend still a split-K GEMM (sgl_mi355_fp8_scaled_mm_partials of the row-parallel layer): the
// GEMM epilogue -- slices summed in order, x w_scale, x x_scale, + bias (rank 0 only, as RowParallelLinear), one rounding to
// the 16-bit dtype -- runs while the row is staged into the IPC buffer.  Bit-identical to sgl_mi355_fp8_scaled_mm_finalize
// followed by sgl_mi355_ar_fused_add_rmsnorm; one launch less per row-parallel layer.
extern "C" int sgl_mi355_ar_fused_add_rmsnorm_partials(void* comm, const float* partials, int64_t num_slices,
                                                       const float* scales_a, const float* scales_b, const void* bias,
                                                       void* residual, const void* weight, void* out, void* out_q,
                                                       float* out_s, int64_t num_tokens, int64_t hidden, float eps, int dtype,
                                                       void* stream) {
  SGLM_CHECK_ARG(partials && scales_a && scales_b && num_slices >= 1, "ar_fused_add_rmsnorm_partials: null partials / scales");
  SGLM_CHECK_ARG(reinterpret_cast<uintptr_t>(partials) % 16 == 0 && reinterpret_cast<uintptr_t>(scales_b) % 16 == 0,
                 "ar_fused_add_rmsnorm_partials: partials and scales_b must be 16-byte aligned");
  const PartialSrc ps{partials, (int)num_slices, num_tokens * hidden, scales_a, scales_b, bias, (int)hidden};
  return ar_fused_add_rmsnorm_impl(comm, nullptr, ps, residual, weight, out, out_q, out_s, num_tokens, hidden, eps, dtype, stream);
}

extern "C" int sgl_mi355_ar_timed_out(void* comm, int* flag_out) {
  SGLM_CHECK_ARG(comm && flag_out, "ar_timed_out: null argument");
  ArComm* c = (ArComm*)comm;
  *flag_out = (int)*(volatile uint32_t*)c->status_host;  // pinned host memory: no device synchronisation
  return 0;
}

extern "C" int sgl_mi355_ar_destroy(void* comm) {
  if (!comm) return 0;
  ArComm* c = (ArComm*)comm;
  for (int r = 0; r < c->world; ++r)
    if (c->opened[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
  (void)hipFree(c->base);
  if (c->status_host) (void)hipHostFree(c->status_host);
  delete c;
  return 0;
}
