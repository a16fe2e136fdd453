// Paged (token-level) decode attention for MI355X / gfx950.
//
// Replaces (see include/sgl_mi355.h for the full citations):
//   * decode_attention_cpu            sgl-kernel/csrc/cpu/decode.cpp:1375-1575
//   * decode_attention_fwd (Triton)   python/sglang/srt/layers/attention/triton_ops/decode_attention.py:677-728
//
// Design (MI355X-first, not a translation of either reference):
//   * One workgroup = 4 waves per (request, kv-head [, head-block], kv-split).  The GQA group
//     (<= 16 q heads) is the N dim of v_mfma_f32_16x16x32, so K/V are read once per group.
//   * Every wave is an independent pipeline: it owns a private LDS ring (2 stages x
//     {K tile, V tile} of 32 tokens) filled by LDS-DMA (global_load_lds_dwordx4, one
//     256-B K/V row = 16 lanes x 16 B, so each wave instruction moves four whole rows
//     of the token-level page table -- fully coalesced), waits with counted s_waitcnt
//     vmcnt(N) and never meets a workgroup barrier inside the loop.  ~24 KB per wave /
//     96 KB per CU stay in flight, which is what hides HBM latency.
//   * The page-table slice of the split is staged once into LDS (int32), so the loop
//     has no ordinary global loads (hipcc would drain the DMA queue with vmcnt(0) at
//     each of them).
//   * QK^T: A = K rows (tokens) read by ds_read_b128 from an XOR-swizzled image
//     (conflict-free), B = Q^T kept in registers.  S^T lands with the head on the lane
//     (col = lane&15), so the softmax max needs two wave shuffles and everything else
//     is lane-local.
//   * PV: O^T = V^T P^T.  V^T fragments come from ds_read_b64_tr_b16 (hardware
//     transpose) on the same swizzled image; P^T is the S^T accumulator packed to
//     16-bit (k-slot order permuted identically on both operands).  The head is again
//     on the lane, so the online-softmax rescale is lane-local.
//   * fp32 accumulate; p rounded to the KV dtype before PV (as decode_attention.py:373).
//   * 4 waves are merged through LDS; with one split the result is written directly,
//     otherwise fp32 partials + LSE go to the caller's scratch and a small merge
//     kernel finishes (same math as _fwd_kernel_stage2 / decode_accumulate_kv_splits).
//   * Head sizes other than 64/128 (or Dv != D) take a generic wave-per-head kernel.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "partials.h"

namespace sglm {
namespace {

constexpr int kTile = 32;        // tokens per pipeline stage
constexpr int kStages = 2;
constexpr int kMaxIdx = 2048;    // page-table entries staged in LDS per pass (x kWaves/2)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
#ifndef SGLM_DEC_TIMING
#define SGLM_DEC_TIMING 0
#endif

// K/V tile pieces go in with the NON-TEMPORAL cache policy: every K/V row is read once per step by one CU, and without
// the hint the stream (537 MB per launch at bs=64, ctx 2048) churns L2 and the Infinity Cache.  Same-box A/B
// (tools/ab_variants.py, bs=64 32/8/128, random page table, S = 512 / 2048 / 4096): 30.2 / 98.5 / 185 us default policy
// vs 28.3 / 88.4 / 170 us nt -- 5.47 -> 6.09 TB/s at S = 2048.  -DSGLM_KV_DMA_NT=0 builds the default-policy variant.
#ifndef SGLM_KV_DMA_NT
#define SGLM_KV_DMA_NT 1
#endif
__device__ __forceinline__ void kv_dma16(const void* gsrc, uint32_t lds_addr) {
#if SGLM_KV_DMA_NT
  lds_dma16_nt(gsrc, lds_addr);
#else
  lds_dma16(gsrc, lds_addr);
#endif
}

struct DecodeArgs {
  const void* q;
  int64_t q_sb, q_sh;
  const void* k;
  int64_t k_sn, k_sh;
  const void* v;
  int64_t v_sn, v_sh;
  void* out;
  int64_t o_sb, o_sh;
  float* mid_o;  // fp32 partials [b][h][split][dv]
  int64_t mo_sb, mo_sh, mo_ss;
  float* mid_lse;  // natural-log LSE per (b,h,split)
  int64_t ml_sb, ml_sh, ml_ss;
  const void* indices;            // page table (int32 or int64)
  const int32_t* kv_indptr;       // mode 0: slice = indices[kv_indptr[b] : kv_indptr[b+1]]
  const int64_t* req_pool_indices;  // mode 1: slice = indices[req_pool_indices[b]*r2t_stride : +seq_lens[b]]
  const int64_t* seq_lens;
  int64_t r2t_stride;
  const int32_t* num_kv_splits;  // optional per-request split count
  int num_splits;                // grid extent along splits
  int split_align;               // split length is rounded up to this many tokens
  int num_heads, num_kv_heads, group;
  float sm_scale, logit_cap;
  int mode;
  // non-null (pairs-of-items kernel only): row_absmax[b] = max |output| over all heads of request b, as the 16-bit values
  // that are stored (atomic max; the caller zeroes it) -- the per-token scale of the w8a8 o_proj input without a pass
  // over the output (sgl_mi355_decode_attention_absmax)
  float* row_absmax;
  int kv8;  // 1 / 2: the pool is e4m3fn / e5m2 bytes (strides in elements = bytes); K is upcast, P is rounded to that format before PV
  // non-null (split kernel with fp32 partials only): merge_counters[b] counts the workgroups of request b that have
  // published their partial; the one that arrives last merges the request's kv-splits (and, if mq_out_q is given,
  // quantises the row per token) in the same launch, then zeroes the counter again.  Caller: zero once; the buffer
  // belongs to ONE stream and ONE geometry at a time (a launch that found a counter non-zero -- another stream's launch in
  // flight, or an earlier launch with another target -- would never complete the count and leave `out` unwritten).
  int32_t* merge_counters;
  uint8_t* mq_out_q;  // optional e4m3 [B][Hq * Dv] ...
  float* mq_out_s;    // ... with its scale [B]
  int pair_deal;      // pairs-of-items kernel: 1 = the second item continues the first one's deal of tiles (SGL_MI355_DECODE_PAIR_DEAL=0: both from wave 0, A/B aid)
#if SGLM_DEC_TIMING
  unsigned long long* tstamp;  // timing build: [kDecTimingWgs][kDecTimingStamps] s_memtime stamps of wave 0 (split kernel)
#endif
};

// Timing build (-DSGLM_DEC_TIMING=1, variant library only): wave 0 of the first kDecTimingWgs workgroups of the split kernel
// stamps s_memtime at its phase boundaries; tools/exp/decode_phase_times.py reads them through sgl_mi355_decode_timing_dump.
#if SGLM_DEC_TIMING
constexpr int kDecTimingWgs = 1024, kDecTimingStamps = 16;
#define DEC_STAMP(slot)                                                                          \
  do {                                                                                           \
    if (a.tstamp != nullptr && threadIdx.x < 64 && blockIdx.x < kDecTimingWgs) {                 \
      const unsigned long long tnow = __builtin_readcyclecounter();                              \
      if (threadIdx.x == 0) a.tstamp[blockIdx.x * kDecTimingStamps + (slot)] = tnow;             \
    }                                                                                            \
  } while (0)
#else
#define DEC_STAMP(slot) do {} while (0)
#endif

// The qkv GEMM of the SAME decode step, still split-K partial sums (sgl_mi355_decode_attention_qkv_partials): the pair
// kernel finishes it in its prologue -- epilogue, RoPE on q and k, k/v rows into the pool at loc[b] -- instead of a
// separate RoPE/KV-write launch in front of the attention.
struct FusedQkv {
  PartialSrc ps;             // [slices][B][(Hq + 2 Hk) D]
  const int64_t* positions;  // [B]
  const void* loc;           // [B] pool rows of the new tokens (int32 / int64)
  int loc_is64;
  const float* cos_sin;      // [max_pos][D] = cos | sin (rot_dim == D, neox pairs (d, d + D/2))
  // FUSED == 2 (round 5, sgl_mi355_decode_attention_newkv): the new token's K / V rows are finished tensors (RoPE already
  // applied by the caller's operator): [B, Hk, D] with these element strides.  Only `loc` above is used besides them.
  const void* k_new;
  const void* v_new;
  int64_t kn_sb, kn_sh, vn_sb, vn_sh;
};

__device__ __forceinline__ void split_range(const DecodeArgs& a, int b, int split, int64_t& base, int& s0, int& s1) {
  int len;
  if (a.mode == 0) {
    base = a.kv_indptr[b];
    len = a.kv_indptr[b + 1] - (int)base;
  } else {
    base = a.req_pool_indices[b] * a.r2t_stride;
    len = (int)a.seq_lens[b];
  }
  int splits = a.num_kv_splits ? a.num_kv_splits[b] : a.num_splits;
  splits = splits < 1 ? 1 : splits;
  int per = ceil_div(ceil_div(len, splits), a.split_align) * a.split_align;
  s0 = per * split;
  s1 = s0 + per < len ? s0 + per : len;
  if (split >= splits) s1 = s0;  // no work
}

// Per-token FP8 quant of one 16-bit row staged in LDS (`row`, R elements, R % 8 == 0; `amax` = this thread's max |x| over
// the elements it staged): scale = absmax / 448, q = clamp(x * (1 / scale)) -- sgl_per_token_quant_fp8's arithmetic
// (per_token_quant_fp8.cu:15-87) on that row.  All NT threads of the workgroup call it; `red`: NT / 64 floats of LDS.
template <int DTYPE>
__device__ __forceinline__ void quant_staged_row(const typename Half16<DTYPE>::T* row, int R, float amax,
                                                 uint8_t* __restrict__ out_q, float* __restrict__ out_s, int b, int tid,
                                                 int NT, float* red) {
  using H = Half16<DTYPE>;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = red[0];
  for (int i = 1; i < NT / 64; ++i) amax = fmaxf(amax, red[i]);
  const float scale = amax / 448.0f;
  if (tid == 0) out_s[b] = scale;
  const float sinv = scale == 0.f ? 0.f : 1.0f / scale;
  for (int v = tid; v < (R >> 3); v += NT) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(H::to_f32(row[8 * v + j]) * sinv, -448.0f), 448.0f);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    reinterpret_cast<uint2*>(out_q + (int64_t)b * R)[v] = uint2{(unsigned)lo, (unsigned)hi};
  }
}

template <int DTYPE, bool COHERENT>
__device__ void merge_quant_row(const DecodeArgs& a, int Dv, uint8_t* __restrict__ out_q, float* __restrict__ out_s, int b,
                                int tid, int NT, char* smem, float* red);

// Split kernels, fused merge (DecodeArgs::merge_counters): called by every thread of a workgroup after its partial (or
// its "empty split" marker) is stored.  The other workgroups of the request may sit on another XCD (own L2), and a
// device-scope fence (`__threadfence()` = L2 write-back + invalidate per workgroup) costs more than the launch it saves
// (measured: 61.6 us against 24.9 for the two launches at 64 x 1 kv head x 4 splits).  So the partials themselves are
// written and read as agent-scope relaxed atomics (`sc1`: written through to / read from the coherence point, no cache
// maintenance): stores -> s_waitcnt vmcnt(0) (they are acknowledged) -> barrier -> one lane counts the workgroup in; the
// workgroup that completes the request's count reads all partials with coherent loads and merges.
// `tail`: 64 bytes at the end of the workgroup's dynamic LDS (flag + reduction scratch; no static LDS: the largest variants
// of the kernel already ask for all 160 KB).
__device__ __forceinline__ void store_coherent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_coherent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void store_coherent16(T* p, T v) {  // a 16-bit value, written through like store_coherent
  static_assert(sizeof(T) == 2, "16-bit storage types");
  __hip_atomic_store(reinterpret_cast<unsigned short*>(p), __builtin_bit_cast(unsigned short, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
template <int DTYPE>
__device__ __forceinline__ void arrive_and_merge(const DecodeArgs& a, int Dv, int b, int tid, int NT, char* smem, char* tail) {
  int* s_last = reinterpret_cast<int*>(tail);
  float* s_red = reinterpret_cast<float*>(tail + 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's partial stores are acknowledged
  __syncthreads();                                  // ... and so are everyone's; every wave is done with the LDS
  DEC_STAMP(6);
  if (tid == 0) {
    const int nhb = (a.group + 15) >> 4;
    const int target = a.num_kv_heads * nhb * a.num_splits;
    const int old = __hip_atomic_fetch_add(a.merge_counters + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = old == target - 1;
    // the next launch starts from zero again (nobody else touches the counter before then)
    if (old == target - 1) __hip_atomic_store(a.merge_counters + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  DEC_STAMP(7);
  if (*s_last) {
    merge_quant_row<DTYPE, true>(a, Dv, a.mq_out_q, a.mq_out_s, b, tid, NT, smem, s_red);
#if SGLM_DEC_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DEC_STAMP(8);
#endif
  }
}

// 16-B chunk swizzle of a [token][D] 16-bit tile (see DESIGN.md "LDS image"):
// XOR the 32-B unit index with a per-row value so that both the ds_read_b128 row reads of the
// QK^T operand and the ds_read_b64_tr_b16 transposed reads of the PV operand are conflict-free.
template <int D>
__device__ __forceinline__ int swz_chunk(int c, int row) {
  const int f = (D == 128) ? (row & 7) : ((row >> 1) & 3);
  return (((c >> 1) ^ f) << 1) | (c & 1);
}

// KV8 byte tiles: XOR applied to the 16-B chunk index of byte row `row` (D bytes per row).  With it the ds_read_b64
// K-fragment reads (16 rows x 16 B per half-wave) and the ds_read_b64_tr_b8 V reads (16 rows x 16 B per half-wave)
// each touch all 64 banks once.
template <int D>
__device__ __forceinline__ int swz8(int row) {
  return (D == 128) ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

// KV8 = 1: the pool holds e4m3 bytes.  Tiles are DMA'd as bytes (four stages per wave); K fragments are upcast in
// registers on their way into the 16-bit QK^T MFMA, P is packed to e4m3 and P.V runs on the FP8 MFMA with V^T taken by
// ds_read_b64_tr_b8 (profiles/r01_tr_b8_probe.txt) -- no 16-bit copy of the tile, no LDS writes.  HBM traffic halves.
// E5: the pool bytes are e5m2 instead of e4m3fn (only the conversions and the byte MFMA differ)
template <int DTYPE, int D, typename IdxT, bool DIRECT_OUT, int kWaves, int KV8 = 0, bool E5 = false>
__global__ __launch_bounds__(kWaves * 64) void decode_mfma_kernel(DecodeArgs a) {
  using H = Half16<DTYPE>;
  using T = typename H::T;
  using x8 = typename H::x8;
  using x4 = typename H::x4;
  constexpr int ROWB = D * 2;               // bytes per K/V row
  constexpr int CH = ROWB / 16;             // 16-B chunks per row
  constexpr int ROWS_PER_DMA = 1024 / ROWB; // rows moved by one wave-instruction
  constexpr int NI = kTile / ROWS_PER_DMA;  // DMA instructions per K (or V) tile
  constexpr int TILE_BYTES = kTile * ROWB;
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  // KV8 == 1: byte tiles only, so the 32 KB ring of a wave holds four stages (one workgroup per CU, deepest pipeline);
  // KV8 == 2: two stages and a 16 KB page-table window, 80 KB in all, so that two workgroups share a CU and one's
  //           prologue / merge overlaps the other's streaming (what a grid of more than one workgroup per CU wants).
  constexpr int kStages8 = (KV8 == 2) ? 2 : 4;
  constexpr int WAVE_BYTES = KV8 ? kStages8 * 2 * kTile * D : kStages * STAGE_BYTES;
  constexpr int KS = D / 32;                // MFMA k-steps for QK^T
  constexpr int NDV = D / 16;               // 16-wide output column blocks
  // KV8: byte rows of D bytes; 1024 / D rows per DMA instruction; staging tile = kTile * D bytes
  constexpr int KVB = KV8 ? 1 : 2;          // bytes per pool element
  constexpr int CH8 = D / 16;               // 16-B chunks per byte row
  constexpr int ROWS8 = 1024 / D;
  constexpr int NI8 = kTile / ROWS8;
  constexpr int TILE8 = kTile * D;
  constexpr int NIQ = KV8 ? NI8 : NI;       // DMA instructions per K (or V) tile in the vmcnt queue

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kIdxCap = (KV8 == 2) ? 2 * kMaxIdx : (kWaves >= 4) ? 4 * kMaxIdx : kMaxIdx;
  int32_t* idx_lds = reinterpret_cast<int32_t*>(smem + kWaves * WAVE_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = lane & 15;  // head within block (S^T / O^T column)
  const int g = lane >> 4;   // lane group

  // blockIdx.x = ((b * (Hkv*nhb) + hblk) * num_splits + split)
  const int nhb = (a.group + 15) >> 4;
  int bid = blockIdx.x;
  const int split = bid % a.num_splits;
  bid /= a.num_splits;
  const int hblk = bid % (a.num_kv_heads * nhb);
  const int b = bid / (a.num_kv_heads * nhb);
  const int kvh = hblk / nhb;
  const int hb = hblk - kvh * nhb;
  const int h0 = kvh * a.group + hb * 16;
  const int nh = (a.group - hb * 16) < 16 ? (a.group - hb * 16) : 16;

  DEC_STAMP(0);
  int64_t base;
  int s0, s1;
  split_range(a, b, split, base, s0, s1);
#if SGLM_DEC_TIMING
  asm volatile("" ::"s"(__builtin_amdgcn_readfirstlane(s1)));
  DEC_STAMP(1);
#endif

  if (s0 >= s1) {
    // Empty split: tell the merge pass to ignore it.  (With DIRECT_OUT an empty range
    // means an empty sequence: the output row is zero.)
    if (tid < nh) {
      if (DIRECT_OUT) {
        T* o = reinterpret_cast<T*>(a.out) + (int64_t)b * a.o_sb + (int64_t)(h0 + tid) * a.o_sh;
        for (int d = 0; d < D; ++d) o[d] = H::from_f32(0.f);
      } else {
        float* lp = a.mid_lse + (int64_t)b * a.ml_sb + (int64_t)(h0 + tid) * a.ml_sh + (int64_t)split * a.ml_ss;
        if (a.merge_counters != nullptr) store_coherent(lp, -INFINITY);
        else *lp = -INFINITY;
      }
    }
    if constexpr (!DIRECT_OUT)
      if (a.merge_counters != nullptr)
        arrive_and_merge<DTYPE>(a, D, b, tid, kWaves * 64, smem, smem + kWaves * WAVE_BYTES + kIdxCap * 4 - 64);
    return;
  }

  // ---- Q^T fragments (B operand): lane (hl, g) holds Q[h0+hl][32*ks + 8*g .. +8]
  x8 qf[KS];
  {
    const T* qp = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.q_sb + (int64_t)(h0 + hl) * a.q_sh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (hl < nh) {
        qf[ks] = *reinterpret_cast<const x8*>(qp + 32 * ks + 8 * g);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (T)0.f;
      }
    }
    // Consume the fragments here so that hipcc's wait for these (ordinary) loads lands before
    // the loop; left to itself it puts `s_waitcnt vmcnt(0)` at their first use INSIDE the
    // loop, which would drain the DMA queue every iteration.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
  }

  const char* kbase = reinterpret_cast<const char*>(a.k) + (int64_t)kvh * a.k_sh * KVB;
  const char* vbase = reinterpret_cast<const char*>(a.v) + (int64_t)kvh * a.v_sh * KVB;
  const int64_t k_row_bytes = a.k_sn * KVB;
  const int64_t v_row_bytes = a.v_sn * KVB;
  char* wave_lds = smem + wave * WAVE_BYTES;

  // DMA source mapping of this lane: row within a piece and the (unswizzled) chunk it fetches
  const int dma_row = lane / CH;
  const int dma_pos = lane % CH;

  // online-softmax state (log2 domain), one head per lane column
  float m_run = -INFINITY;
  float l_run = 0.f;  // partial over this lane's token slots
  f32x4 o_acc[NDV];
#pragma unroll
  for (int i = 0; i < NDV; ++i) o_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float scale_log2 = a.sm_scale * kLog2e;
  const bool has_cap = a.logit_cap > 0.f;

  for (int p0 = s0; p0 < s1; p0 += kIdxCap) {
    const int n_pass = (s1 - p0) < kIdxCap ? (s1 - p0) : kIdxCap;
    __syncthreads();  // previous pass finished with idx_lds
    {
      const IdxT* src = reinterpret_cast<const IdxT*>(a.indices) + base + p0;
      for (int i = tid; i < n_pass; i += kWaves * 64) idx_lds[i] = (int32_t)src[i];
    }
    __syncthreads();
    if (p0 == s0) DEC_STAMP(2);

    const int ntiles = ceil_div(n_pass, kTile);
    const int nt = (ntiles - wave + kWaves - 1) / kWaves;  // tiles wave, wave+4, ... of this pass

    auto issue = [&](int jt, int stage, bool is_v) {
      const int tok0 = (wave + kWaves * jt) * kTile;
      const char* gb = is_v ? vbase : kbase;
      const int64_t rb = is_v ? v_row_bytes : k_row_bytes;
      if constexpr (KV8) {
        // byte rows of D bytes; lane i of an instruction fills chunk position i % CH8 of LDS row i / CH8 with source
        // chunk (i % CH8) ^ swz8(row).  K rows are the tile's tokens in order; V rows are the PV k-slots:
        // row 8g + j holds token 4g + j (j < 4) or 16 + 4g + (j - 4), so that ds_read_b64_tr_b8 hands lane group g
        // the 8 tokens its P fragment is ordered by.
        const uint32_t dst8 = __builtin_amdgcn_readfirstlane(
            lds_addr_of(wave_lds + stage * 2 * TILE8 + (is_v ? TILE8 : 0)));
        int32_t tok8[NI8];
        int src_off[NI8];
#pragma unroll
        for (int i = 0; i < NI8; ++i) {
          const int row = i * ROWS8 + lane / CH8;
          const int slot = is_v ? ((row & 4) ? 16 + 4 * (row >> 3) + (row & 3) : 4 * (row >> 3) + (row & 3)) : row;
          int tp = tok0 + slot;
          tp = tp < n_pass ? tp : n_pass - 1;
          tok8[i] = idx_lds[tp];
          src_off[i] = ((lane % CH8) ^ swz8<D>(row)) * 16;
        }
#pragma unroll
        for (int i = 0; i < NI8; ++i) kv_dma16(gb + (int64_t)tok8[i] * rb + src_off[i], dst8 + i * 1024);
        return;
      }
      const uint32_t dst = __builtin_amdgcn_readfirstlane(
          lds_addr_of(wave_lds + stage * STAGE_BYTES + (is_v ? TILE_BYTES : 0)));
      int32_t tok[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int tp = tok0 + i * ROWS_PER_DMA + dma_row;
        tp = tp < n_pass ? tp : n_pass - 1;  // tail rows re-read a valid token; masked below
        tok[i] = idx_lds[tp];
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = swz_chunk<D>(dma_pos, i * ROWS_PER_DMA + dma_row);
        kv_dma16(gb + (int64_t)tok[i] * rb + c * 16, dst + i * 1024);
      }
    };

    if constexpr (KV8) {
      // ---- e4m3 pool: K fragments are converted in registers straight from the byte tile (K is upcast,
      // decode_attention.py:336), P is packed to e4m3 (:373) and P.V runs on the FP8 MFMA with V^T read by
      // ds_read_b64_tr_b8: no 16-bit copy of the tile, four byte stages in flight per wave.
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      // `units` DMA groups of NI8 instructions may stay in flight (K or V tile = one unit)
      auto wait_units = [&](int units) __attribute__((always_inline)) {
        switch (units) {
          case 0: wait_vmcnt<0>(); break;
          case 1: wait_vmcnt<NI8>(); break;
          case 2: wait_vmcnt<2 * NI8>(); break;
          case 3: wait_vmcnt<3 * NI8>(); break;
          case 4: wait_vmcnt<4 * NI8>(); break;
          case 5: wait_vmcnt<5 * NI8>(); break;
          case 6: wait_vmcnt<6 * NI8>(); break;
          default: wait_vmcnt<7 * NI8>(); break;
        }
      };
#pragma unroll
      for (int pj = 0; pj < kStages8; ++pj)
        if (pj < nt) {
          issue(pj, pj, false);
          issue(pj, pj, true);
        }
      for (int jt = 0; jt < nt; ++jt) {
        const int st = jt & (kStages8 - 1);
        const char* kst = wave_lds + st * 2 * TILE8;
        const char* vst = kst + TILE8;
        const int tok0 = (wave + kWaves * jt) * kTile;
        // in the queue behind K(jt): V(jt), K,V of the next `after` tiles
        const int after = (nt - 1 - jt) < (kStages8 - 1) ? (nt - 1 - jt) : (kStages8 - 1);
        const bool refill = jt + kStages8 < nt;
        wait_units(1 + 2 * after);

        // ---- S^T = K Q^T: lane (hl, g) reads bytes 32ks + 8g .. +8 of row 16th + hl
        f32x4 s_acc[2];
#pragma unroll
        for (int th = 0; th < 2; ++th) {
          s_acc[th] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int row = 16 * th + hl;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int c = (2 * ks + (g >> 1)) ^ swz8<D>(row);
            const uint2 raw = *reinterpret_cast<const uint2*>(kst + row * D + c * 16 + 8 * (g & 1));
            x8 kf;
            const f32x2_t a0 = cvt_pk_f32_kv<E5, false>((int)raw.x), a1 = cvt_pk_f32_kv<E5, true>((int)raw.x);
            const f32x2_t b0 = cvt_pk_f32_kv<E5, false>((int)raw.y), b1 = cvt_pk_f32_kv<E5, true>((int)raw.y);
            kf[0] = H::from_f32(a0[0]); kf[1] = H::from_f32(a0[1]); kf[2] = H::from_f32(a1[0]); kf[3] = H::from_f32(a1[1]);
            kf[4] = H::from_f32(b0[0]); kf[5] = H::from_f32(b0[1]); kf[6] = H::from_f32(b1[0]); kf[7] = H::from_f32(b1[1]);
            s_acc[th] = H::mfma16(kf, qf[ks], s_acc[th]);
          }
        }
        wait_lgkmcnt0();  // the K bytes are in registers: refill the slot
        if (refill) issue(jt + kStages8, st, false);

        // ---- online softmax (log2 domain); token of (th, r) = tok0 + 16*th + 4*g + r
        float sv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sv[i] = s_acc[i >> 2][i & 3] * (has_cap ? a.sm_scale : scale_log2);
        if (has_cap) {
#pragma unroll
          for (int i = 0; i < 8; ++i) sv[i] = a.logit_cap * tanhf(sv[i] / a.logit_cap) * kLog2e;
        }
        float m_tile = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool valid = (tok0 + 16 * (i >> 2) + 4 * g + (i & 3)) < n_pass;
          sv[i] = valid ? sv[i] : -INFINITY;
          m_tile = fmaxf(m_tile, sv[i]);
        }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 16));
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);  // finite: every tile holds >= 1 valid token
        const float alpha = exp2f(m_run - m_new);
        float psum = 0.f;
        float pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          pv[i] = exp2f(sv[i] - m_new);
          psum += pv[i];  // the row sum keeps the unrounded p (decode_attention.py:375)
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // P^T as e4m3 bytes, k-slot order j = 0..7 (decode_attention.py:373: p.to(v.dtype))
        int p_lo = cvt_pk_kv_f32<E5, false>(pv[0], pv[1], 0);
        p_lo = cvt_pk_kv_f32<E5, true>(pv[2], pv[3], p_lo);
        int p_hi = cvt_pk_kv_f32<E5, false>(pv[4], pv[5], 0);
        p_hi = cvt_pk_kv_f32<E5, true>(pv[6], pv[7], p_hi);
        const long pf8 = (long)(((unsigned long)(unsigned)p_hi << 32) | (unsigned long)(unsigned)p_lo);
        if (__ballot(alpha != 1.f) != 0) {  // the running max of some head moved: rescale (exact no-op otherwise)
#pragma unroll
          for (int i = 0; i < NDV; ++i) o_acc[i] *= alpha;
        }

        // ---- wait for V(jt): behind it the next `after` tiles [+ K(jt + stages)]
        wait_units(2 * after + (refill ? 1 : 0));
        // ---- O^T += V^T P^T on the FP8 MFMA; lane 2q+p of group g points at row 8g+q, bytes 8p.. of the block
        {
          const int vrow = 8 * g + ((lane & 15) >> 1);
          const char* vrp = vst + vrow * D + 8 * (lane & 1);
          const int vsw = swz8<D>(vrow);
#pragma unroll
          for (int dvb = 0; dvb < NDV; ++dvb) {
            typedef int v2i_t __attribute__((ext_vector_type(2)));
            const v2i_t vr = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
                (__attribute__((address_space(3))) v2i_t*)(vrp + ((dvb ^ vsw) * 16)));
            const long vf8 = (long)(((unsigned long)(unsigned)vr[1] << 32) | (unsigned long)(unsigned)vr[0]);
            o_acc[dvb] = mfma_kv8<E5>(vf8, pf8, o_acc[dvb]);
          }
        }
        wait_lgkmcnt0();  // the V bytes are in registers: refill the slot
        if (refill) issue(jt + kStages8, st, true);
      }
      continue;  // next page-table pass
    }

    if (nt > 0) {
      issue(0, 0, false);
      issue(0, 0, true);
    }
    if (nt > 1) {
      issue(1, 1, false);
      issue(1, 1, true);
    }

    for (int jt = 0; jt < nt; ++jt) {
      const int st = jt & 1;
      const char* kst = wave_lds + st * STAGE_BYTES;
      const char* vst = kst + TILE_BYTES;
      const int tok0 = (wave + kWaves * jt) * kTile;
      const bool more1 = jt + 1 < nt;
      const bool more2 = jt + 2 < nt;

      // ---- wait for K(jt): younger ops allowed in flight = V(jt) [+ K,V(jt+1)]
      if (more1) wait_vmcnt<3 * NIQ>(); else wait_vmcnt<NIQ>();
      if (jt == 0 && p0 == s0) DEC_STAMP(3);

      // ---- S^T = K Q^T  (rows = tokens, cols = heads)
      f32x4 s_acc[2];
#pragma unroll
      for (int th = 0; th < 2; ++th) {
        s_acc[th] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int row = 16 * th + hl;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int c = swz_chunk<D>(4 * ks + g, row);
          const x8 kf = *reinterpret_cast<const x8*>(kst + row * ROWB + c * 16);
          s_acc[th] = H::mfma16(kf, qf[ks], s_acc[th]);
        }
      }
      wait_lgkmcnt0();  // K fragments are in registers: the K buffer may be refilled
      if (more2) issue(jt + 2, st, false);

      // ---- online softmax (log2 domain); token of (th, r) = tok0 + 16*th + 4*g + r
      float sv[8];
      float m_tile = -INFINITY;
#pragma unroll
      for (int th = 0; th < 2; ++th) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = s_acc[th][r];
          if (has_cap) {
            s = s * a.sm_scale;
            s = a.logit_cap * tanhf(s / a.logit_cap) * kLog2e;
          } else {
            s = s * scale_log2;
          }
          const bool valid = (tok0 + 16 * th + 4 * g + r) < n_pass;
          s = valid ? s : -INFINITY;
          sv[th * 4 + r] = s;
          m_tile = fmaxf(m_tile, s);
        }
      }
      m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 16));
      m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
      const float m_new = fmaxf(m_run, m_tile);  // finite: every tile holds >= 1 valid token
      const float alpha = exp2f(m_run - m_new);
      float psum = 0.f;
      x8 pf;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float p = exp2f(sv[i] - m_new);
        psum += p;  // the row sum keeps the unrounded p (decode_attention.py:375)
        pf[i] = H::from_f32(p);
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < NDV; ++i) o_acc[i] *= alpha;

      // ---- wait for V(jt): younger ops allowed = [K,V(jt+1)] [+ K(jt+2)]
      if (more2) wait_vmcnt<3 * NIQ>(); else if (more1) wait_vmcnt<2 * NIQ>(); else wait_vmcnt<0>();

      // ---- O^T += V^T P^T ; k-slot (g, j): j<4 -> token 4g+j, j>=4 -> token 16+4g+(j-4)
      {
        const int q4 = (lane >> 2) & 3;  // row within the 4-row transposed block
        const int p4 = lane & 3;         // 8-B piece within the 32-B column block
        const int row_lo = 4 * g + q4;
#pragma unroll
        for (int dvb = 0; dvb < NDV; ++dvb) {
          const int c = 2 * dvb + (p4 >> 1);
          const int off = swz_chunk<D>(c, row_lo) * 16 + 8 * (p4 & 1);  // same for row_lo + 16
          const x4 v_lo = H::ds_read_tr(vst + row_lo * ROWB + off);
          const x4 v_hi = H::ds_read_tr(vst + (row_lo + 16) * ROWB + off);
          x8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vf[j] = v_lo[j];
            vf[4 + j] = v_hi[j];
          }
          o_acc[dvb] = H::mfma16(vf, pf, o_acc[dvb]);
        }
      }
      wait_lgkmcnt0();  // V fragments are in registers: the V buffer may be refilled
      if (more2) issue(jt + 2, st, true);
    }
  }

  // ---- reduce l over the 4 lane groups, then merge the 4 waves through LDS
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
#if SGLM_DEC_TIMING
  asm volatile("" ::"s"(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, o_acc[NDV - 1][3]))));
  DEC_STAMP(4);
#endif

  __syncthreads();  // every wave is done with its ring (all DMA waited, all reads retired)
  DEC_STAMP(5);
  float* mrg_o = reinterpret_cast<float*>(smem);               // [wave][16][D]
  float* mrg_m = mrg_o + kWaves * 16 * D;                      // [wave][16]
  float* mrg_l = mrg_m + kWaves * 16;                          // [wave][16]
  if (hl < nh) {
    float* dst = mrg_o + (wave * 16 + hl) * D;
#pragma unroll
    for (int dvb = 0; dvb < NDV; ++dvb) *reinterpret_cast<f32x4*>(dst + dvb * 16 + 4 * g) = o_acc[dvb];
    if (g == 0) {
      mrg_m[wave * 16 + hl] = m_run;
      mrg_l[wave * 16 + hl] = l_run;
    }
  }
  __syncthreads();
  for (int e = tid; e < nh * D; e += kWaves * 64) {
    const int h = e / D;
    const int dv = e - h * D;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) M = fmaxf(M, mrg_m[w * 16 + h]);
    float L = 0.f, val = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const float f = exp2f(mrg_m[w * 16 + h] - M);  // a wave without tiles has m = -inf -> 0
      L += mrg_l[w * 16 + h] * f;
      val += mrg_o[(w * 16 + h) * D + dv] * f;
    }
    const float r = val / L;
    if (DIRECT_OUT) {
      reinterpret_cast<T*>(a.out)[(int64_t)b * a.o_sb + (int64_t)(h0 + h) * a.o_sh + dv] = H::from_f32(r);
    } else {
      float* op = a.mid_o + (int64_t)b * a.mo_sb + (int64_t)(h0 + h) * a.mo_sh + (int64_t)split * a.mo_ss + dv;
      float* lp = a.mid_lse + (int64_t)b * a.ml_sb + (int64_t)(h0 + h) * a.ml_sh + (int64_t)split * a.ml_ss;
      if (a.merge_counters != nullptr) {  // read by another workgroup of this launch: see arrive_and_merge
        store_coherent(op, r);
        if (dv == 0) store_coherent(lp, (M + log2f(L)) * kLn2);
      } else {
        *op = r;
        if (dv == 0) *lp = (M + log2f(L)) * kLn2;
      }
    }
  }
  if constexpr (!DIRECT_OUT)
    if (a.merge_counters != nullptr)
      arrive_and_merge<DTYPE>(a, D, b, tid, kWaves * 64, smem, smem + kWaves * WAVE_BYTES + kIdxCap * 4 - 64);
}

// --------------------------------------------------------------------------------------
// Two (request, kv-head block) items per workgroup, back to back in ONE DMA stream.
//
// With more items than CUs the kernel above runs them in rounds, and on every CU the second workgroup's prologue (page
// table + Q, then the first tiles: two dependent memory round trips, ~5 us under load) and the first one's merge cannot
// overlap anything -- the LDS rings leave room for one workgroup.  Here a workgroup owns items 2i and 2i+1: both page
// tables and both Q blocks are fetched in the one prologue, every wave walks its tiles of the first item and continues
// straight into its tiles of the second (the two-stage ring keeps prefetching across the boundary), and both merges run
// at the end.  16-bit pools, one split, 4 waves; items longer than the staged window fall back to one item at a time.
//
// QOUT: the per-token FP8 quant of the finished row (the w8a8 o_proj input) in the same launch.  A request's heads are
// spread over Hkv items in different workgroups (and XCDs), so the 16-bit outputs are stored write-through (sc1), every
// workgroup counts its items in on merge_counters[b], and the one that completes request b's count reads the row back
// with coherent loads and quantises it with sgl_per_token_quant_fp8's arithmetic (quant_staged_row): same bits as the
// quant launch it replaces, ~0.1 us of work for one workgroup in 32 instead of a 4.8 us launch behind the slowest one.
// FUSED: 0 = every token comes from the pool; 1 = q / k / v of this step out of the qkv GEMM's partial sums (opt-in build);
// 2 = the new token's K / V rows are tensors (fq.k_new / v_new): the launch writes them to the pool rows loc[b] for the steps
// to come and takes them into the softmax as one more partial state at the merge, exactly like form 1 -- the stream covers
// the len - 1 older tokens, and the KV-write launch in front of the attention disappears (the reference call order:
// attn_backend.forward(save_kv_cache=True)).  Contract of form 2: loc[b] IS the page-table entry of position seq_lens[b] - 1.
template <int DTYPE, int D, typename IdxT, bool KV8 = false, int FUSED = 0, bool QOUT = false>
__global__ __launch_bounds__(256) void decode_mfma_pair_kernel(DecodeArgs a, int num_items, FusedQkv fq) {
  using H = Half16<DTYPE>;
  using T = typename H::T;
  using x8 = typename H::x8;
  using x4 = typename H::x4;
  constexpr int kWaves = 4;
  constexpr int ROWB = D * 2;
  constexpr int CH = ROWB / 16;
  constexpr int ROWS_PER_DMA = 1024 / ROWB;
  constexpr int NI = kTile / ROWS_PER_DMA;
  constexpr int TILE_BYTES = kTile * ROWB;
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  // KV8 (e4m3 pool): byte tiles, four stages per wave, K upcast in registers, P.V on the FP8 MFMA (see decode_mfma_kernel)
  constexpr int KVB = KV8 ? 1 : 2;
  constexpr int CH8 = D / 16, ROWS8 = 1024 / D, NI8 = kTile / ROWS8, TILE8 = kTile * D, kStages8 = 4;
  constexpr int WAVE_BYTES = KV8 ? kStages8 * 2 * TILE8 : kStages * STAGE_BYTES;
  constexpr int KS = D / 32;
  constexpr int NDV = D / 16;
  constexpr int CAP = 2 * kMaxIdx;  // page-table entries staged per item (the two halves of the 32 KB window)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* idx_lds = reinterpret_cast<int32_t*>(smem + kWaves * WAVE_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = lane & 15;
  const int g = lane >> 4;
  const int nhb = (a.group + 15) >> 4;
  const int dma_row = lane / CH, dma_pos = lane % CH;
  const int64_t k_row_bytes = a.k_sn * KVB, v_row_bytes = a.v_sn * KVB;
  char* wave_lds = smem + wave * WAVE_BYTES;
  const float scale_log2 = a.sm_scale * kLog2e;
  const bool has_cap = a.logit_cap > 0.f;

  // ---- the two items
  struct Item {
    int b, h0, nh, len;
    int64_t base;
    const char *kbase, *vbase;
  } it[2];
  x8 qfs[2][KS];
  bool newtok[2];       // FUSED: the item has a new token (always, unless seq_lens[b] == 0)
  float s_new[2];       // FUSED: its score for head hl, log2 domain (what the streamed tiles call sv)
  float v_new[2];       // FUSED: its value at dv = tid % D (the merge loop's dv of this thread)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int item = 2 * (int)blockIdx.x + i;
    const bool ok = item < num_items;
    const int id = ok ? item : num_items - 1;
    const int hblk = id % (a.num_kv_heads * nhb);
    const int b = id / (a.num_kv_heads * nhb);
    const int kvh = hblk / nhb, hb = hblk - kvh * nhb;
    it[i].b = b;
    it[i].h0 = kvh * a.group + hb * 16;
    it[i].nh = (a.group - hb * 16) < 16 ? (a.group - hb * 16) : 16;
    int s0, s1;
    split_range(a, b, 0, it[i].base, s0, s1);
    // -1: no such item.  FUSED: the new token (position s1 - 1) is not streamed from the pool -- it joins at the merge
    it[i].len = ok ? (FUSED ? (s1 > 0 ? s1 - 1 : 0) : s1) : -1;
    newtok[i] = FUSED && ok && s1 > 0;
    it[i].kbase = reinterpret_cast<const char*>(a.k) + (int64_t)kvh * a.k_sh * KVB;
    it[i].vbase = reinterpret_cast<const char*>(a.v) + (int64_t)kvh * a.v_sh * KVB;
    if constexpr (FUSED != 1) {
      const T* qp = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.q_sb + (int64_t)(it[i].h0 + hl) * a.q_sh;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (hl < it[i].nh) {
          qfs[i][ks] = *reinterpret_cast<const x8*>(qp + 32 * ks + 8 * g);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) qfs[i][ks][j] = (T)0.f;
        }
      }
    }
  }
  // page tables: the first window of both items goes in with the one prologue
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = it[i].len < CAP ? it[i].len : CAP;
    const IdxT* src = reinterpret_cast<const IdxT*>(a.indices) + it[i].base;
    for (int e = tid; e < n; e += 256) idx_lds[i * CAP + e] = (int32_t)src[e];
  }
  if constexpr (FUSED == 2) {
    // The new token of each item: score of every head against its K row (fp32 dot of the 16-bit values, the products the
    // MFMA would form), its V element for this thread's output column, and both rows into the pool -- loads that depend on
    // b alone, in flight beside the page-table loads above.
    static_assert(!KV8 && 256 % D == 0, "new-token form: 16-bit pools, head size dividing 256");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      s_new[i] = 0.f;
      v_new[i] = 0.f;
      if (!newtok[i]) continue;  // (workgroup-uniform)
      const int b = it[i].b, kvh = it[i].h0 / a.group;
      const int64_t row = fq.loc_is64 ? reinterpret_cast<const int64_t*>(fq.loc)[b]
                                      : (int64_t) reinterpret_cast<const int32_t*>(fq.loc)[b];
      const T* kn = reinterpret_cast<const T*>(fq.k_new) + (int64_t)b * fq.kn_sb + (int64_t)kvh * fq.kn_sh;
      const T* vn = reinterpret_cast<const T*>(fq.v_new) + (int64_t)b * fq.vn_sb + (int64_t)kvh * fq.vn_sh;
      float dot = 0.f;
      if (hl < it[i].nh) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const x8 kv = *reinterpret_cast<const x8*>(kn + 32 * ks + 8 * g);
#pragma unroll
          for (int j = 0; j < 8; ++j) dot += H::to_f32(qfs[i][ks][j]) * H::to_f32(kv[j]);
        }
      }
      dot += __shfl_xor(dot, 16);
      dot += __shfl_xor(dot, 32);
      s_new[i] = has_cap ? a.logit_cap * tanhf(dot * a.sm_scale / a.logit_cap) * kLog2e : dot * scale_log2;
      v_new[i] = H::to_f32(vn[tid % D]);
      // the pool rows for the steps to come (several head blocks of one kv head write the same bytes: harmless)
      if (tid < D / 8) {
        T* kp = reinterpret_cast<T*>(const_cast<void*>(a.k)) + row * a.k_sn + (int64_t)kvh * a.k_sh;
        *reinterpret_cast<x8*>(kp + 8 * tid) = *reinterpret_cast<const x8*>(kn + 8 * tid);
      } else if (tid < D / 4) {
        T* vp = reinterpret_cast<T*>(const_cast<void*>(a.v)) + row * a.v_sn + (int64_t)kvh * a.v_sh;
        *reinterpret_cast<x8*>(vp + 8 * (tid - D / 8)) = *reinterpret_cast<const x8*>(vn + 8 * (tid - D / 8));
      }
    }
  }
  if constexpr (FUSED == 1) {
    // q/k/v of this step come out of the qkv GEMM's partial sums.  Tasks of 8 columns (and their RoPE partners D/2 columns
    // on) are dealt over the 256 threads: per item nh q heads x HC chunk pairs, HC k chunk pairs, 2 HC v chunks.  Rotated
    // q, k and v are staged in LDS (wave 0's ring, before any DMA lands there): [item][16 heads | k | v][D].  k and v also
    // go to the pool row loc[b] for the steps to come; THIS launch does not read them back -- the stream covers the
    // len - 1 older tokens and the new one joins at the merge as one more partial state (m = its score, l = 1, o = v).
    static_assert(!KV8 && 256 % D == 0, "fused qkv prologue: 16-bit pools, head size dividing 256");
    constexpr int HC = D / 16;  // 8-column chunks in half a head
    T* q_lds = reinterpret_cast<T*>(smem);
    int64_t pos[2], row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // loads that depend on b alone go out with the page-table loads above
      pos[i] = fq.positions[it[i].b];
      row[i] = fq.loc_is64 ? reinterpret_cast<const int64_t*>(fq.loc)[it[i].b]
                           : (int64_t) reinterpret_cast<const int32_t*>(fq.loc)[it[i].b];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!newtok[i]) continue;
      const int b = it[i].b, kvh = it[i].h0 / a.group;
      const int ntask = (it[i].nh + 3) * HC;
      const float* cs = fq.cos_sin + pos[i] * D;
      T* stage = q_lds + i * 18 * D;
      for (int task = tid; task < ntask; task += 256) {
        const int hq = task / HC, c = task - hq * HC;
        if (hq <= it[i].nh) {  // a q head (hq < nh) or the k head (hq == nh): rotate the pair of chunks (8c, 8c + D/2)
          const bool is_k = hq == it[i].nh;
          const int col = (is_k ? (a.num_heads + kvh) : (it[i].h0 + hq)) * D + 8 * c;
          const x8 x1 = gemm_row8<DTYPE>(fq.ps, b, col), x2 = gemm_row8<DTYPE>(fq.ps, b, col + D / 2);
          const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + 8 * c), c1 = *reinterpret_cast<const f32x4*>(cs + 8 * c + 4);
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs + D / 2 + 8 * c),
                      s1 = *reinterpret_cast<const f32x4*>(cs + D / 2 + 8 * c + 4);
          x8 o1, o2;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float r1, r2;
            rope_pair(H::to_f32(x1[j]), H::to_f32(x2[j]), j < 4 ? c0[j] : c1[j - 4], j < 4 ? s0[j] : s1[j - 4], r1, r2);
            o1[j] = H::from_f32(r1);
            o2[j] = H::from_f32(r2);
          }
          T* dst = stage + (is_k ? 16 : hq) * D;
          *reinterpret_cast<x8*>(dst + 8 * c) = o1;
          *reinterpret_cast<x8*>(dst + D / 2 + 8 * c) = o2;
          if (is_k) {
            T* kp = reinterpret_cast<T*>(const_cast<void*>(a.k)) + row[i] * a.k_sn + (int64_t)kvh * a.k_sh;
            *reinterpret_cast<x8*>(kp + 8 * c) = o1;
            *reinterpret_cast<x8*>(kp + D / 2 + 8 * c) = o2;
          }
        } else {  // v chunk
          const int vc = task - (it[i].nh + 1) * HC;
          const x8 x = gemm_row8<DTYPE>(fq.ps, b, (a.num_heads + a.num_kv_heads + kvh) * D + 8 * vc);
          *reinterpret_cast<x8*>(stage + 17 * D + 8 * vc) = x;
          *reinterpret_cast<x8*>(reinterpret_cast<T*>(const_cast<void*>(a.v)) + row[i] * a.v_sn + (int64_t)kvh * a.v_sh +
                                 8 * vc) = x;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const T* stage = q_lds + i * 18 * D;
      float dot = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (hl < it[i].nh && newtok[i]) {
          qfs[i][ks] = *reinterpret_cast<const x8*>(stage + hl * D + 32 * ks + 8 * g);
          const x8 kn = *reinterpret_cast<const x8*>(stage + 16 * D + 32 * ks + 8 * g);
#pragma unroll
          for (int j = 0; j < 8; ++j) dot += H::to_f32(qfs[i][ks][j]) * H::to_f32(kn[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) qfs[i][ks][j] = (T)0.f;
        }
      }
      dot += __shfl_xor(dot, 16);
      dot += __shfl_xor(dot, 32);
      s_new[i] = has_cap ? a.logit_cap * tanhf(dot * a.sm_scale / a.logit_cap) * kLog2e : dot * scale_log2;
      v_new[i] = newtok[i] ? H::to_f32(stage[17 * D + (tid % D)]) : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qfs[i][ks]));  // waits for these loads stay out of the loop
  __syncthreads();

  // online-softmax state of the item being streamed, and the finished first item
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o_acc[NDV];
#pragma unroll
  for (int i = 0; i < NDV; ++i) o_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_first = -INFINITY, l_first = 0.f;
  f32x4 o_first[NDV];
#pragma unroll
  for (int i = 0; i < NDV; ++i) o_first[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = qfs[0][ks];

  bool parked = false;  // the first item's result sits in *_first, the Q fragments are the second item's
  auto park_first = [&]() __attribute__((always_inline)) {
    m_first = m_run;
    l_first = l_run;
    m_run = -INFINITY;
    l_run = 0.f;
#pragma unroll
    for (int k = 0; k < NDV; ++k) {
      o_first[k] = o_acc[k];
      o_acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = qfs[1][ks];
    parked = true;
  };

  // One stream over up to two segments: (sel0: idx window off0, n0 tokens, nt0 tiles of this wave) then (sel1: ...).
  // At the boundary the running state moves to *_first and the Q fragments switch.
  auto stream = [&](int sel0, int off0, int n0, int sel1, int off1, int n1) __attribute__((always_inline)) {
    // tiles are dealt round-robin over the waves, and the second segment CONTINUES the first one's deal (its tile j goes to
    // wave (tiles0 + j) % kWaves): the waves' totals differ by at most one tile.  With both segments starting at wave 0, two
    // items of 65 tiles (ctx 2049..2080) gave wave 0 34 tiles and the others 32; now 33 / 33 / 32 / 32.
    const int tiles0 = n0 > 0 ? ceil_div(n0, kTile) : 0;
    const int w1 = a.pair_deal ? (wave - tiles0) & (kWaves - 1) : wave;  // this wave's place in the second segment's deal
    const int nt0 = n0 > 0 ? (tiles0 - wave + kWaves - 1) / kWaves : 0;
    const int nt1 = n1 > 0 ? (ceil_div(n1, kTile) - w1 + kWaves - 1) / kWaves : 0;
    const int nvt = nt0 + nt1;
    if constexpr (KV8) {
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      auto issue8 = [&](int vt, int stage, bool is_v) __attribute__((always_inline)) {
        const bool second = vt >= nt0;
        const int jt = second ? vt - nt0 : vt;
        const int sel = second ? sel1 : sel0;
        const int off = second ? off1 : off0;
        const int n = second ? n1 : n0;
        const int tok0 = ((second ? w1 : wave) + kWaves * jt) * kTile;
        const char* gb = sel ? (is_v ? it[1].vbase : it[1].kbase) : (is_v ? it[0].vbase : it[0].kbase);
        const int64_t rb = is_v ? v_row_bytes : k_row_bytes;
        const uint32_t dst8 = __builtin_amdgcn_readfirstlane(
            lds_addr_of(wave_lds + stage * 2 * TILE8 + (is_v ? TILE8 : 0)));
        int32_t tok8[NI8];
        int src_off[NI8];
#pragma unroll
        for (int i = 0; i < NI8; ++i) {
          const int row = i * ROWS8 + lane / CH8;
          const int slot = is_v ? ((row & 4) ? 16 + 4 * (row >> 3) + (row & 3) : 4 * (row >> 3) + (row & 3)) : row;
          int tp = tok0 + slot;
          tp = tp < n ? tp : n - 1;
          tok8[i] = idx_lds[off + tp];
          src_off[i] = ((lane % CH8) ^ swz8<D>(row)) * 16;
        }
#pragma unroll
        for (int i = 0; i < NI8; ++i) kv_dma16(gb + (int64_t)tok8[i] * rb + src_off[i], dst8 + i * 1024);
      };
      auto wait_units = [&](int units) __attribute__((always_inline)) {
        switch (units) {
          case 0: wait_vmcnt<0>(); break;
          case 1: wait_vmcnt<NI8>(); break;
          case 2: wait_vmcnt<2 * NI8>(); break;
          case 3: wait_vmcnt<3 * NI8>(); break;
          case 4: wait_vmcnt<4 * NI8>(); break;
          case 5: wait_vmcnt<5 * NI8>(); break;
          case 6: wait_vmcnt<6 * NI8>(); break;
          default: wait_vmcnt<7 * NI8>(); break;
        }
      };
#pragma unroll
      for (int pj = 0; pj < kStages8; ++pj)
        if (pj < nvt) {
          issue8(pj, pj, false);
          issue8(pj, pj, true);
        }
      for (int vt = 0; vt < nvt; ++vt) {
        if (vt == nt0 && nt1 > 0 && sel1 != sel0) park_first();  // the second item starts
        const bool second = vt >= nt0;
        const int jt = second ? vt - nt0 : vt;
        const int n_pass = second ? n1 : n0;
        const int st = vt & (kStages8 - 1);
        const char* kst = wave_lds + st * 2 * TILE8;
        const char* vst = kst + TILE8;
        const int tok0 = ((second ? w1 : wave) + kWaves * jt) * kTile;
        const int after = (nvt - 1 - vt) < (kStages8 - 1) ? (nvt - 1 - vt) : (kStages8 - 1);
        const bool refill = vt + kStages8 < nvt;
        wait_units(1 + 2 * after);
        f32x4 s_acc[2];
#pragma unroll
        for (int th = 0; th < 2; ++th) {
          s_acc[th] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int row = 16 * th + hl;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int c = (2 * ks + (g >> 1)) ^ swz8<D>(row);
            const uint2 raw = *reinterpret_cast<const uint2*>(kst + row * D + c * 16 + 8 * (g & 1));
            x8 kf;
            const f32x2_t a0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)raw.x, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)raw.x, true);
            const f32x2_t b0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)raw.y, false), b1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)raw.y, true);
            kf[0] = H::from_f32(a0[0]); kf[1] = H::from_f32(a0[1]); kf[2] = H::from_f32(a1[0]); kf[3] = H::from_f32(a1[1]);
            kf[4] = H::from_f32(b0[0]); kf[5] = H::from_f32(b0[1]); kf[6] = H::from_f32(b1[0]); kf[7] = H::from_f32(b1[1]);
            s_acc[th] = H::mfma16(kf, qf[ks], s_acc[th]);
          }
        }
        wait_lgkmcnt0();
        if (refill) issue8(vt + kStages8, st, false);
        float sv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sv[i] = s_acc[i >> 2][i & 3] * (has_cap ? a.sm_scale : scale_log2);
        if (has_cap) {
#pragma unroll
          for (int i = 0; i < 8; ++i) sv[i] = a.logit_cap * tanhf(sv[i] / a.logit_cap) * kLog2e;
        }
        float m_tile = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool valid = (tok0 + 16 * (i >> 2) + 4 * g + (i & 3)) < n_pass;
          sv[i] = valid ? sv[i] : -INFINITY;
          m_tile = fmaxf(m_tile, sv[i]);
        }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 16));
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);
        const float alpha = exp2f(m_run - m_new);
        float psum = 0.f;
        float pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          pv[i] = exp2f(sv[i] - m_new);
          psum += pv[i];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        int p_lo = __builtin_amdgcn_cvt_pk_fp8_f32(pv[0], pv[1], 0, false);
        p_lo = __builtin_amdgcn_cvt_pk_fp8_f32(pv[2], pv[3], p_lo, true);
        int p_hi = __builtin_amdgcn_cvt_pk_fp8_f32(pv[4], pv[5], 0, false);
        p_hi = __builtin_amdgcn_cvt_pk_fp8_f32(pv[6], pv[7], p_hi, true);
        const long pf8 = (long)(((unsigned long)(unsigned)p_hi << 32) | (unsigned long)(unsigned)p_lo);
        if (__ballot(alpha != 1.f) != 0) {
#pragma unroll
          for (int i = 0; i < NDV; ++i) o_acc[i] *= alpha;
        }
        wait_units(2 * after + (refill ? 1 : 0));
        {
          const int vrow = 8 * g + ((lane & 15) >> 1);
          const char* vrp = vst + vrow * D + 8 * (lane & 1);
          const int vsw = swz8<D>(vrow);
#pragma unroll
          for (int dvb = 0; dvb < NDV; ++dvb) {
            typedef int v2i_t __attribute__((ext_vector_type(2)));
            const v2i_t vr = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
                (__attribute__((address_space(3))) v2i_t*)(vrp + ((dvb ^ vsw) * 16)));
            const long vf8 = (long)(((unsigned long)(unsigned)vr[1] << 32) | (unsigned long)(unsigned)vr[0]);
            o_acc[dvb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(vf8, pf8, o_acc[dvb], 0, 0, 0);
          }
        }
        wait_lgkmcnt0();
        if (refill) issue8(vt + kStages8, st, true);
      }
      return;
    }
    auto issue = [&](int vt, int stage, bool is_v) __attribute__((always_inline)) {
      const bool second = vt >= nt0;
      const int jt = second ? vt - nt0 : vt;
      const int sel = second ? sel1 : sel0;
      const int off = second ? off1 : off0;
      const int n = second ? n1 : n0;
      const int tok0 = ((second ? w1 : wave) + kWaves * jt) * kTile;
      const char* gb = sel ? (is_v ? it[1].vbase : it[1].kbase) : (is_v ? it[0].vbase : it[0].kbase);
      const int64_t rb = is_v ? v_row_bytes : k_row_bytes;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(
          lds_addr_of(wave_lds + stage * STAGE_BYTES + (is_v ? TILE_BYTES : 0)));
      int32_t tok[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int tp = tok0 + i * ROWS_PER_DMA + dma_row;
        tp = tp < n ? tp : n - 1;
        tok[i] = idx_lds[off + tp];
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = swz_chunk<D>(dma_pos, i * ROWS_PER_DMA + dma_row);
        kv_dma16(gb + (int64_t)tok[i] * rb + c * 16, dst + i * 1024);
      }
    };
    if (nvt > 0) {
      issue(0, 0, false);
      issue(0, 0, true);
    }
    if (nvt > 1) {
      issue(1, 1, false);
      issue(1, 1, true);
    }
    for (int vt = 0; vt < nvt; ++vt) {
      if (vt == nt0 && nt1 > 0 && sel1 != sel0) park_first();  // the second item starts
      const bool second = vt >= nt0;
      const int jt = second ? vt - nt0 : vt;
      const int n_pass = second ? n1 : n0;
      const int st = vt & 1;
      const char* kst = wave_lds + st * STAGE_BYTES;
      const char* vst = kst + TILE_BYTES;
      const int tok0 = ((second ? w1 : wave) + kWaves * jt) * kTile;
      const bool more1 = vt + 1 < nvt;
      const bool more2 = vt + 2 < nvt;

      if (more1) wait_vmcnt<3 * NI>(); else wait_vmcnt<NI>();
      f32x4 s_acc[2];
#pragma unroll
      for (int th = 0; th < 2; ++th) {
        s_acc[th] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int row = 16 * th + hl;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int c = swz_chunk<D>(4 * ks + g, row);
          const x8 kf = *reinterpret_cast<const x8*>(kst + row * ROWB + c * 16);
          s_acc[th] = H::mfma16(kf, qf[ks], s_acc[th]);
        }
      }
      wait_lgkmcnt0();
      if (more2) issue(vt + 2, st, false);

      float sv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) sv[i] = s_acc[i >> 2][i & 3] * (has_cap ? a.sm_scale : scale_log2);
      if (has_cap) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sv[i] = a.logit_cap * tanhf(sv[i] / a.logit_cap) * kLog2e;
      }
      float m_tile = -INFINITY;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool valid = (tok0 + 16 * (i >> 2) + 4 * g + (i & 3)) < n_pass;
        sv[i] = valid ? sv[i] : -INFINITY;
        m_tile = fmaxf(m_tile, sv[i]);
      }
      m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 16));
      m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
      const float m_new = fmaxf(m_run, m_tile);
      const float alpha = exp2f(m_run - m_new);
      float psum = 0.f;
      x8 pf;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float p = exp2f(sv[i] - m_new);
        psum += p;
        pf[i] = H::from_f32(p);
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < NDV; ++i) o_acc[i] *= alpha;

      if (more2) wait_vmcnt<3 * NI>(); else if (more1) wait_vmcnt<2 * NI>(); else wait_vmcnt<0>();
      {
        const int q4 = (lane >> 2) & 3;
        const int p4 = lane & 3;
        const int row_lo = 4 * g + q4;
#pragma unroll
        for (int dvb = 0; dvb < NDV; ++dvb) {
          const int c = 2 * dvb + (p4 >> 1);
          const int off = swz_chunk<D>(c, row_lo) * 16 + 8 * (p4 & 1);
          const x4 v_lo = H::ds_read_tr(vst + row_lo * ROWB + off);
          const x4 v_hi = H::ds_read_tr(vst + (row_lo + 16) * ROWB + off);
          x8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vf[j] = v_lo[j];
            vf[4 + j] = v_hi[j];
          }
          o_acc[dvb] = H::mfma16(vf, pf, o_acc[dvb]);
        }
      }
      wait_lgkmcnt0();
      if (more2) issue(vt + 2, st, true);
    }
  };

  // One call site for `stream` (a second inlined copy of the loop would double the kernel): either the fused pair,
  // or, for contexts longer than the staged window, one item at a time in windows of the whole LDS window.
  const int len0 = it[0].len, len1 = it[1].len;  // len1 may be -1 (no second item)
  const bool fused = len0 <= CAP && len1 <= CAP;
  int cur = 0, p0 = 0;  // long path: item and window start
  for (;;) {
    int sel0, off0, n0, sel1, off1, n1;
    if (fused) {
      sel0 = 0, off0 = 0, n0 = len0 > 0 ? len0 : 0, sel1 = 1, off1 = CAP, n1 = len1 > 0 ? len1 : 0;
    } else {
      while (cur < 2 && p0 >= (cur ? len1 : len0)) {  // next item (lengths are workgroup-uniform)
        ++cur;
        p0 = 0;
      }
      if (cur >= 2) break;
      if (cur == 1 && !parked) park_first();
      const int len_c = cur ? len1 : len0;
      const int n = (len_c - p0) < 2 * CAP ? (len_c - p0) : 2 * CAP;
      __syncthreads();  // the previous window is no longer read
      const IdxT* src = reinterpret_cast<const IdxT*>(a.indices) + (cur ? it[1].base : it[0].base) + p0;
      for (int e = tid; e < n; e += 256) idx_lds[e] = (int32_t)src[e];
      __syncthreads();
      sel0 = sel1 = cur, off0 = off1 = 0, n0 = n, n1 = 0;
      p0 += n;
    }
    stream(sel0, off0, n0, sel1, off1, n1);
    if (fused) break;
  }
  if (!parked) park_first();  // this wave never crossed into a second item: what it streamed belongs to the first

  // ---- merges: first item from *_first, second from the running state
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  l_first += __shfl_xor(l_first, 16);
  l_first += __shfl_xor(l_first, 32);
  // both items' partials go to LDS behind ONE barrier (the dead rings hold 2 x [wave][16][D] floats + m, l), then every
  // thread finishes output elements of both
  constexpr int MRG = kWaves * 16 * D + 2 * kWaves * 16 + 16;  // floats per item (+ the new token's scores, FUSED)
  float* mrg = reinterpret_cast<float*>(smem);
  __syncthreads();  // every wave is done with its ring
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (it[i].len < 0 || (it[i].len == 0 && !newtok[i])) continue;
    float* mrg_o = mrg + i * MRG;
    float* mrg_m = mrg_o + kWaves * 16 * D;
    float* mrg_l = mrg_m + kWaves * 16;
    if (FUSED && wave == 0 && g == 0) mrg_l[kWaves * 16 + hl] = s_new[i];
    if (hl < it[i].nh) {
      float* dst = mrg_o + (wave * 16 + hl) * D;
#pragma unroll
      for (int dvb = 0; dvb < NDV; ++dvb)
        *reinterpret_cast<f32x4*>(dst + dvb * 16 + 4 * g) = i == 0 ? o_first[dvb] : o_acc[dvb];
      if (g == 0) {
        mrg_m[wave * 16 + hl] = i == 0 ? m_first : m_run;
        mrg_l[wave * 16 + hl] = i == 0 ? l_first : l_run;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int len = it[i].len;
    if (len < 0) break;  // no such item (workgroup-uniform)
    const int nh = it[i].nh;
    if (len == 0 && !newtok[i]) {  // empty sequence: zero rows
      if (tid < nh) {
        T* o = reinterpret_cast<T*>(a.out) + (int64_t)it[i].b * a.o_sb + (int64_t)(it[i].h0 + tid) * a.o_sh;
        for (int d = 0; d < D; ++d) {
          if constexpr (QOUT) store_coherent16(o + d, H::from_f32(0.f));
          else o[d] = H::from_f32(0.f);
        }
      }
      continue;
    }
    const float* mrg_o = mrg + i * MRG;
    const float* mrg_m = mrg_o + kWaves * 16 * D;
    const float* mrg_l = mrg_m + kWaves * 16;
    float amx = 0.f;  // max |stored output| of this thread's elements (row_absmax)
    for (int e = tid; e < nh * D; e += kWaves * 64) {
      const int h = e / D;
      const int dv = e - h * D;
      float M = -INFINITY;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) M = fmaxf(M, mrg_m[w * 16 + h]);
      float mn = -INFINITY;
      if constexpr (FUSED) {
        mn = newtok[i] ? mrg_l[kWaves * 16 + h] : -INFINITY;
        M = fmaxf(M, mn);
      }
      float L = 0.f, val = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        const float f = exp2f(mrg_m[w * 16 + h] - M);
        L += mrg_l[w * 16 + h] * f;
        val += mrg_o[(w * 16 + h) * D + dv] * f;
      }
      if constexpr (FUSED) {  // the new token: one more partial state, (m, l, o) = (score, 1, v); dv == tid % D here
        const float f = exp2f(mn - M);
        L += f;
        val += v_new[i] * f;
      }
      const T ov = H::from_f32(val / L);
      T* op = reinterpret_cast<T*>(a.out) + (int64_t)it[i].b * a.o_sb + (int64_t)(it[i].h0 + h) * a.o_sh + dv;
      if constexpr (QOUT) store_coherent16(op, ov);
      else *op = ov;
      amx = fmaxf(amx, fabsf(H::to_f32(ov)));
    }
    if (a.row_absmax != nullptr) {  // non-negative floats order like their bit patterns: an integer atomic max is exact
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) amx = fmaxf(amx, __shfl_xor(amx, off));
      if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(a.row_absmax) + it[i].b, __float_as_uint(amx));
    }
  }
  if constexpr (QOUT) {
    // LDS from here on (the merge staging is dead behind the barrier): [0, 8) last-arriver flags of the two items,
    // [16, 32) reduction scratch, [64, 64 + 2 Hq D) the row being quantised
    int* s_last = reinterpret_cast<int*>(smem);
    float* s_red = reinterpret_cast<float*>(smem + 16);
    T* row = reinterpret_cast<T*>(smem + 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's output stores are acknowledged
    __syncthreads();                                  // ... and so are everyone's
    if (tid == 0) {
      const int target = a.num_kv_heads * nhb;  // items per request
      const bool two = it[1].len >= 0;
      const bool same = two && it[1].b == it[0].b;
      const int old0 = __hip_atomic_fetch_add(a.merge_counters + it[0].b, same ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last0 = old0 + (same ? 2 : 1) == target;
      bool last1 = false;
      if (two && !same)
        last1 = __hip_atomic_fetch_add(a.merge_counters + it[1].b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == target;
      // the next launch starts from zero again (nobody else touches a completed counter before then)
      if (last0) __hip_atomic_store(a.merge_counters + it[0].b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (last1) __hip_atomic_store(a.merge_counters + it[1].b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last[0] = last0;
      s_last[1] = last1;
    }
    __syncthreads();
    const int R = a.num_heads * D;
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
      if (!s_last[i]) continue;  // workgroup-uniform
      const int b = it[i].b;
      // 8-byte coherent loads (four 16-bit values), all of a thread's in flight together
      constexpr int U = 4;
      float amax = 0.f;
      for (int v0 = tid; v0 < (R >> 2); v0 += U * 256) {
        unsigned long long w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int v = v0 + u * 256 < (R >> 2) ? v0 + u * 256 : (R >> 2) - 1;
          const int h = (4 * v) / D, d = 4 * v - h * D;
          w[u] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(
                                       reinterpret_cast<const T*>(a.out) + (int64_t)b * a.o_sb + (int64_t)h * a.o_sh + d),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int v = v0 + u * 256;
          if (v < (R >> 2)) {
            *reinterpret_cast<unsigned long long*>(row + 4 * v) = w[u];
            const x4 xv = __builtin_bit_cast(x4, w[u]);
#pragma unroll
            for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(H::to_f32(xv[j])));
          }
        }
      }
      __syncthreads();
      quant_staged_row<DTYPE>(row, R, amax, a.mq_out_q, a.mq_out_s, b, tid, 256, s_red);
      __syncthreads();  // the row and the scratch are free again
    }
  }
}

// --------------------------------------------------------------------------------------
// Generic fallback: one wave per (request, head, split); any head sizes up to 1024.
// Lane l owns elements l, l+64, ...  Correctness path for the odd shapes the reference
// tests (D = 13, 33/55, 80, 512, 576/512 ...), not a performance path.
template <int DTYPE, typename IdxT, bool DIRECT_OUT>
__global__ __launch_bounds__(64) void decode_generic_kernel(DecodeArgs a, int D, int Dv) {
  using H = Half16<DTYPE>;
  using T = typename H::T;
  constexpr int MAXE = 16;
  const int lane = threadIdx.x;
  int bid = blockIdx.x;
  const int split = bid % a.num_splits;
  bid /= a.num_splits;
  const int h = bid % a.num_heads;
  const int b = bid / a.num_heads;
  const int kvh = h / a.group;

  int64_t base;
  int s0, s1;
  split_range(a, b, split, base, s0, s1);
  if (s0 >= s1) {
    if (DIRECT_OUT) {
      T* o = reinterpret_cast<T*>(a.out) + (int64_t)b * a.o_sb + (int64_t)h * a.o_sh;
      for (int d = lane; d < Dv; d += 64) o[d] = H::from_f32(0.f);
    } else if (lane == 0) {
      a.mid_lse[(int64_t)b * a.ml_sb + (int64_t)h * a.ml_sh + (int64_t)split * a.ml_ss] = -INFINITY;
    }
    return;
  }
  float qv[MAXE], acc[MAXE];
  const T* qp = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.q_sb + (int64_t)h * a.q_sh;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int d = lane + 64 * i;
    qv[i] = d < D ? H::to_f32(qp[d]) : 0.f;
    acc[i] = 0.f;
  }
  float m_run = -INFINITY, l_run = 0.f;
  const IdxT* idx = reinterpret_cast<const IdxT*>(a.indices) + base;
  const T* kb = reinterpret_cast<const T*>(a.k) + (int64_t)kvh * a.k_sh;
  const T* vb = reinterpret_cast<const T*>(a.v) + (int64_t)kvh * a.v_sh;
  for (int n = s0; n < s1; ++n) {
    const int64_t tok = (int64_t)idx[n];
    const T* kp = kb + tok * a.k_sn;
    const T* vp = vb + tok * a.v_sn;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int d = lane + 64 * i;
      if (d < D) s += qv[i] * H::to_f32(kp[d]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    s *= a.sm_scale;
    if (a.logit_cap > 0.f) s = a.logit_cap * tanhf(s / a.logit_cap);
    const float m_new = fmaxf(m_run, s);
    const float alpha = expf(m_run - m_new);
    const float p = expf(s - m_new);
    const float pr = H::to_f32(H::from_f32(p));
    l_run = l_run * alpha + p;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int d = lane + 64 * i;
      if (d < Dv) acc[i] = acc[i] * alpha + pr * H::to_f32(vp[d]);
    }
  }
  const float inv = 1.f / l_run;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int d = lane + 64 * i;
    if (d < Dv) {
      if (DIRECT_OUT)
        reinterpret_cast<T*>(a.out)[(int64_t)b * a.o_sb + (int64_t)h * a.o_sh + d] = H::from_f32(acc[i] * inv);
      else
        a.mid_o[(int64_t)b * a.mo_sb + (int64_t)h * a.mo_sh + (int64_t)split * a.mo_ss + d] = acc[i] * inv;
    }
  }
  if (!DIRECT_OUT && lane == 0)
    a.mid_lse[(int64_t)b * a.ml_sb + (int64_t)h * a.ml_sh + (int64_t)split * a.ml_ss] = m_run + logf(l_run);
}

// --------------------------------------------------------------------------------------
// Merge of the kv-splits: out = sum_s exp(lse_s - M) o_s / sum_s exp(lse_s - M)
// (decode_accumulate_kv_splits, decode.cpp:812-860; _fwd_kernel_stage2, decode_attention.py:491-548)
template <int DTYPE>
__global__ __launch_bounds__(64) void decode_merge_kernel(DecodeArgs a, int Dv) {
  using H = Half16<DTYPE>;
  using T = typename H::T;
  const int lane = threadIdx.x;
  const int h = blockIdx.x % a.num_heads;
  const int b = blockIdx.x / a.num_heads;
  const float* lse = a.mid_lse + (int64_t)b * a.ml_sb + (int64_t)h * a.ml_sh;
  const float* mo = a.mid_o + (int64_t)b * a.mo_sb + (int64_t)h * a.mo_sh;
  float M = -INFINITY;
  for (int s = 0; s < a.num_splits; ++s) M = fmaxf(M, lse[(int64_t)s * a.ml_ss]);
  T* o = reinterpret_cast<T*>(a.out) + (int64_t)b * a.o_sb + (int64_t)h * a.o_sh;
  if (M == -INFINITY) {  // empty sequence
    for (int d = lane; d < Dv; d += 64) o[d] = H::from_f32(0.f);
    return;
  }
  float L = 0.f;
  for (int s = 0; s < a.num_splits; ++s) L += expf(lse[(int64_t)s * a.ml_ss] - M);
  const float inv = 1.f / L;
  for (int d = lane; d < Dv; d += 64) {
    float acc = 0.f;
    for (int s = 0; s < a.num_splits; ++s) {
      const float l = lse[(int64_t)s * a.ml_ss];
      if (l != -INFINITY) acc += expf(l - M) * mo[(int64_t)s * a.mo_ss + d];
    }
    o[d] = H::from_f32(acc * inv);
  }
}


// Merge of the kv-splits fused with the per-token FP8 quant of the attention output (what feeds o_proj in the w8a8
// model): one workgroup per request merges all heads with decode_merge_kernel's arithmetic, rounds to the 16-bit
// dtype, takes the row absmax and writes e4m3 + scale exactly as sgl_per_token_quant_fp8 would on that 16-bit row
// (per_token_quant_fp8.cu:15-87: scale = absmax / 448, q = clamp(x * (1 / scale))) -- bit-identical to the two
// launches it replaces.  The 16-bit row itself is written only if `a.out` is given.
template <int DTYPE, bool COHERENT>
__device__ void merge_quant_row(const DecodeArgs& a, int Dv, uint8_t* __restrict__ out_q, float* __restrict__ out_s, int b,
                                int tid, int NT, char* smem, float* red) {
  using H = Half16<DTYPE>;
  using T = typename H::T;
  auto ld = [](const float* p) __attribute__((always_inline)) { return COHERENT ? load_coherent(p) : *p; };
  const int Hq = a.num_heads, S = a.num_splits, R = Hq * Dv;
  T* row = reinterpret_cast<T*>(smem);                                   // [R] merged output, 16-bit
  float* ew = reinterpret_cast<float*>(smem + ((R * 2 + 15) & ~15));     // [Hq][S] exp(lse - M); then [Hq] 1 / L
  float* inv = ew + Hq * S;
  // all LSEs in one round trip (one (head, split) pair per thread), then the per-head weights from LDS
  for (int i = tid; i < Hq * S; i += NT) {
    const int h = i / S, sp = i - h * S;
    ew[i] = ld(a.mid_lse + (int64_t)b * a.ml_sb + (int64_t)h * a.ml_sh + (int64_t)sp * a.ml_ss);
  }
  __syncthreads();
  for (int h = tid; h < Hq; h += NT) {
    float* lw = ew + h * S;  // in: lse; out: exp(lse - M)
    float M = -INFINITY;
    for (int s = 0; s < S; ++s) M = fmaxf(M, lw[s]);
    float L = 0.f;
    for (int s = 0; s < S; ++s) {
      const float l = lw[s];
      const float e = (M == -INFINITY || l == -INFINITY) ? 0.f : expf(l - M);  // 0: an empty split, partial undefined
      lw[s] = e;
      if (M != -INFINITY) L += expf(l - M);
    }
    inv[h] = M == -INFINITY ? 0.f : 1.f / L;  // empty sequence: zero row
  }
  __syncthreads();
  float amax = 0.f;
  auto finish = [&](int e, int h, int d, float acc) __attribute__((always_inline)) {
    const T o = H::from_f32(acc * inv[h]);
    row[e] = o;
    amax = fmaxf(amax, fabsf(H::to_f32(o)));
    if (a.out) reinterpret_cast<T*>(a.out)[(int64_t)b * a.o_sb + (int64_t)h * a.o_sh + d] = o;
  };
  if (S <= 8) {
    // four elements x up to eight splits: all of a thread's loads of an iteration in flight together (this row is usually
    // the whole job of the thread); the sum stays in split order
    for (int e0 = tid; e0 < R; e0 += 4 * NT) {
      float m[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NT < R ? e0 + u * NT : R - 1;
        const int h = e / Dv, d = e - h * Dv;
        const float* mo = a.mid_o + (int64_t)b * a.mo_sb + (int64_t)h * a.mo_sh + d;
#pragma unroll
        for (int k = 0; k < 8; ++k) m[u][k] = ld(mo + (int64_t)(k < S ? k : S - 1) * a.mo_ss);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NT;
        if (e < R) {
          const int h = e / Dv, d = e - h * Dv;
          const float* w = ew + h * S;
          float acc = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (k < S) {
              const float wk = w[k];
              if (wk != 0.f) acc += wk * m[u][k];
            }
          finish(e, h, d, acc);
        }
      }
    }
  } else {
    for (int e = tid; e < R; e += NT) {
      const int h = e / Dv, d = e - h * Dv;
      const float* mo = a.mid_o + (int64_t)b * a.mo_sb + (int64_t)h * a.mo_sh + d;
      const float* w = ew + h * S;
      float acc = 0.f;
      for (int s0 = 0; s0 < S; s0 += 8) {  // eight splits' loads in flight; the sum stays in split order
        float m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) m[u] = ld(mo + (int64_t)(s0 + u < S ? s0 + u : S - 1) * a.mo_ss);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (s0 + u < S) {
            const float wu = w[s0 + u];
            if (wu != 0.f) acc += wu * m[u];
          }
      }
      finish(e, h, d, acc);
    }
  }
  if (out_q == nullptr) return;  // merge only (the 16-bit row went to a.out)
  quant_staged_row<DTYPE>(row, R, amax, out_q, out_s, b, tid, NT, red);
}

template <int DTYPE, int NT>
__global__ __launch_bounds__(NT) void decode_merge_quant_kernel(DecodeArgs a, int Dv, uint8_t* __restrict__ out_q,
                                                                float* __restrict__ out_s) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[NT / 64];
  merge_quant_row<DTYPE, false>(a, Dv, out_q, out_s, blockIdx.x, threadIdx.x, NT, smem, red);
}

template <int D, int kWaves, int KV8 = 0>
constexpr int mfma_lds_bytes() {
  const int ring = KV8 ? ((KV8 == 2) ? 2 : 4) * 2 * kTile * D : kStages * 2 * kTile * D * 2;
  return kWaves * ring + ((KV8 == 2) ? 2 * kMaxIdx : (kWaves >= 4) ? 4 * kMaxIdx : kMaxIdx) * 4;
}

// Waves per workgroup.  2-wave workgroups need <= 72 KB of LDS, so two of them share a CU and
// one's prologue/merge overlaps the other's streaming; 4-wave workgroups (one per CU) finish a
// single long sequence sooner.  SGL_MI355_DECODE_WAVES=2|4 overrides (tuning aid).
inline int pick_waves(int64_t workgroups) {
  static const int forced = [] {
    const char* e = getenv("SGL_MI355_DECODE_WAVES");
    return e ? atoi(e) : 0;
  }();
  if (forced == 2 || forced == 4) return forced;
  return workgroups > 1024 ? 2 : 4;
}

#if SGLM_DEC_TIMING
inline unsigned long long* dec_timing_buffer() {
  static unsigned long long* buf = [] {
    void* p = nullptr;
    const size_t n = (size_t)kDecTimingWgs * kDecTimingStamps * 8;
    if (hipMalloc(&p, n) != hipSuccess) return (unsigned long long*)nullptr;
    (void)hipMemset(p, 0, n);
    return (unsigned long long*)p;
  }();
  return buf;
}
#endif

template <typename K>
int set_max_lds(K kernel, int bytes) {
  return check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes),
      "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
}

template <int DTYPE, int D, typename IdxT, bool DIRECT, int kWaves, int KV8 = 0, bool E5 = false>
int launch_mfma_w(const DecodeArgs& a, int64_t grid, hipStream_t stream) {
  auto kern = decode_mfma_kernel<DTYPE, D, IdxT, DIRECT, kWaves, KV8, E5>;
  constexpr int lds = mfma_lds_bytes<D, kWaves, KV8>();
  static int attr_rc = set_max_lds(kern, lds);
  if (attr_rc != 0) return attr_rc;
#if SGLM_DEC_TIMING
  DecodeArgs at = a;
  at.tstamp = dec_timing_buffer();
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kWaves * 64), lds, stream, at);
  return check_hip(hipGetLastError(), "decode_mfma_kernel launch");
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kWaves * 64), lds, stream, a);
  return check_hip(hipGetLastError(), "decode_mfma_kernel launch");
}

// More items than CUs and one split: pairs of items per workgroup (decode_mfma_pair_kernel).
// SGL_MI355_DECODE_PAIR=0|1 overrides (tuning aid).
inline bool pair_eligible(const DecodeArgs& a, int64_t grid) {
  static const int pair_env = [] { const char* e = getenv("SGL_MI355_DECODE_PAIR"); return e ? atoi(e) : -1; }();
  return a.num_splits == 1 && a.num_kv_splits == nullptr && (pair_env >= 0 ? pair_env != 0 : grid > 256);
}

inline int pair_deal_default() {  // DecodeArgs::pair_deal
  static const int v = [] { const char* e = getenv("SGL_MI355_DECODE_PAIR_DEAL"); return e ? atoi(e) : 1; }();
  return v;
}

// set by sgl_mi355_decode_attention_qkv_partials around its call of the regular entry point
[[maybe_unused]] thread_local const FusedQkv* tl_fq = nullptr;  // (read only in builds with SGLM_OPTIN_FUSIONS)
[[maybe_unused]] thread_local bool tl_fq_used = false;
// set by sgl_mi355_decode_attention_newkv around its call of the regular entry point (FUSED == 2 of the pair kernel)
[[maybe_unused]] thread_local const FusedQkv* tl_newkv = nullptr;
[[maybe_unused]] thread_local bool tl_newkv_used = false;
// set by sgl_mi355_decode_attention_absmax around its call of the regular entry point
thread_local float* tl_row_absmax = nullptr;
// set by sgl_mi355_decode_attention_merged around its call of the regular entry point
struct MergeFused { int32_t* counters; uint8_t* out_q; float* out_s; };
thread_local MergeFused tl_merge{nullptr, nullptr, nullptr};
// set by sgl_mi355_decode_attention_quant around its call of the regular entry point (one split: pairs-of-items kernel, QOUT)
[[maybe_unused]] thread_local MergeFused tl_pair_quant{nullptr, nullptr, nullptr};
[[maybe_unused]] thread_local bool tl_pair_quant_used = false;

template <int DTYPE, int D, typename IdxT, bool DIRECT>
int launch_mfma(const DecodeArgs& a, int64_t batch, hipStream_t stream) {
  const int nhb = (a.group + 15) / 16;
  const int64_t grid = batch * a.num_kv_heads * nhb * a.num_splits;
  if constexpr (DIRECT) {
    // More items than CUs, one split, 16-bit pool: pairs of items per workgroup in one DMA stream (see
    // decode_mfma_pair_kernel).  Measured, bs=64 x 8 kv heads, S = 256 / 1024 / 2048 / 8192: 20.2 / 53.4 / 99.0 / 352 us
    // vs 25.6 / 57.5 / 103.0 / 375 one item per workgroup; 2048 items (MHA, 32 kv heads): 363 vs 375 us with the 2-wave
    // workgroups.  SGL_MI355_DECODE_PAIR=0|1 overrides (tuning aid).
    const bool pair = pair_eligible(a, grid);
#if SGLM_OPTIN_FUSIONS
    if (pair && !a.kv8 && tl_fq != nullptr) {  // + the qkv GEMM epilogue, RoPE and the KV write in the prologue
      auto kern = decode_mfma_pair_kernel<DTYPE, D, IdxT, false, true>;
      constexpr int lds = mfma_lds_bytes<D, 4, 0>();
      static int attr_rc = set_max_lds(kern, lds);
      if (attr_rc != 0) return attr_rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)((grid + 1) / 2)), dim3(256), lds, stream, a, (int)grid, *tl_fq);
      tl_fq_used = true;
      return check_hip(hipGetLastError(), "decode_mfma_pair_kernel (qkv partials) launch");
    }
    if (pair && !a.kv8 && tl_pair_quant.counters != nullptr) {  // + the per-token FP8 quant of the finished rows
      auto kern = decode_mfma_pair_kernel<DTYPE, D, IdxT, false, false, true>;
      constexpr int lds = mfma_lds_bytes<D, 4, 0>();
      static int attr_rc = set_max_lds(kern, lds);
      if (attr_rc != 0) return attr_rc;
      if (64 + (int64_t)a.num_heads * D * 2 > lds) {
        set_error("decode_attention_quant: a row of %d heads does not fit the workgroup's LDS", a.num_heads);
        return SGL_MI355_ERR_UNSUPPORTED;
      }
      DecodeArgs aq = a;
      aq.merge_counters = tl_pair_quant.counters;
      aq.mq_out_q = tl_pair_quant.out_q;
      aq.mq_out_s = tl_pair_quant.out_s;
      hipLaunchKernelGGL(kern, dim3((unsigned)((grid + 1) / 2)), dim3(256), lds, stream, aq, (int)grid, FusedQkv{});
      tl_pair_quant_used = true;
      return check_hip(hipGetLastError(), "decode_mfma_pair_kernel (fp8 out) launch");
    }
    if (pair && !a.kv8 && tl_newkv != nullptr) {  // + the new token's K / V rows: pool write + one more partial state at the merge
      auto kern = decode_mfma_pair_kernel<DTYPE, D, IdxT, false, 2>;
      constexpr int lds = mfma_lds_bytes<D, 4, 0>();
      static int attr_rc = set_max_lds(kern, lds);
      if (attr_rc != 0) return attr_rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)((grid + 1) / 2)), dim3(256), lds, stream, a, (int)grid, *tl_newkv);
      tl_newkv_used = true;
      return check_hip(hipGetLastError(), "decode_mfma_pair_kernel (new-token K/V) launch");
    }
#endif  // SGLM_OPTIN_FUSIONS
    if (pair && !a.kv8) {
      auto kern = decode_mfma_pair_kernel<DTYPE, D, IdxT>;
      constexpr int lds = mfma_lds_bytes<D, 4, 0>();
      static int attr_rc = set_max_lds(kern, lds);
      if (attr_rc != 0) return attr_rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)((grid + 1) / 2)), dim3(256), lds, stream, a, (int)grid, FusedQkv{});
      return check_hip(hipGetLastError(), "decode_mfma_pair_kernel launch");
    }
    static const int pair8_env = [] { const char* e = getenv("SGL_MI355_DECODE_PAIR_KV8"); return e ? atoi(e) : 0; }();
    // e4m3 pool: opt-in (SGL_MI355_DECODE_PAIR_KV8=1).  Measured against the two-workgroups-per-CU layout at bs=64 x 8 kv
    // heads, S = 512 / 1024 / 2048 / 4096: 28.0 / 38.2 / 63.2 / 112.4 us vs 22.5 / 35.7 / 65.1 / 122.6 -- better only
    // from ~2k tokens, and the launcher cannot see the lengths.
    if (pair && a.kv8 == 1 && pair8_env) {
      auto kern = decode_mfma_pair_kernel<DTYPE, D, IdxT, true>;
      constexpr int lds = mfma_lds_bytes<D, 4, 1>();
      static int attr_rc = set_max_lds(kern, lds);
      if (attr_rc != 0) return attr_rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)((grid + 1) / 2)), dim3(256), lds, stream, a, (int)grid, FusedQkv{});
      return check_hip(hipGetLastError(), "decode_mfma_pair_kernel (e4m3 pool) launch");
    }
  }
  if (a.kv8 == 2)  // e5m2 pool: the two-workgroups-per-CU layout only (one set of instantiations; same structure)
    return launch_mfma_w<DTYPE, D, IdxT, DIRECT, 4, 2, true>(a, grid, stream);
  if (a.kv8) {
    // measured at bs=64 x 8 kv heads (512 workgroups), S = 512 / 2048 / 8192: two per CU 22 / 64 / 224 us, one per CU
    // with four stages 30 / 70 / 205 us.  SGL_MI355_DECODE_KV8_STAGES=2|4 overrides (tuning aid).
    static const int forced = [] {
      const char* e = getenv("SGL_MI355_DECODE_KV8_STAGES");
      return e ? atoi(e) : 0;
    }();
    const bool two_per_cu = forced == 2 || (forced != 4 && grid > 256);
    return two_per_cu ? launch_mfma_w<DTYPE, D, IdxT, DIRECT, 4, 2>(a, grid, stream)
                      : launch_mfma_w<DTYPE, D, IdxT, DIRECT, 4, 1>(a, grid, stream);
  }
  return pick_waves(grid) == 2 ? launch_mfma_w<DTYPE, D, IdxT, DIRECT, 2>(a, grid, stream)
                               : launch_mfma_w<DTYPE, D, IdxT, DIRECT, 4>(a, grid, stream);
}

template <int DTYPE, typename IdxT, bool DIRECT>
int launch_generic(const DecodeArgs& a, int64_t batch, int D, int Dv, hipStream_t stream) {
  const int64_t grid = batch * a.num_heads * a.num_splits;
  hipLaunchKernelGGL((decode_generic_kernel<DTYPE, IdxT, DIRECT>), dim3((unsigned)grid), dim3(64), 0, stream, a, D, Dv);
  return check_hip(hipGetLastError(), "decode_generic_kernel launch");
}

template <int DTYPE, typename IdxT>
int dispatch_stage1(const DecodeArgs& a, int64_t batch, int D, int Dv, bool direct, hipStream_t stream) {
  const bool aligned = (a.q_sb % 8 == 0) && (a.q_sh % 8 == 0) && (a.k_sn % 8 == 0) && (a.k_sh % 8 == 0) &&
                       (a.v_sn % 8 == 0) && (a.v_sh % 8 == 0) &&
                       (reinterpret_cast<uintptr_t>(a.q) % 16 == 0) && (reinterpret_cast<uintptr_t>(a.k) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(a.v) % 16 == 0);
  if (a.kv8 && !(D == Dv && (D == 128 || D == 64) && (a.k_sn % 16 == 0) && (a.k_sh % 16 == 0) && (a.v_sn % 16 == 0) &&
                 (a.v_sh % 16 == 0) && aligned)) {
    set_error("decode_attention: the FP8 KV cache path needs head sizes 64 / 128 (D == Dv) and 16-byte aligned rows");
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  if (D == Dv && aligned && (D == 128 || D == 64)) {
    if (D == 128)
      return direct ? launch_mfma<DTYPE, 128, IdxT, true>(a, batch, stream)
                    : launch_mfma<DTYPE, 128, IdxT, false>(a, batch, stream);
    return direct ? launch_mfma<DTYPE, 64, IdxT, true>(a, batch, stream)
                  : launch_mfma<DTYPE, 64, IdxT, false>(a, batch, stream);
  }
  return direct ? launch_generic<DTYPE, IdxT, true>(a, batch, D, Dv, stream)
                : launch_generic<DTYPE, IdxT, false>(a, batch, D, Dv, stream);
}

template <int DTYPE>
int run_decode(DecodeArgs a, int64_t batch, int D, int Dv, bool idx64, hipStream_t stream) {
  // a.out == nullptr: stage 1 only -- the caller merges (sgl_mi355_decode_merge_quant_fp8)
  const bool direct = (a.num_splits == 1 && a.num_kv_splits == nullptr && a.out != nullptr && a.merge_counters == nullptr);
  int rc = idx64 ? dispatch_stage1<DTYPE, int64_t>(a, batch, D, Dv, direct, stream)
                 : dispatch_stage1<DTYPE, int32_t>(a, batch, D, Dv, direct, stream);
  if (rc != 0 || direct || a.merge_counters != nullptr || a.out == nullptr) return rc;  // (fused: the last workgroup of a request merged)
  hipLaunchKernelGGL((decode_merge_kernel<DTYPE>), dim3((unsigned)(batch * a.num_heads)), dim3(64), 0, stream, a, Dv);
  return check_hip(hipGetLastError(), "decode_merge_kernel launch");
}

int check_common(
    int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t D, int64_t Dv, int64_t splits, int dtype) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "decode: dtype must be bf16 (0) or fp16 (1), got %d", dtype);
  SGLM_CHECK_ARG(batch >= 0 && num_heads > 0 && num_kv_heads > 0, "decode: bad sizes batch=%ld heads=%ld kv_heads=%ld",
                 (long)batch, (long)num_heads, (long)num_kv_heads);
  SGLM_CHECK_ARG(num_heads % num_kv_heads == 0, "decode: num_heads (%ld) must be a multiple of num_kv_heads (%ld)",
                 (long)num_heads, (long)num_kv_heads);
  SGLM_CHECK_ARG(D > 0 && Dv > 0 && D <= 1024 && Dv <= 1024, "decode: head sizes must be in [1,1024], got %ld/%ld",
                 (long)D, (long)Dv);
  SGLM_CHECK_ARG(splits >= 1 && splits <= 65535, "decode: num_kv_splits must be in [1,65535], got %ld", (long)splits);
  SGLM_CHECK_ARG(batch * num_heads * splits < (1ll << 31), "decode: grid too large");
  return 0;
}

}  // namespace
}  // namespace sglm

using namespace sglm;

// set by the *_fp8kv entry points around a call of the regular ones (same argument list, pool = e4m3 bytes)
static thread_local int tl_kv8 = 0;

extern "C" int sgl_mi355_decode_attention(
    const void* query, void* k_cache, void* v_cache, void* output, const void* key, const void* value,
    const int64_t* loc, float* attn_logits, const void* req_to_token, int req_to_token_is64,
    const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t num_seqs, int64_t max_context_len,
    int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v, int64_t num_kv_splits,
    int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
    int64_t v_stride_h, int64_t key_stride_n, int64_t key_stride_h, int64_t value_stride_n, int64_t value_stride_h,
    int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap, int dtype, void* stream) {
  int rc = check_common(num_seqs, num_heads, num_kv_heads, head_size, head_size_v, num_kv_splits, dtype);
  if (rc) return rc;
  if (num_seqs == 0) return 0;
  SGLM_CHECK_ARG(query && k_cache && v_cache && req_to_token && req_pool_indices && seq_lens,
                 "decode_attention: null tensor pointer");
  SGLM_CHECK_ARG(attn_logits != nullptr || (num_kv_splits == 1 && output != nullptr),
                 "decode_attention: attn_logits is required when num_kv_splits > 1 or output is null (stage 1 only)");
  SGLM_CHECK_ARG(!(tl_kv8 && loc != nullptr), "decode_attention_fp8kv: write the pool with set_kv_buffer_fp8 (loc must be null)");
  if (loc != nullptr) {
    SGLM_CHECK_ARG(key && value, "decode_attention: key/value are required when loc is given");
    rc = sgl_mi355_set_kv_buffer(k_cache, v_cache, key, value, loc, 1, num_seqs, num_kv_heads, head_size, head_size_v,
                                 k_stride_n, k_stride_h, v_stride_n, v_stride_h, key_stride_n, key_stride_h,
                                 value_stride_n, value_stride_h, dtype, stream);
    if (rc) return rc;
  }
  DecodeArgs a{};
  a.q = query; a.q_sb = q_stride_b; a.q_sh = q_stride_h;
  a.k = k_cache; a.k_sn = k_stride_n; a.k_sh = k_stride_h;
  a.v = v_cache; a.v_sn = v_stride_n; a.v_sh = v_stride_h;
  a.out = output; a.o_sb = o_stride_b; a.o_sh = o_stride_h;
  // attn_logits [B][Hq][splits][Dv+1], LSE in column Dv (decode.cpp:989-994)
  const int64_t l2 = head_size_v + 1, l1 = num_kv_splits * l2, l0 = num_heads * l1;
  a.mid_o = attn_logits; a.mo_sb = l0; a.mo_sh = l1; a.mo_ss = l2;
  a.mid_lse = attn_logits ? attn_logits + head_size_v : nullptr; a.ml_sb = l0; a.ml_sh = l1; a.ml_ss = l2;
  a.indices = req_to_token; a.mode = 1;
  a.req_pool_indices = req_pool_indices; a.seq_lens = seq_lens; a.r2t_stride = max_context_len;
  a.num_kv_splits = nullptr; a.num_splits = (int)num_kv_splits;
  a.split_align = 1;  // SPLIT_SIZE = div_up(seq_len, num_kv_splits) (decode.cpp:916)
  a.num_heads = (int)num_heads; a.num_kv_heads = (int)num_kv_heads; a.group = (int)(num_heads / num_kv_heads);
  a.sm_scale = sm_scale; a.logit_cap = logit_cap; a.kv8 = tl_kv8;
  a.row_absmax = tl_row_absmax;
  a.merge_counters = tl_merge.counters; a.mq_out_q = tl_merge.out_q; a.mq_out_s = tl_merge.out_s;
  a.pair_deal = pair_deal_default();
  hipStream_t s = as_stream(stream);
  return dtype == SGL_MI355_BF16
             ? run_decode<SGL_MI355_BF16>(a, num_seqs, (int)head_size, (int)head_size_v, req_to_token_is64 != 0, s)
             : run_decode<SGL_MI355_FP16>(a, num_seqs, (int)head_size, (int)head_size_v, req_to_token_is64 != 0, s);
}

extern "C" int sgl_mi355_decode_attention_fwd(
    const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
    const int32_t* kv_indices, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
    int64_t max_kv_splits, int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream) {
  int rc = check_common(batch, num_heads, num_kv_heads, head_size, head_size_v, max_kv_splits, dtype);
  if (rc) return rc;
  if (batch == 0) return 0;
  SGLM_CHECK_ARG(q && k_buffer && v_buffer && o && kv_indptr, "decode_attention_fwd: null tensor pointer");
  const bool direct = (max_kv_splits == 1 && num_kv_splits == nullptr);
  SGLM_CHECK_ARG(direct || (attn_logits && attn_lse), "decode_attention_fwd: attn_logits/attn_lse are required when splitting");
  DecodeArgs a{};
  a.q = q; a.q_sb = q_stride_b; a.q_sh = q_stride_h;
  a.k = k_buffer; a.k_sn = k_stride_n; a.k_sh = k_stride_h;
  a.v = v_buffer; a.v_sn = v_stride_n; a.v_sh = v_stride_h;
  a.out = o; a.o_sb = o_stride_b; a.o_sh = o_stride_h;
  a.pair_deal = pair_deal_default();
  // attn_logits [B][Hq][max_kv_splits][Dv], attn_lse [B][Hq][max_kv_splits] (triton_backend.py:207-216)
  a.mid_o = attn_logits; a.mo_ss = head_size_v; a.mo_sh = max_kv_splits * head_size_v; a.mo_sb = num_heads * a.mo_sh;
  a.mid_lse = attn_lse; a.ml_ss = 1; a.ml_sh = max_kv_splits; a.ml_sb = num_heads * max_kv_splits;
  a.indices = kv_indices; a.mode = 0; a.kv_indptr = kv_indptr;
  a.num_kv_splits = num_kv_splits; a.num_splits = (int)max_kv_splits;
  a.split_align = 32;  // _MIN_BLOCK_KV (decode_attention.py:303-307)
  a.num_heads = (int)num_heads; a.num_kv_heads = (int)num_kv_heads; a.group = (int)(num_heads / num_kv_heads);
  a.sm_scale = sm_scale; a.logit_cap = logit_cap; a.kv8 = tl_kv8;
  hipStream_t s = as_stream(stream);
  return dtype == SGL_MI355_BF16 ? run_decode<SGL_MI355_BF16>(a, batch, (int)head_size, (int)head_size_v, false, s)
                                 : run_decode<SGL_MI355_FP16>(a, batch, (int)head_size, (int)head_size_v, false, s);
}

// FP8 (e4m3fn) KV pool: identical argument lists, k/v pointers address bytes and their strides count bytes.
extern "C" int sgl_mi355_decode_attention_fp8kv(
    const void* query, void* k_cache, void* v_cache, void* output, float* attn_logits, const void* req_to_token,
    int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t num_seqs,
    int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v,
    int64_t num_kv_splits, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream) {
  tl_kv8 = 1;
  const int rc = sgl_mi355_decode_attention(query, k_cache, v_cache, output, nullptr, nullptr, nullptr, attn_logits,
                                            req_to_token, req_to_token_is64, req_pool_indices, seq_lens, num_seqs,
                                            max_context_len, num_heads, num_kv_heads, head_size, head_size_v,
                                            num_kv_splits, q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n,
                                            v_stride_h, 0, 0, 0, 0, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype,
                                            stream);
  tl_kv8 = 0;
  return rc;
}

// sgl_mi355_decode_attention (page-table form, no KV write) with kv-splits whose merge -- and, when out_q / out_s are
// given, the per-token FP8 quant of the merged row (what sgl_mi355_decode_merge_quant_fp8 computes) -- happens in the SAME
// launch: every workgroup counts itself in on merge_counters[b] (int32 [num_seqs], zero before the first call; left zero)
// after publishing its partial, and the workgroup that completes request b's count merges it.  Same bits as the two (or
// three) launches it replaces.  kv_format: 0 16-bit pool, 1 e4m3fn bytes, 2 e5m2 bytes.  `output` (16-bit row) and
// out_q / out_s are each optional, at least one is required.  Shapes outside the MFMA split kernel (head size 64 / 128 =
// v head size, 16-byte aligned rows) or rows that do not fit its LDS return SGL_MI355_ERR_UNSUPPORTED without launching.
extern "C" int sgl_mi355_decode_attention_merged(
    const void* query, void* k_cache, void* v_cache, void* output, void* out_q, float* out_s, float* attn_logits,
    int32_t* merge_counters, const void* req_to_token, int req_to_token_is64, const int64_t* req_pool_indices,
    const int64_t* seq_lens, int64_t num_seqs, int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads,
    int64_t head_size, int64_t num_kv_splits, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n,
    int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale,
    float logit_cap, int kv_format, int dtype, void* stream) {
  SGLM_CHECK_ARG(attn_logits != nullptr && merge_counters != nullptr, "decode_attention_merged: null attn_logits / merge_counters");
  SGLM_CHECK_ARG(output != nullptr || (out_q != nullptr && out_s != nullptr), "decode_attention_merged: no output given");
  SGLM_CHECK_ARG((out_q == nullptr) == (out_s == nullptr), "decode_attention_merged: out_q and out_s go together");
  SGLM_CHECK_ARG(kv_format >= 0 && kv_format <= 2, "decode_attention_merged: kv_format must be 0, 1 or 2");
  SGLM_CHECK_ARG(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0 && num_kv_splits >= 1,
                 "decode_attention_merged: bad head counts / splits");
  const int64_t esz = kv_format ? 16 : 8;  // 16-byte rows: elements per 16 bytes
  const bool aligned = q_stride_b % 8 == 0 && q_stride_h % 8 == 0 && k_stride_n % esz == 0 && k_stride_h % esz == 0 &&
                       v_stride_n % esz == 0 && v_stride_h % esz == 0 && reinterpret_cast<uintptr_t>(query) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(k_cache) % 16 == 0 && reinterpret_cast<uintptr_t>(v_cache) % 16 == 0;
  const int64_t R = num_heads * head_size;
  const int64_t lds_need = ((R * 2 + 15) & ~15ll) + (num_heads * num_kv_splits + num_heads) * 4;
  // the smallest dynamic LDS any variant of the split kernel is launched with (two waves, 16-bit tiles)
  const int64_t lds_have = (head_size == 128 ? mfma_lds_bytes<128, 2, 0>() : mfma_lds_bytes<64, 2, 0>()) - 64;  // (its last 64 bytes: flag + scratch)
  if (!((head_size == 128 || head_size == 64) && aligned && R % 8 == 0 && lds_need <= lds_have)) {
    set_error("decode_attention_merged: shape outside the fused form (head size 64 / 128, 16-byte aligned rows, "
              "num_heads * head_size * 2 + num_heads * (num_kv_splits + 1) * 4 bytes of LDS <= %ld)", (long)lds_have);
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  tl_merge = MergeFused{merge_counters, static_cast<uint8_t*>(out_q), out_s};
  tl_kv8 = kv_format;
  const int rc = sgl_mi355_decode_attention(query, k_cache, v_cache, output, nullptr, nullptr, nullptr, attn_logits,
                                            req_to_token, req_to_token_is64, req_pool_indices, seq_lens, num_seqs,
                                            max_context_len, num_heads, num_kv_heads, head_size, head_size, num_kv_splits,
                                            q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, 0, 0, 0,
                                            0, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype, stream);
  tl_kv8 = 0;
  tl_merge = MergeFused{nullptr, nullptr, nullptr};
  return rc;
}

// sgl_mi355_decode_attention (page-table form, no KV write, one split, 16-bit pool) that ALSO leaves
// row_absmax[b] = max over all heads and dims of |output[b]| (the stored 16-bit values) by atomic max into a buffer the
// caller zeroed: the absmax pass of sgl_per_token_quant_fp8 on the attention output (per_token_quant_fp8.cu:15-60, the
// w8a8 o_proj input) comes for free, and sgl_mi355_fp8_scaled_mm_partials_a16 quantises while it stages the activations.
// Only the pairs-of-items kernel has the epilogue (num_seqs * num_kv_heads > 256, head size 64 / 128, group <= 16): any
// other shape returns SGL_MI355_ERR_UNSUPPORTED without launching.
extern "C" int sgl_mi355_decode_attention_absmax(
    const void* query, void* k_cache, void* v_cache, void* output, float* row_absmax, const void* req_to_token,
    int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t num_seqs,
    int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t q_stride_b,
    int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b,
    int64_t o_stride_h, float sm_scale, float logit_cap, int dtype, void* stream) {
#if !SGLM_OPTIN_FUSIONS
  set_error("%s: an opt-in fusion, not in this build of the library (build with -DSGLM_OPTIN_FUSIONS=1)", "decode_attention_absmax");
  return SGL_MI355_ERR_UNSUPPORTED;
#else
  SGLM_CHECK_ARG(row_absmax != nullptr && output != nullptr, "decode_attention_absmax: null output / row_absmax");
  SGLM_CHECK_ARG(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0, "decode_attention_absmax: bad head counts");
  DecodeArgs probe{};
  probe.num_splits = 1;
  const bool aligned = q_stride_b % 8 == 0 && q_stride_h % 8 == 0 && k_stride_n % 8 == 0 && k_stride_h % 8 == 0 &&
                       v_stride_n % 8 == 0 && v_stride_h % 8 == 0 && reinterpret_cast<uintptr_t>(query) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(k_cache) % 16 == 0 && reinterpret_cast<uintptr_t>(v_cache) % 16 == 0;
  if (!(pair_eligible(probe, num_seqs * num_kv_heads) && (head_size == 128 || head_size == 64) &&
        num_heads / num_kv_heads <= 16 && aligned)) {
    set_error("decode_attention_absmax: shape outside the pairs-of-items kernel (needs > 256 (request, kv head) items, head "
              "size 64 / 128, group <= 16, aligned rows)");
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  tl_row_absmax = row_absmax;
  const int rc = sgl_mi355_decode_attention(query, k_cache, v_cache, output, nullptr, nullptr, nullptr, nullptr, req_to_token,
                                            req_to_token_is64, req_pool_indices, seq_lens, num_seqs, max_context_len,
                                            num_heads, num_kv_heads, head_size, head_size, 1, q_stride_b, q_stride_h,
                                            k_stride_n, k_stride_h, v_stride_n, v_stride_h, 0, 0, 0, 0, o_stride_b,
                                            o_stride_h, sm_scale, logit_cap, dtype, stream);
  tl_row_absmax = nullptr;
  return rc;
#endif
}

// sgl_mi355_decode_attention (page-table form, no KV write, one split, 16-bit pool) that ALSO leaves the per-token FP8
// quant of the output rows -- out_q e4m3 [num_seqs][num_heads * head_size], out_s float32 [num_seqs]: exactly what
// sgl_per_token_quant_fp8 (per_token_quant_fp8.cu:15-87) gives on `output` -- in the same launch: the workgroup that
// finishes a request's last head block quantises its row (decode_mfma_pair_kernel, QOUT).  merge_counters: int32
// [num_seqs], zero before the first call, left zero; one stream and one geometry at a time, like
// sgl_mi355_decode_attention_merged's.  `output` is written too (it is how the heads reach the quantising workgroup), with
// rows that 8-byte loads can read (o strides % 4 == 0).  Only the pairs-of-items kernel has the epilogue (num_seqs *
// num_kv_heads > 256, head size 64 / 128, group <= 16): any other shape returns SGL_MI355_ERR_UNSUPPORTED without launching.
extern "C" int sgl_mi355_decode_attention_quant(
    const void* query, void* k_cache, void* v_cache, void* output, void* out_q, float* out_s, int32_t* merge_counters,
    const void* req_to_token, int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens,
    int64_t num_seqs, int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
    int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap, int dtype, void* stream) {
#if !SGLM_OPTIN_FUSIONS
  set_error("%s: an opt-in fusion, not in this build of the library (build with -DSGLM_OPTIN_FUSIONS=1)", "decode_attention_quant");
  return SGL_MI355_ERR_UNSUPPORTED;
#else
  SGLM_CHECK_ARG(output != nullptr && out_q != nullptr && out_s != nullptr && merge_counters != nullptr,
                 "decode_attention_quant: null output / out_q / out_s / merge_counters");
  SGLM_CHECK_ARG(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0, "decode_attention_quant: bad head counts");
  DecodeArgs probe{};
  probe.num_splits = 1;
  const bool aligned = q_stride_b % 8 == 0 && q_stride_h % 8 == 0 && k_stride_n % 8 == 0 && k_stride_h % 8 == 0 &&
                       v_stride_n % 8 == 0 && v_stride_h % 8 == 0 && o_stride_b % 4 == 0 && o_stride_h % 4 == 0 &&
                       reinterpret_cast<uintptr_t>(query) % 16 == 0 && reinterpret_cast<uintptr_t>(k_cache) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(v_cache) % 16 == 0 && reinterpret_cast<uintptr_t>(output) % 8 == 0 &&
                       reinterpret_cast<uintptr_t>(out_q) % 8 == 0;
  if (!(pair_eligible(probe, num_seqs * num_kv_heads) && (head_size == 128 || head_size == 64) &&
        num_heads / num_kv_heads <= 16 && aligned && num_heads * head_size <= 16384)) {
    set_error("decode_attention_quant: shape outside the pairs-of-items kernel (needs > 256 (request, kv head) items, head "
              "size 64 / 128, group <= 16, aligned rows, at most 16384 elements per row)");
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  tl_pair_quant = MergeFused{merge_counters, static_cast<uint8_t*>(out_q), out_s};
  tl_pair_quant_used = false;
  const int rc = sgl_mi355_decode_attention(query, k_cache, v_cache, output, nullptr, nullptr, nullptr, nullptr, req_to_token,
                                            req_to_token_is64, req_pool_indices, seq_lens, num_seqs, max_context_len,
                                            num_heads, num_kv_heads, head_size, head_size, 1, q_stride_b, q_stride_h,
                                            k_stride_n, k_stride_h, v_stride_n, v_stride_h, 0, 0, 0, 0, o_stride_b,
                                            o_stride_h, sm_scale, logit_cap, dtype, stream);
  tl_pair_quant = MergeFused{nullptr, nullptr, nullptr};
  if (rc == 0 && !tl_pair_quant_used) {  // (cannot happen after the checks above; never report a quant that did not run)
    set_error("decode_attention_quant: the launch did not take the pairs-of-items kernel");
    return SGL_MI355_ERR_RUNTIME;
  }
  return rc;
#endif
}

// Decode attention straight off the qkv GEMM's split-K partial sums: one launch does the GEMM epilogue, RoPE on q/k,
// the k/v pool write at loc[b] and the attention over [0, seq_lens[b]) (which includes the new token).  Same result, bit
// for bit, as sgl_mi355_rotary_embedding_set_kv_from_partials followed by sgl_mi355_decode_attention.  Only the
// pairs-of-items form takes it (one split, more than 256 (request, kv head) items, 16-bit pool, head size 64/128 = rot_dim,
// neox pairs, group <= 16): anything else returns SGL_MI355_ERR_UNSUPPORTED WITHOUT launching, and the caller runs the two
// calls instead.
extern "C" int sgl_mi355_decode_attention_qkv_partials(
    const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b, const void* bias,
    const int64_t* positions, const float* cos_sin_cache, int64_t rot_dim, int is_neox, const void* loc, int loc_is64,
    void* k_cache, void* v_cache, void* output, const void* req_to_token, int req_to_token_is64,
    const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t num_seqs, int64_t max_context_len,
    int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream) {
#if !SGLM_OPTIN_FUSIONS
  set_error("%s: an opt-in fusion, not in this build of the library (build with -DSGLM_OPTIN_FUSIONS=1)", "decode_attention_qkv_partials");
  return SGL_MI355_ERR_UNSUPPORTED;
#else
  int rc = check_common(num_seqs, num_heads, num_kv_heads, head_size, head_size, 1, dtype);
  if (rc) return rc;
  if (num_seqs == 0) return 0;
  SGLM_CHECK_ARG(partials && scales_a && scales_b && positions && cos_sin_cache && loc && k_cache && v_cache && output &&
                     req_to_token && req_pool_indices && seq_lens && num_slices >= 1,
                 "decode_attention_qkv_partials: null tensor pointer");
  DecodeArgs probe{};
  probe.num_splits = 1;
  const bool ok = pair_eligible(probe, num_seqs * num_kv_heads) && (head_size == 128 || head_size == 64) &&
                  rot_dim == head_size && is_neox && num_heads / num_kv_heads <= 16 && k_stride_n % 8 == 0 &&
                  k_stride_h % 8 == 0 && v_stride_n % 8 == 0 && v_stride_h % 8 == 0 &&
                  reinterpret_cast<uintptr_t>(k_cache) % 16 == 0 && reinterpret_cast<uintptr_t>(v_cache) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(partials) % 16 == 0 && reinterpret_cast<uintptr_t>(scales_b) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(cos_sin_cache) % 16 == 0;
  if (!ok) {
    set_error("decode_attention_qkv_partials: shape outside the fused form (needs > 256 (request, kv head) items, head "
              "size 64/128 == rot_dim, neox, group <= 16, 16-byte aligned rows)");
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  const int64_t N = (num_heads + 2 * num_kv_heads) * head_size;
  const FusedQkv fq{PartialSrc{partials, (int)num_slices, num_seqs * N, scales_a, scales_b, bias, (int)N}, positions, loc,
                    loc_is64, cos_sin_cache};
  tl_fq = &fq;
  tl_fq_used = false;
  // the query pointer is not read in this form (q comes from the partials); strides as a [B, Hq, D] tensor would have
  rc = sgl_mi355_decode_attention(partials, k_cache, v_cache, output, nullptr, nullptr, nullptr, nullptr, req_to_token,
                                  req_to_token_is64, req_pool_indices, seq_lens, num_seqs, max_context_len, num_heads,
                                  num_kv_heads, head_size, head_size, 1, num_heads * head_size, head_size, k_stride_n,
                                  k_stride_h, v_stride_n, v_stride_h, 0, 0, 0, 0, o_stride_b, o_stride_h, sm_scale,
                                  logit_cap, dtype, stream);
  tl_fq = nullptr;
  if (rc == 0 && !tl_fq_used) {
    set_error("decode_attention_qkv_partials: internal error, the dispatcher did not take the fused kernel");
    return SGL_MI355_ERR_RUNTIME;
  }
  return rc;
#endif
}

// Paged decode (the request page table, one split) WITH the KV write of the step inside the same launch -- what
// AttentionBackend.forward(..., save_kv_cache=True) does in two steps (memory_pool.py:369-407 set_kv_buffer, then the decode
// kernel, triton_backend.py:686-732).  key / value [B, Hk, D] are the new tokens' finished rows (RoPE applied); loc[b] is the
// pool row they go to and MUST be the page-table entry of position seq_lens[b] - 1 (out_cache_loc of a decode batch: always).
// The launch writes both rows to the pool for the steps to come and takes the new token into the softmax from the tensors
// (one more partial state at the merge) instead of streaming it back from the pool.  Pairs-of-items kernel only: returns
// SGL_MI355_ERR_UNSUPPORTED -- nothing launched, nothing written -- for batches outside it (at most 256 (request, kv head
// block) items, FP8 pools, head sizes other than 64 / 128, unaligned rows); the caller then makes the two calls.
extern "C" int sgl_mi355_decode_attention_newkv(
    const void* query, void* k_cache, void* v_cache, void* output, const void* key, const void* value, const void* loc,
    int loc_is64, const void* req_to_token, int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens,
    int64_t num_seqs, int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t q_stride_b,
    int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int64_t key_stride_b,
    int64_t key_stride_h, int64_t value_stride_b, int64_t value_stride_h, int64_t o_stride_b, int64_t o_stride_h,
    float sm_scale, float logit_cap, int dtype, void* stream) {
#if !SGLM_OPTIN_FUSIONS
  set_error("%s: an opt-in fusion, not in this build of the library (build with -DSGLM_OPTIN_FUSIONS=1)", "decode_attention_newkv");
  return SGL_MI355_ERR_UNSUPPORTED;
#else
  int rc = check_common(num_seqs, num_heads, num_kv_heads, head_size, head_size, 1, dtype);
  if (rc) return rc;
  if (num_seqs == 0) return 0;
  SGLM_CHECK_ARG(query && k_cache && v_cache && output && key && value && loc && req_to_token && req_pool_indices && seq_lens,
                 "decode_attention_newkv: null tensor pointer");
  DecodeArgs probe{};
  probe.num_splits = 1;
  const int64_t group = num_heads / num_kv_heads, nhb = (group + 15) / 16;
  const bool ok = pair_eligible(probe, num_seqs * num_kv_heads * nhb) && (head_size == 128 || head_size == 64) &&
                  k_stride_n % 8 == 0 && k_stride_h % 8 == 0 && v_stride_n % 8 == 0 && v_stride_h % 8 == 0 &&
                  key_stride_b % 8 == 0 && key_stride_h % 8 == 0 && value_stride_b % 8 == 0 && value_stride_h % 8 == 0 &&
                  q_stride_b % 8 == 0 && q_stride_h % 8 == 0 && reinterpret_cast<uintptr_t>(k_cache) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(v_cache) % 16 == 0 && reinterpret_cast<uintptr_t>(key) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(value) % 16 == 0 && reinterpret_cast<uintptr_t>(query) % 16 == 0;
  if (!ok) {
    set_error("decode_attention_newkv: batch outside the fused form (needs > 256 (request, kv head block) items, head size "
              "64 / 128, 16-byte aligned rows)");
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  FusedQkv fq{};
  fq.loc = loc; fq.loc_is64 = loc_is64;
  fq.k_new = key; fq.v_new = value;
  fq.kn_sb = key_stride_b; fq.kn_sh = key_stride_h; fq.vn_sb = value_stride_b; fq.vn_sh = value_stride_h;
  tl_newkv = &fq;
  tl_newkv_used = false;
  rc = sgl_mi355_decode_attention(query, k_cache, v_cache, output, nullptr, nullptr, nullptr, nullptr, req_to_token,
                                  req_to_token_is64, req_pool_indices, seq_lens, num_seqs, max_context_len, num_heads,
                                  num_kv_heads, head_size, head_size, 1, q_stride_b, q_stride_h, k_stride_n, k_stride_h,
                                  v_stride_n, v_stride_h, 0, 0, 0, 0, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype, stream);
  tl_newkv = nullptr;
  if (rc == 0 && !tl_newkv_used) {
    // (the dispatcher took another kernel after all -- e.g. SGL_MI355_DECODE_PAIR=0: the attention ran WITHOUT the new token)
    set_error("decode_attention_newkv: internal error, the dispatcher did not take the pairs-of-items kernel");
    return SGL_MI355_ERR_RUNTIME;
  }
  return rc;
#endif
}

extern "C" int sgl_mi355_decode_attention_fwd_fp8kv(
    const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
    const int32_t* kv_indices, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
    int64_t max_kv_splits, int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream) {
  tl_kv8 = 1;
  const int rc = sgl_mi355_decode_attention_fwd(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse,
                                                num_kv_splits, max_kv_splits, batch, num_heads, num_kv_heads, head_size,
                                                head_size_v, q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n,
                                                v_stride_h, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype, stream);
  tl_kv8 = 0;
  return rc;
}

// float8_e5m2 pools (`--kv-cache-dtype fp8_e5m2`): same argument lists and kernels, the byte format aside
extern "C" int sgl_mi355_decode_attention_fp8kv_e5m2(
    const void* query, void* k_cache, void* v_cache, void* output, float* attn_logits, const void* req_to_token,
    int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t num_seqs,
    int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v,
    int64_t num_kv_splits, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream) {
  tl_kv8 = 2;
  const int rc = sgl_mi355_decode_attention(query, k_cache, v_cache, output, nullptr, nullptr, nullptr, attn_logits,
                                            req_to_token, req_to_token_is64, req_pool_indices, seq_lens, num_seqs,
                                            max_context_len, num_heads, num_kv_heads, head_size, head_size_v,
                                            num_kv_splits, q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n,
                                            v_stride_h, 0, 0, 0, 0, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype,
                                            stream);
  tl_kv8 = 0;
  return rc;
}

extern "C" int sgl_mi355_decode_attention_fwd_fp8kv_e5m2(
    const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
    const int32_t* kv_indices, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
    int64_t max_kv_splits, int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream) {
  tl_kv8 = 2;
  const int rc = sgl_mi355_decode_attention_fwd(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse,
                                                num_kv_splits, max_kv_splits, batch, num_heads, num_kv_heads, head_size,
                                                head_size_v, q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n,
                                                v_stride_h, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype, stream);
  tl_kv8 = 0;
  return rc;
}

extern "C" int sgl_mi355_has_optin_fusions(void) { return SGLM_OPTIN_FUSIONS; }

#if SGLM_DEC_TIMING
// timing build only: copy the stamps out (device-synchronising) and clear them
extern "C" int sgl_mi355_decode_timing_dump(void* host_buf, int64_t bytes) {
  const int64_t n = (int64_t)kDecTimingWgs * kDecTimingStamps * 8;
  SGLM_CHECK_ARG(host_buf && bytes >= n, "decode_timing_dump: buffer of %ld bytes needed", (long)n);
  unsigned long long* d = dec_timing_buffer();
  SGLM_CHECK_ARG(d != nullptr, "decode_timing_dump: no debug buffer");
  SGLM_CHECK_HIP(hipDeviceSynchronize());
  SGLM_CHECK_HIP(hipMemcpy(host_buf, d, n, hipMemcpyDeviceToHost));
  SGLM_CHECK_HIP(hipMemset(d, 0, n));
  return 0;
}
#endif

extern "C" int sgl_mi355_decode_merge_quant_fp8(const float* attn_logits, int64_t num_seqs, int64_t num_heads,
                                                int64_t head_size_v, int64_t num_kv_splits, void* output,
                                                int64_t o_stride_b, int64_t o_stride_h, void* out_q, float* out_s,
                                                int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "decode_merge_quant_fp8: dtype must be bf16 (0) or fp16 (1)");
  SGLM_CHECK_ARG(num_seqs >= 0 && num_heads > 0 && head_size_v > 0 && num_kv_splits >= 1 && num_kv_splits <= 65535,
                 "decode_merge_quant_fp8: bad shape");
  SGLM_CHECK_ARG((num_heads * head_size_v) % 8 == 0 && num_heads * head_size_v <= 16384 && num_heads * num_kv_splits <= 4096,
                 "decode_merge_quant_fp8: num_heads * head_size_v must be a multiple of 8 and <= 16384");
  if (num_seqs == 0) return 0;
  SGLM_CHECK_ARG(attn_logits && out_q && out_s, "decode_merge_quant_fp8: null tensor pointer");
  DecodeArgs a{};
  const int64_t l2 = head_size_v + 1, l1 = num_kv_splits * l2, l0 = num_heads * l1;
  a.mid_o = const_cast<float*>(attn_logits); a.mo_sb = l0; a.mo_sh = l1; a.mo_ss = l2;
  a.mid_lse = const_cast<float*>(attn_logits) + head_size_v; a.ml_sb = l0; a.ml_sh = l1; a.ml_ss = l2;
  a.out = output; a.o_sb = o_stride_b; a.o_sh = o_stride_h;
  a.num_heads = (int)num_heads; a.num_splits = (int)num_kv_splits;
  const int R = (int)(num_heads * head_size_v);
  const int lds = ((R * 2 + 15) & ~15) + (int)(num_heads * num_kv_splits + num_heads) * 4;
  hipStream_t s = as_stream(stream);
  // few requests: 1024 threads per row (the row is a chain of dependent memory round trips, not bandwidth)
#define MQ_GO(DT, NT_)                                                                                              \
  hipLaunchKernelGGL((decode_merge_quant_kernel<DT, NT_>), dim3((unsigned)num_seqs), dim3(NT_), lds, s, a,            \
                     (int)head_size_v, (uint8_t*)out_q, out_s)
  const bool wide = num_seqs <= 128 && R >= 2048;
  if (dtype == SGL_MI355_BF16) {
    if (wide) MQ_GO(SGL_MI355_BF16, 1024); else MQ_GO(SGL_MI355_BF16, 256);
  } else {
    if (wide) MQ_GO(SGL_MI355_FP16, 1024); else MQ_GO(SGL_MI355_FP16, 256);
  }
#undef MQ_GO
  return check_hip(hipGetLastError(), "decode_merge_quant launch");
}
