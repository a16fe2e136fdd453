// Ragged prefix + extend ("prefill") attention for MI355X / gfx950.
//
// Replaces (see include/sgl_mi355.h):
//   * extend_attention_fwd (Triton)  python/sglang/srt/layers/attention/triton_ops/extend_attention.py:306-438
//                                    (kernel :41-303), called from triton_backend.py:632-685
//   * extend_attention_cpu           sgl-kernel/csrc/cpu/extend.cpp:579-723 (impl :224-560)
//
// Math: for request b, query row r of its extend part (absolute position prefix_len + r), softmax over
//   stage 1: every cached prefix token, gathered from the KV pool through the page table;
//   stage 2: the new tokens k_extend/v_extend[start + j], j <= r (causal) -- contiguous memory;
// fp32 accumulation, p rounded to the 16-bit dtype before p@V (extend.cpp:409-410,
// extend_attention.py:196,276), out = acc / sum.
//
// Design (FlashAttention-2 shape, tiled for 64-wide waves and 32x32x16 MFMA):
//   * Workgroup = 4 waves = up to 4 q-heads of one GQA group x 32 query positions (or fewer heads and
//     more positions for small groups), so one K/V tile in LDS serves the whole group: K/V are read once
//     per group, not once per head.
//   * K/V tiles of 64 tokens arrive by LDS-DMA (per-lane source address: the same instruction gathers
//     page-table rows in stage 1 and streams contiguous rows in stage 2), double-buffered; the prefix
//     page-table slice is staged in LDS so the loop issues no ordinary global loads.
//   * Per wave S^T = K Q^T with v_mfma_f32_32x32x16 (A = K rows via ds_read_b128 from a swizzled image,
//     B = Q^T in registers).  The query row sits on the lane (col = lane&31), so the online softmax is
//     lane-local plus one cross-half shuffle.  O^T += V^T P^T with V^T fragments from
//     ds_read_b64_tr_b16 and P^T taken straight from the S^T accumulator registers (no LDS round trip).
//   * Head sizes other than 64/128 (e.g. the reference test's D=128, Dv=96) use a generic kernel.
#include <math.h>
#include <type_traits>

#include "common.h"

namespace sglm {
namespace {

constexpr int kBN = 64;          // keys per tile
constexpr int kIdxCap = 4096;    // prefix page-table entries staged in LDS per pass
constexpr float kLog2e = 1.4426950408889634f;

struct ExtendArgs {
  const void* q;  int64_t q_st, q_sh;      // q_extend [T,Hq,D]
  const void* ke; int64_t ke_st, ke_sh;    // k_extend [T,Hkv,D]
  const void* ve; int64_t ve_st, ve_sh;
  void* o;        int64_t o_st, o_sh;      // o_extend [T,Hq,Dv]
  const void* kb; int64_t kb_sn, kb_sh;    // pool
  const void* vb; int64_t vb_sn, vb_sh;
  // mode 0 (Triton form): qo_indptr/kv_indptr int32 [B+1], kv_indices int32
  const int32_t* qo_indptr;
  const int32_t* kv_indptr;
  // mode 1 (CPU-op form): req_to_token rows, int64 per-request vectors
  const int64_t* req_pool_indices;
  const int64_t* seq_lens;
  const int64_t* extend_seq_lens;
  const int64_t* extend_start_loc;
  int64_t r2t_stride;
  const void* indices;  // kv_indices (int32) or req_to_token (int32/int64)
  int num_heads, num_kv_heads, group;
  int num_mblocks;      // grid extent along query blocks
  float sm_scale, logit_cap;
  int causal, mode;
  // optional masks of the Triton kernel (extend_attention.py:171-189, 246-259); mask == nullptr && window <= 0: none
  const uint8_t* mask;         // per sequence [ext][prefix + ext] bytes, flattened back to back
  const int64_t* mask_indptr;  // [B+1]
  int skip_prefix_mask;        // SKIP_PREFIX_CUSTOM_MASK
  int window;                  // SLIDING_WINDOW_SIZE (prefix stage only: q <= n + window)
  int kv8;                     // 1 / 2: pool rows are e4m3fn / e5m2 bytes (strides in elements = bytes)
  // KV-range parts (round 4, PARTS kernels): an item = (query block, head group, request) whose prefix + extend tiles number
  // more than part_tiles is cut into up to smax consecutive tile ranges, one workgroup each; every part stores its
  // (O, m, l) and counts itself in on part_counters[item]; the one that completes the count merges all of them in part order.
  float* part_ws;              // [item][smax][owner wave][(Dv / 32) * 16 + 2][64] floats
  int32_t* part_counters;      // [items], zero before the first launch; left zero
  int smax, part_tiles;
};

__device__ __forceinline__ void seq_info(const ExtendArgs& a, int b, int64_t& idx_base, int& prefix, int& ext,
                                         int64_t& q_start) {
  if (a.mode == 0) {
    q_start = a.qo_indptr[b];
    ext = a.qo_indptr[b + 1] - (int)q_start;
    idx_base = a.kv_indptr[b];
    prefix = a.kv_indptr[b + 1] - (int)idx_base;
  } else {
    q_start = a.extend_start_loc[b];
    ext = (int)a.extend_seq_lens[b];
    prefix = (int)a.seq_lens[b] - ext;
    idx_base = a.req_pool_indices[b] * a.r2t_stride;
  }
}

// Timing build (-DSGLM_EXT_TIMING=1, variant library only): wave 0 of the first kExtTimingWgs workgroups stamps s_memtime at
// six points of every tile step into a debug buffer (ExtendArgs::part_ws of a launch without parts); each stamp sits behind a
// v_readfirstlane of a value the phase produces, so that the in-order issue of the wave places it after the phase's results.
// tools/exp/extend_phase_times.py reads the buffer through sgl_mi355_extend_timing_dump.
#ifndef SGLM_EXT_TIMING
#define SGLM_EXT_TIMING 0
#endif
#if SGLM_EXT_TIMING
constexpr int kExtTimingWgs = 64, kExtTimingTiles = 80, kExtTimingStamps = 6;
#define EXT_STAMP(slot, dep)                                                                                         \
  do {                                                                                                               \
    if (tstamp != nullptr && tstep < kExtTimingTiles) {                                                              \
      asm volatile("" ::"s"(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)(dep)))));                 \
      const unsigned long long tnow = __builtin_readcyclecounter();                                                  \
      if (lane == 0) tstamp[tstep * kExtTimingStamps + (slot)] = tnow;                                               \
    }                                                                                                                \
  } while (0)
#else
#define EXT_STAMP(slot, dep) do {} while (0)
#endif

template <int D>
__device__ __forceinline__ int swz_k(int c, int row) {  // ds_read_b128 row reads, 32 rows per operand
  return (D == 128) ? (c ^ (row & 15)) : (c ^ ((row >> 1) & 7));
}
template <int D>
__device__ __forceinline__ int swz_v(int c, int row) {  // ds_read_b64_tr_b16: keep 32-B units intact
  const int f = (D == 128) ? ((row & 3) << 1) : (((row >> 1) & 1) << 1);
  return (((c >> 1) ^ f) << 1) | (c & 1);
}

// 8 e4m3 bytes -> 8 values of the 16-bit dtype (exact)
// (e5: the bytes are e5m2 -- a workgroup-uniform flag, not a template parameter: this path is not the hot one)
template <int DTYPE>
__device__ __forceinline__ typename Half16<DTYPE>::x8 cvt8_fp8(const uint2& raw, bool e5) {
  using Hh = Half16<DTYPE>;
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typename Hh::x8 out;
  const uint32_t w[2] = {raw.x, raw.y};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x2_t lo = e5 ? __builtin_amdgcn_cvt_pk_f32_bf8((int)w[i], false) : __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], false);
    const f32x2_t hi = e5 ? __builtin_amdgcn_cvt_pk_f32_bf8((int)w[i], true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], true);
    out[4 * i + 0] = Hh::from_f32(lo[0]);
    out[4 * i + 1] = Hh::from_f32(lo[1]);
    out[4 * i + 2] = Hh::from_f32(hi[0]);
    out[4 * i + 3] = Hh::from_f32(hi[1]);
  }
  return out;
}
// fp32 rounded to e4m3 (saturating) and back: `x.to(fp8)` of the Triton kernel for Q (extend_attention.py:149) and P (:200)
__device__ __forceinline__ float round_fp8(float x, bool e5) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  if (e5) {
    const int pk = __builtin_amdgcn_cvt_pk_bf8_f32(x, 0.f, 0, false);
    const f32x2_t r = __builtin_amdgcn_cvt_pk_f32_bf8(pk, false);
    return r[0];
  }
  const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x, 0.f, 0, false);
  const f32x2_t r = __builtin_amdgcn_cvt_pk_f32_fp8(pk, false);
  return r[0];
}

// GH = q heads per workgroup (1, 2 or 4); the other 4/GH waves take further 32-row position blocks.
// KV8: the pool holds e4m3 bytes.  The prefix stage DMAs byte tiles into a two-deep staging area (the LDS of stage 1),
// converts each (exactly) into the 16-bit tile image of stage 0 and runs the usual MFMAs on it with Q and P rounded to
// FP8 first, which is what the Triton kernel computes (fp8 x fp8 products are exact in the 16-bit MFMA as well); the
// extend stage is the 16-bit code unchanged.
// KSPLIT = 2: the 64 keys of every tile are shared by TWO waves per (head, position block) -- each takes one 32-key half
// through QK^T, softmax and P.V and the two partial (m, l, O) are merged through LDS at the end.  Per-wave work per tile
// halves, so the longest query block (the critical path of a short single-request prefill, where there is one
// workgroup per CU or less) finishes in half the time; the workgroup covers 2 (GH = 2) or 1 head(s).
// NSTAGE (round 3): K/V tiles in LDS; two everywhere (more tiles in flight at one workgroup per CU were measured slower for
// the short single-request prefill, see dispatch()).
// PARTS (round 4): the tiles of an item are cut into consecutive ranges over several workgroups (ExtendArgs::part_ws) -- for
// the launches that leave most of the chip idle behind a long serial chain: one short request behind a long cached prefix
// (32-64 workgroups of 20-70 tiles each) and the heaviest query blocks of a single 1024-token prefill.
template <int DTYPE, int D, typename IdxT, int GH, bool MASKED, bool KV8 = false, int KSPLIT = 1, int NSTAGE = 2,
          bool PARTS = false>
__global__ __launch_bounds__(256, 2) void extend_mfma_kernel(ExtendArgs a) {
  static_assert(NSTAGE == 2, "two LDS stages: the tile loop is unrolled over them (run_phase), the FP8 staging area is laid out for them");
  static_assert(!PARTS || (!KV8 && !MASKED), "KV-range parts: 16-bit pools, no custom mask / window");
  using H = Half16<DTYPE>;
  using T = typename H::T;
  using x8 = typename H::x8;
  using x4 = typename H::x4;
  const bool e5 = a.kv8 == 2;  // the pool bytes are e5m2 (KV8 kernels only)
  constexpr int ROWB = D * 2;
  constexpr int CH = ROWB / 16;
  constexpr int ROWS_PER_DMA = 1024 / ROWB;
  constexpr int PIECES = kBN / ROWS_PER_DMA;    // DMA pieces per K (or V) tile
  constexpr int PPW = PIECES / 4;               // pieces per wave
  constexpr int TILE_BYTES = kBN * ROWB;
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  constexpr int KS = D / 16;                    // 32x32x16 k-steps over the head dim
  constexpr int NDVB = D / 32;                  // 32-wide output blocks
  constexpr int NPB = 4 / (GH * KSPLIT);        // position blocks per workgroup
  constexpr int NTH = 2 / KSPLIT;               // 32-key halves of a tile this wave works on
  static_assert(GH * KSPLIT <= 4 && (KSPLIT == 1 || KSPLIT == 2), "4 waves = heads x position blocks x key halves");
  constexpr int BP = 32 * NPB;                  // query positions per workgroup

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* idx_lds = reinterpret_cast<int32_t*>(smem + NSTAGE * STAGE_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, hh = lane >> 5;

  // blockIdx.x = mblk' * (B * Hq/GH) + (b * Hq/GH + hgrp): ALL the heavy (late, long-prefix) query blocks of every
  // request and head group are dispatched before any lighter one (longest-processing-time-first over the whole grid)
  const int hgroups = a.num_heads / GH;
  int bid = blockIdx.x;
  int part = 0;
  if constexpr (PARTS) {  // the parts of an item are neighbours in the launch order: dispatched, and done, at about the same time
    part = bid % a.smax;
    bid /= a.smax;
  }
  const int item = bid;
  const int nbh = (int)(gridDim.x / (PARTS ? a.smax : 1) / a.num_mblocks);  // B * hgroups
  const int mblk = a.num_mblocks - 1 - (bid / nbh);
  bid %= nbh;
  const int hgrp = bid % hgroups;
  const int b = bid / hgroups;
  const int head0 = hgrp * GH;
  const int kvh = head0 / a.group;

  int64_t idx_base, q_start;
  int prefix, ext;
  seq_info(a, b, idx_base, prefix, ext, q_start);
  const int p0 = mblk * BP;
  if (p0 >= ext) return;

  const int head = head0 + (wave % GH);
  const int pw0 = p0 + 32 * ((wave / GH) % NPB);  // first query position of this wave
  const int kh = wave / (GH * NPB);                // KSPLIT == 2: the key half of every tile this wave takes
  const int qpos = pw0 + col;             // this lane's query position within the extend part
  const bool q_ok = qpos < ext;

  // ---- Q^T fragments: lane (col, hh) holds Q[qpos][head][16s + 8hh .. +8]
  x8 qf[KS];
  auto load_q = [&](bool to_fp8) __attribute__((always_inline)) {
    const T* qp = reinterpret_cast<const T*>(a.q) + (q_start + (q_ok ? qpos : 0)) * a.q_st + (int64_t)head * a.q_sh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (q_ok) {
        qf[s] = *reinterpret_cast<const x8*>(qp + 16 * s + 8 * hh);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[s][j] = (T)0.f;
      }
    }
    if (to_fp8) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[s][j] = H::from_f32(round_fp8(H::to_f32(qf[s][j]), e5));
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[s]));  // keep hipcc's wait for these loads out of the loop
  };
  load_q(KV8 && prefix > 0);

  // (with a custom mask AND is_causal the reference still stops at the end of the query block,
  //  extend_attention.py:199-203; masks that are subsets of the causal mask -- tree attention -- do not notice)
  const int n_ext_keys = a.causal ? ((p0 + BP) < ext ? (p0 + BP) : ext) : ext;
  const uint8_t* mrow = nullptr;  // this lane's row of the custom mask
  if constexpr (MASKED) {
    if (a.mask != nullptr && q_ok) mrow = a.mask + a.mask_indptr[b] + (int64_t)qpos * (prefix + ext);
  }
  const int nt1 = (prefix + kBN - 1) / kBN;
  const int nt2 = (n_ext_keys + kBN - 1) / kBN;
  // this workgroup's tiles: [pt0, pt1) of the item's nt1 + nt2 (prefix tiles first)
  int pt0 = 0, pt1 = nt1 + nt2, nparts = 1;
  if constexpr (PARTS) {
    const int ntot = nt1 + nt2;
    nparts = (ntot + a.part_tiles - 1) / a.part_tiles;
    nparts = nparts < 1 ? 1 : (nparts > a.smax ? a.smax : nparts);
    if (part >= nparts) return;
    const int per = (ntot + nparts - 1) / nparts;
    pt0 = part * per;
    pt1 = pt0 + per < ntot ? pt0 + per : ntot;
    if (pt0 >= pt1) { pt0 = pt1 = 0; }  // (an empty trailing part still counts itself in below)
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o_acc[NDVB];
#pragma unroll
  for (int i = 0; i < NDVB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
#if SGLM_EXT_TIMING
  unsigned long long* tstamp = nullptr;
  int tstep = 0;
  if constexpr (!PARTS) {
    if (a.part_ws != nullptr && blockIdx.x < kExtTimingWgs && wave == 0)
      tstamp = reinterpret_cast<unsigned long long*>(a.part_ws) + (int64_t)blockIdx.x * kExtTimingTiles * kExtTimingStamps;
  }
#endif

  const float scale_log2 = a.sm_scale * kLog2e;
  const bool has_cap = MASKED && a.logit_cap > 0.f;  // (the dispatcher sends a logit cap to the MASKED kernels)
  const int dma_row = lane / CH, dma_pos = lane % CH;
  const char* kpool = reinterpret_cast<const char*>(a.kb) + (int64_t)kvh * a.kb_sh * 2;
  const char* vpool = reinterpret_cast<const char*>(a.vb) + (int64_t)kvh * a.vb_sh * 2;
  const char* kext = reinterpret_cast<const char*>(a.ke) + (q_start * a.ke_st + (int64_t)kvh * a.ke_sh) * 2;
  const char* vext = reinterpret_cast<const char*>(a.ve) + (q_start * a.ve_st + (int64_t)kvh * a.ve_sh) * 2;

  // DMA addressing (round 4).  Of the ~305 vector instructions this kernel issued per 32-MFMA tile, 60 were 64-bit
  // address arithmetic for its eight DMA pieces (SQ counters, profiles/r04_extend_pipelined_negative.txt).  Now:
  //   * new tokens (contiguous rows): a wave-uniform 64-bit base per piece (scalar ALU) + a per-lane 32-bit offset =
  //     row-in-piece x row stride + the swizzled 16-byte chunk, one multiply-add per piece (lds_dma16_s);
  //   * prefix (gathered rows): pool base + slot x row stride as ONE 32 x 32 -> 64-bit multiply-add per piece.
  // The row strides in bytes fit 32 bits (the dispatcher checks); pieces that reach past the last key of a ragged tile
  // keep the clamped per-lane form.
  const uint32_t ke_rowb = (uint32_t)(a.ke_st * 2), ve_rowb = (uint32_t)(a.ve_st * 2);
  const uint32_t kb_rowb = (uint32_t)(a.kb_sn * 2), vb_rowb = (uint32_t)(a.vb_sn * 2);
  // byte offset of this lane's swizzled chunk in K piece i of the wave (kept in registers, except in the FP8-pool kernels,
  // which have none to spare and recompute it); V's is the same for every piece: swz_v looks at row & 3 (D = 128) /
  // (row >> 1) & 1 (D = 64), i.e. at the row in the piece
  uint32_t ksw_reg[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) ksw_reg[i] = swz_k<D>(dma_pos, (wave * PPW + i) * ROWS_PER_DMA + dma_row) * 16;
  auto ksw_of = [&](int i) __attribute__((always_inline)) -> uint32_t { return ksw_reg[i]; };
  const uint32_t vsw = swz_v<D>(dma_pos, dma_row) * 16;
  // issue this wave's share (PPW K pieces + PPW V pieces) of tile `t` of the given phase into `stage`
  auto issue = [&](int phase, int t, int stage, int idx_off, int n_keys) __attribute__((always_inline)) {
    const uint32_t kdst = __builtin_amdgcn_readfirstlane(lds_addr_of(smem + stage * STAGE_BYTES)) + wave * (PPW * 1024);
    const uint32_t vdst = kdst + TILE_BYTES;
    const int row0 = t * kBN + wave * (PPW * ROWS_PER_DMA);  // first key of this wave's first piece
    if constexpr (KV8) {
      // (the FP8-pool kernels have no registers to spare for the forms below: the first version's arithmetic, extend stage only)
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int row = (wave * PPW + i) * ROWS_PER_DMA + dma_row;
        int kidx = t * kBN + row;
        kidx = kidx < n_keys ? kidx : n_keys - 1;
        lds_dma16(kext + (int64_t)kidx * a.ke_st * 2 + swz_k<D>(dma_pos, row) * 16, kdst + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int row = (wave * PPW + i) * ROWS_PER_DMA + dma_row;
        int kidx = t * kBN + row;
        kidx = kidx < n_keys ? kidx : n_keys - 1;
        lds_dma16(vext + (int64_t)kidx * a.ve_st * 2 + swz_v<D>(dma_pos, row) * 16, vdst + i * 1024);
      }
    } else if (phase == 0) {
      const char* ka[PPW];
      const char* va[PPW];
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        int kidx = row0 + i * ROWS_PER_DMA + dma_row;
        kidx = kidx < n_keys ? kidx : n_keys - 1;  // tail rows re-read a valid key; masked in the softmax
        const uint32_t slot = (uint32_t)idx_lds[kidx - idx_off];
        ka[i] = kpool + ((uint64_t)slot * kb_rowb + ksw_of(i));
        va[i] = vpool + ((uint64_t)slot * vb_rowb + vsw);
      }
#pragma unroll
      for (int i = 0; i < PPW; ++i) lds_dma16(ka[i], kdst + i * 1024);
#pragma unroll
      for (int i = 0; i < PPW; ++i) lds_dma16(va[i], vdst + i * 1024);
    } else if (row0 + PPW * ROWS_PER_DMA <= n_keys) {  // (wave-uniform) every row of the wave's pieces is a real key
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        lds_dma16_s(kext + (uint64_t)(uint32_t)(row0 + i * ROWS_PER_DMA) * ke_rowb, dma_row * ke_rowb + ksw_of(i), kdst + i * 1024);
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        lds_dma16_s(vext + (uint64_t)(uint32_t)(row0 + i * ROWS_PER_DMA) * ve_rowb, dma_row * ve_rowb + vsw, vdst + i * 1024);
    } else {  // the ragged last tile
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        int kidx = row0 + i * ROWS_PER_DMA + dma_row;
        kidx = kidx < n_keys ? kidx : n_keys - 1;
        lds_dma16(kext + ((uint64_t)(uint32_t)kidx * ke_rowb + ksw_of(i)), kdst + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        int kidx = row0 + i * ROWS_PER_DMA + dma_row;
        kidx = kidx < n_keys ? kidx : n_keys - 1;
        lds_dma16(vext + ((uint64_t)(uint32_t)kidx * ve_rowb + vsw), vdst + i * 1024);
      }
    }
  };

  // One piece of this wave's share of extend-stage tile `t` (j < PPW: K piece j, else V piece j - PPW) into `stage`: the
  // contiguous, all-rows-valid form only (scalar base + lane offset, no vector arithmetic).  Round 5: these are issued
  // BETWEEN the MFMAs of the step that precedes the tile's (dma_between below) instead of in a block at the loop top, where
  // stamps put them at ~160 cycles apiece = 1 300 cycles of every wave's step with nothing of its own in the matrix pipe
  // (profiles/r05_extend_phase_times.txt; MI355X_MICROARCH.md: an LDS-DMA piece costs ~60 cycles of issue among bare MFMAs).
  auto issue_piece = [&](int t, int stage, int j) __attribute__((always_inline)) {
    const uint32_t kdst = __builtin_amdgcn_readfirstlane(lds_addr_of(smem + stage * STAGE_BYTES)) + wave * (PPW * 1024);
    const int row0 = t * kBN + wave * (PPW * ROWS_PER_DMA);
    if (j < PPW)
      lds_dma16_s(kext + (uint64_t)(uint32_t)(row0 + j * ROWS_PER_DMA) * ke_rowb, dma_row * ke_rowb + ksw_of(j), kdst + j * 1024);
    else
      lds_dma16_s(vext + (uint64_t)(uint32_t)(row0 + (j - PPW) * ROWS_PER_DMA) * ve_rowb, dma_row * ve_rowb + vsw,
                  kdst + TILE_BYTES + (j - PPW) * 1024);
  };
  // the 2 PPW pieces of the next tile spread over this step's MFMAs: one after every second QK^T MFMA, what is left after
  // every second PV MFMA (GH = 4 form: 16 + 16 MFMAs, 8 pieces, all beside QK^T; key-split form: 8 + 8 MFMAs, 4 + 4)
  constexpr int NQK = NTH * KS, NPV = NTH * 2 * NDVB;
  constexpr int NP_QK = (2 * PPW) < (NQK / 2) ? (2 * PPW) : (NQK / 2);
  constexpr int NP_PV = 2 * PPW - NP_QK;
  static_assert(KV8 || NP_PV <= NPV / 2, "the pieces of a tile must fit between the MFMAs of a step");

  // `stage` arrives as std::integral_constant: with the stage a compile-time constant every LDS read of the tile is a
  // lane-constant address register + an immediate offset (round 5: the v_add_u32 that re-based the address registers on the
  // runtime stage every tile are gone; the tile loop is unrolled over its two stages instead).
  // `inter` (wave-uniform, run time): issue tile t + 1's pieces (into the other stage) between this step's MFMAs -- a scalar
  // branch around each piece, ONE body (two instantiations joined behind an `if` cost 32 v_mov_b64 per tile to reconcile
  // the O accumulators' registers)
  auto compute = [&](int phase, int t, auto stage_c, int n_keys, const bool inter) __attribute__((always_inline)) {
    constexpr int stage = decltype(stage_c)::value;
    const char* kst = smem + stage * STAGE_BYTES;
    const char* vst = kst + TILE_BYTES;
    // ---- S^T = K Q^T for the two 32-key halves
    f32x16 s_acc[NTH];
#pragma unroll
    for (int ti = 0; ti < NTH; ++ti) {
      const int th = KSPLIT == 2 ? kh : ti;
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[ti][r] = 0.f;
      const int row = 32 * th + col;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const x8 kf = *reinterpret_cast<const x8*>(kst + row * ROWB + swz_k<D>(2 * s + hh, row) * 16);
        s_acc[ti] = H::mfma32(kf, qf[s], s_acc[ti]);
        if constexpr (!KV8) {
          constexpr int every = NQK / NP_QK;
          const int m = ti * KS + s;  // (compile-time after unrolling)
          if (m % every == every - 1 && inter) issue_piece(t + 1, stage ^ 1, m / every);
        }
      }
    }
    // ---- online softmax; key of (th, r) = t*64 + 32th + (r&3) + 8(r>>2) + 4hh
    float m_tile = -INFINITY;
    // tiles that lie wholly inside the valid keys and (extend stage, causal) wholly at or below this wave's first query
    // position need no per-element mask: most tiles of a long block (wave-uniform test, the same values either way)
    const bool unmasked_tile = !MASKED && !has_cap && (t + 1) * kBN <= n_keys &&
                               !(phase == 1 && a.causal && (t + 1) * kBN - 1 > pw0);
    if (unmasked_tile) {
#pragma unroll
      for (int ti = 0; ti < NTH; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float s = s_acc[ti][r] * scale_log2;
          s_acc[ti][r] = s;
          m_tile = fmaxf(m_tile, s);
        }
    } else
#pragma unroll
    for (int ti = 0; ti < NTH; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int th = KSPLIT == 2 ? kh : ti;
        float s = s_acc[ti][r];
        if (has_cap) {
          s = s * a.sm_scale;
          s = a.logit_cap * tanhf(s / a.logit_cap) * kLog2e;
        } else {
          s = s * scale_log2;
        }
        const int key = t * kBN + 32 * th + (r & 3) + 8 * (r >> 2) + 4 * hh;
        bool ok = key < n_keys;
        if constexpr (MASKED) {
          if (phase == 0) {
            if (a.window > 0) ok = ok && (qpos <= key + a.window);
            if (a.mask != nullptr && !a.skip_prefix_mask && ok && q_ok) ok = mrow[key] != 0;
          } else {
            if (a.mask != nullptr) {
              if (ok && q_ok) ok = mrow[prefix + key] != 0;
            } else if (a.causal) {
              ok = ok && (key <= qpos);
            }
          }
        } else {
          if (phase == 1 && a.causal) ok = ok && (key <= qpos);
        }
        s = ok ? s : -INFINITY;
        s_acc[ti][r] = s;
        m_tile = fmaxf(m_tile, s);
      }
    {  // the other 32-lane half holds the other keys of the same query row: one v_permlane32_swap (round 5) instead of a
       // ds_bpermute round trip through the LDS pipe + 5 address instructions in the middle of the softmax chain.
       // Inline asm on purpose: this hipcc folds __builtin_amdgcn_permlane32_swap(x, x) to {x, x} (as if swapping a value with
       // itself did nothing -- it exchanges the two HALVES), which silently leaves every row with the maximum of half 0 only.
       // a = {m.lo, m.lo}, b = {m.hi, m.hi} afterwards; s_nop 1: the VALU-write -> permlane-swap hazard of the ISA.
      float ma = m_tile, mb = m_tile;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
      m_tile = fmaxf(ma, mb);
    }
    EXT_STAMP(2, m_tile);  // QK^T MFMAs + scale + row maximum done
    const float m_new = fmaxf(m_run, m_tile);
    // a row may have seen no visible key yet (its first tile fully masked): keep everything at zero
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    float psum = 0.f;
    x8 pf[NTH][2];
#pragma unroll
    for (int ti = 0; ti < NTH; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s_acc[ti][r] - m_safe);
        psum += p;
        pf[ti][r >> 3][r & 7] = H::from_f32((KV8 && phase == 0) ? round_fp8(p, e5) : p);
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    EXT_STAMP(3, psum);  // exp2 / row sum done (P packed)
    // the running maximum moves in the first tiles of a row and then rarely: when NO lane of the wave has a new maximum
    // alpha is exactly 1 everywhere and the 16 NDVB multiplies (a quarter of the tile's vector instructions) are skipped --
    // bit-identical, since x * 1.0f == x
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int i = 0; i < NDVB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
    }
    // ---- O^T += V^T P^T
    const int grp = lane >> 4;       // 16-lane group: column half (grp&1), k half hh = grp>>1
    const int q4 = (lane >> 2) & 3;  // row within a 4-row transposed block
    const int p4 = lane & 3;         // 8-B piece of the 32-B column block
#pragma unroll
    for (int ti = 0; ti < NTH; ++ti)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int th = KSPLIT == 2 ? kh : ti;
        const int r_lo = 32 * th + 16 * s2 + 4 * hh + q4;
        const int r_hi = r_lo + 8;
#pragma unroll
        for (int dvb = 0; dvb < NDVB; ++dvb) {
          const int c = 4 * dvb + 2 * (grp & 1) + (p4 >> 1);
          const x4 v_lo = H::ds_read_tr(vst + r_lo * ROWB + swz_v<D>(c, r_lo) * 16 + 8 * (p4 & 1));
          const x4 v_hi = H::ds_read_tr(vst + r_hi * ROWB + swz_v<D>(c, r_hi) * 16 + 8 * (p4 & 1));
          x8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vf[j] = v_lo[j];
            vf[4 + j] = v_hi[j];
          }
          o_acc[dvb] = H::mfma32(vf, pf[ti][s2], o_acc[dvb]);
          if constexpr (!KV8 && NP_PV > 0) {
            constexpr int every = NP_PV > 0 ? NPV / NP_PV : 1;
            const int m = (ti * 2 + s2) * NDVB + dvb;
            if (m % every == every - 1 && inter) issue_piece(t + 1, stage ^ 1, NP_QK + m / every);
          }
        }
      }
  };

  // tile sequence: stage 1 (prefix, in passes of kIdxCap page-table entries), then stage 2 (new tokens)
  // Every pass runs the same double-buffered loop.
  auto run_phase = [&](int phase, int t_begin, int t_end, int idx_off, int n_keys) __attribute__((always_inline)) {
    static_assert(NSTAGE == 2, "the tile loop is unrolled over two stages");
    if (t_begin >= t_end) return;
    issue(phase, t_begin, 0, idx_off, n_keys);  // prologue: one tile in flight
    // one step = tile t out of stage S while tile t + 1 lands in the other stage
    auto step = [&](int t, auto stage_c) __attribute__((always_inline)) {
      constexpr int S = decltype(stage_c)::value;
      // extend stage, 16-bit kernels: when every row of this wave's share of tile t + 1 is a real key, its pieces go out
      // between this step's MFMAs (issue_piece); the ragged last tile and the prefix stage keep the block at the top
      // (measured, same box, us, block at the top -> between the MFMAs: L = 4096 173.3 -> 168.5, 8192 554.9 -> 549.6, 4 x 2048
      //  160.4 -> 160.7 on the GH = 4 form; the key-split form of short prefills LOSES 3-5 % -- 8 MFMAs per phase leave no room
      //  beside them: L = 1024 24.7 -> 25.5 -- and keeps the block)
      bool inter = false;
      if constexpr (!KV8 && KSPLIT == 1) {
        inter = phase == 1 && t + 1 < t_end && (t + 1) * kBN + (wave + 1) * (PPW * ROWS_PER_DMA) <= n_keys;
      }
      if (t + 1 < t_end && !inter) {
        issue(phase, t + 1, S ^ 1, idx_off, n_keys);  // the stage freed by the previous step's trailing barrier
        wait_vmcnt<2 * PPW>();                        // tile t's pieces are older than these: landed
      } else {
        wait_vmcnt<0>();                              // nothing younger than tile t's pieces is in flight
      }
      EXT_STAMP(0, 0.f);             // own share of tile t landed
      __builtin_amdgcn_s_barrier();  // every wave's share of tile t has landed
      EXT_STAMP(1, 0.f);
      compute(phase, t, stage_c, n_keys, inter);
      wait_lgkmcnt0();
      EXT_STAMP(4, o_acc[NDVB - 1][15]);  // PV MFMAs done
      __builtin_amdgcn_s_barrier();  // everyone is done reading stage S before it is refilled
      EXT_STAMP(5, 0.f);
#if SGLM_EXT_TIMING
      ++tstep;
#endif
    };
    for (int t = t_begin; t < t_end; t += 2) {
      step(t, std::integral_constant<int, 0>{});
      if (t + 1 < t_end) step(t + 1, std::integral_constant<int, 1>{});
    }
  };

  // ---- KV8 prefix stage: byte tiles -> staging (stage 1's LDS, two buffers of K+V) -> 16-bit image in stage 0
  constexpr int TILE8 = kBN * D;                 // bytes of one e4m3 K (or V) tile
  constexpr int CH8 = D / 16;                    // 16-B chunks per e4m3 row
  constexpr int ROWS8 = 1024 / D;                // rows per 1-KB DMA piece
  constexpr int PPW8 = (kBN / ROWS8) / 4;        // pieces per wave
  static_assert(2 * 2 * TILE8 <= STAGE_BYTES, "staging must fit the second stage");
  auto issue8 = [&](int t, int sb, int idx_off, int n_keys) __attribute__((always_inline)) {
    const uint32_t kdst = __builtin_amdgcn_readfirstlane(lds_addr_of(smem + STAGE_BYTES + sb * 2 * TILE8));
    const uint32_t vdst = kdst + TILE8;
    const char* kp8 = reinterpret_cast<const char*>(a.kb) + (int64_t)kvh * a.kb_sh;
    const char* vp8 = reinterpret_cast<const char*>(a.vb) + (int64_t)kvh * a.vb_sh;
    const int r8 = lane / CH8, c8 = lane % CH8;
    int64_t koff[PPW8], voff[PPW8];
#pragma unroll
    for (int i = 0; i < PPW8; ++i) {
      int kidx = t * kBN + (wave * PPW8 + i) * ROWS8 + r8;
      kidx = kidx < n_keys ? kidx : n_keys - 1;
      const int64_t slot = idx_lds[kidx - idx_off];
      koff[i] = slot * a.kb_sn + c8 * 16;
      voff[i] = slot * a.vb_sn + c8 * 16;
    }
#pragma unroll
    for (int i = 0; i < PPW8; ++i) lds_dma16(kp8 + koff[i], kdst + (wave * PPW8 + i) * 1024);
#pragma unroll
    for (int i = 0; i < PPW8; ++i) lds_dma16(vp8 + voff[i], vdst + (wave * PPW8 + i) * 1024);
  };
  auto convert8 = [&](int sb) __attribute__((always_inline)) {
    const char* ks8 = smem + STAGE_BYTES + sb * 2 * TILE8;
    const char* vs8 = ks8 + TILE8;
    char* kst = smem;
    char* vst = smem + TILE_BYTES;
    constexpr int CPR = D / 8;                   // 8-element chunks per row
    constexpr int NCH = kBN * CPR / 256;         // chunks per thread
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + 256 * i, row = c / CPR, ch = c % CPR;
      const uint2 kr = *reinterpret_cast<const uint2*>(ks8 + row * D + ch * 8);
      const uint2 vr = *reinterpret_cast<const uint2*>(vs8 + row * D + ch * 8);
      *reinterpret_cast<x8*>(kst + row * ROWB + swz_k<D>(ch, row) * 16) = cvt8_fp8<DTYPE>(kr, e5);
      *reinterpret_cast<x8*>(vst + row * ROWB + swz_v<D>(ch, row) * 16) = cvt8_fp8<DTYPE>(vr, e5);
    }
  };
  auto run_phase8 = [&](int t_begin, int t_end, int idx_off, int n_keys) __attribute__((always_inline)) {
    if (t_begin >= t_end) return;
    issue8(t_begin, 0, idx_off, n_keys);
    for (int t = t_begin; t < t_end; ++t) {
      const int sb = (t - t_begin) & 1;
      if (t + 1 < t_end) {
        issue8(t + 1, sb ^ 1, idx_off, n_keys);
        wait_vmcnt<2 * PPW8>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();  // byte tile t has landed; every wave is done with the 16-bit image of tile t-1
      convert8(sb);
      wait_lgkmcnt0();
      __builtin_amdgcn_s_barrier();  // the 16-bit image of tile t is complete; staging sb may be refilled
      compute(0, t, std::integral_constant<int, 0>{}, n_keys, false);
    }
    wait_lgkmcnt0();
  };

  for (int i0 = 0; i0 < prefix; i0 += kIdxCap) {
    const int n = (prefix - i0) < kIdxCap ? (prefix - i0) : kIdxCap;
    if constexpr (PARTS) {  // passes outside this part's range are skipped (workgroup-uniform)
      if ((i0 + n + kBN - 1) / kBN <= pt0 || i0 / kBN >= pt1) continue;
    }
    {
      const IdxT* src = reinterpret_cast<const IdxT*>(a.indices) + idx_base + i0;
      for (int i = tid; i < n; i += 256) idx_lds[i] = (int32_t)src[i];
    }
    __syncthreads();
    if constexpr (KV8) {
      run_phase8(i0 / kBN, (i0 + n + kBN - 1) / kBN, i0, i0 + n);
    } else {
      if constexpr (PARTS) {
        const int tb = i0 / kBN > pt0 ? i0 / kBN : pt0;
        const int te = (i0 + n + kBN - 1) / kBN < pt1 ? (i0 + n + kBN - 1) / kBN : pt1;
        run_phase(0, tb, te, i0, i0 + n);
      } else {
        run_phase(0, i0 / kBN, (i0 + n + kBN - 1) / kBN, i0, i0 + n);  // kIdxCap is a multiple of kBN
      }
    }
    __syncthreads();
  }
  if constexpr (KV8) {
    if (prefix > 0) load_q(false);
  }
  if constexpr (PARTS) {
    const int tb = pt0 > nt1 ? pt0 - nt1 : 0;
    const int te = pt1 > nt1 ? pt1 - nt1 : 0;
    run_phase(1, tb, te, 0, n_ext_keys);
  } else {
    run_phase(1, 0, nt2, 0, n_ext_keys);
  }

  // ---- epilogue: out = acc / l ; lane (col = qrow, hh), reg r -> dv = 32*dvb + (r&3) + 8(r>>2) + 4hh
  l_run += __shfl_xor(l_run, 32);
  bool owner = true;  // this wave holds a (head, position block)'s result
  if constexpr (KSPLIT == 2) {
    // merge the two key halves: wave kh = 1 hands (O, m, l) to its partner through LDS, element i of lane l at [i][l]
    __syncthreads();  // every wave is past its last tile: the stages are dead
    float* mo = reinterpret_cast<float*>(smem) + (wave % (GH * NPB)) * ((NDVB * 16 + 2) * 64);
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < NDVB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) mo[(16 * i + r) * 64 + lane] = o_acc[i][r];
      mo[(NDVB * 16) * 64 + lane] = m_run;
      mo[(NDVB * 16 + 1) * 64 + lane] = l_run;
    }
    __syncthreads();
    if constexpr (!PARTS) {
      if (kh == 1) return;
    }
    owner = kh == 0;  // (PARTS: the other waves stay for the barriers below)
    if (owner) {
      const float m_b = mo[(NDVB * 16) * 64 + lane], l_b = mo[(NDVB * 16 + 1) * 64 + lane];
      const float m_new = fmaxf(m_run, m_b);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      const float fa = __builtin_amdgcn_exp2f(m_run - m_safe), fb = __builtin_amdgcn_exp2f(m_b - m_safe);
      l_run = l_run * fa + l_b * fb;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < NDVB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] = o_acc[i][r] * fa + mo[(16 * i + r) * 64 + lane] * fb;
    }
  }
  if constexpr (PARTS) {
    if (nparts > 1) {  // (workgroup-uniform)
      // Every part stores its (O, m, l) write-through (agent-scope relaxed atomics = sc1: the other parts may run on
      // another XCD, and a device-scope fence costs more than it saves -- attention_decode.hip arrive_and_merge), counts
      // itself in, and the part that completes the count merges ALL parts in part order (its own included: the result does
      // not depend on which part came last).
      constexpr int OW = GH * NPB;                 // owner waves per workgroup
      constexpr int PART = (NDVB * 16 + 2) * 64;   // floats per stored partial
      float* base = a.part_ws + ((int64_t)item * a.smax * OW + (wave % OW)) * PART;
      // a partial = (NDVB * 8 + 1) pairs per lane, pair e of lane l at [e][l]: 8-byte stores / loads, 512 B per wave-instruction
      typedef unsigned long long u64;
      auto pack2 = [](float x, float y) -> u64 {
        return (u64)__builtin_bit_cast(unsigned, x) | ((u64)__builtin_bit_cast(unsigned, y) << 32);
      };
      if (owner) {
        u64* mine = reinterpret_cast<u64*>(base + (int64_t)part * OW * PART);
#pragma unroll
        for (int i = 0; i < NDVB; ++i)
#pragma unroll
          for (int r = 0; r < 16; r += 2)
            __hip_atomic_store(mine + (8 * i + (r >> 1)) * 64 + lane, pack2(o_acc[i][r], o_acc[i][r + 1]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + (NDVB * 8) * 64 + lane, pack2(m_run, l_run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores are acknowledged
      __syncthreads();                                  // ... and everyone's; idx_lds is dead
      int* s_last = reinterpret_cast<int*>(idx_lds);
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(a.part_counters + item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = old == nparts - 1;
        // the next launch starts from zero again (nobody else touches the counter before then)
        if (old == nparts - 1) __hip_atomic_store(a.part_counters + item, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (!*s_last || !owner) return;
      m_run = -INFINITY;
      l_run = 0.f;
#pragma unroll
      for (int i = 0; i < NDVB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
      for (int q = 0; q < nparts; ++q) {
        const u64* pq = reinterpret_cast<const u64*>(base + (int64_t)q * OW * PART);
        u64 w[NDVB * 8 + 1];  // every load of the part in flight before the first use
#pragma unroll
        for (int e = 0; e < NDVB * 8 + 1; ++e)
          w[e] = __hip_atomic_load(pq + e * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float m_b = __builtin_bit_cast(float, (unsigned)w[NDVB * 8]);
        const float l_b = __builtin_bit_cast(float, (unsigned)(w[NDVB * 8] >> 32));
        const float m_new = fmaxf(m_run, m_b);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float fa = __builtin_amdgcn_exp2f(m_run - m_safe), fb = __builtin_amdgcn_exp2f(m_b - m_safe);
        l_run = l_run * fa + l_b * fb;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < NDVB; ++i)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const u64 v = w[8 * i + (r >> 1)];
            o_acc[i][r] = o_acc[i][r] * fa + __builtin_bit_cast(float, (unsigned)v) * fb;
            o_acc[i][r + 1] = o_acc[i][r + 1] * fa + __builtin_bit_cast(float, (unsigned)(v >> 32)) * fb;
          }
      }
    } else if (!owner) {
      return;
    }
  }
  if (q_ok) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    T* op = reinterpret_cast<T*>(a.o) + (q_start + qpos) * a.o_st + (int64_t)head * a.o_sh;
#pragma unroll
    for (int dvb = 0; dvb < NDVB; ++dvb)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = H::from_f32(o_acc[dvb][4 * r4 + j] * inv);
        *reinterpret_cast<x4*>(op + 32 * dvb + 8 * r4 + 4 * hh) = v;
      }
  }
}

// Generic fallback: one wave per (query row, head); any head sizes up to 1024.
template <int DTYPE, typename IdxT>
__global__ __launch_bounds__(64) void extend_generic_kernel(ExtendArgs a, int D, int Dv, int max_len_extend) {
  using H = Half16<DTYPE>;
  using T = typename H::T;
  constexpr int MAXE = 16;
  const int lane = threadIdx.x;
  int bid = blockIdx.x;
  const int r = bid % max_len_extend;
  bid /= max_len_extend;
  const int h = bid % a.num_heads;
  const int b = bid / a.num_heads;
  int64_t idx_base, q_start;
  int prefix, ext;
  seq_info(a, b, idx_base, prefix, ext, q_start);
  if (r >= ext) return;
  const int kvh = h / a.group;
  float qv[MAXE], acc[MAXE];
  const T* qp = reinterpret_cast<const T*>(a.q) + (q_start + r) * a.q_st + (int64_t)h * a.q_sh;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int d = lane + 64 * i;
    qv[i] = d < D ? H::to_f32(qp[d]) : 0.f;
    acc[i] = 0.f;
  }
  float m_run = -INFINITY, l_run = 0.f;
  const IdxT* idx = reinterpret_cast<const IdxT*>(a.indices) + idx_base;
  const int n_new = a.causal ? r + 1 : ext;
  const uint8_t* mrow = a.mask ? a.mask + a.mask_indptr[b] + (int64_t)r * (prefix + ext) : nullptr;
  for (int n = 0; n < prefix + n_new; ++n) {
    if (n < prefix) {
      if (a.window > 0 && !(r <= n + a.window)) continue;
      if (mrow && !a.skip_prefix_mask && !mrow[n]) continue;
    } else if (mrow && !mrow[n]) {
      continue;
    }
    const T *kp, *vp;
    if (n < prefix) {
      const int64_t tok = (int64_t)idx[n];
      kp = reinterpret_cast<const T*>(a.kb) + tok * a.kb_sn + (int64_t)kvh * a.kb_sh;
      vp = reinterpret_cast<const T*>(a.vb) + tok * a.vb_sn + (int64_t)kvh * a.vb_sh;
    } else {
      kp = reinterpret_cast<const T*>(a.ke) + (q_start + n - prefix) * a.ke_st + (int64_t)kvh * a.ke_sh;
      vp = reinterpret_cast<const T*>(a.ve) + (q_start + n - prefix) * a.ve_st + (int64_t)kvh * a.ve_sh;
    }
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
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  T* op = reinterpret_cast<T*>(a.o) + (q_start + r) * a.o_st + (int64_t)h * a.o_sh;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int d = lane + 64 * i;
    if (d < Dv) op[d] = H::from_f32(acc[i] * inv);
  }
}

#if SGLM_EXT_TIMING
inline unsigned long long* ext_timing_buffer() {
  static unsigned long long* buf = [] {
    void* p = nullptr;
    const size_t n = (size_t)kExtTimingWgs * kExtTimingTiles * kExtTimingStamps * 8;
    if (hipMalloc(&p, n) != hipSuccess) return (unsigned long long*)nullptr;
    (void)hipMemset(p, 0, n);
    return (unsigned long long*)p;
  }();
  return buf;
}
#endif

template <int DTYPE, int D, typename IdxT, int GH, bool MASKED, bool KV8 = false, int KSPLIT = 1, int NSTAGE = 2,
          bool PARTS = false>
int launch_mfma(ExtendArgs a, int64_t batch, int max_len_extend, hipStream_t s) {
  auto kern = extend_mfma_kernel<DTYPE, D, IdxT, GH, MASKED, KV8, KSPLIT, NSTAGE, PARTS>;
  constexpr int lds = NSTAGE * 2 * kBN * D * 2 + kIdxCap * 4;
  static int attr_rc = check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
      "hipFuncSetAttribute");
  if (attr_rc) return attr_rc;
  constexpr int BP = 32 * (4 / (GH * KSPLIT));
  a.num_mblocks = (max_len_extend + BP - 1) / BP;
  const int64_t grid = batch * (a.num_heads / GH) * a.num_mblocks * (PARTS ? a.smax : 1);
  if (grid <= 0) return 0;
  if (grid >= (1ll << 31)) {
    set_error("extend_attention: grid too large");
    return SGL_MI355_ERR_INVALID_ARGUMENT;
  }
  // (Tried, round 3: with 512 workgroups on 256 CUs, the second 256 in ASCENDING weight so that a CU gets ranks j and
  //  511 - j instead of j and 256 + j -- no change, 35.9 vs 36.1 us at 1024 tokens: profiles/r03_extend_fold.txt.)
#if SGLM_EXT_TIMING
  if constexpr (!PARTS) {
    if (a.part_ws == nullptr) a.part_ws = reinterpret_cast<float*>(ext_timing_buffer());
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a);
  return check_hip(hipGetLastError(), "extend_mfma_kernel launch");
}

// what the caller of the _parts entry point knows about the batch and lends to the launch
struct PartsHint {
  int64_t max_prefix_len = 0;
  int64_t ws_floats = 0, n_counters = 0;
};

// KV-range parts for this form (GH heads x NPB position blocks per workgroup, D = 128)?  Sets a.smax / a.part_tiles.
// Only launches that leave the chip part empty behind long chains: at most 128 items whose longest has 12 tiles or more
// (up to 256 from 48 tiles, two parts); up to 4 parts per item, 8 up to 32 items (the merging workgroup reads them all back: more parts, longer tail),
// parts of at least 4 tiles.  SGL_MI355_EXTEND_PARTS=0 switches it off, SGL_MI355_EXTEND_SMAX=n sets the cap, SGL_MI355_EXTEND_PARTS_MAX_ITEMS=n
// the item limit (A/B aids).
inline bool plan_parts(ExtendArgs& a, const PartsHint& h, int64_t batch, int GH, int NPB, int max_len_extend) {
  static const int parts_env = [] { const char* e = getenv("SGL_MI355_EXTEND_PARTS"); return e ? atoi(e) : 1; }();
  static const int smax_env = [] { const char* e = getenv("SGL_MI355_EXTEND_SMAX"); return e ? atoi(e) : 0; }();
  if (!parts_env || a.part_ws == nullptr || a.part_counters == nullptr) return false;
  const int BP = 32 * NPB;
  const int64_t items = batch * (a.num_heads / GH) * ((max_len_extend + BP - 1) / BP);
  const int64_t ntot_max = (h.max_prefix_len + kBN - 1) / kBN + (max_len_extend + kBN - 1) / kBN;
  // (measured, profiles/r04_extend_parts.txt: with more than 128 items the partials' write-through stores and the merging
  //  workgroups' reads cost more than the shorter chains give back -- 4 requests of 128 new tokens behind 1024-token prefixes
  //  27.9 -> 45.7 us, the single 1024-token prefill 27.0 -> 33.8; one such request 24.9 -> 20.4, behind 4096 / 16384 tokens
  //  74.9 / 262.6 -> 34.4 / 89.8)
  //  Up to 256 items TWO parts still pay once the chains are long (48 tiles): 4 x 128 / 2 x 256 / 1 x 512 new tokens behind 4096
  //  84.2 / 83.4 / 82.2 -> 77.4 / 68.3 / 66.2 us, 3 x 128 behind 8192 138.5 -> 105.7; three parts give it back (81.1 / 81.4 / 80.2).
  static const int items_env = [] { const char* e = getenv("SGL_MI355_EXTEND_PARTS_MAX_ITEMS"); return e ? atoi(e) : 0; }();
  const bool few = items <= (items_env > 0 ? items_env : 128);
  const bool some = items_env <= 0 && items <= 256 && ntot_max >= 48;
  if (!(few || some) || items > 512 || ntot_max < 12 || items > h.n_counters) return false;
  // parts per item: 4, 8 up to 32 items (same box, us, 4 -> 8 parts: 64 new tokens behind 8192 50.7 -> 39.6, one kv head's rank
  // 30.2 -> 29.0; but 64 items 34.5 -> 38.2, 128 items 44.2 -> 68.0: the merging workgroup reads every part back)
  const int64_t cap = smax_env > 0 ? smax_env : (items <= 32 ? 8 : (few ? 4 : 2));
  int64_t smax = 1024 / items;
  smax = smax > cap ? cap : smax;
  if (smax > (ntot_max + 3) / 4) smax = (ntot_max + 3) / 4;
  if (smax < 2) return false;
  const int64_t part_floats = (128 / 32 * 16 + 2) * 64;
  if (items * smax * GH * NPB * part_floats > h.ws_floats) return false;
  a.smax = (int)smax;
  a.part_tiles = (int)((ntot_max + smax - 1) / smax);
  if (a.part_tiles < 4) a.part_tiles = 4;
  return true;
}

template <int DTYPE, typename IdxT>
int dispatch(ExtendArgs a, int64_t batch, int D, int Dv, int max_len_extend, hipStream_t s, const PartsHint& hint = PartsHint{}) {
  const bool aligned = (a.q_st % 8 == 0) && (a.q_sh % 8 == 0) && (a.ke_st % 8 == 0) && (a.ke_sh % 8 == 0) &&
                       (a.ve_st % 8 == 0) && (a.ve_sh % 8 == 0) && (a.kb_sn % 8 == 0) && (a.kb_sh % 8 == 0) &&
                       (a.vb_sn % 8 == 0) && (a.vb_sh % 8 == 0) && (a.o_st % 4 == 0) && (a.o_sh % 4 == 0) &&
                       (reinterpret_cast<uintptr_t>(a.q) % 16 == 0) && (reinterpret_cast<uintptr_t>(a.ke) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(a.ve) % 16 == 0) && (reinterpret_cast<uintptr_t>(a.kb) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(a.vb) % 16 == 0) && (reinterpret_cast<uintptr_t>(a.o) % 8 == 0);
  if (a.kv8) {
    // e4m3 pool: Triton form only (the CPU op has no FP8 pool), MFMA kernel only
    if constexpr (std::is_same<IdxT, int32_t>::value) {
      const bool ok8 = D == Dv && (D == 128 || D == 64) && (a.q_st % 8 == 0) && (a.q_sh % 8 == 0) && (a.ke_st % 8 == 0) &&
                       (a.ke_sh % 8 == 0) && (a.ve_st % 8 == 0) && (a.ve_sh % 8 == 0) && (a.kb_sn % 16 == 0) &&
                       (a.kb_sh % 16 == 0) && (a.vb_sn % 16 == 0) && (a.vb_sh % 16 == 0) && (a.o_st % 4 == 0) &&
                       (a.o_sh % 4 == 0) && (reinterpret_cast<uintptr_t>(a.q) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(a.ke) % 16 == 0) && (reinterpret_cast<uintptr_t>(a.ve) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(a.kb) % 16 == 0) && (reinterpret_cast<uintptr_t>(a.vb) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(a.o) % 8 == 0) && a.mode == 0;
      if (!ok8) {
        set_error("extend_attention_fwd_fp8kv: needs head_size == head_size_v in {64, 128} and 16-byte aligned rows");
        return SGL_MI355_ERR_UNSUPPORTED;
      }
      const int gh = (a.group % 4 == 0) ? 4 : (a.group % 2 == 0 ? 2 : 1);
#define EXT_LAUNCH8(DD, GG) return launch_mfma<DTYPE, DD, int32_t, GG, true, true>(a, batch, max_len_extend, s)
      if (D == 128) {
        if (gh == 4) EXT_LAUNCH8(128, 4);
        if (gh == 2) EXT_LAUNCH8(128, 2);
        EXT_LAUNCH8(128, 1);
      } else {
        if (gh == 4) EXT_LAUNCH8(64, 4);
        if (gh == 2) EXT_LAUNCH8(64, 2);
        EXT_LAUNCH8(64, 1);
      }
#undef EXT_LAUNCH8
    } else {
      set_error("extend_attention: e4m3 pools are supported by the kv_indices form only");
      return SGL_MI355_ERR_UNSUPPORTED;
    }
  }
  // (row strides below 2^26 elements: the MFMA kernel's DMA offsets are 32-bit byte counts)
  const int64_t max_row = int64_t(1) << 26;
  const bool rows32 = a.ke_st > 0 && a.ve_st > 0 && a.kb_sn > 0 && a.vb_sn > 0 && a.ke_st < max_row && a.ve_st < max_row &&
                      a.kb_sn < max_row && a.vb_sn < max_row;
  if (D == Dv && aligned && rows32 && (D == 128 || D == 64)) {
    const int gh = (a.group % 4 == 0) ? 4 : (a.group % 2 == 0 ? 2 : 1);
    // (a logit cap takes the MASKED instantiation too: its per-element tanh is not in the plain kernels at all -- round 5)
    const bool masked = a.mask != nullptr || a.window > 0 || a.logit_cap > 0.f;
    if constexpr (std::is_same<IdxT, int32_t>::value) {
      // few workgroups (a short prefill of one or two requests): split every tile's keys over wave pairs so that the
      // longest query block, which sets the kernel time when there are at most about two workgroups per CU, takes half
      // as long.  Measured, bs=1 Llama-3-8B heads, L = 512 / 1024 / 2048 / 4096 tokens: 19 / 35 / 82 / 295 us split,
      // 28 / 51 / 100 / 242 us unsplit.  SGL_MI355_EXTEND_KSPLIT=0|1 overrides (tuning aid).
      static const int ks_env = [] { const char* e = getenv("SGL_MI355_EXTEND_KSPLIT"); return e ? atoi(e) : -1; }();
      const int bp1 = 32 * (4 / gh);
      const int64_t grid1 = batch * (a.num_heads / gh) * ((max_len_extend + bp1 - 1) / bp1);
      // Round 4: short extends as well while the unsplit form has at most 128 workgroups (half a workgroup per CU): same box, us,
      // unsplit -> split: one request of 64 / 128 / 192 new tokens 9.1 / 10.1 / 11.5 -> 7.4 / 8.1 / 9.1; 128 new tokens behind
      // a 1024- / 4096-token prefix 32.6 / 97.1 -> 24.6 / 74.2; four such requests 34.6 -> 28.9; eight tie (38.5 / 39.0),
      // sixteen lose (58.5 -> 94.4)  (profiles/r04_extend_short_ksplit.txt)
      const bool ksplit = ks_env >= 0 ? ks_env != 0 : ((grid1 <= 512 && max_len_extend >= 256) || grid1 <= 128);
      if (ksplit && !masked && D == 128) {
        const int ghk = gh == 4 ? 2 : 1;
        if (plan_parts(a, hint, batch, ghk, 4 / (ghk * 2), max_len_extend)) {
          if (ghk == 2) return launch_mfma<DTYPE, 128, int32_t, 2, false, false, 2, 2, true>(a, batch, max_len_extend, s);
          return launch_mfma<DTYPE, 128, int32_t, 1, false, false, 2, 2, true>(a, batch, max_len_extend, s);
        }
      }
      // (the forms without the key split start at 129 workgroups: never few enough items for parts)
      if (ksplit && !masked) {
        // (Round 3: more tiles in flight -- three or four LDS stages, one workgroup per CU instead of two -- do NOT help: 38.8 /
        //  38.9 us against 35.8 at 1024 tokens, 83 against 54 at 1536 (profiles/r03_extend_nstage.txt): the longest block's
        //  chain is issue-bound, not a chain of exposed DMA round trips.  Since round 5 the tile loop is unrolled over exactly
        //  two stages (immediate LDS offsets), and the deeper A/B variants are gone from the source.)
#define EXT_KS(DD, GG) return launch_mfma<DTYPE, DD, int32_t, GG, false, false, 2, 2>(a, batch, max_len_extend, s)
        if (D == 128) {
          if (gh == 4) EXT_KS(128, 2);
          EXT_KS(128, 1);
        }
        if (gh == 4) EXT_KS(64, 2);
        EXT_KS(64, 1);
#undef EXT_KS
      }
    }
#define EXT_LAUNCH(DD, GG)                                                             \
  return masked ? launch_mfma<DTYPE, DD, IdxT, GG, true>(a, batch, max_len_extend, s)   \
                : launch_mfma<DTYPE, DD, IdxT, GG, false>(a, batch, max_len_extend, s)
    if (D == 128) {
      if (gh == 4) EXT_LAUNCH(128, 4);
      if (gh == 2) EXT_LAUNCH(128, 2);
      EXT_LAUNCH(128, 1);
    } else {
      if (gh == 4) EXT_LAUNCH(64, 4);
      if (gh == 2) EXT_LAUNCH(64, 2);
      EXT_LAUNCH(64, 1);
    }
#undef EXT_LAUNCH
  }
  const int64_t grid = batch * a.num_heads * (int64_t)max_len_extend;
  if (grid <= 0) return 0;
  if (grid >= (1ll << 31)) {
    set_error("extend_attention: grid too large for the generic kernel");
    return SGL_MI355_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL((extend_generic_kernel<DTYPE, IdxT>), dim3((unsigned)grid), dim3(64), 0, s, a, D, Dv, max_len_extend);
  return check_hip(hipGetLastError(), "extend_generic_kernel launch");
}

int check_common(int64_t batch, int64_t Hq, int64_t Hkv, int64_t D, int64_t Dv, int64_t max_len_extend, int dtype) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "extend_attention: dtype must be bf16 (0) or fp16 (1)");
  SGLM_CHECK_ARG(batch >= 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "extend_attention: bad head counts %ld/%ld", (long)Hq, (long)Hkv);
  SGLM_CHECK_ARG(D > 0 && Dv > 0 && D <= 1024 && Dv <= 1024, "extend_attention: head sizes must be in [1,1024]");
  SGLM_CHECK_ARG(max_len_extend >= 0 && max_len_extend < (1ll << 30), "extend_attention: bad max_len_extend");
  return 0;
}

}  // namespace
}  // namespace sglm

using namespace sglm;

#if SGLM_EXT_TIMING
// timing build only: copy the stamps out (device-synchronising) and clear them
extern "C" int sgl_mi355_extend_timing_dump(void* host_buf, int64_t bytes) {
  const int64_t n = (int64_t)kExtTimingWgs * kExtTimingTiles * kExtTimingStamps * 8;
  SGLM_CHECK_ARG(host_buf && bytes >= n, "extend_timing_dump: buffer of %ld bytes needed", (long)n);
  unsigned long long* d = ext_timing_buffer();
  SGLM_CHECK_ARG(d != nullptr, "extend_timing_dump: no debug buffer");
  SGLM_CHECK_HIP(hipDeviceSynchronize());
  SGLM_CHECK_HIP(hipMemcpy(host_buf, d, n, hipMemcpyDeviceToHost));
  SGLM_CHECK_HIP(hipMemset(d, 0, n));
  return 0;
}
#endif

static int extend_fwd_impl(int kv8,
    
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend, const void* k_buffer,
    const void* v_buffer, const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int is_causal,
    int64_t max_len_extend, int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h, int64_t kb_stride_n,
    int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h, float sm_scale, float logit_cap,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t sliding_window_size,
    int dtype, void* stream, int64_t max_prefix_len = 0, float* workspace = nullptr, int64_t workspace_floats = 0,
    int32_t* counters = nullptr, int64_t num_counters = 0) {
  int rc = check_common(batch, num_heads, num_kv_heads, head_size, head_size_v, max_len_extend, dtype);
  if (rc) return rc;
  if (batch == 0 || max_len_extend == 0) return 0;
  SGLM_CHECK_ARG(!kv8 || (k_buffer && v_buffer && kv_indices), "extend_attention_fwd_fp8kv: null pool pointer");
  SGLM_CHECK_ARG(q_extend && k_extend && v_extend && o_extend && qo_indptr && kv_indptr,
                 "extend_attention_fwd: null tensor pointer");
  SGLM_CHECK_ARG(custom_mask == nullptr || mask_indptr != nullptr, "extend_attention_fwd: custom_mask needs mask_indptr");
  SGLM_CHECK_ARG(sliding_window_size < (1ll << 30), "extend_attention_fwd: bad sliding_window_size");
  ExtendArgs a{};
  a.q = q_extend; a.q_st = q_stride_t; a.q_sh = q_stride_h;
  a.ke = k_extend; a.ke_st = ke_stride_t; a.ke_sh = ke_stride_h;
  a.ve = v_extend; a.ve_st = ve_stride_t; a.ve_sh = ve_stride_h;
  a.o = o_extend; a.o_st = o_stride_t; a.o_sh = o_stride_h;
  a.kb = k_buffer; a.kb_sn = kb_stride_n; a.kb_sh = kb_stride_h;
  a.vb = v_buffer; a.vb_sn = vb_stride_n; a.vb_sh = vb_stride_h;
  a.qo_indptr = qo_indptr; a.kv_indptr = kv_indptr; a.indices = kv_indices; a.mode = 0;
  a.num_heads = (int)num_heads; a.num_kv_heads = (int)num_kv_heads; a.group = (int)(num_heads / num_kv_heads);
  a.sm_scale = sm_scale; a.logit_cap = logit_cap; a.causal = is_causal;
  a.mask = custom_mask; a.mask_indptr = mask_indptr; a.skip_prefix_mask = skip_prefix_custom_mask;
  a.window = sliding_window_size > 0 ? (int)sliding_window_size : 0;
  a.kv8 = kv8;
  PartsHint hint;
  if (workspace != nullptr && counters != nullptr && !kv8 && max_prefix_len >= 0) {
    a.part_ws = workspace;
    a.part_counters = counters;
    hint.max_prefix_len = max_prefix_len;
    hint.ws_floats = workspace_floats;
    hint.n_counters = num_counters;
  }
  hipStream_t s = as_stream(stream);
  return dtype == SGL_MI355_BF16
             ? dispatch<SGL_MI355_BF16, int32_t>(a, batch, (int)head_size, (int)head_size_v, (int)max_len_extend, s, hint)
             : dispatch<SGL_MI355_FP16, int32_t>(a, batch, (int)head_size, (int)head_size_v, (int)max_len_extend, s, hint);
}

// The same with what lets short launches use the whole chip (round 4): max_prefix_len = an upper bound of the batch's cached
// prefix lengths (the kernel reads the true ones from kv_indptr; the bound only sizes the split), an fp32 workspace and
// int32 counters (ZERO before the first call; the kernel leaves them zero), both owned by the caller and not shared with a
// launch that may run concurrently.  With at most 512 (query block, head group) items whose longest has 8 key tiles or
// more, every item's tiles are cut into up to 4 consecutive ranges over as many workgroups; each stores its (O, m, l), the
// last one to finish merges them in range order (deterministic).  16-bit pools, head size 128, no custom mask / window; anything else, a
// null workspace or one that is too small runs exactly as sgl_mi355_extend_attention_fwd.
extern "C" int sgl_mi355_extend_attention_fwd_parts(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend, const void* k_buffer,
    const void* v_buffer, const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int is_causal,
    int64_t max_len_extend, int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h, int64_t kb_stride_n,
    int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h, float sm_scale, float logit_cap,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t sliding_window_size,
    int dtype, void* stream, int64_t max_prefix_len, float* workspace, int64_t workspace_floats, int32_t* counters,
    int64_t num_counters) {
  SGLM_CHECK_ARG(max_prefix_len >= 0 && workspace_floats >= 0 && num_counters >= 0, "extend_attention_fwd_parts: negative size");
  return extend_fwd_impl(0, q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
                         is_causal, max_len_extend, batch, num_heads, num_kv_heads, head_size, head_size_v, q_stride_t,
                         q_stride_h, ke_stride_t, ke_stride_h, ve_stride_t, ve_stride_h, o_stride_t, o_stride_h,
                         kb_stride_n, kb_stride_h, vb_stride_n, vb_stride_h, sm_scale, logit_cap, custom_mask, mask_indptr,
                         skip_prefix_custom_mask, sliding_window_size, dtype, stream, max_prefix_len, workspace,
                         workspace_floats, counters, num_counters);
}

extern "C" int sgl_mi355_extend_attention_fwd(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend, const void* k_buffer,
    const void* v_buffer, const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int is_causal,
    int64_t max_len_extend, int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h, int64_t kb_stride_n,
    int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h, float sm_scale, float logit_cap,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t sliding_window_size,
    int dtype, void* stream) {
  return extend_fwd_impl(0, q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
                         is_causal, max_len_extend, batch, num_heads, num_kv_heads, head_size, head_size_v, q_stride_t,
                         q_stride_h, ke_stride_t, ke_stride_h, ve_stride_t, ve_stride_h, o_stride_t, o_stride_h,
                         kb_stride_n, kb_stride_h, vb_stride_n, vb_stride_h, sm_scale, logit_cap, custom_mask, mask_indptr,
                         skip_prefix_custom_mask, sliding_window_size, dtype, stream);
}

// Same contract with k_buffer / v_buffer holding e4m3 bytes (pool strides in elements = bytes).
extern "C" int sgl_mi355_extend_attention_fwd_fp8kv(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend, const void* k_buffer,
    const void* v_buffer, const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int is_causal,
    int64_t max_len_extend, int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h, int64_t kb_stride_n,
    int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h, float sm_scale, float logit_cap,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t sliding_window_size,
    int dtype, void* stream) {
  return extend_fwd_impl(1, q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
                         is_causal, max_len_extend, batch, num_heads, num_kv_heads, head_size, head_size_v, q_stride_t,
                         q_stride_h, ke_stride_t, ke_stride_h, ve_stride_t, ve_stride_h, o_stride_t, o_stride_h,
                         kb_stride_n, kb_stride_h, vb_stride_n, vb_stride_h, sm_scale, logit_cap, custom_mask, mask_indptr,
                         skip_prefix_custom_mask, sliding_window_size, dtype, stream);
}

// float8_e5m2 pool: Q and P of the prefix stage are rounded to e5m2 instead
extern "C" int sgl_mi355_extend_attention_fwd_fp8kv_e5m2(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend, const void* k_buffer,
    const void* v_buffer, const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int is_causal,
    int64_t max_len_extend, int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h, int64_t kb_stride_n,
    int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h, float sm_scale, float logit_cap,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t sliding_window_size,
    int dtype, void* stream) {
  return extend_fwd_impl(2, q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
                         is_causal, max_len_extend, batch, num_heads, num_kv_heads, head_size, head_size_v, q_stride_t,
                         q_stride_h, ke_stride_t, ke_stride_h, ve_stride_t, ve_stride_h, o_stride_t, o_stride_h,
                         kb_stride_n, kb_stride_h, vb_stride_n, vb_stride_h, sm_scale, logit_cap, custom_mask, mask_indptr,
                         skip_prefix_custom_mask, sliding_window_size, dtype, stream);
}

extern "C" int sgl_mi355_extend_attention(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend, const void* k_buffer,
    const void* v_buffer, const void* req_to_token, int req_to_token_is64, const int64_t* req_pool_indices,
    const int64_t* seq_lens, const int64_t* extend_seq_lens, const int64_t* extend_start_loc, int64_t max_len_extend,
    int64_t num_seqs, int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h, int64_t kb_stride_n,
    int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h, float sm_scale, float logit_cap, int dtype,
    void* stream) {
  int rc = check_common(num_seqs, num_heads, num_kv_heads, head_size, head_size_v, max_len_extend, dtype);
  if (rc) return rc;
  if (num_seqs == 0 || max_len_extend == 0) return 0;
  SGLM_CHECK_ARG(q_extend && k_extend && v_extend && o_extend && req_to_token && req_pool_indices && seq_lens &&
                     extend_seq_lens && extend_start_loc,
                 "extend_attention: null tensor pointer");
  ExtendArgs a{};
  a.q = q_extend; a.q_st = q_stride_t; a.q_sh = q_stride_h;
  a.ke = k_extend; a.ke_st = ke_stride_t; a.ke_sh = ke_stride_h;
  a.ve = v_extend; a.ve_st = ve_stride_t; a.ve_sh = ve_stride_h;
  a.o = o_extend; a.o_st = o_stride_t; a.o_sh = o_stride_h;
  a.kb = k_buffer; a.kb_sn = kb_stride_n; a.kb_sh = kb_stride_h;
  a.vb = v_buffer; a.vb_sn = vb_stride_n; a.vb_sh = vb_stride_h;
  a.req_pool_indices = req_pool_indices; a.seq_lens = seq_lens; a.extend_seq_lens = extend_seq_lens;
  a.extend_start_loc = extend_start_loc; a.r2t_stride = max_context_len; a.indices = req_to_token; a.mode = 1;
  a.num_heads = (int)num_heads; a.num_kv_heads = (int)num_kv_heads; a.group = (int)(num_heads / num_kv_heads);
  a.sm_scale = sm_scale; a.logit_cap = logit_cap; a.causal = 1;  // extend.cpp is always causal
  hipStream_t s = as_stream(stream);
  const int D = (int)head_size, Dv = (int)head_size_v, ml = (int)max_len_extend;
  if (dtype == SGL_MI355_BF16)
    return req_to_token_is64 ? dispatch<SGL_MI355_BF16, int64_t>(a, num_seqs, D, Dv, ml, s)
                             : dispatch<SGL_MI355_BF16, int32_t>(a, num_seqs, D, Dv, ml, s);
  return req_to_token_is64 ? dispatch<SGL_MI355_FP16, int64_t>(a, num_seqs, D, Dv, ml, s)
                           : dispatch<SGL_MI355_FP16, int32_t>(a, num_seqs, D, Dv, ml, s);
}
