// AWQ INT4 decode GEMM on a k-packed copy of the weights (M <= 64).
//
// Replaces (fused): AWQLinearMethod.apply = awq_dequantize(qweight, scales, qzeros) then x @ W
//   -- python/sglang/srt/layers/quantization/awq.py:401-418, awq_triton.py:13-107 / awq_kernel.cu:126-221.
//
// Why a repack.  The checkpoint layout packs 8 OUTPUT columns into one int32 (qweight int32 [K, N/8], nibble
// ORDER[j] = column 8c+j, awq_triton.py:56-69), i.e. the contraction index k is the slow dimension, while an MFMA
// operand register holds 8 consecutive CONTRACTION values of one column.  Streaming the checkpoint layout therefore
// needs a transposition per tile (the first kernel, awq.hip: dequantise into LDS, read back transposed -- 0.05 of the
// HBM roofline at M = 64).  `process_weights_after_loading` is allowed to repack, so the linear method
// keeps a second copy built ONCE by awq_repack_kernel:
//     wp uint32 [N][K/8] : dword (n, kk) = the 8 nibbles of column n for k = 8kk .. 8kk+7, stored in nibble order
//                          [v0 v2 v4 v6 v1 v3 v5 v7] so that ((w >> 4t) & 0x000F000F) | 0x64006400 is the fp16 pair
//                          (1024 + v_2t, 1024 + v_2t+1);
//     sz uint32 [N][K/G] : {fp16 scale, fp16 (1024 + zero)} of column n, group g.
// A lane (column n, k-group kg) then loads ONE dwordx4 = its 32 k-values of a 128-k step and dequantises in
// registers with the reference's exact arithmetic: (1024+nib) - (1024+zero) is exact in fp16, times scale rounds once
// -- the same `(w - z) * s` in the scales dtype as awq_triton.py:101-104 / awq_kernel.cu:60-115.
//
// The streaming structure is the FP8 one (gemm_fp8.hip, fp8_gemm_wstream_kernel): consumer waves own a 16-column
// block, weights through registers PB steps ahead with hand-kept vmcnt counts, the fp16 activations of a phase are
// LDS-DMA'd by a producer wave into a double-buffered, XOR-swizzled [step][row][256 B] image; optional split-K slabs.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include "common.h"

namespace sglm {
namespace {

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

struct AwqPArgs {
  const uint8_t* x;   // fp16 [M][K]
  int64_t x_sm;       // bytes between rows
  const uint8_t* wp;  // uint32 [N][K/8]
  const uint32_t* sz; // uint32 [N][K/G]
  const void* bias;   // fp16 [N] or null
  void* out;          // fp16 [M][N]
  int M, N, K;        // K here is the PADDED K (multiple of 512): the packed buffers are built for it
  int gshift;         // log2(G / 128): sz index of k-step s is s >> gshift
  int ngroups;        // padded K / G (row length of sz)
  int real_steps;     // ceil(real K / 128): activation steps that exist; later steps are zero weights
};

constexpr int ORDER[8] = {0, 4, 1, 5, 2, 6, 3, 7};  // awq_triton.py:56-69: column 8c+j lives in nibble ORDER[j]

// Round 2: both buffers are FRAGMENT-MAJOR (as the FP8 weights, gemm_fp8.hip GemmArgs::b_shuf).  A decode lane
// (column r16 of block nb, k-group kg) loads the 16 bytes = dwords 16 ks + 4 kg .. + 3 of its column for k-step ks; row-major
// wp made that 16 rows x 64 B per instruction, rows Kp/2 bytes apart.  Stored instead as one contiguous KiB per
// (block, step), [kg][r16][16 B]; sz as 64 contiguous bytes per (block, group), [r16].  Logical shapes (wp [N][Kp/8],
// sz [N][Kp/G]) are unchanged; the buffers hold ceil(N / 16) * 16 columns.
__host__ __device__ __forceinline__ int64_t wp_index(int n, int kk, int Kp) {  // dword index of (column n, dword kk)
  return ((int64_t)(n >> 4) * (Kp >> 7) + (kk >> 4)) * 256 + (((kk >> 2) & 3) * 16 + (n & 15)) * 4 + (kk & 3);
}
__host__ __device__ __forceinline__ int64_t sz_index(int n, int gp, int ngp) {
  return ((int64_t)(n >> 4) * ngp + gp) * 16 + (n & 15);
}

// ---------------------------------------------------------------- repack (once per layer)
__global__ __launch_bounds__(256) void awq_repack_kernel(const uint32_t* __restrict__ qweight,
                                                         const _Float16* __restrict__ scales,
                                                         const uint32_t* __restrict__ qzeros, uint32_t* __restrict__ wp,
                                                         uint32_t* __restrict__ sz, int K, int Kp, int N8, int G) {
  // K is padded to Kp (multiple of 512) with weights that dequantise to exactly 0: nibble == zero point of the
  // last group (the kernel multiplies them with a valid, finite activation step).
  const int64_t total = (int64_t)(Kp / 8) * N8;
  const int N = N8 * 8;
  const int ng = K / G, ngp = (Kp + G - 1) / G;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % N8);  // consecutive threads: consecutive int32 columns of the same 8 rows (coalesced)
    const int kk = (int)(i / N8);
    uint32_t w[8];
    const uint32_t zlast = qzeros[(int64_t)(ng - 1) * N8 + c];
#pragma unroll
    for (int r = 0; r < 8; ++r) w[r] = (8 * kk + r) < K ? qweight[(int64_t)(8 * kk + r) * N8 + c] : zlast;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = (w[r] >> (4 * ORDER[j])) & 0xFu;
      const uint32_t d = v[0] | (v[2] << 4) | (v[4] << 8) | (v[6] << 12) | (v[1] << 16) | (v[3] << 20) | (v[5] << 24) |
                         (v[7] << 28);
      wp[wp_index(8 * c + j, kk, Kp)] = d;
    }
  }
  const int64_t total2 = (int64_t)ngp * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total2; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i % N), gp = (int)(i / N);
    const int g = gp < ng ? gp : ng - 1;
    const uint32_t z = (qzeros[(int64_t)g * N8 + n / 8] >> (4 * ORDER[n % 8])) & 0xFu;
    const _Float16 zb = (_Float16)(float)(1024 + (int)z);
    const _Float16 sc = scales[(int64_t)g * N + n];
    sz[sz_index(n, gp, ngp)] = (uint32_t)__builtin_bit_cast(uint16_t, sc) | ((uint32_t)__builtin_bit_cast(uint16_t, zb) << 16);
  }
}

// ---------------------------------------------------------------- hand-scheduled loads (see gemm_fp8.hip 4.3.1)
struct WFrag {
  i32x4 w;      // 32 nibbles: this lane's 32 k-values of the step
  uint32_t sz;  // {scale, 1024 + zero} of the step's group
};
// NT: non-temporal hint on the weight stream (read once per step); the scale/zero words stay cacheable
template <bool NT>
__device__ __forceinline__ void wload_asm(WFrag& f, const uint8_t* wbase, uint32_t woff, const uint32_t* sbase, uint32_t soff) {
  if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(f.w) : "v"(woff), "s"(wbase) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f.w) : "v"(woff), "s"(wbase) : "memory");
  asm volatile("global_load_dword %0, %1, %2" : "=v"(f.sz) : "v"(soff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_wfrag(WFrag& f) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(f.w), "+v"(f.sz) : "n"(N) : "memory");
}
template <int PB>
__device__ __forceinline__ void drain_wfrags(WFrag (&q)[PB]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < PB; ++i) asm volatile("" ::"v"(q[i].w), "v"(q[i].sz));
}

// 8 nibbles of one dword -> 8 fp16 (MFMA operand order), exact reference arithmetic
__device__ __forceinline__ f16x8 dequant8(uint32_t w, f16x2 zb2, f16x2 sc2) {
  f16x8 r;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t bits = ((w >> (4 * t)) & 0x000F000Fu) | 0x64006400u;
    f16x2 h = __builtin_bit_cast(f16x2, bits);
    h = (h - zb2) * sc2;  // (1024+nib) - (1024+zero) exact; one rounding in the product
    r[2 * t] = h[0];
    r[2 * t + 1] = h[1];
  }
  return r;
}

// Timing ablations (variant builds only, WRONG results): -DSGLM_AWQ_ABL_NODEQ=1 feeds the raw nibble words to the MFMA (no
// dequant VALU work), -DSGLM_AWQ_ABL_NOMFMA=1 drops the A-fragment LDS reads and the MFMAs, -DSGLM_AWQ_ABL_NODMA=1 drops the
// producer's activation DMA (the barriers stay).
#ifndef SGLM_AWQ_ABL_NODEQ
#define SGLM_AWQ_ABL_NODEQ 0
#endif
#ifndef SGLM_AWQ_ABL_NOMFMA
#define SGLM_AWQ_ABL_NOMFMA 0
#endif
#ifndef SGLM_AWQ_ABL_NODMA
#define SGLM_AWQ_ABL_NODMA 0
#endif

template <int MB, int PH, bool SLAB>
__global__ __launch_bounds__(576) void awq_wstream_kernel(AwqPArgs p, float* slabs, int phases_per_slice) {
  static_assert(PH * MB <= 16, "one fp16 A buffer is at most 64 KiB");
  constexpr int ROWS = 16 * MB;
  constexpr int STEP_BYTES = ROWS * 256;      // 128 k x 2 B per row
  constexpr int BUF_BYTES = PH * STEP_BYTES;  // <= 64 KiB
  // weight k-steps in flight per wave: shallow, as in gemm_fp8.hip (deeper queues only let the waves drift apart).
  // Same-box A/B of 8 / 4 / 2: M = 64 qkv 22.0 / 22.0 / 21.3, gate_up 30.7 / 30.3 / 29.3, down 24.2 / 24.0 / 22.8 us;
  // M = 1 qkv 15.8 / 14.7 / 14.9, down 15.7 / 14.0 / 16.1 us.
  // (round 2, fragment-major weights: the unsplit kernel takes four steps and the nt hint, as in gemm_fp8.hip)
#ifndef SGLM_AWQ_UNSPLIT_DEEP
#define SGLM_AWQ_UNSPLIT_DEEP 1
#endif
  constexpr bool kDeep = SGLM_AWQ_UNSPLIT_DEEP && !SLAB;
  constexpr int PB = (MB == 4 && !kDeep) ? 2 : (PH >= 4 ? 4 : PH);
  constexpr int UPS = 4 * MB;                 // 1-KiB DMA units (4 rows x 256 B) per k-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NC = (int)(blockDim.x >> 6) - 1;  // consumer waves; wave NC is the DMA producer
  const int r16 = lane & 15, kg = lane >> 4;
  const int P_total = (p.K >> 7) / PH;
  const int ph0 = SLAB ? (int)blockIdx.y * phases_per_slice : 0;
  const int ph1 = SLAB ? (ph0 + phases_per_slice < P_total ? ph0 + phases_per_slice : P_total) : P_total;
  const int nph = ph1 - ph0;
  const int nb = blockIdx.x * NC + wave;
  const int n = nb * 16 + r16;
  const bool n_ok = n < p.N;
  const uint32_t smem_base = lds_addr_of(smem);

  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (wave == NC) {
    // ---------------- producer: lane i of a unit lands at chunk i & 15 of row i >> 4 and fetches source chunk
    // (i & 15) ^ (row & 15) of that row (swizzle on the source side, inside one coalesced 256-B segment)
    const int lrow = lane >> 4;
    const uint8_t* a_lane[UPS];
#pragma unroll
    for (int u = 0; u < UPS; ++u) {
      const int drow = u * 4 + lrow;
      const int dj = (lane & 15) ^ (drow & 15);
      a_lane[u] = p.x + (int64_t)(drow < p.M ? drow : p.M - 1) * p.x_sm + 16 * dj;  // rows past M: never stored
    }
    auto dma_phase = [&](int ph, int buf) __attribute__((always_inline)) {
#if SGLM_AWQ_ABL_NODMA
      return;
#endif
      for (int sl = 0; sl < PH; ++sl) {
        int asl = ph * PH + sl;  // padded steps re-read the last real one (finite values x zero weights)
        asl = asl < p.real_steps ? asl : p.real_steps - 1;
#pragma unroll
        for (int u = 0; u < UPS; ++u)
          lds_dma16(a_lane[u] + (int64_t)asl * 256, smem_base + buf * BUF_BYTES + sl * STEP_BYTES + u * 1024);
      }
    };
    dma_phase(ph0, 0);
    for (int lp = 0; lp < nph; ++lp) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // phase lp has landed
      __syncthreads();                                   // barrier #lp
      if (lp + 1 < nph) dma_phase(ph0 + lp + 1, (lp + 1) & 1);
    }
    __syncthreads();  // the consumers' final barrier
    return;
  }

  // ---------------- consumers
  const int nbc = nb * 16 < p.N ? nb : 0;  // a wave past the last block streams block 0 (never stored)
  const uint8_t* w_blk = p.wp + (int64_t)nbc * (p.K >> 7) * 1024;  // fragment-major: one KiB per (block, step)
  const uint32_t* s_blk = p.sz + (int64_t)nbc * p.ngroups * 16;
  const uint32_t w_off = (uint32_t)lane * 16;
  const uint32_t s_off = (uint32_t)r16 * 4;
  const int rot = (nb * 3) & (PH - 1);
  const int last = ph1 * PH - 1;
  int f_pf = ph0 * PH;
  auto refill = [&](WFrag& fr) __attribute__((always_inline)) {
    const int f = f_pf < last ? f_pf : last;
    const int ks = (f & ~(PH - 1)) + ((f + rot) & (PH - 1));
    wload_asm<kDeep>(fr, w_blk + (int64_t)ks * 1024, w_off, s_blk + (ks >> p.gshift) * 16, s_off);
    ++f_pf;
  };
  WFrag wq[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) refill(wq[i]);

  // A fragment (row 16mb + r16, chunk 4kg + j) sits at row*256 + 16*((4kg + j) ^ r16)
#pragma clang loop unroll(disable)  // exactly one copy of the step body (gemm_fp8.hip 4.3.1 f)
  for (int lp = 0; lp < nph; ++lp) {
    __syncthreads();  // barrier #lp: phase lp is in LDS
    const uint32_t abuf = (lp & 1) * BUF_BYTES;
    for (int s0 = 0; s0 < PH; s0 += PB) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int t = (s0 + i + rot) & (PH - 1);
        wait_wfrag<2 * (PB - 1)>(wq[i]);
        const uint32_t szv = wq[i].sz;
        const f16x2 sc2 = __builtin_bit_cast(f16x2, (szv & 0xFFFFu) | (szv << 16));
        const f16x2 zb2 = __builtin_bit_cast(f16x2, (szv >> 16) | (szv & 0xFFFF0000u));
        f16x8 bf[4];
#if SGLM_AWQ_ABL_NODEQ
        {
          (void)zb2;
          const i32x4 raw = {wq[i].w[0], wq[i].w[1], wq[i].w[2], wq[i].w[3] ^ (int)__builtin_bit_cast(uint32_t, sc2)};
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[j] = __builtin_bit_cast(f16x8, raw);
        }
#else
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = dequant8((uint32_t)wq[i].w[j], zb2, sc2);
#endif
        const char* arow = smem + abuf + t * STEP_BYTES + r16 * 256;
#if SGLM_AWQ_ABL_NOMFMA
        (void)arow;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] += (float)bf[j][0];
#else
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f16x8 af = *reinterpret_cast<const f16x8*>(arow + mb * 4096 + 16 * ((4 * kg + j) ^ r16));
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[mb], 0, 0, 0);
          }
#endif
        refill(wq[i]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  drain_wfrags(wq);
  __syncthreads();  // the A buffers are dead: reuse their memory
  if constexpr (SLAB) {
    float* ep = reinterpret_cast<float*>(smem) + wave * (ROWS * 20);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(16 * mb + 4 * kg + r) * 20 + r16] = acc[mb][r];
    wait_lgkmcnt0();
    float* dst = slabs + (int64_t)blockIdx.y * p.M * p.N;
    for (int c = lane; c < ROWS * 4; c += 64) {
      const int m = c >> 2, q = c & 3;
      const int nn = nb * 16 + q * 4;
      if (m < p.M && nn < p.N)
        *reinterpret_cast<f32x4*>(dst + (int64_t)m * p.N + nn) = *reinterpret_cast<const f32x4*>(ep + m * 20 + q * 4);
    }
  } else {
    _Float16* ep = reinterpret_cast<_Float16*>(smem) + wave * (ROWS * 24);
    const float bv = p.bias ? (float)reinterpret_cast<const _Float16*>(p.bias)[n_ok ? n : p.N - 1] : 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(16 * mb + 4 * kg + r) * 24 + r16] = (_Float16)(acc[mb][r] + bv);
    wait_lgkmcnt0();
    for (int c = lane; c < ROWS * 2; c += 64) {
      const int m = c >> 1, half = c & 1;
      const int nn = nb * 16 + half * 8;
      if (m < p.M && nn < p.N)
        *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(p.out) + (int64_t)m * p.N + nn) =
            *reinterpret_cast<const uint4*>(ep + m * 24 + half * 8);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Round 3: TWO column blocks per consumer wave, deep weight queue, dequant pipelined under the MFMAs.
//
// Timing ablations of the kernel above (profiles/r03_awq_ablation.txt, M = 64, 4096 -> 22016): 27.6 us as built; 20.6
// without the dequant VALU work; 17.1 without the LDS fragment reads + MFMAs; 15.7 with neither -- i.e. the bare
// streaming skeleton already takes twice what the 45 MB need, and dequant and MFMA run one after the other on top.
//   * skeleton: a wave kept 4 k-steps x 1 KiB in flight and a CU 6 such waves: 24 KiB per CU against ~2 us of loaded
//     memory latency = 12 GB/s per CU (the FP8 streamer keeps 56 KiB).  Here a wave owns 32 columns (two 16-column
//     blocks, each with its own queue) and keeps PB = 8 steps of both in flight: 16 KiB per wave, 48-64 KiB per CU.
//   * LDS: the fp16 A fragments of a k-step (16 KiB at M = 64) are read ONCE per wave and feed both blocks -- half the
//     LDS bytes per weight byte (the old kernel's 8 waves x 16 KiB per k-step round = 1024 clk of LDS per CU).
//   * dequant: 13 VALU per dword instead of 19.  The odd nibbles are not shifted down: (w & 0x00F000F0) | 0x54005400 is
//     the fp16 pair (64 + n) because the mantissa bit 4 of an fp16 in [64, 128) weighs 1, so one shift by 8 serves all four
//     pairs of a dword; the second zero-point constant is 64 + z.  (64 + n) - (64 + z) = n - z exactly, times the scale
//     rounds once: the reference's arithmetic, bit for bit.  v_and_or_b32 with the mask in an SGPR and the magic in a VGPR
//     (a VOP3 instruction of this ISA takes neither a literal nor two scalar operands).
//   * one wave per SIMD (<= 4 consumers + the DMA producer), so the in-order issue of a wave must itself overlap VALU and
//     MFMA: the loop is rotated by half a step -- while the 16 MFMAs of block b run, the other block's next 32 k-values are
//     dequantised; the raw registers of a slot are refilled right after its dequant, not after its MFMAs.
__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask_s, uint32_t magic_v) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mask_s), "v"(magic_v));
  return r;
}
struct DqConst {
  uint32_t magic_lo, magic_hi;  // 0x64006400 / 0x54005400 in VGPRs
};
__device__ __forceinline__ f16x8 dequant8_v2(uint32_t w, f16x2 zlo, f16x2 zhi, f16x2 sc2, const DqConst& c) {
  const uint32_t w8 = w >> 8;
  const f16x2 h0 = (__builtin_bit_cast(f16x2, and_or(w, 0x000F000Fu, c.magic_lo)) - zlo) * sc2;
  const f16x2 h1 = (__builtin_bit_cast(f16x2, and_or(w, 0x00F000F0u, c.magic_hi)) - zhi) * sc2;
  const f16x2 h2 = (__builtin_bit_cast(f16x2, and_or(w8, 0x000F000Fu, c.magic_lo)) - zlo) * sc2;
  const f16x2 h3 = (__builtin_bit_cast(f16x2, and_or(w8, 0x00F000F0u, c.magic_hi)) - zhi) * sc2;
  return f16x8{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
}

template <int MB, int PH, int PB, bool SLAB>
__global__ __launch_bounds__(320) void awq_wstream2_kernel(AwqPArgs p, float* slabs, int phases_per_slice) {
  static_assert(PH * MB <= 16, "one fp16 A buffer is at most 64 KiB");
  static_assert(PB % PH == 0 || PH % PB == 0, "PB and PH are powers of two");
  constexpr int ROWS = 16 * MB;
  constexpr int STEP_BYTES = ROWS * 256;
  constexpr int BUF_BYTES = PH * STEP_BYTES;
  constexpr int UPS = 4 * MB;
  constexpr int NQ = 2 * PB;  // half-steps (block, k-step) in flight per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NC = (int)(blockDim.x >> 6) - 1;
  const int r16 = lane & 15, kg = lane >> 4;
  const int P_total = (p.K >> 7) / PH;
  const int ph0 = SLAB ? (int)blockIdx.y * phases_per_slice : 0;
  const int ph1 = SLAB ? (ph0 + phases_per_slice < P_total ? ph0 + phases_per_slice : P_total) : P_total;
  const int nph = ph1 - ph0;
  const uint32_t smem_base = lds_addr_of(smem);

  if (wave == NC) {  // ---------------- producer (as in awq_wstream_kernel)
    const int lrow = lane >> 4;
    const uint8_t* a_lane[UPS];
#pragma unroll
    for (int u = 0; u < UPS; ++u) {
      const int drow = u * 4 + lrow;
      const int dj = (lane & 15) ^ (drow & 15);
      a_lane[u] = p.x + (int64_t)(drow < p.M ? drow : p.M - 1) * p.x_sm + 16 * dj;
    }
    auto dma_phase = [&](int ph, int buf) __attribute__((always_inline)) {
      for (int sl = 0; sl < PH; ++sl) {
        int asl = ph * PH + sl;
        asl = asl < p.real_steps ? asl : p.real_steps - 1;
#pragma unroll
        for (int u = 0; u < UPS; ++u)
          lds_dma16(a_lane[u] + (int64_t)asl * 256, smem_base + buf * BUF_BYTES + sl * STEP_BYTES + u * 1024);
      }
    };
    dma_phase(ph0, 0);
    for (int lp = 0; lp < nph; ++lp) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (lp + 1 < nph) dma_phase(ph0 + lp + 1, (lp + 1) & 1);
    }
    __syncthreads();
    return;
  }

  // ---------------- consumers: unit = column blocks 2u and 2u + 1
  const int unit = blockIdx.x * NC + wave;
  const int nblocks = (p.N + 15) >> 4;
  const uint8_t* w_blk[2];
  const uint32_t* s_blk[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int nb = 2 * unit + b;
    const int nbc = nb < nblocks ? nb : 0;  // a block past the end streams block 0 (never stored)
    w_blk[b] = p.wp + (int64_t)nbc * (p.K >> 7) * 1024;
    s_blk[b] = p.sz + (int64_t)nbc * p.ngroups * 16;
  }
  const uint32_t w_off = (uint32_t)lane * 16;
  const uint32_t s_off = (uint32_t)r16 * 4;
  const int rot = (unit * 3) & (PH - 1);
  const int first = ph0 * PH, last = ph1 * PH - 1;
  DqConst dq;
  asm volatile("v_mov_b32 %0, 0x64006400\n\tv_mov_b32 %1, 0x54005400" : "=v"(dq.magic_lo), "=v"(dq.magic_hi));

  // half-step h (h even: block 0, odd: block 1) of flat step first + h / 2; loads past the end re-read the last step
  int h_pf = 0;
  auto refill = [&](WFrag& fr, int b) __attribute__((always_inline)) {
    int f = first + (h_pf >> 1);
    f = f < last ? f : last;
    const int ks = (f & ~(PH - 1)) + ((f + rot) & (PH - 1));
    wload_asm<!SLAB>(fr, w_blk[b] + (int64_t)ks * 1024, w_off, s_blk[b] + (ks >> p.gshift) * 16, s_off);
    ++h_pf;
  };
  WFrag wq[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) refill(wq[i], i & 1);

  f32x4 acc[2][MB];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[b][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto dequant_slot = [&](WFrag& fr, f16x8 (&bf)[4]) __attribute__((always_inline)) {
    const uint32_t szv = fr.sz;
    const f16x2 sc2 = __builtin_bit_cast(f16x2, (szv & 0xFFFFu) | (szv << 16));
    const f16x2 zlo = __builtin_bit_cast(f16x2, (szv >> 16) | (szv & 0xFFFF0000u));
    const f16x2 zhi = zlo - f16x2{(_Float16)960.f, (_Float16)960.f};  // (1024 + z) - 960 = 64 + z, exact
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = dequant8_v2((uint32_t)fr.w[j], zlo, zhi, sc2, dq);
  };

  f16x8 bf[2][4];  // [parity of the half-step]
  wait_wfrag<2 * (NQ - 1)>(wq[0]);
  dequant_slot(wq[0], bf[0]);
  refill(wq[0], 0);

  f16x8 af[MB][4];
  const int nsteps = nph * PH;  // a multiple of PB (launcher)
#pragma clang loop unroll(disable)  // exactly one copy of the body (gemm_fp8.hip 4.3.1 f)
  for (int f0 = 0; f0 < nsteps; f0 += PB) {
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int fl = f0 + i;  // local flat step
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int cur = (2 * i + b) & 1, nxt = cur ^ 1;
        const int slot_n = (2 * i + b + 1) % NQ;  // the next half-step's slot
        if (b == 0) {
          // a new phase starts here: wait until the producer's DMA of it is in LDS (static for PH <= PB)
          if ((i % PH == 0) && (PH <= PB || (f0 & (PH - 1)) == 0)) __syncthreads();
          const int t = (fl + rot) & (PH - 1);
          const char* arow = smem + ((fl / PH) & 1) * BUF_BYTES + t * STEP_BYTES + r16 * 256;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              af[mb][j] = *reinterpret_cast<const f16x8*>(arow + mb * 4096 + 16 * ((4 * kg + j) ^ r16));
        }
        wait_wfrag<2 * (NQ - 1)>(wq[slot_n]);
        dequant_slot(wq[slot_n], bf[nxt]);  // next half-step's operands, under this half-step's MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j)  // j outside: consecutive MFMAs go to different accumulators (each still sums j = 0..3 in order)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) acc[b][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb][j], bf[cur][j], acc[b][mb], 0, 0, 0);
        // the dequantised operands are complete HERE (pure VALU ops would otherwise sink to their use in the next
        // half-step, in front of its MFMAs instead of under these), interleaved 1 MFMA : 4 VALU by the groups below
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bf[nxt][j]));
        if (b == 0) __builtin_amdgcn_sched_group_barrier(0x100, 4 * MB, 0);  // the step's A fragments first
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                    // ... and a first run of VALU under their latency
#pragma unroll
        for (int k = 0; k < 4 * MB; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, (56 + 4 * MB - 1) / (4 * MB), 0);
        }
        refill(wq[slot_n], (b + 1) & 1);    // its raw registers are free again
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  drain_wfrags(wq);
  __syncthreads();  // the A buffers are dead: reuse their memory
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int nb = 2 * unit + b;
    const int n = nb * 16 + r16;
    const bool n_ok = n < p.N;
    if constexpr (SLAB) {
      float* ep = reinterpret_cast<float*>(smem) + (wave * 2 + b) * (ROWS * 20);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) ep[(16 * mb + 4 * kg + r) * 20 + r16] = acc[b][mb][r];
      wait_lgkmcnt0();
      float* dst = slabs + (int64_t)blockIdx.y * p.M * p.N;
      for (int c = lane; c < ROWS * 4; c += 64) {
        const int m = c >> 2, q = c & 3;
        const int nn = nb * 16 + q * 4;
        if (m < p.M && nn < p.N)
          *reinterpret_cast<f32x4*>(dst + (int64_t)m * p.N + nn) = *reinterpret_cast<const f32x4*>(ep + m * 20 + q * 4);
      }
    } else {
      _Float16* ep = reinterpret_cast<_Float16*>(smem) + (wave * 2 + b) * (ROWS * 24);
      const float bv = p.bias ? (float)reinterpret_cast<const _Float16*>(p.bias)[n_ok ? n : p.N - 1] : 0.f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) ep[(16 * mb + 4 * kg + r) * 24 + r16] = (_Float16)(acc[b][mb][r] + bv);
      wait_lgkmcnt0();
      for (int c = lane; c < ROWS * 2; c += 64) {
        const int m = c >> 1, half = c & 1;
        const int nn = nb * 16 + half * 8;
        if (m < p.M && nn < p.N)
          *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(p.out) + (int64_t)m * p.N + nn) =
              *reinterpret_cast<const uint4*>(ep + m * 24 + half * 8);
      }
    }
  }
}

// Sum of the K slices (in slice order) + bias.  64-thread workgroups: the slabs (SK x M x N floats, several MB) come
// from beyond this XCD's L2 and a CU draws only ~10 B/clk from there, so the read is spread over as many CUs as there
// are 512-element pieces (round 3: 256-thread workgroups left half of the chip idle at N = 4096: 4.3-4.6 us per launch).
__global__ __launch_bounds__(64) void awq_packed_finalize_kernel(AwqPArgs p, const float* slabs, int SK) {
  const int64_t total = (int64_t)p.M * p.N / 8;
  const int64_t sstride = (int64_t)p.M * p.N;
  for (int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 64) {
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s0 = 0; s0 < SK; s0 += 4) {  // four slices in flight; sums in slice order
      f32x4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* src = slabs + (int64_t)(s0 + u < SK ? s0 + u : SK - 1) * sstride + i * 8;
        a[u] = *reinterpret_cast<const f32x4*>(src);
        b[u] = *reinterpret_cast<const f32x4*>(src + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u < SK) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] += a[u][j];
            v[4 + j] += b[u][j];
          }
        }
    }
    const int64_t e = i * 8;
    const int nn = (int)(e % p.N);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float r = v[j];
      if (p.bias) r += (float)reinterpret_cast<const _Float16*>(p.bias)[nn + j];
      o[j] = (_Float16)r;
    }
    *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(p.out) + e) = o;
  }
}

// set by sgl_mi355_awq_gemm_packed_partials around its call: leave the K slices unsummed and report their count; decline
// (nothing launched) where the dispatcher takes the unsplit form
thread_local int32_t* tl_awq_slices = nullptr;

template <bool SLAB>
inline int awq_partials_precheck() {
  if (tl_awq_slices != nullptr && !SLAB) {
    set_error("awq_gemm_packed_partials: this shape runs unsplit (no partial sums to hand over)");
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  return 0;
}

template <int MB, int PH, bool SLAB>
int launch_ph(const AwqPArgs& p, float* slabs, int SK, int pps, int nc, int groups, hipStream_t s) {
  if (int pre = awq_partials_precheck<SLAB>()) return pre;
  auto kern = awq_wstream_kernel<MB, PH, SLAB>;
  constexpr int lds = 2 * PH * 16 * MB * 256;
  static int attr_rc = check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
      "hipFuncSetAttribute");
  if (attr_rc) return attr_rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)groups, (unsigned)SK), dim3(64 * (nc + 1)), lds, s, p, slabs, pps);
  int rc = check_hip(hipGetLastError(), "awq_wstream launch");
  if (rc || !SLAB) return rc;
  if (tl_awq_slices != nullptr) { *tl_awq_slices = SK; return 0; }  // the consumer sums the slices (+ bias, one rounding)
  const int64_t total = (int64_t)p.M * p.N / 8;
  hipLaunchKernelGGL(awq_packed_finalize_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, p,
                     (const float*)slabs, SK);
  return check_hip(hipGetLastError(), "awq_packed_finalize launch");
}

template <int MB, int PH, int PB, bool SLAB>
int launch2_ph(const AwqPArgs& p, float* slabs, int SK, int pps, int nc, int groups, hipStream_t s) {
  if (int pre = awq_partials_precheck<SLAB>()) return pre;
  auto kern = awq_wstream2_kernel<MB, PH, PB, SLAB>;
  constexpr int lds = 2 * PH * 16 * MB * 256;
  static int attr_rc = check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
      "hipFuncSetAttribute");
  if (attr_rc) return attr_rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)groups, (unsigned)SK), dim3(64 * (nc + 1)), lds, s, p, slabs, pps);
  int rc = check_hip(hipGetLastError(), "awq_wstream2 launch");
  if (rc || !SLAB) return rc;
  if (tl_awq_slices != nullptr) { *tl_awq_slices = SK; return 0; }
  const int64_t total = (int64_t)p.M * p.N / 8;
  hipLaunchKernelGGL(awq_packed_finalize_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, p,
                     (const float*)slabs, SK);
  return check_hip(hipGetLastError(), "awq_packed_finalize launch");
}

// The two-blocks-per-wave kernel: (phase length, consumer waves <= 4, K slices) by the same reasoning as launch() below --
// one workgroup per CU, fewest weight bytes on the busiest CU -- with the per-workgroup fixed cost weighing two "waves".
template <int MB>
int launch2(const AwqPArgs& p, float* slabs, int64_t slab_floats, hipStream_t s, bool& used) {
  used = false;
  const int steps = p.K >> 7;
  const int units = ((p.N + 15) / 16 + 1) / 2;
  constexpr int PH_BIG = 16 / MB;
  const int ph = (steps % PH_BIG == 0) ? PH_BIG : 4;
  if (steps % ph != 0) return 0;
  const int P = steps / ph;
  static const int force_nc = [] { const char* e = getenv("SGL_MI355_AWQ2_NC"); return e ? atoi(e) : 0; }();  // tuning aid
  int nc = 0, SK = 1, pps = 0, best = 1 << 30;
  for (int c = 4; c >= 2; --c) {
    if (force_nc && c != force_nc) continue;
    const int groups_c = (units + c - 1) / c;
    int sk = 1, pp = P;
    if (slabs != nullptr && groups_c < 200) {
      int sk_max = 256 / groups_c;
      if (sk_max > P) sk_max = P;
      if (sk_max < 1) sk_max = 1;
      pp = (P + sk_max - 1) / sk_max;
      sk = (P + pp - 1) / pp;
      if ((int64_t)sk * p.M * p.N > slab_floats) { sk = 1; pp = P; }
    }
    const int rounds = (groups_c * sk + 255) / 256;
    const int cost = rounds * pp * ph * (c + 2) + (sk > 1 ? 40 : 0);  // + the finalize launch
    if (cost < best) { best = cost; nc = c; SK = sk; pps = pp; }
  }
  if (nc == 0) return 0;
  const int groups = (units + nc - 1) / nc;
  const bool pb8 = (pps * ph) % 8 == 0;
  used = true;
#define AWQ2_GO(PH_, PB_)                                                                          \
  return SK > 1 ? launch2_ph<MB, PH_, PB_, true>(p, slabs, SK, pps, nc, groups, s)                 \
                : launch2_ph<MB, PH_, PB_, false>(p, nullptr, 1, pps, nc, groups, s)
  if constexpr (PH_BIG > 4) { if (ph == PH_BIG) AWQ2_GO(PH_BIG, 8); }
  if (pb8) AWQ2_GO(4, 8);
  AWQ2_GO(4, 4);
#undef AWQ2_GO
}

template <int MB>
int launch(const AwqPArgs& p, float* slabs, int64_t slab_floats, hipStream_t s) {
  // Which kernel (same box, M = 1 / 16 / 64, us, v2 vs v1, profiles/r03_awq_v2_points.txt): 4096 -> 22016 17.0 / 17.5 / 24.7 vs
  // 17.2 / 17.6 / 25.5; 11008 -> 4096 12.6 / 14.4 / 19.2 vs 12.9 / 13.3 / 22.0; 4096 -> 12288 13.8 / 14.2 / 19.8 vs 13.2 / 13.6 /
  // 19.7; 4096 -> 4096 11.6 / 12.5 / 15.5 vs 8.5 / 8.9 / 13.2.  The two-blocks-per-wave kernel wins where a workgroup has a
  // long K walk (unsplit wide N, or long K at M > 32), the shallow one where a slice is a handful of steps.
  // SGL_MI355_AWQ_V1=1 / SGL_MI355_AWQ_V2=1 force one of them (A/B aid).
  static const bool v1 = getenv("SGL_MI355_AWQ_V1") != nullptr;
  static const bool v2 = getenv("SGL_MI355_AWQ_V2") != nullptr;
  const bool wide = (p.N + 15) / 16 >= 2 * 3 * 200;          // >= 200 workgroups of three 32-column units without splitting K
  const bool long_k = (p.K >> 7) >= 64 && MB == 4;           // K >= 8192 at M > 32
  if (!v1 && (v2 || wide || long_k)) {
    bool used = false;
    const int rc = launch2<MB>(p, slabs, slab_floats, s, used);
    if (used || rc) return rc;
  }
  const int steps = p.K >> 7;
  const int nblocks = (p.N + 15) / 16;
  // (PH, consumer waves, K slices) as in gemm_fp8.hip launch_wstream: fewest k-steps on the busiest CU.
  // Split K only when the slab workspace is there and the unsplit grid leaves CUs idle.
  int PH = 0, nc = 8, SK = 1, pps = 0, best = 1 << 30;
  for (int ph = 16 / MB; ph >= 4; ph >>= 1) {
    if (steps % ph != 0) continue;
    const int P = steps / ph;
    for (int c = 8; c >= 4; --c) {
      const int groups_c = (nblocks + c - 1) / c;
      int sk = 1, pp = P;
      if (slabs != nullptr && groups_c < 200) {
        int sk_max = 256 / groups_c;
        if (sk_max > P) sk_max = P;
        if (sk_max < 1) sk_max = 1;
        pp = (P + sk_max - 1) / sk_max;
        sk = (P + pp - 1) / pp;
        if ((int64_t)sk * p.M * p.N > slab_floats) { sk = 1; pp = P; }
      }
      const int rounds = (groups_c * sk + 255) / 256;
      const int cost = rounds * c * pp * (ph + 4) + (sk > 1 ? 6 * c : 0);  // + the finalize launch
      if (cost < best) { best = cost; PH = ph; nc = c; SK = sk; pps = pp; }
    }
  }
  if (PH == 0) {
    set_error("awq_gemm_packed: K / 128 (%d) must be a multiple of 4", steps);
    return SGL_MI355_ERR_INVALID_ARGUMENT;
  }
  const int groups = (nblocks + nc - 1) / nc;
#define AWQ_GO(PH_)                                                                              \
  return SK > 1 ? launch_ph<MB, PH_, true>(p, slabs, SK, pps, nc, groups, s)                      \
                : launch_ph<MB, PH_, false>(p, nullptr, 1, pps, nc, groups, s)
  if constexpr (MB == 1) { if (PH == 16) AWQ_GO(16); }
  if constexpr (MB <= 2) { if (PH == 8) AWQ_GO(8); }
  AWQ_GO(4);
#undef AWQ_GO
}

}  // namespace
}  // namespace sglm

using namespace sglm;

extern "C" int64_t sgl_mi355_awq_packed_k(int64_t K) { return (K + 511) / 512 * 512; }

extern "C" int sgl_mi355_awq_repack(const int32_t* qweight, const void* scales, const int32_t* qzeros, uint32_t* wp,
                                    uint32_t* sz, int64_t K, int64_t N, int64_t group_size, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_FP16, "awq_repack: the packed path is fp16 only (scales dtype of AWQ checkpoints)");
  SGLM_CHECK_ARG(K > 0 && N > 0 && N % 8 == 0 && K % 128 == 0 && group_size > 0 && K % group_size == 0,
                 "awq_repack: bad shape K=%ld N=%ld G=%ld", (long)K, (long)N, (long)group_size);
  SGLM_CHECK_ARG(K < (1ll << 30) && N < (1ll << 31), "awq_repack: shape too large");
  SGLM_CHECK_ARG(qweight && scales && qzeros && wp && sz, "awq_repack: null tensor pointer");
  const int64_t Kp = sgl_mi355_awq_packed_k(K);
  const int64_t total = (Kp / 8) * (N / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 65535 * 8 ? (total + 255) / 256 : 65535 * 8);
  hipLaunchKernelGGL(awq_repack_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const uint32_t*)qweight,
                     (const _Float16*)scales, (const uint32_t*)qzeros, wp, sz, (int)K, (int)Kp, (int)(N / 8),
                     (int)group_size);
  return check_hip(hipGetLastError(), "awq_repack launch");
}

extern "C" int sgl_mi355_awq_gemm_packed(const void* x, const uint32_t* wp, const uint32_t* sz, const void* bias, void* out,
                                         float* workspace, int64_t workspace_floats, int64_t M, int64_t N, int64_t K,
                                         int64_t group_size, int64_t x_stride_m, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_FP16, "awq_gemm_packed: fp16 only");
  SGLM_CHECK_ARG(M >= 0 && M <= 64, "awq_gemm_packed: M <= 64 (got %ld); larger M goes through awq_dequantize + GEMM", (long)M);
  SGLM_CHECK_ARG(K > 0 && N > 0 && N % 8 == 0 && K % 128 == 0, "awq_gemm_packed: N %% 8 == 0 and K %% 128 == 0 required (K=%ld N=%ld)",
                 (long)K, (long)N);
  SGLM_CHECK_ARG(group_size >= 128 && (group_size & (group_size - 1)) == 0 && K % group_size == 0,
                 "awq_gemm_packed: group_size must be a power of two >= 128 dividing K (got %ld)", (long)group_size);
  const int64_t Kp = sgl_mi355_awq_packed_k(K);
  const int64_t ngp = (Kp + group_size - 1) / group_size;
  SGLM_CHECK_ARG(N * (Kp / 2) < (1ll << 32) && N * ngp * 4 < (1ll << 32), "awq_gemm_packed: weight too large for 32-bit offsets");
  if (M == 0) return 0;
  SGLM_CHECK_ARG(x && wp && sz && out, "awq_gemm_packed: null tensor pointer");
  SGLM_CHECK_ARG(reinterpret_cast<uintptr_t>(x) % 16 == 0 && (x_stride_m * 2) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0,
                 "awq_gemm_packed: x rows and out must be 16-byte aligned");
  AwqPArgs p{(const uint8_t*)x, x_stride_m * 2, (const uint8_t*)wp, sz, bias, out, (int)M, (int)N, (int)Kp, 0, (int)ngp,
             (int)(K / 128)};
  for (int64_t g = group_size / 128; g > 1; g >>= 1) ++p.gshift;
  hipStream_t s = as_stream(stream);
  if (M <= 16) return launch<1>(p, workspace, workspace_floats, s);
  if (M <= 32) return launch<2>(p, workspace, workspace_floats, s);
  return launch<4>(p, workspace, workspace_floats, s);
}

// sgl_mi355_awq_gemm_packed's split-K form WITHOUT its finalize launch: the fp32 partial sums [num_slices][M][N] stay in
// `workspace` for a consumer that runs the epilogue itself (sum in slice order, + bias, one rounding: awq_packed_finalize_kernel's
// arithmetic -- what the FP8 path's *_from_partials entry points do on unit scales).  SGL_MI355_ERR_UNSUPPORTED, nothing
// launched, where the dispatcher would run the shape unsplit.
extern "C" int sgl_mi355_awq_gemm_packed_partials(const void* x, const uint32_t* wp, const uint32_t* sz, float* workspace,
                                                  int64_t workspace_floats, int64_t M, int64_t N, int64_t K, int64_t group_size,
                                                  int64_t x_stride_m, int dtype, int32_t* num_slices, void* stream) {
  SGLM_CHECK_ARG(workspace != nullptr && workspace_floats > 0 && num_slices != nullptr && M > 0,
                 "awq_gemm_packed_partials: null workspace / num_slices, or no rows");
  *num_slices = 0;
  tl_awq_slices = num_slices;
  // (`out` is not written in this form; the workspace stands in for the non-null / alignment checks)
  const int rc = sgl_mi355_awq_gemm_packed(x, wp, sz, nullptr, workspace, workspace, workspace_floats, M, N, K, group_size, x_stride_m,
                                           dtype, stream);
  tl_awq_slices = nullptr;
  return rc;
}
