// AWQ INT4 prefill GEMM (M > 64) on the MFMA cores with the weights kept as INT4 in HBM.
//
// Replaces (fused): AWQLinearMethod.apply at prefill sizes = awq_dequantize(qweight, scales, qzeros) then x @ W
//   -- python/sglang/srt/layers/quantization/awq.py:401-418; the reference's own fused form is awq_gemm_triton
//   (awq_triton.py:110-229, 289-339).  Before this kernel the linear method dequantised the whole weight to fp16 (4x the
//   INT4 footprint, kept per layer) and called the library GEMM.
//
// Operands: x fp16 [M][K]; the k-packed weight copy of awq_packed.hip (wp uint32 [N][Kp/8]: 8 nibbles of column n per
// dword, nibble order such that ((w >> 4t) & 0x000F000F) | 0x64006400 is the fp16 pair (1024 + v_2t, 1024 + v_2t+1);
// sz uint32 [N][Kp/G] = {scale, 1024 + zero}); Kp = K padded to 512 with weights that dequantise to exactly 0.
// Dequantisation is the reference's arithmetic, done in registers on the way into the MFMA: (1024 + nib) - (1024 + zero)
// is exact in fp16 and the product with the scale rounds once -- `(w - z) * s` in the scales dtype (awq_triton.py:101-104).
//
// Structure: block tile 128 x 128 x 64, four waves side by side along N (each 128 rows x 32 columns: a wave dequantises
// only its own columns -- no redundant unpacking), three LDS stages filled by LDS-DMA two k-steps ahead, ONE barrier per
// k-step:
//   A  [128 rows][128 B]  = 16 KiB / stage, 16-B chunk index XOR-swizzled by (row >> 1) & 7 on the SOURCE side (a DMA
//                           instruction lands 8 rows x 128 B linearly); fragments by ds_read_b128, conflict-free;
//   W  [128 cols][32 B]   =  4 KiB / stage (64 k-values x 4 bit per column), wave w moves and reads only columns 32w..;
//   sz [4 waves][64 dwords]: the step's group constants of the wave's 32 columns, by 4-byte LDS-DMA.
// Per k-step a lane (column r16 of a block, k-group kg) reads ONE ds_read_b64 of W per column block = its 16 k-values
// (k = 16 kg + 8 ks + t for MFMA ks), so the matching activation chunk of row r is chunk 2 kg + ks of the 128-B row.
// 64.5 KiB of LDS per workgroup: two workgroups per CU hide each other's barriers.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace sglm {
namespace {

typedef __attribute__((ext_vector_type(2))) _Float16 h2;

struct AwqTArgs {
  const uint8_t* x;    // fp16 [M][K]
  int64_t x_sm;        // bytes between rows
  const uint8_t* wp;   // packed weights, Kp / 2 bytes per column
  const uint32_t* sz;  // [N][ngroups]
  const void* bias;    // fp16 [N] or null
  void* out;           // fp16 [M][N]
  int M, N, Kp;        // Kp: padded K (multiple of 512)
  int real_steps;      // K / 64: activation k-steps that exist (later ones multiply zero weights)
  int gshift;          // log2(G / 64): group of k-step s is s >> gshift
  int ngroups;         // Kp / G
};

// 4-byte LDS-DMA piece: lane i's dword lands at lds_addr + 4 i
__device__ __forceinline__ void lds_dma4(const void* gsrc, uint32_t lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}

// 8 nibbles of one dword -> 8 fp16 in MFMA operand order, exact reference arithmetic (see awq_packed.hip dequant8)
__device__ __forceinline__ f16x8 unpack8(uint32_t w, h2 zb2, h2 sc2) {
  f16x8 r;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    h2 h = __builtin_bit_cast(h2, ((w >> (4 * t)) & 0x000F000Fu) | 0x64006400u);
    h = (h - zb2) * sc2;
    r[2 * t] = h[0];
    r[2 * t + 1] = h[1];
  }
  return r;
}

constexpr int kTN = 128;
constexpr int kOpW = kTN * 32;            // 4 KiB
constexpr int kOpS = 4 * 256;             // 1 KiB: 64 dwords per wave

// RI = 16-row blocks per tile: 8 (128 rows) or 4 (64 rows, for shapes whose 128-row tiles would not fill the chip)
template <int kStagesT, int RI>
__global__ __launch_bounds__(256, 2) void awq_tiled_kernel(AwqTArgs p) {
  constexpr int kTM = 16 * RI;
  constexpr int kOpA = kTM * 128;           // 16 / 8 KiB
  constexpr int kStageT = kOpA + kOpW + kOpS;
  constexpr int UA = RI / 2;                // 1-KiB A pieces per wave and stage
  constexpr int kOpsPerStage = UA + 1 + 1;  // DMA instructions per wave and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kg = lane >> 4;

  // tile order: consecutive workgroups of one XCD (ids = same value mod 8) walk the M tiles of one N tile, so the packed
  // weight columns of a tile are fetched into that XCD's L2 once
  const int tiles_m = (p.M + kTM - 1) / kTM, tiles_n = (p.N + kTN - 1) / kTN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * kTM, n0 = tn * kTN;

  // ---- DMA sources of this lane
  const uint8_t* a_src[UA];
#pragma unroll
  for (int u = 0; u < UA; ++u) {
    const int row = (UA * wave + u) * 8 + (lane >> 3);
    const int j = (lane & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;  // rows past the edge re-read a valid row; never stored
    a_src[u] = p.x + (int64_t)m * p.x_sm + 16 * j;
  }
  // W (fragment-major, awq_packed.hip wp_index): the wave's 32 columns are two 16-column blocks; of a block's KiB per
  // 128-k step this 64-k stage takes k-groups 2 half, 2 half + 1 = 512 contiguous bytes.  Lane = (block lane >> 5,
  // k-group (lane >> 4) & 1, column lane & 15); the LDS image is that order verbatim.
  const int nblocks = (p.N + 15) >> 4;
  int wnb = (n0 >> 4) + 2 * wave + (lane >> 5);
  wnb = wnb < nblocks ? wnb : nblocks - 1;
  const uint8_t* w_src = p.wp + (int64_t)wnb * (p.Kp >> 7) * 1024 + ((lane >> 4) & 1) * 256 + (lane & 15) * 16;
  int snb = (n0 >> 4) + 2 * wave + ((lane & 31) >> 4);  // sz: lanes 32..63 repeat lanes 0..31
  snb = snb < nblocks ? snb : nblocks - 1;
  const uint32_t* s_src = p.sz + (int64_t)snb * p.ngroups * 16 + (lane & 15);

  const uint32_t smem_base = lds_addr_of(smem);
  auto dma_stage = [&](int stage, int kt) __attribute__((always_inline)) {
    const uint32_t dst = smem_base + stage * kStageT;
    const int akt = kt < p.real_steps ? kt : p.real_steps - 1;  // padded steps: finite activations x zero weights
#pragma unroll
    for (int u = 0; u < UA; ++u) lds_dma16(a_src[u] + (int64_t)akt * 128, dst + (UA * wave + u) * 1024);
    lds_dma16(w_src + (int64_t)(kt >> 1) * 1024 + (kt & 1) * 512, dst + kOpA + wave * 1024);
    lds_dma4(s_src + (kt >> p.gshift) * 16, dst + kOpA + kOpW + wave * 256);
  };

  f32x4 acc[RI][2];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment offsets: activation row (16 i + r16), chunk 2 kg + ks, swizzled; weight column 16 cb + r16 of the wave
  const int sw = (r16 >> 1) & 7;
  const uint32_t ca0 = r16 * 128 + 16 * ((2 * kg) ^ sw), ca1 = r16 * 128 + 16 * ((2 * kg + 1) ^ sw);
  const uint32_t wo = kOpA + wave * 1024 + (kg >> 1) * 256 + r16 * 16 + 8 * (kg & 1);  // image [block][k-group pair][column][16 B]
  const uint32_t so = kOpA + kOpW + wave * 256 + r16 * 4;

  const int nk = p.Kp >> 6;
#pragma unroll
  for (int st = 0; st < kStagesT - 1; ++st)
    if (st < nk) dma_stage(st, st);
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's pieces of stage kt have landed once at most the younger stage's are outstanding
    // (kStagesT - 2 younger stages stay in flight; the tail drains)
    const int younger = (nk - 1 - kt) < (kStagesT - 2) ? (nk - 1 - kt) : (kStagesT - 2);
    switch (younger) {
      case 0: wait_vmcnt<0>(); break;
      case 1: wait_vmcnt<kOpsPerStage>(); break;
      case 2: wait_vmcnt<2 * kOpsPerStage>(); break;
      case 3: wait_vmcnt<3 * kOpsPerStage>(); break;
      default: wait_vmcnt<4 * kOpsPerStage>(); break;
    }
    __syncthreads();  // everyone's pieces landed; everyone is done reading the stage refilled next
    if (kt + kStagesT - 1 < nk) dma_stage((kt + kStagesT - 1) % kStagesT, kt + kStagesT - 1);
    const char* st_ = smem + (kt % kStagesT) * kStageT;

    f16x8 bf[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const uint2 w2 = *reinterpret_cast<const uint2*>(st_ + wo + cb * 512);
      const uint32_t szv = *reinterpret_cast<const uint32_t*>(st_ + so + cb * 64);
      const h2 sc2 = __builtin_bit_cast(h2, (szv & 0xFFFFu) | (szv << 16));
      const h2 zb2 = __builtin_bit_cast(h2, (szv >> 16) | (szv & 0xFFFF0000u));
      bf[cb][0] = unpack8(w2.x, zb2, sc2);
      bf[cb][1] = unpack8(w2.y, zb2, sc2);
    }
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const f16x8 a0 = *reinterpret_cast<const f16x8*>(st_ + i * 2048 + ca0);
      const f16x8 a1 = *reinterpret_cast<const f16x8*>(st_ + i * 2048 + ca1);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bf[cb][0], acc[i][cb], 0, 0, 0);
        acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bf[cb][1], acc[i][cb], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // all stages dead: the epilogue reuses the memory

  // ---- epilogue through a wave-private [rows][32] (+8 pad) fp16 patch: 16-B row segments to global
  _Float16* ep = reinterpret_cast<_Float16*>(smem) + wave * (kTM * 40);
  float bv[2] = {0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = n0 + 32 * wave + 16 * cb + r16;
      bv[cb] = (float)reinterpret_cast<const _Float16*>(p.bias)[n < p.N ? n : p.N - 1];
    }
  }
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(16 * i + 4 * kg + r) * 40 + 16 * cb + r16] = (_Float16)(acc[i][cb][r] + bv[cb]);
  wait_lgkmcnt0();  // wave-private patch
#pragma unroll
  for (int it = 0; it < RI; ++it) {
    const int c = lane + 64 * it;  // 4 segments per row: row c >> 2, 8-column piece c & 3
    const int ml = c >> 2, nl = (c & 3) * 8;
    const int m = m0 + ml, n = n0 + 32 * wave + nl;
    if (m < p.M && n < p.N)  // N % 8 == 0: a piece is all in or all out
      *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(p.out) + (int64_t)m * p.N + n) =
          *reinterpret_cast<const uint4*>(ep + ml * 40 + nl);
  }
}

}  // namespace
}  // namespace sglm

using namespace sglm;

extern "C" int64_t sgl_mi355_awq_packed_k(int64_t K);

extern "C" int sgl_mi355_awq_gemm_packed_tiled(const void* x, const uint32_t* wp, const uint32_t* sz, const void* bias,
                                               void* out, int64_t M, int64_t N, int64_t K, int64_t group_size,
                                               int64_t x_stride_m, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_FP16, "awq_gemm_packed_tiled: fp16 only");
  SGLM_CHECK_ARG(M >= 0 && M < (1ll << 31) && N > 0 && N % 8 == 0 && K > 0 && K % 128 == 0,
                 "awq_gemm_packed_tiled: N %% 8 == 0 and K %% 128 == 0 required (M=%ld K=%ld N=%ld)", (long)M, (long)K, (long)N);
  SGLM_CHECK_ARG(group_size >= 128 && (group_size & (group_size - 1)) == 0 && K % group_size == 0,
                 "awq_gemm_packed_tiled: group_size must be a power of two >= 128 dividing K (got %ld)", (long)group_size);
  if (M == 0) return 0;
  SGLM_CHECK_ARG(x && wp && sz && out, "awq_gemm_packed_tiled: null tensor pointer");
  SGLM_CHECK_ARG(reinterpret_cast<uintptr_t>(x) % 16 == 0 && (x_stride_m * 2) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0,
                 "awq_gemm_packed_tiled: x rows and out must be 16-byte aligned");
  const int64_t Kp = sgl_mi355_awq_packed_k(K);
  const int64_t ngp = (Kp + group_size - 1) / group_size;
  AwqTArgs p{(const uint8_t*)x, x_stride_m * 2, (const uint8_t*)wp, sz, bias, out, (int)M, (int)N, (int)Kp, (int)(K / 64),
             0, (int)ngp};
  for (int64_t g = group_size / 64; g > 1; g >>= 1) ++p.gshift;
  // Tile height and pipeline depth by tile count (same-box A/B, Llama-2-7B shapes, M = 512 / 1024 / 4096): 128-row tiles with
  // 2 stages (43 KiB: three workgroups per CU) where they fill the chip three times over, 3 stages (64.5 KiB, two per CU)
  // otherwise; 64-row tiles when even that leaves CUs idle.  SGL_MI355_AWQ_TILED="stages,rows" overrides (tuning aid).
  const int64_t tn_ = (N + kTN - 1) / kTN;
  const int64_t tiles128 = ((M + 127) / 128) * tn_;
  int nst = tiles128 > 512 ? 2 : 3, rows = tiles128 >= 256 ? 128 : 64;
  {
    static const char* force = getenv("SGL_MI355_AWQ_TILED");
    int fs = 0, fr = 0;
    if (force && sscanf(force, "%d,%d", &fs, &fr) == 2 && (fs == 2 || fs == 3) && (fr == 64 || fr == 128)) nst = fs, rows = fr;
  }
  const int64_t tiles = ((M + rows - 1) / rows) * tn_;
  SGLM_CHECK_ARG(tiles < (1ll << 31), "awq_gemm_packed_tiled: too many tiles");
#define AWQ_T(NST, RI_)                                                                                                 \
  do {                                                                                                                  \
    constexpr int lds = NST * (16 * RI_ * 128 + kOpW + kOpS);                                                           \
    static int attr_rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(awq_tiled_kernel<NST, RI_>),       \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds),                \
                                   "hipFuncSetAttribute");                                                              \
    if (attr_rc) return attr_rc;                                                                                        \
    hipLaunchKernelGGL((awq_tiled_kernel<NST, RI_>), dim3((unsigned)tiles), dim3(256), lds, as_stream(stream), p);      \
  } while (0)
  if (rows == 128) { if (nst == 2) AWQ_T(2, 8); else AWQ_T(3, 8); }
  else { if (nst == 2) AWQ_T(2, 4); else AWQ_T(3, 4); }
#undef AWQ_T
  return check_hip(hipGetLastError(), "awq_tiled launch");
}
