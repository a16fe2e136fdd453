// 16-bit weight-streaming GEMM for decode-sized batches: out[M, N] = x[M, K] @ W[N, K]^T (+ bias), M <= 128
// (up to 64 rows: MB <= 4 row blocks; 65..128: MB = 8 with phases of at most four k-steps, round 3).
//
// Replaces: the LM-head matmul of LogitsProcessor._get_logits -- `torch.matmul(hidden_states, lm_head.weight.T)`
//   (python/sglang/srt/layers/logits_processor.py:430-505) -- which at Llama-3-8B is a 4096 x 128256 bf16 matrix:
//   1.05 GB read per decode step, as much as five decoder layers' weights.  (Also any unquantised decode linear.)
//
// HBM-bound (AI = 2 M / 2 B = 64 flop/B at M = 64): the structure is the FP8 weight streamer's (gemm_fp8.hip
// fp8_gemm_wstream_kernel) with 16-bit operands:
//   * a consumer wave owns 16 output columns over all of K; its weights go HBM -> registers, two 128-B row segments
//     (64 k) ahead, hand-counted vmcnt (the loads are inline asm: hipcc would drain the queue at every use);
//   * the activations of a phase (PH k-steps of 64, <= 64 KiB) are put into LDS by a producer wave with LDS-DMA into an
//     XOR-swizzled [step][row][128 B] image, double-buffered, one workgroup barrier per phase;
//   * v_mfma_f32_16x16x32_{bf16,f16}: lane group g takes 16-B chunks g and 4 + g of a 128-B segment for both operands
//     (whole 64-B sectors per load instruction); fp32 accumulation, one rounding at the end (+ bias in fp32).
#include <stdlib.h>

#include "common.h"

namespace sglm {
namespace {

struct G16Args {
  const uint8_t* a;  // activations [M][K] 16-bit
  int64_t a_sm;      // bytes between rows
  const uint8_t* b;  // weights [N][K] 16-bit (K contiguous)
  int64_t b_sn;      // bytes between weight rows
  const void* bias;  // [N] 16-bit or null
  void* out;         // [M][N] 16-bit
  int M, N, KB;      // KB = K * 2: contraction length in BYTES (128 per k-step)
  int throttle;      // producer: k-steps of activation DMA allowed in flight (0 = the whole phase at once)
  float* slabs;      // split-K form: fp32 partials [slices][M][N] (SLAB kernels), summed by gemm16_finalize_kernel
  int pps;           // ... phases per K slice
  int b_shuf;        // weights are FRAGMENT-MAJOR (round 3): the 2 KiB of a (16-column block, 64-k step) stored as
                     // [half][lane group g][row r16][16 B] -- the byte layout of the pre-shuffled FP8 weights
                     // (include/sgl_mi355.h), so a wave's load instruction covers one contiguous KiB
};

union Seg32 {  // this lane's 2 x 16 B of a 128-B row segment = two MFMA operands
  uint4 v[2];
  i32x4 x[2];
};

template <bool SHUF>
__device__ __forceinline__ void wload(Seg32& f, const uint8_t* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f.x[0]) : "v"(voff), "s"(sbase) : "memory");
  if constexpr (SHUF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(f.x[1]) : "v"(voff), "s"(sbase) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(f.x[1]) : "v"(voff), "s"(sbase) : "memory");
}
template <bool SHUF>
__device__ __forceinline__ void wload_nt(Seg32& f, const uint8_t* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(f.x[0]) : "v"(voff), "s"(sbase) : "memory");
  if constexpr (SHUF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 nt" : "=v"(f.x[1]) : "v"(voff), "s"(sbase) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2 offset:64 nt" : "=v"(f.x[1]) : "v"(voff), "s"(sbase) : "memory");
}
template <int N, int NB>
__device__ __forceinline__ void wait_segs(Seg32 (&f)[NB]) {
  static_assert(NB == 1 || NB == 2, "NB");
  if constexpr (NB == 1)
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(f[0].x[0]), "+v"(f[0].x[1]) : "n"(N) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(f[0].x[0]), "+v"(f[0].x[1]), "+v"(f[1].x[0]), "+v"(f[1].x[1]) : "n"(N) : "memory");
}
// the never-consumed tail refills: wait, THEN keep every queue register alive across the wait (gemm_fp8.hip 4.3.1 g)
template <int PB, int NB>
__device__ __forceinline__ void drain_segs(Seg32 (&q)[PB][NB]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < PB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) asm volatile("" ::"v"(q[i][j].x[0]), "v"(q[i][j].x[1]));
}

// SLAB (round 3): K is split over blockIdx.y in whole phases; a workgroup writes the fp32 partial of its slice and
// gemm16_finalize_kernel sums the slices in order, adds the bias and rounds once -- for narrow N (o_proj, down_proj, qkv of
// the bf16 config), where one workgroup per 4-8 column blocks over all of K leaves most CUs idle.
template <int DTYPE, int MB, int PH, bool NT, int NB, int PB = 2, bool SHUF = false, bool SLAB = false>
__global__ __launch_bounds__(576) void gemm16_wstream_kernel(G16Args p) {
  static_assert(!SLAB || NB == 1, "the split-K form is built for one column block per wave");
  static_assert(PH * MB <= 32, "one A buffer is at most 64 KiB");
  using H = Half16<DTYPE>;
  using T = typename H::T;
  using x8 = typename H::x8;
  constexpr int ROWS = 16 * MB;
  constexpr int STEP_BYTES = ROWS * 128;
  constexpr int BUF_BYTES = PH * STEP_BYTES;
  // PB: weight k-steps in flight per wave (2: gemm_fp8.hip 4.3.1 l found deeper queues slower there)
  constexpr int UPS = 2 * MB;     // 1-KiB DMA pieces (8 rows x 128 B) per k-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NC = (int)(blockDim.x >> 6) - 1;  // consumer waves; wave NC is the DMA producer
  const int r16 = lane & 15, g = lane >> 4;
  const int P_total = (p.KB >> 7) / PH;
  const int ph0 = SLAB ? (int)blockIdx.y * p.pps : 0;
  const int ph1 = SLAB ? (ph0 + p.pps < P_total ? ph0 + p.pps : P_total) : P_total;
  const int nph = ph1 - ph0;
  // NB 16-column blocks per consumer wave (NB = 2 at 64 rows: the activation image, re-read from L2 by every workgroup,
  // is then amortised over twice the weight bytes -- at M = 64 it is half as many bytes per k-step as 8 waves' weights)
  const int nb0 = (blockIdx.x * NC + wave) * NB;
  const uint32_t smem_base = lds_addr_of(smem);

  f32x4 acc[NB][MB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[j][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (wave == NC) {
    // producer: lane i of a piece lands at chunk i & 7 of row i >> 3 and fetches source chunk (i & 7) ^ ((row >> 1) & 7)
    const uint8_t* a_lane[UPS];
#pragma unroll
    for (int rg = 0; rg < UPS; ++rg) {
      const int drow = rg * 8 + (lane >> 3);
      const int dj = (lane & 7) ^ ((drow >> 1) & 7);
      a_lane[rg] = p.a + (int64_t)(drow < p.M ? drow : p.M - 1) * p.a_sm + 16 * dj;  // rows past M: never stored
    }
    auto dma_phase = [&](int ph, int buf) __attribute__((always_inline)) {
      for (int sl = 0; sl < PH; ++sl) {
#pragma unroll
        for (int rg = 0; rg < UPS; ++rg)
          lds_dma16(a_lane[rg] + ((ph * PH + sl) << 7), smem_base + buf * BUF_BYTES + sl * STEP_BYTES + rg * 1024);
        // pace the image: a whole phase (64 KiB) issued at once crowds the consumers' weight loads out of the CU's
        // memory pipe for microseconds; it is not needed before the NEXT phase barrier
        if (p.throttle == 1) wait_vmcnt<UPS>();
        else if (p.throttle == 2) wait_vmcnt<(2 * UPS <= 63 ? 2 * UPS : 63)>();
        else if (p.throttle == 4) wait_vmcnt<(4 * UPS <= 63 ? 4 * UPS : 63)>();
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

  // consumers
  uint32_t lane_off[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = (nb0 + j) * 16 + r16;
    if constexpr (SHUF)  // block nb starts at nb * (K / 64) * 2 KiB; the lane's 16 bytes of a KiB sit at lane * 16
      lane_off[j] = (uint32_t)((int64_t)((nb0 + j) * 16 < p.N ? nb0 + j : 0) * (p.KB >> 7) * 2048 + lane * 16);
    else
      lane_off[j] = (uint32_t)((int64_t)(n < p.N ? n : 0) * p.b_sn + 16 * g);
  }
  const int rot = (nb0 * 3) & (PH - 1);  // per-wave rotation of the sweep inside a phase
  const int last = ph1 * PH - 1;
  const int sw = (r16 >> 1) & 7;
  const uint32_t o0 = r16 * 128 + 16 * (g ^ sw);
  const uint32_t o1 = r16 * 128 + 16 * ((4 + g) ^ sw);

  int f_pf = ph0 * PH;
  auto refill = [&](Seg32 (&fr)[NB]) __attribute__((always_inline)) {
    const int f = f_pf < last ? f_pf : last;  // tail refills re-read the last step, never consumed
    const int ks = (f & ~(PH - 1)) + ((f + rot) & (PH - 1));
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const uint8_t* sb = p.b + ((int64_t)ks << (SHUF ? 11 : 7));
      if constexpr (NT) wload_nt<SHUF>(fr[j], sb, lane_off[j]);
      else wload<SHUF>(fr[j], sb, lane_off[j]);
    }
    ++f_pf;
  };
  Seg32 bq[PB][NB];
#pragma unroll
  for (int i = 0; i < PB; ++i) refill(bq[i]);

#pragma clang loop unroll(disable)  // exactly one copy of the step body (gemm_fp8.hip 4.3.1 f)
  for (int lp = 0; lp < nph; ++lp) {
    __syncthreads();  // barrier #lp: phase lp is in LDS
    const uint32_t abuf = (lp & 1) * BUF_BYTES;
    for (int s0 = 0; s0 < PH; s0 += PB) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int t = (s0 + i + rot) & (PH - 1);
        // the slot's 2 NB loads are the oldest in the queue: 2 NB (PB - 1) younger ones may stay in flight
        wait_segs<2 * NB * (PB - 1), NB>(bq[i]);
        const char* a0 = smem + abuf + t * STEP_BYTES + o0;
        const char* a1 = smem + abuf + t * STEP_BYTES + o1;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const x8 f0 = *reinterpret_cast<const x8*>(a0 + mb * 2048);
          const x8 f1 = *reinterpret_cast<const x8*>(a1 + mb * 2048);
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            acc[j][mb] = H::mfma16(f0, __builtin_bit_cast(x8, bq[i][j].v[0]), acc[j][mb]);
            acc[j][mb] = H::mfma16(f1, __builtin_bit_cast(x8, bq[i][j].v[1]), acc[j][mb]);
          }
        }
        refill(bq[i]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  drain_segs(bq);
  __syncthreads();  // the A buffers are dead: reuse their memory
  if constexpr (SLAB) {
    float* ep = reinterpret_cast<float*>(smem) + wave * (ROWS * 20);  // [ROWS][16] (+4 pad)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(16 * mb + 4 * g + r) * 20 + r16] = acc[0][mb][r];
    wait_lgkmcnt0();
    float* dst = p.slabs + (int64_t)blockIdx.y * p.M * p.N;
    for (int c = lane; c < ROWS * 4; c += 64) {
      const int m = c >> 2, q = c & 3;
      const int nn = nb0 * 16 + q * 4;
      if (m < p.M && nn < p.N)
        *reinterpret_cast<f32x4*>(dst + (int64_t)m * p.N + nn) = *reinterpret_cast<const f32x4*>(ep + m * 20 + q * 4);
    }
    return;
  }
  T* ep = reinterpret_cast<T*>(smem) + wave * (ROWS * 24);  // [ROWS][16] (+8 pad), one column block at a time
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int nb = nb0 + j;
    const int n = nb * 16 + r16;
    const float bv = p.bias ? H::to_f32(reinterpret_cast<const T*>(p.bias)[n < p.N ? n : p.N - 1]) : 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(16 * mb + 4 * g + r) * 24 + r16] = H::from_f32(acc[j][mb][r] + bv);
    wait_lgkmcnt0();
    for (int c = lane; c < ROWS * 2; c += 64) {
      const int m = c >> 1, half = c & 1;
      const int nn = nb * 16 + half * 8;
      if (m < p.M && nn < p.N)
        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.N + nn) =
            *reinterpret_cast<const uint4*>(ep + m * 24 + half * 8);
    }
    wait_lgkmcnt0();  // the patch is rewritten by the next column block
  }
}

// sum of the K slices (in slice order) + bias, one rounding: the epilogue of the split-K form
template <int DTYPE>
__global__ __launch_bounds__(64) void gemm16_finalize_kernel(G16Args p, int SK) {
  using H = Half16<DTYPE>;
  using T = typename H::T;
  const int64_t total = (int64_t)p.M * p.N / 8;
  const int64_t sstride = (int64_t)p.M * p.N;
  for (int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 64) {
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s0 = 0; s0 < SK; s0 += 4) {  // four slices in flight
      f32x4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* src = p.slabs + (int64_t)(s0 + u < SK ? s0 + u : SK - 1) * sstride + i * 8;
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
    typename H::x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float r = v[j];
      if (p.bias) r += H::to_f32(reinterpret_cast<const T*>(p.bias)[nn + j]);
      o[j] = H::from_f32(r);
    }
    *reinterpret_cast<typename H::x8*>(reinterpret_cast<T*>(p.out) + e) = o;
  }
}

template <int DTYPE, int MB, int PH, bool NT, int NB, int PB, bool SHUF>
int launch16_k(const G16Args& p, int nc, int groups, hipStream_t s) {
  auto kern = gemm16_wstream_kernel<DTYPE, MB, PH, NT, NB, PB, SHUF>;
  constexpr int lds = 2 * PH * 16 * MB * 128;
  static int attr_rc = check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
      "hipFuncSetAttribute");
  if (attr_rc) return attr_rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(64 * (nc + 1)), lds, s, p);
  return check_hip(hipGetLastError(), "gemm16_wstream launch");
}

template <int DTYPE, int MB, int PH, bool NT, int NB>
int launch16_ph(const G16Args& p, int nc, int groups, hipStream_t s) {
  static const int pb_env = [] { const char* e = getenv("SGL_MI355_GEMM16_PB"); return e ? atoi(e) : 0; }();  // A/B aid
  if constexpr (NB == 1) {
    if (p.b_shuf) {  // fragment-major weights: contiguous 1-KiB loads
      if constexpr (PH >= 4) {
        if (pb_env == 4) return launch16_k<DTYPE, MB, PH, NT, 1, 4, true>(p, nc, groups, s);
      }
      return launch16_k<DTYPE, MB, PH, NT, 1, 2, true>(p, nc, groups, s);
    }
  }
  if constexpr (PH >= 4 && NB == 1 && !NT) {
    if (pb_env == 4) return launch16_k<DTYPE, MB, PH, NT, NB, 4, false>(p, nc, groups, s);
  }
  return launch16_k<DTYPE, MB, PH, NT, NB, 2, false>(p, nc, groups, s);
}

// Split-K form (pre-shuffled weights): (phase length, consumer waves, K slices) as gemm_fp8.hip launch_wstream picks them --
// one workgroup per CU, fewest k-steps on the busiest CU -- then the finalize kernel.  `used` = false: the shape has no
// split-K form (K / 64 not a multiple of a phase, or the workspace is too small); nothing was launched.
// set by sgl_mi355_gemm16_nt_wshuffled_partials around its call: leave the slices unsummed and report their count
thread_local int32_t* tl_g16_slices = nullptr;

template <int DTYPE, int MB, int PH, bool NT>
int launch16_slab_k(const G16Args& p, int nc, int groups, int SK, hipStream_t s) {
  auto kern = gemm16_wstream_kernel<DTYPE, MB, PH, NT, 1, 2, true, true>;
  constexpr int stage = 2 * PH * 16 * MB * 128;
  constexpr int patch = 8 * 16 * MB * 20 * 4;
  constexpr int lds = stage > patch ? stage : patch;
  static int attr_rc = check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
      "hipFuncSetAttribute");
  if (attr_rc) return attr_rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)groups, (unsigned)SK), dim3(64 * (nc + 1)), lds, s, p);
  int rc = check_hip(hipGetLastError(), "gemm16_wstream (split-K) launch");
  if (rc) return rc;
  if (tl_g16_slices != nullptr) {  // the consumer runs the epilogue (sum in slice order, + bias, one rounding)
    *tl_g16_slices = SK;
    return 0;
  }
  const int64_t total = (int64_t)p.M * p.N / 8;
  hipLaunchKernelGGL((gemm16_finalize_kernel<DTYPE>), dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, p, SK);
  return check_hip(hipGetLastError(), "gemm16_finalize launch");
}

template <int DTYPE, int MB>
int launch16_splitk(G16Args p, float* slabs, int64_t slab_floats, hipStream_t s, bool& used) {
  used = false;
  if (!p.b_shuf || slabs == nullptr || (p.N & 7) != 0) return 0;
  const int steps = p.KB >> 7;
  const int nblocks = (p.N + 15) / 16;
  int PH = 0, nc = 8, SK = 1, pps = 0, best = 1 << 30;
  for (int ph = 8; ph >= 4; ph >>= 1) {
    if (ph * MB > 32 || steps % ph != 0) continue;
    const int P = steps / ph;
    for (int c = 8; c >= 4; --c) {
      const int groups_c = (nblocks + c - 1) / c;
      int sk_max = 256 / groups_c;
      if (sk_max < 1) sk_max = 1;
      if (sk_max > P) sk_max = P;
      const int pp = (P + sk_max - 1) / sk_max;
      const int sk = (P + pp - 1) / pp;
      const int rounds = (groups_c * sk + 255) / 256;
      const int cost = rounds * c * pp * (ph + 4);
      if (cost < best) { best = cost; PH = ph; nc = c; SK = sk; pps = pp; }
    }
  }
  if (PH == 0 || SK < 2 || slab_floats < (int64_t)SK * p.M * p.N) return 0;
  p.slabs = slabs;
  p.pps = pps;
  const int groups = (nblocks + nc - 1) / nc;
  // non-temporal weight loads: ahead for up to 32 rows and for long K at 64 (Llama-3-8B, M = 1 / 64, us, nt vs plain: o 10.6 /
  // 16.1 vs 11.6 / 14.8, qkv 12.6 / 16.9 vs 13.4 / 16.6, down 24.9 / 32.1 vs 26.2 / 33.1; profiles/r03_linear16_shapes_splitk.txt)
  static const int nt_env = [] { const char* e = getenv("SGL_MI355_GEMM16_SLAB_NT"); return e ? atoi(e) : -1; }();  // A/B aid
  const int nt = nt_env >= 0 ? nt_env : ((MB < 4 || steps >= 128) ? 1 : 0);
  used = true;
  if (PH == 8) {
    if constexpr (MB <= 4) return nt ? launch16_slab_k<DTYPE, MB, 8, true>(p, nc, groups, SK, s) : launch16_slab_k<DTYPE, MB, 8, false>(p, nc, groups, SK, s);
  }
  return nt ? launch16_slab_k<DTYPE, MB, 4, true>(p, nc, groups, SK, s) : launch16_slab_k<DTYPE, MB, 4, false>(p, nc, groups, SK, s);
}

template <int DTYPE, int MB>
int launch16(const G16Args& p, hipStream_t s) {
  const int steps = p.KB >> 7;
  // two column blocks per wave at 64 rows and wide N (SGL_MI355_GEMM16_NB=1|2: A/B aid)
  static const int nb_env = [] { const char* e = getenv("SGL_MI355_GEMM16_NB"); return e ? atoi(e) : 0; }();
  constexpr bool kCanNB2 = MB == 4;
  const bool nb2 = kCanNB2 && nb_env == 2 && !p.b_shuf;  // measured (LM head, M = 64): 243 us vs 218 with one block per wave -- off by default
  const int nblocks = ((p.N + 15) / 16 + (nb2 ? 1 : 0)) / (nb2 ? 2 : 1);  // units of NB column blocks
  // Phase length.  At 64 rows FOUR k-steps (2 x 32 KiB of LDS: two workgroups per CU) beat eight (one workgroup per CU)
  // and two: LM head 4096 x 128256, same box, M = 64: PH 8 / 4 / 2 = 234 / 218 / 245 us (hipBLASLt 210).  Fewer rows take
  // the longest phase that divides K (M = 16: 180 us, hipBLASLt 181).  SGL_MI355_GEMM16_PH caps it (A/B aid).
  static const int ph_cap = [] { const char* e = getenv("SGL_MI355_GEMM16_PH"); return e ? atoi(e) : 32; }();
  int PH = 0;
  for (int ph = (MB == 4 ? 4 : 32 / MB); ph >= 2; ph >>= 1)
    if (steps % ph == 0 && ph <= ph_cap) { PH = ph; break; }
  if (PH == 0) {
    set_error("gemm16: K (%d elements) must be a multiple of 256", p.KB / 2);
    return SGL_MI355_ERR_INVALID_ARGUMENT;
  }
  // consumer waves: the count that leaves the fewest k-steps on the busiest CU (one workgroup per CU)
  int nc = 8, best = 1 << 30;
  for (int c = 8; c >= 4; --c) {
    const int groups_c = (nblocks + c - 1) / c;
    const int cost = ((groups_c + 255) / 256) * c;
    if (cost < best) { best = cost; nc = c; }
  }
  const int groups = (nblocks + nc - 1) / nc;
  // weight-load cache policy (SGL_MI355_GEMM16_NT=0|1: A/B aid)
  // (measured, LM head 4096 x 128256, M = 1 / 16 / 64: nt 188 / 196 / 249 us, default policy 175 / 180 / 231 us: unlike
  //  the LDS-DMA K/V stream of the decode kernel, register loads do not gain from the hint)
  // Round 3, fragment-major weights (contiguous 1-KiB loads): there the hint DOES pay -- LM head, M = 1 / 16 / 64: 155 / 159 /
  // 184 us with nt against 174 / 179 / 195 without (hipBLASLt 183-187 / 187-188 / 210; profiles/r03_lm_head_points.jsonl) --
  // as it does for the unsplit FP8 streamer (gemm_fp8.hip).  Default: on for pre-shuffled weights, off for row-major ones.
  static const int nt_env = [] { const char* e = getenv("SGL_MI355_GEMM16_NT"); return e ? atoi(e) : -1; }();
  const int nt = nt_env >= 0 ? nt_env : (p.b_shuf ? 1 : 0);
#define G16_GO(PH_)                                                                                   \
  do {                                                                                                \
    if constexpr (kCanNB2) {                                                                          \
      if (nb2) return nt ? launch16_ph<DTYPE, MB, PH_, true, 2>(p, nc, groups, s)                     \
                         : launch16_ph<DTYPE, MB, PH_, false, 2>(p, nc, groups, s);                   \
    }                                                                                                 \
    return nt ? launch16_ph<DTYPE, MB, PH_, true, 1>(p, nc, groups, s)                                \
              : launch16_ph<DTYPE, MB, PH_, false, 1>(p, nc, groups, s);                              \
  } while (0)
  if constexpr (MB == 1) { if (PH == 32) G16_GO(32); }
  if constexpr (MB <= 2) { if (PH == 16) G16_GO(16); }
  if (PH == 4) G16_GO(4);
  if (PH == 2) G16_GO(2);
  if constexpr (MB <= 4) G16_GO(8);
  set_error("gemm16: no phase length for K = %d elements at %d rows", p.KB / 2, p.M);
  return SGL_MI355_ERR_INVALID_ARGUMENT;
#undef G16_GO
}


// ------------------------------------------------------------------------------------------
// Tiled form for MORE than 128 rows (round 4): prefill of an unquantised model, LM head over many rows -- until now
// `F.linear` -> the vendor library (layers/quantization/unquant.py:111-123).  Fragment-major weights only (the copy the decode
// streamer reads: a wave's load instruction is one contiguous KiB).  The structure is gemm_fp8.hip's fp8_gemm_tiled3_kernel with
// 16-bit operands -- the byte geometry is the same (a k-step is 128 B of a row = 64 values, two v_mfma_f32_16x16x32 per
// fragment pair on 16-B chunks g and 4 + g of both operands):
//   * A goes global -> LDS by LDS-DMA, three stages, [row][128 B] image with the 16-B chunk index XOR-swizzled by
//     (row >> 1) & 7 on the source side; fragments by ds_read_b128;
//   * B never touches LDS: each wave streams the weight fragments of its own 64 columns global -> VGPR, every fragment
//     refilled one k-step ahead right behind its last MFMA, hand-counted vmcnt (loads retire in order);
//   * block tile (16 RI WM) x (64 WN): 128 x 256 with four waves and TWO workgroups per CU, 256 x 256 with eight waves, or
//     128 x 128 (RI 4, 2 x 2 waves) where the wider tiles would leave CUs without one;
//   * fp32 accumulation, + bias in fp32, one rounding; stores in 16-B row segments through a wave-private LDS patch;
//   * SLAB: K is cut over blockIdx.y into slices of p.pps k-steps, each workgroup writes the fp32 partial of its slice and
//     gemm16_finalize_kernel sums the slices in order, adds the bias and rounds once -- for the shapes whose tiles would
//     otherwise cover a fraction of the chip for a long K loop (o_proj / down_proj at a few hundred rows).
template <int DTYPE, int RI, int WM, int WN, bool SLAB = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm16_tiled_kernel(G16Args p, int raster_gn) {
  using H = Half16<DTYPE>;
  using T = typename H::T;
  using x8 = typename H::x8;
  constexpr int NSTAGE = 3;
  constexpr int CB = 4;                 // 16-column blocks per wave
  constexpr int NW = WM * WN;
  constexpr int TMB = 16 * RI * WM;     // block rows
  constexpr int TNB = 16 * CB * WN;     // block columns
  constexpr int STAGE = TMB * 128;      // A tile: TMB rows x 128 B
  constexpr int UA = TMB / 8 / NW;      // 1-KiB DMA units (8 rows x 128 B) per wave and stage
  static_assert((TMB / 8) % NW == 0, "DMA units must divide over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r16 = lane & 15, g = lane >> 4;

  const int tiles_m = (p.M + TMB - 1) / TMB, tiles_n = (p.N + TNB - 1) / TNB;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // consecutive workgroup ids of one XCD (ids congruent mod 8) take consecutive tiles
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn;
  {  // groups of raster_gn column tiles, row tile fastest across the group's columns: an XCD's resident tiles share both operands
    const int gsz = raster_gn * tiles_m;
    const int grp = bid / gsz, r = bid - grp * gsz;
    const int left = tiles_n - grp * raster_gn;
    const int gn = left < raster_gn ? left : raster_gn;
    tm = r / gn;
    tn = grp * raster_gn + (r - tm * gn);
  }
  const int m0 = tm * TMB, n0 = tn * TNB;

  const uint8_t* a_src[UA];
#pragma unroll
  for (int u = 0; u < UA; ++u) {
    const int row = (UA * wave + u) * 8 + (lane >> 3);  // tile-local row
    const int j = (lane & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;  // rows past the edge re-read a valid row; never stored
    a_src[u] = p.a + (int64_t)m * p.a_sm + 16 * j;
  }
  const uint32_t smem_base = lds_addr_of(smem);
  auto dma_stage = [&](int stage, int kt) __attribute__((always_inline)) {
    const uint32_t dst = smem_base + stage * STAGE;
#pragma unroll
    for (int u = 0; u < UA; ++u) lds_dma16(a_src[u] + (int64_t)kt * 128, dst + (UA * wave + u) * 1024);
  };
  // weight fragments: column block (n0 / 16 + 4 wn + j), k-step kt = 2 KiB at block * 16 KB + 2048 kt; lane i takes bytes
  // 16 i.. of each KiB.  Blocks past N re-read the last one (their columns are never stored).
  const uint8_t* b_blk[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    int nb = (n0 >> 4) + wn * CB + j;
    nb = nb < (p.N >> 4) ? nb : (p.N >> 4) - 1;
    b_blk[j] = p.b + (int64_t)nb * 16 * p.KB;
  }
  const int kt0 = SLAB ? (int)blockIdx.y * p.pps : 0;
  const int nk = SLAB ? ((p.KB >> 7) < kt0 + p.pps ? (p.KB >> 7) : kt0 + p.pps) : (p.KB >> 7);  // this workgroup: k-steps [kt0, nk)

  f32x4 acc[RI][CB];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A fragment offsets inside the stage: row (16 (RI wm + i) + r16), chunks g and 4 + g, swizzled as the DMA wrote them
  const int sw = (r16 >> 1) & 7;
  const uint32_t a_row = (wm * 16 * RI + r16) * 128;
  const uint32_t c0 = 16 * (g ^ sw), c1 = 16 * ((4 + g) ^ sw);

  // VMEM issue order per step: the A DMAs of step kt + 2 (UA per wave), then refills R_0..R_3 (two loads each), each behind
  // its column block's MFMAs.  In
  // front of column block j "R_j of the previous step has landed" is vmcnt(6) for j = 0 (R_1..R_3 of the previous step are
  // younger) and vmcnt(6 + UA) after it; the wait at the last column also proves this wave's DMAs of step kt + 1, which is
  // what the next barrier publishes.  Tail steps re-issue the last step's DMAs into the dead stage and re-read the last
  // weights, so the counts never change.
  Seg32 bq[CB];
  constexpr int YB = 2 * (CB - 1);
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) dma_stage(st, kt0 + st < nk ? kt0 + st : nk - 1);
  {
    const uint32_t voff = (uint32_t)lane * 16 + (uint32_t)kt0 * 2048;
#pragma unroll
    for (int j = 0; j < CB; ++j) wload<true>(bq[j], b_blk[j], voff);
  }
  wait_vmcnt<(NSTAGE - 2) * UA + 2 * CB>();  // stage 0 landed (this wave's part)
#pragma clang loop unroll(disable)
  for (int kt = kt0; kt < nk; ++kt) {
    __syncthreads();  // everyone's DMAs of stage kt landed; everyone finished reading the stage refilled below
    const char* sa_ = smem + ((kt - kt0) % NSTAGE) * STAGE + a_row;
    Seg32 af[RI];
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      af[i].v[0] = *reinterpret_cast<const uint4*>(sa_ + i * 2048 + c0);
      af[i].v[1] = *reinterpret_cast<const uint4*>(sa_ + i * 2048 + c1);
    }
    const uint32_t voff = (uint32_t)lane * 16 + (uint32_t)(kt + 1 < nk ? kt + 1 : nk - 1) * 2048;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      if (j == 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bq[j].x[0]), "+v"(bq[j].x[1]) : "n"(YB) : "memory");
      else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bq[j].x[0]), "+v"(bq[j].x[1]) : "n"(YB + UA) : "memory");
      const x8 b0 = __builtin_bit_cast(x8, bq[j].v[0]), b1 = __builtin_bit_cast(x8, bq[j].v[1]);
#pragma unroll
      for (int i = 0; i < RI; ++i) acc[i][j] = H::mfma16(__builtin_bit_cast(x8, af[i].v[0]), b0, acc[i][j]);
#pragma unroll
      for (int i = 0; i < RI; ++i) acc[i][j] = H::mfma16(__builtin_bit_cast(x8, af[i].v[1]), b1, acc[i][j]);
      if (j == 0) {
        const int kn = kt + NSTAGE - 1;
        dma_stage((kn - kt0) % NSTAGE, kn < nk ? kn : nk - 1);
      }
      wload<true>(bq[j], b_blk[j], voff);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the never-consumed tail refills and DMAs
#pragma unroll
  for (int j = 0; j < CB; ++j) asm volatile("" ::"v"(bq[j].x[0]), "v"(bq[j].x[1]));
  __syncthreads();  // all stages dead: the epilogue reuses the memory

  if constexpr (SLAB) {
    // fp32 partial of this K slice: 16 rows at a time through a wave-private [16][64] (+4 pad) patch, 16-B segments
    float* fp = reinterpret_cast<float*>(smem) + wave * (16 * 68);
    float* dst = p.slabs + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < RI; ++i) {
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) fp[(4 * g + r) * 68 + 16 * j + r16] = acc[i][j][r];
      wait_lgkmcnt0();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = lane + 64 * it;  // 256 segments of four floats
        const int ml = c >> 4, nl = (c & 15) * 4;
        const int m = m0 + wm * 16 * RI + 16 * i + ml, n = n0 + wn * 16 * CB + nl;
        if (m < p.M && n < p.N)  // N % 16 == 0: a segment is all-in or all-out
          *reinterpret_cast<f32x4*>(dst + (int64_t)m * p.N + n) = *reinterpret_cast<const f32x4*>(fp + ml * 68 + nl);
      }
      wait_lgkmcnt0();
    }
    return;
  }
  // epilogue: passes of 64 rows through a wave-private [64][64] (+8 pad) patch that turns the MFMA layout into 16-B segments
  T* ep = reinterpret_cast<T*>(smem) + wave * (64 * 72);
  constexpr int RP = RI < 4 ? RI : 4;
#pragma unroll
  for (int pass = 0; pass < RI / RP; ++pass) {
    const int mw0 = m0 + wm * 16 * RI + 16 * RP * pass;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int n = n0 + wn * 16 * CB + 16 * j + r16;
      const float bv = p.bias ? H::to_f32(reinterpret_cast<const T*>(p.bias)[n < p.N ? n : p.N - 1]) : 0.f;
#pragma unroll
      for (int i = 0; i < RP; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) ep[(16 * i + 4 * g + r) * 72 + 16 * j + r16] = H::from_f32(acc[RP * pass + i][j][r] + bv);
    }
    wait_lgkmcnt0();  // wave-private patch: a wave-level LDS wait is enough
    constexpr int SEG = 2 * CB;            // 16-byte row segments per patch row
    constexpr int ITEMS = 16 * RP * SEG;
#pragma unroll
    for (int it = 0; it < (ITEMS + 63) / 64; ++it) {
      const int c = lane + 64 * it;
      const int ml = c / SEG, nl = (c % SEG) * 8;
      const int m = mw0 + ml, n = n0 + wn * 16 * CB + nl;
      if (c < ITEMS && m < p.M && n < p.N)
        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.N + n) =
            *reinterpret_cast<const uint4*>(ep + ml * 72 + nl);
    }
    wait_lgkmcnt0();  // the patch is rewritten by the next pass
  }
}

template <int DTYPE, int RI, int WM, int WN, bool SLAB = false>
int launch16_tiled_k(const G16Args& p, int SK, hipStream_t s) {
  auto kern = gemm16_tiled_kernel<DTYPE, RI, WM, WN, SLAB>;
  constexpr int tmb = 16 * RI * WM, tnb = 64 * WN;
  constexpr int lds_st = 3 * tmb * 128, lds_ep = SLAB ? WM * WN * 16 * 68 * 4 : WM * WN * 64 * 72 * 2;
  constexpr int lds = lds_st > lds_ep ? lds_st : lds_ep;
  static int attr_rc = check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
      "hipFuncSetAttribute");
  if (attr_rc) return attr_rc;
  static const int gn_env = [] { const char* e = getenv("SGL_MI355_G16T_GN"); return e ? atoi(e) : 0; }();  // tuning aid
  const unsigned grid = (unsigned)(((p.M + tmb - 1) / tmb) * ((p.N + tnb - 1) / tnb));
  hipLaunchKernelGGL(kern, dim3(grid, (unsigned)SK), dim3(64 * WM * WN), lds, s, p, gn_env > 0 ? gn_env : 4);
  int rc = check_hip(hipGetLastError(), "gemm16_tiled_kernel launch");
  if (rc || !SLAB) return rc;
  const int64_t total = (int64_t)p.M * p.N / 8;
  const int64_t blocks = (total + 63) / 64;
  hipLaunchKernelGGL((gemm16_finalize_kernel<DTYPE>), dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(64), 0, s, p, SK);
  return check_hip(hipGetLastError(), "gemm16_finalize launch");
}

// Tile choice: 128 x 256 (four waves, two workgroups per CU) from 192 such tiles, else 128 x 128; where even those leave the
// chip part empty or one workgroup per CU for a long K loop (at most 128 tiles from 64 k-steps, at most 256 from 128) and
// the caller gave a workspace, K is cut into slices (about 512-768 workgroups in all, at least 16 k-steps each) + the finalize launch.
// A/B aids: SGL_MI355_G16T_TILE=1|2|3 forces 256x256 / 128x256 / 128x128; SGL_MI355_G16T_SK=n forces n slices (1: never split).
template <int DTYPE>
int launch16_tiled(G16Args p, float* slabs, int64_t slab_floats, hipStream_t s) {
  static const int tile_env = [] { const char* e = getenv("SGL_MI355_G16T_TILE"); return e ? atoi(e) : 0; }();
  static const int sk_env = [] { const char* e = getenv("SGL_MI355_G16T_SK"); return e ? atoi(e) : 0; }();
  const int64_t t128w = (int64_t)((p.M + 127) / 128) * ((p.N + 255) / 256);
  const int64_t t128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const int tile = tile_env >= 1 && tile_env <= 3 ? tile_env : (t128w >= 192 ? 2 : 3);
  const int steps = p.KB >> 7;
  if (tile == 3 && slabs != nullptr && sk_env != 1) {
    // (measured: with 64 k-steps a split pays up to 128 tiles -- o_proj at 256 / 512 rows 35.8 / 37.1 -> 22.0 / 33.1 us, but 192
    //  or 256 tiles 50.0 / 40.3 -> 61.7 / 47.1; from 128 k-steps up to 256 tiles: down_proj 14336 at 1024 rows 156 -> 146)
    const bool split = (t128 <= 128 && steps >= 64) || (t128 <= 256 && steps >= 128);
    int sk = sk_env > 1 ? sk_env : (split ? (int)(640 / t128) : 1);
    if (sk > steps / 16) sk = steps / 16;
    if (sk >= 2) {
      const int pps = (steps + sk - 1) / sk;
      sk = (steps + pps - 1) / pps;
      if (sk >= 2 && (int64_t)sk * p.M * p.N <= slab_floats) {
        p.slabs = slabs;
        p.pps = pps;
        return launch16_tiled_k<DTYPE, 4, 2, 2, true>(p, sk, s);
      }
    }
  }
  if (tile == 1) return launch16_tiled_k<DTYPE, 8, 2, 4>(p, 1, s);
  if (tile == 2) return launch16_tiled_k<DTYPE, 8, 1, 4>(p, 1, s);
  return launch16_tiled_k<DTYPE, 4, 2, 2>(p, 1, s);
}

}  // namespace
}  // namespace sglm

using namespace sglm;

static int gemm16_impl(const void* x, const void* weight, const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                       int64_t x_stride_m, int64_t w_stride_n, int b_shuf, int dtype, void* stream, float* workspace = nullptr,
                       int64_t workspace_floats = 0) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "gemm16_nt: dtype must be bfloat16 or float16");
  SGLM_CHECK_ARG(M >= 0 && (M <= 128 || b_shuf) && M < (1ll << 31),
                 "gemm16_nt: row-major weights go through the weight-streaming kernel, M <= 128 rows (got %ld); more rows need the "
                 "fragment-major weight (sgl_mi355_gemm16_nt_wshuffled)", (long)M);
  SGLM_CHECK_ARG(N > 0 && N % 8 == 0 && K > 0 && K % 256 == 0 && N < (1ll << 31) && K < (1ll << 30),
                 "gemm16_nt: N %% 8 == 0 and K %% 256 == 0 required (N=%ld K=%ld)", (long)N, (long)K);
  SGLM_CHECK_ARG(!b_shuf || N % 16 == 0, "gemm16_nt (pre-shuffled weight): N %% 16 == 0 required (N=%ld)", (long)N);
  SGLM_CHECK_ARG(x_stride_m >= K && w_stride_n >= K && x_stride_m % 8 == 0 && w_stride_n % 8 == 0,
                 "gemm16_nt: row strides must be >= K and multiples of 8 elements");
  SGLM_CHECK_ARG(N * w_stride_n * 2 < (1ll << 32), "gemm16_nt: weight larger than 4 GiB (32-bit lane offsets)");
  if (M == 0) return 0;
  SGLM_CHECK_ARG(x && weight && out, "gemm16_nt: null tensor pointer");
  SGLM_CHECK_ARG(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(weight) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(out) % 16 == 0, "gemm16_nt: operands must be 16-byte aligned");
  static const int throttle = [] { const char* e = getenv("SGL_MI355_GEMM16_THROTTLE"); return e ? atoi(e) : 0; }();  // A/B aid
  G16Args p{(const uint8_t*)x, x_stride_m * 2, (const uint8_t*)weight, w_stride_n * 2, bias, out, (int)M, (int)N, (int)(K * 2),
            throttle, nullptr, 0, b_shuf};
  hipStream_t s = as_stream(stream);
  if (M > 128) {  // prefill-sized batches: the tiled kernel
    SGLM_CHECK_ARG(M * N < (1ll << 40), "gemm16_nt: output too large");
    if (dtype == SGL_MI355_BF16) return launch16_tiled<SGL_MI355_BF16>(p, workspace, workspace_floats, s);
    return launch16_tiled<SGL_MI355_FP16>(p, workspace, workspace_floats, s);
  }
  if (workspace != nullptr) {  // split-K form where the unsplit grid would leave CUs idle (fewer than ~200 workgroups)
    bool used = false;
    int rc = 0;
    const bool narrow = (N + 15) / 16 < 8 * 200;
    if (narrow) {
#define G16_S(D)                                                                                   \
      rc = M <= 16   ? launch16_splitk<D, 1>(p, workspace, workspace_floats, s, used)              \
           : M <= 32 ? launch16_splitk<D, 2>(p, workspace, workspace_floats, s, used)              \
           : M <= 64 ? launch16_splitk<D, 4>(p, workspace, workspace_floats, s, used)              \
                     : launch16_splitk<D, 8>(p, workspace, workspace_floats, s, used)
      if (dtype == SGL_MI355_BF16) { G16_S(SGL_MI355_BF16); } else { G16_S(SGL_MI355_FP16); }
#undef G16_S
      if (rc || used) return rc;
    }
  }
  if (tl_g16_slices != nullptr) {
    set_error("gemm16_nt_wshuffled_partials: this shape has no split-K form (M=%ld N=%ld K=%ld)", (long)M, (long)N, (long)K);
    return SGL_MI355_ERR_UNSUPPORTED;  // nothing launched: the caller runs the GEMM that finishes itself
  }
#define G16_D(D)                                   \
  do {                                             \
    if (M <= 16) return launch16<D, 1>(p, s);      \
    if (M <= 32) return launch16<D, 2>(p, s);      \
    if (M <= 64) return launch16<D, 4>(p, s);      \
    return launch16<D, 8>(p, s);                   \
  } while (0)
  if (dtype == SGL_MI355_BF16) G16_D(SGL_MI355_BF16);
  G16_D(SGL_MI355_FP16);
#undef G16_D
}

extern "C" int sgl_mi355_gemm16_nt(const void* x, const void* weight, const void* bias, void* out, int64_t M, int64_t N,
                                   int64_t K, int64_t x_stride_m, int64_t w_stride_n, int dtype, void* stream) {
  return gemm16_impl(x, weight, bias, out, M, N, K, x_stride_m, w_stride_n, 0, dtype, stream);
}

// The same product on a weight re-laid by sgl_mi355_fp8_shuffle_weight applied to its BYTES ([N][2 K] bytes: the
// fragment-major layout is defined on 128-byte k-steps, i.e. 64 16-bit values): N % 16 == 0, K % 256 == 0.
extern "C" int sgl_mi355_gemm16_nt_wshuffled(const void* x, const void* weight_shuffled, const void* bias, void* out, int64_t M,
                                             int64_t N, int64_t K, int64_t x_stride_m, int dtype, void* stream) {
  return gemm16_impl(x, weight_shuffled, bias, out, M, N, K, x_stride_m, K, 1, dtype, stream);
}

// ... with an fp32 workspace for the split-K form: narrow N (fewer than ~200 workgroups of 8 column blocks) is cut along K into
// slices of whole phases, partial sums go to `workspace` (num_slices * M * N floats, at most 16 * M * N) and a finalize kernel
// sums them in slice order, adds the bias and rounds once.  Wide N runs the unsplit kernel as sgl_mi355_gemm16_nt_wshuffled.
// The split-K form WITHOUT its finalize launch: fp32 partial sums [num_slices][M][N] stay in `workspace` for a consumer that
// runs the epilogue itself -- the *_from_partials entry points of the FP8 path with unit scales (sum in slice order, + bias, one
// rounding: gemm16_finalize_kernel's arithmetic).  Decode sizes only (M <= 128); SGL_MI355_ERR_UNSUPPORTED, nothing launched,
// where the shape has no split-K form.
extern "C" int sgl_mi355_gemm16_nt_wshuffled_partials(const void* x, const void* weight_shuffled, float* workspace,
                                                      int64_t workspace_floats, int64_t M, int64_t N, int64_t K,
                                                      int64_t x_stride_m, int dtype, int32_t* num_slices, void* stream) {
  SGLM_CHECK_ARG(workspace != nullptr && workspace_floats > 0 && num_slices != nullptr,
                 "gemm16_nt_wshuffled_partials: null workspace / num_slices");
  SGLM_CHECK_ARG(M > 0 && M <= 128, "gemm16_nt_wshuffled_partials: decode sizes only (1..128 rows, got %ld)", (long)M);
  *num_slices = 0;
  tl_g16_slices = num_slices;
  // (`out` is not written in this form; the workspace stands in for the non-null / alignment checks)
  const int rc = gemm16_impl(x, weight_shuffled, nullptr, workspace, M, N, K, x_stride_m, K, 1, dtype, stream, workspace, workspace_floats);
  tl_g16_slices = nullptr;
  return rc;
}

extern "C" int sgl_mi355_gemm16_nt_wshuffled_splitk(const void* x, const void* weight_shuffled, const void* bias, void* out,
                                                    float* workspace, int64_t workspace_floats, int64_t M, int64_t N, int64_t K,
                                                    int64_t x_stride_m, int dtype, void* stream) {
  SGLM_CHECK_ARG(workspace != nullptr && workspace_floats > 0, "gemm16_nt_wshuffled_splitk: null workspace");
  return gemm16_impl(x, weight_shuffled, bias, out, M, N, K, x_stride_m, K, 1, dtype, stream, workspace, workspace_floats);
}
