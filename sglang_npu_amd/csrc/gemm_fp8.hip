// FP8 (OCP e4m3fn) w8a8 path for MI355X / gfx950: per-token dynamic activation quant and
// the rowwise-scaled GEMM.
//
// Replaces:
//   * sgl_per_token_quant_fp8(input, output_q, output_s)
//       sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:15-87,166-227
//       (scale = absmax/448; scale_inv = scale==0 ? 0 : 1/scale; q = cast(clamp(x*scale_inv)))
//   * fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias)
//       sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1071-1146, epilogue :498-546
//       D = (A B)_f32 * scale_b[n] * scale_a[m] (+ bias[n]) -> bf16/fp16
//   both reached from apply_fp8_linear, python/sglang/srt/layers/quantization/fp8_utils.py:653-704.
//
// Layouts: A [M,K] row-major e4m3; B is the reference's "column-major [K,N]" = W[N][K] K-major
// (w8a8_fp8.py:115,132 stores weight.t()), which is already the MFMA-friendly layout: both
// operands feed v_mfma_f32_16x16x32_fp8_fp8 with 8 contiguous K bytes per lane.
//
// Two regimes:
//   * M <= 64 (decode): HBM-bound weight streaming.  `skinny` kernel: no LDS for operands, every
//     wave streams its 16 weight rows straight into registers 4 k-steps ahead (full 128-B lines),
//     activations come from L2; the K range is split across the waves of a workgroup and reduced
//     through LDS, so even N = 4096 fills the chip without a split-K pass over HBM.
//   * M > 64 (prefill): MFMA-bound.  `tiled` kernel: 128x128x128 tiles, double-buffered LDS in a
//     fragment-major image ([k-chunk][row][32 B], conflict-free ds_read_b128), register-staged
//     prefetch of the next tile under the current tile's 64 MFMAs per wave, LDS-transposed
//     epilogue with 16-B stores.
// The MFMA k index is contracted, so both kernels hand lane group g the contiguous 32 B
// [32g, 32g+32) of each 128-wide k-step (four MFMAs' worth) -- the same permutation on A and B.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include "common.h"

namespace sglm {
namespace {

constexpr float kFp8Max = 448.0f;

// Which GEMM kernel family the last fp8_scaled_mm(_partials) call of this thread launched (sgl_mi355_fp8_last_kernel):
// "skinny", "oneshot", "astat", "astat_direct", "wstream", "wstream_slab", "tiled", "tiled2", "tiled3", "tiled3_silu".  A test aid: the
// dispatch table of run_gemm is long, and tests/test_fp8_gpu.py names a shape that reaches each family.
thread_local const char* g_last_kernel = "";

// ------------------------------------------------------------------------------------------
// per-token quant: one workgroup (256 threads) per row; two passes (second read is L2-hot)
template <int DTYPE, int NT = 256>
__global__ __launch_bounds__(NT) void per_token_quant_fp8_kernel(
    const typename Half16<DTYPE>::T* __restrict__ x, uint8_t* __restrict__ q, float* __restrict__ s, int K) {
  using H = Half16<DTYPE>;
  using x8 = typename H::x8;
  __shared__ float red[NT / 64];
  const int t = blockIdx.x;
  const x8* xr = reinterpret_cast<const x8*>(x + (int64_t)t * K);
  const int nv = K >> 3;
  float amax = 0.f;
  for (int i = threadIdx.x; i < nv; i += NT) {
    const x8 v = xr[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(H::to_f32(v[j])));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) amax = fmaxf(amax, red[w]);
  const float scale = amax / kFp8Max;
  if (threadIdx.x == 0) s[t] = scale;
  const float inv = scale == 0.f ? 0.f : 1.0f / scale;
  uint2* qr = reinterpret_cast<uint2*>(q + (int64_t)t * K);
  for (int i = threadIdx.x; i < nv; i += NT) {
    const x8 v = xr[i];
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(H::to_f32(v[j]) * inv, -kFp8Max), kFp8Max);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    qr[i] = uint2{(unsigned)lo, (unsigned)hi};
  }
}

// ------------------------------------------------------------------------------------------
struct GemmArgs {
  const uint8_t* a;
  int64_t a_sm;  // bytes between rows of A
  const uint8_t* b;
  int64_t b_sn;  // bytes between rows of W (= columns of B)
  const float* sa;
  const float* sb;
  const void* bias;
  void* out;
  int M, N, K;
  // 1: W is stored fragment-major ("pre-shuffled", sgl_mi355_fp8_shuffle_weight / include/sgl_mi355.h): for column block
  // nb = n / 16 and k-step ks = k / 128 one 2-KiB piece [half][g][r16][16 B] holding bytes 64 half + 16 g .. + 16 of row
  // 16 nb + r16 -- exactly the two 1-KiB load instructions of a decode wave, contiguous.  b_sn is unused then.
  int b_shuf = 0;
  // 16-bit activations quantised while they are staged (sgl_mi355_fp8_scaled_mm_partials_a16; `a` is unused then):
  // row m of a16 (a16_sm elements apart) is multiplied by 448 / a_absmax[m] and cast to e4m3 exactly as
  // sgl_per_token_quant_fp8 does; a_scale_out[m] = a_absmax[m] / 448 is written for the consumer's epilogue
  const void* a16 = nullptr;
  int64_t a16_sm = 0;
  const float* a_absmax = nullptr;
  float* a_scale_out = nullptr;
  int raster_gn = 1;  // tiled v3: column tiles per rasterisation group (see the kernel)
  // tiled v3, RAWK form (prefill split-K over workgroups, round 5): blockIdx.y = K slice of `k_steps_per_slice` 128-byte
  // k-steps; the raw fp32 accumulators go to slabs[slice][M][N] for a consumer that runs the epilogue (ops.GemmPartials)
  float* slabs = nullptr;
  int k_steps_per_slice = 0;
};

// Where a decode wave finds its weight fragments: scalar base of (column block, k-step 0), bytes between k-steps, and
// the two per-lane offsets of the step's load instructions (lane = 16 g + r16 holds row 16 nb + r16, bytes 16 g.. and
// 64 + 16 g.. of the step).  Row-major: 16 rows x 64 B per instruction; pre-shuffled: 1 KiB contiguous.
struct WFrag {
  int64_t base;
  int step;
  uint32_t v0, v1;
};
__device__ __forceinline__ WFrag wfrag_addr(const GemmArgs& p, int nb, int lane) {
  WFrag w;
  if (p.b_shuf) {
    w.base = (int64_t)nb * 16 * p.K;
    w.step = 2048;
    w.v0 = (uint32_t)lane * 16;
    w.v1 = w.v0 + 1024;
  } else {
    const int n = nb * 16 + (lane & 15);
    w.base = 0;
    w.step = 128;
    w.v0 = (uint32_t)((int64_t)(n < p.N ? n : p.N - 1) * p.b_sn + 16 * (lane >> 4));
    w.v1 = w.v0 + 64;
  }
  return w;
}

union Frag32 {  // 32 contiguous K bytes of one row = four MFMA operands
  uint4 v[2];
  i32x4 x[2];
  long l[4];
};

// Hand-scheduled weight loads for the EXACT A-stationary loop.  hipcc does not track inline-asm loads in its
// s_waitcnt pass, so the loop owns the counts: a slot is refilled right after its MFMAs, and the wait in
// front of slot i's MFMAs is vmcnt(2 * (PB - 1)) -- loads retire in order, and any compiler-issued VMEM in
// between only makes the count stricter, never wrong.  (With ordinary loads hipcc put a single vmcnt(0) at
// the loop head and sank the refills to the loop end: no load/MFMA overlap inside a wave.)
// Address = scalar base (saddr) + 32-bit per-lane offset: zero VALU per load.
#ifndef SGLM_W_NT
#define SGLM_W_NT 0
#endif
// timing ablations of fp8_gemm_wstream_kernel (WRONG RESULTS; tools/ab_variants.py only): no phase barriers / no LDS
// reads + MFMA / another weight-queue depth
#ifndef SGLM_ABL_SHUF
#define SGLM_ABL_SHUF 0
#endif
#ifndef SGLM_WS_ABL_NOBAR
#define SGLM_WS_ABL_NOBAR 0
#endif
#ifndef SGLM_WS_ABL_NOMFMA
#define SGLM_WS_ABL_NOMFMA 0
#endif
#ifdef SGLM_WS_PB
#define SGLM_WS_PB_SET 1
#endif
#ifndef SGLM_SLAB_SC1
#define SGLM_SLAB_SC1 0
#endif
__device__ __forceinline__ void gload32_asm(Frag32& f, const uint8_t* sbase, uint32_t voff) {
#if SGLM_W_NT
  asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(f.x[0]) : "v"(voff), "s"(sbase) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:64 nt" : "=v"(f.x[1]) : "v"(voff), "s"(sbase) : "memory");
#else
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f.x[0]) : "v"(voff), "s"(sbase) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(f.x[1]) : "v"(voff), "s"(sbase) : "memory");
#endif
}
// the same with the second instruction's offset in a register (row-major weights: + 64; pre-shuffled: + 1024)
#ifndef SGLM_W_NT_UNSPLIT
#define SGLM_W_NT_UNSPLIT 1  // the unsplit weight-streaming kernel reads its weights with the nt hint (see PB below)
#endif
template <bool NT = false>
__device__ __forceinline__ void gload32_asm2(Frag32& f, const uint8_t* sbase, uint32_t voff0, uint32_t voff1) {
  if constexpr (NT || SGLM_W_NT) {
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(f.x[0]) : "v"(voff0), "s"(sbase) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(f.x[1]) : "v"(voff1), "s"(sbase) : "memory");
  } else {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f.x[0]) : "v"(voff0), "s"(sbase) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f.x[1]) : "v"(voff1), "s"(sbase) : "memory");
  }
}
// (a non-temporal hint on these loads was measured: no difference on any decode shape, profiles/README.md)
// End of a hand-scheduled loop: the tail refills are never consumed, so hipcc considers their destination
// registers free right after the asm that issued them and may place epilogue address arithmetic there --
// which the load, landing later, overwrites (seen as a GPU memory fault).  Wait for everything, THEN "use"
// every queue register so that they stay allocated across the wait.
template <int PB>
__device__ __forceinline__ void drain_frags(Frag32 (&q)[PB]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < PB; ++i) asm volatile("" ::"v"(q[i].x[0]), "v"(q[i].x[1]));
}
template <int N>
__device__ __forceinline__ void wait_frag(Frag32& f) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(f.x[0]), "+v"(f.x[1]) : "n"(N) : "memory");
}

// Lane group g takes bytes [16g, 16g+16) and [64+16g, 64+16g+16) of each 128-wide k-step (`p` already
// includes the 16g offset, `kend` = K - 16g): the first load instruction of a wave then covers bytes 0..63
// of every row, the second 64..127 -- whole 64-B sectors instead of interleaved 16-B pieces.  The MFMA k
// index is contracted, so any k permutation is fine as long as A and B use the same one.
//
// The loads are UNCONDITIONAL (address clamped into the row, data masked afterwards).  A guarded load
// (`ok ? *p : 0`) makes hipcc branch around each load and put `s_waitcnt vmcnt(0)` right behind it, which
// serialises the whole prefetch queue -- measured: every skinny-GEMM variant ran at one memory latency per
// k-step until this was removed.  Rows beyond M / N need no masking at all: a garbage row of A (column of
// B) only reaches output rows (columns) that are never stored.  Only the K tail must read as zero.
// ld16 returns RAW data from a clamped (always valid) address; the K-tail zeroing is a separate mask
// applied where the fragment is CONSUMED -- masking at the load site would itself be an immediate use
// of the loaded value and pin a wait right behind the load.
__device__ __forceinline__ uint4 ld16(const uint8_t* p, int k, int kend) {
  // stay inside the row: p carries this lane's 16 g offset and kend = K - 16 g, so p + (kend - 16) is the row's LAST chunk
  // (K >= 16) whatever g is -- also when the lane's own offset lies beyond a short row (K < 128: kend - 16 < 0; clamping that to
  // 0 read up to 112 bytes past the row, i.e. past the allocation on the tensor's last row)
  const int kc = k < kend ? k : kend - 16;
  return *reinterpret_cast<const uint4*>(p + kc);
}
__device__ __forceinline__ uint4 mask16(uint4 v, int k, int kend) {
  const uint32_t m = k < kend ? 0xFFFFFFFFu : 0u;
  v.x &= m; v.y &= m; v.z &= m; v.w &= m;
  return v;
}
__device__ __forceinline__ uint4 ld16_masked(const uint8_t* p, int k, int kend) {  // prologue-only use
  return mask16(ld16(p, k, kend), k, kend);
}
__device__ __forceinline__ void load32(Frag32& f, const uint8_t* p, int k, int kend, bool /*row_ok*/) {
  f.v[0] = ld16(p, k, kend);
  f.v[1] = ld16(p, k + 64, kend);
}
// keep or zero a whole fragment (padding steps of a branch-free pipeline); a data select, not a branch
__device__ __forceinline__ void keep32(Frag32& f, bool keep) {
  const uint32_t m = keep ? 0xFFFFFFFFu : 0u;
  f.v[0].x &= m; f.v[0].y &= m; f.v[0].z &= m; f.v[0].w &= m;
  f.v[1].x &= m; f.v[1].y &= m; f.v[1].z &= m; f.v[1].w &= m;
}
// zero the part of a fragment that lies beyond K (only when K % 128 != 0; `ktail` is wave-uniform)
__device__ __forceinline__ void mask32(Frag32& f, int k, int kend, bool ktail) {
  if (ktail) {
    f.v[0] = mask16(f.v[0], k, kend);
    f.v[1] = mask16(f.v[1], k + 64, kend);
  }
}

// skinny: M <= 16*MB.  Workgroup = WK waves splitting K; every wave owns NB column blocks of 16
// (the same activation fragment feeds NB MFMAs, so the L2 traffic for A is 1/NB of the weight stream).
template <int OUT_DTYPE, int MB, int NB, int WK>
__global__ __launch_bounds__(64 * WK) void fp8_gemm_skinny_kernel(GemmArgs p) {
  using H = Half16<OUT_DTYPE>;
  using T = typename H::T;
  constexpr int PB = (MB * NB >= 8) ? 2 : 4;  // prefetch distance in k-steps (register budget)
  constexpr int COLS = 16 * NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [WK][MB*16][COLS]

  const int lane = threadIdx.x & 63;
  const int wk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * COLS;

  // this wave's K range, in 128-wide steps
  const int steps_total = (p.K + 127) >> 7;
  const int steps_per = (steps_total + WK - 1) / WK;
  const int s_begin = wk * steps_per;
  const int s_end = (s_begin + steps_per) < steps_total ? (s_begin + steps_per) : steps_total;
  const int nsteps = s_end > s_begin ? s_end - s_begin : 0;

  const uint8_t* brow[NB];
  bool n_ok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + 16 * nb + r16;
    n_ok[nb] = n < p.N;
    brow[nb] = p.b + (int64_t)(n_ok[nb] ? n : 0) * p.b_sn + 16 * g;
  }
  const uint8_t* arow[MB];
  bool a_ok[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = 16 * mb + r16;
    a_ok[mb] = m < p.M;
    arow[mb] = p.a + (int64_t)(a_ok[mb] ? m : 0) * p.a_sm + 16 * g;
  }

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The k sum is order-free, so every column block starts its sweep at a different k-step:
  // weight rows are K bytes apart, and a chip full of waves all reading the same residue
  // mod 4 KiB would camp on a few HBM channels.
  const int rot = nsteps > 0 ? (int)((blockIdx.x * 5u) % (unsigned)nsteps) : 0;
  auto kof = [&](int s) {
    int t = s + rot;
    t = t >= nsteps ? t - nsteps : t;
    return (s_begin + t) << 7;
  };
  const int kend = p.K - 16 * g;
  const bool ktail = (p.K & 127) != 0;

  // Software pipeline.  vmcnt retires IN ORDER, so consuming any load waits for every older one: the
  // activation fragments (L2-fast) must therefore be issued together with the weights of the SAME
  // future step -- otherwise using A(s) would drain the younger-than-needed weight prefetches and the
  // effective prefetch distance collapses to one step.  Both operands are PB steps ahead.
  // The loop body is BRANCH-FREE: every reload is unconditional (step index clamped to the last real step)
  // and padding steps are neutralised by zeroing the weight fragment where it is consumed.  A load under an
  // `if` makes hipcc merge old/new registers with copies right behind the load -- an immediate use, i.e. a
  // `vmcnt(0)` after every load and no prefetch at all (seen in the ISA of the first versions).
  Frag32 bq[PB][NB];
  Frag32 aq[PB][MB];
  const int last = nsteps > 0 ? nsteps - 1 : 0;
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int kk = kof(i < nsteps ? i : last);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) load32(aq[i][mb], arow[mb], kk, kend, true);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) load32(bq[i][nb], brow[nb], kk, kend, true);
  }

  for (int s0 = 0; s0 < nsteps; s0 += PB) {
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int s = s0 + i;
      const int kcur = kof(s < nsteps ? s : last);
      Frag32 af[MB], bf[NB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        af[mb] = aq[i][mb];
        mask32(af[mb], kcur, kend, ktail);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        bf[nb] = bq[i][nb];
        mask32(bf[nb], kcur, kend, ktail);
        keep32(bf[nb], s < nsteps);
      }
      {
        const int sn = s + PB;
        const int kn = kof(sn < nsteps ? sn : last);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) load32(aq[i][mb], arow[mb], kn, kend, true);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) load32(bq[i][nb], brow[nb], kn, kend, true);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af[mb].l[ks], bf[nb].l[ks], acc[mb][nb], 0, 0, 0);
    }
  }

  // ---- cross-wave K reduction + transposed epilogue through LDS
  // acc[mb][nb][r] = C[m = 16mb + 4g + r][n = n0 + 16nb + r16]
  constexpr int ROWS = MB * 16;
  {
    float* dst = red + wk * ROWS * COLS;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * mb + 4 * g + r) * COLS + 16 * nb + r16] = acc[mb][nb][r];
  }
  __syncthreads();
  // each thread finishes 8 consecutive columns of one row
  constexpr int CHUNKS = ROWS * (COLS / 8);
  for (int c = threadIdx.x; c < CHUNKS; c += 64 * WK) {
    const int m = c / (COLS / 8);
    const int cc = (c - m * (COLS / 8)) * 8;
    const int nn = n0 + cc;
    if (m >= p.M || nn >= p.N) continue;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
    for (int kk = 0; kk < WK; ++kk) {
      const float* src = red + (kk * ROWS + m) * COLS + cc;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] += lo[j];
        v[4 + j] += hi[j];
      }
    }
    const float sa = p.sa[m];
    typename H::x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // epilogue order of fp8_gemm_kernel.cu:498-546: acc * w_scale[col], then * x_scale[row], then + bias
      float r = v[j] * p.sb[nn + j] * sa;
      if (p.bias) r += H::to_f32(reinterpret_cast<const T*>(p.bias)[nn + j]);
      o[j] = H::from_f32(r);
    }
    *reinterpret_cast<typename H::x8*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.N + nn) = o;
  }
}

// ------------------------------------------------------------------------------------------
// One-shot skinny GEMM: K small enough (<= 8 waves x 4 k-steps, K % 128 == 0) that a wave can put its WHOLE
// K-slice in flight at once.  Workgroup = one 16-column block, WK waves splitting K, S (compile-time) k-steps
// per wave.  Every weight load of the wave is issued first (HBM latency), then every activation load (L2),
// all through inline asm with scalar-base addressing (no VALU, no compiler-placed waits); the MFMAs of step
// s then wait on an explicit count.  The whole kernel is one memory latency + the transfer time, where the
// pipelined kernel above pays one latency per PB steps plus ~100 VALU per step of address / mask work.
template <int N, int MB>
__device__ __forceinline__ void wait_step(Frag32& b, Frag32 (&a)[MB]) {
  if constexpr (MB == 1)
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b.x[0]), "+v"(b.x[1]), "+v"(a[0].x[0]), "+v"(a[0].x[1]) : "n"(N) : "memory");
  else if constexpr (MB == 2)
    asm volatile("s_waitcnt vmcnt(%6)"
                 : "+v"(b.x[0]), "+v"(b.x[1]), "+v"(a[0].x[0]), "+v"(a[0].x[1]), "+v"(a[1].x[0]), "+v"(a[1].x[1])
                 : "n"(N) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+v"(b.x[0]), "+v"(b.x[1]), "+v"(a[0].x[0]), "+v"(a[0].x[1]), "+v"(a[1].x[0]), "+v"(a[1].x[1]),
                   "+v"(a[2].x[0]), "+v"(a[2].x[1]), "+v"(a[3].x[0]), "+v"(a[3].x[1])
                 : "n"(N) : "memory");
}

template <int OUT_DTYPE, int MB, int S>
__global__ __launch_bounds__(512) void fp8_gemm_oneshot_kernel(GemmArgs p) {
  static_assert(MB == 1 || MB == 2 || MB == 4, "MB");
  using H = Half16<OUT_DTYPE>;
  using T = typename H::T;
  constexpr int ROWS = MB * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [WK][ROWS][16]

  const int lane = threadIdx.x & 63;
  const int wk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int WK = blockDim.x >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const WFrag wf = wfrag_addr(p, blockIdx.x, lane);
  uint32_t aoff[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = 16 * mb + r16;
    aoff[mb] = (uint32_t)((int64_t)(m < p.M ? m : p.M - 1) * p.a_sm + 16 * g);  // rows past M are never stored
  }
  const int k0 = (wk * S) << 7;

  Frag32 bq[S];
  Frag32 aq[S][MB];
#pragma unroll
  for (int s = 0; s < S; ++s) gload32_asm2(bq[s], p.b + wf.base + (int64_t)(wk * S + s) * wf.step, wf.v0, wf.v1);
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) gload32_asm(aq[s][mb], p.a + k0 + (s << 7), aoff[mb]);

  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // step s may run once at most the activation loads of the later steps are outstanding (in-order retirement)
#define SGLM_ONESHOT_STEP(s_)                                                                              \
  if constexpr (s_ < S) {                                                                                  \
    wait_step<(S - 1 - s_) * 2 * MB, MB>(bq[s_ < S ? s_ : 0], aq[s_ < S ? s_ : 0]);                         \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                        \
    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                       \
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(aq[s_ < S ? s_ : 0][mb].l[kk],                   \
                                                            bq[s_ < S ? s_ : 0].l[kk], acc[mb], 0, 0, 0);    \
    __builtin_amdgcn_sched_barrier(0); /* or hipcc hoists the later waits above these MFMAs */             \
  }
  SGLM_ONESHOT_STEP(0)
  SGLM_ONESHOT_STEP(1)
  SGLM_ONESHOT_STEP(2)
  SGLM_ONESHOT_STEP(3)
#undef SGLM_ONESHOT_STEP

  // ---- cross-wave K reduction + epilogue through LDS: acc[mb][r] = C[16mb + 4g + r][n0 + r16]
  {
    float* dst = red + wk * ROWS * 16;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(16 * mb + 4 * g + r) * 16 + r16] = acc[mb][r];
  }
  __syncthreads();
  // each thread finishes 8 consecutive columns of one row
  for (int c = threadIdx.x; c < ROWS * 2; c += blockDim.x) {
    const int m = c >> 1;
    const int cc = (c & 1) * 8;
    const int nn = n0 + cc;
    if (m >= p.M || nn >= p.N) continue;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    for (int kk = 0; kk < WK; ++kk) {
      const float* src = red + (kk * ROWS + m) * 16 + cc;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] += lo[j];
        v[4 + j] += hi[j];
      }
    }
    const float sa = p.sa[m];
    typename H::x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // epilogue order of fp8_gemm_kernel.cu:498-546: acc * w_scale[col], then * x_scale[row], then + bias
      float r = v[j] * p.sb[nn + j] * sa;
      if (p.bias) r += H::to_f32(reinterpret_cast<const T*>(p.bias)[nn + j]);
      o[j] = H::from_f32(r);
    }
    *reinterpret_cast<typename H::x8*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.N + nn) = o;
  }
}

// picks (S, WK) with WK * S == K / 128, S <= 4, WK <= 8; `used` false when the shape does not qualify
template <int OUT_DTYPE, int MB>
int launch_oneshot(const GemmArgs& p, hipStream_t s, bool& used) {
  used = false;
  if ((p.K & 127) != 0 || (p.N & 7) != 0) return 0;
  if ((!p.b_shuf && (int64_t)p.N * p.b_sn >= ((int64_t)1 << 32)) || (int64_t)p.M * p.a_sm >= ((int64_t)1 << 32)) return 0;
  const int steps = p.K >> 7;
  int S = 0;
  for (int c = 1; c <= 4; ++c)
    if (steps % c == 0 && steps / c <= 8) { S = c; break; }
  if (S == 0) return 0;
  const int WK = steps / S;
  const unsigned grid = (unsigned)((p.N + 15) / 16);
  // M > 32: 167 VGPRs -> one workgroup per CU; a second round of workgroups (N = 6144: 384) costs more than the
  // pipelined kernel's 32-column tiles (measured 23.0 vs 20.6 us), so only take shapes that fit one round.
  static const unsigned max_grid4 = [] { const char* e = getenv("SGL_MI355_ONESHOT_MAX_GRID"); return e ? (unsigned)atoi(e) : 256u; }();  // tuning aid
  // (one k-step per wave is cheap enough for several rounds: K = 1024, N = 8192 -- 512 workgroups -- 8.2 us vs 14.4)
  if (MB == 4 && grid > max_grid4 && !(S == 1 && grid <= 1024)) return 0;
  const int lds = WK * MB * 16 * 16 * 4;
#define OS_GO(S_)                                                                                          \
  hipLaunchKernelGGL((fp8_gemm_oneshot_kernel<OUT_DTYPE, MB, S_>), dim3(grid), dim3(64 * WK), lds, s, p)
  if (S == 1) OS_GO(1);
  else if (S == 2) OS_GO(2);
  else if (S == 3) OS_GO(3);
  else OS_GO(4);
#undef OS_GO
  g_last_kernel = "oneshot";
  used = true;
  return check_hip(hipGetLastError(), "fp8_gemm_oneshot launch");
}

// ------------------------------------------------------------------------------------------
// A-stationary skinny GEMM (M <= 64): the decode-time weight streamer.
//
// Measured on the register-only kernel above: at M = 64 it is bound by the ACTIVATION loads, not by
// HBM -- every wave re-fetches fragment-shaped pieces of A (16 rows x 16 B per instruction) from L2
// and the texture-address path saturates long before the weight stream does.  160 KB of LDS per CU
// changes the structure: one persistent workgroup per CU keeps its K-slice of A (<= 64 x 2048 B)
// resident in LDS for the whole kernel, in the fragment-major image the MFMA wants, and its 16 waves
// then do nothing but stream weights: each wave walks its own column blocks, 4 k-steps of 2 KB ahead
// (full 128-B lines), reading A fragments by conflict-free ds_read_b128.  No cross-wave reduction, no
// barrier after the fill.  K is split over SK workgroup classes (fp32 slabs, summed by the epilogue
// kernel that also applies the scales / bias) so that the slice fits LDS and 256 CUs have work.
constexpr int kAsWaves = 16;

template <int MB>
struct AStat {
  static constexpr int ROWS = 16 * MB;
  static constexpr int REGION = ROWS * 32 + 16;   // one 32-B k-chunk column of a step, +16 B skew
  static constexpr int STEP_BYTES = 4 * REGION;   // 128 k per step
  static constexpr int MAX_STEPS = (150 * 1024) / STEP_BYTES;
};

// Cooperative fill of the A-stationary LDS image: rows [0,ROWS) x k in [k_begin, k_begin + nsteps*128).
// Eight independent 16-B loads per thread are in flight before the first is consumed (a plain
// load->store loop would pay one L2 round trip per 16 B).
template <int ROWS, int REGION, int STEP_BYTES>
__device__ __forceinline__ void astat_fill(char* smem, const GemmArgs& p, int k_begin, int nsteps, int tid, int nthreads) {
  const int per_row = nsteps * 8;  // 16-B chunks per row
  const int total = ROWS * per_row;
  for (int base = tid; base < total; base += nthreads * 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int idx = base + u * nthreads;
      idx = idx < total ? idx : total - 1;
      const int row = idx / per_row;
      const int c = idx - row * per_row;
      const int rc = row < p.M ? row : p.M - 1;  // rows past M are never stored; keep the load unconditional
      v[u] = ld16(p.a + (int64_t)rc * p.a_sm, k_begin + c * 16, p.K);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * nthreads;
      if (idx < total) {
        const int row = idx / per_row;
        const int c = idx - row * per_row;
        const int st = c >> 3, j = c & 7;
        *reinterpret_cast<uint4*>(smem + st * STEP_BYTES + (j & 3) * REGION + row * 32 + (j >> 2) * 16) =
            mask16(v[u], k_begin + c * 16, p.K);
      }
    }
  }
}

template <int MB>
__global__ __launch_bounds__(64 * kAsWaves) void fp8_gemm_astat_kernel(GemmArgs p, float* slabs, int SK, int steps_per_slice) {
  using L = AStat<MB>;
  constexpr int PB = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int ks = blockIdx.x % SK;
  const int ng = blockIdx.x / SK;
  const int ngroups = gridDim.x / SK;
  const int nwaves = blockDim.x >> 6;
  const int steps_total = (p.K + 127) >> 7;
  const int st_begin = ks * steps_per_slice;
  const int nsteps = (st_begin + steps_per_slice) <= steps_total ? steps_per_slice : (steps_total - st_begin);
  if (nsteps <= 0) return;  // workgroup-uniform
  const int k_begin = st_begin << 7;

  // ---- this wave's flat work list: column blocks gw, gw + stride, ... times nsteps k-steps
  const int nblocks = (p.N + 15) >> 4;
  const int gw = ng * nwaves + wave;
  const int stride = ngroups * nwaves;
  const int my_blocks = gw < nblocks ? (nblocks - gw + stride - 1) / stride : 0;
  const int total_f = my_blocks * nsteps;
  const int kend = p.K - 16 * g;
  const bool ktail = (p.K & 127) != 0;

  auto issue = [&](Frag32& f, int fi) __attribute__((always_inline)) {
    const int bi = fi / nsteps;
    const int st = fi - bi * nsteps;
    const int n = (gw + bi * stride) * 16 + r16;
    const bool ok = n < p.N;
    load32(f, p.b + (int64_t)(ok ? n : 0) * p.b_sn + 16 * g, k_begin + (st << 7), kend, ok);
  };

  // the first weight loads go out BEFORE the A fill: they do not depend on it and their HBM latency
  // then overlaps the fill.  All loads are unconditional (flat index clamped), see the skinny kernel.
  const int last_f = total_f > 0 ? total_f - 1 : 0;
  Frag32 bq[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) issue(bq[i], i < total_f ? i : last_f);

  // ---- fill: A[0:ROWS][k_begin : k_begin + nsteps*128] -> LDS, fragment-major
  astat_fill<L::ROWS, L::REGION, L::STEP_BYTES>(smem, p, k_begin, nsteps, tid, (int)blockDim.x);
  __syncthreads();

  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* abase = smem + g * L::REGION + r16 * 32;
  int st = 0, bi = 0;
  for (int f0 = 0; f0 < total_f; f0 += PB) {
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int f = f0 + i;
      Frag32 bf = bq[i];
      mask32(bf, k_begin + (st << 7), kend, ktail);  // A's tail is zero in LDS already; keep B finite too
      keep32(bf, f < total_f);
      {
        const int fn = f + PB;
        issue(bq[i], fn < total_f ? fn : last_f);
      }
      const char* ap = abase + st * L::STEP_BYTES;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        Frag32 af;
        af.v[0] = *reinterpret_cast<const uint4*>(ap + mb * 512);
        af.v[1] = *reinterpret_cast<const uint4*>(ap + mb * 512 + 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af.l[kk], bf.l[kk], acc[mb], 0, 0, 0);
      }
      if (++st == nsteps) {
        // column block finished: acc[mb][r] = C[m = 16mb + 4g + r][n]; store the fp32 partial
        const int n = (gw + bi * stride) * 16 + r16;
        if (bi < my_blocks && n < p.N) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int m = 16 * mb + 4 * g + r;
              if (m < p.M) slabs[((int64_t)ks * p.M + m) * p.N + n] = acc[mb][r];
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        st = 0;
        ++bi;
      }
    }
  }
}

// A-stationary, DIRECT epilogue: for wide N there are enough column blocks that every wave can own ONE
// block for the whole K range.  The workgroup then walks K in phases (each phase refills the LDS image of
// A with the next K-slice), the accumulator never leaves registers, and the epilogue (scales, bias, cast)
// happens in the same kernel -- no fp32 slabs, no second launch.  The next phase's first weight loads are
// issued before the refill barrier so the HBM stream never drains.
template <int OUT_DTYPE, int MB, bool EXACT>
__global__ __launch_bounds__(512) void fp8_gemm_astat_direct_kernel(GemmArgs p, int SK, int steps_per_slice) {
  using H = Half16<OUT_DTYPE>;
  using T = typename H::T;
  using L = AStat<MB>;
  constexpr int PB = 8;  // 16 KB of weights in flight per wave
  const int NW = blockDim.x >> 6;  // 4..8 waves, chosen by the launcher so the grid fills the CUs evenly
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int steps_total = (p.K + 127) >> 7;
  const int nb = blockIdx.x * NW + wave;  // this wave's column block
  const int n = nb * 16 + r16;
  const bool n_ok = n < p.N;
  // Weight address = (scalar) base + k  +  (per-lane, 32-bit) row offset: the per-step address math stays
  // on the scalar unit and the load uses the saddr form -- no VALU per load.
  const uint32_t lane_off = (uint32_t)((int64_t)(n_ok ? n : 0) * p.b_sn + 16 * g);
  const int kend = p.K - 16 * g;
  // EXACT (compile time): K % 128 == 0 and every phase length is a multiple of PB -> no masks at all.

  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Phases are contiguous runs of k-steps; inside a phase the sweep is rotated per column block (the sum
  // is order-free) so that waves do not all read the same residue mod 4 KiB at the same time. Only two
  // phase lengths exist (full / last), so rotation and length are scalar selects, no division in the loop.
  const int ns_last = steps_total - (SK - 1) * steps_per_slice;
  const int rot_full = (int)(((unsigned)(nb * 6) % (unsigned)steps_per_slice) & ~1u);
  const int rot_last = (int)(((unsigned)(nb * 6) % (unsigned)ns_last) & ~1u);
  auto ldw = [&](Frag32& f, int kstep_) __attribute__((always_inline)) {
    const int k = kstep_ << 7;
    if constexpr (EXACT) {
      gload32_asm(f, p.b + k, lane_off);  // scalar base
    } else {
      f.v[0] = ld16(p.b + lane_off, k, kend);
      f.v[1] = ld16(p.b + lane_off, k + 64, kend);
    }
  };

  // prefetch cursor: (phase, local step) of the NEXT load to issue; branch-free scalar updates
  int pf_ph = 0, pf_sl = 0;
  auto next_kstep = [&]() __attribute__((always_inline)) {
    const bool last = pf_ph == SK - 1;
    const int ns = last ? ns_last : steps_per_slice;
    const int rot = last ? rot_last : rot_full;
    int t = pf_sl + rot;
    t = t >= ns ? t - ns : t;
    const int ks = pf_ph * steps_per_slice + t;
    const bool wrap = pf_sl + 1 >= ns;
    // after the very last step stay on it: the extra loads are harmless and never consumed
    pf_sl = wrap ? (last ? pf_sl : 0) : pf_sl + 1;
    pf_ph = (wrap && !last) ? pf_ph + 1 : pf_ph;
    return ks;
  };

  Frag32 bq[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) ldw(bq[i], next_kstep());

  const char* abase = smem + g * L::REGION + r16 * 32;
  for (int ph = 0; ph < SK; ++ph) {
    const int st_begin = ph * steps_per_slice;
    const int nsteps = ph == SK - 1 ? ns_last : steps_per_slice;
    const int rot = ph == SK - 1 ? rot_last : rot_full;
    if (ph > 0) __syncthreads();  // everyone is done reading the previous slice
    astat_fill<L::ROWS, L::REGION, L::STEP_BYTES>(smem, p, st_begin << 7, nsteps, tid, 64 * NW);
    // vmcnt(0), visible to hipcc's waitcnt pass (the fill just drained everything anyway): otherwise a
    // "maybe pending" fill load survives into the loop and costs a vmcnt(0) per 8 steps.
    if constexpr (EXACT) __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int s0 = 0; s0 < nsteps; s0 += PB) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {  // queue slot i == (flat step) % PB (phase lengths are multiples of PB
        const int sl = s0 + i;        //  except possibly the last one, whose padding steps are zeroed)
        int t = (sl < nsteps ? sl : nsteps - 1) + rot;
        t = t >= nsteps ? t - nsteps : t;  // local (rotated) step inside the LDS image
        if constexpr (EXACT) {
          wait_frag<2 * (PB - 1)>(bq[i]);
        } else {
          mask32(bq[i], (st_begin + t) << 7, kend, true);
          keep32(bq[i], sl < nsteps);
        }
        const char* ap = abase + t * L::STEP_BYTES;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          Frag32 af;
          af.v[0] = *reinterpret_cast<const uint4*>(ap + mb * 512);
          af.v[1] = *reinterpret_cast<const uint4*>(ap + mb * 512 + 16);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af.l[kk], bq[i].l[kk], acc[mb], 0, 0, 0);
        }
        ldw(bq[i], next_kstep());  // refill the slot AFTER its MFMAs consumed it: no register copy
        __builtin_amdgcn_sched_barrier(0);  // keep the refill here (the scheduler would sink all 8 to the loop end)
      }
    }
  }

  // ---- epilogue through a wave-private LDS patch (transposes to 16-B row segments)
  if constexpr (EXACT) drain_frags(bq);  // the never-consumed tail loads
  __syncthreads();  // the A image is dead: reuse its memory
  T* ep = reinterpret_cast<T*>(smem) + wave * (L::ROWS * 24);  // [ROWS][16] (+8 pad)
  const float sbv = p.sb[n_ok ? n : p.N - 1];
  const float bv = p.bias ? H::to_f32(reinterpret_cast<const T*>(p.bias)[n_ok ? n : p.N - 1]) : 0.f;
  // unconditional (clamped) scale loads, all in flight together: `m < M ? sa[m] : 0` compiles to a branch
  // plus vmcnt(0) per row -- 16 serial L2 round trips at M = 64
  float sav[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 16 * mb + 4 * g + r;
      sav[mb][r] = p.sa[m < p.M ? m : p.M - 1];
    }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 16 * mb + 4 * g + r;
      ep[m * 24 + r16] = H::from_f32(acc[mb][r] * sbv * sav[mb][r] + bv);
    }
  wait_lgkmcnt0();
  for (int c = lane; c < L::ROWS * 2; c += 64) {
    const int m = c >> 1, half = c & 1;
    const int nn = nb * 16 + half * 8;
    if (m < p.M && nn < p.N)
      *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.N + nn) =
          *reinterpret_cast<const uint4*>(ep + m * 24 + half * 8);
  }
}

// ------------------------------------------------------------------------------------------
// Weight-streaming skinny GEMM for wide N (gate_up at decode): one column block per wave over ALL of K,
// activations double-buffered through LDS by LDS-DMA while the weights stream through registers.
//
// What the phase-filled kernel above loses at M = 64: its A image is rebuilt per phase by a
// load -> mask -> ds_write loop of ~40 VALU per 16 B behind two barriers, during which no new weight load is
// issued (2 x 128 KB per workgroup at K = 4096; gate_up: 41 us at M = 64 vs 28 us at M = 16).  Here
//   * the image is [step][row][128 B] with the 16-B chunk index XOR-swizzled by (row >> 1) & 7: a DMA
//     instruction lands 8 rows x 128 B = 1 KiB contiguous in LDS from 8 fully coalesced 128-B row segments
//     (the swizzle is applied on the SOURCE side, inside one 128-B line), and the MFMA-fragment
//     ds_read_b128 of 16 rows x 16 B hits 16 distinct bank groups;
//   * a dedicated PRODUCER wave (the last one of the workgroup) DMAs phase p+1 (PH k-steps, <= 64 KiB) into the
//     other buffer right after the barrier that opens phase p: one barrier per phase, no VGPR round trip,
//     and the 4..8 consumer waves run a pure weight loop with a single wait count (vmcnt(14));
//   * the consumer count is chosen per shape so that the workgroups fill the 256 CUs evenly (a CU moves
//     only ~10 B/clk from HBM, so an idle CU is lost bandwidth: N = 28672 -> 7 consumers x 256 workgroups).
template <int OUT_DTYPE, int MB, int PH, bool SLAB>
__global__ __launch_bounds__(576) void fp8_gemm_wstream_kernel(GemmArgs p, float* slabs, int phases_per_slice) {
  static_assert(PH == 2 || PH == 4 || PH == 8 || PH == 16 || PH == 32, "PH");
  static_assert(PH * MB <= 32, "one A buffer is at most 64 KiB");
  using H = Half16<OUT_DTYPE>;
  using T = typename H::T;
  constexpr int ROWS = 16 * MB;
  constexpr int STEP_BYTES = ROWS * 128;
  constexpr int BUF_BYTES = PH * STEP_BYTES;  // <= 64 KiB
  // Weight k-steps in flight per wave.  Round 1 (row-major weights, 16 rows x 64 B per load): same-box A/B of 8 / 4 / 2 / 1
  // gave gate_up 33.1 / 32.4 / 30.8 / 37.8 us, down 24.6 / 23.4 / 22.6 / 26.5, qkv 16.6 / 15.7 / 15.0 / 16.6 at M = 64 --
  // two it was.  Round 2 (contiguous 1-KiB loads on pre-shuffled weights): the unsplit kernel (gate_up) does better with
  // FOUR steps and a non-temporal hint on the weight stream (25.4 -> 23.8 us; whole step 6.16 -> 6.10 ms), the split-K
  // kernels stay at two without the hint (down 20.5 vs 20.75 with four; the hint costs them ~0.5 us each).
#ifdef SGLM_WS_PB_SET
  constexpr int PB = SGLM_WS_PB;
#else
  constexpr int PB = (SLAB || PH < 4) ? 2 : 4;  // (never more than a phase: PH = 2 is the 256-row form)
#endif
  constexpr int UPS = 2 * MB;                 // 1-KiB DMA units per k-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NC = (int)(blockDim.x >> 6) - 1;  // consumer waves; wave NC is the DMA producer
  const int r16 = lane & 15, g = lane >> 4;
  const int P_total = (p.K >> 7) / PH;
  // SLAB: blockIdx.y owns phases [ph0, ph1) and writes an fp32 partial; otherwise all of K
  const int ph0 = SLAB ? (int)blockIdx.y * phases_per_slice : 0;
  const int ph1 = SLAB ? (ph0 + phases_per_slice < P_total ? ph0 + phases_per_slice : P_total) : P_total;
  const int nph = ph1 - ph0;
  const int nb = blockIdx.x * NC + wave;  // a consumer wave's column block
  const int n = nb * 16 + r16;
  const bool n_ok = n < p.N;
  const uint32_t smem_base = lds_addr_of(smem);

  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // a16 form (single-phase slices only): every wave of the workgroup converts its share of the slice's 16-bit
  // activations to e4m3 straight into the phase image -- 32 source bytes -> one 16-byte chunk at its swizzled position
  auto fill_a16 = [&]() __attribute__((always_inline)) {
    const T* src = reinterpret_cast<const T*>(p.a16);
    const int per_row = PH * 8;  // 16-byte chunks of one row in the phase
    const int total = ROWS * per_row;
    constexpr int UB = 8;        // units whose loads are in flight together (one memory round trip per batch)
    for (int u0 = tid; u0 < total; u0 += UB * (int)blockDim.x) {
      typename H::x8 v0[UB], v1[UB];
      float amax[UB];
#pragma unroll
      for (int i = 0; i < UB; ++i) {  // unconditional loads from clamped addresses (gemm_fp8.hip 4.3.1 c)
        int u = u0 + i * (int)blockDim.x;
        u = u < total ? u : total - 1;
        const int row = u / per_row, kpos = u - row * per_row;
        const int mrow = row < p.M ? row : p.M - 1;  // rows past M: never stored
        const typename H::x8* sp =
            reinterpret_cast<const typename H::x8*>(src + (int64_t)mrow * p.a16_sm + ((int64_t)ph0 * PH << 7) + kpos * 16);
        v0[i] = sp[0];
        v1[i] = sp[1];
        amax[i] = p.a_absmax[mrow];
      }
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int u = u0 + i * (int)blockDim.x;
        if (u < total) {
          const int row = u / per_row, kpos = u - row * per_row;
          const float scale = amax[i] / 448.0f;
          const float inv = scale == 0.f ? 0.f : 1.0f / scale;
          uint32_t w[4];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const typename H::x8 v = h ? v1[i] : v0[i];
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(H::to_f32(v[j]) * inv, -448.0f), 448.0f);
            int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
            int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
            w[2 * h] = (uint32_t)lo;
            w[2 * h + 1] = (uint32_t)hi;
          }
          const int step = kpos >> 3, j = kpos & 7;
          *reinterpret_cast<uint4*>(smem + step * STEP_BYTES + row * 128 + 16 * (j ^ ((row >> 1) & 7))) =
              uint4{w[0], w[1], w[2], w[3]};
        }
      }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && p.a_scale_out != nullptr && tid < p.M) p.a_scale_out[tid] = p.a_absmax[tid] / 448.0f;
  };

  if (wave == NC) {
    // ---------------- producer: the A image of every phase, one 1-KiB unit (8 rows x 128 B) per instruction.
    // Lane i of a unit lands at LDS chunk i & 7 of row i >> 3 and therefore fetches source chunk
    // (i & 7) ^ ((row >> 1) & 7) of that row: the swizzle is applied inside one coalesced 128-B line.
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
      }
    };
    // One workgroup barrier per phase.  (LDS flags instead of barriers were tried -- they would let the waves
    // drift apart -- and are NOT valid: LDS-DMA data is ordered for another wave's ds_read only by the issuing
    // wave's vmcnt followed by a barrier the reader has passed; with a ds_write flag the M = 7 down_proj case
    // read stale rows.  They were not faster either.)
    if (p.a16 != nullptr) fill_a16(); else dma_phase(ph0, 0);
    for (int lp = 0; lp < nph; ++lp) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // phase lp has landed
#if !SGLM_WS_ABL_NOBAR
      __syncthreads();                                   // barrier #lp: consumers may read it; the other buffer is free
#endif
      if (lp + 1 < nph) dma_phase(ph0 + lp + 1, (lp + 1) & 1);
    }
    __syncthreads();  // the consumers' final barrier
    return;
  }

  // ---------------- consumers: one column block each, weights through registers, A fragments from LDS
  const WFrag wf = wfrag_addr(p, (nb * 16 < p.N) ? nb : 0, lane);
  const uint8_t* wbase = p.b + wf.base;
  const int rot = (nb * 3) & (PH - 1);   // per-column-block rotation of the sweep inside a phase
  const int last = ph1 * PH - 1;
  // A fragment offsets inside a k-step: row r16 (+16 mb), chunks g and 4 + g, swizzled
  const int sw = (r16 >> 1) & 7;
  const uint32_t o0 = r16 * 128 + 16 * (g ^ sw);
  const uint32_t o1 = r16 * 128 + 16 * ((4 + g) ^ sw);

  int f_pf = ph0 * PH;  // next flat step to prefetch; phases are contiguous runs of PH flat steps
  auto refill = [&](Frag32& fr) __attribute__((always_inline)) {
    const int f = f_pf < last ? f_pf : last;  // tail refills re-read the last step, never consumed
    const int ks = (f & ~(PH - 1)) + ((f + rot) & (PH - 1));
    gload32_asm2<(SGLM_W_NT_UNSPLIT != 0) && !SLAB>(fr, wbase + (int64_t)ks * wf.step, wf.v0, wf.v1);
    ++f_pf;
  };
  Frag32 bq[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) refill(bq[i]);
  if (p.a16 != nullptr) fill_a16();  // the weight prefetch is in flight meanwhile

  // ONE copy of the 8-step body (a second copy -- e.g. a variant with another wait count -- makes hipcc merge
  // the in-flight weight registers of the two paths with v_mov copies placed in front of the wait, i.e. reads
  // of registers whose loads have not landed).
#pragma clang loop unroll(disable)  // no peeled copies either: see above
  for (int lp = 0; lp < nph; ++lp) {
#if !SGLM_WS_ABL_NOBAR
    __syncthreads();  // barrier #lp: phase lp is in LDS
#endif
    const uint32_t abuf = (lp & 1) * BUF_BYTES;
    for (int s0 = 0; s0 < PH; s0 += PB) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int t = (s0 + i + rot) & (PH - 1);  // local step held by slot i
        wait_frag<2 * (PB - 1)>(bq[i]);
        const char* a0 = smem + abuf + t * STEP_BYTES + o0;
        const char* a1 = smem + abuf + t * STEP_BYTES + o1;
#if SGLM_WS_ABL_NOMFMA
        (void)a0; (void)a1;
        acc[0][0] += __builtin_bit_cast(float, (int)bq[i].l[0]);
#else
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          Frag32 af;
          af.v[0] = *reinterpret_cast<const uint4*>(a0 + mb * 2048);
          af.v[1] = *reinterpret_cast<const uint4*>(a1 + mb * 2048);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af.l[kk], bq[i].l[kk], acc[mb], 0, 0, 0);
        }
#endif
        refill(bq[i]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  drain_frags(bq);  // the never-consumed tail refills
  __syncthreads();  // the A buffers are dead: reuse their memory
  if constexpr (SLAB) {
    // ---- fp32 partial of this K slice -> slabs[blockIdx.y][m][n], 16-B stores through a wave-private patch
    float* ep = reinterpret_cast<float*>(smem) + wave * (ROWS * 20);  // [ROWS][16] (+4 pad)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(16 * mb + 4 * g + r) * 20 + r16] = acc[mb][r];
    wait_lgkmcnt0();
    float* dst = slabs + (int64_t)blockIdx.y * p.M * p.N;
    for (int c = lane; c < ROWS * 4; c += 64) {
      const int m = c >> 2, q = c & 3;
      const int nn = nb * 16 + q * 4;
      if (m < p.M && nn < p.N) {
#if SGLM_SLAB_SC1
        // write-through: the slab leaves L2 while the kernel is still streaming instead of at its end
        const f32x4 val = *reinterpret_cast<const f32x4*>(ep + m * 20 + q * 4);
        float* gp = dst + (int64_t)m * p.N + nn;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(gp), "v"(val) : "memory");
#else
        *reinterpret_cast<f32x4*>(dst + (int64_t)m * p.N + nn) = *reinterpret_cast<const f32x4*>(ep + m * 20 + q * 4);
#endif
      }
    }
  } else {
    // ---- epilogue through a wave-private LDS patch (transposes to 16-B row segments)
    T* ep = reinterpret_cast<T*>(smem) + wave * (ROWS * 24);  // [ROWS][16] (+8 pad)
    const float sbv = p.sb[n_ok ? n : p.N - 1];
    const float bv = p.bias ? H::to_f32(reinterpret_cast<const T*>(p.bias)[n_ok ? n : p.N - 1]) : 0.f;
    // unconditional (clamped) scale loads, all in flight together: `m < M ? sa[m] : 0` compiles to a branch
    // plus vmcnt(0) per row -- 16 serial L2 round trips at M = 64
    float sav[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * mb + 4 * g + r;
        sav[mb][r] = p.sa[m < p.M ? m : p.M - 1];
      }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * mb + 4 * g + r;
        ep[m * 24 + r16] = H::from_f32(acc[mb][r] * sbv * sav[mb][r] + bv);
      }
    wait_lgkmcnt0();
    for (int c = lane; c < ROWS * 2; c += 64) {
      const int m = c >> 1, half = c & 1;
      const int nn = nb * 16 + half * 8;
      if (m < p.M && nn < p.N)
        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.N + nn) =
            *reinterpret_cast<const uint4*>(ep + m * 24 + half * 8);
    }
  }
}

// finalize of the slab variants (defined below)
template <int OUT_DTYPE>
__global__ void fp8_gemm_finalize_kernel(GemmArgs p, const float* slabs, int SK);

template <int OUT_DTYPE, int MB, int PH, bool SLAB>
int launch_wstream_ph(const GemmArgs& p, float* slabs, int SK, int phases_per_slice, int nc, int groups, hipStream_t s,
                      bool finalize = true) {
  auto kern = fp8_gemm_wstream_kernel<OUT_DTYPE, MB, PH, SLAB>;
  constexpr int lds_ab = 2 * PH * 16 * MB * 128;                       // the two A buffers
  constexpr int lds_ep = 8 * 16 * MB * (SLAB ? 20 * 4 : 24 * 2);        // epilogue patches of up to 8 consumer waves
  constexpr int lds = lds_ab > lds_ep ? lds_ab : lds_ep;
  static_assert(lds <= 160 * 1024, "LDS");
  static int attr_rc = check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
      "hipFuncSetAttribute");
  if (attr_rc) return attr_rc;
  g_last_kernel = SLAB ? "wstream_slab" : "wstream";
  hipLaunchKernelGGL(kern, dim3((unsigned)groups, (unsigned)SK), dim3(64 * (nc + 1)), lds, s, p, slabs,
                     phases_per_slice);
  int rc = check_hip(hipGetLastError(), "fp8_gemm_wstream launch");
  if (rc || !SLAB || !finalize) return rc;
  const int64_t total = (int64_t)p.M * p.N / 8;
  hipLaunchKernelGGL((fp8_gemm_finalize_kernel<OUT_DTYPE>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p,
                     (const float*)slabs, SK);
  return check_hip(hipGetLastError(), "fp8_gemm_finalize launch");
}

// Shapes: K % 128 == 0 and K / 128 a multiple of a phase length PH in {32, 16, 8} (PH * MB <= 32).
// slabs == nullptr: one workgroup per 8 column blocks over all of K (wide N).  Otherwise K is split into SK
// slices of whole phases so that ~256 workgroups exist, partials go to fp32 slabs and a finalize kernel applies
// the reference epilogue (narrow N, long K: down_proj).
template <int OUT_DTYPE, int MB>
int launch_wstream(const GemmArgs& p, float* slabs, int64_t slab_floats, hipStream_t s, bool& used,
                   int* partial_slices = nullptr) {
  // partial_slices != nullptr: leave the raw fp32 partial sums in `slabs` (any slice count >= 1) and report the
  // count; the epilogue runs inside the consumer kernel (sgl_mi355_*_from_partials).
  used = false;
  if ((p.K & 127) != 0 || (p.N & 7) != 0 || (p.a_sm & 15) != 0) return 0;
  if (!p.b_shuf && (int64_t)p.N * p.b_sn >= ((int64_t)1 << 32)) return 0;  // per-lane weight offsets are 32-bit
  const int steps = p.K >> 7;
  static const int ph_cap = [] { const char* e = getenv("SGL_MI355_WSTREAM_PH"); return e ? atoi(e) : 32; }();  // tuning aid
  const int nblocks = (p.N + 15) / 16;
  // Pick (PH, consumer waves, K slices): one workgroup per CU (the A buffers fill most of the LDS) and a CU draws
  // only ~10 B/clk from HBM, so minimise the k-steps on the busiest CU (+ ~4 steps' worth of barrier per phase).
  // Without slabs: all of K per workgroup, longest phase (N = 28672: 7 consumers -> exactly 256 workgroups).
  int PH = 0, nc = 8, SK = 1, pps = 0, best = 1 << 30;
  // K / 128 not a multiple of 8 (e.g. 3584 = 28 steps): phases of 4 steps, four weight steps in flight
  // (phases of 2 steps for the split-K form -- 192 workgroups instead of 48 on the TP = 8 qkv shape 4096 x 768 -- were
  //  tried: one rank's TP = 4 / 8 step 3.07 -> 3.04 / 2.42 -> 2.45 ms, i.e. nothing; those shapes are launch-bound)
  const int ph_min = MB >= 16 ? 2 : (steps % 8 == 0 && MB <= 4) ? 8 : 4;  // (MB = 8 / 16: 4 / 2 steps are the 64-KiB buffer)
  for (int ph = 32 / MB; ph >= ph_min; ph >>= 1) {
    if (ph > ph_cap && ph > 8) continue;
    if (steps % ph != 0) continue;
    const int P = steps / ph;
    for (int c = 8; c >= 4; --c) {
      const int groups_c = (nblocks + c - 1) / c;
      int sk = 1, pp = P;
      if (slabs != nullptr) {
        int sk_max = 256 / groups_c;
        if (sk_max < 1) sk_max = 1;
        if (sk_max > P) sk_max = P;
        pp = (P + sk_max - 1) / sk_max;
        sk = (P + pp - 1) / pp;
      }
      const int rounds = (groups_c * sk + 255) / 256;
      const int cost = rounds * c * pp * (ph + (MB >= 16 ? 1 : 4));  // (a barrier per two steps is the only form at 256 rows)
      if (cost < best) { best = cost; PH = ph; nc = c; SK = sk; pps = pp; }
    }
    if (slabs == nullptr) break;  // the longest phase that divides K
  }
  {  // tuning aid: SGL_MI355_WSTREAM_FORCE="PH,nc" for the unsplit (no-slab) form
    static const char* force = getenv("SGL_MI355_WSTREAM_FORCE");
    int fph = 0, fnc = 0;
    if (force && slabs == nullptr && sscanf(force, "%d,%d", &fph, &fnc) == 2 && (fph == 4 || fph == 8 || fph == 16 || fph == 32) &&
        fph * MB <= 32 && steps % fph == 0 && fnc >= 1 && fnc <= 8) {
      PH = fph;
      nc = fnc;
      SK = 1;
      pps = steps / fph;
    }
  }
  {  // tuning aid: SGL_MI355_WSTREAM_SLAB_FORCE="PH,nc,SK" for the split-K (slab) form
    static const char* sforce = getenv("SGL_MI355_WSTREAM_SLAB_FORCE");
    int fph = 0, fnc = 0, fsk = 0;
    if (sforce && slabs != nullptr && sscanf(sforce, "%d,%d,%d", &fph, &fnc, &fsk) == 3 && (fph == 4 || fph == 8 || fph == 16 || fph == 32) &&
        fph * MB <= 32 && steps % fph == 0 && fnc >= 1 && fnc <= 8 && fsk >= 1) {
      const int P = steps / fph;
      PH = fph;
      nc = fnc;
      pps = (P + fsk - 1) / fsk;
      SK = (P + pps - 1) / pps;
    }
  }
  if (PH == 0) return 0;
  if (p.a16 != nullptr && (slabs == nullptr || pps != 1)) return 0;  // the a16 fill covers single-phase slices only
  const int groups = (nblocks + nc - 1) / nc;
  if (slabs != nullptr && ((SK < 2 && partial_slices == nullptr) || slab_floats < (int64_t)SK * p.M * p.N)) return 0;
  if (partial_slices != nullptr) *partial_slices = SK;
  const bool finalize = partial_slices == nullptr;
  used = true;
#define WS_GO(PH_)                                                                                         \
  return slabs ? launch_wstream_ph<OUT_DTYPE, MB, PH_, true>(p, slabs, SK, pps, nc, groups, s, finalize)    \
               : launch_wstream_ph<OUT_DTYPE, MB, PH_, false>(p, nullptr, 1, pps, nc, groups, s)
  if constexpr (MB == 1) { if (PH == 32) WS_GO(32); }
  if constexpr (MB <= 2) { if (PH == 16) WS_GO(16); }
  if constexpr (MB <= 8) { if (PH == 4) WS_GO(4); }
  if constexpr (MB <= 4) WS_GO(8);
  if constexpr (MB == 16) { if (PH == 2) WS_GO(2); }
  used = false;  // (MB = 8 / 16 have phases of four / two k-steps only)
  return 0;
#undef WS_GO
}

template <int OUT_DTYPE, int MB>
int launch_astat_direct(const GemmArgs& p, hipStream_t s, bool& used) {
  using L = AStat<MB>;
  used = false;
  const int steps_total = (p.K + 127) >> 7;
  const int nblocks = (p.N + 15) / 16;
  int max_steps = L::MAX_STEPS & ~7;  // multiple of the prefetch depth
  int SK = (steps_total + max_steps - 1) / max_steps;
  int steps_per_slice = (((steps_total + SK - 1) / SK) + 7) & ~7;
  if (steps_per_slice > max_steps) return 0;
  SK = (steps_total + steps_per_slice - 1) / steps_per_slice;
  if ((int64_t)p.N * p.b_sn >= ((int64_t)1 << 32)) return 0;  // per-lane weight offsets are 32-bit
  const int ns_last = steps_total - (SK - 1) * steps_per_slice;
  const bool exact = (p.K & 127) == 0 && (ns_last & 7) == 0;
  auto kern = exact ? fp8_gemm_astat_direct_kernel<OUT_DTYPE, MB, true> : fp8_gemm_astat_direct_kernel<OUT_DTYPE, MB, false>;
  static int attr_rc = [] {
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(fp8_gemm_astat_direct_kernel<OUT_DTYPE, MB, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                       "hipFuncSetAttribute");
    if (rc) return rc;
    return check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(fp8_gemm_astat_direct_kernel<OUT_DTYPE, MB, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                     "hipFuncSetAttribute");
  }();
  if (attr_rc) return attr_rc;
  int lds = steps_per_slice * L::STEP_BYTES;
  const int ep_bytes = 8 * L::ROWS * 24 * 2;
  if (lds < ep_bytes) lds = ep_bytes;
  // waves per workgroup: one workgroup per CU (the A image fills the LDS), so pick the count that leaves
  // the fewest column blocks on the busiest CU (N = 28672: 7 waves -> 256 workgroups, 8 -> only 224).
  // (measured at N = 28672: M = 16 28.0 us with 7 waves vs 29.0 with 8; M = 64 prefers 8 waves: 39.9 vs 42.3)
  int nw = 8, best = 1 << 30;
  for (int c = 8; c >= (MB == 4 ? 8 : 4); --c) {
    const int blocks = (nblocks + c - 1) / c;
    const int cost = ((blocks + 255) / 256) * c;
    if (cost < best) { best = cost; nw = c; }
  }
  g_last_kernel = "astat_direct";
  hipLaunchKernelGGL(kern, dim3((unsigned)((nblocks + nw - 1) / nw)), dim3(64 * nw), lds, s, p, SK, steps_per_slice);
  used = true;
  return check_hip(hipGetLastError(), "fp8_gemm_astat_direct launch");
}

// epilogue of the A-stationary kernel: sum the K-slices, then the reference's epilogue
// (fp8_gemm_kernel.cu:498-546): * w_scale[col], * x_scale[row], + bias, cast.
template <int OUT_DTYPE>
__global__ __launch_bounds__(256) void fp8_gemm_finalize_kernel(GemmArgs p, const float* slabs, int SK) {
  using H = Half16<OUT_DTYPE>;
  using T = typename H::T;
  const int64_t total = (int64_t)p.M * p.N / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t sstride = (int64_t)p.M * p.N;
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
    const int m = (int)(e / p.N), n = (int)(e - (int64_t)m * p.N);
    const float sa = p.sa[m];
    typename H::x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // explicit roundings (no fma contraction): the fused *_from_partials consumers repeat exactly this
#pragma clang fp contract(off)
      float r = (v[j] * p.sb[n + j]) * sa;
      if (p.bias) r = r + H::to_f32(reinterpret_cast<const T*>(p.bias)[n + j]);
      o[j] = H::from_f32(r);
    }
    *reinterpret_cast<typename H::x8*>(reinterpret_cast<T*>(p.out) + e) = o;
  }
}

template <int OUT_DTYPE, int MB>
int launch_astat(const GemmArgs& p, float* slabs, int64_t slab_floats, hipStream_t s, bool& used) {
  using L = AStat<MB>;
  const int steps_total = (p.K + 127) >> 7;
  const int nblocks = (p.N + 15) / 16;
  // K-slices: at least enough for the slice of A to fit LDS; more when N is narrow, so that there are
  // >= ~2048 (column block, slice) items for 256 CUs x 8+ waves -- but never below 4 k-steps per item.
  int SK = (steps_total + L::MAX_STEPS - 1) / L::MAX_STEPS;
  const int want = (2048 + nblocks - 1) / nblocks;
  if (SK < want) SK = want;
  if (SK > steps_total / 4) SK = steps_total / 4;
  if (SK < 1) SK = 1;
  const int steps_per_slice = (steps_total + SK - 1) / SK;
  SK = (steps_total + steps_per_slice - 1) / steps_per_slice;
  used = false;
  if (steps_per_slice > L::MAX_STEPS) return 0;
  if (slabs == nullptr || slab_floats < (int64_t)SK * p.M * p.N) return 0;
  int ngroups = 256 / SK;
  if (ngroups > nblocks) ngroups = nblocks;
  if (ngroups < 1) ngroups = 1;
  int nwaves = (nblocks + ngroups - 1) / ngroups;
  if (nwaves > kAsWaves) nwaves = kAsWaves;
  auto kern = fp8_gemm_astat_kernel<MB>;
  const int lds = steps_per_slice * L::STEP_BYTES;
  static int attr_rc = check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
      "hipFuncSetAttribute");
  if (attr_rc) return attr_rc;
  g_last_kernel = "astat";
  hipLaunchKernelGGL(kern, dim3((unsigned)(ngroups * SK)), dim3(64 * nwaves), lds, s, p, slabs, SK, steps_per_slice);
  int rc = check_hip(hipGetLastError(), "fp8_gemm_astat launch");
  if (rc) return rc;
  const int64_t total = (int64_t)p.M * p.N / 8;
  hipLaunchKernelGGL((fp8_gemm_finalize_kernel<OUT_DTYPE>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p,
                     (const float*)slabs, SK);
  used = true;
  return check_hip(hipGetLastError(), "fp8_gemm_finalize launch");
}

// ------------------------------------------------------------------------------------------
// tiled: 128x128x128, 4 waves as 2x2, each wave 64x64 (4x4 fragments of 16x16).
constexpr int kTM = 128, kTN = 128, kTK = 128;
constexpr int kRegion = kTM * 32 + 16;      // one 32-B k-chunk column of the tile, +16 B skew
constexpr int kOperand = 4 * kRegion + 48;  // 16512 B, keeps 16-B alignment
constexpr int kStageBytes = 2 * kOperand;   // A + B

// SCALED: one v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, E8M0 block scales fixed at 2^0) per 128-k step instead of
// four v_mfma_f32_16x16x32_fp8_fp8.  Same products and fp32 accumulation, but the block-scaled instruction runs at twice
// the FP8 rate on gfx950 (MI355X_MICROARCH.md: non-scaled FP8 MFMA = the BF16 rate, 2.5 PFLOP/s; scaled = 5 PFLOP/s).
// The lane's 32 contiguous k-bytes of the fragment-major LDS image are exactly one operand of it.
typedef int v8i32_t __attribute__((ext_vector_type(8)));

template <int OUT_DTYPE, bool SCALED>
__global__ __launch_bounds__(256) void fp8_gemm_tiled_kernel(GemmArgs p) {
  using H = Half16<OUT_DTYPE>;
  using T = typename H::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;

  // XCD-aware tile order: blocks that share an XCD (blockIdx % 8) walk neighbouring tiles, so the
  // A / W panels they share are L2 hits.  Bijective for any grid size.
  const int tiles_m = (p.M + kTM - 1) / kTM, tiles_n = (p.N + kTN - 1) / kTN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // walk M fastest inside a column panel of W so the weight panel stays hot
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * kTM, n0 = tn * kTN;

  // global -> register staging map: 4 x 16 B per thread per operand
  int ld_row[4], ld_j[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    ld_row[i] = idx >> 3;
    ld_j[i] = idx & 7;
  }
  const uint8_t* a_ptr[4];
  const uint8_t* b_ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + ld_row[i];
    m = m < p.M ? m : p.M - 1;  // rows past the edge re-read a valid row; never stored
    int n = n0 + ld_row[i];
    n = n < p.N ? n : p.N - 1;
    a_ptr[i] = p.a + (int64_t)m * p.a_sm + 16 * ld_j[i];
    b_ptr[i] = p.b + (int64_t)n * p.b_sn + 16 * ld_j[i];
  }
  uint4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // unconditional raw loads (see ld16); a_ptr/b_ptr already include 16*ld_j; masked in lstore
      ra[i] = ld16(a_ptr[i], k0, p.K - 16 * ld_j[i]);
      rb[i] = ld16(b_ptr[i], k0, p.K - 16 * ld_j[i]);
    }
  };
  const bool ktail = (p.K & 127) != 0;
  auto lstore = [&](int stage, int k0) {
    char* sa_ = smem + stage * kStageBytes;
    char* sb_ = sa_ + kOperand;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (ld_j[i] >> 1) * kRegion + ld_row[i] * 32 + (ld_j[i] & 1) * 16;
      uint4 va = ra[i], vb = rb[i];
      if (ktail) {
        va = mask16(va, k0, p.K - 16 * ld_j[i]);
        vb = mask16(vb, k0, p.K - 16 * ld_j[i]);
      }
      *reinterpret_cast<uint4*>(sa_ + off) = va;
      *reinterpret_cast<uint4*>(sb_ + off) = vb;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + kTK - 1) / kTK;
  gload(0);
  lstore(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * kTK);  // in flight under this tile's MFMAs
    const char* sa_ = smem + st * kStageBytes + g * kRegion;
    const char* sb_ = sa_ + kOperand;
    Frag32 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* pa = sa_ + (wm * 64 + 16 * i + r16) * 32;
      af[i].v[0] = *reinterpret_cast<const uint4*>(pa);
      af[i].v[1] = *reinterpret_cast<const uint4*>(pa + 16);
      const char* pb = sb_ + (wn * 64 + 16 * i + r16) * 32;
      bf[i].v[0] = *reinterpret_cast<const uint4*>(pb);
      bf[i].v[1] = *reinterpret_cast<const uint4*>(pb + 16);
    }
    if constexpr (SCALED) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              __builtin_bit_cast(v8i32_t, af[i]), __builtin_bit_cast(v8i32_t, bf[j]), acc[i][j], 0 /* A: e4m3 */,
              0 /* B: e4m3 */, 0, 0x7F7F7F7F /* scale 2^0 */, 0, 0x7F7F7F7F);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af[i].l[ks], bf[j].l[ks], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(st ^ 1, (kt + 1) * kTK);
    __syncthreads();
  }

  // ---- epilogue: scale in fp32, convert, transpose through LDS, 16-B row-segment stores
  // acc[i][j][r] = C[m0 + wm*64 + 16i + 4g + r][n0 + wn*64 + 16j + r16]
  T* ep = reinterpret_cast<T*>(smem) + wave * (64 * 72);  // per wave [64][64] (+8 pad), 9216 B
  float sbv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + 16 * j + r16;
    sbv[j] = n < p.N ? p.sb[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ml = 16 * i + 4 * g + r;
      const int m = m0 + wm * 64 + ml;
      const float sa = m < p.M ? p.sa[m] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nl = 16 * j + r16;
        float v = acc[i][j][r] * sbv[j] * sa;
        if (p.bias) {
          const int n = n0 + wn * 64 + nl;
          if (n < p.N) v += H::to_f32(reinterpret_cast<const T*>(p.bias)[n]);
        }
        ep[ml * 72 + nl] = H::from_f32(v);
      }
    }
  // wave-private region: a wave-level LDS wait is enough (no cross-wave sharing here)
  wait_lgkmcnt0();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = lane + 64 * it;  // 512 chunks of 8 columns
    const int ml = c >> 3, nl = (c & 7) * 8;
    const int m = m0 + wm * 64 + ml, n = n0 + wn * 64 + nl;
    if (m < p.M && n < p.N)  // N % 8 == 0, so a chunk is all-in or all-out
      *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.N + n) =
          *reinterpret_cast<const uint4*>(ep + ml * 72 + nl);
  }
}

// Epilogue of the tiled kernels: each wave scales its (16 RI) x 64 accumulator tile, adds the bias and stores it in passes
// of 64 rows through a wave-private [64][64] (+8 pad) LDS patch (9216 B) that turns the MFMA layout into 16-B row segments.
template <int OUT_DTYPE, int RI, int CB = 4>
__device__ __forceinline__ void tiled_epilogue(const GemmArgs& p, char* smem, f32x4 (&acc)[RI][CB], int m0, int n0, int wm,
                                               int wn, int wave, int lane) {
  // CB: 16-column blocks per wave (4: the 64-column wave tile; 3: the 48-column one of the 192-wide tiled3 tiles)
  using H = Half16<OUT_DTYPE>;
  using T = typename H::T;
  const int r16 = lane & 15, g = lane >> 4;
  T* ep = reinterpret_cast<T*>(smem) + wave * (64 * 72);
  float sbv[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int n = n0 + wn * 16 * CB + 16 * j + r16;
#ifndef SGLM_EPI_ABL_NOSCALE
#define SGLM_EPI_ABL_NOSCALE 0  // timing ablation (WRONG RESULTS): the tiled epilogues without their scale loads
#endif
#if SGLM_EPI_ABL_NOSCALE
    sbv[j] = 1.0f + (float)n * 1e-9f;
#else
    sbv[j] = p.sb[n < p.N ? n : p.N - 1];
#endif
  }
  constexpr int RP = RI < 4 ? RI : 4;  // 16-row fragments per epilogue pass
#pragma unroll
  for (int pass = 0; pass < RI / RP; ++pass) {
    const int mw0 = m0 + wm * 16 * RI + 16 * RP * pass;
#pragma unroll
    for (int i = 0; i < RP; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ml = 16 * i + 4 * g + r;
        const int m = mw0 + ml;
#if SGLM_EPI_ABL_NOSCALE
        const float sa = 1.0f + (float)m * 1e-9f;
#else
        const float sa = p.sa[m < p.M ? m : p.M - 1];
#endif
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          const int nl = 16 * j + r16;
          float v;
          {
#pragma clang fp contract(off)  // two products and a sum, never an fma: tiled_epilogue_silu must give the same bits
            v = acc[RP * pass + i][j][r] * sbv[j] * sa;
            if (p.bias) {
              const int n = n0 + wn * 16 * CB + nl;
              v = v + H::to_f32(reinterpret_cast<const T*>(p.bias)[n < p.N ? n : p.N - 1]);
            }
          }
          ep[ml * 72 + nl] = H::from_f32(v);
        }
      }
    wait_lgkmcnt0();  // wave-private patch: a wave-level LDS wait is enough
    constexpr int SEG = 2 * CB;              // 16-byte row segments per patch row
    constexpr int ITEMS = 16 * RP * SEG;     // segments of this pass
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

// Epilogue of the gate_up GEMM with SiLU(gate) * up folded in (fp8_gemm_tiled3_kernel, SILU): a wave holds gate columns
// n0 + 32 wn + [0, 32) in column blocks 0, 1 and the up columns I + (the same) in blocks 2, 3, i.e. gate and up of one output
// element sit in the same lane.  Both are finished and rounded to the 16-bit dtype as tiled_epilogue would store them, then
// out = silu(g) * u with silu_mul_kernel's two roundings (elementwise.hip; activation.cu:56-60) -- the same bits as the GEMM
// followed by sgl_mi355_silu_and_mul.  out is [M][I], I = N / 2; 64-row passes through a wave-private [64][32] (+8 pad) patch.
template <int OUT_DTYPE, int RI>
__device__ __forceinline__ void tiled_epilogue_silu(const GemmArgs& p, char* smem, f32x4 (&acc)[RI][4], int m0, int n0, int wm,
                                                    int wn, int wave, int lane) {
  using H = Half16<OUT_DTYPE>;
  using T = typename H::T;
  const int r16 = lane & 15, g = lane >> 4;
  const int I = p.N >> 1;
  T* ep = reinterpret_cast<T*>(smem) + wave * (64 * 72);
  float sbg[2], sbu[2], bg[2], bu[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int n = n0 + wn * 32 + 16 * j + r16;
    n = n < I ? n : I - 1;
#if SGLM_EPI_ABL_NOSCALE
    sbg[j] = 1.0f + (float)n * 1e-9f;
    sbu[j] = 1.0f + (float)n * 2e-9f;
#else
    sbg[j] = p.sb[n];
    sbu[j] = p.sb[I + n];
#endif
    bg[j] = p.bias ? H::to_f32(reinterpret_cast<const T*>(p.bias)[n]) : 0.f;
    bu[j] = p.bias ? H::to_f32(reinterpret_cast<const T*>(p.bias)[I + n]) : 0.f;
  }
  constexpr int RP = RI < 4 ? RI : 4;
#pragma unroll
  for (int pass = 0; pass < RI / RP; ++pass) {
    const int mw0 = m0 + wm * 16 * RI + 16 * RP * pass;
#pragma unroll
    for (int i = 0; i < RP; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ml = 16 * i + 4 * g + r;
        const int m = mw0 + ml;
#if SGLM_EPI_ABL_NOSCALE
        const float sa = 1.0f + (float)m * 1e-9f;
#else
        const float sa = p.sa[m < p.M ? m : p.M - 1];
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // (acc * w_scale) * x_scale, then + bias as an operation of its own, as in tiled_epilogue (whose addition sits in
          // another basic block): left to itself hipcc contracts the product and the sum here into one fma, one ulp off now and then
          float gv, uv;
          {
#pragma clang fp contract(off)
            gv = acc[RP * pass + i][j][r] * sbg[j] * sa;
            uv = acc[RP * pass + i][j + 2][r] * sbu[j] * sa;
            if (p.bias) {
              gv = gv + bg[j];
              uv = uv + bu[j];
            }
          }
          const float af = H::to_f32(H::from_f32(gv));
          const float sl = H::to_f32(H::from_f32(af / (1.0f + __expf(-af))));
          ep[ml * 40 + 16 * j + r16] = H::from_f32(sl * H::to_f32(H::from_f32(uv)));
        }
      }
    wait_lgkmcnt0();  // wave-private patch: a wave-level LDS wait is enough
    constexpr int ITEMS = 16 * RP * 4;  // 16-byte segments of this pass (four per 32-column row)
#pragma unroll
    for (int it = 0; it < (ITEMS + 63) / 64; ++it) {
      const int c = lane + 64 * it;
      const int ml = c >> 2, nl = (c & 3) * 8;
      const int m = mw0 + ml, n = n0 + wn * 32 + nl;
      if (c < ITEMS && m < p.M && n < I)
        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + (int64_t)m * I + n) =
            *reinterpret_cast<const uint4*>(ep + ml * 40 + nl);
    }
    wait_lgkmcnt0();  // the patch is rewritten by the next pass
  }
}

// ------------------------------------------------------------------------------------------
// tiled v2 (K % 128 == 0): the same 128x128x128 tile and 2x2 waves, but
//   * operands go global -> LDS by LDS-DMA (no VGPR staging, no ds_write), NSTAGE stages, prefetch distance NSTAGE-1;
//   * the LDS image is [row][128 B] with the 16-B chunk index XOR-swizzled by (row >> 1) & 7 -- a DMA instruction
//     lands 8 rows x 128 B from 8 coalesced row segments (swizzle on the source side), and the fragment
//     ds_read_b128 of 16 rows hits 16 distinct bank groups;
//   * one block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (scales 2^0) per fragment pair and k-step.
// Block tile = (16 RI WM) x (64 WN) x 128 with WM x WN waves, each wave 16 RI rows x 64 columns:
//   RI 4, 2x2 waves: 128x128 (2 stages -> two workgroups per CU; 3 stages when there is at most one tile per CU)
//   RI 8, 2x2 waves: 256x128, 3 stages
//   RI 4, 4x2 waves: 256x128 with 8 waves, 3 stages (B staged once per 256 rows)
//   RI 8, 2x4 waves: 256x256 with 8 waves, 2 stages (0.75x the LDS fragment reads per flop of the 64x64 wave tile)
template <int OUT_DTYPE, int NSTAGE, int RI, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void fp8_gemm_tiled2_kernel(GemmArgs p) {
  constexpr int NW = WM * WN;
  constexpr int TMB = 16 * RI * WM;     // block rows
  constexpr int TNB = 64 * WN;          // block columns
  constexpr int OPA = TMB * 128;        // A tile: TMB rows x 128 B
  constexpr int OPB = TNB * 128;
  constexpr int STAGE = OPA + OPB;
  constexpr int UA = TMB / 8 / NW;      // 1-KiB DMA units (8 rows x 128 B) per wave and stage
  constexpr int UB = TNB / 8 / NW;
  static_assert((TMB / 8) % NW == 0 && (TNB / 8) % NW == 0, "DMA units must divide over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r16 = lane & 15, g = lane >> 4;

  const int tiles_m = (p.M + TMB - 1) / TMB, tiles_n = (p.N + TNB - 1) / TNB;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * TMB, n0 = tn * TNB;

  // DMA: wave w moves units UA*w .. of A and UB*w .. of B; lane i of a unit lands at chunk i & 7 of row i >> 3 and
  // fetches source chunk (i & 7) ^ ((row >> 1) & 7) of that row
  const uint8_t* a_src[UA];
  const uint8_t* b_src[UB];
#pragma unroll
  for (int u = 0; u < UA; ++u) {
    const int row = (UA * wave + u) * 8 + (lane >> 3);  // tile-local row
    const int j = (lane & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;  // rows past the edge re-read a valid row; never stored
    a_src[u] = p.a + (int64_t)m * p.a_sm + 16 * j;
  }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    if (p.b_shuf) {
      // pre-shuffled W: the 2-KiB piece of (16-column block, k-step) is copied VERBATIM, one contiguous KiB per DMA
      // instruction; the B image of a stage is then [block][half][g][r16][16 B] and a fragment read of 16 rows x 16 B is
      // 256 contiguous bytes (no swizzle needed)
      const int q = UB * wave + u;  // tile-local KiB: block q >> 1, half q & 1
      int nb = (n0 >> 4) + (q >> 1);
      nb = nb < (p.N >> 4) ? nb : (p.N >> 4) - 1;
      b_src[u] = p.b + (int64_t)nb * 16 * p.K + (q & 1) * 1024 + lane * 16;
    } else {
      const int row = (UB * wave + u) * 8 + (lane >> 3);
      const int j = (lane & 7) ^ ((row >> 1) & 7);
      int n = n0 + row;
      n = n < p.N ? n : p.N - 1;
      b_src[u] = p.b + (int64_t)n * p.b_sn + 16 * j;
    }
  }
  const int b_step = p.b_shuf ? 2048 : 128;
  const uint32_t smem_base = lds_addr_of(smem);
  auto dma_stage = [&](int stage, int kt) __attribute__((always_inline)) {
    const uint32_t dst = smem_base + stage * STAGE;
#pragma unroll
    for (int u = 0; u < UA; ++u) lds_dma16(a_src[u] + (int64_t)kt * 128, dst + (UA * wave + u) * 1024);
#pragma unroll
    for (int u = 0; u < UB; ++u) lds_dma16(b_src[u] + (int64_t)kt * b_step, dst + OPA + (UB * wave + u) * 1024);
  };

  f32x4 acc[RI][4];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment byte offsets inside an operand tile: row (.. + 16 i + r16), chunks 2g and 2g+1
  const int sw = (r16 >> 1) & 7;  // (row >> 1) & 7 with row = 16 x + r16
  const uint32_t c0 = 16 * ((2 * g) ^ sw), c1 = 16 * ((2 * g + 1) ^ sw);
  const uint32_t a_row = (wm * 16 * RI + r16) * 128;
  // B fragment i (16 columns): row-major image as A; verbatim pre-shuffled image: block wn * 4 + i, chunks 2g / 2g + 1
  // = (half g >> 1, lane group (2g) & 3 and the next one), row r16
  const uint32_t b_row = p.b_shuf ? wn * 4 * 2048 + r16 * 16 : (wn * 64 + r16) * 128;
  const uint32_t bc0 = p.b_shuf ? (g >> 1) * 1024 + ((2 * g) & 3) * 256 : c0;
  const uint32_t bc1 = p.b_shuf ? bc0 + 256 : c1;

  const int nk = p.K >> 7;
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st)
    if (st < nk) dma_stage(st, st);
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's DMAs of stage kt have landed when at most the younger stage's (UA + UB) are outstanding
    const int younger = (nk - 1 - kt) < (NSTAGE - 2) ? (nk - 1 - kt) : (NSTAGE - 2);
    if (younger >= 1) wait_vmcnt<UA + UB>();  // NSTAGE == 3
    else wait_vmcnt<0>();
    __syncthreads();  // everyone's DMAs of stage kt landed; everyone finished reading the stage refilled below
    if (kt + NSTAGE - 1 < nk) dma_stage((kt + NSTAGE - 1) % NSTAGE, kt + NSTAGE - 1);
    const char* sa_ = smem + (kt % NSTAGE) * STAGE;
    const char* sb_ = sa_ + OPA;
    // fragment pipeline: B (4 frags) and the first two A frags up front, then the reads of A frag i+2 are issued
    // behind the MFMAs of A frag i (sched_group_barrier pins that interleave; left alone, hipcc loads each A frag
    // right before its MFMAs and exposes one LDS latency per 8 MFMAs)
    Frag32 af[RI], bf[4];
    // source order = the order the scheduling groups below consume the reads: A0, B0..B3, A1, A2, ...
    af[0].v[0] = *reinterpret_cast<const uint4*>(sa_ + a_row + c0);
    af[0].v[1] = *reinterpret_cast<const uint4*>(sa_ + a_row + c1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf[i].v[0] = *reinterpret_cast<const uint4*>(sb_ + b_row + i * 2048 + bc0);
      bf[i].v[1] = *reinterpret_cast<const uint4*>(sb_ + b_row + i * 2048 + bc1);
    }
#pragma unroll
    for (int i = 1; i < RI; ++i) {
      af[i].v[0] = *reinterpret_cast<const uint4*>(sa_ + a_row + i * 2048 + c0);
      af[i].v[1] = *reinterpret_cast<const uint4*>(sa_ + a_row + i * 2048 + c1);
    }
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
            __builtin_bit_cast(v8i32_t, af[i]), __builtin_bit_cast(v8i32_t, bf[j]), acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0,
            0x7F7F7F7F);
    // the first MFMA starts after 3 fragments (A0, B0, B1), not after all 6 of the step's first burst
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);  // A0, B0, B1
#pragma unroll
    for (int j = 0; j < 4; ++j) {                        // row 0: one MFMA, then the reads of B2, B3, A1, A2
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (j < 2 || j - 1 < RI) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
#pragma unroll
    for (int i = 1; i < RI; ++i) {                       // rows 1..: the A fragment two rows ahead, then the row's MFMAs
      if (i + 2 < RI) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  }
  __syncthreads();  // all stages dead: the epilogue reuses the memory

  tiled_epilogue<OUT_DTYPE, RI>(p, smem, acc, m0, n0, wm, wn, wave, lane);
}

// Tiled v3 (pre-shuffled weights only): the B operand never touches LDS.  In the fragment-major layout the 32 bytes a lane
// feeds to one block-scaled MFMA are two 16-B pieces 1 KiB apart, and a wave's load instruction covers one contiguous KiB,
// so every wave streams the weight fragments of its own 64 columns global -> VGPR (each fragment refilled one k-step ahead
// right after its last MFMA, hand-counted vmcnt as in the weight-streaming decode kernel) while only A goes global -> LDS.  Per k-step of a
// 256 x 256 tile that leaves 128 KiB of LDS fragment reads + 32 KiB of DMA writes (v2: 192 + 64), i.e. the LDS pipe is
// busy for less than half of the step's MFMA time, at the price of each weight byte crossing L2 -> CU once per wave row
// (WM times).  k permutation inside a step: lane group g holds bytes [16g, +16) and [64 + 16g, +16) -- on both operands.
// Shapes: 256 x 256 with 8 waves (2 x 4), or 128 x 256 with 4 waves (1 x 4: no weight byte is loaded twice) and TWO
// workgroups per CU, whose barriers and first-fragment latencies then overlap each other's MFMAs.
// CB (round 3): 16-column blocks per wave.  4 = the 128 x 256 tile; 3 = a 128 x 192 tile for widths whose 256-column tiling
// leaves a ragged last round (qkv 4096 -> 6144: 768 tiles of 256 on 512 slots = 1.5 rounds, 1024 tiles of 192 = exactly 2
// rounds of three quarters the work; at M = 1024 256 tiles instead of 192: every CU has one).
// SILU (round 3, the gate_up GEMM of a gated MLP, W = [gate | up], N = 2 I): the tile is 128 rows x 128 OUTPUT columns -- every
// wave streams the weight fragments of 32 gate columns and of the 32 up columns I further on (the same 2 KiB contiguous pieces,
// just other blocks), and the epilogue writes silu(gate) * up, [M][I], instead of the [M][2 I] product (tiled_epilogue_silu).
// KS = 2 (round 4): TWO wave groups per workgroup, group kg taking the k-steps 2 i + kg of the SAME tile (its own A image per
// step, its own weight fragments: no byte enters the CU twice) and handing its accumulators to group 0 through LDS at the end.
// For the shapes that give every CU exactly one 128-row tile (M = 1024 qkv: 256 tiles of 128 x 192), where the four-wave
// workgroup leaves one wave per SIMD and every LDS / barrier / first-fragment latency is exposed; two co-resident waves per
// SIMD are what the two-workgroups-per-CU form has from 512 tiles on.  Sums the k-steps in another order than KS = 1.
// RAWK (round 5): the K dimension is ALSO cut over workgroups (blockIdx.y = slice) and the accumulators leave as raw fp32 partial
// sums -- the prefill counterpart of the decode streamers' split-K slabs, for narrow outputs with a long K (down_proj at 1024
// rows: 128 tiles of 128 x 256 x two slices fill the chip with HALF the operand bytes per CU of the 256 tiles of 128 x 128).
// The epilogue belongs to the consumer (RMSNorm from partials through deferred.py).
template <int OUT_DTYPE, int NSTAGE, int RI, int WM, int WN, int CB = 4, bool SILU = false, int KS = 1, bool RAWK = false>
__global__ __launch_bounds__(64 * WM * WN * KS, 2) void fp8_gemm_tiled3_kernel(GemmArgs p) {
  static_assert(!RAWK || (KS == 1 && !SILU), "the raw split-K form is the plain one-group kernel");
  static_assert(NSTAGE >= 3, "the wait count below assumes A(kt) was issued before B(kt)");
  static_assert(CB == 3 || CB == 4, "column blocks per wave");
  static_assert(!SILU || CB == 4, "gate and up: two column blocks each");
  static_assert(KS == 1 || (KS == 2 && !SILU), "one or two k groups");
  constexpr int NW = WM * WN * KS;
  constexpr int TMB = 16 * RI * WM;     // block rows
  constexpr int TNB = SILU ? 8 * CB * WN : 16 * CB * WN;  // block columns (SILU: output columns = gate columns)
  constexpr int OPA = TMB * 128;        // one A image: TMB rows x 128 B
  constexpr int STAGE = KS * OPA;       // a stage holds the images of KS consecutive k-steps
  constexpr int UA = KS * TMB / 8 / NW; // 1-KiB DMA units (8 rows x 128 B) per wave and stage
  static_assert((KS * TMB / 8) % NW == 0, "DMA units must divide over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef SGLM_T3_TIMING
#define SGLM_T3_TIMING 0  // instrumentation build (tools/exp/gemm_phase_times.py): `bias` is a uint64 buffer, four 100-MHz timestamps per workgroup
#endif
#if SGLM_T3_TIMING
  const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
  uint64_t t_loop = 0, t_loop_end = 0;
#endif
  const int kg = wave / (WM * WN);      // k group of this wave
  const int wave_t = wave % (WM * WN);  // its place in the tile
  const int wm = wave_t / WN, wn = wave_t % WN;
  const int r16 = lane & 15, g = lane >> 4;

  const int n_cols = SILU ? p.N >> 1 : p.N;  // columns the tiles cover
  const int tiles_m = (p.M + TMB - 1) / TMB, tiles_n = (n_cols + TNB - 1) / TNB;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // Rasterisation: tiles go in groups of raster_gn column tiles, row tile fastest across the group's columns, so the
  // workgroups resident on an XCD at any time (a contiguous run of `bid`) form a compact block of tiles and both operands
  // are shared through that XCD's L2 (with one column per group every resident workgroup streams its own rows of A).
  int tm, tn;
  {
    const int gsz = p.raster_gn * tiles_m;
    const int grp = bid / gsz, r = bid - grp * gsz;
    const int left = tiles_n - grp * p.raster_gn;
    const int gn = left < p.raster_gn ? left : p.raster_gn;
    tm = r / gn;
    tn = grp * p.raster_gn + (r - tm * gn);
  }
  const int m0 = tm * TMB, n0 = tn * TNB;

  const uint8_t* a_src[UA];
#pragma unroll
  for (int u = 0; u < UA; ++u) {
    const int q = UA * wave + u;                         // unit of the stage: image q / (TMB / 8), rows 8 (q % (TMB / 8))..
    const int row = (q % (TMB / 8)) * 8 + (lane >> 3);   // tile-local row
    const int j = (lane & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;  // rows past the edge re-read a valid row; never stored
    a_src[u] = p.a + (int64_t)m * p.a_sm + 16 * j + (q / (TMB / 8)) * 128;
  }
  const int kstep0 = RAWK ? (int)blockIdx.y * p.k_steps_per_slice : 0;  // first 128-byte k-step of this workgroup's K slice
  if constexpr (RAWK) {
#pragma unroll
    for (int u = 0; u < UA; ++u) a_src[u] += (int64_t)kstep0 * 128;
  }
  const uint32_t smem_base = lds_addr_of(smem);
  // (kt below counts loop iterations: KS k-steps each)
  auto dma_stage = [&](int stage, int kt) __attribute__((always_inline)) {
    const uint32_t dst = smem_base + stage * STAGE;
#pragma unroll
    for (int u = 0; u < UA; ++u) lds_dma16(a_src[u] + (int64_t)kt * (128 * KS), dst + (UA * wave + u) * 1024);
  };
  // weight fragments: column block (n0 / 16 + 4 wn + j), k-step kt = 2 KiB at block * 16 K + 2048 kt; lane i takes bytes
  // 16 i.. of each KiB.  Blocks past N re-read the last one (their columns are never stored).
  const uint8_t* b_blk[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    int nb;
    if constexpr (SILU) {  // blocks 0, 1: gate columns n0 + 32 wn + [0, 32); blocks 2, 3: the up columns n_cols further on
      nb = (n0 >> 4) + wn * 2 + (j & 1);
      nb = (nb < (n_cols >> 4) ? nb : (n_cols >> 4) - 1) + (j >> 1) * (n_cols >> 4);
    } else {
      nb = (n0 >> 4) + wn * CB + j;
      nb = nb < (p.N >> 4) ? nb : (p.N >> 4) - 1;
    }
    b_blk[j] = p.b + (int64_t)nb * 16 * p.K + (int64_t)kstep0 * 2048;
  }
  const int nk = RAWK ? p.k_steps_per_slice : (p.K >> 7) / KS;  // loop iterations
  auto load_b = [&](Frag32 (&q)[CB], int kt) __attribute__((always_inline)) {
    const uint32_t voff = (uint32_t)lane * 16 + (uint32_t)(KS * kt + kg) * 2048;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(q[j].x[0]) : "v"(voff), "s"(b_blk[j]) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(q[j].x[1]) : "v"(voff), "s"(b_blk[j]) : "memory");
    }
  };

  f32x4 acc[RI][CB];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A fragment offsets inside the stage: row (16 (RI wm + i) + r16), chunks g and 4 + g, swizzled as the DMA wrote them
  const int sw = (r16 >> 1) & 7;
  const uint32_t a_row = (wm * 16 * RI + r16) * 128;
  const uint32_t c0 = 16 * (g ^ sw), c1 = 16 * ((4 + g) ^ sw);

  // Register budget (two waves per SIMD = 256 per lane): 128 accumulators + 64 for the step's RI A fragments + 32 for ONE
  // set of weight fragments.  The MFMAs therefore go column by column: once the RI MFMAs of column block j are issued,
  // its fragment registers are refilled with the NEXT step's bytes, which have a whole step to arrive.
  // VMEM issue order per step: A DMAs of step kt + 2 (UA per wave), then refills R0..R3 (two loads each).  Loads retire in
  // order, so in front of column j "R_j of the previous step has landed" is vmcnt(6 + UA): R_j+1.. of the previous step,
  // this step's DMAs, this step's R_0..j-1 (vmcnt(6) in front of column 0 when the DMAs are issued behind it,
  // SGLM_T3_DMA_LATE -- a stricter count is never wrong).  The same wait, at column 3, also proves this wave's DMAs of
  // step kt + 1 (older than the previous step's R_3), which is what the next barrier publishes.  Tail steps re-issue the last step's DMAs into
  // the dead stage and re-read the last weights, so the counts never change.
  Frag32 bq[CB];
  constexpr int YB = 2 * (CB - 1);  // loads of the OTHER column blocks' refills (6 with four blocks per wave)
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) dma_stage(st, st < nk ? st : nk - 1);
  load_b(bq, 0);
  wait_vmcnt<(NSTAGE - 2) * UA + 2 * CB>();  // stage 0 landed (this wave's part)
#ifndef SGLM_T3_ABL
#define SGLM_T3_ABL 0  // timing ablations (WRONG RESULTS): 1 no barrier, 2 A fragments read once, 3 no weight refills, 4 no DMA, 5 DMA always of k-step 0 (L2-resident), 6 the flops as 32x32x64 MFMAs
#endif
#if SGLM_T3_ABL == 2
  Frag32 af[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    af[i].v[0] = *reinterpret_cast<const uint4*>(smem + a_row + i * 2048 + c0);
    af[i].v[1] = *reinterpret_cast<const uint4*>(smem + a_row + i * 2048 + c1);
  }
#endif
#if SGLM_T3_ABL == 6
  static_assert(CB == 4, "ablation 6 is written for four column blocks");
  f32x16 acc32[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc32[m][n][r] = 0.f;
#endif
#if SGLM_T3_TIMING
  t_loop = __builtin_amdgcn_s_memrealtime();
#endif
#pragma clang loop unroll(disable)
  for (int kt = 0; kt < nk; ++kt) {
#if SGLM_T3_ABL != 1
    __syncthreads();  // everyone's DMAs of stage kt landed; everyone finished reading the stage refilled below
#endif
#ifndef SGLM_T3_DMA_LATE
#define SGLM_T3_DMA_LATE 1  // the step's A DMAs are written behind the first column: hipcc emits them after the step's fragment reads and the
                            // first weight wait, still in front of that column's MFMAs (1-2 % over issuing them right at the barrier)
#endif
    auto issue_dma = [&]() __attribute__((always_inline)) {
#if SGLM_T3_ABL != 4
      const int kn = kt + NSTAGE - 1;
#if SGLM_T3_ABL == 5
      dma_stage(kn % NSTAGE, 0);
#else
      dma_stage(kn % NSTAGE, kn < nk ? kn : nk - 1);
#endif
#endif
    };
#if !SGLM_T3_DMA_LATE
    issue_dma();
#endif
#if SGLM_T3_ABL == 2
#pragma unroll
    for (int i = 0; i < RI; ++i) asm volatile("" : "+v"(af[i].x[0]), "+v"(af[i].x[1]));
#else
    const char* sa_ = smem + (kt % NSTAGE) * STAGE + kg * OPA + a_row;
    Frag32 af[RI];
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      af[i].v[0] = *reinterpret_cast<const uint4*>(sa_ + i * 2048 + c0);
      af[i].v[1] = *reinterpret_cast<const uint4*>(sa_ + i * 2048 + c1);
    }
#endif
    const uint32_t voff = (uint32_t)lane * 16 + (uint32_t)(KS * (kt + 1 < nk ? kt + 1 : nk - 1) + kg) * 2048;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
#if SGLM_T3_ABL == 4
      wait_frag<YB>(bq[j]);
#elif SGLM_T3_ABL == 3
      wait_frag<0>(bq[j]);
#elif SGLM_T3_DMA_LATE
      if (j == 0) wait_frag<YB>(bq[j]);
      else wait_frag<YB + UA>(bq[j]);
#else
      wait_frag<YB + UA>(bq[j]);
#endif
#if SGLM_T3_ABL == 6
      // the step's flops as 32x32x64 MFMAs on the same operand registers (garbage products): does the other shape hold a
      // higher clock in this loop?  16 per step (4 x 2 tiles of 32 x 32, two k halves): 4 per column phase here
#pragma unroll
      for (int m = 0; m < 4; ++m)
        acc32[m][j >> 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
            __builtin_bit_cast(v8i32_t, af[2 * m + (j & 1)]), __builtin_bit_cast(v8i32_t, bq[j]), acc32[m][j >> 1], 0, 0, 0,
            0x7F7F7F7F, 0, 0x7F7F7F7F);
#else
#pragma unroll
      for (int i = 0; i < RI; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
            __builtin_bit_cast(v8i32_t, af[i]), __builtin_bit_cast(v8i32_t, bq[j]), acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0,
            0x7F7F7F7F);
#endif
#if SGLM_T3_DMA_LATE
      if (j == 0) issue_dma();  // (see SGLM_T3_DMA_LATE for where hipcc puts them)
#endif
#if SGLM_T3_ABL != 3
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bq[j].x[0]) : "v"(voff), "s"(b_blk[j]) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(bq[j].x[1]) : "v"(voff), "s"(b_blk[j]) : "memory");
#else
      (void)voff;
#endif
    }
  }
  drain_frags(bq);  // the never-consumed tail refills (and the tail DMAs)
#if SGLM_T3_TIMING
  t_loop_end = __builtin_amdgcn_s_memrealtime();
  GemmArgs pt = p;
  pt.bias = nullptr;
#define SGLM_T3_P pt
#else
#define SGLM_T3_P p
#endif
#if SGLM_T3_ABL == 6
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = acc32[i >> 1][j >> 1][(i & 1) * 8 + (j & 1) * 4 + r];
#endif
  __syncthreads();  // all stages dead: the epilogue reuses the memory
  if constexpr (KS == 2) {
    // group 1 hands its accumulators over: element e of lane l at [e][l] of the receiving wave's region
    float* hand = reinterpret_cast<float*>(smem) + wave_t * (RI * CB * 4 * 64);
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) hand[((i * CB + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] += hand[((i * CB + j) * 4 + r) * 64 + lane];
    }
    __syncthreads();  // every wave is done with the hand-over area: the epilogue patches below overlap it
    if (kg == 1) return;
  }
  if constexpr (RAWK) {
    // raw partial sums: element (row 16 i + 4 g + r, column 16 j + r16) of the wave's tile; 16 lanes write 64 contiguous bytes
    float* slab = p.slabs + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 16 * RI + 16 * i + 4 * g + r;
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          const int n = n0 + wn * 16 * CB + 16 * j + r16;
          if (m < p.M && n < p.N) slab[(int64_t)m * p.N + n] = acc[i][j][r];
        }
      }
  } else if constexpr (SILU) tiled_epilogue_silu<OUT_DTYPE, RI>(SGLM_T3_P, smem, acc, m0, n0, wm, wn, wave_t, lane);
  else tiled_epilogue<OUT_DTYPE, RI, CB>(SGLM_T3_P, smem, acc, m0, n0, wm, wn, wave_t, lane);
#undef SGLM_T3_P
#if SGLM_T3_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the output stores have left
  if (tid == 0 && p.bias != nullptr) {
    uint64_t* tb = reinterpret_cast<uint64_t*>(const_cast<void*>(p.bias)) + 4 * (int64_t)blockIdx.x;
    tb[0] = t_start; tb[1] = t_loop; tb[2] = t_loop_end; tb[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

template <int OUT_DTYPE, int MB, int NB, int WK>
int launch_skinny(const GemmArgs& p, hipStream_t s) {
  auto kern = fp8_gemm_skinny_kernel<OUT_DTYPE, MB, NB, WK>;
  constexpr int lds = WK * MB * 16 * 16 * NB * 4;
  static int attr_rc = lds <= 65536 ? 0 : check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
      "hipFuncSetAttribute");
  if (attr_rc) return attr_rc;
  const unsigned grid = (unsigned)((p.N + 16 * NB - 1) / (16 * NB));
  g_last_kernel = "skinny";
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WK), lds, s, p);
  return check_hip(hipGetLastError(), "fp8_gemm_skinny launch");
}

// Tuning override: SGL_MI355_SKINNY="NB,WK" (NB in {1,2,4}, WK in {2,4,8}).
inline void skinny_override(int& nb, int& wk) {
  static const char* e = getenv("SGL_MI355_SKINNY");
  if (e) sscanf(e, "%d,%d", &nb, &wk);
}

template <int OUT_DTYPE, int MB>
int dispatch_skinny(const GemmArgs& p, hipStream_t s) {
  // Wide N: 64-column wave tiles (activation L2 traffic = 1x the weight stream), K split 4 ways.
  // Narrow N: 32-column tiles and a deeper K split so that enough waves are in flight.
  int nb = (p.N >= 12288) ? 4 : 2;
  int wk = (p.N >= 12288) ? 4 : 8;
  skinny_override(nb, wk);
#define SK_GO(NB_, WK_) return launch_skinny<OUT_DTYPE, MB, NB_, WK_>(p, s)
  if (nb == 4) { if (wk == 2) SK_GO(4, 2); if (wk == 8) SK_GO(4, 8); SK_GO(4, 4); }
  if (nb == 1) { if (wk == 2) SK_GO(1, 2); if (wk == 4) SK_GO(1, 4); SK_GO(1, 8); }
  if (wk == 2) SK_GO(2, 2);
  if (wk == 4) SK_GO(2, 4);
  SK_GO(2, 8);
#undef SK_GO
}

template <int OUT_DTYPE>
int run_gemm(const GemmArgs& p, float* workspace, int64_t workspace_floats, hipStream_t s) {
  if (p.M <= 64) {
    static const bool no_astat = getenv("SGL_MI355_NO_ASTAT") != nullptr;  // tuning / A-B aid
    static const bool no_wstream = getenv("SGL_MI355_NO_WSTREAM") != nullptr;  // tuning / A-B aid
    static const int direct_min_n = [] { const char* e = getenv("SGL_MI355_WSTREAM_MIN_N"); return e ? atoi(e) : 16 * 8 * 100; }();  // tuning aid
    // >= ~100 workgroups of 8 column blocks (M = 64, K = 4096: N = 14336 21.6 us here vs 24.1 one-shot; at N = 7168 and
    // below the two tie or the one-shot kernel wins -- those shapes are latency-, not bandwidth-bound)
    if (!no_wstream && p.N >= direct_min_n) {
      bool used = false;
      int rc = p.M <= 16   ? launch_wstream<OUT_DTYPE, 1>(p, nullptr, 0, s, used)
               : p.M <= 32 ? launch_wstream<OUT_DTYPE, 2>(p, nullptr, 0, s, used)
                           : launch_wstream<OUT_DTYPE, 4>(p, nullptr, 0, s, used);
      if (rc || used) return rc;
    }
    if (!no_astat && !p.b_shuf && p.N >= 16 * 8 * 160) {  // same shapes, K tails / odd step counts: phase-filled A image
      bool used = false;
      int rc = p.M <= 16   ? launch_astat_direct<OUT_DTYPE, 1>(p, s, used)
               : p.M <= 32 ? launch_astat_direct<OUT_DTYPE, 2>(p, s, used)
                           : launch_astat_direct<OUT_DTYPE, 4>(p, s, used);
      if (rc || used) return rc;
    }
    // split-K slabs pay a finalize launch (~4 us): worth it from 40 Mi weights, and from 20 Mi at M > 32 where
    // the alternative re-reads the activations from L2 per wave (qkv 4096x6144: 12.6 + 4.2 us vs 21.2)
    static const int slab_env = [] { const char* e = getenv("SGL_MI355_SLAB_MIN_MI"); return e ? atoi(e) : 0; }();  // tuning aid
    const int steps128 = p.K >> 7;  // the one-shot kernel takes K/128 = WK * S with S <= 4, WK <= 8
    const bool oneshot_ok = (p.K & 127) == 0 && (steps128 <= 8 || (steps128 <= 32 && (steps128 % 4 == 0 || steps128 % 3 == 0 ||
                                                                                        (steps128 % 2 == 0 && steps128 <= 16))));
    const int64_t slab_min = (int64_t)(slab_env ? slab_env : (p.M > 32 ? (oneshot_ok ? 20 : 8) : 40)) << 20;
    if (!no_wstream && workspace != nullptr && (int64_t)p.N * p.K >= slab_min) {
      bool used = false;  // narrow N, long K: K split over workgroups, fp32 slabs
      int rc = p.M <= 16   ? launch_wstream<OUT_DTYPE, 1>(p, workspace, workspace_floats, s, used)
               : p.M <= 32 ? launch_wstream<OUT_DTYPE, 2>(p, workspace, workspace_floats, s, used)
                           : launch_wstream<OUT_DTYPE, 4>(p, workspace, workspace_floats, s, used);
      if (rc || used) return rc;
    }
    static const int astat_min = [] { const char* e = getenv("SGL_MI355_ASTAT_MIN_MI"); return e ? atoi(e) : 40; }();  // tuning aid
    if (!no_astat && !p.b_shuf && workspace != nullptr && (int64_t)p.N * p.K >= ((int64_t)astat_min << 20)) {
      bool used = false;
      int rc = p.M <= 16   ? launch_astat<OUT_DTYPE, 1>(p, workspace, workspace_floats, s, used)
               : p.M <= 32 ? launch_astat<OUT_DTYPE, 2>(p, workspace, workspace_floats, s, used)
                           : launch_astat<OUT_DTYPE, 4>(p, workspace, workspace_floats, s, used);
      if (rc || used) return rc;
    }
    static const bool no_oneshot = getenv("SGL_MI355_NO_ONESHOT") != nullptr;  // tuning / A-B aid
    if (!no_oneshot) {
      bool used = false;
      int rc = p.M <= 16   ? launch_oneshot<OUT_DTYPE, 1>(p, s, used)
               : p.M <= 32 ? launch_oneshot<OUT_DTYPE, 2>(p, s, used)
                           : launch_oneshot<OUT_DTYPE, 4>(p, s, used);
      if (rc || used) return rc;
    }
    if (p.b_shuf) {  // a pre-shuffled weight is only read by the weight-streaming, one-shot and tiled-v2 kernels
      bool used = false;
      int rc = p.M <= 16   ? launch_wstream<OUT_DTYPE, 1>(p, nullptr, 0, s, used)
               : p.M <= 32 ? launch_wstream<OUT_DTYPE, 2>(p, nullptr, 0, s, used)
                           : launch_wstream<OUT_DTYPE, 4>(p, nullptr, 0, s, used);
      if (rc || used) return rc;
      set_error("fp8_scaled_mm (pre-shuffled weight): no kernel for M=%d N=%d K=%d", p.M, p.N, p.K);
      return SGL_MI355_ERR_UNSUPPORTED;
    }
    if (p.M <= 16) return dispatch_skinny<OUT_DTYPE, 1>(p, s);
    if (p.M <= 32) return dispatch_skinny<OUT_DTYPE, 2>(p, s);
    return dispatch_skinny<OUT_DTYPE, 4>(p, s);
  }
  if (p.M <= 128) {
    // 65..128 rows (a decode batch of up to 128): still weight-bound, and the tiled kernels below would put these shapes on
    // N / 128 CUs (down_proj 4096 x 14336 at M = 128: 32 tiles, 64.8 us).  The weight streamer with 128-row A phases (MB = 8,
    // phases of four k-steps) keeps every CU streaming: direct form for wide N, split-K slabs + finalize otherwise.
    static const bool no_wstream128 = getenv("SGL_MI355_NO_WSTREAM") != nullptr || getenv("SGL_MI355_NO_WSTREAM_M128") != nullptr;  // A/B aid
    static const int direct_min_n128 = [] { const char* e = getenv("SGL_MI355_WSTREAM_MIN_N"); return e ? atoi(e) : 16 * 8 * 100; }();
    if (!no_wstream128 && (p.K & 511) == 0) {
      bool used = false;
      if (p.N >= direct_min_n128) {
        int rc = launch_wstream<OUT_DTYPE, 8>(p, nullptr, 0, s, used);
        if (rc || used) return rc;
      }
      // split-K form from 8 Mi weights.  Eager launch loops, M = 128, us, tiled -> streamer: down_proj (4096 x 14336) 64.8 -> 25.8,
      // gate_up direct 31.5 -> 30.1, but o_proj (4096 x 4096) 21.3 -> 30.9 and qkv 22.0 -> 23.3; in the graph-replayed model step
      // (where the finalize launch costs a boundary, not a launch gap) taking all of them is still ahead: Llama-3-8B bs = 128
      // 11.18 ms tiled -> 10.03 with down / gate_up only -> 9.86 with all four; bs = 96 8.86 -> 8.63 (same box)
      if (workspace != nullptr && (int64_t)p.N * p.K >= ((int64_t)8 << 20)) {
        int rc = launch_wstream<OUT_DTYPE, 8>(p, workspace, workspace_floats, s, used);
        if (rc || used) return rc;
      }
    }
  }
  if (p.M <= 256) {
    // 129..256 rows: the same again with a 256-row A image and phases of two k-steps (MB = 16).  Measured: see DESIGN 4.8.9.
    static const bool no_wstream256 = getenv("SGL_MI355_NO_WSTREAM") != nullptr || getenv("SGL_MI355_NO_WSTREAM_M256") != nullptr;  // A/B aid
    static const int direct_min_n256 = [] { const char* e = getenv("SGL_MI355_WSTREAM_MIN_N"); return e ? atoi(e) : 16 * 8 * 100; }();
    // At 256 rows the streamer is MFMA- / LDS-bound itself (64 MFMAs and 32 KB of fragment reads per wave and k-step), so only
    // the long-K, narrow-N layers gain, where the tiled kernels run on N / 128 CUs (eager loops, M = 256, us, tiled -> streamer:
    // down_proj 4096 x 14336 64.4 -> 37.7; gate_up 44.9 -> 54.7, qkv 22.4 -> 27.7, o 20.8 -> 25.7 stay tiled).
    // SGL_MI355_WSTREAM_M256_ALL=1 takes it for every shape (A/B aid).
    static const bool all256 = getenv("SGL_MI355_WSTREAM_M256_ALL") != nullptr;
    if (!no_wstream256 && (p.K & 511) == 0) {
      bool used = false;
      if (all256 && p.N >= direct_min_n256) {
        int rc = launch_wstream<OUT_DTYPE, 16>(p, nullptr, 0, s, used);
        if (rc || used) return rc;
      }
      if (workspace != nullptr && (int64_t)p.N * p.K >= ((int64_t)(all256 ? 8 : 40) << 20) && (all256 || p.K >= 2 * p.N)) {
        int rc = launch_wstream<OUT_DTYPE, 16>(p, workspace, workspace_floats, s, used);
        if (rc || used) return rc;
      }
    }
  }
  if (p.b_shuf && (p.K & 127) != 0) {
    set_error("fp8_scaled_mm (pre-shuffled weight): K must be a multiple of 128");
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  // v2 (LDS-DMA + block-scaled MFMA) whenever K has no tail; variant by tile count (see the kernel's comment).
  static const int v2_env = [] { const char* e = getenv("SGL_MI355_TILED_V2"); return e ? atoi(e) : -1; }();  // 0 off; 2, 3, 8, 48, 84 force a variant
  const unsigned grid_s = (unsigned)(((p.M + kTM - 1) / kTM) * ((p.N + kTN - 1) / kTN));
  // Selection (measured, M = 4096 / 512, PFLOP/s): 256x256 x 8 waves 1.62-2.26 wherever there are enough of the big
  // tiles (>= 192: three quarters of the CUs) -- e.g. 2.05 vs 1.57 at 4096x4096x4096; 128x128: with at most one tile per
  // CU three stages and EIGHT waves (32x64 per wave: two waves per SIMD hide each other's latency; M = 1024 o/down
  // 26.7 / 74.4 us vs 29.0 / 81.9 with four waves), else two stages x two workgroups per CU.
  // (A 4-wave 256x256 with 128x128 wave tiles -- MFMA-bound on paper -- reached only 1.47-2.12: with one wave per SIMD
  //  every barrier and first-fragment latency is exposed.)
  const unsigned grid_b = (unsigned)(((p.M + 255) / 256) * ((p.N + 255) / 256));
  int v2 = v2_env >= 0 ? v2_env : (grid_b >= 192 ? 84 : (grid_s <= 256 ? 22 : 2));
  if (p.b_shuf && v2 == 0) v2 = 2;  // the register-staged predecessor below does not read the pre-shuffled layout
  // v3 (weights global -> VGPR, pre-shuffled weights only) from 192 tiles of 128 x 256 (same box, us, v2 -> v3: M = 1024 qkv
  // 38.4 -> 36.5, gate_up 116.4 -> 111.6; M = 2048 o 40.6 -> 37.6, down 122.0 -> 100.4; M = 4096 qkv 123.9 -> 93.5, gate_up
  // 436 -> 396; M = 8192 gate_up 920 -> 774, down 397 -> 358; below that the 128 x 128 v2 tiles fill the chip better).
  // Column tiles per rasterisation group: 4 (1 / 2 / 4 / 8 / 16 at M = 4096 gate_up: 437 / 418 / 408 / 403 / 435 us).
  static const int v3_env = [] { const char* e = getenv("SGL_MI355_TILED_V3"); return e ? atoi(e) : -1; }();  // 0 off; 1 / 2 force 256x256x8 waves / 128x256x4 waves (A/B aid)
  static const int gn_env = [] { const char* e = getenv("SGL_MI355_T3_GN"); return e ? atoi(e) : 0; }();       // tuning aid
  // 128 x 256 tiles, or 128 x 192 where that leaves less work on the busiest CU: tiles per CU (rounded up) x tile width.  Only
  // the qkv-like widths gain (4096 -> 6144, same box, 256- vs 192-wide: M = 1024 36.5 -> 33.9 us, 192 -> 256 tiles; M = 2048
  // 53.0 -> 49.6, 384 -> 512 tiles); wherever the products tie the wider tile wins (M = 4096 93.2 vs 102.4) and forcing the
  // narrow one elsewhere costs 10-20 % (profiles/r03_prefill_gemm_cb3.txt).  SGL_MI355_T3_CB=3|4 forces one (A/B aid).
  static const int cb_env = [] { const char* e = getenv("SGL_MI355_T3_CB"); return e ? atoi(e) : 0; }();
  const unsigned tiles_m3 = (unsigned)((p.M + 127) / 128);
  const unsigned grid_4 = tiles_m3 * (unsigned)((p.N + 255) / 256), grid_c3 = tiles_m3 * (unsigned)((p.N + 191) / 192);
  const unsigned cost4 = ((grid_4 + 255) / 256) * 4, cost3 = ((grid_c3 + 255) / 256) * 3;
  const int cb = (cb_env == 3 || cb_env == 4) ? cb_env : (cost3 < cost4 ? 3 : 4);
  const unsigned grid_3 = cb == 3 ? grid_c3 : grid_4;
  if (p.b_shuf && (p.K & 511) == 0 && (v3_env > 0 || (v3_env != 0 && grid_3 >= 192))) {
#define TILED3_GO(WM_, CB_)                                                                                        \
  {                                                                                                               \
    GemmArgs p3 = p;                                                                                              \
    p3.raster_gn = gn_env > 0 ? gn_env : 4;                                                                       \
    auto k3 = fp8_gemm_tiled3_kernel<OUT_DTYPE, 3, 8, WM_, 4, CB_>;                                               \
    constexpr int tmb = 128 * WM_, tnb = 64 * CB_;                                                                \
    constexpr int lds_st = 3 * tmb * 128, lds_ep = WM_ * 4 * 64 * 72 * 2;                                         \
    constexpr int lds3 = lds_st > lds_ep ? lds_st : lds_ep;                                                       \
    static int a3 = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(k3),                              \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds3), "hipFuncSetAttribute"); \
    if (a3) return a3;                                                                                            \
    const unsigned g3 = (unsigned)(((p.M + tmb - 1) / tmb) * ((p.N + tnb - 1) / tnb));                            \
    g_last_kernel = "tiled3";                                                                                     \
    hipLaunchKernelGGL(k3, dim3(g3), dim3(256 * WM_), lds3, s, p3);                                               \
    return check_hip(hipGetLastError(), "fp8_gemm_tiled3 launch");                                                \
  }
    // one tile per CU or fewer (M = 1024 qkv: 256 tiles of 128 x 192): the eight-wave form with two k groups (see the kernel)
    // SGL_MI355_T3_KS=1|2 forces one (A/B aid)
    static const int ks_env = [] { const char* e = getenv("SGL_MI355_T3_KS"); return e ? atoi(e) : 0; }();
    if (v3_env <= 0 && (ks_env == 2 || (ks_env != 1 && grid_3 <= 256))) {
#define TILED3_KS2(CB_)                                                                                           \
  {                                                                                                               \
    GemmArgs p3 = p;                                                                                              \
    p3.raster_gn = gn_env > 0 ? gn_env : 4;                                                                       \
    auto k3 = fp8_gemm_tiled3_kernel<OUT_DTYPE, 3, 8, 1, 4, CB_, false, 2>;                                       \
    constexpr int lds_st = 3 * 2 * 128 * 128, lds_hand = 4 * 8 * CB_ * 4 * 64 * 4;                                \
    constexpr int lds3 = lds_st > lds_hand ? lds_st : lds_hand;                                                   \
    static int a3 = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(k3),                              \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds3), "hipFuncSetAttribute"); \
    if (a3) return a3;                                                                                            \
    const unsigned g3 = (unsigned)(((p.M + 127) / 128) * ((p.N + 64 * CB_ - 1) / (64 * CB_)));                    \
    g_last_kernel = "tiled3_ks2";                                                                                 \
    hipLaunchKernelGGL(k3, dim3(g3), dim3(512), lds3, s, p3);                                                     \
    return check_hip(hipGetLastError(), "fp8_gemm_tiled3 (two k groups) launch");                                 \
  }
      if (cb == 3) TILED3_KS2(3)
      TILED3_KS2(4)
#undef TILED3_KS2
    }
    if (v3_env == 1) TILED3_GO(2, 4)
    // Round 5: few row tiles x a very wide output (M = 1024 gate_up of Llama-3-8B: 4 x 112 tiles of 256 x 256 = 1.75 rounds of one
    // eight-wave workgroup per CU) take the 256 x 256 form: 112.0 vs 124.4-126.8 us for the 128 x 256 form's 896 tiles (the same
    // 1.75 rounds at two per CU, half the operand reuse per byte entering the CU), and ahead of the vendor library's 116.4
    // (profiles/r05_prefill_gemm_forms.txt).  Everywhere else measured the 256 x 256 form loses (M = 1024 qkv / o / down: 53 / 50 /
    // 147 us against 33 / 25 / 72; M = 4096 qkv / gate_up / down 108 / 418 / 186 against 92 / 393 / 180), so the rule is narrow.
    if (v3_env < 0 && (p.M + 255) / 256 <= 4 && grid_b >= 384 && grid_b <= 512) TILED3_GO(2, 4)
    if (cb == 3) TILED3_GO(1, 3)
    TILED3_GO(1, 4)
#undef TILED3_GO
  }
  if (v2 && (p.K & 127) == 0) {
#define TILED2_GO(NST, RI_, WM_, WN_)                                                                             \
  {                                                                                                               \
    auto k2 = fp8_gemm_tiled2_kernel<OUT_DTYPE, NST, RI_, WM_, WN_>;                                              \
    constexpr int tmb = 16 * RI_ * WM_, tnb = 64 * WN_;                                                           \
    constexpr int lds_st = NST * (tmb + tnb) * 128, lds_ep = WM_ * WN_ * 64 * 72 * 2;                             \
    constexpr int lds2 = lds_st > lds_ep ? lds_st : lds_ep;                                                       \
    static int a2 = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(k2),                              \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds2), "hipFuncSetAttribute"); \
    if (a2) return a2;                                                                                            \
    const unsigned g2 = (unsigned)(((p.M + tmb - 1) / tmb) * ((p.N + tnb - 1) / tnb));                            \
    g_last_kernel = "tiled2";                                                                                         \
    hipLaunchKernelGGL(k2, dim3(g2), dim3(64 * WM_ * WN_), lds2, s, p);                                           \
    return check_hip(hipGetLastError(), "fp8_gemm_tiled2 launch");                                                \
  }
    if (v2 == 22) TILED2_GO(3, 2, 4, 2)  // 128x128 with 8 waves (32x64 per wave)
    if (v2 == 84) TILED2_GO(2, 8, 2, 4)
    if (v2 == 48) TILED2_GO(3, 4, 4, 2)
    if (v2 == 8) TILED2_GO(3, 8, 2, 2)
    if (v2 == 2) TILED2_GO(2, 4, 2, 2)
    TILED2_GO(3, 4, 2, 2)
#undef TILED2_GO
  }
  static const bool scaled = [] { const char* e = getenv("SGL_MI355_TILED_SCALED"); return e ? atoi(e) != 0 : true; }();  // A-B aid
  auto kern = scaled ? fp8_gemm_tiled_kernel<OUT_DTYPE, true> : fp8_gemm_tiled_kernel<OUT_DTYPE, false>;
  constexpr int lds = 2 * kStageBytes;
  static int attr_rc = [] {
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(fp8_gemm_tiled_kernel<OUT_DTYPE, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute");
    return rc ? rc : check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(fp8_gemm_tiled_kernel<OUT_DTYPE, false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute");
  }();
  if (attr_rc) return attr_rc;
  const unsigned grid = (unsigned)(((p.M + kTM - 1) / kTM) * ((p.N + kTN - 1) / kTN));
  g_last_kernel = "tiled";
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p);
  return check_hip(hipGetLastError(), "fp8_gemm_tiled launch");
}

// Row-major [N][K] FP8 weight <-> the fragment-major layout the decode kernels stream (GemmArgs::b_shuf).  One thread
// per 16-byte piece: piece (nb, ks, half, lane = 16 g + r16) <-> bytes 128 ks + 64 half + 16 g .. + 16 of row 16 nb + r16.
__global__ __launch_bounds__(256) void fp8_shuffle_weight_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                 int64_t row_stride, int N, int K, int inverse) {
  const int64_t piece = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (piece >= (int64_t)N * K / 16) return;
  const int lane = (int)(piece & 63);
  const int half = (int)((piece >> 6) & 1);
  const int64_t blk = piece >> 7;  // nb * (K / 128) + ks
  const int ksteps = K >> 7;
  const int ks = (int)(blk % ksteps);
  const int64_t nb = blk / ksteps;
  const int64_t rm = (nb * 16 + (lane & 15)) * row_stride + ks * 128 + half * 64 + (lane >> 4) * 16;
  if (inverse) *reinterpret_cast<uint4*>(dst + rm) = *reinterpret_cast<const uint4*>(src + piece * 16);
  else *reinterpret_cast<uint4*>(dst + piece * 16) = *reinterpret_cast<const uint4*>(src + rm);
}

}  // namespace
}  // namespace sglm

using namespace sglm;

extern "C" int sgl_mi355_per_token_quant_fp8(
    const void* input, void* output_q, float* output_s, int64_t num_tokens, int64_t hidden_dim, int dtype,
    void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "per_token_quant_fp8: bad dtype %d", dtype);
  // per_token_quant_fp8.cu:173: "Hidden dimension must be divisible by 8"
  SGLM_CHECK_ARG(hidden_dim > 0 && hidden_dim % 8 == 0, "per_token_quant_fp8: hidden_dim (%ld) must be divisible by 8",
                 (long)hidden_dim);
  SGLM_CHECK_ARG(num_tokens >= 0 && num_tokens < (1ll << 31), "per_token_quant_fp8: bad num_tokens %ld", (long)num_tokens);
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(input && output_q && output_s, "per_token_quant_fp8: null tensor pointer");
  SGLM_CHECK_ARG(reinterpret_cast<uintptr_t>(input) % 16 == 0 && reinterpret_cast<uintptr_t>(output_q) % 8 == 0,
                 "per_token_quant_fp8: input must be 16-byte and output 8-byte aligned");
  hipStream_t s = as_stream(stream);
  // few long rows (decode): 512 threads per row -- the row is a chain of memory round trips, not bandwidth
  const bool wide = num_tokens <= 2048 && hidden_dim >= 4096;
#define PTQ_GO(DT, TT, NT_)                                                                                         \
  hipLaunchKernelGGL((per_token_quant_fp8_kernel<DT, NT_>), dim3((unsigned)num_tokens), dim3(NT_), 0, s, (const TT*)input, \
                     (uint8_t*)output_q, output_s, (int)hidden_dim)
  if (dtype == SGL_MI355_BF16) {
    if (wide) PTQ_GO(SGL_MI355_BF16, __bf16, 512); else PTQ_GO(SGL_MI355_BF16, __bf16, 256);
  } else {
    if (wide) PTQ_GO(SGL_MI355_FP16, _Float16, 512); else PTQ_GO(SGL_MI355_FP16, _Float16, 256);
  }
#undef PTQ_GO
  return check_hip(hipGetLastError(), "per_token_quant_fp8 launch");
}

// K % 512 == 0 (whole phases of four 128-byte k-steps) and N % 16 == 0: what every kernel that reads the pre-shuffled
// layout can take at any M
static bool shuffle_shape_ok(int64_t N, int64_t K) { return N > 0 && K > 0 && N % 16 == 0 && K % 512 == 0; }

static int fp8_scaled_mm_impl(
    int b_shuf, const void* mat_a, const void* mat_b, const float* scales_a, const float* scales_b, const void* bias, void* out,
    float* workspace, int64_t workspace_floats, int64_t M, int64_t N, int64_t K, int64_t a_stride_m,
    int64_t b_stride_n, int out_dtype, void* stream) {
  // preconditions of fp8_gemm_kernel.cu:1078-1108
  SGLM_CHECK_ARG(out_dtype == SGL_MI355_BF16 || out_dtype == SGL_MI355_FP16, "fp8_scaled_mm: out_dtype must be Half or BFloat16");
  SGLM_CHECK_ARG(M >= 0 && N > 0 && K > 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "fp8_scaled_mm: bad shape");
  SGLM_CHECK_ARG(K % 16 == 0, "fp8_scaled_mm: mat_a must be multiple of 16 bytes for memory alignment (K=%ld)", (long)K);
  SGLM_CHECK_ARG((N * 2) % 16 == 0, "fp8_scaled_mm: out must be multiple of 16 bytes for memory alignment (N=%ld)", (long)N);
  SGLM_CHECK_ARG(a_stride_m % 16 == 0 && b_stride_n % 16 == 0 && a_stride_m >= K && b_stride_n >= K,
                 "fp8_scaled_mm: row strides must be >= K and multiples of 16 bytes");
  if (M == 0) return 0;
  SGLM_CHECK_ARG(mat_a && mat_b && scales_a && scales_b && out, "fp8_scaled_mm: null tensor pointer");
  SGLM_CHECK_ARG(reinterpret_cast<uintptr_t>(mat_a) % 16 == 0 && reinterpret_cast<uintptr_t>(mat_b) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(out) % 16 == 0,
                 "fp8_scaled_mm: operands must be 16-byte aligned");
  SGLM_CHECK_ARG(!b_shuf || shuffle_shape_ok(N, K), "fp8_scaled_mm_wshuffled: N %% 16 == 0 and K %% 512 == 0 required (N=%ld K=%ld)",
                 (long)N, (long)K);
  GemmArgs p{(const uint8_t*)mat_a, a_stride_m, (const uint8_t*)mat_b, b_stride_n, scales_a, scales_b, bias, out,
             (int)M, (int)N, (int)K};
  p.b_shuf = b_shuf;
#if SGLM_ABL_SHUF  // variant builds only (-DSGLM_ABL_SHUF=1), never in the default library
  // timing ablation (WRONG RESULTS): read a row-major weight as if it were pre-shuffled -- same bytes, contiguous loads
  static const bool abl_shuf = getenv("SGL_MI355_ABL_SHUF") != nullptr;
  if (abl_shuf && shuffle_shape_ok(N, K)) p.b_shuf = 1;
#endif
  hipStream_t s = as_stream(stream);
  return out_dtype == SGL_MI355_BF16 ? run_gemm<SGL_MI355_BF16>(p, workspace, workspace_floats, s)
                                     : run_gemm<SGL_MI355_FP16>(p, workspace, workspace_floats, s);
}

extern "C" int sgl_mi355_fp8_scaled_mm(
    const void* mat_a, const void* mat_b, const float* scales_a, const float* scales_b, const void* bias, void* out,
    float* workspace, int64_t workspace_floats, int64_t M, int64_t N, int64_t K, int64_t a_stride_m,
    int64_t b_stride_n, int out_dtype, void* stream) {
  return fp8_scaled_mm_impl(0, mat_a, mat_b, scales_a, scales_b, bias, out, workspace, workspace_floats, M, N, K, a_stride_m,
                            b_stride_n, out_dtype, stream);
}

// mat_b in the pre-shuffled layout of sgl_mi355_fp8_shuffle_weight (N % 16 == 0, K % 512 == 0); otherwise as above
extern "C" int sgl_mi355_fp8_scaled_mm_wshuffled(
    const void* mat_a, const void* mat_b, const float* scales_a, const float* scales_b, const void* bias, void* out,
    float* workspace, int64_t workspace_floats, int64_t M, int64_t N, int64_t K, int64_t a_stride_m, int out_dtype,
    void* stream) {
  return fp8_scaled_mm_impl(1, mat_a, mat_b, scales_a, scales_b, bias, out, workspace, workspace_floats, M, N, K, a_stride_m,
                            K, out_dtype, stream);
}

// gate_up GEMM of a gated MLP with the activation folded into its epilogue: out[m][i] = silu(y[m][i]) * y[m][I + i], y =
// fp8_scaled_mm(mat_a, mat_b, ...) [M][N], I = N / 2 -- rounded exactly like sgl_mi355_fp8_scaled_mm_wshuffled followed by
// sgl_mi355_silu_and_mul (fp8_utils.py:696-704, activation.py:59-83), without the [M][N] round trip through HBM.  Prefill
// sizes on a pre-shuffled weight only: at least 192 tiles of 128 rows x 128 output columns (M = 1024 x I = 14336: 896);
// anything else returns SGL_MI355_ERR_UNSUPPORTED WITHOUT launching and the caller makes the two calls.
template <int OUT_DTYPE>
static int run_gemm_silu(const GemmArgs& p, hipStream_t s) {
  GemmArgs p3 = p;
  static const int gn_env = [] { const char* e = getenv("SGL_MI355_T3_GN"); return e ? atoi(e) : 0; }();  // tuning aid
  p3.raster_gn = gn_env > 0 ? gn_env : 4;
  auto k3 = fp8_gemm_tiled3_kernel<OUT_DTYPE, 3, 8, 1, 4, 4, true>;
  constexpr int lds_st = 3 * 128 * 128, lds_ep = 4 * 64 * 72 * 2;
  constexpr int lds3 = lds_st > lds_ep ? lds_st : lds_ep;
  static int a3 = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, lds3),
                            "hipFuncSetAttribute");
  if (a3) return a3;
  const unsigned g3 = (unsigned)(((p.M + 127) / 128) * (((p.N >> 1) + 127) / 128));
  g_last_kernel = "tiled3_silu";
  hipLaunchKernelGGL(k3, dim3(g3), dim3(256), lds3, s, p3);
  return check_hip(hipGetLastError(), "fp8_gemm_tiled3 (silu * mul) launch");
}

extern "C" int sgl_mi355_fp8_scaled_mm_silu_mul_wshuffled(
    const void* mat_a, const void* mat_b, const float* scales_a, const float* scales_b, const void* bias, void* out, int64_t M,
    int64_t N, int64_t K, int64_t a_stride_m, int out_dtype, void* stream) {
  SGLM_CHECK_ARG(out_dtype == SGL_MI355_BF16 || out_dtype == SGL_MI355_FP16, "fp8_scaled_mm_silu_mul: out_dtype must be Half or BFloat16");
  SGLM_CHECK_ARG(M >= 0 && N > 0 && K > 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "fp8_scaled_mm_silu_mul: bad shape");
  SGLM_CHECK_ARG(N % 32 == 0 && shuffle_shape_ok(N, K),
                 "fp8_scaled_mm_silu_mul: N %% 32 == 0 (gate and up halves of whole 16-column blocks) and K %% 512 == 0 required (N=%ld K=%ld)",
                 (long)N, (long)K);
  SGLM_CHECK_ARG(a_stride_m % 16 == 0 && a_stride_m >= K, "fp8_scaled_mm_silu_mul: row stride must be >= K and a multiple of 16 bytes");
  if (M == 0) return 0;
  SGLM_CHECK_ARG(mat_a && mat_b && scales_a && scales_b && out, "fp8_scaled_mm_silu_mul: null tensor pointer");
  SGLM_CHECK_ARG(reinterpret_cast<uintptr_t>(mat_a) % 16 == 0 && reinterpret_cast<uintptr_t>(mat_b) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(out) % 16 == 0,
                 "fp8_scaled_mm_silu_mul: operands must be 16-byte aligned");
  const int64_t tiles = ((M + 127) / 128) * ((N / 2 + 127) / 128);
  static const int min_tiles = [] { const char* e = getenv("SGL_MI355_SILU_GEMM_MIN_TILES"); return e ? atoi(e) : 192; }();  // tuning aid
  if (M <= 64 || tiles < min_tiles) {
    set_error("fp8_scaled_mm_silu_mul: only prefill sizes (M > 64, at least %d tiles of 128 x 128 outputs) have the fused form", min_tiles);
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  GemmArgs p{(const uint8_t*)mat_a, a_stride_m, (const uint8_t*)mat_b, K, scales_a, scales_b, bias, out, (int)M, (int)N, (int)K};
  p.b_shuf = 1;
  hipStream_t s = as_stream(stream);
  return out_dtype == SGL_MI355_BF16 ? run_gemm_silu<SGL_MI355_BF16>(p, s) : run_gemm_silu<SGL_MI355_FP16>(p, s);
}

// Split-K partial sums only: workspace[slice][m][n] = sum over the slice's k of a[m][k] * b[n][k] (raw fp32, no
// scales).  The epilogue (x w_scale, x x_scale, + bias, round) is applied by the consumer: sgl_mi355_fp8_scaled_mm_finalize
// or one of the fused *_from_partials kernels, which sum the slices in the same order -- bit-identical to
// sgl_mi355_fp8_scaled_mm on the split-K path.  UNSUPPORTED when the shape has no split-K path (caller falls back).
static int fp8_scaled_mm_partials_impl(int b_shuf, const void* mat_a, const void* mat_b, float* workspace,
                                       int64_t workspace_floats, int64_t M, int64_t N, int64_t K,
                                       int64_t a_stride_m, int64_t b_stride_n, int32_t* num_slices,
                                       void* stream) {
  SGLM_CHECK_ARG(num_slices != nullptr && workspace != nullptr, "fp8_scaled_mm_partials: null workspace / num_slices");
  SGLM_CHECK_ARG(M > 0 && N > 0 && K > 0 && N < (1ll << 31) && K < (1ll << 31), "fp8_scaled_mm_partials: bad shape");
  SGLM_CHECK_ARG(a_stride_m % 16 == 0 && b_stride_n % 16 == 0 && a_stride_m >= K && b_stride_n >= K,
                 "fp8_scaled_mm_partials: row strides must be >= K and multiples of 16 bytes");
  SGLM_CHECK_ARG(mat_a && mat_b && reinterpret_cast<uintptr_t>(mat_a) % 16 == 0 && reinterpret_cast<uintptr_t>(mat_b) % 16 == 0,
                 "fp8_scaled_mm_partials: operands must be 16-byte aligned");
  *num_slices = 0;
  if (M > 128) {
    // prefill sizes (round 5): the tiled kernel's raw split-K form where 128 x 256 tiles alone would leave the chip half empty
    // and K is long enough to cut -- down_proj 14336 -> 4096 at 512 / 1024 rows (tools/exp/splitk_prefill_probe.py: the GEMM body
    // 85 -> 57 us at 1024 rows).  Everything else has no such form: the caller runs the GEMM that finishes itself.
    const int64_t tiles = ((M + 127) / 128) * ((N + 255) / 256);
    const int64_t steps = K >> 7;
    int S = tiles > 0 ? (int)(256 / tiles) : 0;
    if (S > 4) S = 4;
    while (S >= 2 && (steps % S != 0 || (steps / S) % 4 != 0 || steps / S < 16)) --S;
    static const bool off = getenv("SGL_MI355_NO_PREFILL_SPLITK") != nullptr;  // A/B aid
    if (off || !b_shuf || !shuffle_shape_ok(N, K) || K < 8192 || S < 2 || (N & 7) != 0 ||
        workspace_floats < (int64_t)S * M * N) {
      set_error("fp8_scaled_mm_partials: no split-K form for M=%ld N=%ld K=%ld (prefill sizes: pre-shuffled weight, K >= 8192, "
                "at most 128 tiles of 128 x 256)", (long)M, (long)N, (long)K);
      return SGL_MI355_ERR_UNSUPPORTED;
    }
    GemmArgs p3{(const uint8_t*)mat_a, a_stride_m, (const uint8_t*)mat_b, K, nullptr, nullptr, nullptr, nullptr, (int)M, (int)N, (int)K};
    p3.b_shuf = 1;
    p3.raster_gn = 4;
    p3.slabs = workspace;
    p3.k_steps_per_slice = (int)(steps / S);
    auto k3 = fp8_gemm_tiled3_kernel<SGL_MI355_BF16, 3, 8, 1, 4, 4, false, 1, true>;
    constexpr int lds3 = 3 * 128 * 128;
    static int a3 = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, lds3),
                              "hipFuncSetAttribute");
    if (a3) return a3;
    g_last_kernel = "tiled3_rawk";
    hipLaunchKernelGGL(k3, dim3((unsigned)tiles, (unsigned)S), dim3(256), lds3, as_stream(stream), p3);
    int rc3 = check_hip(hipGetLastError(), "fp8_gemm_tiled3 (raw split-K) launch");
    if (rc3) return rc3;
    *num_slices = S;
    return 0;
  }
  SGLM_CHECK_ARG(!b_shuf || shuffle_shape_ok(N, K), "fp8_scaled_mm_partials_wshuffled: N %% 16 == 0 and K %% 512 == 0 required");
  GemmArgs p{(const uint8_t*)mat_a, a_stride_m, (const uint8_t*)mat_b, b_stride_n, nullptr, nullptr, nullptr, nullptr,
             (int)M, (int)N, (int)K};
  p.b_shuf = b_shuf;
#if SGLM_ABL_SHUF  // timing ablation, see fp8_scaled_mm_impl
  static const bool abl_shuf = getenv("SGL_MI355_ABL_SHUF") != nullptr;
  if (abl_shuf && shuffle_shape_ok(N, K)) p.b_shuf = 1;
#endif
  hipStream_t s = as_stream(stream);
  bool used = false;
  int sk = 0;
  int rc = M <= 16   ? launch_wstream<SGL_MI355_BF16, 1>(p, workspace, workspace_floats, s, used, &sk)
           : M <= 32 ? launch_wstream<SGL_MI355_BF16, 2>(p, workspace, workspace_floats, s, used, &sk)
           : M <= 64 ? launch_wstream<SGL_MI355_BF16, 4>(p, workspace, workspace_floats, s, used, &sk)
                     : launch_wstream<SGL_MI355_BF16, 8>(p, workspace, workspace_floats, s, used, &sk);  // 128-row phases
  if (rc) return rc;
  if (!used) {
    set_error("fp8_scaled_mm_partials: shape (M=%ld N=%ld K=%ld) is not on the split-K weight-streaming path", (long)M,
              (long)N, (long)K);
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  *num_slices = sk;
  return 0;
}

extern "C" int sgl_mi355_fp8_scaled_mm_partials(const void* mat_a, const void* mat_b, float* workspace,
                                                int64_t workspace_floats, int64_t M, int64_t N, int64_t K,
                                                int64_t a_stride_m, int64_t b_stride_n, int32_t* num_slices,
                                                void* stream) {
  return fp8_scaled_mm_partials_impl(0, mat_a, mat_b, workspace, workspace_floats, M, N, K, a_stride_m, b_stride_n, num_slices,
                                     stream);
}

extern "C" int sgl_mi355_fp8_scaled_mm_partials_wshuffled(const void* mat_a, const void* mat_b, float* workspace,
                                                          int64_t workspace_floats, int64_t M, int64_t N, int64_t K,
                                                          int64_t a_stride_m, int32_t* num_slices, void* stream) {
  return fp8_scaled_mm_partials_impl(1, mat_a, mat_b, workspace, workspace_floats, M, N, K, a_stride_m, K, num_slices, stream);
}

// sgl_mi355_fp8_scaled_mm_partials on 16-bit activations whose per-token absmax is already known (left by
// sgl_mi355_decode_attention_absmax): the kernel applies sgl_per_token_quant_fp8's arithmetic (scale = absmax / 448,
// x * (1 / scale) clamped to +-448, cast to e4m3fn) while it stages its K slice into LDS, and writes scales_a_out[m] for
// the consumer's epilogue.  Partial sums and scales are bit-identical to sgl_per_token_quant_fp8 followed by
// sgl_mi355_fp8_scaled_mm_partials.  b_shuffled != 0: mat_b in the pre-shuffled layout (b_stride_n unused).
// UNSUPPORTED (nothing launched) unless M <= 64 and the split-K choice gives single-phase slices (K / slices <= 1024,
// e.g. o_proj 4096 -> 4096).
extern "C" int sgl_mi355_fp8_scaled_mm_partials_a16(const void* mat_a16, int64_t a_stride_m, const float* row_absmax,
                                                    float* scales_a_out, const void* mat_b, int b_shuffled,
                                                    int64_t b_stride_n, float* workspace, int64_t workspace_floats,
                                                    int64_t M, int64_t N, int64_t K, int a_dtype, int32_t* num_slices,
                                                    void* stream) {
#if !SGLM_OPTIN_FUSIONS
  set_error("%s: an opt-in fusion, not in this build of the library (build with -DSGLM_OPTIN_FUSIONS=1)", "fp8_scaled_mm_partials_a16");
  return SGL_MI355_ERR_UNSUPPORTED;
#else
  SGLM_CHECK_ARG(num_slices != nullptr && workspace != nullptr, "fp8_scaled_mm_partials_a16: null workspace / num_slices");
  SGLM_CHECK_ARG(a_dtype == SGL_MI355_BF16 || a_dtype == SGL_MI355_FP16, "fp8_scaled_mm_partials_a16: activations must be bf16 / fp16");
  SGLM_CHECK_ARG(M > 0 && N > 0 && K > 0 && N < (1ll << 31) && K < (1ll << 31) && K % 16 == 0, "fp8_scaled_mm_partials_a16: bad shape");
  SGLM_CHECK_ARG(a_stride_m % 8 == 0 && a_stride_m >= K, "fp8_scaled_mm_partials_a16: activation rows must be 16-byte aligned");
  SGLM_CHECK_ARG(mat_a16 && mat_b && row_absmax && scales_a_out && reinterpret_cast<uintptr_t>(mat_a16) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(mat_b) % 16 == 0,
                 "fp8_scaled_mm_partials_a16: null or misaligned tensor pointer");
  SGLM_CHECK_ARG(!b_shuffled || shuffle_shape_ok(N, K), "fp8_scaled_mm_partials_a16: N %% 16 == 0 and K %% 512 == 0 required for a pre-shuffled weight");
  SGLM_CHECK_ARG(b_shuffled || (b_stride_n % 16 == 0 && b_stride_n >= K), "fp8_scaled_mm_partials_a16: bad weight row stride");
  *num_slices = 0;
  if (M > 64) {
    set_error("fp8_scaled_mm_partials_a16: only the decode kernels (M <= 64) have this form");
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  GemmArgs p{nullptr, K, (const uint8_t*)mat_b, b_shuffled ? K : b_stride_n, nullptr, nullptr, nullptr, nullptr, (int)M, (int)N, (int)K};
  p.b_shuf = b_shuffled ? 1 : 0;
  p.a16 = mat_a16; p.a16_sm = a_stride_m; p.a_absmax = row_absmax; p.a_scale_out = scales_a_out;
  hipStream_t s = as_stream(stream);
  bool used = false;
  int sk = 0;
  int rc;
  if (a_dtype == SGL_MI355_BF16)
    rc = M <= 16   ? launch_wstream<SGL_MI355_BF16, 1>(p, workspace, workspace_floats, s, used, &sk)
         : M <= 32 ? launch_wstream<SGL_MI355_BF16, 2>(p, workspace, workspace_floats, s, used, &sk)
                   : launch_wstream<SGL_MI355_BF16, 4>(p, workspace, workspace_floats, s, used, &sk);
  else
    rc = M <= 16   ? launch_wstream<SGL_MI355_FP16, 1>(p, workspace, workspace_floats, s, used, &sk)
         : M <= 32 ? launch_wstream<SGL_MI355_FP16, 2>(p, workspace, workspace_floats, s, used, &sk)
                   : launch_wstream<SGL_MI355_FP16, 4>(p, workspace, workspace_floats, s, used, &sk);
  if (rc) return rc;
  if (!used) {
    set_error("fp8_scaled_mm_partials_a16: shape (M=%ld N=%ld K=%ld) has no single-phase split-K form", (long)M, (long)N, (long)K);
    return SGL_MI355_ERR_UNSUPPORTED;
  }
  *num_slices = sk;
  return 0;
#endif
}

extern "C" int sgl_mi355_fp8_scaled_mm_finalize(const float* partials, int64_t num_slices, const float* scales_a,
                                                const float* scales_b, const void* bias, void* out, int64_t M, int64_t N,
                                                int out_dtype, void* stream) {
  SGLM_CHECK_ARG(out_dtype == SGL_MI355_BF16 || out_dtype == SGL_MI355_FP16, "fp8_scaled_mm_finalize: bad out_dtype");
  SGLM_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && num_slices >= 1, "fp8_scaled_mm_finalize: bad shape");
  SGLM_CHECK_ARG(partials && scales_a && scales_b && out, "fp8_scaled_mm_finalize: null tensor pointer");
  GemmArgs p{nullptr, 0, nullptr, 0, scales_a, scales_b, bias, out, (int)M, (int)N, 0};
  const int64_t total = M * N / 8;
  hipStream_t s = as_stream(stream);
  if (out_dtype == SGL_MI355_BF16)
    hipLaunchKernelGGL((fp8_gemm_finalize_kernel<SGL_MI355_BF16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p,
                       partials, (int)num_slices);
  else
    hipLaunchKernelGGL((fp8_gemm_finalize_kernel<SGL_MI355_FP16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p,
                       partials, (int)num_slices);
  return check_hip(hipGetLastError(), "fp8_gemm_finalize launch");
}

// Re-lay a row-major FP8 weight [N][K] (row stride in bytes) into the layout sgl_mi355_fp8_scaled_mm_wshuffled reads
// (inverse != 0: back to row-major; row_stride then describes dst).  N % 16 == 0, K % 512 == 0; src and dst must not
// overlap.  A one-time cost in process_weights_after_loading (w8a8_fp8.py:104-134 is the reference's hook for repacking).
extern "C" int sgl_mi355_fp8_shuffle_weight(const void* src, void* dst, int64_t N, int64_t K, int64_t row_stride, int inverse,
                                            void* stream) {
  SGLM_CHECK_ARG(shuffle_shape_ok(N, K) && N < (1ll << 31) && K < (1ll << 31),
                 "fp8_shuffle_weight: N %% 16 == 0 and K %% 512 == 0 required (N=%ld K=%ld)", (long)N, (long)K);
  SGLM_CHECK_ARG(src && dst && src != dst && row_stride >= K && row_stride % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0,
                 "fp8_shuffle_weight: 16-byte aligned, distinct buffers; row stride >= K, multiple of 16");
  const int64_t pieces = N * K / 16;
  hipLaunchKernelGGL(fp8_shuffle_weight_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, as_stream(stream),
                     (const uint8_t*)src, (uint8_t*)dst, row_stride, (int)N, (int)K, inverse);
  return check_hip(hipGetLastError(), "fp8_shuffle_weight launch");
}

extern "C" const char* sgl_mi355_fp8_last_kernel(void) { return g_last_kernel; }
