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
#include "common.h"

namespace sglm {
namespace {

constexpr float kFp8Max = 448.0f;

// ------------------------------------------------------------------------------------------
// per-token quant: one workgroup (256 threads) per row; two passes (second read is L2-hot)
template <int DTYPE>
__global__ __launch_bounds__(256) void per_token_quant_fp8_kernel(
    const typename Half16<DTYPE>::T* __restrict__ x, uint8_t* __restrict__ q, float* __restrict__ s, int K) {
  using H = Half16<DTYPE>;
  using x8 = typename H::x8;
  __shared__ float red[4];
  const int t = blockIdx.x;
  const x8* xr = reinterpret_cast<const x8*>(x + (int64_t)t * K);
  const int nv = K >> 3;
  float amax = 0.f;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const x8 v = xr[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(H::to_f32(v[j])));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float scale = amax / kFp8Max;
  if (threadIdx.x == 0) s[t] = scale;
  const float inv = scale == 0.f ? 0.f : 1.0f / scale;
  uint2* qr = reinterpret_cast<uint2*>(q + (int64_t)t * K);
  for (int i = threadIdx.x; i < nv; i += 256) {
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
};

union Frag32 {  // 32 contiguous K bytes of one row = four MFMA operands
  uint4 v[2];
  long l[4];
};

__device__ __forceinline__ void load32(Frag32& f, const uint8_t* p, int k, int kend, bool row_ok) {
  // K % 16 == 0, so each 16-B half is entirely inside or outside [0, kend)
  f.v[0] = (row_ok && k < kend) ? *reinterpret_cast<const uint4*>(p + k) : uint4{0, 0, 0, 0};
  f.v[1] = (row_ok && k + 16 < kend) ? *reinterpret_cast<const uint4*>(p + k + 16) : uint4{0, 0, 0, 0};
}

// skinny: M <= 16*MB.  Workgroup = WN*WK waves: WN column blocks of 16, K split WK ways.
template <int OUT_DTYPE, int MB, int WN, int WK>
__global__ __launch_bounds__(64 * WN * WK) void fp8_gemm_skinny_kernel(GemmArgs p) {
  using H = Half16<OUT_DTYPE>;
  using T = typename H::T;
  constexpr int PB = 4;  // weight prefetch distance in k-steps
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [WK][WN][MB*16][16]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave % WN, wk = wave / WN;
  const int r16 = lane & 15, g = lane >> 4;
  const int n0 = (blockIdx.x * WN + wn) * 16;
  const int n = n0 + r16;
  const bool n_ok = n < p.N;

  // this wave's K range, in 128-wide steps
  const int steps_total = (p.K + 127) >> 7;
  const int steps_per = (steps_total + WK - 1) / WK;
  const int s_begin = wk * steps_per;
  const int s_end = (s_begin + steps_per) < steps_total ? (s_begin + steps_per) : steps_total;
  const int nsteps = s_end - s_begin;

  const uint8_t* brow = p.b + (int64_t)(n_ok ? n : 0) * p.b_sn + 32 * g;
  const uint8_t* arow[MB];
  bool a_ok[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = 16 * mb + r16;
    a_ok[mb] = m < p.M;
    arow[mb] = p.a + (int64_t)(a_ok[mb] ? m : 0) * p.a_sm + 32 * g;
  }

  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The k sum is order-free, so every column block starts its sweep at a different k-step:
  // weight rows are K bytes apart, and a chip full of waves all reading the same residue
  // mod 4 KiB would camp on a few HBM channels.
  const int rot = nsteps > 0 ? (int)((blockIdx.x * 5u + wn * 3u) % (unsigned)nsteps) : 0;
  auto kof = [&](int s) {
    int t = s + rot;
    t = t >= nsteps ? t - nsteps : t;
    return (s_begin + t) << 7;
  };

  Frag32 bq[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i)
    if (i < nsteps) load32(bq[i], brow, kof(i), p.K - 32 * g, n_ok);

  for (int s0 = 0; s0 < nsteps; s0 += PB) {
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int s = s0 + i;
      if (s < nsteps) {
        const int k = kof(s);
        Frag32 af[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) load32(af[mb], arow[mb], k, p.K - 32 * g, a_ok[mb]);
        const Frag32 bf = bq[i];
        if (s + PB < nsteps) load32(bq[i], brow, kof(s + PB), p.K - 32 * g, n_ok);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af[mb].l[ks], bf.l[ks], acc[mb], 0, 0, 0);
        }
      }
    }
  }

  // ---- cross-wave K reduction + transposed epilogue through LDS
  // acc[mb][r] = C[m = 16mb + 4g + r][n = n0 + r16]
  {
    float* dst = red + ((wk * WN + wn) * MB * 16) * 16;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(16 * mb + 4 * g + r) * 16 + r16] = acc[mb][r];
  }
  __syncthreads();
  // each thread finishes 8 consecutive columns of one row
  constexpr int ROWS = MB * 16;
  constexpr int CHUNKS = ROWS * WN * 2;
  for (int c = threadIdx.x; c < CHUNKS; c += 64 * WN * WK) {
    const int half = c & 1;
    const int w = (c >> 1) % WN;
    const int m = (c >> 1) / WN;
    const int nn = (blockIdx.x * WN + w) * 16 + half * 8;
    if (m >= p.M || nn >= p.N) continue;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
    for (int kk = 0; kk < WK; ++kk) {
      const float* src = red + ((kk * WN + w) * ROWS + m) * 16 + half * 8;
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
// tiled: 128x128x128, 4 waves as 2x2, each wave 64x64 (4x4 fragments of 16x16).
constexpr int kTM = 128, kTN = 128, kTK = 128;
constexpr int kRegion = kTM * 32 + 16;      // one 32-B k-chunk column of the tile, +16 B skew
constexpr int kOperand = 4 * kRegion + 48;  // 16512 B, keeps 16-B alignment
constexpr int kStageBytes = 2 * kOperand;   // A + B

template <int OUT_DTYPE>
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
      const bool ok = k0 + 16 * ld_j[i] < p.K;
      ra[i] = ok ? *reinterpret_cast<const uint4*>(a_ptr[i] + k0) : uint4{0, 0, 0, 0};
      rb[i] = ok ? *reinterpret_cast<const uint4*>(b_ptr[i] + k0) : uint4{0, 0, 0, 0};
    }
  };
  auto lstore = [&](int stage) {
    char* sa_ = smem + stage * kStageBytes;
    char* sb_ = sa_ + kOperand;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (ld_j[i] >> 1) * kRegion + ld_row[i] * 32 + (ld_j[i] & 1) * 16;
      *reinterpret_cast<uint4*>(sa_ + off) = ra[i];
      *reinterpret_cast<uint4*>(sb_ + off) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + kTK - 1) / kTK;
  gload(0);
  lstore(0);
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
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af[i].l[ks], bf[j].l[ks], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) lstore(st ^ 1);
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

template <int OUT_DTYPE, int MB, int WN, int WK>
int launch_skinny(const GemmArgs& p, hipStream_t s) {
  const int lds = WK * WN * MB * 16 * 16 * 4;
  const unsigned grid = (unsigned)((p.N + 16 * WN - 1) / (16 * WN));
  hipLaunchKernelGGL((fp8_gemm_skinny_kernel<OUT_DTYPE, MB, WN, WK>), dim3(grid), dim3(64 * WN * WK), lds, s, p);
  return check_hip(hipGetLastError(), "fp8_gemm_skinny launch");
}

template <int OUT_DTYPE, int MB>
int dispatch_skinny(const GemmArgs& p, hipStream_t s) {
  // Aim for >= ~2 workgroups per CU; wide N needs no K split, narrow N splits K across waves.
  const int nblk = (p.N + 15) / 16;
  if (nblk >= 1024) return launch_skinny<OUT_DTYPE, MB, 2, 2>(p, s);
  if (nblk >= 384) return launch_skinny<OUT_DTYPE, MB, 1, 4>(p, s);
  return launch_skinny<OUT_DTYPE, MB, 1, 8>(p, s);
}

template <int OUT_DTYPE>
int run_gemm(const GemmArgs& p, hipStream_t s) {
  if (p.M <= 16) return dispatch_skinny<OUT_DTYPE, 1>(p, s);
  if (p.M <= 32) return dispatch_skinny<OUT_DTYPE, 2>(p, s);
  if (p.M <= 64) return dispatch_skinny<OUT_DTYPE, 4>(p, s);
  auto kern = fp8_gemm_tiled_kernel<OUT_DTYPE>;
  constexpr int lds = 2 * kStageBytes;
  static int attr_rc = check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
      "hipFuncSetAttribute");
  if (attr_rc) return attr_rc;
  const unsigned grid = (unsigned)(((p.M + kTM - 1) / kTM) * ((p.N + kTN - 1) / kTN));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p);
  return check_hip(hipGetLastError(), "fp8_gemm_tiled launch");
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
  if (dtype == SGL_MI355_BF16)
    hipLaunchKernelGGL((per_token_quant_fp8_kernel<SGL_MI355_BF16>), dim3((unsigned)num_tokens), dim3(256), 0, s,
                       (const __bf16*)input, (uint8_t*)output_q, output_s, (int)hidden_dim);
  else
    hipLaunchKernelGGL((per_token_quant_fp8_kernel<SGL_MI355_FP16>), dim3((unsigned)num_tokens), dim3(256), 0, s,
                       (const _Float16*)input, (uint8_t*)output_q, output_s, (int)hidden_dim);
  return check_hip(hipGetLastError(), "per_token_quant_fp8 launch");
}

extern "C" int sgl_mi355_fp8_scaled_mm(
    const void* mat_a, const void* mat_b, const float* scales_a, const float* scales_b, const void* bias, void* out,
    int64_t M, int64_t N, int64_t K, int64_t a_stride_m, int64_t b_stride_n, int out_dtype, void* stream) {
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
  GemmArgs p{(const uint8_t*)mat_a, a_stride_m, (const uint8_t*)mat_b, b_stride_n, scales_a, scales_b, bias, out,
             (int)M, (int)N, (int)K};
  hipStream_t s = as_stream(stream);
  return out_dtype == SGL_MI355_BF16 ? run_gemm<SGL_MI355_BF16>(p, s) : run_gemm<SGL_MI355_FP16>(p, s);
}
