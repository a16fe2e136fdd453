// AWQ INT4 (group-wise, zero-point) weight-only path for MI355X / gfx950.
//
// Replaces:
//   * awq_dequantize(qweight, scales, qzeros) -> [K, N]
//       sgl-kernel/csrc/gemm/awq_kernel.cu:126-221; python/sglang/srt/layers/quantization/awq_triton.py:13-107
//   * AWQLinearMethod.apply = awq_dequantize + torch.matmul (+ bias)
//       python/sglang/srt/layers/quantization/awq.py:401-418
//
// Format (awq.py:355-394): qweight int32 [K, N/8], qzeros int32 [K/G, N/8], scales fp16/bf16 [K/G, N];
//   out[k][8c+j] = (nib(qweight[k][c], ORDER[j]) - nib(qzeros[k/G][c], ORDER[j])) * scales[k/G][8c+j],
//   ORDER = [0,4,1,5,2,6,3,7], nib(x,i) = (x >> 4i) & 15.  (w - z) is exact in the 16-bit dtype and
//   the product is rounded once, so the dequantised weight is bit-exact with the reference.
//   The nibble order is the one that makes `(word >> 4i) & 0x000F000F` a PAIR of adjacent columns
//   (2i, 2i+1) -- the fp16 path unpacks two columns per VALU op with the 0x6400 (=1024.0h) bias trick.
//
// Fused GEMM (decode, M <= 64): the dequant never touches HBM.  The weight is [K][N] with N
// contiguous, i.e. the contraction index is the SLOW dimension -- the same situation as V in
// attention -- so each wave dequantises a [32 k][128 n] tile into its private LDS region and reads
// MFMA B-fragments back with ds_read_b64_tr_b16 (hardware transpose).  K is split over the 4 waves
// of a workgroup (no barrier in the loop) and optionally over workgroups (fp32 slabs + a finalize
// kernel) so that narrow layers still fill the chip.  Accumulation is fp32, the weight is rounded
// to the 16-bit dtype first exactly as the reference's two-step path does.
#include "common.h"

namespace sglm {
namespace {

__device__ __forceinline__ int awq_nib(uint32_t w, int j) {
  // column j of the 8 packed in w
  const int order = ((j & 1) << 2) | (j >> 1);  // [0,4,1,5,2,6,3,7][j]
  return (int)((w >> (4 * order)) & 0xFu);
}

template <int DTYPE>
__global__ __launch_bounds__(256) void awq_dequant_kernel(
    const uint32_t* __restrict__ qweight, const typename Half16<DTYPE>::T* __restrict__ scales,
    const uint32_t* __restrict__ qzeros, typename Half16<DTYPE>::T* __restrict__ out, int K, int Nc, int G) {
  using H = Half16<DTYPE>;
  using x8 = typename H::x8;
  const int64_t total = (int64_t)K * Nc;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i / Nc), c = (int)(i - (int64_t)k * Nc);
    const int g = k / G;
    const uint32_t w = qweight[i];
    const uint32_t z = qzeros[(int64_t)g * Nc + c];
    const x8 s = *reinterpret_cast<const x8*>(scales + ((int64_t)g * Nc + c) * 8);
    x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = H::from_f32((float)(awq_nib(w, j) - awq_nib(z, j)) * H::to_f32(s[j]));
    *reinterpret_cast<x8*>(out + i * 8) = o;
  }
}

struct AwqArgs {
  const void* x;            // [M, K] 16-bit, row-major
  const uint32_t* qweight;  // [K, Nc]
  const void* scales;       // [K/G, N]
  const uint32_t* qzeros;   // [K/G, Nc]
  const void* bias;         // [N] or null
  void* out;                // [M, N] 16-bit
  float* slabs;             // [SK][M][N] fp32 when SK > 1
  int M, N, K, G, SK;
};

template <int D>
__device__ __forceinline__ int swz_w(int c, int row) {  // 16-B chunk swizzle of the [32][128] fp16 tile
  const int f = (row & 3) | (((row >> 3) & 1) << 2);
  return (((c >> 1) ^ f) << 1) | (c & 1);
}

// One workgroup = 4 waves = a [M<=16*MB] x [128 n] output tile over a K slice; wave w sweeps a quarter of it.
template <int DTYPE, int MB>
__global__ __launch_bounds__(256) void awq_gemm_kernel(AwqArgs p) {
  using H = Half16<DTYPE>;
  using T = typename H::T;
  using x8 = typename H::x8;
  using x4 = typename H::x4;
  constexpr int TILE_BYTES = 32 * 256;  // [32 k][128 n] 16-bit
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 128;
  const int Nc = p.N >> 3;
  const int sk = blockIdx.y;

  // K slice of this workgroup, then of this wave, in 32-row steps
  const int steps_total = p.K >> 5;
  const int steps_wg = (steps_total + p.SK - 1) / p.SK;
  const int wg_begin = sk * steps_wg;
  const int wg_end = (wg_begin + steps_wg) < steps_total ? (wg_begin + steps_wg) : steps_total;
  const int steps_w = (wg_end - wg_begin + 3) / 4;
  const int s_begin = wg_begin + wave * steps_w;
  const int s_end = (s_begin + steps_w) < wg_end ? (s_begin + steps_w) : wg_end;

  // weight-load map: lane -> (row kr of 16, four packed words = 32 columns)
  const int kr = lane >> 2, c4 = lane & 3;
  const int wc = (n0 >> 3) + 4 * c4;           // first packed column of this lane
  const bool col_ok = (n0 + 32 * c4) < p.N;    // N % 32 == 0 is required, so all-or-nothing
  char* tile = smem + wave * TILE_BYTES;

  f32x4 acc[MB][8];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* xrow[MB];
  bool x_ok[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = 16 * mb + r16;
    x_ok[mb] = m < p.M;
    xrow[mb] = reinterpret_cast<const T*>(p.x) + (int64_t)(x_ok[mb] ? m : 0) * p.K + 8 * g;
  }

  int cur_group = -1;
  uint4 zq = uint4{0, 0, 0, 0};
  x8 sc[4];
  for (int s = s_begin; s < s_end; ++s) {
    const int k0 = s << 5;
    const int grp = k0 / p.G;
    if (grp != cur_group) {  // wave-uniform: a 32-row step never straddles a group (G % 32 == 0)
      cur_group = grp;
      if (col_ok) {
        zq = *reinterpret_cast<const uint4*>(p.qzeros + (int64_t)grp * Nc + wc);
#pragma unroll
        for (int w = 0; w < 4; ++w)
          sc[w] = *reinterpret_cast<const x8*>(reinterpret_cast<const T*>(p.scales) + (int64_t)grp * p.N + n0 + 32 * c4 + 8 * w);
      }
    }
    // ---- load + dequantise two 16-row halves into the wave's LDS tile
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int row = 16 * hf + kr;
      uint4 wq = uint4{0, 0, 0, 0};
      if (col_ok) wq = *reinterpret_cast<const uint4*>(p.qweight + (int64_t)(k0 + row) * Nc + wc);
      const uint32_t ww[4] = {wq.x, wq.y, wq.z, wq.w};
      const uint32_t zz[4] = {zq.x, zq.y, zq.z, zq.w};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o[j] = col_ok ? H::from_f32((float)(awq_nib(ww[w], j) - awq_nib(zz[w], j)) * H::to_f32(sc[w][j])) : (T)0.f;
        *reinterpret_cast<x8*>(tile + row * 256 + swz_w<128>(4 * c4 + w, row) * 16) = o;
      }
    }
    // ---- activations: lane (m = r16, g) holds x[m][k0 + 8g .. +8]
    x8 xf[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      if (x_ok[mb]) {
        xf[mb] = *reinterpret_cast<const x8*>(xrow[mb] + k0);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[mb][j] = (T)0.f;
      }
    }
    // ---- B fragments by transposed LDS reads; B[k = 8g + j][n]: rows 8g..8g+3 then 8g+4..8g+7
    const int q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int r_lo = 8 * g + q4, r_hi = r_lo + 4;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const int c = 2 * nb + (p4 >> 1);
      const x4 lo = H::ds_read_tr(tile + r_lo * 256 + swz_w<128>(c, r_lo) * 16 + 8 * (p4 & 1));
      const x4 hi = H::ds_read_tr(tile + r_hi * 256 + swz_w<128>(c, r_hi) * 16 + 8 * (p4 & 1));
      x8 wf;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wf[j] = lo[j];
        wf[4 + j] = hi[j];
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = H::mfma16(xf[mb], wf, acc[mb][nb]);
    }
  }

  // ---- reduce the 4 waves through LDS (wave 0 collects), then store
  // acc[mb][nb][r] = C[m = 16mb + 4g + r][n = n0 + 16nb + r16]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);  // [16*MB][128] fp32
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(16 * mb + 4 * g + r) * 128 + 16 * nb + r16] = acc[mb][nb][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mb][nb][r] += red[(16 * mb + 4 * g + r) * 128 + 16 * nb + r16];
    }
    __syncthreads();
  }
  if (wave == 0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(16 * mb + 4 * g + r) * 128 + 16 * nb + r16] = acc[mb][nb][r];
  }
  __syncthreads();
  // 256 threads: each finishes 8 consecutive columns of one row
  for (int ch = threadIdx.x; ch < 16 * MB * 16; ch += 256) {
    const int m = ch >> 4, nn = (ch & 15) * 8;
    if (m >= p.M || n0 + nn >= p.N) continue;
    const float* src = red + m * 128 + nn;
    if (p.SK > 1) {
      float* dst = p.slabs + ((int64_t)sk * p.M + m) * p.N + n0 + nn;
      *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
      *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(src + 4);
    } else {
      x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        T v = H::from_f32(src[j]);
        if (p.bias) v = H::from_f32(H::to_f32(v) + H::to_f32(reinterpret_cast<const T*>(p.bias)[n0 + nn + j]));
        o[j] = v;
      }
      *reinterpret_cast<x8*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.N + n0 + nn) = o;
    }
  }
}

template <int DTYPE>
__global__ __launch_bounds__(256) void awq_finalize_kernel(AwqArgs p) {
  using H = Half16<DTYPE>;
  using T = typename H::T;
  const int64_t total = (int64_t)p.M * p.N / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < p.SK; ++s) {
      const float* src = p.slabs + (int64_t)s * p.M * p.N + i * 8;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] += a[j];
        v[4 + j] += b[j];
      }
    }
    const int n = (int)((i * 8) % p.N);
    typename H::x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      T r = H::from_f32(v[j]);
      if (p.bias) r = H::from_f32(H::to_f32(r) + H::to_f32(reinterpret_cast<const T*>(p.bias)[n + j]));
      o[j] = r;
    }
    *reinterpret_cast<typename H::x8*>(reinterpret_cast<T*>(p.out) + i * 8) = o;
  }
}

template <int DTYPE, int MB>
int launch_gemm(const AwqArgs& p, hipStream_t s) {
  auto kern = awq_gemm_kernel<DTYPE, MB>;
  constexpr int lds = 4 * 32 * 256;  // 32 KB: four wave tiles; reused as the [64][128] fp32 reduce buffer
  hipLaunchKernelGGL(kern, dim3((unsigned)((p.N + 127) / 128), (unsigned)p.SK), dim3(256), lds, s, p);
  int rc = check_hip(hipGetLastError(), "awq_gemm launch");
  if (rc || p.SK == 1) return rc;
  const int64_t total = (int64_t)p.M * p.N / 8;
  hipLaunchKernelGGL((awq_finalize_kernel<DTYPE>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
  return check_hip(hipGetLastError(), "awq_finalize launch");
}

int check_awq(int64_t K, int64_t N, int64_t G, int dtype, const char* op) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "%s: bad dtype %d", op, dtype);
  SGLM_CHECK_ARG(K > 0 && N > 0 && N % 8 == 0, "%s: N (%ld) must be a positive multiple of 8", op, (long)N);
  SGLM_CHECK_ARG(G > 0 && K % G == 0, "%s: K (%ld) must be a multiple of the group size (%ld)", op, (long)K, (long)G);
  return 0;
}

}  // namespace
}  // namespace sglm

using namespace sglm;

extern "C" int sgl_mi355_awq_dequantize(
    const int32_t* qweight, const void* scales, const int32_t* qzeros, void* out, int64_t K, int64_t N,
    int64_t group_size, int dtype, void* stream) {
  int rc = check_awq(K, N, group_size, dtype, "awq_dequantize");
  if (rc) return rc;
  SGLM_CHECK_ARG(qweight && scales && qzeros && out, "awq_dequantize: null tensor pointer");
  SGLM_CHECK_ARG(K < (1ll << 31) && N < (1ll << 31), "awq_dequantize: shape too large");
  const int64_t total = K * (N / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 65535 * 8 ? (total + 255) / 256 : 65535 * 8);
  hipStream_t s = as_stream(stream);
  if (dtype == SGL_MI355_BF16)
    hipLaunchKernelGGL((awq_dequant_kernel<SGL_MI355_BF16>), dim3(grid), dim3(256), 0, s, (const uint32_t*)qweight,
                       (const __bf16*)scales, (const uint32_t*)qzeros, (__bf16*)out, (int)K, (int)(N / 8), (int)group_size);
  else
    hipLaunchKernelGGL((awq_dequant_kernel<SGL_MI355_FP16>), dim3(grid), dim3(256), 0, s, (const uint32_t*)qweight,
                       (const _Float16*)scales, (const uint32_t*)qzeros, (_Float16*)out, (int)K, (int)(N / 8), (int)group_size);
  return check_hip(hipGetLastError(), "awq_dequantize launch");
}

extern "C" int sgl_mi355_awq_gemm(
    const void* x, const int32_t* qweight, const void* scales, const int32_t* qzeros, const void* bias, void* out,
    float* workspace, int64_t workspace_floats, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype,
    void* stream) {
  int rc = check_awq(K, N, group_size, dtype, "awq_gemm");
  if (rc) return rc;
  SGLM_CHECK_ARG(M >= 0 && M <= 64, "awq_gemm: the fused kernel takes M <= 64 (got %ld); larger M goes through awq_dequantize + GEMM", (long)M);
  SGLM_CHECK_ARG(N % 32 == 0 && K % 32 == 0 && group_size % 32 == 0, "awq_gemm: N, K and group_size must be multiples of 32");
  if (M == 0) return 0;
  SGLM_CHECK_ARG(x && qweight && scales && qzeros && out, "awq_gemm: null tensor pointer");
  // split K over workgroups until there are ~2 per CU (narrow layers), if the workspace allows it
  const int64_t ntiles = (N + 127) / 128;
  int sk = 1;
  while (sk < 16 && ntiles * sk < 384 && (K / 32) / (sk * 2) >= 4) sk *= 2;
  if (sk > 1 && (workspace == nullptr || workspace_floats < (int64_t)sk * M * N)) sk = 1;
  AwqArgs p{x, (const uint32_t*)qweight, scales, (const uint32_t*)qzeros, bias, out, workspace,
            (int)M, (int)N, (int)K, (int)group_size, sk};
  hipStream_t s = as_stream(stream);
#define AWQ_GO(DT)                                      \
  if (M <= 16) return launch_gemm<DT, 1>(p, s);         \
  if (M <= 32) return launch_gemm<DT, 2>(p, s);         \
  return launch_gemm<DT, 4>(p, s)
  if (dtype == SGL_MI355_BF16) { AWQ_GO(SGL_MI355_BF16); }
  AWQ_GO(SGL_MI355_FP16);
#undef AWQ_GO
}
