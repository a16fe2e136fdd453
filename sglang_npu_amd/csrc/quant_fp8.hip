// The other two FP8 activation quantisers of the quant-linear path (SURVEY 8b "next" ops), MI355X / gfx950.
//
// Replaces:
//   * sgl_per_token_group_quant_fp8   sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu:15-215
//                                     (schema csrc/common_extension.cc:116-119; Triton twin
//                                     python/sglang/srt/layers/quantization/fp8_kernel.py:115-155, on HIP the
//                                     per-token path of apply_fp8_linear is this op with group_size = K, fp8_utils.py:676-678)
//   * sgl_per_tensor_quant_fp8        sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu:9-120 (schema common_extension.cc:126-127)
//
// Both are byte movers (read 2 B, write 1 B per element): 16-B vector loads, everything between the load and the
// store in registers.  Arithmetic follows the reference to the rounding:
//   group:   absmax starts at eps; y_s = absmax / fp8_max; q = clamp(x / y_s, fp8_min, fp8_max) -- a DIVISION here
//            (per_token_group_quant_8bit.cu:99), unlike the per-token op's reciprocal multiply;
//   tensor:  scale = max over the tensor of |x| / 448 (atomicMax into the caller's zero-initialised word),
//            q = clamp(x * (1 / scale), -448, 448) (per_tensor_quant_fp8.cu:47-61).
// gfx950 FP8 is OCP e4m3fn (max 448), so the reference's fnuz branch (USE_ROCM on gfx94x) does not apply.
#include "common.h"

namespace sglm {
namespace {

template <int DTYPE>
struct In {  // 8 consecutive input elements as floats
  static __device__ __forceinline__ void load8(const void* base, int64_t elem, float* f) {
    using Hh = Half16<DTYPE>;
    const typename Hh::x8 v = *reinterpret_cast<const typename Hh::x8*>(reinterpret_cast<const typename Hh::T*>(base) + elem);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = Hh::to_f32(v[j]);
  }
};
template <>
struct In<2> {  // float32 input
  static __device__ __forceinline__ void load8(const void* base, int64_t elem, float* f) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem);
    const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[j] = a[j];
      f[4 + j] = b[j];
    }
  }
};

__device__ __forceinline__ uint2 pack8(const float* f) {
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return uint2{(unsigned)lo, (unsigned)hi};
}

// 16 lanes per group (four groups per wave), VPL 8-element vectors per lane: group_size = 128 * VPL / ... see launcher.
// The group's elements stay in registers between the absmax pass and the quantising pass (one read of the input).
// UE8M0 (round 3): the scale is rounded UP to a power of two, y_s = exp2(ceil(log2(max(absmax / fp8_max, 1e-10)))), and
// stored as its biased exponent byte, four to an int32, column-major (per_token_group_quant_8bit.cu:52-60, 87-96): byte
// (col / 4) * s_stride_group * 4 + row * 4 + col % 4 (s_stride_group = int32 elements between packed columns).  The
// exponent comes from the float's bits -- ceil(log2 y) = exponent + (mantissa != 0) for a normal y, and y >= 1e-10 is one
// -- which is what exp2f(ceilf(log2f(y))) evaluates to for every y a 16-bit absmax / 448 can be.
template <int DTYPE, int VPL, bool UE8M0 = false>
__global__ __launch_bounds__(256) void group_quant_kernel(
    const void* __restrict__ x, uint8_t* __restrict__ q, float* __restrict__ s, int64_t num_groups, int group_size,
    int groups_per_row, int64_t s_stride_row, int64_t s_stride_group, float eps, float fmin_, float fmax_) {
  const int lane16 = threadIdx.x & 15;
  const int64_t grp = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = grp < num_groups;
  const int64_t g = live ? grp : num_groups - 1;   // dead groups shadow the last one (no divergent shuffles), never store
  const int nvec = group_size >> 3;
  float v[VPL][8];
  float amax = eps;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane16 + 16 * i;
    if (vi < nvec) {
      In<DTYPE>::load8(x, g * group_size + 8 * vi, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[i][j]));
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  float y_s = amax / fmax_;
  if constexpr (UE8M0) {
    const uint32_t bits = __float_as_uint(fmaxf(y_s, 1e-10f));
    const uint32_t e = (bits >> 23) + ((bits & 0x7FFFFFu) != 0u ? 1u : 0u);  // biased exponent of the next power of two
    y_s = __uint_as_float(e << 23);
    if (live && lane16 == 0) {
      const int64_t row = g / groups_per_row, col = g % groups_per_row;
      reinterpret_cast<uint8_t*>(s)[(col >> 2) * s_stride_group * 4 + row * 4 + (col & 3)] = (uint8_t)e;
    }
  } else {
    if (live && lane16 == 0) s[(g / groups_per_row) * s_stride_row + (g % groups_per_row) * s_stride_group] = y_s;
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane16 + 16 * i;
    if (live && vi < nvec) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(v[i][j] / y_s, fmin_), fmax_);
      *reinterpret_cast<uint2*>(q + g * group_size + 8 * vi) = pack8(f);
    }
  }
}

template <int DTYPE>
__global__ __launch_bounds__(256) void tensor_absmax_kernel(const void* __restrict__ x, float* __restrict__ s, int64_t n) {
  __shared__ float red[4];
  float amax = 0.f;
  const int64_t nvec = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float f[8];
    In<DTYPE>::load8(x, 8 * i, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {  // tail elements
    const int64_t e = (nvec << 3) + threadIdx.x;
    float val;
    if constexpr (DTYPE == 2) val = reinterpret_cast<const float*>(x)[e];
    else val = Half16<DTYPE == 2 ? 0 : DTYPE>::to_f32(reinterpret_cast<const typename Half16<DTYPE == 2 ? 0 : DTYPE>::T*>(x)[e]);
    amax = fmaxf(amax, fabsf(val));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // non-negative floats order like their bit patterns: the reference's atomicMaxFloat (utils.h) on an unsigned word
    atomicMax(reinterpret_cast<unsigned int*>(s), __float_as_uint(amax / 448.0f));
  }
}

template <int DTYPE>
__global__ __launch_bounds__(256) void tensor_quant_kernel(const void* __restrict__ x, uint8_t* __restrict__ q,
                                                           const float* __restrict__ s, int64_t n) {
  const float inv = 1.0f / *s;
  const int64_t nvec = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float f[8];
    In<DTYPE>::load8(x, 8 * i, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fmaxf(fminf(f[j] * inv, 448.0f), -448.0f);
    *reinterpret_cast<uint2*>(q + 8 * i) = pack8(f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t e = (nvec << 3) + threadIdx.x;
    float val;
    if constexpr (DTYPE == 2) val = reinterpret_cast<const float*>(x)[e];
    else val = Half16<DTYPE == 2 ? 0 : DTYPE>::to_f32(reinterpret_cast<const typename Half16<DTYPE == 2 ? 0 : DTYPE>::T*>(x)[e]);
    val = fmaxf(fminf(val * inv, 448.0f), -448.0f);
    q[e] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(val, 0.f, 0, false) & 0xff);
  }
}

template <int DTYPE, bool UE8M0 = false>
int launch_group(const void* x, void* q, float* s, int64_t num_groups, int group_size, int groups_per_row, int64_t ssr,
                 int64_t ssg, float eps, float mn, float mx, hipStream_t st) {
  const unsigned blocks = (unsigned)((num_groups + 15) / 16);
  const int vpl = (group_size / 8 + 15) / 16;
#define GQ(V)                                                                                                         \
  hipLaunchKernelGGL((group_quant_kernel<DTYPE, V, UE8M0>), dim3(blocks), dim3(256), 0, st, x, (uint8_t*)q, s, num_groups, \
                     group_size, groups_per_row, ssr, ssg, eps, mn, mx)
  if (vpl <= 1) GQ(1);
  else if (vpl <= 2) GQ(2);
  else if (vpl <= 4) GQ(4);
  else GQ(8);
#undef GQ
  return check_hip(hipGetLastError(), "per_token_group_quant_fp8 launch");
}

}  // namespace
}  // namespace sglm

using namespace sglm;

extern "C" int sgl_mi355_per_token_group_quant_fp8(
    const void* input, void* output_q, float* output_s, int64_t num_tokens, int64_t hidden_dim, int64_t group_size,
    int64_t s_stride_token, int64_t s_stride_group, float eps, float fp8_min, float fp8_max, int scale_ue8m0, int dtype,
    void* stream) {
  SGLM_CHECK_ARG(dtype >= 0 && dtype <= 2, "per_token_group_quant_fp8: bad dtype %d", dtype);
  SGLM_CHECK_ARG(num_tokens >= 0 && hidden_dim > 0, "per_token_group_quant_fp8: bad shape");
  SGLM_CHECK_ARG(group_size >= 8 && group_size % 8 == 0 && group_size <= 1024 && hidden_dim % group_size == 0,
                 "per_token_group_quant_fp8: group_size (%ld) must be a multiple of 8, at most 1024, and divide the hidden "
                 "dimension (%ld)", (long)group_size, (long)hidden_dim);
  SGLM_CHECK_ARG(fp8_max > 0.f && fp8_min < 0.f, "per_token_group_quant_fp8: bad fp8_min / fp8_max");
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(input && output_q && output_s, "per_token_group_quant_fp8: null tensor pointer");
  const int gpr = (int)(hidden_dim / group_size);
  const int64_t ng = num_tokens * gpr;
  hipStream_t st = as_stream(stream);
  if (scale_ue8m0) {
    // output_s: int32 [num_tokens][ceil(gpr / 4)] column-major (s_stride_token == 1: the kernel asserts is_column_major,
    // per_token_group_quant_8bit.cu:171-186), s_stride_group = int32 elements between packed columns >= num_tokens
    SGLM_CHECK_ARG((s_stride_token == 1 || num_tokens == 1) && s_stride_group >= num_tokens,
                   "per_token_group_quant_fp8 (UE8M0): output_s must be the column-major packed int32 tensor "
                   "(stride(0) == 1, stride(1) >= num_tokens; got %ld, %ld)", (long)s_stride_token, (long)s_stride_group);
    if (dtype == SGL_MI355_BF16)
      return launch_group<SGL_MI355_BF16, true>(input, output_q, output_s, ng, (int)group_size, gpr, 0, s_stride_group, eps, fp8_min, fp8_max, st);
    if (dtype == SGL_MI355_FP16)
      return launch_group<SGL_MI355_FP16, true>(input, output_q, output_s, ng, (int)group_size, gpr, 0, s_stride_group, eps, fp8_min, fp8_max, st);
    return launch_group<2, true>(input, output_q, output_s, ng, (int)group_size, gpr, 0, s_stride_group, eps, fp8_min, fp8_max, st);
  }
  if (dtype == SGL_MI355_BF16)
    return launch_group<SGL_MI355_BF16>(input, output_q, output_s, ng, (int)group_size, gpr, s_stride_token, s_stride_group, eps, fp8_min, fp8_max, st);
  if (dtype == SGL_MI355_FP16)
    return launch_group<SGL_MI355_FP16>(input, output_q, output_s, ng, (int)group_size, gpr, s_stride_token, s_stride_group, eps, fp8_min, fp8_max, st);
  return launch_group<2>(input, output_q, output_s, ng, (int)group_size, gpr, s_stride_token, s_stride_group, eps, fp8_min, fp8_max, st);
}

extern "C" int sgl_mi355_per_tensor_quant_fp8(const void* input, void* output_q, float* output_s, int64_t num_elements,
                                              int is_static, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype >= 0 && dtype <= 2, "per_tensor_quant_fp8: bad dtype %d", dtype);
  SGLM_CHECK_ARG(num_elements >= 0, "per_tensor_quant_fp8: bad size");
  if (num_elements == 0) return 0;
  SGLM_CHECK_ARG(input && output_q && output_s, "per_tensor_quant_fp8: null tensor pointer");
  SGLM_CHECK_ARG(((uintptr_t)input & 15) == 0 && ((uintptr_t)output_q & 7) == 0, "per_tensor_quant_fp8: unaligned tensor");
  hipStream_t st = as_stream(stream);
  const int64_t nvec = (num_elements + 7) / 8;
  const unsigned blocks = (unsigned)(nvec < 256 * 2048 ? (nvec + 255) / 256 : 2048);
#define PT(D)                                                                                                   \
  do {                                                                                                          \
    if (!is_static)                                                                                             \
      hipLaunchKernelGGL((tensor_absmax_kernel<D>), dim3(blocks), dim3(256), 0, st, input, output_s, num_elements); \
    hipLaunchKernelGGL((tensor_quant_kernel<D>), dim3(blocks), dim3(256), 0, st, input, (uint8_t*)output_q,    \
                       (const float*)output_s, num_elements);                                                   \
  } while (0)
  if (dtype == SGL_MI355_BF16) PT(SGL_MI355_BF16);
  else if (dtype == SGL_MI355_FP16) PT(SGL_MI355_FP16);
  else PT(2);
#undef PT
  return check_hip(hipGetLastError(), "per_tensor_quant_fp8 launch");
}
