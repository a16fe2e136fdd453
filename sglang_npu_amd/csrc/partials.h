// Helpers shared by the kernels that consume a split-K GEMM while it is still partial sums (elementwise.hip,
// attention_decode.hip): the fp8_scaled_mm epilogue applied on the fly, and the RoPE rotation every RoPE kernel uses.
#pragma once
#include "common.h"

namespace sglm {

// Source of a row that is still a split-K GEMM in flight (sgl_mi355_fp8_scaled_mm_partials): the consumer applies the
// fp8_scaled_mm epilogue itself -- slices summed in slice order, x w_scale[col], x x_scale[row], + bias, ONE rounding to
// the 16-bit dtype (fp8_gemm_finalize_kernel, gemm_fp8.hip) -- and continues with the rounded value, so the result is
// bit-identical to running the GEMM to completion first.  Saves the finalize launch and one activation round trip.
struct PartialSrc {
  const float* partials;  // [num_slices][M][N] fp32
  int num_slices;
  int64_t slice_stride;   // M * N
  const float* sa;        // x_scale [M]
  const float* sb;        // w_scale [N]
  const void* bias;       // 16-bit [N] or null
  int N;
};

template <int DTYPE>
__device__ __forceinline__ typename Half16<DTYPE>::x8 gemm_row8(const PartialSrc& ps, int64_t row, int col) {
  using Hh = Half16<DTYPE>;
  float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float* src = ps.partials + row * ps.N + col;
  // EIGHT slices' loads in flight at a time: the decode GEMMs leave 6-8 slices, so the whole reduction is one memory
  // round trip (a plain slice loop serialises one per slice; four at a time made it two).  The sums still run in
  // slice order, as in fp8_gemm_finalize_kernel.
  constexpr int U = 8;
  const float sa = ps.sa[row];
  const f32x4 sb0 = *reinterpret_cast<const f32x4*>(ps.sb + col), sb1 = *reinterpret_cast<const f32x4*>(ps.sb + col + 4);
  for (int s0 = 0; s0 < ps.num_slices; s0 += U) {
    f32x4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int si = s0 + u < ps.num_slices ? s0 + u : ps.num_slices - 1;
      a[u] = *reinterpret_cast<const f32x4*>(src + si * ps.slice_stride);
      b[u] = *reinterpret_cast<const f32x4*>(src + si * ps.slice_stride + 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s0 + u < ps.num_slices) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] += a[u][j];
          v[4 + j] += b[u][j];
        }
      }
  }
  typename Hh::x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma clang fp contract(off)  // explicit roundings (no fma contraction): must match fp8_gemm_finalize_kernel bit for bit
    float r = (v[j] * (j < 4 ? sb0[j] : sb1[j - 4])) * sa;
    if (ps.bias) r = r + Hh::to_f32(reinterpret_cast<const typename Hh::T*>(ps.bias)[col + j]);
    o[j] = Hh::from_f32(r);
  }
  return o;
}
template <int DTYPE>
__device__ __forceinline__ typename Half16<DTYPE>::T gemm_elem(const PartialSrc& ps, int64_t row, int col) {
  using Hh = Half16<DTYPE>;
  float v = 0.f;
  const float* src = ps.partials + row * ps.N + col;
  for (int s0 = 0; s0 < ps.num_slices; s0 += 8) {
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = src[(s0 + u < ps.num_slices ? s0 + u : ps.num_slices - 1) * ps.slice_stride];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < ps.num_slices) v += a[u];
  }
  float r;
  {
#pragma clang fp contract(off)
    r = (v * ps.sb[col]) * ps.sa[row];
    if (ps.bias) r = r + Hh::to_f32(reinterpret_cast<const typename Hh::T*>(ps.bias)[col]);
  }
  return Hh::from_f32(r);
}

// One rotation, every product and sum rounded on its own (no fma contraction): the four RoPE kernels below must agree
// bit for bit -- the fused paths are tested against the unfused sequence with torch.equal.
__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float s, float& o1, float& o2) {
#pragma clang fp contract(off)  // __fmul_rn & co. are plain operators in the HIP headers and DO get fused otherwise
  const float a = x1 * c, b = x2 * s, d = x2 * c, e = x1 * s;
  o1 = a - b;
  o2 = d + e;
}

}  // namespace sglm
