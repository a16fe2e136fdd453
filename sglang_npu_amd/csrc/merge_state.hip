// merge_state: combine two partial attention results by their log-sum-exp (cascade / chunked-prefix prefill).
// Replaces: merge_state_triton  python/sglang/srt/layers/attention/triton_ops/merge_state.py:8-96
//           merge_state / merge_state_v2  sgl-kernel/csrc/attention/merge_attn_states.cu (tests
//           sgl-kernel/tests/test_merge_state_v2.py)
//   lse == +inf is treated as -inf (merge_state.py:29-30); max = max(p, s); out_se = e^(p-max) + e^(s-max);
//   out = p_out * e^(p-max)/out_se + s_out * e^(s-max)/out_se (fp32, one rounding); out_lse = log(out_se) + max.
// HBM-bound byte mover: one wave per (token, head), 16-B loads where the head size allows.
#include "common.h"

namespace sglm {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void merge_state_kernel(const T* __restrict__ p_out, const float* __restrict__ p_lse,
                                                          const T* __restrict__ s_out, const float* __restrict__ s_lse,
                                                          T* __restrict__ out, float* __restrict__ out_lse,
                                                          int64_t pairs, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= pairs) return;
  float pl = p_lse[pair], sl = s_lse[pair];
  pl = pl == INFINITY ? -INFINITY : pl;
  sl = sl == INFINITY ? -INFINITY : sl;
  const float mx = fmaxf(pl, sl);
  const float pe = expf(pl - mx), se = expf(sl - mx);
  const float out_se = pe + se;
  if (out_lse && lane == 0) out_lse[pair] = logf(out_se) + mx;
  const float ps = pe / out_se, ss = se / out_se;
  const T* a = p_out + pair * D;
  const T* b = s_out + pair * D;
  T* o = out + pair * D;
  for (int d = lane; d < D; d += 64) o[d] = (T)((float)a[d] * ps + (float)b[d] * ss);
}

template <typename T>
int launch(const void* p_out, const float* p_lse, const void* s_out, const float* s_lse, void* out, float* out_lse,
           int64_t pairs, int D, hipStream_t s) {
  hipLaunchKernelGGL((merge_state_kernel<T>), dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, (const T*)p_out, p_lse,
                     (const T*)s_out, s_lse, (T*)out, out_lse, pairs, D);
  return check_hip(hipGetLastError(), "merge_state launch");
}

}  // namespace
}  // namespace sglm

using namespace sglm;

extern "C" int sgl_mi355_merge_state(const void* prefix_output, const float* prefix_lse, const void* suffix_output,
                                     const float* suffix_lse, void* output, float* output_lse, int64_t num_tokens,
                                     int64_t num_heads, int64_t head_size, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16 || dtype == SGL_MI355_FP32,
                 "merge_state: dtype must be bf16 (0), fp16 (1) or fp32 (2)");
  SGLM_CHECK_ARG(num_tokens >= 0 && num_heads > 0 && head_size > 0 && head_size <= 4096, "merge_state: bad shape");
  SGLM_CHECK_ARG(num_tokens * num_heads < (1ll << 33), "merge_state: too many (token, head) pairs");
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(prefix_output && prefix_lse && suffix_output && suffix_lse && output, "merge_state: null tensor pointer");
  const int64_t pairs = num_tokens * num_heads;
  hipStream_t s = as_stream(stream);
  if (dtype == SGL_MI355_BF16)
    return launch<__bf16>(prefix_output, prefix_lse, suffix_output, suffix_lse, output, output_lse, pairs, (int)head_size, s);
  if (dtype == SGL_MI355_FP16)
    return launch<_Float16>(prefix_output, prefix_lse, suffix_output, suffix_lse, output, output_lse, pairs, (int)head_size, s);
  return launch<float>(prefix_output, prefix_lse, suffix_output, suffix_lse, output, output_lse, pairs, (int)head_size, s);
}
