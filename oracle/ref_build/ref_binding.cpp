// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Registration shim for the *unmodified* reference CPU kernels of the hot path.  The
// functions declared below are defined in the reference tree
// (/root/reference/sgl-kernel/csrc/cpu/decode.cpp:1375, extend.cpp:579, norm.cpp:244,273,
// activation.cpp:59, rope.cpp:241); the Makefile next to this file compiles those
// translation units where they lie and links them with this shim into
// oracle/_ref/libsgl_ref_cpu.so.  Nothing from the reference is copied into this repository.
//
// The reference registers its ops through torch_extension_cpu.cpp, which drags in all
// nineteen translation units (MoE, shm, numa ...).  We only need the hot path, so we
// register just these under our own namespace `sgl_ref` (declarations as
// torch_extension_cpu.cpp:23-33,221-227).
#include <ATen/ATen.h>
#include <torch/library.h>

void decode_attention_cpu(
    at::Tensor& query,
    at::Tensor& k_cache,
    at::Tensor& v_cache,
    at::Tensor& output,
    at::Tensor& key,
    at::Tensor& value,
    at::Tensor& loc,
    at::Tensor& attn_logits,
    at::Tensor& req_to_token,
    at::Tensor& req_pool_indices,
    at::Tensor& seq_lens,
    double sm_scale,
    double logit_cap);

void extend_attention_cpu(
    at::Tensor& q_extend,
    at::Tensor& k_extend,
    at::Tensor& v_extend,
    at::Tensor& o_extend,
    at::Tensor& k_buffer,
    at::Tensor& v_buffer,
    at::Tensor& req_to_token,
    at::Tensor& req_pool_indices,
    at::Tensor& seq_lens,
    at::Tensor& extend_seq_lens,
    at::Tensor& extend_start_loc,
    int64_t max_len_extend,
    double sm_scale,
    double logit_cap);

at::Tensor silu_and_mul_cpu(at::Tensor& input);
at::Tensor rmsnorm_cpu(at::Tensor& input, at::Tensor& weight, double eps);
void fused_add_rmsnorm_cpu(at::Tensor& input, at::Tensor& residual, at::Tensor& weight, double eps);
std::tuple<at::Tensor, at::Tensor> rotary_embedding_cpu(
    at::Tensor& positions,
    at::Tensor& query,
    at::Tensor& key,
    int64_t head_size,
    at::Tensor& cos_sin_cache,
    bool is_neox);

TORCH_LIBRARY(sgl_ref, m) {
  m.def("silu_and_mul_cpu(Tensor input) -> Tensor");
  m.impl("silu_and_mul_cpu", c10::DispatchKey::CPU, &silu_and_mul_cpu);
  m.def("rmsnorm_cpu(Tensor input, Tensor weight, float eps) -> Tensor");
  m.impl("rmsnorm_cpu", c10::DispatchKey::CPU, &rmsnorm_cpu);
  m.def("fused_add_rmsnorm_cpu(Tensor(a!) input, Tensor(b!) residual, Tensor weight, float eps) -> ()");
  m.impl("fused_add_rmsnorm_cpu", c10::DispatchKey::CPU, &fused_add_rmsnorm_cpu);
  m.def(
      "rotary_embedding_cpu(Tensor positions, Tensor query, Tensor key, int head_size, Tensor cos_sin_cache, "
      "bool is_neox) -> (Tensor, Tensor)");
  m.impl("rotary_embedding_cpu", c10::DispatchKey::CPU, &rotary_embedding_cpu);
  m.def(
      "decode_attention_cpu(Tensor query, Tensor k_cache, Tensor v_cache, Tensor(a!) output, Tensor key, "
      "Tensor value, Tensor loc, Tensor(b!) attn_logits, Tensor req_to_token, Tensor req_pool_indices, "
      "Tensor seq_lens, float sm_scale, float logit_cap) -> ()");
  m.impl("decode_attention_cpu", c10::DispatchKey::CPU, &decode_attention_cpu);
  m.def(
      "extend_attention_cpu(Tensor q_extend, Tensor k_extend, Tensor v_extend, Tensor(a!) o_extend, "
      "Tensor k_buffer, Tensor v_buffer, Tensor req_to_token, Tensor req_pool_indices, Tensor seq_lens, "
      "Tensor extend_seq_lens, Tensor extend_start_loc, int max_len_extend, float sm_scale, "
      "float logit_cap) -> ()");
  m.impl("extend_attention_cpu", c10::DispatchKey::CPU, &extend_attention_cpu);
}
