"""``torch.ops.sgl_kernel.*`` registration for the MI355X backend.

SGLang reaches its native kernels through the torch dispatcher: ``python/sgl_kernel/*.py`` and the CPU backend call
``torch.ops.sgl_kernel.<op>`` (e.g. ``intel_amx_backend.py:91-125``, ``sgl_kernel/gemm.py:7-42,100-145``,
``sgl_kernel/elementwise.py``), and the reference registers those names in
``sgl-kernel/csrc/common_extension.cc:56-130`` / ``csrc/cpu/torch_extension_cpu.cpp:263-275`` /
``csrc/torch_extension_rocm.cc:21-116`` with ``TORCH_LIBRARY_FRAGMENT(sgl_kernel, m)``.

``register()`` puts the SAME schemas (argument names, order, mutability annotations) into the same ``sgl_kernel``
namespace through ``torch.library`` and binds each to this backend for the ``CUDA`` dispatch key (ROCm devices are
"cuda" to torch), so an SGLang process that imports this module instead of the ``sgl_kernel`` wheel resolves
``torch.ops.sgl_kernel.fp8_scaled_mm`` etc. to the HIP library.  The two attention ops keep the names the reference
gave them (``decode_attention_cpu`` / ``extend_attention_cpu`` -- the only op-level attention ABI in the tree,
SURVEY 8b) and are registered for device tensors.

If the real ``sgl_kernel`` extension is already loaded in the process its definitions exist and ``define`` would
clash: ``register()`` then only adds the CUDA implementations it can (``impl`` on an existing schema), or leaves the op
alone when one is present, and reports what it did.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops

_SCHEMAS = {
    # sgl-kernel/csrc/common_extension.cc:98-130
    "awq_dequantize": "(Tensor qweight, Tensor scales, Tensor qzeros) -> Tensor",
    "fp8_scaled_mm": "(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, ScalarType out_dtype, Tensor? bias) -> Tensor",
    "sgl_per_token_group_quant_fp8": "(Tensor input, Tensor output_q, Tensor output_s, int group_size, float eps, "
                                     "float fp8_min, float fp8_max, bool scale_ue8m0) -> ()",
    "sgl_per_tensor_quant_fp8": "(Tensor input, Tensor output_q, Tensor output_s, bool is_static) -> ()",
    "sgl_per_token_quant_fp8": "(Tensor input, Tensor output_q, Tensor output_s) -> ()",
    # common_extension.cc:56-93
    "merge_state": "(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor! v_merged, Tensor! s_merged) -> ()",
    "merge_state_v2": "(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor! v_merged, Tensor! s_merged) -> ()",
    "rmsnorm": "(Tensor! output, Tensor input, Tensor weight, float eps, bool enable_pdl) -> ()",
    "fused_add_rmsnorm": "(Tensor! input, Tensor! residual, Tensor weight, float eps, bool enable_pdl) -> ()",
    "silu_and_mul": "(Tensor! out, Tensor input) -> ()",
    "apply_rope_pos_ids_cos_sin_cache": "(Tensor q, Tensor k, Tensor! q_rope, Tensor! k_rope, Tensor cos_sin_cache, "
                                        "Tensor pos_ids, bool interleave, int cuda_stream) -> ()",
    # sgl-kernel/csrc/cpu/torch_extension_cpu.cpp:263-275 (sic: "v_cahce")
    "decode_attention_cpu": "(Tensor query, Tensor k_cache, Tensor v_cahce, Tensor output, Tensor key, Tensor value, "
                            "Tensor loc, Tensor attn_logits, Tensor req_to_token, Tensor req_pool_indices, "
                            "Tensor seq_lens, float sm_scale, float logit_cap) -> ()",
    "extend_attention_cpu": "(Tensor q_extend, Tensor k_extend, Tensor v_extend, Tensor o_extend, Tensor k_buffer, "
                            "Tensor v_buffer, Tensor req_to_token, Tensor req_pool_indices, Tensor seq_lens, "
                            "Tensor extend_seq_lens, Tensor extend_start_loc, int max_len_extend, float sm_scale, "
                            "float logit_cap) -> ()",
}


def _rmsnorm(output, input, weight, eps, enable_pdl):
    ops.rmsnorm(input, weight, eps, out=output)


def _fused_add_rmsnorm(input, residual, weight, eps, enable_pdl):
    ops.fused_add_rmsnorm(input, residual, weight, eps)


def _silu_and_mul(out, input):
    ops.silu_and_mul(input, out=out)


def _merge_state(v_a, s_a, v_b, s_b, v_merged, s_merged):
    ops.merge_state(v_a, s_a, v_b, s_b, v_merged, s_merged)


def _apply_rope(q, k, q_rope, k_rope, cos_sin_cache, pos_ids, interleave, cuda_stream):
    """sgl_kernel.apply_rope_with_cos_sin_cache_inplace (elementwise.py) passes q.view(T, -1, head), the same tensors as
    outputs, interleave = not is_neox, and the raw stream handle (ignored: the caller's current stream is used)."""
    if q_rope.data_ptr() != q.data_ptr():
        q_rope.copy_(q)
    if k_rope.data_ptr() != k.data_ptr():
        k_rope.copy_(k)
    head = q.size(-1)
    ops.apply_rope_with_cos_sin_cache_inplace(pos_ids, q_rope.view(q.size(0), -1), k_rope.view(k.size(0), -1), head,
                                              cos_sin_cache, not interleave)


_IMPLS = {
    "awq_dequantize": ops.awq_dequantize,
    "fp8_scaled_mm": ops.fp8_scaled_mm,
    "sgl_per_token_group_quant_fp8": ops.sgl_per_token_group_quant_fp8,
    "sgl_per_tensor_quant_fp8": ops.sgl_per_tensor_quant_fp8,
    "sgl_per_token_quant_fp8": ops.sgl_per_token_quant_fp8,
    "merge_state": _merge_state,
    "merge_state_v2": _merge_state,
    "rmsnorm": _rmsnorm,
    "fused_add_rmsnorm": _fused_add_rmsnorm,
    "silu_and_mul": _silu_and_mul,
    "apply_rope_pos_ids_cos_sin_cache": _apply_rope,
    "decode_attention_cpu": ops.decode_attention,
    "extend_attention_cpu": ops.extend_attention,
}

_libs = []
_report: Optional[Dict[str, str]] = None


def _has_schema(name: str) -> bool:
    try:
        torch._C._dispatch_find_schema_or_throw(f"sgl_kernel::{name}", "")
        return True
    except RuntimeError:
        return False


def _has_cuda_kernel(name: str) -> bool:
    try:
        return torch._C._dispatch_has_kernel_for_dispatch_key(f"sgl_kernel::{name}", "CUDA")
    except RuntimeError:
        return False


def register() -> Dict[str, str]:
    """Define (where absent) and implement the ``sgl_kernel`` ops for device tensors.  Idempotent.  Returns
    {op: "defined+impl" | "impl" | "kept existing"}."""
    global _report
    if _report is not None:
        return _report
    frag = torch.library.Library("sgl_kernel", "FRAGMENT")
    _libs.append(frag)  # registrations live as long as the Library object
    report = {}
    for name, schema in _SCHEMAS.items():
        if _has_schema(name):
            if _has_cuda_kernel(name):
                report[name] = "kept existing"
                continue
            frag.impl(name, _IMPLS[name], "CUDA")
            report[name] = "impl"
        else:
            frag.define(name + schema)
            frag.impl(name, _IMPLS[name], "CUDA")
            report[name] = "defined+impl"
    _report = report
    return report
