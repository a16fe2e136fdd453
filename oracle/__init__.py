"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MI355X hot path.

Thin ctypes wrapper over ``oracle/sgl_oracle.c`` (a plain-C restatement of the
reference's algorithms, each function citing the reference file:line it follows).
The signatures mirror the reference ops (``torch.ops.sgl_kernel.decode_attention_cpu``
etc., sgl-kernel/csrc/cpu/torch_extension_cpu.cpp:263-275 and
sgl-kernel/python/sgl_kernel/gemm.py:7-42,140-145) so that the parity tests read like
the reference's own tests.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``sglang_npu_amd``) never does: it fails loudly when
its HIP library is missing instead of falling back to anything here.

Parity pin: see ``tests/golden/make_golden.py`` -- this oracle was checked in the build
container against the reference's own compiled CPU kernels (``oracle/_ref``) and the
pure-torch references embedded in the reference's tests; the vectors are committed
under ``tests/golden/``.  Round-2 additions and what pins them:
* float8_e5m2 conversions (``f32_to_e5m2`` / ``e5m2_to_f32``, ``kv_dtype=`` on the ``*_fp8kv`` functions): torch's own
  cast, checked here against ``x.to(torch.float8_e5m2)`` on every fp16 and bf16 bit pattern and on fp32 samples
  (tests/test_fp8kv_e5m2_gpu.py::test_e5m2_cast_every_16_bit_value_matches_torch);
* ``oracle/quick_reduce.py`` (numpy): the QuickReduce codec formulas of quick_all_reduce.cuh and the bound of the
  reference's test_quick_allreduce.py; the reference's device code cannot run in the build container, so it is pinned
  to the codec definition and that bound only ("parity unpinned" against the reference kernel's own output -- said so in
  its header).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "sgl_oracle.c")
_LIB_PATH = os.path.join(_HERE, "_build", "libsgl_oracle.so")
_lib = None

_c = ctypes
_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_I = ctypes.c_int
_F = ctypes.c_float


def build(native: bool = False, out: Optional[str] = None, force: bool = False) -> str:
    """Compile the C oracle with gcc.  ``native=True`` uses -march=native (for the
    cpu_baseline timing on the machine that runs it); the default build targets
    x86-64-v3 so the prebuilt .so also runs on the GPU box's host CPU."""
    out = out or _LIB_PATH
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(_SRC):
        return out
    arch = "-march=native" if native else "-march=x86-64-v3"
    cmd = ["gcc", "-O3", arch, "-fopenmp", "-shared", "-fPIC", "-o", out, _SRC, "-lm"]
    subprocess.check_call(cmd)
    return out


def load(path: Optional[str] = None):
    global _lib
    if path is not None:
        return ctypes.CDLL(path)
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def num_threads() -> int:
    return int(load().orc_num_threads())


# ----------------------------------------------------------------------------- helpers
def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return 0
    if t.dtype == torch.float16:
        return 1
    raise TypeError(f"oracle: expected bfloat16/float16, got {t.dtype}")


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.device.type == "cpu", "oracle runs on CPU tensors"
    return ctypes.c_void_p(t.data_ptr())


def _is64(t: torch.Tensor) -> int:
    if t.dtype == torch.int64:
        return 1
    if t.dtype == torch.int32:
        return 0
    raise TypeError(f"oracle: expected int32/int64 index tensor, got {t.dtype}")


def _i64(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int64 else t.to(torch.int64)


# ----------------------------------------------------------------------------- page table
def create_kv_indices(req_to_token, req_pool_indices, page_kernel_lens, kv_indptr, kv_start_idx, kv_indices,
                      lib=None):
    """create_flashinfer_kv_indices_triton (layers/attention/utils.py:10-46)."""
    lib = lib or load()
    assert req_to_token.dtype == torch.int32 and kv_indptr.dtype == torch.int32 and kv_indices.dtype == torch.int32
    lib.orc_create_kv_indices(
        _ptr(req_to_token), _I64(req_to_token.stride(0)),
        _ptr(req_pool_indices), _I(_is64(req_pool_indices)),
        _ptr(page_kernel_lens), _I(_is64(page_kernel_lens)),
        _ptr(kv_indptr),
        _ptr(kv_start_idx), _I(_is64(kv_start_idx) if kv_start_idx is not None else 0),
        _ptr(kv_indices), _I64(req_pool_indices.numel()))
    return kv_indices


def set_kv_buffer(k_buffer, v_buffer, key, value, loc, lib=None):
    """MHATokenToKVPool.set_kv_buffer (mem_cache/memory_pool.py:369-407)."""
    lib = lib or load()
    loc = _i64(loc).contiguous()
    lib.orc_set_kv_buffer(
        _ptr(k_buffer), _ptr(v_buffer), _ptr(key), _ptr(value), _ptr(loc),
        _I64(loc.numel()), _I64(k_buffer.size(1)), _I64(k_buffer.size(2)), _I64(v_buffer.size(2)),
        _I64(k_buffer.stride(0)), _I64(k_buffer.stride(1)), _I64(v_buffer.stride(0)), _I64(v_buffer.stride(1)),
        _I64(key.stride(0)), _I64(key.stride(1)), _I64(value.stride(0)), _I64(value.stride(1)))


# ----------------------------------------------------------------------------- attention
def decode_attention(query, k_buffer, v_buffer, output, key, value, loc, attn_logits, req_to_token,
                     req_pool_indices, seq_lens, sm_scale, logit_cap, p_round: bool = False, lib=None,
                     blocked: bool = False):
    """decode_attention_cpu (sgl-kernel/csrc/cpu/decode.cpp:1375-1575): same argument
    list and in-place behaviour (writes K/V at ``loc`` into the pool, fills
    ``attn_logits`` [B,Hq,splits,Dv+1] and ``output`` [B,Hq,Dv]).  ``loc=None`` skips the
    KV write.  ``blocked=True`` runs the BLOCK_N-keys form of decode.cpp:942-985 (the one bench.py times as the CPU
    baseline); the default token-at-a-time loop is the checker."""
    lib = lib or load()
    assert query.dim() == 3 and k_buffer.dim() == 3 and v_buffer.dim() == 3
    for t in (query, k_buffer, v_buffer):
        assert t.stride(-1) == 1
    B = seq_lens.numel()
    assert attn_logits.dtype == torch.float32 and attn_logits.is_contiguous()
    assert attn_logits.shape == (B, query.size(1), attn_logits.size(2), v_buffer.size(2) + 1)
    assert output.stride(-1) == 1
    rpi = _i64(req_pool_indices).contiguous()
    sl = _i64(seq_lens).contiguous()
    locc = _i64(loc).contiguous() if loc is not None else None
    if key is None:
        key, value = query[:, :0], query[:, :0]  # unused
        nk = (0, 0, 0, 0)
    else:
        nk = (key.stride(0), key.stride(1), value.stride(0), value.stride(1))
    (lib.orc_decode_attention_blocked if blocked else lib.orc_decode_attention)(
        _ptr(query), _ptr(k_buffer), _ptr(v_buffer), _ptr(output), _ptr(key), _ptr(value), _ptr(locc),
        _ptr(attn_logits), _ptr(req_to_token), _I(_is64(req_to_token)), _ptr(rpi), _ptr(sl),
        _I64(B), _I64(req_to_token.size(1)), _I64(query.size(1)), _I64(k_buffer.size(1)),
        _I64(query.size(2)), _I64(v_buffer.size(2)), _I64(attn_logits.size(2)),
        _I64(query.stride(0)), _I64(query.stride(1)), _I64(k_buffer.stride(0)), _I64(k_buffer.stride(1)),
        _I64(v_buffer.stride(0)), _I64(v_buffer.stride(1)), _I64(nk[0]), _I64(nk[1]), _I64(nk[2]), _I64(nk[3]),
        _I64(output.stride(0)), _I64(output.stride(1)),
        _F(sm_scale), _F(logit_cap), _I(_dt(query)), _I(1 if p_round else 0))
    return output


def extend_attention(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, req_to_token, req_pool_indices,
                     seq_lens, extend_seq_lens, extend_start_loc, max_len_extend, sm_scale, logit_cap,
                     p_round: bool = True, causal: bool = True, custom_mask=None, mask_indptr=None,
                     skip_prefix_custom_mask: bool = True, sliding_window_size: int = -1, lib=None):
    """extend_attention_cpu (sgl-kernel/csrc/cpu/extend.cpp:579-723), same argument list; the keyword masks
    are those of the Triton kernel (extend_attention.py:171-189, 246-259)."""
    lib = lib or load()
    cm = mi = None
    if custom_mask is not None:
        cm = custom_mask.to(torch.uint8).contiguous()
        mi = _i64(mask_indptr).contiguous()
    del max_len_extend  # a launch-shape hint in the reference (extend.cpp:670-672); not needed here
    rpi = _i64(req_pool_indices).contiguous()
    sl = _i64(seq_lens).contiguous()
    esl = _i64(extend_seq_lens).contiguous()
    est = _i64(extend_start_loc).contiguous()
    lib.orc_extend_attention(
        _ptr(q_extend), _ptr(k_extend), _ptr(v_extend), _ptr(o_extend), _ptr(k_buffer), _ptr(v_buffer),
        _ptr(req_to_token), _I(_is64(req_to_token)), _ptr(rpi), _ptr(sl), _ptr(esl), _ptr(est),
        _I64(sl.numel()), _I64(req_to_token.size(1)), _I64(q_extend.size(1)), _I64(k_extend.size(1)),
        _I64(q_extend.size(2)), _I64(v_extend.size(2)),
        _I64(q_extend.stride(0)), _I64(q_extend.stride(1)), _I64(k_extend.stride(0)), _I64(k_extend.stride(1)),
        _I64(v_extend.stride(0)), _I64(v_extend.stride(1)), _I64(k_buffer.stride(0)), _I64(k_buffer.stride(1)),
        _I64(v_buffer.stride(0)), _I64(v_buffer.stride(1)), _I64(o_extend.stride(0)), _I64(o_extend.stride(1)),
        _F(sm_scale), _F(logit_cap), _I(_dt(q_extend)), _I(1 if p_round else 0), _I(1 if causal else 0),
        _ptr(cm), _ptr(mi), _I(1 if skip_prefix_custom_mask else 0), _I64(int(sliding_window_size)))
    return o_extend


# ----------------------------------------------------------------------------- FP8
def per_token_quant_fp8(input, output_q, output_s, lib=None):
    """sgl_per_token_quant_fp8(input, output_q, output_s) (per_token_quant_fp8.cu:166-227)."""
    lib = lib or load()
    assert input.is_contiguous() and output_q.is_contiguous() and output_s.is_contiguous()
    assert output_q.dtype in (torch.float8_e4m3fn, torch.uint8) and output_s.dtype == torch.float32
    T, K = input.shape
    lib.orc_per_token_quant_fp8(_ptr(input), _ptr(output_q), _ptr(output_s), _I64(T), _I64(K), _I(_dt(input)))


def _in_code(t):
    return {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}[t.dtype]


def per_token_group_quant_fp8(input, output_q, output_s, group_size, eps, fp8_min, fp8_max, lib=None):
    """sgl_per_token_group_quant_fp8 with row-major scales (per_token_group_quant_8bit.cu:15-215)."""
    lib = lib or load()
    assert input.is_contiguous() and output_q.is_contiguous() and output_s.is_contiguous()
    lib.orc_per_token_group_quant_fp8(_ptr(input), _ptr(output_q), _ptr(output_s), _I64(input.numel() // group_size),
                                      _I64(group_size), _F(eps), _F(fp8_min), _F(fp8_max), _I(_in_code(input)))


def per_token_group_quant_fp8_ue8m0(input, output_q, output_s, group_size, eps, fp8_min, fp8_max, lib=None):
    """sgl_per_token_group_quant_fp8(..., scale_ue8m0=True): power-of-two scales as exponent bytes packed four to an int32,
    column-major (per_token_group_quant_8bit.cu:24-137).  output_s: int32 [T, ceil(K / G / 4)] with stride(0) == 1, as
    create_per_token_group_quant_fp8_output_scale (fp8_kernel.py:308-319) makes it."""
    lib = lib or load()
    assert input.is_contiguous() and output_q.is_contiguous() and input.dim() == 2
    assert output_s.dtype == torch.int32 and output_s.dim() == 2 and (output_s.stride(0) == 1 or output_s.size(0) == 1)
    T, K = input.shape
    lib.orc_per_token_group_quant_fp8_ue8m0(_ptr(input), _ptr(output_q), _ptr(output_s), _I64(T), _I64(K), _I64(group_size),
                                            _I64(output_s.stride(1)), _F(eps), _F(fp8_min), _F(fp8_max), _I(_in_code(input)))


def per_tensor_quant_fp8(input, output_q, output_s, is_static, lib=None):
    """sgl_per_tensor_quant_fp8 (per_tensor_quant_fp8.cu:90-120)."""
    lib = lib or load()
    assert input.is_contiguous() and output_q.is_contiguous() and output_s.numel() == 1
    lib.orc_per_tensor_quant_fp8(_ptr(input), _ptr(output_q), _ptr(output_s), _I64(input.numel()),
                                 _I(1 if is_static else 0), _I(_in_code(input)))


def fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias=None, bias_after_round: bool = False, lib=None):
    """fp8_scaled_mm (fp8_gemm_kernel.cu:1071-1146).  mat_b is [K,N] with stride(0)==1."""
    lib = lib or load()
    assert mat_a.dim() == 2 and mat_b.dim() == 2 and mat_a.stride(1) == 1 and mat_b.stride(0) == 1
    M, K = mat_a.shape
    N = mat_b.size(1)
    out = torch.empty((M, N), dtype=out_dtype)
    sa = scales_a.reshape(-1).contiguous().float()
    sb = scales_b.reshape(-1).contiguous().float()
    lib.orc_fp8_scaled_mm(
        _ptr(mat_a), _ptr(mat_b), _ptr(sa), _ptr(sb), _ptr(bias), _ptr(out), _I64(M), _I64(N), _I64(K),
        _I64(mat_a.stride(0)), _I64(mat_b.stride(1)), _I(_dt(out)), _I(1 if bias_after_round else 0))
    return out


# ----------------------------------------------------------------------------- AWQ
def awq_dequantize(qweight, scales, qzeros, lib=None):
    """awq_dequantize(qweight, scales, qzeros) -> [K, N] (awq_kernel.cu:186-221)."""
    lib = lib or load()
    assert qweight.dtype == torch.int32 and qzeros.dtype == torch.int32
    qweight, scales, qzeros = qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    K, Nc = qweight.shape
    group = K // scales.size(0)
    out = torch.empty((K, Nc * 8), dtype=scales.dtype)
    lib.orc_awq_dequantize(_ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(out), _I64(K), _I64(Nc), _I64(group),
                           _I(_dt(scales)))
    return out


def awq_gemm(x, qweight, scales, qzeros, bias=None, lib=None):
    """AWQLinearMethod.apply (layers/quantization/awq.py:401-418)."""
    lib = lib or load()
    qweight, scales, qzeros = qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    K, Nc = qweight.shape
    x2 = x.reshape(-1, K).contiguous()
    group = K // scales.size(0)
    out = torch.empty((x2.size(0), Nc * 8), dtype=x.dtype)
    lib.orc_awq_gemm(_ptr(x2), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(bias), _ptr(out),
                     _I64(x2.size(0)), _I64(K), _I64(Nc), _I64(group), _I(_dt(x)))
    return out.reshape(x.shape[:-1] + (Nc * 8,))


# ----------------------------------------------------------------------------- elementwise ("next" rows)
def rmsnorm(x, weight, eps, residual=None, lib=None):
    lib = lib or load()
    x = x.contiguous()
    out = torch.empty_like(x)
    T, H = x.reshape(-1, x.size(-1)).shape
    lib.orc_rmsnorm(_ptr(x), _ptr(residual), _ptr(weight), _ptr(out), _I64(T), _I64(H), _F(eps), _I(_dt(x)))
    return out


def silu_and_mul(x, lib=None):
    lib = lib or load()
    x = x.contiguous()
    d = x.size(-1) // 2
    out = torch.empty(x.shape[:-1] + (d,), dtype=x.dtype)
    lib.orc_silu_and_mul(_ptr(x), _ptr(out), _I64(x.numel() // (2 * d)), _I64(d), _I(_dt(x)))
    return out


def rope_neox(x, positions, cos_sin_cache, rot_dim=None, lib=None):
    """In-place neox rotary on x [T, H, D]."""
    lib = lib or load()
    assert x.dim() == 3 and x.stride(2) == 1 and cos_sin_cache.dtype == torch.float32
    rot_dim = rot_dim or cos_sin_cache.size(1)
    pos = _i64(positions).contiguous()
    lib.orc_rope_neox(_ptr(x), _ptr(pos), _ptr(cos_sin_cache.contiguous()), _I64(x.size(0)), _I64(x.size(1)),
                      _I64(x.size(2)), _I64(rot_dim), _I64(x.stride(0)), _I64(x.stride(1)), _I(_dt(x)))
    return x


def merge_state(prefix_output, prefix_lse, suffix_output, suffix_lse, lib=None):
    """merge_state_triton (triton_ops/merge_state.py:68-96): returns (output, output_lse)."""
    lib = lib or load()
    code = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}[prefix_output.dtype]
    po, so = prefix_output.contiguous(), suffix_output.contiguous()
    out = torch.empty_like(po)
    out_lse = torch.empty_like(prefix_lse)
    N, H, D = po.shape
    lib.orc_merge_state(_ptr(po), _ptr(prefix_lse.contiguous()), _ptr(so), _ptr(suffix_lse.contiguous()), _ptr(out),
                        _ptr(out_lse), _I64(N * H), _I64(D), _I(code))
    return out, out_lse


# ----------------------------------------------------------------------------- FP8 (e4m3 / e5m2) KV cache
class _kv_format:
    """`with _kv_format(lib, kv_dtype):` -- the pool format the *_fp8kv C functions read for the duration of one call
    (float8_e5m2, or the default float8_e4m3fn; a uint8 view needs the explicit kv_dtype)."""

    def __init__(self, lib, kv_dtype, *pools):
        if kv_dtype is None and any(p.dtype == torch.float8_e5m2 for p in pools):
            kv_dtype = torch.float8_e5m2
        self.lib, self.fmt = lib, 2 if kv_dtype == torch.float8_e5m2 else 1

    def __enter__(self):
        self.lib.orc_set_kv_format(_I(self.fmt))

    def __exit__(self, *exc):
        self.lib.orc_set_kv_format(_I(1))


def cvt_f32_to_e5m2(x: torch.Tensor, lib=None) -> torch.Tensor:
    """torch's fp32 -> float8_e5m2 cast as the oracle restates it (uint8 bytes)."""
    lib = lib or load()
    x = x.float().contiguous()
    y = torch.empty(x.shape, dtype=torch.uint8)
    lib.orc_cvt_f32_to_e5m2(_ptr(x), _ptr(y), _I64(x.numel()))
    return y


def set_kv_buffer_fp8(k_buffer, v_buffer, key, value, loc, k_scale=None, v_scale=None, kv_dtype=None, lib=None):
    """MHATokenToKVPool.set_kv_buffer with dtype float8_e4m3fn / float8_e5m2 (memory_pool.py:369-407): pools are
    uint8 / fp8 [N,Hkv,D], key/value the 16-bit new entries [T,Hkv,D]."""
    lib = lib or load()
    T, Hkv, D = key.shape
    with _kv_format(lib, kv_dtype, k_buffer, v_buffer):
        _set_kv_buffer_fp8(lib, k_buffer, v_buffer, key, value, loc, k_scale, v_scale, T, Hkv, D)


def _set_kv_buffer_fp8(lib, k_buffer, v_buffer, key, value, loc, k_scale, v_scale, T, Hkv, D):
    lib.orc_set_kv_buffer_fp8(
        _ptr(k_buffer), _ptr(v_buffer), _ptr(key), _ptr(value), _ptr(_i64(loc).contiguous()), _I64(T), _I64(Hkv), _I64(D),
        _I64(value.size(2)), _I64(k_buffer.stride(0)), _I64(k_buffer.stride(1)), _I64(v_buffer.stride(0)),
        _I64(v_buffer.stride(1)), _I64(key.stride(0)), _I64(key.stride(1)), _I64(value.stride(0)), _I64(value.stride(1)),
        _F(k_scale if k_scale else 0.0), _F(v_scale if v_scale else 0.0), _I(_dt(key)))


def decode_attention_fp8kv(query, k_buffer, v_buffer, output, attn_logits, req_to_token, req_pool_indices, seq_lens,
                           sm_scale, logit_cap=0.0, p_fp8: bool = True, kv_dtype=None, lib=None):
    """Triton decode over an FP8 KV pool (decode_attention.py:240-488): K upcast, P rounded to the pool format per
    32-token block.  attn_logits [B,Hq,splits,Dv+1] fp32 scratch; p_fp8=False keeps P in fp32 (the truth for noise
    measurements)."""
    lib = lib or load()
    with _kv_format(lib, kv_dtype, k_buffer, v_buffer):
        return _decode_attention_fp8kv(lib, query, k_buffer, v_buffer, output, attn_logits, req_to_token, req_pool_indices,
                                       seq_lens, sm_scale, logit_cap, p_fp8)


def _decode_attention_fp8kv(lib, query, k_buffer, v_buffer, output, attn_logits, req_to_token, req_pool_indices, seq_lens,
                            sm_scale, logit_cap, p_fp8):
    B = seq_lens.numel()
    lib.orc_decode_attention_fp8kv(
        _ptr(query), _ptr(k_buffer), _ptr(v_buffer), _ptr(output), _ptr(attn_logits), _ptr(req_to_token),
        _I(_is64(req_to_token)), _ptr(_i64(req_pool_indices).contiguous()), _ptr(_i64(seq_lens).contiguous()), _I64(B),
        _I64(req_to_token.size(1)), _I64(query.size(1)), _I64(k_buffer.size(1)), _I64(query.size(2)),
        _I64(v_buffer.size(2)), _I64(attn_logits.size(2)), _I64(query.stride(0)), _I64(query.stride(1)),
        _I64(k_buffer.stride(0)), _I64(k_buffer.stride(1)), _I64(v_buffer.stride(0)), _I64(v_buffer.stride(1)),
        _I64(output.stride(0)), _I64(output.stride(1)), _F(sm_scale), _F(logit_cap), _I(_dt(query)), _I(1 if p_fp8 else 0))
    return output


def extend_attention_fp8kv(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, req_to_token, req_pool_indices,
                           seq_lens, extend_seq_lens, extend_start_loc, sm_scale, logit_cap=0.0, p_round: bool = True,
                           causal: bool = True, custom_mask=None, mask_indptr=None, skip_prefix_custom_mask: bool = True,
                           sliding_window_size: int = -1, q_fp8: bool = True, p_fp8: bool = True, kv_dtype=None,
                           lib=None):
    """Extend attention over an FP8 pool (uint8 / float8_e4m3fn / float8_e5m2 k_buffer, v_buffer), Triton-kernel
    semantics (extend_attention.py:124-303: Q and P rounded to the pool format in the prefix stage, blocks of 64 keys)."""
    lib = lib or load()
    with _kv_format(lib, kv_dtype, k_buffer, v_buffer):
        return _extend_attention_fp8kv(lib, q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, req_to_token,
                                       req_pool_indices, seq_lens, extend_seq_lens, extend_start_loc, sm_scale, logit_cap,
                                       p_round, causal, custom_mask, mask_indptr, skip_prefix_custom_mask,
                                       sliding_window_size, q_fp8, p_fp8)


def _extend_attention_fp8kv(lib, q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, req_to_token, req_pool_indices,
                            seq_lens, extend_seq_lens, extend_start_loc, sm_scale, logit_cap, p_round, causal, custom_mask,
                            mask_indptr, skip_prefix_custom_mask, sliding_window_size, q_fp8, p_fp8):
    assert k_buffer.element_size() == 1 and v_buffer.element_size() == 1
    cm = mi = None
    if custom_mask is not None:
        cm = custom_mask.to(torch.uint8).contiguous()
        mi = _i64(mask_indptr).contiguous()
    rpi = _i64(req_pool_indices).contiguous()
    sl = _i64(seq_lens).contiguous()
    esl = _i64(extend_seq_lens).contiguous()
    est = _i64(extend_start_loc).contiguous()
    lib.orc_extend_attention_fp8kv(
        _ptr(q_extend), _ptr(k_extend), _ptr(v_extend), _ptr(o_extend), _ptr(k_buffer), _ptr(v_buffer),
        _ptr(req_to_token), _I(_is64(req_to_token)), _ptr(rpi), _ptr(sl), _ptr(esl), _ptr(est),
        _I64(sl.numel()), _I64(req_to_token.size(1)), _I64(q_extend.size(1)), _I64(k_extend.size(1)),
        _I64(q_extend.size(2)), _I64(v_extend.size(2)),
        _I64(q_extend.stride(0)), _I64(q_extend.stride(1)), _I64(k_extend.stride(0)), _I64(k_extend.stride(1)),
        _I64(v_extend.stride(0)), _I64(v_extend.stride(1)), _I64(k_buffer.stride(0)), _I64(k_buffer.stride(1)),
        _I64(v_buffer.stride(0)), _I64(v_buffer.stride(1)), _I64(o_extend.stride(0)), _I64(o_extend.stride(1)),
        _F(sm_scale), _F(logit_cap), _I(_dt(q_extend)), _I(1 if p_round else 0), _I(1 if causal else 0),
        _ptr(cm), _ptr(mi), _I(1 if skip_prefix_custom_mask else 0), _I64(int(sliding_window_size)),
        _I(1 if q_fp8 else 0), _I(1 if p_fp8 else 0))
    return o_extend


def vocab_parallel_embedding(ids, table, org_vocab_start, org_vocab_end, num_org_vocab_padding=0):
    """One rank's VocabParallelEmbedding.forward before the all-reduce (vocab_parallel_embedding.py:462-482), the
    original vocabulary only: get_masked_input_and_mask (:126-150) -> F.embedding on the rank's shard ->
    masked_fill_(~vocab_mask, 0).  Plain torch on CPU tensors (index arithmetic and a byte copy: exact)."""
    org_vocab_mask = (ids >= org_vocab_start) & (ids < org_vocab_end)
    valid_offset = org_vocab_start * org_vocab_mask
    masked = org_vocab_mask * (ids - valid_offset)
    out = torch.nn.functional.embedding(masked.long(), table)
    out.masked_fill_((~org_vocab_mask).unsqueeze(-1), 0)
    return out
