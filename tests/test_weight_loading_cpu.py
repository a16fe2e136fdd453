"""CPU: the quant-linear methods' parameters can be filled by SGLang's checkpoint loaders.

(A) always: this repo's harness layers drive the local parameter classes with full ("checkpoint") tensors and shard
    ids, TP = 1 / 2 / 4 -- every rank must end up with exactly its slice.
(B) build container only (needs /root/reference): the REFERENCE's own ``layers/parameter.py`` and ``layers/linear.py``
    are loaded by file path (their sglang-internal imports satisfied by small placeholder modules; nothing of the
    reference is stored here), the reference's QKVParallelLinear / MergedColumnParallelLinear / RowParallelLinear are
    constructed with THIS repo's quant configs, and the reference's ``weight_loader`` / ``weight_loader_v2`` fill the
    parameters our ``create_weights`` made.  That is the path models/llama.py:600-625 takes.
"""
import importlib.util
import os
import sys
import types

import pytest
import torch

from sglang_npu_amd import distributed as D
from sglang_npu_amd import parameter as P
from sglang_npu_amd.quantization import AWQConfig, W8A8Fp8Config

REF = "/root/reference/python/sglang/srt"


def _set_tp(rank, world):
    D.set_tp_group(D.GroupCoordinator(None, rank, world, None))


@pytest.fixture(autouse=True)
def _restore_tp():
    yield
    _set_tp(0, 1)
    P.classes(refresh=True)


def _expected_qkv(full, sizes_total, head, rank, tp, total_kv):
    """Rows of the fused [q | k | v] checkpoint matrix that belong to `rank`."""
    q, k, v = full.split(sizes_total, 0)
    nq = q.size(0) // tp
    rep = max(1, tp // total_kv)
    nkv = max(1, total_kv // tp) * head
    kv_block = rank // rep
    return torch.cat([q[rank * nq:(rank + 1) * nq], k[kv_block * nkv:(kv_block + 1) * nkv],
                      v[kv_block * nkv:(kv_block + 1) * nkv]], 0)


# ----------------------------------------------------------------------------- (A) harness layers, local classes
@pytest.mark.parametrize("tp", [1, 2, 4])
def test_harness_layers_load_fp8_checkpoint_shards(tp):
    from sglang_npu_amd.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
    g = torch.Generator().manual_seed(tp)
    hidden, head, Hq, Hkv, inter = 64, 16, 8, 2, 96
    cfg = W8A8Fp8Config(is_checkpoint_fp8_serialized=True)
    wq = torch.randn(Hq * head, hidden, generator=g)
    wk, wv = torch.randn(Hkv * head, hidden, generator=g), torch.randn(Hkv * head, hidden, generator=g)
    sq, sk, sv = torch.rand(Hq * head, 1, generator=g), torch.rand(Hkv * head, 1, generator=g), torch.rand(Hkv * head, 1, generator=g)
    wg, wu = torch.randn(inter, hidden, generator=g), torch.randn(inter, hidden, generator=g)
    wd, sd = torch.randn(hidden, inter, generator=g), torch.rand(hidden, 1, generator=g)
    f8 = lambda t: t.to(torch.float8_e4m3fn)  # noqa: E731
    for rank in range(tp):
        _set_tp(rank, tp)
        qkv = QKVParallelLinear(hidden, head, Hq, Hkv, params_dtype=torch.bfloat16, quant_config=cfg)
        assert type(qkv.weight).__name__ == "ModelWeightParameter" and type(qkv.weight_scale).__name__ == "ChannelQuantScaleParameter"
        for sid, w, s in (("q", wq, sq), ("k", wk, sk), ("v", wv, sv)):
            qkv.weight.weight_loader(qkv.weight, f8(w), sid)
            qkv.weight_scale.weight_loader(qkv.weight_scale, s, sid)
        exp_w = _expected_qkv(torch.cat([wq, wk, wv]), [Hq * head, Hkv * head, Hkv * head], head, rank, tp, Hkv)
        exp_s = _expected_qkv(torch.cat([sq, sk, sv]), [Hq * head, Hkv * head, Hkv * head], head, rank, tp, Hkv)
        assert torch.equal(qkv.weight.data.float(), f8(exp_w).float()) and torch.equal(qkv.weight_scale.data, exp_s)
        # fused-on-disk form (no shard id) gives the same parameter
        qkv2 = QKVParallelLinear(hidden, head, Hq, Hkv, params_dtype=torch.bfloat16, quant_config=cfg)
        qkv2.weight.weight_loader(qkv2.weight, f8(torch.cat([wq, wk, wv])))
        assert torch.equal(qkv2.weight.data.float(), qkv.weight.data.float())

        gate_up = MergedColumnParallelLinear(hidden, [inter, inter], params_dtype=torch.bfloat16, quant_config=cfg)
        gate_up.weight.weight_loader(gate_up.weight, f8(wg), 0)
        gate_up.weight.weight_loader(gate_up.weight, f8(wu), 1)
        n = inter // tp
        exp = torch.cat([wg[rank * n:(rank + 1) * n], wu[rank * n:(rank + 1) * n]])
        assert torch.equal(gate_up.weight.data.float(), f8(exp).float())

        down = RowParallelLinear(inter, hidden, params_dtype=torch.bfloat16, quant_config=cfg)
        down.weight.weight_loader(down.weight, f8(wd))
        down.weight_scale.weight_loader(down.weight_scale, sd)
        assert torch.equal(down.weight.data.float(), f8(wd[:, rank * n:(rank + 1) * n]).float())
        assert torch.equal(down.weight_scale.data, sd), "per-output-channel scales are not split by a row-parallel layer"
        down.quant_method.process_weights_after_loading(down)
        assert down.weight.shape == (n, hidden) and down.weight.stride(0) == 1   # K-major view


@pytest.mark.parametrize("tp", [1, 2])
def test_harness_layers_load_awq_checkpoint_shards(tp):
    from sglang_npu_amd.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
    g = torch.Generator().manual_seed(7)
    K, head, Hq, Hkv, inter, G = 256, 16, 8, 2, 128, 128
    cfg = AWQConfig(4, G, True)
    ri = lambda *s: torch.randint(0, 2 ** 31 - 1, s, dtype=torch.int32, generator=g)  # noqa: E731
    # checkpoint tensors: qweight [K, N/8] int32, qzeros [K/G, N/8], scales [K/G, N]
    ck = {sid: (ri(K, n // 8), ri(K // G, n // 8), torch.rand(K // G, n, generator=g).half())
          for sid, n in (("q", Hq * head), ("k", Hkv * head), ("v", Hkv * head))}
    gu = [(ri(K, inter // 8), ri(K // G, inter // 8), torch.rand(K // G, inter, generator=g).half()) for _ in range(2)]
    dn = (ri(inter, K // 8), ri(inter // G * tp // tp, K // 8), torch.rand(inter // G, K, generator=g).half())
    for rank in range(tp):
        _set_tp(rank, tp)
        qkv = QKVParallelLinear(K, head, Hq, Hkv, params_dtype=torch.float16, quant_config=cfg)
        assert type(qkv.qweight).__name__ == "PackedvLLMParameter" and qkv.qweight.packed_factor == 8
        assert type(qkv.scales).__name__ == "GroupQuantScaleParameter"
        for sid, (qw, qz, sc) in ck.items():
            qkv.qweight.weight_loader(qkv.qweight, qw, sid)
            qkv.qzeros.weight_loader(qkv.qzeros, qz, sid)
            qkv.scales.weight_loader(qkv.scales, sc, sid)
        sizes = [Hq * head, Hkv * head, Hkv * head]
        for name, idx, div in (("qweight", 0, 8), ("qzeros", 1, 8), ("scales", 2, 1)):
            full = torch.cat([ck[s][idx] for s in "qkv"], 1).t()  # rows = output columns (packed by `div`)
            exp = _expected_qkv(full, [s // div for s in sizes], head // div, rank, tp, Hkv).t()
            assert torch.equal(getattr(qkv, name).data, exp), name
        gate_up = MergedColumnParallelLinear(K, [inter, inter], params_dtype=torch.float16, quant_config=cfg)
        for i, (qw, qz, sc) in enumerate(gu):
            gate_up.qweight.weight_loader(gate_up.qweight, qw, i)
            gate_up.scales.weight_loader(gate_up.scales, sc, i)
        n = inter // tp
        assert torch.equal(gate_up.qweight.data, torch.cat([gu[0][0][:, rank * n // 8:(rank + 1) * n // 8],
                                                            gu[1][0][:, rank * n // 8:(rank + 1) * n // 8]], 1))
        assert torch.equal(gate_up.scales.data, torch.cat([gu[0][2][:, rank * n:(rank + 1) * n],
                                                           gu[1][2][:, rank * n:(rank + 1) * n]], 1))
        if inter // tp % G == 0:
            down = RowParallelLinear(inter, K, params_dtype=torch.float16, quant_config=cfg)
            down.qweight.weight_loader(down.qweight, dn[0])
            down.scales.weight_loader(down.scales, dn[2])
            assert torch.equal(down.qweight.data, dn[0][rank * n:(rank + 1) * n])       # K (rows) is the sharded dim
            assert torch.equal(down.scales.data, dn[2][rank * n // G:(rank + 1) * n // G])


def test_get_quant_method_gates_on_linear_layers_only():
    from sglang_npu_amd.linear import RowParallelLinear

    class LinearBase(torch.nn.Module):          # stands for sglang.srt.layers.linear.LinearBase (another package)
        pass

    class ForeignRowParallel(LinearBase):
        pass

    assert type(W8A8Fp8Config(True).get_quant_method(ForeignRowParallel(), "")).__name__ == "W8A8Fp8LinearMethod"
    assert type(AWQConfig(4, 128, True).get_quant_method(ForeignRowParallel(), "")).__name__ == "AWQLinearMethod"
    assert W8A8Fp8Config(True).get_quant_method(torch.nn.Linear(4, 4), "") is None   # not a LinearBase: w8a8_fp8.py:92
    assert RowParallelLinear(128, 16, quant_config=AWQConfig(4, 128, True), params_dtype=torch.float16).quant_method is not None


# ----------------------------------------------------------------------------- (B) the reference's own layers + loaders
def _load_reference_linear(tp_rank_default=0, tp_size_default=1):
    """Load the reference's parameter.py and linear.py by path.  Placeholders cover only what those two files import
    from the rest of sglang at module import time (distributed helpers, set_weight_attrs, is_cpu / is_npu, the
    unquantised method)."""
    made = []

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []  # behaves as a package for sub-imports
        sys.modules[name] = m
        made.append(name)
        return m

    def set_weight_attrs(weight, attrs):
        for k, v in (attrs or {}).items():
            setattr(weight, k, v)

    class UnquantizedLinearMethod:  # never used below (a quant_config is always given)
        pass

    mod("sglang"), mod("sglang.srt"), mod("sglang.srt.layers"), mod("sglang.srt.layers.quantization")
    mod("sglang.srt.utils", is_cpu=lambda: False, is_npu=lambda: False, set_weight_attrs=set_weight_attrs)
    mod("sglang.srt.distributed", divide=lambda a, b: a // b, get_tensor_model_parallel_rank=lambda: tp_rank_default,
        get_tensor_model_parallel_world_size=lambda: tp_size_default, parallel_state=types.SimpleNamespace(),
        split_tensor_along_last_dim=None, tensor_model_parallel_all_gather=None, tensor_model_parallel_all_reduce=None)
    mod("sglang.srt.distributed.device_communicators")
    mod("sglang.srt.distributed.device_communicators.pynccl_allocator", use_symmetric_memory=None)
    mod("sglang.srt.layers.quantization.unquant", UnquantizedLinearMethod=UnquantizedLinearMethod)
    # imported inside the load_* methods but only called on the CPU backend (parameter.py:104-117)
    mod("sglang.srt.model_loader")
    mod("sglang.srt.model_loader.weight_utils", narrow_padded_param_and_loaded_weight=None)

    def by_path(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        made.append(name)
        spec.loader.exec_module(m)
        return m

    param_mod = by_path("sglang.srt.layers.parameter", f"{REF}/layers/parameter.py")
    linear_mod = by_path("sglang.srt.layers.linear", f"{REF}/layers/linear.py")
    return param_mod, linear_mod, made


@pytest.mark.skipif(not os.path.exists(f"{REF}/layers/linear.py"), reason="needs /root/reference (build container)")
@pytest.mark.parametrize("tp", [1, 2, 8])
def test_reference_layers_and_loaders_fill_our_parameters(tp):
    param_mod, ref_linear, made = _load_reference_linear()
    try:
        cls = P.classes(refresh=True)
        assert cls.source == "sglang" and cls.ModelWeightParameter is param_mod.ModelWeightParameter
        g = torch.Generator().manual_seed(tp)
        hidden, head, Hq, Hkv, inter, G = 256, 16, 16, 4, 1024, 128
        f8 = lambda t: t.to(torch.float8_e4m3fn)  # noqa: E731
        ri = lambda *s: torch.randint(0, 2 ** 31 - 1, s, dtype=torch.int32, generator=g)  # noqa: E731
        fp8 = W8A8Fp8Config(is_checkpoint_fp8_serialized=True)
        awq = AWQConfig(4, G, True)
        w = {"q": torch.randn(Hq * head, hidden, generator=g), "k": torch.randn(Hkv * head, hidden, generator=g),
             "v": torch.randn(Hkv * head, hidden, generator=g)}
        sc = {k: torch.rand(v.size(0), 1, generator=g) for k, v in w.items()}
        aw = {k: (ri(hidden, v.size(0) // 8), ri(hidden // G, v.size(0) // 8), torch.rand(hidden // G, v.size(0), generator=g).half())
              for k, v in w.items()}
        wg, wu, wd = (torch.randn(inter, hidden, generator=g), torch.randn(inter, hidden, generator=g),
                      torch.randn(hidden, inter, generator=g))
        sizes = [Hq * head, Hkv * head, Hkv * head]
        for rank in range(tp):
            # ---- FP8 w8a8: the reference picks its legacy `weight_loader` for this method name (linear.py:44-61,315)
            qkv = ref_linear.QKVParallelLinear(hidden, head, Hq, Hkv, bias=False, params_dtype=torch.bfloat16,
                                               quant_config=fp8, tp_rank=rank, tp_size=tp)
            assert type(qkv.quant_method).__name__ == "W8A8Fp8LinearMethod"
            assert isinstance(qkv.weight, param_mod.ModelWeightParameter)
            for sid in "qkv":
                qkv.weight.weight_loader(qkv.weight, f8(w[sid]), sid)          # what models/llama.py:622 does
                qkv.weight_scale.weight_loader(qkv.weight_scale, sc[sid], sid)
            exp_w = _expected_qkv(torch.cat([w[s] for s in "qkv"]), sizes, head, rank, tp, Hkv)
            exp_s = _expected_qkv(torch.cat([sc[s] for s in "qkv"]), sizes, head, rank, tp, Hkv)
            assert torch.equal(qkv.weight.data.float(), f8(exp_w).float()) and torch.equal(qkv.weight_scale.data, exp_s)
            qkv.quant_method.process_weights_after_loading(qkv)
            assert qkv.weight.shape == (hidden, exp_w.size(0)) and qkv.weight.stride(0) == 1

            gate_up = ref_linear.MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16,
                                                            quant_config=fp8, tp_rank=rank, tp_size=tp)
            gate_up.weight.weight_loader(gate_up.weight, f8(wg), 0)
            gate_up.weight.weight_loader(gate_up.weight, f8(wu), 1)
            n = inter // tp
            assert torch.equal(gate_up.weight.data.float(),
                               f8(torch.cat([wg[rank * n:(rank + 1) * n], wu[rank * n:(rank + 1) * n]])).float())
            down = ref_linear.RowParallelLinear(inter, hidden, bias=False, params_dtype=torch.bfloat16, quant_config=fp8,
                                                tp_rank=rank, tp_size=tp)
            down.weight.weight_loader(down.weight, f8(wd))
            assert torch.equal(down.weight.data.float(), f8(wd[:, rank * n:(rank + 1) * n]).float())

            # ---- AWQ: `AWQLinearMethod` is in WEIGHT_LOADER_V2_SUPPORTED -> the v2 loaders and isinstance dispatch
            qkv = ref_linear.QKVParallelLinear(hidden, head, Hq, Hkv, bias=False, params_dtype=torch.float16,
                                               quant_config=awq, tp_rank=rank, tp_size=tp)
            assert type(qkv.quant_method).__name__ == "AWQLinearMethod"
            assert isinstance(qkv.qweight, param_mod.PackedvLLMParameter)
            assert isinstance(qkv.scales, param_mod.GroupQuantScaleParameter)
            assert qkv.qweight.weight_loader == qkv.weight_loader_v2
            for sid in "qkv":
                for name, t in zip(("qweight", "qzeros", "scales"), aw[sid]):
                    p = getattr(qkv, name)
                    p.weight_loader(p, t, sid)
            for name, idx, div in (("qweight", 0, 8), ("qzeros", 1, 8), ("scales", 2, 1)):
                full = torch.cat([aw[s][idx] for s in "qkv"], 1).t()
                exp = _expected_qkv(full, [s // div for s in sizes], head // div, rank, tp, Hkv).t()
                assert torch.equal(getattr(qkv, name).data, exp), (name, rank, tp)
            # fused-on-disk qkv (no shard id) through the reference's _load_fused_module_from_checkpoint
            qkv2 = ref_linear.QKVParallelLinear(hidden, head, Hq, Hkv, bias=False, params_dtype=torch.float16,
                                                quant_config=awq, tp_rank=rank, tp_size=tp)
            qkv2.qweight.weight_loader(qkv2.qweight, torch.cat([aw[s][0] for s in "qkv"], 1))
            assert torch.equal(qkv2.qweight.data, qkv.qweight.data)
            if (inter // tp) % G == 0:
                down = ref_linear.RowParallelLinear(inter, hidden, bias=False, params_dtype=torch.float16, quant_config=awq,
                                                    tp_rank=rank, tp_size=tp)
                qw, scs = ri(inter, hidden // 8), torch.rand(inter // G, hidden, generator=g).half()
                down.qweight.weight_loader(down.qweight, qw)
                down.scales.weight_loader(down.scales, scs)
                assert torch.equal(down.qweight.data, qw[rank * n:(rank + 1) * n])
                assert torch.equal(down.scales.data, scs[rank * n // G:(rank + 1) * n // G])
    finally:
        for name in made:
            sys.modules.pop(name, None)
        P.classes(refresh=True)
