"""GPU: operands at device addresses whose bit 31 is set.

Regression for round 4's GPU memory fault in the extend kernel (commit 9b73860, csrc/common.h lds_dma16_s): the wave-uniform
DMA base was rebuilt from two `__builtin_amdgcn_readfirstlane` halves, the builtin returns `int`, and the LOW half was
sign-extended over the high one -- every q / k_extend / v_extend / pool buffer whose address has bit 31 set was read from
0xffffffff_xxxxxxxx.  A fresh process's small allocations rarely land there, so the suite never saw it; here every operand
is carved out of one > 4 GiB allocation at such an address on purpose, for the extend kernel (all stages: contiguous
new-token rows by scalar-base DMA, gathered prefix rows) and, because they use the same DMA helpers, the decode kernels,
the KV write and the FP8 weight streamer.  Checked against the oracle like the ordinary tests."""
import pytest
import torch

import oracle
from conftest import assert_elem_close
from sglang_npu_amd import ops
from test_extend_gpu import _case, _p_term, _triton_meta

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class HighArena:
    """A > 4 GiB device allocation and a bump allocator over the part of it whose addresses have bit 31 set."""

    def __init__(self, gib: int = 6):
        self.buf = torch.empty(gib << 30, dtype=torch.uint8, device=DEV)
        base, size = self.buf.data_ptr(), self.buf.numel()
        # windows of addresses with bit 31 set are [k * 4 GiB + 2 GiB, (k + 1) * 4 GiB); take the largest piece of one that lies
        # inside the buffer (a 6 GiB buffer always holds at least 1 GiB of one, wherever it starts)
        best = (0, 0)
        k = base >> 32
        for w in (k - 1, k, k + 1, k + 2):
            lo, hi = max(base, (w << 32) + (1 << 31)), min(base + size, (w + 1) << 32)
            if hi - lo > best[1] - best[0]:
                best = (lo, hi)
        assert best[1] - best[0] >= 1 << 30, "no 1 GiB window with bit 31 set inside the arena"
        self.off = (best[0] - base + 4095) & ~4095
        self.end = best[1] - base

    def put(self, t: torch.Tensor) -> torch.Tensor:
        n = t.numel() * t.element_size()
        assert self.off + n <= self.end, "arena window exhausted"
        view = self.buf[self.off:self.off + n].view(t.dtype).view(t.shape)
        view.copy_(t)
        self.off = (self.off + n + 4095) & ~4095
        assert view.data_ptr() & (1 << 31), hex(view.data_ptr())
        return view


@pytest.fixture(scope="module")
def arena():
    a = HighArena()
    yield a
    del a.buf
    torch.cuda.empty_cache()


@pytest.mark.parametrize("Hq,Hkv,D,dtype", [(32, 8, 128, torch.bfloat16), (8, 1, 128, torch.bfloat16), (14, 2, 64, torch.float16)])
def test_extend_operands_with_bit31_set(arena, Hq, Hkv, D, dtype):
    B = 3
    c = _case(B, Hq, Hkv, D, 700, 300, dtype, seed=31 + Hq, pin_first_prefix=True)
    T = c["q"].size(0)
    o_ref = torch.zeros(T, Hq, D, dtype=dtype)
    oracle.extend_attention(c["q"], c["ke"], c["ve"], o_ref, c["kb"], c["vb"], c["r2t"], c["rpi"], c["seq"], c["ext"],
                            c["start"], int(c["ext"].max()), D ** -0.5, 0.0)
    d = {k: arena.put(v.to(DEV)) for k, v in c.items()}
    qo_indptr, kv_indptr, kv_indices = _triton_meta(d, B)
    qo_indptr, kv_indptr, kv_indices = arena.put(qo_indptr), arena.put(kv_indptr), arena.put(kv_indices)
    o = arena.put(torch.zeros(T, Hq, D, dtype=dtype, device=DEV))
    term = _p_term(c, dtype, D ** -0.5)
    # the Triton-form entry point (backend glue), with and without the KV-range parts, and the op form
    ops.extend_attention_fwd(d["q"], d["ke"], d["ve"], o, d["kb"], d["vb"], qo_indptr, kv_indptr, kv_indices, None,
                             True, None, int(c["ext"].max()), D ** -0.5, 0.0)
    torch.cuda.synchronize()
    assert_elem_close(o, o_ref, dtype, pair=True, what="extend_attention_fwd at high addresses", extra=term)
    o.zero_()
    scratch = ops.ExtendPartsScratch(DEV)
    ops.extend_attention_fwd(d["q"], d["ke"], d["ve"], o, d["kb"], d["vb"], qo_indptr, kv_indptr, kv_indices, None,
                             True, None, int(c["ext"].max()), D ** -0.5, 0.0, max_prefix_len=int(c["prefix"].max()),
                             parts_scratch=scratch)
    torch.cuda.synchronize()
    assert_elem_close(o, o_ref, dtype, pair=True, what="extend_attention_fwd (parts) at high addresses", extra=term)
    o.zero_()
    ops.extend_attention(d["q"], d["ke"], d["ve"], o, d["kb"], d["vb"], d["r2t"], d["rpi"], d["seq"], d["ext"], d["start"],
                         int(c["ext"].max()), D ** -0.5, 0.0)
    torch.cuda.synchronize()
    assert_elem_close(o, o_ref, dtype, pair=True, what="extend_attention (op form) at high addresses", extra=term)


@pytest.mark.parametrize("Hq,Hkv,D,S,B", [(32, 8, 128, 700, 5), (8, 1, 128, 3000, 3), (32, 32, 128, 300, 4), (32, 8, 128, 2100, 64)])
def test_decode_and_kv_write_with_bit31_set(arena, Hq, Hkv, D, S, B):
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(S + B)
    n_tok = B * S + 1
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    kb, vb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype), torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    key, val = torch.randn(B, Hkv, D, generator=g).to(dtype), torch.randn(B, Hkv, D, generator=g).to(dtype)
    r2t = (torch.randperm(n_tok - 1, generator=g) + 1).to(torch.int32).view(B, S)
    seq = torch.randint(S // 2, S + 1, (B,), generator=g)
    seq[0] = S
    rpi = torch.arange(B)
    loc = torch.stack([r2t[b, seq[b] - 1] for b in range(B)]).long()
    o_ref = torch.zeros(B, Hq, D, dtype=dtype)
    kb_ref, vb_ref = kb.clone(), vb.clone()
    oracle.decode_attention(q, kb_ref, vb_ref, o_ref, key, val, loc, torch.zeros(B, Hq, 4, D + 1), r2t, rpi, seq, D ** -0.5, 0.0,
                            p_round=True)
    put = lambda t: arena.put(t.to(DEV))  # noqa: E731
    kb_d, vb_d = put(kb), put(vb)
    for splits in (1, 4):
        kb_d.copy_(kb)
        vb_d.copy_(vb)
        o = put(torch.zeros(B, Hq, D, dtype=dtype))
        ops.decode_attention(put(q), kb_d, vb_d, o, put(key), put(val), put(loc), put(torch.zeros(B, Hq, splits, D + 1)),
                             put(r2t), put(rpi), put(seq), D ** -0.5, 0.0)
        torch.cuda.synchronize()
        assert torch.equal(kb_d.cpu().view(torch.int16), kb_ref.view(torch.int16)), "KV write at high addresses"
        assert torch.equal(vb_d.cpu().view(torch.int16), vb_ref.view(torch.int16)), "KV write at high addresses"
        assert_elem_close(o, o_ref, dtype, pair=True, what=f"decode at high addresses, {splits} kv splits")


def test_fp8_linear_with_bit31_set(arena):
    """The FP8 weight streamers (decode rows) and the tiled prefill kernel on operands in the window."""
    g = torch.Generator().manual_seed(5)
    K, N = 4096, 6144
    w = ((torch.rand(N, K, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sb = torch.rand(N, generator=g) * 1e-2
    w_d, sb_d = arena.put(w.to(DEV)), arena.put(sb.to(DEV))
    for M in (1, 64, 1024):
        x = torch.randn(M, K, generator=g).bfloat16()
        xq_ref, xs_ref = torch.empty(M, K, dtype=torch.uint8), torch.empty(M)
        oracle.per_token_quant_fp8(x, xq_ref, xs_ref)
        xq = arena.put(torch.empty(M, K, dtype=torch.float8_e4m3fn, device=DEV))
        xs = arena.put(torch.empty(M, 1, device=DEV))
        ops.sgl_per_token_quant_fp8(arena.put(x.to(DEV)), xq, xs)
        y = ops.fp8_scaled_mm(xq, w_d.t(), xs, sb_d, torch.bfloat16)
        torch.cuda.synchronize()
        assert torch.equal(xq.cpu().view(torch.uint8), xq_ref), "per-token quant must be bit-exact"
        rows = slice(0, min(M, 16))  # the scalar oracle GEMM on a bounded sample of rows
        y_ref = oracle.fp8_scaled_mm(xq_ref[rows].view(torch.float8_e4m3fn), w.t(), xs_ref[rows], sb, torch.bfloat16)
        torch.testing.assert_close(y[rows].float().cpu(), y_ref.float(), rtol=2.0 ** -7, atol=1e-3 * float(y_ref.float().abs().max()))
