"""CPU: the QuickReduce oracle (oracle/quick_reduce.py) against the reference's own acceptance bound, the codec's
invariants, and the host logic of QuickAllReduce (thresholds and switches of quick_all_reduce.py:56-260)."""
import numpy as np
import pytest
import torch

from oracle import quick_reduce as qr


def _bf16_round(x):
    return torch.from_numpy(x).to(torch.bfloat16).float().numpy()


def _f16_round(x):
    return x.astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("regime", [1, 2, 3])
def test_oracle_meets_the_reference_tests_bound(world, regime):
    """test/srt/test_quick_allreduce.py:131-165: integer payloads in [1, 23], atol 1.25 W, rtol 0.5 W vs the exact sum."""
    rng = np.random.default_rng(world * 10 + regime)
    n = 32 * 1024 + 64
    parts = [rng.integers(1, 24, n).astype(np.float32) for _ in range(world)]
    exact = sum(parts)
    out = qr.quick_all_reduce(parts, regime, _f16_round, max_bytes=64 * 1024)  # small staging area: several chunks
    assert np.all(np.abs(out - exact) <= 1.25 * world + 0.5 * world * np.abs(exact))
    # and much tighter than that for INT8: two quantisation steps of absmax / 128 each
    if regime == 1:
        assert np.abs(out - exact).max() <= 2 * (23 * world) / 128 + 0.51 * world * 23 / 128 + 0.5


def test_codec_invariants():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(32 * 50) * 3).astype(np.float32)
    for bits in (8, 6, 4):
        R = 1 << (bits - 1)
        q, dec = qr._encode(x, bits)
        assert q.min() >= 0 and q.max() <= 2 * R - 1 and dec.dtype == np.float16 and np.all(dec.astype(np.float32) <= 0)
        y = qr._decode(q, dec, bits)
        step = np.repeat(np.abs(dec.astype(np.float32)), 32)
        assert np.all(np.abs(y - x) <= 0.5 * step * 1.01 + step * (np.abs(x) >= step * (R - 1)))  # half a step; the clamped +max a full one
        # +absmax of a block is represented exactly up to the half rounding of the scale; zeros stay zeros
        blk = np.zeros(32, dtype=np.float32)
        q0, d0 = qr._encode(blk, bits)
        assert np.all(qr._decode(q0, d0, bits) == 0)
    # a block whose values are all equal and negative: -absmax maps to +R, clamped to R - 1
    q, dec = qr._encode(np.full(32, -2.0, dtype=np.float32), 4)
    assert np.all(q == 15) and np.allclose(qr._decode(q, dec, 4), -2.0 * 7 / 8)


def test_chunking_does_not_change_whole_block_results():
    """Slices are cut at 32-value block boundaries, so a different staging size only moves which rank owns a block."""
    rng = np.random.default_rng(3)
    parts = [(rng.standard_normal(32 * 300) * 2).astype(np.float32) for _ in range(4)]
    parts = [_bf16_round(p) for p in parts]
    a = qr.quick_all_reduce(parts, 1, _bf16_round, max_bytes=16 * 1024 * 1024)
    b = qr.quick_all_reduce(parts, 1, _bf16_round, max_bytes=32 * 1024)
    assert np.array_equal(a, b)


def test_quick_all_reduce_host_logic(monkeypatch):
    from sglang_npu_amd.distributed import QuickAllReduce, QuickReduceRegime

    class FakeCA:
        _comm, world_size, rank, disabled = object(), 8, 0, False

    monkeypatch.delenv("ROCM_QUICK_REDUCE_QUANTIZATION", raising=False)
    assert QuickAllReduce(FakeCA()).disabled                      # default NONE: off (quick_all_reduce.py:190-197)
    assert QuickAllReduce(None, "INT8").disabled
    monkeypatch.setenv("ROCM_QUICK_REDUCE_QUANTIZATION", "INT4")
    q = QuickAllReduce(FakeCA())
    assert not q.disabled and q.qr_quant_level == QuickReduceRegime.INT4 and q.qr_max_size == 2048 << 20
    assert QuickAllReduce(FakeCA(), "bogus").disabled
    monkeypatch.setenv("ROCM_QUICK_REDUCE_MAX_SIZE_BYTES_MB", "64")
    q = QuickAllReduce(FakeCA(), "INT8")
    assert q.qr_max_size == 64 << 20

    class T:  # what should_quick_allreduce looks at
        def __init__(self, n, dtype=torch.float16, cuda=True, contig=True):
            self._n, self.dtype, self.is_cuda, self._c = n, dtype, cuda, contig

        def numel(self):
            return self._n

        def element_size(self):
            return 2

        def is_contiguous(self):
            return self._c

    mb = 1 << 19  # elements per MiB of 16-bit data
    assert q.should_quick_allreduce(T(4 * mb)) and not q.should_quick_allreduce(T(4 * mb - 32))       # INT8, ws 8: >= 4 MiB
    assert not q.should_quick_allreduce(T(65 * mb))                                                  # above the 64 MiB cap
    assert not q.should_quick_allreduce(T(4 * mb + 8))                                               # not whole blocks
    assert not q.should_quick_allreduce(T(4 * mb, torch.float32)) and not q.should_quick_allreduce(T(4 * mb, cuda=False))
    # bf16 takes the fp16 row while ROCM_QUICK_REDUCE_CAST_BF16_TO_FP16 is on (the default), its own row otherwise
    assert q.should_quick_allreduce(T(4 * mb, torch.bfloat16))
    monkeypatch.setenv("ROCM_QUICK_REDUCE_CAST_BF16_TO_FP16", "0")
    q2 = QuickAllReduce(FakeCA(), "INT8")
    assert not q2.should_quick_allreduce(T(4 * mb, torch.bfloat16))  # bf16 ws 8 INT8: 2048 MiB


def test_group_coordinator_dispatch_order():
    """parallel_state.py:519-542: QuickReduce first, then the custom all-reduce, then the process group."""
    from sglang_npu_amd.distributed import GroupCoordinator
    calls = []

    class QR:
        disabled = False

        def __init__(self, accept):
            self.accept = accept

        def should_quick_allreduce(self, x):
            return self.accept

        def quick_all_reduce(self, x):
            calls.append("qr")
            return x + 1

    class CA:
        disabled = False

        def __init__(self, accept):
            self.accept = accept

        def custom_all_reduce(self, x):
            calls.append("ca")
            return x + 2 if self.accept else None

    tp = GroupCoordinator(None, 0, 2, torch.device("cpu"))
    x = torch.zeros(4)
    tp.qr_comm, tp.ca_comm = QR(True), CA(True)
    assert float(tp.all_reduce(x)[0]) == 1 and calls == ["qr"]
    calls.clear()
    tp.qr_comm = QR(False)
    assert float(tp.all_reduce(x)[0]) == 2 and calls == ["ca"]
    calls.clear()
    tp.qr_comm.disabled = True
    tp.qr_comm.accept = True
    assert float(tp.all_reduce(x)[0]) == 2 and calls == ["ca"]
