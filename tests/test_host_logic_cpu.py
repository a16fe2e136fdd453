"""CPU: host-side logic of the backend mirrors (no GPU compute)."""
import pytest
import torch

from sglang_npu_amd.attention_backend import MI355AttnBackend
from sglang_npu_amd.harness import ForwardMode
from sglang_npu_amd.quantization import (AWQConfig, AWQLinearMethod, W8A8Fp8Config, W8A8Fp8LinearMethod,
                                          get_quantization_config, per_channel_quant_fp8_weight)


def _backend(num_heads, num_kv_heads, max_splits=8):
    b = MI355AttnBackend.__new__(MI355AttnBackend)
    b.num_head, b.num_kv_head, b.max_kv_splits = num_heads, num_kv_heads, max_splits
    return b


def test_split_policy_fills_the_chip_without_over_splitting():
    assert _backend(32, 8).choose_num_kv_splits(64) == 1           # 512 workgroups already
    assert _backend(32, 8).choose_num_kv_splits(32) == 1           # 256 = one per CU
    # just past one round of pairs (512 items): finer items instead of a second full round
    assert _backend(32, 8).choose_num_kv_splits(65) == 2           # 520 items, length unknown (graph capture)
    assert _backend(32, 8).choose_num_kv_splits(65, max_seq_len=4096) == 4
    assert _backend(32, 8).choose_num_kv_splits(65, max_seq_len=300) == 1
    assert _backend(32, 8).choose_num_kv_splits(80) == 2 and _backend(32, 8).choose_num_kv_splits(80, max_seq_len=4096) == 2
    assert _backend(32, 8).choose_num_kv_splits(96) == 1 and _backend(32, 8).choose_num_kv_splits(96, max_seq_len=4096) == 2
    assert _backend(32, 8).choose_num_kv_splits(112, max_seq_len=4096) == 1 and _backend(32, 8).choose_num_kv_splits(128) == 1
    assert _backend(32, 32).choose_num_kv_splits(17) == 2          # MHA: 544 items
    assert _backend(32, 8).choose_num_kv_splits(129) == 2 and _backend(32, 8).choose_num_kv_splits(136) == 2   # just past two rounds
    assert _backend(32, 8).choose_num_kv_splits(144) == 1 and _backend(32, 8).choose_num_kv_splits(192) == 1
    assert _backend(32, 8).choose_num_kv_splits(40) == 1           # 320 items: inside the first round
    assert _backend(8, 1).choose_num_kv_splits(64) == 4            # 70B TP8: 64 workgroups -> 4 splits = one per CU
    assert _backend(32, 8).choose_num_kv_splits(1, max_seq_len=600) == 2    # keep >= 256 tokens per split
    assert _backend(32, 8).choose_num_kv_splits(1, max_seq_len=100000) == 8
    assert _backend(128, 1).choose_num_kv_splits(4) == 8           # 8 head blocks x 4 = 32 workgroups


def test_forward_mode_predicates():
    assert ForwardMode.DECODE.is_decode() and ForwardMode.IDLE.is_decode_or_idle()
    assert ForwardMode.EXTEND.is_extend() and ForwardMode.MIXED.is_extend() and not ForwardMode.DECODE.is_extend()


def test_quant_registry_and_weight_layouts():
    assert get_quantization_config("w8a8_fp8") is W8A8Fp8Config and get_quantization_config("awq") is AWQConfig
    with pytest.raises(ValueError):
        get_quantization_config("gptq")
    layer = torch.nn.Module()
    W8A8Fp8LinearMethod(W8A8Fp8Config(True)).create_weights(layer, 64, [32, 16], 64, 48, torch.bfloat16)
    assert layer.weight.shape == (48, 64) and layer.weight.dtype == torch.float8_e4m3fn
    assert layer.weight_scale.shape == (48, 1) and layer.weight_scale.dtype == torch.float32
    layer2 = torch.nn.Module()
    m = AWQLinearMethod(AWQConfig(4, 128, True))
    m.create_weights(layer2, 256, [64, 64], 256, 128, torch.float16)
    assert layer2.qweight.shape == (256, 16) and layer2.qzeros.shape == (2, 16) and layer2.scales.shape == (2, 128)
    with pytest.raises(ValueError, match="input size"):
        m.create_weights(torch.nn.Module(), 200, [64], 200, 64, torch.float16)
    with pytest.raises(ValueError):
        AWQConfig(8, 128, True)
    cfg = AWQConfig.from_config({"w_bit": 4, "q_group_size": 64, "zero_point": True})
    assert cfg.group_size == 64 and cfg.pack_factor == 8


def test_per_channel_weight_quant_matches_reference_formula():
    # w8a8_fp8.py:119-125: scale = rowmax / 448; q = cast(w / scale)
    g = torch.Generator().manual_seed(0)
    w = torch.randn(16, 64, generator=g).bfloat16()
    q, s = per_channel_quant_fp8_weight(w)
    assert q.dtype == torch.float8_e4m3fn and s.shape == (16, 1)
    assert torch.equal(s, w.float().abs().amax(1, keepdim=True) / 448.0)
    assert float((q.float() * s - w.float()).abs().max()) <= float(s.max()) * 16  # within one e4m3 step at the top
    layer = torch.nn.Module()
    meth = W8A8Fp8LinearMethod(W8A8Fp8Config(False))
    meth.create_weights(layer, 64, [16], 64, 16, torch.bfloat16)
    layer.weight.data = w
    meth.process_weights_after_loading(layer)
    assert layer.weight.shape == (64, 16) and layer.weight.stride(0) == 1, "stored as the K-major [K,N] view"


def test_vocab_parallel_embedding_shards_cover_the_vocabulary_once():
    """VocabParallelEmbedding (vocab_parallel_embedding.py:153-486, original vocabulary only): the shard ranges of all
    ranks tile [0, vocab) exactly, and the masked per-rank lookups (oracle restatement of :126-150, :462-482) summed over
    the ranks -- the all-reduce of :483 -- equal the plain lookup, bit for bit."""
    import torch
    import oracle
    from sglang_npu_amd.layers import pad_vocab_size, vocab_shard_range
    g = torch.Generator().manual_seed(0)
    for vocab, world in ((128256, 8), (32000, 2), (151936, 4), (1000, 8), (100, 3)):
        padded = pad_vocab_size(vocab, 64 * world)
        table = torch.randn(vocab, 16, generator=g).bfloat16()
        ids = torch.randint(0, vocab, (3, 7), generator=g)
        ids[0, 0], ids[0, 1] = 0, vocab - 1
        total = torch.zeros(3, 7, 16, dtype=torch.float32)
        covered = torch.zeros(vocab, dtype=torch.int32)
        for rank in range(world):
            start, end, per = vocab_shard_range(vocab, padded, rank, world)
            assert 0 <= start <= end <= vocab and end - start <= per and per * world == padded
            covered[start:end] += 1
            shard = torch.zeros(per, 16, dtype=torch.bfloat16)
            shard[:end - start] = table[start:end]
            total += oracle.vocab_parallel_embedding(ids, shard, start, end).float()
        assert bool((covered == 1).all())
        assert torch.equal(total.bfloat16(), table[ids])  # every id is in exactly one shard: the sum adds zeros
