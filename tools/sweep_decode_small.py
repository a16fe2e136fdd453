import json, os, sys, torch
sys.path.insert(0, "/root/repo")
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B, S = 64, 2048
for name, Hq, Hkv, D in [("8b/tp2", 16, 4, 128), ("8b/tp4", 8, 2, 128), ("8b/tp8", 4, 1, 128), ("70b/tp8", 8, 1, 128)]:
    n_tok = B * S + 1
    NL = 6
    kbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
    vbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
    q = torch.randn(B, Hq, D, device=dev, generator=g).to(torch.bfloat16)
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=dev)
    r2t = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).view(B, S).to(torch.int32).contiguous()
    rpi, seq = torch.arange(B, device=dev), torch.full((B,), S, device=dev)
    res = {}
    for splits in (1, 2, 4, 8, 16):
        logits = torch.zeros(B, Hq, splits, D + 1, device=dev) if splits > 1 else None
        def run(i):
            ops.decode_attention_paged(q, kbs[i % NL], vbs[i % NL], o, r2t, rpi, seq, logits, splits, D ** -0.5, 0.0)
        for i in range(3): run(i)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for i in range(20): run(i)
        en.record(); torch.cuda.synchronize()
        res[splits] = round(st.elapsed_time(en) / 20 * 1e3, 1)
    nbytes = B * S * Hkv * 2 * D * 2
    print(name, "waves", os.environ.get("SGL_MI355_DECODE_WAVES", "auto"), res, "ideal@5.5TB/s %.1f us" % (nbytes / 5.5e6), flush=True)
