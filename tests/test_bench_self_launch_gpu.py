"""GPU: `python bench.py --gpus 2 ...` -- the plain command the driver runs -- launches its own ranks.

Rehearsal on the one GPU of the box: SGL_MI355_SHARE_GPU=1 puts both ranks on device 0 and lets gloo carry the process
group (RCCL refuses two ranks on one device); the P2P all-reduce kernel over IPC buffers is the data plane, as it would be
over xGMI.  A 2-layer stack keeps it short.  What is checked is the contract of the line: one JSON line, n_gpus = 2, the
world size the ranks actually saw, the all-reduce report and the launcher's own record."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_plain_command_at_two_ranks_prints_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SGL_MI355_SHARE_GPU="1", SGL_MI355_BENCH_LAUNCH_TIMEOUT="800")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "4",
                        "--warmup", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and d["steps"] == 4 and d["warmup"] == 2
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["unit"] == "tokens/s" and d["scaling"] == "strong"
    assert d["config"]["parallelism"] == "tp2" and d["config"]["layers"] == 2
    assert "incomplete" not in d["launcher"] and d["launcher"]["ranks"] == 2
    ar = d["allreduce"]
    for key in ("backend", "overhead_frac", "latency_vs_size", "dispatch_per_step", "step_under_graph_replay",
                "ms_per_step_allreduce_stubbed"):
        assert key in ar, key
    # every collective of the step went through a data plane that exists (P2P kernel, fused with the norm, or the group)
    assert sum(ar["dispatch_per_step"].values()) >= 2 * 2 + 1
    assert all(p.get("p2p_exact", True) and p.get("rccl_exact", True) for p in ar["latency_vs_size"]["points"])
    assert d["roofline"]["frac"] > 0 and d["value_dropin"] and d["value_fused"]
