"""CPU: bench.py's self-launch machinery without a GPU -- the rank-spawning helper with stand-in
commands: every rank gets its RANK / WORLD_SIZE / MASTER_* environment, a failing rank ends the others (only the process
groups the launcher created), a hung job ends at the deadline, and `--gpus N > 1` without RANK / WORLD_SIZE routes to the
launcher before anything could touch a GPU."""
import json
import os
import subprocess
import sys
import time

from conftest import ROOT

sys.path.insert(0, ROOT)


def _bench():
    import importlib
    return importlib.import_module("bench")


def test_rank_children_get_their_environment(tmp_path):
    b = _bench()
    code = ("import os, json; print(json.dumps({k: os.environ[k] for k in "
            "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    ok, why, files = b._run_rank_children([sys.executable, "-c", code], 3, env, "t", 60.0, str(tmp_path))
    assert ok and why is None
    seen = [json.loads(open(out).read()) for out, _ in files]
    assert [s["RANK"] for s in seen] == ["0", "1", "2"] and [s["LOCAL_RANK"] for s in seen] == ["0", "1", "2"]
    assert all(s["WORLD_SIZE"] == "3" and s["MASTER_ADDR"] == "127.0.0.1" for s in seen)
    assert len({s["MASTER_PORT"] for s in seen}) == 1


def test_a_failing_rank_ends_the_others(tmp_path):
    b = _bench()
    code = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(600)"
    t0 = time.time()
    ok, why, _ = b._run_rank_children([sys.executable, "-c", code], 3, dict(os.environ), "t", 300.0, str(tmp_path))
    assert not ok and "rank 1 exited with code 7" in why
    assert time.time() - t0 < 60, "the sleeping ranks must be ended, not waited for"


def test_a_hung_job_ends_at_the_deadline(tmp_path):
    b = _bench()
    t0 = time.time()
    ok, why, _ = b._run_rank_children([sys.executable, "-c", "import time; time.sleep(600)"], 2, dict(os.environ), "t", 2.0,
                                      str(tmp_path))
    assert not ok and "not finished" in why and time.time() - t0 < 60


def test_gpus_n_without_a_job_environment_becomes_the_launcher():
    """No GPU here: the ranks the launcher starts exit with bench.py's own "needs a GPU" message, the launcher reports the
    failing rank and exits non-zero -- instead of the old SystemExit that told the caller to use torch.distributed.run."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["SGL_MI355_BENCH_LAUNCH_TIMEOUT"] = "240"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-graph"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "torch.distributed.run" not in r.stderr
    assert "[bench launcher] no result: ranks: rank" in r.stderr and "needs a GPU" in r.stderr
