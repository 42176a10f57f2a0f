"""`python bench.py --gpus N` must produce the JSON line by itself: with no torch.distributed environment it
starts its own N ranks (child `python -m torch.distributed.run`), relays rank 0's line and exits with the
children's code.  Rehearsed here with --dry (gloo, CPU: launch plumbing only, no search).  CPU only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=600)


def test_self_launch_two_ranks_prints_one_json_line():
    out = _run("--gpus", "2", "--steps", "4", "--warmup", "1", "--dry", "--rows-per-gpu", "10000000")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["dry"] is True
    assert d["scaling"] == "weak" and d["config"]["rows"] == 20_000_000       # config 4 form: rows per GPU x N
    # MAX over ranks: rank 1 sleeps 2 ms per step, rank 0 only 1 ms
    assert d["ms_per_step"] >= 2.0


def test_single_rank_needs_no_launcher():
    out = _run("--gpus", "1", "--steps", "2", "--dry")
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong"


def test_rank_count_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
