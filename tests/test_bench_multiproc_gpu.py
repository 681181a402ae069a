"""bench.py's N > 1 code path (process group, parameter broadcast, bucketed gradient all-reduce from the backward hooks,
max-over-ranks timing, single JSON line from rank 0), rehearsed with two ranks on the one GPU of the test box over gloo.
The RCCL transport itself is exercised only by the driver's multi-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_over_gloo():
    env = dict(os.environ, SGL_BENCH_BACKEND="gloo", SGL_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--config", "hostile", "--batch", "3"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 6 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 6 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3
    assert "roofline" not in d and "cpu_baseline" not in d   # those legs run at N = 1 only


def test_bench_self_launches_its_ranks_when_called_as_the_driver_calls_it():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent starts the ranks as children (no GPU call, no exec in the
    parent) and relays exactly one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SGL_BENCH_BACKEND="gloo", SGL_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config",
           "hostile", "--batch", "3", "--wire", "bf16"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["per_rank_images_per_sec"]) == 2
    assert d["wire"] == "bf16" and d["exposed_comm_ms"] >= 0 and d["collectives_per_step"] >= 1
    assert d["config"]["global_batch"] == 6
