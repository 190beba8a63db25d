"""bench.py as the driver runs it at N > 1: one rank per GPU under torch.distributed.run, ONE JSON line from rank 0 as the last line of stdout.
This box has one GPU, so both ranks share cuda:0 and the process group is gloo (bench.py test hooks UNET_BENCH_BACKEND / UNET_BENCH_ONE_DEVICE);
everything else -- the launcher branch, rank / world plumbing, sharded synthetic data, the data-parallel engine program with its sync points and
gradient buckets, barrier + max-over-ranks timing, the final line -- is the code the 8-GPU run executes."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _env():
    env = dict(os.environ)
    env.update({"UNET_BENCH_BACKEND": "gloo", "UNET_BENCH_ONE_DEVICE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _check(out, n):
    lines = [l for l in out.strip().splitlines() if l.strip()]
    d = json.loads(lines[-1])                                        # the JSON is the LAST line of stdout
    assert d["n_gpus"] == n and d["config"]["parallelism"] == f"dp{n}" and d["config"]["global_batch"] == 2 * n
    assert d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and abs(d["value"] - 2 * n * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]          # whole-job images / the max-over-ranks time
    assert d["unit"] == "images/sec" and "roofline" in d and "cpu_baseline" not in d                        # (the CPU baseline is an N = 1 leg)
    assert all(v == v for v in d["config"]["last_loss_dice"])
    return d


def test_bench_two_ranks_as_the_driver_launches_it():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--batch", "2", "--steps", "2", "--warmup", "1", "--settle", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    _check(r.stdout, 2)


def test_bench_bare_gpus_flag_relaunches_itself_under_torchrun():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--batch", "2", "--steps", "2", "--warmup", "1", "--settle", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    _check(r.stdout, 2)


def test_bench_world_size_mismatch_is_refused():
    env = _env(); env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--batch", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
