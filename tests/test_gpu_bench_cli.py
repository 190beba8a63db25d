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
    _check_comm(d, n)
    return d


def _check_comm(d, n, buckets=True, sync_bn=True):
    """the N > 1 line explains itself (all-reduce time exposed vs hidden)"""
    c = d["comm"]
    assert c["world"] == n and c["comm_status"] == 0 and c["sync_bn"] is sync_bn and c["grad_buckets"] is buckets
    assert c["small_allreduce_in_use"] in (("device (comm.hip)", "torch.distributed") if sync_bn else (None,))
    assert (c["small_allreduce_fallback"] is None) == (c["small_allreduce_in_use"] != "torch.distributed" or c["small_allreduce_requested"] == "rccl")
    nb = 5 if buckets else 1
    assert len(c["buckets_rank0"]) == nb == len(c["buckets_ms_max_over_ranks"])
    assert abs(sum(b["mb"] for b in c["buckets_rank0"]) - 7762401 * 4 / 1e6) < 0.1                           # the five buckets are the whole gradient buffer
    assert all(b["ms"] > 0 for b in c["buckets_rank0"])
    ar, ex, hid = c["allreduce_ms_per_step"], c["exposed_ms_per_step"], c["hidden_ms_per_step"]
    assert ar["max"] >= ar["rank0"] > 0 and ex["max"] >= ex["rank0"] >= 0 and hid["rank0"] >= 0
    assert abs(ar["rank0"] - ex["rank0"] - hid["rank0"]) < 1e-3 or hid["rank0"] == 0
    assert c["small_reductions_per_step"] == (17 if sync_bn else 0)
    r = c["rank_ms_per_step"]
    assert len(r["all"]) == n and r["min"] <= r["max"] <= d["ms_per_step"] * 1.001 + 1e-3


def test_bench_two_ranks_as_the_driver_launches_it():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--batch", "2", "--steps", "2", "--warmup", "1", "--settle", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    _check(r.stdout, 2)


@pytest.mark.parametrize("n", [4, 8])
def test_bench_four_and_eight_ranks_on_the_one_gpu(n):
    """The world sizes the driver's scaling run uses beyond 2 (8 ranks): bucket order, `comm` keys, per-rank spread and the whole-job value at
    4 and 8 ranks -- every rank on cuda:0, gloo for the buckets, the device-side small all-reduce through IPC between 4 / 8 processes."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--size", "64", "--batch", "2", "--steps", "2", "--warmup", "1", "--settle", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check(r.stdout, n)
    assert len(d["comm"]["rank_ms_per_step"]["all"]) == n


def test_bench_comm_selftest_prints_a_diagnosis_without_a_training_step():
    """bench.py --comm-selftest (4 ranks on the one GPU): IPC mapping + self-test of the device-side small all-reduce, 1000 calls through either path (exact on every
    rank), ten all-reduces of the 31 MB gradient buffer -- one JSON line, no step."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "4", "--size", "64", "--batch", "2", "--comm-selftest"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.strip()][-1])["comm_selftest"]
    assert d["world"] == 4 and d["all_exact"] is True and d["statuses"] == [0, 0, 0, 0]
    r0 = d["rank0"]
    assert r0["small_allreduce"] == "device (comm.hip)" and r0["device_us_per_call"] > 0 and r0["torch_distributed_us_per_call"] > 0
    assert abs(r0["gradient_buffer_mb"] - 7762401 * 4 / 1e6) < 0.1 and r0["gradient_allreduce_ms"] > 0 and set(d["max_over_ranks"]) == {"device_us_per_call", "torch_distributed_us_per_call", "gradient_allreduce_ms"}


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


def test_bench_comm_object_at_world_one_and_the_ab_switches():
    """UNET_BENCH_FORCE_PG=1 walks the data-parallel program (RCCL backend) at world 1: the line carries the same `comm` object; --no-buckets is ONE all-reduce whose whole
    duration is exposed, --no-sync-bn has no small reductions"""
    env = _env(); env.pop("UNET_BENCH_BACKEND"); env["UNET_BENCH_FORCE_PG"] = "1"
    for extra, kw in (([], {}), (["--no-buckets"], {"buckets": False}), (["--no-sync-bn"], {"sync_bn": False})):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--size", "64", "--batch", "2", "--steps", "2", "--warmup", "1", "--settle", "0", "--no-cpu-baseline", "--no-strict-leg",
               "--no-fit-leg"] + extra
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([l for l in r.stdout.strip().splitlines() if l.strip()][-1])
        assert d["n_gpus"] == 1
        _check_comm(d, 1, **kw)
        if extra == ["--no-buckets"]:
            c = d["comm"]
            assert abs(c["exposed_ms_per_step"]["rank0"] - c["allreduce_ms_per_step"]["rank0"]) < 1e-6 and c["hidden_ms_per_step"]["rank0"] == 0
