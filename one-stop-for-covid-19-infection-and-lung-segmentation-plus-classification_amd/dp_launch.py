"""Data-parallel launcher of the task runners: one process per GPU on one node, RCCL (torch.distributed backend "nccl") over xGMI.

The reference trains single-process (Keras on one device, T3:989-1009); BASELINE.json configs[2] asks for runner_lung_segmentation() at a
global batch of 64 on 8 MI355X.  With UNET_GPUS=N (N > 1) in the environment a runner called from ONE process -- app.py's `six`
through dropin/run_app.py, or directly -- re-launches itself as N ranks:

    python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=N -m covidseg_amd.dp_launch <runner>

Every rank builds the same model (same seed), walks the same shuffled mini-batches and takes its contiguous shard of each
(keras_like.dp_shard); the engine reduces the BatchNorm / Dice sums inline and the gradient buckets on a side stream (engine.py, dp.py), so
N ranks x batch B/N compute the single-device batch-B step.  Rank 0 prints and writes the checkpoints; the launching process gets the
runner's outputs back through a JSON file -- a REDUCED return value: scalars and lists (arrays arrive as nested lists), no "model" entry
(the trained weights are in the checkpoint files rank 0 wrote: load them with UNetModel.load_weights).

Environment: UNET_GPUS (ranks), UNET_DP_BACKEND (nccl | gloo: the gloo path stages reductions through the host, used by the single-GPU
tests), UNET_DP_ONE_DEVICE=1 (all ranks on cuda:0, tests)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _jsonable(v):
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items() if _jsonable(x) is not None or x is None}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    return None                                   # model objects etc. do not travel


def maybe_launch(runner_name: str, kw: dict):
    """Called at the top of every runner.  Returns None when the runner should just run in this process (UNET_GPUS unset / 1, or this
    process already is a rank); otherwise launches the ranks, waits, and returns the runner's JSON-able outputs."""
    n = int(os.environ.get("UNET_GPUS", "1") or 1)
    if n <= 1 or "WORLD_SIZE" in os.environ or kw.get("backend") is not None or kw.get("process_group") is not None:
        return None
    with tempfile.TemporaryDirectory(prefix="unet_dp_") as tmp:
        env = dict(os.environ)
        kw = dict(kw)
        if kw.get("data") is not None:            # arrays travel through an .npz the ranks read (runners._get_data: UNET_DATA_NPZ)
            x, y = kw.pop("data")
            np.savez(os.path.join(tmp, "data.npz"), x=np.asarray(x), y=np.asarray(y))
            env["UNET_DATA_NPZ"] = os.path.join(tmp, "data.npz")
        if kw.get("init_weights") is not None:
            np.savez(os.path.join(tmp, "init.npz"), **{k: np.asarray(v) for k, v in kw.pop("init_weights").items()})
            env["UNET_DP_INIT_NPZ"] = os.path.join(tmp, "init.npz")
        kw["workdir"] = os.path.abspath(kw.get("workdir", "."))
        bad = [k for k, v in kw.items() if _jsonable(v) is None and v is not None]
        if bad:
            raise ValueError(f"data-parallel launch: arguments {bad} cannot be passed to the ranks")
        env["UNET_DP_KW"] = json.dumps(_jsonable(kw))
        env["UNET_DP_RESULT"] = os.path.join(tmp, "result.json")
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        # --standalone: torchrun picks and owns the rendezvous port (no bind-close-reuse race with other processes)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
               "-m", "covidseg_amd.dp_launch", runner_name]
        rc = subprocess.run(cmd, env=env).returncode
        if rc != 0:
            raise RuntimeError(f"data-parallel run of {runner_name} on {n} ranks failed (exit code {rc})")
        with open(env["UNET_DP_RESULT"]) as f:
            return json.load(f)


def main(argv):
    import torch
    import torch.distributed as dist
    from . import runners
    name = argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("UNET_DP_BACKEND", "nccl")
    dev = 0 if os.environ.get("UNET_DP_ONE_DEVICE") else local
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    kw = json.loads(os.environ.get("UNET_DP_KW", "{}"))
    for k in ("device", "process_group"):         # set by this launcher per rank below: a forwarded copy would collide
        kw.pop(k, None)
    if os.environ.get("UNET_DP_INIT_NPZ"):
        kw["init_weights"] = dict(np.load(os.environ["UNET_DP_INIT_NPZ"]))
    real_stdout = sys.stdout
    if rank != 0:
        sys.stdout = open(os.devnull, "w")        # rank 0 speaks for the job
    try:
        out = getattr(runners, name)(process_group=dist.group.WORLD, device=dev, **kw)
    finally:
        sys.stdout = real_stdout
    if rank == 0 and os.environ.get("UNET_DP_RESULT"):
        res = _jsonable({k: v for k, v in out.items() if k != "model"})
        res["world_size"] = world
        with open(os.environ["UNET_DP_RESULT"], "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv)
