"""Data-parallel orchestration of an op program (one process per GPU, RCCL over xGMI).

The reference is single-process (no tf.distribute / Horovod anywhere); data parallelism is new.
A program is a list of ops that can be run in [begin,end) slices; `sync_points` say after which op a
buffer must be SUM-reduced across ranks before the next op may run:
  kind 0/1/2 -- small fp64 vectors on the critical path (BatchNorm sums, Dice/BCE sums, BN-backward sums):
                reduced inline on the compute stream (latency-bound, <= 2*1024 doubles);
  kind 3     -- a contiguous range of the flat gradient buffer whose producers have all been launched:
                reduced on the side (comm) stream so it overlaps the rest of backward.
With these reductions every rank computes exactly the single-device large-batch step
(batch-global BN statistics and Dice, gradients SUMMED with the loss normalised by the global count).
"""
from __future__ import annotations

KIND_BN_FWD, KIND_LOSS, KIND_BN_BWD, KIND_GRAD_BUCKET = 0, 1, 2, 3


def run_program(run_range, nops, sync_points, reduce_small, reduce_bucket, finish_buckets, enabled_kinds=(0, 1, 2, 3)):
    """run_range(begin, end): launch ops [begin, end).
    sync_points: iterable of (after_op, kind, handle, count) sorted by after_op.
    reduce_small(handle, count): blocking-in-stream SUM all-reduce of a small buffer.
    reduce_bucket(handle, count): async SUM all-reduce of a gradient range (side stream).
    finish_buckets(): make the compute stream wait for all bucket reductions."""
    begin = 0
    used_bucket = False
    for after_op, kind, handle, count in sync_points:
        if kind not in enabled_kinds:
            continue
        if after_op + 1 > begin:
            run_range(begin, after_op + 1)
            begin = after_op + 1
        if kind == KIND_GRAD_BUCKET:
            reduce_bucket(handle, count); used_bucket = True
        else:
            reduce_small(handle, count)
    if begin < nops:
        run_range(begin, nops)
    if used_bucket:
        finish_buckets()
