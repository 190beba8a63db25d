"""Data-parallel orchestration of an op program (one process per GPU, RCCL over xGMI).

The reference is single-process (no tf.distribute / Horovod anywhere); data parallelism is new.
A program is a list of ops that can be run in [begin,end) slices; `sync_points` say after which op a
buffer must be SUM-reduced across ranks and which op is the first to read the reduced values:
  kind 0/1/2 -- small fp64 vectors (BatchNorm sums, Dice/BCE sums, BN-backward sums; <= 2*1024 doubles, latency-bound).
                use_op == after_op + 1: the very next op needs them -> reduced inline on the compute stream;
                use_op  > after_op + 1: the ops in between are independent (the backward programs put a weight gradient there) ->
                reduced on a side stream beside them, the compute stream waits just before use_op;
  kind 3     -- a contiguous range of the flat gradient buffer whose producers have all been launched:
                reduced on the side (comm) stream so it overlaps the rest of backward; the optimizer waits.
With these reductions every rank computes exactly the single-device large-batch step
(batch-global BN statistics and Dice, gradients SUMMED with the loss normalised by the global count).
"""
from __future__ import annotations

KIND_BN_FWD, KIND_LOSS, KIND_BN_BWD, KIND_GRAD_BUCKET = 0, 1, 2, 3


def run_program(run_range, nops, sync_points, reduce_small, reduce_bucket, finish_buckets, enabled_kinds=(0, 1, 2, 3),
                reduce_small_async=None, wait_small=None):
    """run_range(begin, end): launch ops [begin, end).
    sync_points: iterable of (after_op, kind, handle, count[, use_op]) sorted by after_op.
    reduce_small(handle, count): blocking-in-stream SUM all-reduce of a small buffer.
    reduce_small_async(handle, count) -> token / wait_small(token): the same on a side stream; the wait is issued right before op `use_op`
    (only used when both are given and use_op > after_op + 1).
    reduce_bucket(handle, count): async SUM all-reduce of a gradient range (side stream).
    finish_buckets(): make the compute stream wait for all bucket reductions."""
    begin = 0
    used_bucket = False
    pending = []                                   # (use_op, token), in program order

    def run_to(end):
        nonlocal begin
        while pending and pending[0][0] < end:     # a deferred reduction is needed inside [begin, end): run up to its reader, then wait
            use_op, tok = pending.pop(0)
            if use_op > begin:
                run_range(begin, use_op); begin = use_op
            wait_small(tok)
        if end > begin:
            run_range(begin, end); begin = end

    for sp in sync_points:
        after_op, kind, handle, count = sp[:4]
        use_op = sp[4] if len(sp) > 4 else after_op + 1
        if kind not in enabled_kinds:
            continue
        run_to(after_op + 1)
        if kind == KIND_GRAD_BUCKET:
            reduce_bucket(handle, count); used_bucket = True
        elif reduce_small_async is not None and wait_small is not None and use_op > after_op + 1:
            pending.append((use_op, reduce_small_async(handle, count)))
            pending.sort(key=lambda t: t[0])
        else:
            reduce_small(handle, count)
    run_to(nops)
    for _, tok in pending:                         # (a reader beyond the program's end: nothing may stay in flight)
        wait_small(tok)
    if used_bucket:
        finish_buckets()
