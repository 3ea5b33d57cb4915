"""Batch-sharded mode: many independent QPs, no collective (BASELINE config 5).

Problem p of a batch is owned by rank `p % world` (operators.batch_partition).  On its GPU a rank
keeps `lanes` solves in flight at once: every solve runs on its own HIP stream with its own C-ABI
context (scalar slab + pinned mirror), and the solver loops are generators that yield at each point
where the host needs scalars.  The scheduler below answers a yield by enqueueing the asynchronous
slab read, parking that solve and resuming the next lane -- so while one problem waits for its
scalars the GPU runs the kernels of the others and the host keeps launching.  Every solve still
executes exactly the reference's sequence of operations, so per-problem results and MV counts are
identical to a sequential run (asserted in tests/test_gpu_batch.py).
"""
import contextlib
import io
import time

import numpy as np
import torch

from ._engine import get_engine, make_lanes
from .operators import batch_partition


def solve_batch(make_solver, problems, lanes=4, device=None, quiet=True):
    """Solve independent problems concurrently on one GPU.

    make_solver : () -> a fresh CCQPSolver* instance
    problems    : sequence of (A, b, x0, convex_proj_op) with A / b device-resident (operator or tensor)
    Returns the list of solver objects (same order), each with the usual result properties.
    """
    problems = list(problems)
    base = get_engine(device)
    nl = max(1, min(lanes, len(problems)))
    lane_engines = make_lanes(base.index, nl)
    torch.cuda.synchronize(base.device)        # inputs were produced on other streams
    solvers = [make_solver() for _ in problems]
    out = io.StringIO()
    pending = list(range(len(problems)))[::-1]
    active = {}                                # lane -> [pidx, run, gen, request, event, t0]

    def start(lane):
        pidx = pending.pop()
        A, b, x0, op = problems[pidx]
        eng = lane_engines[lane]
        t0 = time.time()
        set_stream(eng.own_stream)
        print("solving " + solvers[pidx]._label, file=out)
        run, gen = solvers[pidx]._start(A, b, x0, op, eng)
        state = [pidx, run, gen, None, None, t0]
        advance(state, None)
        return state

    def advance(state, values):
        """Resume a solve until its next sync point (or its end)."""
        pidx, run, gen = state[0], state[1], state[2]
        try:
            req = next(gen) if values is None and state[3] is None else gen.send(values)
            state[3] = req
            run.eng.read_async(*req)
        except StopIteration as stop:
            x, res, mv = stop.value
            solvers[pidx]._finish(run, x, res, mv, state[5])
            state[2] = None

    set_stream = torch.cuda.set_stream
    caller_stream = torch.cuda.current_stream(base.device)
    try:
      with np.errstate(all="ignore"):
        for lane in range(nl):
            if pending:
                active[lane] = start(lane)
        while active:
            for lane in list(active):
                state = active[lane]
                if state[2] is None:
                    del active[lane]
                    if pending:
                        active[lane] = start(lane)
                    continue
                eng = state[1].eng
                vals = eng.fetch(*state[3])     # usually already there: other lanes ran meanwhile
                set_stream(eng.own_stream)      # allocations below belong to this lane's stream
                advance(state, vals)
    finally:
        set_stream(caller_stream)
    torch.cuda.synchronize(base.device)
    if not quiet:
        print(out.getvalue(), end="")
    return solvers


def solve_batch_sharded(make_solver, make_problem, nproblems, group=None, lanes=4, device=None):
    """One process per GPU: this rank generates and solves problems p = rank, rank+world, ...
    (`make_problem(p)` -> (A, b, x0, op) on this rank's GPU).  Returns {p: solver} for the local
    problems; no collective is issued."""
    world, rank = 1, 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        world = torch.distributed.get_world_size(group)
        rank = torch.distributed.get_rank(group)
    mine = batch_partition(nproblems, world, rank)
    solved = solve_batch(make_solver, [make_problem(p) for p in mine], lanes=lanes, device=device)
    return dict(zip(mine, solved))
