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


def solve_batch_sharded(make_solver, make_problem, nproblems, group=None, lanes=None, device=None, fused="auto",
                        gather=False):
    """One process per GPU: this rank generates and solves problems p = rank, rank+world, ...
    (`make_problem(p)` -> (A, b, x0, op) on this rank's GPU).  Returns {p: solver} for the local
    problems; no collective is issued on the data path.  fused=True uses the lock-step scheduler with batched
    launches (solve_batch_fused, 32 lanes in 2 groups by default), fused=False the one-stream-per-solve scheduler,
    fused="pipelined" the device-resident lock-step scheduler (solve_batch_pipelined: every solver -- SPG with one
    RandomState per solver --, equal-size dense problems, any set built from the seven operators except under
    MPRGP); the default "auto" takes the lock-step scheduler where it applies
    and the fused one otherwise (all of them give bitwise the same results).
    gather=True adds ONE exchange after all solves ("host gathers the result vectors at the
    end"): every rank then gets {p: BatchResult} for ALL problems, results as host arrays."""
    world, rank = 1, 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        world = torch.distributed.get_world_size(group)
        rank = torch.distributed.get_rank(group)
    mine = batch_partition(nproblems, world, rank)
    probs = [make_problem(p) for p in mine]
    solved = None
    if fused == "pipelined":
        solved = solve_batch_pipelined(make_solver, probs, device=device)
    elif fused == "auto" and probs:
        # decided on the argument check alone; whatever the lock-step run itself raises propagates
        from . import lockstep
        made = [make_solver() for _ in probs]           # made ONCE: make_solver may be stateful
        make_solver = iter(made).__next__
        if lockstep.why_not_pipelined(made, probs, get_engine(device).device) is None:
            solved = solve_batch_pipelined(make_solver, probs, device=device)
        else:
            lockstep.stats["auto_declined"] += 1
    if solved is None and fused:
        solved = solve_batch_fused(make_solver, probs, lanes=lanes or 32, groups=2, device=device)
    elif solved is None:
        solved = solve_batch(make_solver, probs, lanes=lanes or 4, device=device)
    local = dict(zip(mine, solved))
    if not gather:
        return local
    packed = {p: BatchResult.of(s) for p, s in local.items()}
    if world == 1:
        return packed
    parts = [None] * world
    torch.distributed.all_gather_object(parts, packed, group=group)
    merged = {}
    for part in parts:
        merged.update(part)
    return merged


class BatchResult:
    """Host-side copy of a finished solve (what travels in the end-of-job gather): the reference's six result
    properties, `solution` as a numpy array."""
    __slots__ = ("solution", "solution_residual", "solution_converged", "solution_time",
                 "solution_num_matrix_vector_multiplications", "executed_matrix_vector_multiplications", "name")

    @classmethod
    def of(cls, solver):
        r = cls()
        x = solver.solution
        r.solution = x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        r.solution_residual = float(solver.solution_residual)
        r.solution_converged = bool(solver.solution_converged)
        r.solution_time = float(solver.solution_time)
        r.solution_num_matrix_vector_multiplications = int(solver.solution_num_matrix_vector_multiplications)
        r.executed_matrix_vector_multiplications = int(solver.executed_matrix_vector_multiplications)
        r.name = solver.name
        return r

    def __getstate__(self):
        return {k: getattr(self, k) for k in self.__slots__}

    def __setstate__(self, state):
        for k, v in state.items():
            setattr(self, k, v)


# ---------------------------------------------------------------------------------------------
# Fused scheduler: the same operation of many in-flight solves becomes ONE batched launch.
# ---------------------------------------------------------------------------------------------
class _Lane:
    __slots__ = ("eng", "pidx", "run", "gen", "req", "t0")

    def __init__(self, eng):
        self.eng = eng
        self.pidx = None
        self.gen = None
        self.req = None


def _flush_group(lanes, grp_eng, ops_buf, codes_buf, counts_buf):
    """Merge the queues of `lanes` position by position: all operations at queue position p that have
    the same signature (opcode + shape) form one group = one launch; groups are issued in position
    order, which preserves every lane's own program order.  One descriptor upload for everything."""
    queues = [(ln.eng, ln.eng.take()) for ln in lanes]
    depth = max((len(q) for _, q in queues), default=0)
    nops = ngroups = 0
    for pos in range(depth):
        buckets = {}
        for eng, q in queues:
            if pos < len(q):
                op = q[pos]
                buckets.setdefault(op[1], []).append((eng, op))
        for sig, members in buckets.items():
            codes_buf[ngroups] = sig[0]
            counts_buf[ngroups] = len(members)
            ngroups += 1
            for eng, op in members:
                eng.write_bop(ops_buf[nops], op)
                nops += 1
    if ngroups:
        rc = grp_eng.lib.ccqp_batch_run_f64(grp_eng.ctx, ngroups, codes_buf, counts_buf, ops_buf, grp_eng.stream)
        if rc:
            from . import _lib
            _lib.check(rc)
    for ln in lanes:
        ln.eng.release()
    return nops, ngroups


def solve_batch_fused(make_solver, problems, lanes=16, groups=2, device=None, quiet=True, stats=None,
                      serialize_matvecs=True):
    """Like solve_batch, but the in-flight solves are executed in lock-step ROUNDS: each solve is advanced
    (on the host only) to its next scalar read while its operations are recorded; the recorded
    operations of all solves of a lane group are merged into batched launches (ccqp_batch_run_f64), the
    scalar reads are enqueued, and the host moves on to the other lane group while the GPU works.
    Per problem the operation sequence and every operand are exactly those of a plain solve(), so the
    results are bitwise identical; only the number of launches changes (x `lanes/groups` fewer).
    """
    import ctypes
    from . import _lib
    from ._engine import Engine
    from ._record import RecordingEngine
    problems = list(problems)
    base = get_engine(device)
    nl = max(1, min(lanes, len(problems)))
    groups = max(1, min(groups, nl))
    torch.cuda.synchronize(base.device)
    streams = [torch.cuda.Stream(device=base.index) for _ in range(groups)]
    grp_engs = [Engine(base.index, s) for s in streams]
    if groups > 1 and serialize_matvecs:
        for ge in grp_engs:
            _lib.check(ge.lib.ccqp_batch_serialize_matvecs(ge.ctx, 1))
    lane_objs = [_Lane(RecordingEngine(base.index, streams[i % groups])) for i in range(nl)]
    grp_lanes = [[ln for i, ln in enumerate(lane_objs) if i % groups == g] for g in range(groups)]
    cap = 4096
    ops_buf = (_lib.Bop * cap)()
    codes_buf = (ctypes.c_int * cap)()
    counts_buf = (ctypes.c_int * cap)()
    solvers = [make_solver() for _ in problems]
    out = io.StringIO()
    pending = list(range(len(problems)))[::-1]
    set_stream = torch.cuda.set_stream
    caller_stream = torch.cuda.current_stream(base.device)
    tot_ops = tot_groups = rounds = 0

    def begin(ln):
        ln.pidx = pending.pop()
        A, b, x0, op = problems[ln.pidx]
        ln.t0 = time.time()
        print("solving " + solvers[ln.pidx]._label, file=out)
        ln.run, ln.gen = solvers[ln.pidx]._start(A, b, x0, op, ln.eng)
        ln.req = None

    def step(ln, values):
        """Resume until the next read request; returns False when the solve has finished."""
        try:
            ln.req = next(ln.gen) if values is None else ln.gen.send(values)
            return True
        except StopIteration as stop:
            x, res, mv = stop.value
            ln.eng.flush_self()                     # trailing operations recorded after the last read
            solvers[ln.pidx]._finish(ln.run, x, res, mv, ln.t0)
            ln.gen = None
            return False

    try:
        with np.errstate(all="ignore"):
            live = [True] * groups
            waiting = [None] * groups
            gbuf = (ctypes.c_double * (64 * 16))()
            gslabs = (ctypes.c_void_p * 64)()
            gslot0 = (ctypes.c_int * 64)()
            gcount = (ctypes.c_int * 64)()
            while any(live):
                for g in range(groups):
                    if not live[g]:
                        continue
                    set_stream(streams[g])
                    gathered = {}
                    if waiting[g]:
                        # one wait + one host copy for all lanes of the group (previous round's read)
                        wl = waiting[g]
                        rc = grp_engs[g].lib.ccqp_scalars_gather_host(grp_engs[g].ctx, len(wl), gbuf)
                        if rc:
                            _lib.check(rc)
                        for i, ln in enumerate(wl):
                            base_i = i * 16
                            gathered[id(ln)] = [np.float64(gbuf[base_i + j]) for j in range(ln.req[1])]
                        waiting[g] = None
                    busy = []
                    for ln in grp_lanes[g]:
                        while True:
                            if ln.gen is None:
                                if not pending:
                                    break
                                begin(ln)
                                vals = None
                            else:
                                vals = gathered.pop(id(ln))     # delivered by this group's previous round
                            if step(ln, vals):
                                busy.append(ln)
                                break
                            # finished: try to start the next problem on this lane right away
                    if not busy:
                        live[g] = False
                        continue
                    if len(ops_buf) < sum(len(ln.eng.queue) for ln in busy):
                        raise RuntimeError("solve_batch_fused: descriptor buffer too small")
                    a, b_ = _flush_group(busy, grp_engs[g], ops_buf, codes_buf, counts_buf)
                    tot_ops += a
                    tot_groups += b_
                    rounds += 1
                    for chunk0 in range(0, len(busy), 64):
                        part = busy[chunk0:chunk0 + 64]
                        if chunk0:
                            raise RuntimeError("solve_batch_fused: at most 64 lanes per group")
                        for i, ln in enumerate(part):
                            gslabs[i] = ln.eng.slab_ptr
                            gslot0[i] = ln.req[0]
                            gcount[i] = ln.req[1]
                        rc = grp_engs[g].lib.ccqp_scalars_gather_async(grp_engs[g].ctx, len(part), gslabs, gslot0,
                                                                       gcount, grp_engs[g].stream)
                        if rc:
                            _lib.check(rc)
                    waiting[g] = busy
    finally:
        set_stream(caller_stream)
    torch.cuda.synchronize(base.device)
    if stats is not None:
        stats.update(ops=tot_ops, launches=tot_groups, rounds=rounds)
    if not quiet:
        print(out.getvalue(), end="")
    return solvers


# the lock-step schedulers (device-resident scalars and branch decisions) live in lockstep.py
from .lockstep import StackedResult, solve_batch_pipelined, solve_stacked  # noqa: E402,F401
