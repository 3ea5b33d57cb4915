"""Lock-step batch schedulers: many dense problems of one size advance together, with every scalar and every branch
decision of the solver on the device (csrc/ccqp_bpipe.hip: PGD / BBPGD / BBPGDf and APGD / anti-relaxation APGD;
csrc/ccqp_bmprgp.hip: MPRGP; csrc/ccqp_bsegs.hip: the same on Sphere / Cone / DisjointUnion sets) -- or, for SPG-QP, with the
scalar part of each iteration done in C on the host for the whole batch at once (csrc/ccqp_bspg.hip).

  solve_batch_pipelined   a list of (A, b, x0, set) problems -> solver objects (grouped by size)
  solve_stacked           stacked tensors A (B, n, n), b (B, n) -> stacked results, no per-problem host work

Two execution forms, chosen by size: lock-step ROUNDS (2-5 launches per matvec for the whole batch, the host polls a
progress word in pinned memory) and, for small matrices, ONE launch in which each problem is solved inside its own
workgroup.  Either way problem p gets bit for bit the result of its own `solve()`: the kernels run the bodies of the
single-problem kernels with the same reduction geometry and the same matvec lane map, and the scalar arithmetic of
the host loops operation for operation.  The host-driven schedulers (batch.py) remain the general ones: any solver,
operator and feasible set."""
import time

import numpy as np
import torch

from ._engine import get_engine

# Problems the device-resident runs handed back to the host-driven loop (matvec budget reached, device budget of a
# single-launch kernel, MPRGP's unreachable branches).  The fall-back keeps every result right, so a device-side bug
# can hide behind it: the tests assert that this counter does not move where no fall-back is expected.
stats = {"fallback_solves": 0, "auto_declined": 0, "halved_runs": 0, "grid_solves": 0, "grid_gave_up": 0}

_BPIPE_MAX = 16384        # problems per lock-step group (gridDim.y <= 65535; keeps the workspace bounded)
# The single-launch form (one workgroup solves a whole problem: ccqp_bsolve_*) is a form for MANY SMALL problems: a
# workgroup sits on one CU and walks its matrix alone, so it needs problems for every CU several times over -- and small
# ones -- before it beats the lock-step rounds, which round 4 made cheaper (fused APGD rounds, the fused MPRGP stage
# launch, the lean roll-call).  One workgroup per problem against the rounds (profiles/r04_single_launch_vs_rounds.jsonl:
# whole batches through solve_batch_pipelined, host staging included; r04_stacked_single_launch_vs_rounds.jsonl: stacked
# tensors, device-bound):
#   BBPGD  n <= 128: x1.4-2.4 stacked, level to x1.3 through the list API at every count; 256: x1.03-1.12 from 256
#          problems on, x0.70-0.80 below; 512: x0.87-0.94 stacked, x0.37-0.80 below 1024 problems
#   APGD   (its single-launch kernel back under 128 registers -- the 80 loads in flight of the lock-step decision had cost
#          it half its workgroups per CU) n = 32: x1.06 list, x1.7-1.9 stacked; 64: x0.72-0.79 up to 32 problems, level at
#          256, x1.17-1.22 stacked; 128: x0.47-0.82 up to 256 problems, x1.06-1.26 from 1024 on; 256: x0.46-0.79
#   MPRGP  n = 32: x1.2; 64: level below 256 problems, x1.08-1.21 above; 128: x0.69-0.94 up to 256, level beyond; 256 /
#          512: x0.56-0.69
# Default of `single_launch` (None), per solver: the size up to which it is taken whatever the count, then (size, smallest
# count) steps.  (Until round 4: BBPGD n <= 512, APGD / MPRGP n <= 128, whatever the count.)
_SINGLE_LAUNCH_RULE = {"bbpgd": (128, ((256, 256),)),
                       "apgd": (32, ((64, 256), (128, 1024))),
                       "mprgp": (32, ((64, 256), (128, 1024)))}


def _single_launch_default(kind, n, count):
    always, steps = _SINGLE_LAUNCH_RULE[kind]
    return n <= always or any(n <= size and count >= least for size, least in steps)


def _device_budget(n):
    """Matvecs a single-launch kernel may spend on one problem.  A kernel must END whatever the data (a diverging or
    NaN-poisoned solve never meets its tolerance, and the reference then loops forever -- on the host, where it can be
    interrupted): about two seconds of one workgroup's time, estimated from the chain of an iteration (~15 us) or the
    matrix bytes one CU streams (~20 GB/s).  A problem that needs more goes back to the host-driven loop."""
    return int(min(20000, max(256, 2.0 / max(15e-6, 8.0 * n * n / 20e9))))


_KINDS = None
_lib_mod = _ss_mod = _dense_op = None


def _lockstep_kinds():
    global _KINDS
    if _KINDS is None:
        from .solvers import (CCQPSolverAPGD, CCQPSolverAPGDAntiRelaxation, CCQPSolverBBPGD, CCQPSolverBBPGDf,
                              CCQPSolverMPRGP, CCQPSolverPGD, CCQPSolverSPG)
        _KINDS = (CCQPSolverBBPGD, CCQPSolverBBPGDf, CCQPSolverAPGD, CCQPSolverAPGDAntiRelaxation, CCQPSolverMPRGP,
                  CCQPSolverPGD, CCQPSolverSPG)
    return _KINDS


def _why_not_batch(first):
    """The batch-wide part of the argument check (the solver class)."""
    kind = type(first)
    if kind not in _lockstep_kinds() or getattr(first, "reuse_matvec", False):
        return TypeError("solve_batch_pipelined runs CCQPSolverPGD / BBPGD / BBPGDf / APGD / APGDAntiRelaxation / MPRGP / "
                         "SPG (no matvec reuse), got %s" % kind.__name__)
    return None


_SPG_MAX_M = 64


def _why_not_spg(s):
    """SPG in lock-step: every solver draws from its OWN RandomState (with numpy's global stream -- the reference's
    default -- the draws of problem p start where problem p-1 stopped, which is unknown while they advance together)."""
    if not isinstance(getattr(s, "rng", None), np.random.RandomState):
        return TypeError("solve_batch_pipelined runs CCQPSolverSPG only with a private generator per solver: set "
                         "solver.rng = np.random.RandomState(seed) (numpy's global stream cannot be shared by "
                         "problems that advance together)")
    if s.trace is not None:
        return TypeError("solve_batch_pipelined: a traced SPG solve runs on the host-driven loop")
    if not (isinstance(s.m, (int, np.integer)) and 1 <= s.m <= _SPG_MAX_M):
        return ValueError("solve_batch_pipelined: SPG's queue length m must be an integer in [1, %d]" % _SPG_MAX_M)
    return None


def _why_not_one(kind, max_mv, s, A, b, P, device):
    """One problem of the argument check: (reason or None, n, the set with the default filled in).  No staging, no
    GPU work."""
    global _lib_mod, _ss_mod, _dense_op
    if _dense_op is None:                               # (imported once: this runs per problem of a batch)
        from . import _lib as _lib_mod
        from . import solution_spaces as _ss_mod
        from .operators import DenseOperator as _dense_op
    _lib, ss, DenseOperator = _lib_mod, _ss_mod, _dense_op
    if type(s) is not kind or getattr(s, "reuse_matvec", False):
        return TypeError("solve_batch_pipelined: one solver class for the whole batch"), None, P
    if s.max_matrix_vector_multiplications != max_mv:
        return ValueError("solve_batch_pipelined: one matvec budget for the whole batch"), None, P
    if kind.__name__ == "CCQPSolverSPG":
        reason = _why_not_spg(s)
        if reason is not None:
            return reason, None, P
    if hasattr(A, "matvec") and hasattr(A, "engine"):
        op_ok = isinstance(A, DenseOperator) and A.rows == A.n
        n = getattr(A, "n", None)
        dev = A.A.device if op_ok else None
    else:
        op_ok = getattr(A, "ndim", None) == 2 and A.shape[0] == A.shape[1] and \
            getattr(A, "layout", torch.strided) == torch.strided and not hasattr(A, "tocsr")
        n = A.shape[1] if op_ok else None
        dev = A.device if isinstance(A, torch.Tensor) and A.is_cuda else None
    if not op_ok:
        return TypeError("solve_batch_pipelined needs square dense operators"), None, P
    if b.shape[0] != n:
        return ValueError("solve_batch_pipelined: b does not match A"), n, P
    if dev is not None and dev != device:
        return ValueError("solve_batch_pipelined: operator on another device"), n, P
    P = P if P is not None else ss.IdentityProjOp(n)
    if not hasattr(P, "_dev_bb_tail"):
        if isinstance(P, ss.ProjOpBase):
            P._refresh()                                 # (host-only: its blocks may have been edited or replaced)
        # Sphere / Cone / DisjointUnion: the segmented lock-step forms (ccqp_bsegs.hip; MPRGP: ccqp_bmprgp.hip)
        if not isinstance(P, ss.ProjOpBase) or not P._has_device_form():
            return TypeError("solve_batch_pipelined needs feasible sets built from the seven operators, got %r"
                             % type(P).__name__), n, P
        if kind.__name__ == "CCQPSolverMPRGP" and not P._normal_everywhere_on_device():
            # a Cone block has no normal vector (the reference raises NotImplementedError, solution_spaces.py:465, and so
            # does the host-driven loop this problem is left to); scalar bounds make normal_vector raise a TypeError
            return TypeError("solve_batch_pipelined runs MPRGP on sets whose normal_vector exists on the device (no "
                             "Cone block, array bounds), got %r" % type(P).__name__), n, P
        if getattr(P, "_fixed_dim", False) and P.embedded_dimension != n:
            return ValueError("solve_batch_pipelined: the feasible set does not match b"), n, P
        return None, n, P
    if kind.__name__ == "CCQPSolverMPRGP" and P._scalar_bound_names():
        # (the reference's normal_vector indexes the bounds: a TypeError for scalar ones.  Tiny problems, n <= 64, are
        # taken since round 4: the one order-sensitive decision of an iteration, |x - P(x)| > 1e-8, is summed there as the
        # literal path's thread-per-block kernel sums it -- split_body, ccqp_bodies.h)
        return TypeError("solve_batch_pipelined runs MPRGP on sets with array bounds"), n, P
    return None, n, P


def why_not_pipelined(solvers, problems, device):
    """Argument check of solve_batch_pipelined WITHOUT any staging or GPU work: None if the lock-step scheduler can take
    the batch, else the TypeError / ValueError it would raise (returned, not raised).  `solve_batch_sharded(fused="auto")`
    decides on this alone, so that nothing raised while a batch is being staged or run can be mistaken for "does not
    apply" and silently re-solved by another scheduler."""
    reason = _why_not_batch(solvers[0])
    if reason is not None:
        return reason
    kind, max_mv = type(solvers[0]), solvers[0].max_matrix_vector_multiplications
    for s, (A, b, x0, P) in zip(solvers, problems):
        reason = _why_not_one(kind, max_mv, s, A, b, P, device)[0]
        if reason is not None:
            return reason
    return _why_not_shared_rng(solvers)


def _why_not_shared_rng(solvers):
    if type(solvers[0]).__name__ == "CCQPSolverSPG" and len({id(s.rng) for s in solvers}) != len(solvers):
        return TypeError("solve_batch_pipelined: SPG solvers of one batch must not share a RandomState object")
    return None


def solve_batch_pipelined(make_solver, problems, device=None, quiet=True, ahead=4, tune=0, single_launch=None,
                          timing=None):
    """Many dense problems (grouped by size n: one run per size), each solved by PGD, BBPGD, BBPGDf, APGD, anti-relaxation
    APGD or MPRGP with every scalar and branch decision on the device (ccqp_bpipe_* / ccqp_bapgd_* / ccqp_bmprgp_*):
    lock-step rounds of 2-5 launches per matvec for the whole batch while the host only polls a progress word, or --
    small matrices, many of them (by default: BBPGD-type n <= 128, <= 256 from 256 problems on; APGD and MPRGP n <= 32,
    <= 64 from 256 problems, <= 128 from 1024 -- _SINGLE_LAUNCH_RULE; `single_launch` forces it on, up to
    n = 2048, or off) -- ONE launch in which every problem is solved inside its own workgroup.  The host-driven
    schedulers of batch.py are bound by ~20 us of Python per problem and scalar read at n <= 4096; this one is not.

    make_solver : () -> CCQPSolverPGD, CCQPSolverBBPGD, CCQPSolverBBPGDf, CCQPSolverAPGD,
                  CCQPSolverAPGDAntiRelaxation, CCQPSolverMPRGP (no matvec reuse) or CCQPSolverSPG (each with its own
                  `rng = np.random.RandomState(...)`; its generator is left where its own solve() would leave it),
                  all of one class with the same max_matrix_vector_multiplications
    problems    : sequence of (A, b, x0, convex_proj_op); A an (n, n) device tensor / DenseOperator, the feasible set
                  elementwise (Identity / Lower / Upper / Box) or None
    Returns the list of solver objects with the usual result properties; per problem x, residual, converged flag
    and MV count are bit for bit those of `make_solver().solve(A, b, x0, convex_proj_op)`.  `solution_time` is
    the wall-clock of the whole batch.  Problems that would run into the matvec budget (APGD / MPRGP, whose exit paths
    at the budget have quirks of their own) or into a single-launch kernel's own budget are handed to plain solve()
    and counted in `lockstep.stats`.  `timing`, if given, receives the wall-clock split {staging_s: argument checks
    and per-problem staging on the host, run_s: tables, launches, wait, result objects}."""
    from . import _lib
    from . import solution_spaces as ss
    from .operators import DenseOperator, as_operator
    from .solvers import (CCQPSolverAPGD, CCQPSolverAPGDAntiRelaxation, CCQPSolverBBPGD, CCQPSolverBBPGDf,
                          CCQPSolverMPRGP, CCQPSolverPGD, CCQPSolverSPG)
    problems = list(problems)
    solvers = [make_solver() for _ in problems]
    if not problems:
        return solvers
    t0 = time.time()
    eng = get_engine(device)
    eng.bind()
    kind = type(solvers[0])
    reason = _why_not_batch(solvers[0])
    if reason is not None:
        raise reason
    max_mv = solvers[0].max_matrix_vector_multiplications
    by_size = {}                                        # (n, elementwise set?) -> items, in order of first appearance
    refreshed, memo = set(), {}
    for s, (A, b, x0, P0) in zip(solvers, problems):
        # argument check and staging of one problem in ONE pass (many small problems: the host side counts); the check
        # itself does no GPU work, and the first failing problem raises before anything is launched
        reason, n, P = _why_not_one(kind, max_mv, s, A, b, P0, eng.device)
        if reason is not None:
            raise reason
        op = as_operator(A)
        if id(P) not in refreshed:                       # once per distinct set object and call
            refreshed.add(id(P))
            if hasattr(P, "_dev_bb_tail"):
                P._refresh(memo)                         # (elementwise sets over shared bound arrays: one comparison)
            else:
                P._refresh()
        bd = eng.to_device(b)
        if bd.dim() != 1:                                # (reshape of an already flat tensor: ~1 us per problem for nothing)
            bd = bd.reshape(-1)
        by_size.setdefault((n, hasattr(P, "_dev_bb_tail")), []).append(
            (s, op, bd, None if x0 is None else eng.to_device(x0).reshape(-1), P, isinstance(b, torch.Tensor),
             (A, b, x0, P)))
    reason = _why_not_shared_rng(solvers)
    if reason is not None:
        raise reason
    if not quiet:
        for s in solvers:
            print("solving " + s._label)
    group = {CCQPSolverBBPGD: _bpipe_group, CCQPSolverBBPGDf: _bpipe_group, CCQPSolverPGD: _bpipe_group,
             CCQPSolverAPGD: _bapgd_group,
             CCQPSolverAPGDAntiRelaxation: _bapgd_group, CCQPSolverMPRGP: _bmprgp_group,
             CCQPSolverSPG: _bspg_group}[kind]
    t1 = time.time()
    with np.errstate(all="ignore"):
        for (n, elementwise), items in by_size.items():     # one lock-step run per problem size (and kind of set)
            for lo in range(0, len(items), _BPIPE_MAX):
                if elementwise:
                    group(eng, items[lo:lo + _BPIPE_MAX], n, max_mv, ahead, tune, single_launch)
                elif kind in (CCQPSolverAPGD, CCQPSolverAPGDAntiRelaxation):
                    _bapgd_group(eng, items[lo:lo + _BPIPE_MAX], n, max_mv, ahead, tune, False, segmented=True)
                elif kind is CCQPSolverSPG:
                    _bspg_group(eng, items[lo:lo + _BPIPE_MAX], n, max_mv, ahead, tune, segmented=True)
                elif kind is CCQPSolverMPRGP:
                    _bmprgp_group(eng, items[lo:lo + _BPIPE_MAX], n, max_mv, ahead, tune, segmented=True)
                else:
                    _bsegs_group(eng, items[lo:lo + _BPIPE_MAX], n, max_mv, ahead, tune)
    dt = time.time() - t0
    if timing is not None:                              # host-side staging of the problem list vs the runs proper
        timing.update(staging_s=t1 - t0, run_s=time.time() - t1)
    for s in solvers:
        s._solution_time = dt
    return solvers


def _host_solve(s, A, b, x0, P):
    """The host-driven loop of solver `s` (whose exit paths are the reference's) for a problem the device-resident run
    could not finish: `pipeline` off for the call -- the default None would send the solve through the device-resident
    form of a batch of one first, to find out the same thing again."""
    had = hasattr(s, "pipeline")
    keep = getattr(s, "pipeline", None)
    if had:
        s.pipeline = False
    try:
        return s.solve(A, b, x0=x0, convex_proj_op=P)
    finally:
        if had:
            s.pipeline = keep


def _bpipe_poll_loop(eng, B, launch, last_round, ahead):
    """Queue rounds 1, 2, ... (`launch(k)`) at most `ahead` beyond the last one the device reported, until every
    problem has converged or `last_round` has been queued.  Returns the number of rounds queued."""
    import ctypes
    lib, ctx = eng.lib, eng.ctx
    state = (ctypes.c_double * 2)()
    k = 0
    while k < last_round:
        k += 1
        launch(k)
        if k >= last_round:                             # (the last round of a budget may report nothing)
            break
        lib.ccqp_bpipe_poll(ctx, state)
        spins = 0
        while state[0] < B and k - state[1] >= ahead:
            lib.ccqp_bpipe_poll(ctx, state)
            spins += 1
            if spins > 2000000:                         # mirror silent for ~1 s: wait for the queue itself
                torch.cuda.current_stream(eng.device).synchronize()
                lib.ccqp_bpipe_poll(ctx, state)
                if state[1] != k:
                    raise RuntimeError("solve_batch_pipelined: the device never reported round %d" % k)
                spins = 0
        if state[0] >= B:
            break
    return k


def _bpipe_poll_many(runs, ahead):
    """_bpipe_poll_loop for several independent lock-step runs at once (each with its own engine = its own stream and
    host mirror; `run.eng`, `run.B`, `run.launch(k)`, `run.last`): the host round-robins between them, keeping each at
    most `ahead` rounds beyond its own last report, until each has converged or reached its last round.  With two runs
    on two streams the device always has one run's matvec to execute while the other run's small decision kernels and
    their launch gaps go by.  Sets `run.k` = rounds queued."""
    import ctypes
    state = (ctypes.c_double * 2)()
    active = []
    for r in runs:
        r.k = 0
        if r.last >= 1:                                  # the first round is queued before the first look at the mirror
            r.k = 1
            r.launch(1)
            if r.last > 1:
                active.append(r)
    idle = 0
    while active:
        moved = False
        for r in list(active):
            r.eng.lib.ccqp_bpipe_poll(r.eng.ctx, state)
            if state[0] >= r.B:
                active.remove(r)
                moved = True
            elif r.k - state[1] < ahead:
                r.k += 1
                r.launch(r.k)
                moved = True
                if r.k >= r.last:                        # (the last round of a budget may report nothing)
                    active.remove(r)
        if moved:
            idle = 0
            continue
        idle += 1
        if idle > 2000000:                               # every mirror silent for ~1 s: wait for the queues themselves
            for r in active:
                r.eng.torch_stream.synchronize()
                r.eng.lib.ccqp_bpipe_poll(r.eng.ctx, state)
                if state[1] != r.k and state[0] < r.B:
                    raise RuntimeError("solve_batch_pipelined: the device never reported round %d" % r.k)
            idle = 0


class _Spec:
    """What a lock-step run needs to know about its B problems of size n, as arrays: matrix / right-hand-side
    addresses, leading dimensions, set descriptors (struct ccqp_seg records), tolerances, and the start vectors
    (None: zeros; a (B, n) tensor; or a list of optional n-vectors)."""
    __slots__ = ("B", "n", "A", "lda", "b", "seg", "tol", "x0", "keep", "step", "fallback")

    def __len__(self):
        return self.B

    def part(self, lo, hi):
        """Problems [lo, hi) as a spec of their own (views / slices, nothing copied)."""
        sp = _Spec()
        sp.B, sp.n = hi - lo, self.n
        sp.A, sp.lda, sp.b, sp.seg, sp.tol = self.A[lo:hi], self.lda[lo:hi], self.b[lo:hi], self.seg[lo:hi], self.tol[lo:hi]
        sp.x0 = None if self.x0 is None else self.x0[lo:hi]
        sp.keep = self.keep
        sp.step = None if getattr(self, "step", None) is None else self.step[lo:hi]
        sp.fallback = getattr(self, "fallback", False)
        return sp

    def stage_x0(self, plane):
        x0 = self.x0
        if x0 is None:
            plane.zero_()                               # x0 = 0 (solvers.py:125-126)
        elif isinstance(x0, torch.Tensor):
            plane.copy_(x0)
        else:
            plane.zero_()
            for p, v in enumerate(x0):
                if v is not None:
                    plane[p].copy_(v)


def _spec_from_items(eng, items, n):
    """items: tuples (solver, DenseOperator, b on the device, x0 on the device or None, set, ...)"""
    from . import _lib
    sp = _Spec()
    sp.B, sp.n = len(items), n
    sp.A = np.array([it[1].A.data_ptr() for it in items], dtype=np.uint64)
    sp.lda = np.array([it[1].A.stride(0) for it in items], dtype=np.int64)
    sp.b = np.array([it[2].data_ptr() for it in items], dtype=np.uint64)
    if all(hasattr(it[4], "_dev_bb_tail") for it in items):
        sp.seg = np.frombuffer(b"".join(bytes(it[4]._seg(eng, 0, n)) for it in items), dtype=np.dtype(_lib.Seg))
    else:                                               # segmented form: the sets travel as plans (_bsegs_run)
        sp.seg = np.zeros(len(items), dtype=np.dtype(_lib.Seg))
    sp.tol = np.array([it[0].desired_residual_tol for it in items], dtype=np.float64)
    sp.x0 = None if all(it[3] is None for it in items) else [it[3] for it in items]
    sp.keep = None
    # PGD (solvers.py:71-194): BBPGD's iteration with a constant step
    sp.step = np.array([it[0].step_size for it in items], dtype=np.float64) if hasattr(items[0][0], "step_size") else None
    sp.fallback = bool(getattr(items[0][0], "_fallback", False))       # BBPGDf
    return sp


def _bpipe_table(eng, spec, n, struct, nvec, partial_rows):
    """Workspace + descriptor table shared by the lock-step drivers: `nvec` (B, n) vector planes, zeroed slabs,
    partial-sum areas, counters; returns (vec, slabs, words, tab, base, plane, rows)."""
    from . import _lib
    from .solvers import _GD
    dev = eng.device
    B = len(spec)
    nblk = max(int(eng.lib.ccqp_bpipe_partials_doubles(n)) // 3, 1)
    vec = torch.empty((nvec, B, n), dtype=torch.float64, device=dev)
    slabs = torch.zeros((B, 64), dtype=torch.float64, device=dev)
    partials = torch.empty((B, partial_rows * nblk), dtype=torch.float64, device=dev)
    words = torch.zeros(B + 2, dtype=torch.int32, device=dev)           # per-problem counters + 2 batch words
    dt = np.dtype(struct)
    tab = np.zeros(B, dtype=dt)
    idx = np.arange(B, dtype=np.uint64)
    tab["A"], tab["lda"], tab["b"], tab["seg"] = spec.A, spec.lda, spec.b, spec.seg
    tab["slab"] = np.uint64(slabs.data_ptr()) + idx * np.uint64(8 * 64)
    tab["partials"] = np.uint64(partials.data_ptr()) + idx * np.uint64(8 * partial_rows * nblk)
    tab["counter"] = np.uint64(words.data_ptr()) + idx * np.uint64(4)
    tab["c"] = 1.0 / (3 * n * _GD)                                      # solvers.py:138
    tab["tol"] = spec.tol
    planes = [np.uint64(vec.data_ptr()) + np.uint64(j) * np.uint64(8 * n * B) + idx * np.uint64(8 * n)
              for j in range(nvec)]
    keep = (partials,)
    return vec, slabs, words, tab, planes, keep


def _bpipe_flags(tab, n, tune):
    # bit 0: 16-byte aligned rows and vectors; `tune` = the matvec's tuning word (never changes a result bit)
    # (bits 3 and 7 select solver variants -- PGD / anti-relaxation, BBPGDf -- and are not the caller's to set)
    return int(n % 2 == 0 and not np.any(tab["A"] % 16) and not np.any(tab["lda"] % 2)) | (int(tune) & 0xFFFF & ~(1 | 8 | 128))


def _bpipe_run(eng, spec, n, max_mv, ahead, tune=0, single_launch=None):
    """Lock-step (or single-launch) BBPGD over `spec`; returns (converged mask, index of the final iteration,
    (B, n) device tensor of final iterates, residuals, indices of problems the single-launch kernel gave up on)."""
    from . import _lib
    from .solvers import _EPS10, _GD
    lib, ctx, st, dev = eng.lib, eng.ctx, eng.stream, eng.device
    B = len(spec)
    fallback = bool(getattr(spec, "fallback", False))
    vec, slabs, words, tab, planes, keep = _bpipe_table(eng, spec, n, _lib.PProb, 9 if fallback else 7, 3)
    spec.stage_x0(vec[0])                                               # planes: x ring, g ring, y (, xmin, gmin)
    for j in range(3):
        tab["x"][:, j] = planes[j]
        tab["g"][:, j] = planes[3 + j]
    tab["y"] = planes[6]
    flags = _bpipe_flags(tab, n, tune)
    if getattr(spec, "step", None) is not None:         # PGD: constant steps, in slab slot 48; flag bit 3
        slabs[:, 48] = torch.from_numpy(np.ascontiguousarray(spec.step)).to(dev)
        flags |= 8
    if fallback:                                        # BBPGDf: xmin = gmin = copy(x0) (solvers.py:762-763), bit 7
        vec[7].copy_(vec[0])
        vec[8].copy_(vec[0])
        tab["xmin"], tab["gmin"] = planes[7], planes[8]
        flags |= 128
    dprobs = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(dev)
    dp, batch = dprobs.data_ptr(), words.data_ptr() + 4 * B
    check = _lib.check
    # iteration k with 1 + k >= max_mv is the last one and has no tail: the reference breaks before the
    # residual (:643-644)
    last = max(1, max_mv - 1) if np.isfinite(max_mv) else np.inf

    def launch(k):
        check(lib.ccqp_bpipe_iterate_f64(ctx, dp, B, n, k, _GD, float(_EPS10), flags, int(1 + k < max_mv), batch, st))

    if single_launch is None:
        single_launch = _single_launch_default("bbpgd", n, B)
    single_launch = bool(single_launch) and n <= 2048
    if single_launch:
        # small matrices: the whole solve of every problem in ONE launch, one workgroup per problem
        k = int(min(last, _device_budget(n)))
        check(lib.ccqp_bsolve_bbpgd_f64(ctx, dp, B, n, _GD, float(_EPS10), k, flags & 137, st))
        capped = True                                   # (at the caller's budget or at the device's: see `todo`)
    else:
        check(lib.ccqp_bpipe_start_f64(ctx, dp, B, n, _GD, flags, batch, st))
        k = _bpipe_poll_loop(eng, B, launch, last, ahead)
        capped = 1 + k >= max_mv
    h = slabs[:, 48:52].cpu().numpy()                   # blocking: the queue has drained
    done, iters, res2 = h[:, 1] != 0.0, h[:, 2].astype(np.int64), h[:, 3]
    if not capped and not done.all():
        raise RuntimeError("solve_batch_pipelined: mirror and slabs disagree")
    final = np.where(done, iters, k)
    # stopped by the DEVICE's budget, not the caller's: not a result -- these go to the host-driven loop
    todo = np.flatnonzero(~done) if k < last else np.zeros(0, dtype=np.int64)
    if B == 1:
        sols = vec[int(final[0] % 3)]
    else:
        sel = torch.from_numpy(final % 3).to(dev)
        sols = vec[0:3][sel, torch.arange(B, device=dev)]   # (B, n) gather of every problem's final iterate
    del keep
    return done, final, sols, np.sqrt(res2), todo


def _rows(sols, count):
    """The rows of a (B, n) result tensor as views, taken apart in one call (a batch of ONE hands over its (n,) row)."""
    return sols.unbind(0) if sols.dim() == 2 else (sols,) * count


def _bpipe_group(eng, items, n, max_mv, ahead, tune=0, single_launch=None):
    import contextlib
    import io
    done, final, sols, res, todo = _bpipe_run(eng, _spec_from_items(eng, items, n), n, max_mv, ahead, tune,
                                              single_launch)
    host = None
    todo = set(todo.tolist())
    pgd = hasattr(items[0][0], "step_size")
    # (per problem only attribute stores: the rows, counts and residuals are taken apart once, not indexed 8192 times)
    rows, finals, dones, ress = _rows(sols, len(items)), final.tolist(), done.tolist(), list(res)
    executed = 0
    for p, it in enumerate(items):
        s = it[0]
        if p in todo:                                    # beyond the single-launch kernel's own budget
            A, b, x0, P = it[6]
            with contextlib.redirect_stdout(io.StringIO()):
                stats["fallback_solves"] += 1
                _host_solve(s, A, b, x0, P)
            continue
        k = finals[p]
        mv = 1 + k
        if it[5]:
            s._solution = rows[p]
        else:
            if host is None:
                host = sols.cpu().numpy()
            s._solution = host[p]
        s._solution_converged = mv < max_mv
        s._solution_residual = ress[p]
        s._solution_num_matrix_vector_mults = mv
        # start-up: A x0 and -- BBPGD, unless the start already satisfies the tolerance -- A g0 (executed, not counted)
        s._executed_matrix_vector_mults = ex = mv if (pgd or (dones[p] and k == 0)) else 2 + k
        executed += ex
    eng.mv_executed += executed


def _psegs_table(plans, dev):
    """Device array of struct ccqp_psegs for one lock-step run: per problem its (possibly shared, read-only) block table
    and work lists and its OWN zeroed workspace slice.  Returns (table tensor, workspace tensor, max nwork, max nnorm)."""
    from . import _lib
    B = len(plans)
    wsd = np.array([pl.ws_doubles for pl in plans], dtype=np.int64)
    off = np.concatenate(([0], np.cumsum(wsd)[:-1]))
    ws = torch.zeros(int(wsd.sum()), dtype=torch.float64, device=dev)
    stab = np.zeros(B, dtype=np.dtype(_lib.PSegs))
    stab["table"] = [pl.table_ptr for pl in plans]
    stab["work"] = [pl.work_ptr for pl in plans]
    stab["normwork"] = [pl.normwork_ptr for pl in plans]
    stab["workspace"] = np.uint64(ws.data_ptr()) + off.astype(np.uint64) * np.uint64(8)
    stab["npartials"] = [pl.npartials for pl in plans]
    stab["nwork"] = [pl.nwork for pl in plans]
    stab["nnorm"] = [pl.nnorm for pl in plans]
    dsegs = torch.from_numpy(stab.view(np.uint8).reshape(-1)).to(dev)
    return dsegs, ws, max(int(stab["nwork"].max()), 1), int(stab["nnorm"].max())


def _bsegs_run(eng, spec, plans, n, max_mv, ahead, tune=0):
    """Lock-step BBPGD / PGD over `spec` with the feasible sets given as segment plans (one per problem, shared objects
    allowed): the segmented form of _bpipe_run (ccqp_bsegs.hip).  Returns (converged mask, index of the final
    iteration, (B, n) device tensor of final iterates, residuals)."""
    from . import _lib
    from .solvers import _EPS10, _GD
    lib, ctx, st, dev = eng.lib, eng.ctx, eng.stream, eng.device
    B = len(spec)
    fallback = bool(getattr(spec, "fallback", False))
    vec, slabs, words, tab, planes, keep = _bpipe_table(eng, spec, n, _lib.PProb, 9 if fallback else 7, 3)
    spec.stage_x0(vec[0])
    for j in range(3):
        tab["x"][:, j] = planes[j]
        tab["g"][:, j] = planes[3 + j]
    tab["y"] = planes[6]
    flags = _bpipe_flags(tab, n, tune)
    if getattr(spec, "step", None) is not None:         # PGD: constant steps, in slab slot 48; flag bit 3
        slabs[:, 48] = torch.from_numpy(np.ascontiguousarray(spec.step)).to(dev)
        flags |= 8
    if fallback:                                        # BBPGDf: xmin = gmin = copy(x0) (solvers.py:762-763), bit 7
        vec[7].copy_(vec[0])
        vec[8].copy_(vec[0])
        tab["xmin"], tab["gmin"] = planes[7], planes[8]
        flags |= 128
    dsegs, ws, max_nwork, max_nnorm = _psegs_table(plans, dev)
    dprobs = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(dev)
    dp, dsp, batch = dprobs.data_ptr(), dsegs.data_ptr(), words.data_ptr() + 4 * B
    check = _lib.check
    last = max(1, max_mv - 1) if np.isfinite(max_mv) else np.inf

    def launch(k):
        check(lib.ccqp_bsegs_iterate_f64(ctx, dp, dsp, B, n, k, max_nwork, max_nnorm, _GD, float(_EPS10), flags,
                                         int(1 + k < max_mv), batch, st))

    check(lib.ccqp_bsegs_start_f64(ctx, dp, dsp, B, n, max_nwork, max_nnorm, _GD, flags, batch, st))
    k = _bpipe_poll_loop(eng, B, launch, last, ahead)
    capped = 1 + k >= max_mv
    h = slabs[:, 48:52].cpu().numpy()                   # blocking: the queue has drained
    done, iters, res2 = h[:, 1] != 0.0, h[:, 2].astype(np.int64), h[:, 3]
    if not capped and not done.all():
        raise RuntimeError("solve_batch_pipelined: mirror and slabs disagree")
    final = np.where(done, iters, k)
    if B == 1:
        sols = vec[int(final[0] % 3)]
    else:
        sel = torch.from_numpy(final % 3).to(dev)
        sols = vec[0:3][sel, torch.arange(B, device=dev)]
    del keep, ws, dsegs
    return done, final, sols, np.sqrt(res2)


def _bsegs_group(eng, items, n, max_mv, ahead, tune=0):
    plans = [it[4]._segplan(eng, n) for it in items]
    done, final, sols, res = _bsegs_run(eng, _spec_from_items(eng, items, n), plans, n, max_mv, ahead, tune)
    host = None
    pgd = hasattr(items[0][0], "step_size")
    for p, it in enumerate(items):
        s = it[0]
        mv = 1 + int(final[p])
        if it[5]:
            s._solution = sols[p]
        else:
            if host is None:
                host = sols.cpu().numpy()
            s._solution = host[p]
        s._solution_converged = mv < max_mv
        s._solution_residual = res[p]
        s._solution_num_matrix_vector_mults = mv
        s._executed_matrix_vector_mults = mv if (pgd or (done[p] and final[p] == 0)) else 2 + int(final[p])
        eng.mv_executed += s._executed_matrix_vector_mults


_SPG_ROUNDS = 256         # rounds per call of the C driver (keeps the Python side interruptible)
# A/B switch (tools, tests): True = the C driver advances the batch as ONE group instead of two alternating halves
_SPG_ONE_GROUP = False    # (module attribute, set by the tool / test that wants it; the package reads no environment)
_capsule_pointer = None


def _draw_source(rng):
    """Address of the numpy bitgen_t record of a RandomState: the C driver calls its next_double -- the double
    `uniform` would consume -- exactly when the solver's own loop would."""
    global _capsule_pointer
    import ctypes
    if _capsule_pointer is None:
        f = ctypes.pythonapi.PyCapsule_GetPointer
        f.restype, f.argtypes = ctypes.c_void_p, [ctypes.py_object, ctypes.c_char_p]
        _capsule_pointer = f
    return _capsule_pointer(rng._bit_generator.capsule, b"BitGenerator")


def _bspg_run(eng, spec, solvers, n, max_mv, tune=0, plans=None, rngs=None, params=None):
    """Lock-step SPG over `spec` (ccqp_bspg.hip): the vector work of every problem on the device, the scalar part of
    each iteration on the host inside ONE C call per few hundred rounds.  The solver parameters and generators come
    from `solvers` (CCQPSolverSPG objects) or from `params` = (m, tau, sigma1, sigma2) (scalars or one per problem) +
    `rngs`.  Returns (host records, (B, n) x plane)."""
    import ctypes
    from . import _lib
    lib, ctx, st, dev = eng.lib, eng.ctx, eng.stream, eng.device
    B = len(spec)
    nblk = max(int(lib.ccqp_bpipe_partials_doubles(n)) // 3, 1)
    vec = torch.empty((4, B, n), dtype=torch.float64, device=dev)       # planes: x, g, d, Ad
    spec.stage_x0(vec[0])
    partials = torch.empty((B, 3 * nblk), dtype=torch.float64, device=dev)
    words = torch.zeros(B + 2, dtype=torch.int32, device=dev)
    xchg = torch.zeros(8 * B, dtype=torch.float64, device=dev)          # control words | dot products
    idx = np.arange(B, dtype=np.uint64)
    tab = np.zeros(B, dtype=np.dtype(_lib.PSpg))
    tab["A"], tab["lda"], tab["b"], tab["seg"] = spec.A, spec.lda, spec.b, spec.seg
    for j, name in enumerate(("x", "g", "d", "Ad")):
        tab[name] = np.uint64(vec.data_ptr()) + np.uint64(j) * np.uint64(8 * n * B) + idx * np.uint64(8 * n)
    tab["partials"] = np.uint64(partials.data_ptr()) + idx * np.uint64(8 * 3 * nblk)
    tab["counter"] = np.uint64(words.data_ptr()) + idx * np.uint64(4)
    flags = _bpipe_flags(tab, n, tune) | (8 if _SPG_ONE_GROUP else 0)
    dprobs = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(dev)
    if plans is not None:
        dsegs, seg_ws, max_nwork, max_nnorm = _psegs_table(plans, dev)
        dsp = dsegs.data_ptr()
    else:
        dsp, max_nwork, max_nnorm = None, 0, 0
    host = np.zeros(B, dtype=np.dtype(_lib.SpgHost))
    host["tol"] = spec.tol
    host["max_mv"] = max_mv
    if params is None:
        params = ([s.m for s in solvers], [s.t for s in solvers], [s.sigma1 for s in solvers],
                  [s.sigma2 for s in solvers])
    host["m"], host["tau"], host["sigma1"], host["sigma2"] = params
    qstride = int(host["m"].max())
    queue = np.zeros((B, qstride))
    gens = list(rngs if rngs is not None else [s.rng for s in solvers])
    sources = np.array([_draw_source(r) for r in gens], dtype=np.uint64)
    more = ctypes.c_int(1)
    start = 1
    torch.cuda.current_stream(dev).synchronize()        # x0 staged, tables copied: the C driver runs on `st`
    # The ctypes call releases the GIL and the C driver draws from the generators' bitgen_t records directly: hold each
    # generator's own lock meanwhile (numpy's methods take it before they draw), so another Python thread using the
    # same generator -- the process-wide np.random one of a batch-of-one `pipeline=True` solve above all -- waits
    # instead of racing on the MT19937 state.  (Distinct locks, taken in one global order.)
    import contextlib
    locks = sorted({id(g._bit_generator.lock): g._bit_generator.lock for g in gens}.items())
    with contextlib.ExitStack() as held:
        for _, lk in locks:
            held.enter_context(lk)
        while more.value:
            _lib.check(lib.ccqp_bspg_run_f64(ctx, dprobs.data_ptr(), dsp, host.ctypes.data, queue.ctypes.data, qstride,
                                             sources.ctypes.data, B, n, max_nwork, max_nnorm, flags, start, _SPG_ROUNDS,
                                             xchg.data_ptr(), xchg.data_ptr() + 32 * B, words.data_ptr() + 4 * B,
                                             ctypes.byref(more), st))
            start = 0
    return host, vec[0]


def _bspg_one(eng, solver, op, b, x0, P, n, rng):
    """ONE dense SPG problem on an elementwise set, device-resident: _bspg_run for a batch of one without its per-batch
    host work (a cached workspace per size, staging in ONE launch on the engine's stream -- ccqp_bspg_stage_one_f64 -- and
    no stream synchronisation before the C driver, which runs on the same stream).  Returns (host record, x)."""
    import contextlib
    import ctypes
    from . import _lib
    lib, ctx, st, dev = eng.lib, eng.ctx, eng.stream, eng.device
    A = op.A
    aligned = n % 2 == 0 and A.data_ptr() % 16 == 0 and A.stride(0) % 2 == 0 and b.data_ptr() % 16 == 0
    flags = int(aligned) | (8 if _SPG_ONE_GROUP else 0)
    cache = eng.__dict__.setdefault("_bspg_one_ws", {})
    ws = cache.get(n)
    if ws is None:
        if len(cache) > 8:
            cache.clear()
        ws = cache[n] = _OneWs()
        nblk = max(int(lib.ccqp_bpipe_partials_doubles(n)) // 3, 1)
        stride = (n + 1) // 2 * 2
        ws.vec = torch.empty((4, stride), dtype=torch.float64, device=dev)      # planes: x, g, d, Ad
        ws.partials = torch.empty(3 * nblk, dtype=torch.float64, device=dev)
        ws.words = torch.zeros(4, dtype=torch.int32, device=dev)
        ws.slab = torch.zeros(8, dtype=torch.float64, device=dev)               # (the exchange words: control | dots)
        ws.ddesc = torch.zeros(ctypes.sizeof(_lib.PSpg), dtype=torch.uint8, device=dev)
        d = ws.desc = _lib.PSpg()
        base = ws.vec.data_ptr()
        d.x, d.g, d.d, d.Ad = (base + 8 * stride * j for j in range(4))
        d.partials, d.counter = ws.partials.data_ptr(), ws.words.data_ptr()
    d = ws.desc
    d.A, d.lda, d.b = A.data_ptr(), A.stride(0), b.data_ptr()
    d.seg = P._seg(eng, 0, n)
    _lib.check(lib.ccqp_bspg_stage_one_f64(ctx, ctypes.byref(d), ws.ddesc.data_ptr(),
                                           None if x0 is None else x0.data_ptr(), n, ws.words.data_ptr(),
                                           ws.slab.data_ptr(), st))
    host = np.zeros(1, dtype=np.dtype(_lib.SpgHost))
    host["tol"] = float(solver.desired_residual_tol)
    host["max_mv"] = solver.max_matrix_vector_multiplications
    host["m"], host["tau"], host["sigma1"], host["sigma2"] = solver.m, solver.t, solver.sigma1, solver.sigma2
    qstride = int(solver.m)
    queue = np.zeros((1, qstride))
    sources = np.array([_draw_source(rng)], dtype=np.uint64)
    more = ctypes.c_int(1)
    start = 1
    # (the C driver draws from the generator's bitgen_t record directly, outside the GIL: hold the generator's own lock)
    with contextlib.ExitStack() as held:
        held.enter_context(rng._bit_generator.lock)
        while more.value:
            _lib.check(lib.ccqp_bspg_run_f64(ctx, ws.ddesc.data_ptr(), None, host.ctypes.data, queue.ctypes.data, qstride,
                                             sources.ctypes.data, 1, n, 0, 0, flags, start, _SPG_ROUNDS,
                                             ws.slab.data_ptr(), ws.slab.data_ptr() + 32, ws.words.data_ptr() + 4,
                                             ctypes.byref(more), st))
            start = 0
    # (a VIEW of the workspace: the caller's export copies it on this stream before the next solve can stage over it)
    return host, ws.vec[0, :n]


def _bspg_group(eng, items, n, max_mv, ahead, tune=0, single_launch=None, segmented=False):
    import contextlib
    import io
    plans = [it[4]._segplan(eng, n) for it in items] if segmented else None
    solvers = [it[0] for it in items]
    host, sols = _bspg_run(eng, _spec_from_items(eng, items, n), solvers, n, max_mv, tune, plans)
    hostx = None
    for p, it in enumerate(items):
        s = it[0]
        if host["status"][p] == 3:
            stats["fallback_solves"] += 1
            if host["used"][p]:
                # the reference's draw raises here, after `used` successful ones (the generator is where it would be):
                # numpy's own OverflowError('Range exceeds valid bounds'), solvers.py:959
                np.random.RandomState(0).uniform(low=0.0, high=np.nan)
            # nothing drawn yet (the first draw would raise, or the budget ends before the first d.d:
            # UnboundLocalError): the host-driven loop reproduces the reference's exception
            A, b, x0, P = it[6]
            with contextlib.redirect_stdout(io.StringIO()):
                _host_solve(s, A, b, x0, P)
            continue
        mv = int(host["mv"][p])
        if it[5]:
            s._solution = sols[p]
        else:
            if hostx is None:
                hostx = sols.cpu().numpy()
            s._solution = hostx[p]
        s._solution_converged = mv < max_mv
        s._solution_residual = np.sqrt(host["dd"][p])
        s._solution_num_matrix_vector_mults = mv
        # (the matvec that ends a budget is counted but its result is never used: not executed here)
        s._executed_matrix_vector_mults = mv - (1 if host["status"][p] == 2 else 0)
        eng.mv_executed += s._executed_matrix_vector_mults


# rounds a device-resident SINGLE solve is queued beyond the device's last report: the host needs ~8 us per round, the
# device 15 ... 90, and every round queued past convergence is three to seven no-op launches -- 2 / 3 are 3-5 % ahead of
# 6 / 8 on APGD, BBPGD and MPRGP solves of n = 256 ... 4096 (gpurun_out r04M, `tools/bench_single_mid.py --ahead`)
_ONE_AHEAD = 3
_MPRGP_FUSED_ROUNDS = True  # (module attribute for the A/B of tools / tests: False = one round per launch everywhere)
last_timing = {}            # host-side split of the last _bapgd_run: staging / queueing the rounds / waiting for the queue
_APGD_FUSED_ROUNDS = True   # (module attribute for the A/B of tools / tests: False = two rounds per accepted iteration)
# a batch of ONE problem of n >= 4096: the whole solve in one cooperative launch whose rounds are separated by grid barriers
# (ccqp_bsolve_apgd_grid_f64).  OFF: measured on MI355X a grid barrier over 256 workgroups on 8 XCDs costs ~12 us (the
# agent-scope release / acquire of every workgroup), a kernel boundary ~3.5 us -- the rounds are faster (n = 4096: 69
# against 45 us per round, profiles/r04_single_mid.jsonl).  Kept selectable (same bits and counts, tested).
_APGD_GRID_SOLVE = False


def _sharing_form(eng, count, n, flags, mprgp=False):
    """Does a run that shares passes (lock-step APGD / MPRGP with their look-ahead) take the persistent matvec?"""
    import ctypes
    buf = ctypes.create_string_buffer(64)
    q = (flags & 0xFFFF) | (1 << 16) | ((1 << 18) if mprgp else 0)
    from . import _lib
    _lib.check(eng.lib.ccqp_bpipe_matvec_form(eng.ctx, count, n, q, buf, 64))
    return buf.value.decode() == "bp_gemv_pers_kernel"


def _bapgd_run(eng, spec, n, max_mv, ahead, tune=0, single_launch=None, ar=False, plans=None):
    """Lock-step APGD over `spec`; returns (converged mask, (B, n) device tensor of final iterates, residuals,
    matvec counts) -- rows of problems that did not converge inside the budget are meaningless.  `plans` (one segment
    plan per problem): the feasible sets are not elementwise -- the segmented rounds of ccqp_bsegs.hip, no
    single-launch form."""
    from . import _lib
    from .solvers import _GD
    lib, ctx, st, dev = eng.lib, eng.ctx, eng.stream, eng.device
    B = len(spec)
    t_enter = time.perf_counter()
    # planes: X0 X1 Y0 Y1 gk g1 Ayk Ax1 xhat ga Aya (the last two: products computed ahead, ccqp_bpipe.hip)
    vec, slabs, words, tab, planes, keep = _bpipe_table(eng, spec, n, _lib.PApgd, 11, 8)
    spec.stage_x0(vec[0])
    vec[8].fill_(1.0)                                                   # xhat = ones (solvers.py:432)
    vec[2].copy_(vec[0])                                                # yk = x0 (solvers.py:257-258)
    tab["X"][:, 0], tab["X"][:, 1] = planes[0], planes[1]
    tab["Y"][:, 0], tab["Y"][:, 1] = planes[2], planes[3]
    tab["gk"], tab["g1"], tab["Ayk"], tab["Ax1"], tab["xhat"] = planes[4], planes[5], planes[6], planes[7], planes[8]
    tab["ga"], tab["Aya"] = planes[9], planes[10]
    # bit 3: the anti-relaxation variant; bit 17: every problem carries the two look-ahead buffers (ga / Aya above)
    flags = _bpipe_flags(tab, n, tune) | (8 if ar else 0) | _lib.LOCKSTEP_LOOKAHEAD
    # bit 18, elementwise sets: FUSED ROUNDS -- an accepting round also takes A.y_{k+1} + b over and forms the next trial
    # point (one round per accepted iteration instead of two); the device then enforces the budget itself: max_mv - 1 in
    # slab slot 48 (0 = none), a problem that reaches it stops with slot 49 = 2 and goes to the host-driven loop
    if plans is None and _APGD_FUSED_ROUNDS:
        flags |= _lib.LOCKSTEP_FUSED_ROUNDS
        slabs[:, 48] = float(max_mv - 1) if np.isfinite(max_mv) else 0.0
    # ONE problem whose matvec is the persistent multi-vector form (n >= 4096 by default): the whole solve in ONE cooperative
    # launch on the whole chip, the fused rounds separated by grid barriers (ccqp_bsolve_apgd_grid_f64)
    grid = B == 1 and plans is None and _APGD_FUSED_ROUNDS and _APGD_GRID_SOLVE and single_launch is not True \
        and (flags & 1) and _sharing_form(eng, 1, n, flags) and (not np.isfinite(max_mv) or max_mv >= 4)
    dprobs = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(dev)
    dp, batch = dprobs.data_ptr(), words.data_ptr() + 4 * B
    check = _lib.check
    # round r performs matvec number 1 + r; the rounds stop one short of the budget, where the reference's exit
    # path takes over (:277-278, :288-289): such problems go to the host-driven loop
    last = max(0, max_mv - 2) if np.isfinite(max_mv) else np.inf

    if plans is not None:
        dsegs, seg_ws, max_nwork, max_nnorm = _psegs_table(plans, dev)
        dsp = dsegs.data_ptr()

        def launch(r):
            check(lib.ccqp_bsegs_apgd_round_f64(ctx, dp, dsp, B, n, r, max_nwork, max_nnorm, _GD, flags, batch, st))
    else:
        def launch(r):
            check(lib.ccqp_bapgd_round_f64(ctx, dp, B, n, r, _GD, flags, batch, st))

    if single_launch is None:
        single_launch = _single_launch_default("apgd", n, B)
    single_launch = bool(single_launch) and n <= 2048 and plans is None
    t_staged = time.perf_counter()
    if grid and not single_launch:
        rc = lib.ccqp_bsolve_apgd_grid_f64(ctx, dp, n, _GD, flags & 9, int(last) if np.isfinite(last) else 10 ** 7, batch, st)
        if rc == _lib.ERR_UNSUPPORTED:
            grid = False                                 # (no cooperative launch here: the rounds below)
        else:
            check(rc)
            stats["grid_solves"] += 1
    else:
        grid = False
    if grid:
        pass                                             # (one launch; the blocking read below waits for it)
    elif single_launch:
        # small matrices: the whole solve of every problem in ONE launch, one workgroup per problem
        check(lib.ccqp_bsolve_apgd_f64(ctx, dp, B, n, _GD, float(min(max_mv - 1, _device_budget(n))), flags & 9, st))
    else:
        check(lib.ccqp_bapgd_start_f64(ctx, dp, B, n, flags, batch, st))
        _bpipe_poll_loop(eng, B, launch, last, ahead)
    t_queued = time.perf_counter()
    h = slabs[:, 32:52].cpu().numpy()                   # blocking: the queue has drained
    last_timing.update(staging_s=t_staged - t_enter, rounds_s=t_queued - t_staged, drain_s=time.perf_counter() - t_queued)
    # (slot 49: 1 = converged; 2 = a fused-rounds problem stopped one short of its budget -> not done here)
    done, its, mvs, res2 = h[:, 17] == 1.0, h[:, 8].astype(np.int64), h[:, 18].astype(np.int64), h[:, 19]
    if grid and not done.all() and int(words[B + 1].item()) != 0:
        stats["grid_gave_up"] += 1                       # (the grid barrier gave up: the caller's host-driven loop takes over)
    elif not single_launch and not np.isfinite(max_mv) and not done.all():
        raise RuntimeError("solve_batch_pipelined: mirror and slabs disagree")
    # the solution: x_{k+1} = X[(accepted iterations + 1) % 2]; anti-relaxation: xhat (plane 8) unless the converging
    # iterate was itself the best one so far (slot 43)
    pick = (its + 1) % 2
    if ar:
        pick = np.where(h[:, 11] != 0.0, 8, pick)
    if B == 1:
        sols = vec[int(pick[0])]
    else:
        sols = vec[torch.from_numpy(pick).to(dev), torch.arange(B, device=dev)]
    del keep
    return done, sols, np.sqrt(res2), mvs


class _OneWs:
    """Workspace of the device-resident single APGD solve of one size on one engine, kept between solves: the eleven
    vector planes, slab, partial sums, counters, the device copy of the descriptor and its host original."""
    __slots__ = ("vec", "slab", "partials", "words", "ddesc", "desc", "host", "planes")


def _bapgd_one(eng, solver, op, b, x0, P, n, ahead=None):
    """ONE dense APGD / anti-relaxation APGD problem on an elementwise set, device-resident (the form of
    CCQPSolverAPGD's default pipeline): staging in ONE launch (ccqp_bapgd_stage_one_f64: no per-solve allocation, no
    host-paced torch operation, the descriptor by value); then n <= 32 the whole solve in one launch (one workgroup),
    beyond the FUSED lock-step rounds -- one round per accepted iteration: below n = 4096 both products from one launch of
    the VALU kernel, from 4096 on the persistent three-vector matvec; then the decision, then the update -- all queued `ahead` beyond the device's last report; the result words through the pinned mirror.  Returns (done, x, residual, matvecs); done False: the budget was reached first (the host-driven loop's exits)."""
    import ctypes
    from . import _lib
    from .solvers import _GD
    lib, ctx, st, dev = eng.lib, eng.ctx, eng.stream, eng.device
    A = op.A
    aligned = n % 2 == 0 and A.data_ptr() % 16 == 0 and A.stride(0) % 2 == 0 and b.data_ptr() % 16 == 0
    ar = bool(solver._anti_relaxation)
    max_mv = solver.max_matrix_vector_multiplications
    # fused rounds (one round per accepted iteration, the budget enforced on the device): wherever the library has a form
    # for them -- the persistent three-vector matvec from n = 4096 on, below it both products from one VALU launch
    fused = _lib.LOCKSTEP_LOOKAHEAD | _lib.LOCKSTEP_FUSED_ROUNDS
    flags = int(aligned) | (8 if ar else 0)
    sharing = _APGD_FUSED_ROUNDS and (not np.isfinite(max_mv) or max_mv >= 4) \
        and lib.ccqp_bapgd_fused_form(ctx, 1, n, flags | fused) != 0
    if sharing:
        flags |= fused
    cache = eng.__dict__.setdefault("_apgd_one_ws", {})
    ws = cache.get(n)
    if ws is None:
        if len(cache) > 8:
            cache.clear()
        ws = cache[n] = _OneWs()
        nblk = max(int(lib.ccqp_bpipe_partials_doubles(n)) // 3, 1)
        stride = (n + 1) // 2 * 2                        # (planes 16 bytes apart for an odd n too)
        ws.vec = torch.empty((11, stride), dtype=torch.float64, device=dev)
        ws.slab = torch.zeros(64, dtype=torch.float64, device=dev)
        ws.partials = torch.empty(8 * nblk, dtype=torch.float64, device=dev)
        ws.words = torch.zeros(4, dtype=torch.int32, device=dev)
        ws.ddesc = torch.zeros(ctypes.sizeof(_lib.PApgd), dtype=torch.uint8, device=dev)
        d = ws.desc = _lib.PApgd()
        base = ws.vec.data_ptr()
        plane = [base + 8 * stride * j for j in range(11)]
        # planes: X0 X1 Y0 Y1 gk g1 Ayk Ax1 xhat ga Aya
        d.X[0], d.X[1], d.Y[0], d.Y[1] = plane[0], plane[1], plane[2], plane[3]
        d.gk, d.g1, d.Ayk, d.Ax1, d.xhat, d.ga, d.Aya = plane[4:11]
        d.slab, d.partials, d.counter = ws.slab.data_ptr(), ws.partials.data_ptr(), ws.words.data_ptr()
        d.c = 1.0 / (3 * n * _GD)                        # solvers.py:138
        ws.host = (ctypes.c_double * 20)()
    d = ws.desc
    d.A, d.lda, d.b = A.data_ptr(), A.stride(0), b.data_ptr()
    d.seg = P._seg(eng, 0, n)
    d.tol = float(solver.desired_residual_tol)
    limit = float(max_mv - 1) if (sharing and np.isfinite(max_mv)) else 0.0
    check = _lib.check
    dp, batch = ws.ddesc.data_ptr(), ws.words.data_ptr() + 4
    t0 = time.perf_counter()
    check(lib.ccqp_bapgd_stage_one_f64(ctx, ctypes.byref(d), dp, None if x0 is None else x0.data_ptr(), n, limit,
                                       ws.words.data_ptr(), st))
    last = max(0, max_mv - 2) if np.isfinite(max_mv) else np.inf
    grid = sharing and _APGD_GRID_SOLVE and aligned and _sharing_form(eng, 1, n, 1)
    # (ONE problem: the one-workgroup solve only for tiny matrices -- at n = 128 it takes 1.2 ms, the rounds 0.5, at n = 64
    # four problems 0.93 against 0.71 ms; profiles/r04_single_launch_vs_rounds.jsonl -- it is a form for MANY problems)
    single = n <= 32
    if grid:
        rc = lib.ccqp_bsolve_apgd_grid_f64(ctx, dp, n, _GD, flags & 9, int(last) if np.isfinite(last) else 10 ** 7, batch, st)
        if rc == _lib.ERR_UNSUPPORTED:
            grid = False
        else:
            check(rc)
            stats["grid_solves"] += 1
    t1 = time.perf_counter()
    if grid:
        pass
    elif single:
        check(lib.ccqp_bsolve_apgd_f64(ctx, dp, 1, n, _GD, float(min(max_mv - 1, _device_budget(n))), flags & 9, st))
    else:
        check(lib.ccqp_bapgd_start_f64(ctx, dp, 1, n, flags, batch, st))

        def launch(r):
            check(lib.ccqp_bapgd_round_f64(ctx, dp, 1, n, r, _GD, flags, batch, st))
        _bpipe_poll_loop(eng, 1, launch, last, ahead or _ONE_AHEAD)
    t2 = time.perf_counter()
    h = ws.host
    check(lib.ccqp_read_f64(ctx, ws.slab.data_ptr() + 8 * 32, 20, h, st))      # blocking: the queue has drained
    last_timing.update(staging_s=t1 - t0, rounds_s=t2 - t1, drain_s=time.perf_counter() - t2)
    done = h[17] == 1.0
    if not done:
        if grid and int(ws.words[2].item()) != 0:
            stats["grid_gave_up"] += 1
        return False, None, None, None
    pick = 8 if (ar and h[11] != 0.0) else (int(h[8]) + 1) % 2
    # (a VIEW of the workspace: the caller's export copies it on this stream before the next solve can stage over it)
    return True, ws.vec[pick, :n], np.sqrt(np.float64(h[19])), int(h[18])


def _bpipe_one(eng, solver, op, b, x0, P, n, ahead=None):
    """ONE dense PGD / BBPGDf (/ BBPGD) problem on an elementwise set, device-resident: _bpipe_run for a batch of one
    without its per-batch host work -- a cached workspace per size, staging in ONE launch (ccqp_bpipe_stage_one_f64: the
    descriptor by value, no host-paced torch operation), the result words through the pinned mirror.  n <= 64: the whole
    solve in one launch (one workgroup), beyond the lock-step iterations queued `ahead` beyond the device's last report.
    Returns (done, final iteration, x, residual, gave_up): gave_up = the DEVICE's own budget ended the run."""
    import ctypes
    from . import _lib
    from .solvers import _EPS10, _GD
    lib, ctx, st, dev = eng.lib, eng.ctx, eng.stream, eng.device
    A = op.A
    aligned = n % 2 == 0 and A.data_ptr() % 16 == 0 and A.stride(0) % 2 == 0 and b.data_ptr() % 16 == 0
    pgd, fallback = hasattr(solver, "step_size"), bool(getattr(solver, "_fallback", False))
    flags = int(aligned) | (8 if pgd else 0) | (128 if fallback else 0)
    max_mv = solver.max_matrix_vector_multiplications
    cache = eng.__dict__.setdefault("_bpipe_one_ws", {})
    ws = cache.get(n)
    if ws is None:
        if len(cache) > 8:
            cache.clear()
        ws = cache[n] = _OneWs()
        nblk = max(int(lib.ccqp_bpipe_partials_doubles(n)) // 3, 1)
        stride = (n + 1) // 2 * 2                        # (planes 16 bytes apart for an odd n too)
        ws.vec = torch.empty((9, stride), dtype=torch.float64, device=dev)
        ws.slab = torch.zeros(64, dtype=torch.float64, device=dev)
        ws.partials = torch.empty(3 * nblk, dtype=torch.float64, device=dev)
        ws.words = torch.zeros(4, dtype=torch.int32, device=dev)
        ws.ddesc = torch.zeros(ctypes.sizeof(_lib.PProb), dtype=torch.uint8, device=dev)
        d = ws.desc = _lib.PProb()
        base = ws.vec.data_ptr()
        plane = [base + 8 * stride * j for j in range(9)]
        # planes: x ring, g ring, y, xmin, gmin
        for j in range(3):
            d.x[j], d.g[j] = plane[j], plane[3 + j]
        d.y = plane[6]
        d.slab, d.partials, d.counter = ws.slab.data_ptr(), ws.partials.data_ptr(), ws.words.data_ptr()
        d.c = 1.0 / (3 * n * _GD)                        # solvers.py:138
        ws.host = (ctypes.c_double * 4)()
        ws.planes = plane
    d = ws.desc
    d.A, d.lda, d.b = A.data_ptr(), A.stride(0), b.data_ptr()
    d.seg = P._seg(eng, 0, n)
    d.tol = float(solver.desired_residual_tol)
    d.xmin, d.gmin = (ws.planes[7], ws.planes[8]) if fallback else (None, None)
    check = _lib.check
    dp, batch = ws.ddesc.data_ptr(), ws.words.data_ptr() + 4
    check(lib.ccqp_bpipe_stage_one_f64(ctx, ctypes.byref(d), dp, None if x0 is None else x0.data_ptr(), n,
                                       float(solver.step_size) if pgd else 0.0, ws.words.data_ptr(), st))
    # iteration k with 1 + k >= max_mv is the last one and has no tail: the reference breaks before the residual (:643-644)
    last = max(1, max_mv - 1) if np.isfinite(max_mv) else np.inf
    if n <= 64:
        k = int(min(last, _device_budget(n)))
        check(lib.ccqp_bsolve_bbpgd_f64(ctx, dp, 1, n, _GD, float(_EPS10), k, flags & 137, st))
        capped = True
    else:
        def launch(k):
            check(lib.ccqp_bpipe_iterate_f64(ctx, dp, 1, n, k, _GD, float(_EPS10), flags, int(1 + k < max_mv), batch, st))
        check(lib.ccqp_bpipe_start_f64(ctx, dp, 1, n, _GD, flags, batch, st))
        k = _bpipe_poll_loop(eng, 1, launch, last, ahead or _ONE_AHEAD)
        capped = 1 + k >= max_mv
    h = ws.host
    check(lib.ccqp_read_f64(ctx, ws.slab.data_ptr() + 8 * 48, 4, h, st))       # blocking: the queue has drained
    done, iters, res2 = h[1] != 0.0, int(h[2]), h[3]
    if not capped and not done:
        raise RuntimeError("device-resident solve: mirror and slab disagree")
    final = iters if done else k
    # (a VIEW of the workspace: the caller's export copies it on this stream before the next solve can stage over it)
    return done, final, ws.vec[final % 3, :n], np.sqrt(np.float64(res2)), (not done) and k < last


def _bapgd_group(eng, items, n, max_mv, ahead, tune=0, single_launch=None, segmented=False):
    import contextlib
    import io
    plans = [it[4]._segplan(eng, n) for it in items] if segmented else None
    done, sols, res, mvs = _bapgd_run(eng, _spec_from_items(eng, items, n), n, max_mv, ahead, tune, single_launch,
                                      ar=items[0][0]._anti_relaxation, plans=plans)
    host = None
    rows, dones, mvl, ress = _rows(sols, len(items)), done.tolist(), mvs.tolist(), list(res)
    executed = 0
    for p, it in enumerate(items):
        s = it[0]
        if not dones[p]:                                 # budget reached: the reference's own exit path
            A, b, x0, P = it[6]
            with contextlib.redirect_stdout(io.StringIO()):
                stats["fallback_solves"] += 1
                _host_solve(s, A, b, x0, P)
            continue
        if it[5]:
            s._solution = rows[p]
        else:
            if host is None:
                host = sols.cpu().numpy()
            s._solution = host[p]
        mv = mvl[p]
        s._solution_converged = mv < max_mv
        s._solution_residual = ress[p]
        s._solution_num_matrix_vector_mults = mv
        s._executed_matrix_vector_mults = mv
        executed += mv
    eng.mv_executed += executed


class _MprgpRun:
    """One lock-step MPRGP run over `spec` on engine `eng` (ccqp_bmprgp_*): construction allocates the workspace, stages
    x0 and queues the start-up on the engine's stream; `launch(r)` queues round r; `finish()` waits for the queue and
    returns (status, (B, n) final iterates, residuals, reported matvecs, executed matvecs)."""

    def __init__(self, eng, spec, n, max_mv, tune=0, single_launch=None, plans=None):
        """plans: one segment plan per problem (solution_spaces._SegPlan, shared objects allowed) for feasible sets that
        are not elementwise -- the segmented form (ccqp_bsegs_mprgp_*); None: elementwise sets from spec.seg."""
        from . import _lib
        from .solvers import _EPS10, _GD
        self.eng, self.n, self.max_mv = eng, n, max_mv
        lib, ctx, st, dev = eng.lib, eng.ctx, eng.stream, eng.device
        self.B = B = len(spec)
        names = ("X0", "X1", "G0", "G1", "p", "Ap", "psi", "w", "nv", "txk", "dx", "Adx", "ytmp")
        vec, slabs, words, tab, planes, keep = _bpipe_table(eng, spec, n, _lib.PMprgp, len(names),
                                                             int(lib.ccqp_bmprgp_partial_rows()))
        spec.stage_x0(vec[1] if plans is None else vec[10])             # x0 staged in X[1] (segmented form: in dx)
        tab["X"][:, 0], tab["X"][:, 1] = planes[0], planes[1]
        tab["G"][:, 0], tab["G"][:, 1] = planes[2], planes[3]
        for j, name in enumerate(names[4:]):
            tab[name] = planes[4 + j]
        flags = _bpipe_flags(tab, n, tune)
        if plans is None and _MPRGP_FUSED_ROUNDS:
            # bit 18: FUSED ROUNDS where the library has a form for them (small batches of one-chunk problems: a launch
            # passes through the rounds whose products are there); the device then enforces the budget itself: max_mv - 1 in
            # slab slot 55 (0 = none), a problem that reaches it ends with slot 49 = 4 and goes to the host-driven loop
            flags |= _lib.LOCKSTEP_FUSED_ROUNDS
            slabs[:, 55] = float(max_mv - 1) if np.isfinite(max_mv) else 0.0
        dprobs = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(dev)
        dp, batch = dprobs.data_ptr(), words.data_ptr() + 4 * B
        check = _lib.check
        # the count stands at 2 after the start-up and grows by at most one per round; the rounds stop where a problem
        # could reach the budget, whose exit paths (:1087, :1102, :1154, :1176) belong to the host-driven loop
        self.last = max(0, max_mv - 3) if np.isfinite(max_mv) else np.inf
        gd, eps10 = _GD, float(_EPS10)

        def launch(r):
            check(lib.ccqp_bmprgp_round_f64(ctx, dp, B, n, r, gd, eps10, flags, batch, st))

        self.launch = launch
        if plans is not None:
            dsegs, seg_ws, max_nwork, max_nnorm = _psegs_table(plans, dev)
            dsp = dsegs.data_ptr()

            def launch_segs(r):
                check(lib.ccqp_bsegs_mprgp_round_f64(ctx, dp, dsp, B, n, r, max_nwork, max_nnorm, gd, eps10, flags,
                                                     batch, st))

            self.launch = launch_segs
            self.single_launch = False
            check(lib.ccqp_bsegs_mprgp_start_f64(ctx, dp, dsp, B, n, max_nwork, max_nnorm, gd, flags, batch, st))
            self.vec, self.slabs, self.keep = vec, slabs, (keep, words, dprobs, dsegs, seg_ws)
            return
        if single_launch is None:
            single_launch = _single_launch_default("mprgp", n, B)
        self.single_launch = single_launch = bool(single_launch) and n <= 2048
        if single_launch:
            # small matrices: the whole solve of every problem in ONE launch, one workgroup per problem
            check(lib.ccqp_bsolve_mprgp_f64(ctx, dp, B, n, gd, eps10, float(min(max_mv - 1, _device_budget(n))),
                                            flags & 1, st))
            self.last = 0
        else:
            check(lib.ccqp_bmprgp_start_f64(ctx, dp, B, n, gd, flags, batch, st))
            # what this run's batched matvec launches as (for tests / tools: config 5's share must be the >= 32-problem
            # shape of the three-vector persistent launch)
            import ctypes
            buf = ctypes.create_string_buffer(64)
            check(lib.ccqp_bpipe_matvec_form(ctx, B, n, (flags & 0xFFFF) | (1 << 16) | (1 << 18), buf, 64))
            stats["last_mprgp_run"] = {"problems": B, "n": n, "matvec": buf.value.decode(),
                                       "triple_shape": int(lib.ccqp_bmv3_shape(ctx, B)),
                                       "looks_ahead": buf.value.decode() == "bp_gemv_pers_kernel"
                                       and not eng.get_option("MPRGP_LOOKAHEAD")}
        self.vec, self.slabs, self.keep = vec, slabs, (keep, words, dprobs)

    def finish(self):
        vec, slabs, B, dev = self.vec, self.slabs, self.B, self.eng.device
        h = slabs[:, 32:56].cpu().numpy()               # blocking: the queue has drained
        status, its = h[:, 17], h[:, 8].astype(np.int64)
        mvs, res2, execd = h[:, 18].astype(np.int64), h[:, 19], h[:, 20].astype(np.int64)
        if not self.single_launch and not np.isfinite(self.max_mv) and not (status != 0.0).all():
            raise RuntimeError("solve_batch_pipelined: mirror and slabs disagree")
        if B == 1:
            sols = vec[int((its[0] + 1) % 2)]
        else:
            sel = torch.from_numpy((its + 1) % 2).to(dev)
            sols = vec[0:2][sel, torch.arange(B, device=dev)]
        self.keep = None
        return status, sols, np.sqrt(res2), mvs, execd


def _bmprgp_one(eng, solver, op, b, x0, P, n, ahead=None, device_cap=None):
    """ONE dense MPRGP problem on an elementwise set with array bounds, device-resident: _bmprgp_run for a batch of one
    without its per-batch host work (a cached workspace per size, staging in ONE launch -- ccqp_bmprgp_stage_one_f64 --,
    the result words through the pinned mirror); n <= 32: one launch (one workgroup), beyond the lock-step rounds queued
    `ahead` beyond the device's last report.  Returns (status, x, residual, reported matvecs, executed matvecs); status 1 =
    converged, anything else = not finished here.  `device_cap`: the device gives up after this many matvecs (the caller
    then runs the host-driven loop: a solve that STALLS short of its tolerance -- MPRGP does at tol <= 1e-8 on well
    conditioned boxes, SURVEY B.2 -- must not spend its whole budget here first)."""
    import ctypes
    from . import _lib
    from .solvers import _EPS10, _GD
    lib, ctx, st, dev = eng.lib, eng.ctx, eng.stream, eng.device
    A = op.A
    aligned = n % 2 == 0 and A.data_ptr() % 16 == 0 and A.stride(0) % 2 == 0 and b.data_ptr() % 16 == 0
    flags = int(aligned) | (_lib.LOCKSTEP_FUSED_ROUNDS if _MPRGP_FUSED_ROUNDS else 0)
    max_mv = solver.max_matrix_vector_multiplications
    if device_cap is not None:
        max_mv = min(max_mv, device_cap)
    cache = eng.__dict__.setdefault("_bmprgp_one_ws", {})
    ws = cache.get(n)
    if ws is None:
        if len(cache) > 8:
            cache.clear()
        ws = cache[n] = _OneWs()
        names = ("p", "Ap", "psi", "w", "nv", "txk", "dx", "Adx", "ytmp")
        rows = int(lib.ccqp_bmprgp_partial_rows())
        nblk = max(int(lib.ccqp_bpipe_partials_doubles(n)) // 3, 1)
        stride = (n + 1) // 2 * 2                        # (planes 16 bytes apart for an odd n too)
        ws.vec = torch.empty((4 + len(names), stride), dtype=torch.float64, device=dev)
        ws.slab = torch.zeros(64, dtype=torch.float64, device=dev)
        ws.partials = torch.empty(rows * nblk, dtype=torch.float64, device=dev)
        ws.words = torch.zeros(4, dtype=torch.int32, device=dev)
        ws.ddesc = torch.zeros(ctypes.sizeof(_lib.PMprgp), dtype=torch.uint8, device=dev)
        d = ws.desc = _lib.PMprgp()
        base = ws.vec.data_ptr()
        plane = [base + 8 * stride * j for j in range(4 + len(names))]
        d.X[0], d.X[1], d.G[0], d.G[1] = plane[0:4]
        for j, name in enumerate(names):
            setattr(d, name, plane[4 + j])
        d.slab, d.partials, d.counter = ws.slab.data_ptr(), ws.partials.data_ptr(), ws.words.data_ptr()
        d.c = 1.0 / (3 * n * _GD)                        # solvers.py:138
        ws.host = (ctypes.c_double * 24)()
    d = ws.desc
    d.A, d.lda, d.b = A.data_ptr(), A.stride(0), b.data_ptr()
    d.seg = P._seg(eng, 0, n)
    d.tol = float(solver.desired_residual_tol)
    check = _lib.check
    dp, batch = ws.ddesc.data_ptr(), ws.words.data_ptr() + 4
    check(lib.ccqp_bmprgp_stage_one_f64(ctx, ctypes.byref(d), dp, None if x0 is None else x0.data_ptr(), n,
                                        float(max_mv - 1) if np.isfinite(max_mv) else 0.0, ws.words.data_ptr(), st))
    gd, eps10 = _GD, float(_EPS10)
    if n <= 32:
        check(lib.ccqp_bsolve_mprgp_f64(ctx, dp, 1, n, gd, eps10, float(min(max_mv - 1, _device_budget(n))), flags & 1, st))
    else:
        # the count stands at 2 after the start-up and grows by at most one per round; the rounds stop where the problem
        # could reach the budget, whose exit paths (:1087, :1102, :1154, :1176) belong to the host-driven loop
        last = max(0, max_mv - 3) if np.isfinite(max_mv) else np.inf
        check(lib.ccqp_bmprgp_start_f64(ctx, dp, 1, n, gd, flags, batch, st))
        if last >= 1:
            def launch(r):
                check(lib.ccqp_bmprgp_round_f64(ctx, dp, 1, n, r, gd, eps10, flags, batch, st))
            _bpipe_poll_loop(eng, 1, launch, last, ahead or _ONE_AHEAD)
    h = ws.host
    check(lib.ccqp_read_f64(ctx, ws.slab.data_ptr() + 8 * 32, 24, h, st))      # blocking: the queue has drained
    status, its = h[17], int(h[8])
    # (a VIEW of the workspace: the caller's export copies it on this stream before the next solve can stage over it)
    return status, ws.vec[(its + 1) % 2, :n], np.sqrt(np.float64(h[19])), int(h[18]), int(h[20])


_HALVES_DEFAULT = True           # lock-step MPRGP: large batches advance as two alternating halves (see _bmprgp_run)
_HALF_MIN_PROBLEMS = 96          # ... from this many problems on ...
_HALF_MIN_BYTES = 2.0 ** 35      # ... if one round of matvecs reads at least this much (32 GiB: 16 GiB per half) ...
_HALF_MIN_N = 4096               # ... of matrices at least this large (below, the stage launches are already one)
_HALVES_CHAINED = True           # the halves' matvecs go through ONE stream, one after the other (False: the free-running
                                 # form of the first A/B, whose two matvec kernels share the machine)
_half_lanes = {}


def _bmprgp_run(eng, spec, n, max_mv, ahead, tune=0, single_launch=None, halves=None, plans=None):
    """Lock-step MPRGP over `spec`; returns (status, (B, n) final iterates, residuals, reported matvecs, executed
    matvecs); status 1 = converged, anything else = not finished here (rows meaningless).
    Large batches (`halves`: by default from 96 problems of n >= 4096 reading 32 GiB per round, with the default matvec)
    advance as TWO runs on two streams, the host alternating between them, so that one half's decision stages and
    vector update -- small launches, a few tens of microseconds each, during which the memory system idles -- go by
    under the other half's matvec.  Every problem is independent, so the results are those of one run, bit for bit.
    `_HALVES_CHAINED` (default) sends both halves' matvecs through the device's ONE matvec stream
    (ccqp_batch_serialize_matvecs), so that they run one after the other at full bandwidth; free-running, two 32 GiB
    matvec kernels share the machine worse than one 64 GiB kernel.  Measured on config 5's share (128 x n = 8192, round
    3, profiles/r03_sweep_config5_{halves,chained,memo,lds}.jsonl): free-running 390-411 problems/s against 421-425 for
    one run; chained, with the VALU matvec a tie (the second table, the event chaining and twice the launches cost what
    the overlap gains), with the persistent LDS-DMA matvec -- whose half-batch launch is as efficient as the full one
    -- 434-435 against 430-431 (+1.1 %; 256 x 4096: +0.5 %; 64 x 8192 and 256 x 2048: nothing or a loss, hence the
    thresholds)."""
    B = len(spec)
    single = plans is None and (_single_launch_default("mprgp", n, B) if single_launch is None else bool(single_launch)) \
        and n <= 2048
    if halves is None:
        halves = _HALVES_DEFAULT and (not single) and tune == 0 and n >= _HALF_MIN_N and B >= _HALF_MIN_PROBLEMS \
            and 8.0 * n * n * B >= _HALF_MIN_BYTES
    if not halves or B < 2:
        run = _MprgpRun(eng, spec, n, max_mv, tune, single_launch, plans)
        if run.last >= 1:
            _bpipe_poll_loop(eng, B, run.launch, run.last, ahead)
        return run.finish()
    from . import _lib
    from ._engine import make_lanes
    stats["halved_runs"] += 1
    lanes = _half_lanes.get(eng.index)
    if lanes is None:
        lanes = _half_lanes[eng.index] = make_lanes(eng.index, 2)
    cur = eng.torch_stream
    cut = (B + 1) // 2
    runs = []
    for lane, (lo, hi) in zip(lanes, ((0, cut), (cut, B))):
        lane.torch_stream.wait_stream(cur)               # the caller's matrices / vectors are ready on ITS stream
        lane.gemv_variant = eng.gemv_variant
        _lib.check(lane.lib.ccqp_batch_serialize_matvecs(lane.ctx, 1 if _HALVES_CHAINED else 0))
        with torch.cuda.stream(lane.torch_stream):
            runs.append(_MprgpRun(lane, spec.part(lo, hi), n, max_mv, tune, single_launch,
                                  None if plans is None else plans[lo:hi]))
    _bpipe_poll_many(runs, ahead)
    outs = []
    for r in runs:
        with torch.cuda.stream(r.eng.torch_stream):
            outs.append(r.finish())
    for lane in lanes:
        cur.wait_stream(lane.torch_stream)               # (and every later use of the lanes starts by waiting for `cur`)
    with torch.cuda.stream(cur):
        sols = torch.cat([o[1].reshape(-1, n) for o in outs])
    return (np.concatenate([o[0] for o in outs]), sols, np.concatenate([o[2] for o in outs]),
            np.concatenate([o[3] for o in outs]), np.concatenate([o[4] for o in outs]))


def _bmprgp_group(eng, items, n, max_mv, ahead, tune=0, single_launch=None, segmented=False):
    import contextlib
    import io
    plans = [it[4]._segplan(eng, n) for it in items] if segmented else None
    status, sols, res, mvs, execd = _bmprgp_run(eng, _spec_from_items(eng, items, n), n, max_mv, ahead, tune,
                                                single_launch, plans=plans)
    host = None
    rows, stl, mvl, exl, ress = _rows(sols, len(items)), status.tolist(), mvs.tolist(), execd.tolist(), list(res)
    executed = 0
    for p, it in enumerate(items):
        s = it[0]
        if stl[p] != 1.0:                                # budget reached (or the search that cannot terminate)
            A, b, x0, P = it[6]
            with contextlib.redirect_stdout(io.StringIO()):
                stats["fallback_solves"] += 1
                _host_solve(s, A, b, x0, P)
            continue
        if it[5]:
            s._solution = rows[p]
        else:
            if host is None:
                host = sols.cpu().numpy()
            s._solution = host[p]
        s._solution_converged = mvl[p] < max_mv
        s._solution_residual = ress[p]
        s._solution_num_matrix_vector_mults = mvl[p]
        s._executed_matrix_vector_mults = exl[p]
        executed += exl[p]
    eng.mv_executed += executed


# ---------------------------------------------------------------------------------------------
# Stacked problems: the lock-step runs without any per-problem Python.
# ---------------------------------------------------------------------------------------------
class StackedResult:
    """Results of solve_stacked, one row / entry per problem: `solution` (B, n) device tensor; numpy arrays
    `residual`, `converged`, `num_matrix_vector_multiplications` (the reference's count), `executed_matrix_vector_
    multiplications`; `solution_time` the wall-clock of the whole call."""
    __slots__ = ("solution", "residual", "converged", "num_matrix_vector_multiplications",
                 "executed_matrix_vector_multiplications", "solution_time")


def solve_stacked(solver, A, b, x0=None, lower=None, upper=None, tol=1e-8, max_mv=np.inf, device=None, ahead=4,
                  tune=0, single_launch=None, step_size=0.01, rngs=None, m=5, tau=0.5, sigma1=0.01, sigma2=0.5,
                  feasible_set=None):
    """B dense problems of one size handed over as STACKED device tensors -- A (B, n, n), b (B, n), optional x0
    (B, n) -- and solved by the lock-step schedulers of solve_batch_pipelined with no per-problem work on the host
    (at n <= 512 a Python loop over the problems costs more than the GPU needs to solve them).

    solver        : "PGD" (constant `step_size`, one or one per problem), "BBPGD", "BBPGDf", "APGD",
                    "APGDAntiRelaxation", "MPRGP", "SPG" (`rngs`: B distinct np.random.RandomState objects, problem p
                    draws from rngs[p] and leaves it where its own solve would; m, tau, sigma1, sigma2: SPG's
                    parameters), or the solver class
    lower / upper : bounds of the feasible set: None, a Python / numpy scalar, an (n,) or a (B, n) array / tensor.
                    Neither -> Identity, one -> Lower / Upper bound, both -> Box.  MPRGP needs array bounds
                    (scalar ones raise in the reference's normal_vector).
    feasible_set  : instead of lower / upper: ONE *ProjOp of dimension n built from the seven operators (Sphere, Cone,
                    DisjointUnion of any blocks, ...), shared by all problems -- the segmented lock-step forms
                    (ccqp_bsegs.hip / ccqp_bspg.hip / ccqp_bmprgp.hip; MPRGP: no Cone block, array bounds)
    tol           : one tolerance or one per problem
    Problem p gets bit for bit the result of `Solver(tol[p], max_mv).solve(A[p], b[p], x0[p], P_p)` with P_p the
    corresponding *ProjOp (tests/test_gpu_stacked.py).  Returns a StackedResult."""
    from . import _lib
    from . import solution_spaces as ss
    from . import solvers as sv
    t0 = time.time()
    name = solver if isinstance(solver, str) else {sv.CCQPSolverBBPGD: "BBPGD", sv.CCQPSolverAPGD: "APGD",
                                                   sv.CCQPSolverMPRGP: "MPRGP", sv.CCQPSolverPGD: "PGD",
                                                   sv.CCQPSolverBBPGDf: "BBPGDf", sv.CCQPSolverSPG: "SPG",
                                                   sv.CCQPSolverAPGDAntiRelaxation: "APGDAntiRelaxation"}.get(solver)
    if name not in ("PGD", "BBPGD", "BBPGDf", "APGD", "APGDAntiRelaxation", "MPRGP", "SPG"):
        raise TypeError("solve_stacked runs PGD, BBPGD, BBPGDf, APGD, APGDAntiRelaxation, MPRGP or SPG, got %r"
                        % (solver,))
    eng = get_engine(device if device is not None else (A.device if isinstance(A, torch.Tensor) else None))
    eng.bind()
    A = eng.to_device(A) if not (isinstance(A, torch.Tensor) and A.dtype == torch.float64 and A.device == eng.device) else A
    if A.dim() != 3 or A.shape[1] != A.shape[2]:
        raise ValueError("solve_stacked: A must be (B, n, n)")
    if A.stride(2) != 1 or A.stride(1) < A.shape[2]:
        A = A.contiguous()
    B, n = int(A.shape[0]), int(A.shape[1])
    if B == 0:                                          # an empty batch
        out = StackedResult()
        out.solution = torch.empty((0, n), dtype=torch.float64, device=eng.device)
        out.residual = np.zeros(0)
        out.converged = np.zeros(0, dtype=bool)
        out.num_matrix_vector_multiplications = np.zeros(0, dtype=np.int64)
        out.executed_matrix_vector_multiplications = np.zeros(0, dtype=np.int64)
        out.solution_time = time.time() - t0
        return out
    b = eng.to_device(b).reshape(B, n)
    if x0 is not None:
        x0 = eng.to_device(x0).reshape(B, n)
    if name == "SPG":
        rngs = list(rngs) if rngs is not None else []
        if len(rngs) != B or not all(isinstance(r, np.random.RandomState) for r in rngs) or \
                len({id(r) for r in rngs}) != B:
            raise TypeError("solve_stacked('SPG', ...) needs rngs = B distinct np.random.RandomState objects (numpy's "
                            "global stream cannot be shared by problems that advance together)")
        if not (isinstance(m, (int, np.integer)) and 1 <= m <= _SPG_MAX_M):
            raise ValueError("solve_stacked: SPG's queue length m must be an integer in [1, %d]" % _SPG_MAX_M)

    def is_scalar(v):
        return isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool)

    plans = None
    if feasible_set is not None:
        if lower is not None or upper is not None:
            raise ValueError("solve_stacked: either lower / upper or feasible_set")
        if hasattr(feasible_set, "_dev_bb_tail"):
            raise TypeError("solve_stacked: an elementwise feasible_set is given through lower / upper")
        if not isinstance(feasible_set, ss.ProjOpBase) or not feasible_set._has_device_form():
            raise TypeError("solve_stacked(feasible_set=...) needs a set built from the seven operators, got %r"
                            % type(feasible_set).__name__)
        feasible_set._refresh()
        if name == "MPRGP" and not feasible_set._normal_everywhere_on_device():
            raise TypeError("solve_stacked('MPRGP', feasible_set=...) needs a set whose normal_vector exists on the device "
                            "(no Cone block -- the reference raises NotImplementedError --, array bounds)")
        if getattr(feasible_set, "_fixed_dim", False) and feasible_set.embedded_dimension != n:
            raise ValueError("solve_stacked: the feasible set does not match b")
        plans = [feasible_set._segplan(eng, n)] * B      # (the tables are read-only: one plan for all problems)
    kind = {(False, False): _lib.OP_IDENTITY, (True, False): _lib.OP_LOWER, (False, True): _lib.OP_UPPER,
            (True, True): _lib.OP_BOX}[(lower is not None, upper is not None)]
    if name == "MPRGP" and (is_scalar(lower) or is_scalar(upper)):
        raise TypeError("solve_stacked runs MPRGP on sets with array bounds")
    seg = np.zeros(B, dtype=np.dtype(_lib.Seg))
    seg["kind"], seg["dim"] = kind, n
    keep = []
    for field, sfield, flag, v in (("lb", "lb_s", _lib.SEG_LB_INT, lower), ("ub", "ub_s", _lib.SEG_UB_INT, upper)):
        if v is None:
            continue
        if is_scalar(v):
            seg[sfield] = float(v)
            if isinstance(v, (int, np.integer)):
                seg["flags"] |= flag                    # integer bounds multiply integer masks (solution_spaces.py)
            continue
        if not isinstance(v, torch.Tensor) and np.asarray(v).dtype.kind in "iu":
            seg["flags"] |= flag
        t = eng.to_device(v)
        keep.append(t)
        if t.numel() == n:
            seg[field] = t.data_ptr()
        elif tuple(t.shape) == (B, n):
            seg[field] = np.uint64(t.data_ptr()) + np.arange(B, dtype=np.uint64) * np.uint64(8 * n)
        else:
            raise ValueError("solve_stacked: bounds must be scalars, (n,) or (B, n)")
    sp = _Spec()
    sp.B, sp.n = B, n
    idx = np.arange(B, dtype=np.uint64)
    sp.A = np.uint64(A.data_ptr()) + idx * np.uint64(8 * A.stride(0))
    sp.lda = np.full(B, A.stride(1), dtype=np.int64)
    sp.b = np.uint64(b.data_ptr()) + idx * np.uint64(8 * n)
    sp.seg = seg
    sp.tol = np.broadcast_to(np.asarray(tol, dtype=np.float64), (B,)).copy()
    sp.x0 = x0
    sp.keep = keep
    sp.step = np.broadcast_to(np.asarray(step_size, dtype=np.float64), (B,)).copy() if name == "PGD" else None
    sp.fallback = name == "BBPGDf"
    out = StackedResult()
    with np.errstate(all="ignore"):
        if name in ("BBPGD", "BBPGDf", "PGD"):
            if plans is not None:
                done, final, sols, res = _bsegs_run(eng, sp, plans, n, max_mv, ahead, tune)
                todo = np.zeros(0, dtype=np.int64)
            else:
                done, final, sols, res, todo = _bpipe_run(eng, sp, n, max_mv, ahead, tune, single_launch)
            mv = 1 + final
            execd = mv if name == "PGD" else np.where(done & (final == 0), 1, 2 + final)
        elif name in ("APGD", "APGDAntiRelaxation"):
            done, sols, res, mv = _bapgd_run(eng, sp, n, max_mv, ahead, tune, single_launch,
                                             ar=name == "APGDAntiRelaxation", plans=plans)
            execd = mv
            todo = np.flatnonzero(~done)
        elif name == "SPG":
            host, sols = _bspg_run(eng, sp, None, n, max_mv, tune, plans=plans, rngs=rngs,
                                   params=(m, tau, sigma1, sigma2))
            mv, res = host["mv"].copy(), np.sqrt(host["dd"])
            execd = mv - (host["status"] == 2)           # (the matvec that ends a budget is counted, not executed)
            todo = np.flatnonzero(host["status"] == 3)
            for p in todo:
                if host["used"][p]:                      # the reference's draw raises here: numpy's own OverflowError
                    np.random.RandomState(0).uniform(low=0.0, high=np.nan)
        else:
            status, sols, res, mv, execd = _bmprgp_run(eng, sp, n, max_mv, ahead, tune, single_launch, plans=plans)
            todo = np.flatnonzero(status != 1.0)
    mv, execd, res = np.array(mv, dtype=np.int64), np.array(execd, dtype=np.int64), np.array(res, dtype=np.float64)
    if B == 1:
        sols = sols.reshape(1, n).clone()
    eng.mv_executed += int(np.delete(execd, todo).sum())
    if len(todo):
        # matvec budget reached (APGD / MPRGP), or the single-launch kernel's own budget: the host-driven loop
        import contextlib
        import io
        cls = getattr(sv, "CCQPSolver" + name)

        def bound(v, p):
            if v is None or is_scalar(v):
                return v
            t = v if isinstance(v, torch.Tensor) else np.asarray(v)
            t = t if t.shape[0] == n and t.ndim == 1 else t[p]
            return t.cpu().numpy() if isinstance(t, torch.Tensor) else t
        for p in todo:
            lo, up = bound(lower, p), bound(upper, p)
            P = feasible_set if feasible_set is not None else \
                {_lib.OP_IDENTITY: lambda: ss.IdentityProjOp(n), _lib.OP_LOWER: lambda: ss.LowerBoundProjOp(n, lo),
                 _lib.OP_UPPER: lambda: ss.UpperBoundProjOp(n, up), _lib.OP_BOX: lambda: ss.BoxProjOp(n, lo, up)}[kind]()
            with contextlib.redirect_stdout(io.StringIO()):
                args = (float(sp.tol[p]), max_mv) + ((float(sp.step[p]),) if name == "PGD" else ())
                stats["fallback_solves"] += 1
                s = cls(*args) if name != "SPG" else cls(*args, m=m, tau=tau, sigma1=sigma1, sigma2=sigma2)
                if name == "SPG":                        # (nothing drawn yet: the host loop raises what the reference raises)
                    s.rng = rngs[p]
                s = _host_solve(s, A[p], b[p], None if x0 is None else x0[p], P)
            sols[p].copy_(s.solution)
            res[p], mv[p] = s.solution_residual, s.solution_num_matrix_vector_multiplications
            execd[p] = s.executed_matrix_vector_multiplications
    out.solution = sols
    out.residual = res
    out.num_matrix_vector_multiplications = mv
    out.executed_matrix_vector_multiplications = execd
    out.converged = mv < max_mv
    out.solution_time = time.time() - t0
    return out
