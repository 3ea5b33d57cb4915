"""First-order solvers for  min 1/2 x'Ax + b'x  s.t. x in Omega  -- MI355X (HIP) implementation.

Drop-in for `ccqppy.solvers` (reference: /root/reference/src/ccqppy/solvers.py): same seven class
names, constructor signatures, `solve(A, b, x0=None, convex_proj_op=None) -> self`, the six result
properties, the `name` strings and the stdout line printed per solve.

Division of labour (BASELINE.json north_star): the host loop below is Python and only does what
the reference's loop does with SCALARS (step sizes, Lipschitz test, momentum, MV counting, exit
tests, SPG's draw from numpy's global RNG) -- in numpy float64 arithmetic so branch decisions
match.  Every VECTOR operation (matvec, axpy, dot/norm, projection, residual) is a kernel of
libccqp_hip.so acting on device-resident fp64 vectors; nothing vector-sized crosses PCIe during
a solve.  Host<->device traffic per iteration = one read of a few scalars.

`A` may be a numpy array (uploaded once), a torch tensor on the GPU (zero copy), or an operator
from `ccqppy_amd.operators` (e.g. RowShardedOperator for one-process-per-GPU row sharding).
The returned `solution` is a numpy array unless b was a torch tensor.
"""
from abc import ABC, abstractmethod
from collections import deque
import time

import numpy as np
import torch

from . import solution_spaces as ss
from ._lib import check as _lib_check
from .operators import as_operator

__all__ = ["CCQPSolverBase", "CCQPSolverPGD", "CCQPSolverAPGD", "CCQPSolverAPGDAntiRelaxation",
           "CCQPSolverBBPGD", "CCQPSolverBBPGDf", "CCQPSolverSPG", "CCQPSolverMPRGP"]

_GD = 1e-6                            # solvers.py:137,321,496,629,768,1070
_EPS10 = 10 * np.finfo(float).eps     # solvers.py:656,...


class _Run:
    """Per-solve device state: engine, operator, staged inputs, fresh-vector factory."""

    def __init__(self, A, b, x0, convex_proj_op, engine=None):
        self.n = n = b.shape[0]
        self.P = convex_proj_op if convex_proj_op is not None else ss.IdentityProjOp(n)   # :115-116
        if not isinstance(self.P, ss.ProjOpBase):
            # anything callable is a legal feasible set for the reference (:115-116): run it on the host, slowly
            # (two PCIe copies per projection, solution_spaces.HostCallableProjOp); a non-callable raises the
            # TypeError the reference's first `convex_proj_op(x)` would
            self.P = ss.HostCallableProjOp(n, self.P)
        self.P._refresh()                      # public parameters edited since the last solve are picked up here
        self.op = as_operator(A)
        # `engine` = a lane engine of the batch scheduler (own stream + scalar slab); default: the
        # operator's engine, bound to torch's current stream
        self.eng = eng = engine if engine is not None else self.op.engine
        eng.bind()
        self.torch_out = isinstance(b, torch.Tensor)
        self.b = eng.to_device(b).reshape(-1)
        self.x0_given = x0 is not None
        if x0 is None:
            self.x0 = torch.zeros(n, dtype=torch.float64, device=eng.device)             # :125-126
        else:
            self.x0 = eng.to_device(x0).reshape(-1)
        self.c = 1.0 / (3 * n * _GD)                                                     # :138
        self.mv0 = eng.mv_executed

    def new(self):
        return self.eng.empty(self.n)

    def copy(self, x):
        out = self.new()
        self.eng.copy(x, out)
        return out

    def mv(self, v, with_g=False):
        """A.v (and A.v + b) into fresh vectors."""
        y = self.new()
        if with_g:
            g = self.new()
            self.op.matvec(v, y, self.b, g, self.eng)
            return y, g
        self.op.matvec(v, y, None, None, self.eng)
        return y

    def mv2(self, v1, with_g1, v2, with_g2):
        """(A.v1[, A.v1 + b]), (A.v2[, A.v2 + b]) from ONE call of the operator's matvec2 (one pass over a dense A)."""
        y1, y2 = self.new(), self.new()
        g1 = self.new() if with_g1 else None
        g2 = self.new() if with_g2 else None
        self.op.matvec2(v1, y1, g1, v2, y2, g2, self.b, self.eng)
        return (y1, g1), (y2, g2)

    def mv3(self, vs, with_g):
        """[(A.v, A.v + b | None) | None for v in vs] from ONE call of the operator's matvec3 (None entries of vs: no
        product): one pass over a square dense A."""
        ys = [None if v is None else self.new() for v in vs]
        gs = [self.new() if (v is not None and wg) else None for v, wg in zip(vs, with_g)]
        self.op.matvec3(list(vs), ys, gs, self.b, self.eng)
        return [None if y is None else (y, g) for y, g in zip(ys, gs)]

    def lin(self, a, x, b, y):
        out = self.new()
        self.eng.axpby(a, x, b, y, out)
        return out

    def step_project(self, x, alpha, g):
        """P(x - alpha*g)"""
        out = self.new()
        self.P._dev_step_project(self.eng, 1.0, x, -alpha, g, out)
        return out

    def project(self, x):
        if isinstance(self.P, ss.IdentityProjOp):
            return x                       # same object, as the reference
        out = self.new()
        self.P._dev_project(self.eng, x, out)
        return out

    def residual_sq(self, x, g, slot):
        self.P._dev_residual_sq(self.eng, x, g, self.c, _GD, slot)

    # -- synchronisation points.  The solver loops are GENERATORS: wherever the reference needs a
    # scalar on the host they `yield (slot0, count)` and are resumed with the values.  solve() answers
    # with a blocking read; the batch scheduler (batch.py) parks the solve and runs another lane.
    def read(self, slot0, count):
        return (yield (slot0, count))

    def dots_now(self, pairs, slot0=0):
        self.eng.dots(pairs, slot0)
        return (yield (slot0, len(pairs)))

    def residual_now(self, x, g, slot=0):
        self.residual_sq(x, g, slot)
        v = yield (slot, 1)
        return np.sqrt(v[0])

    def export(self, x):
        if self.torch_out:
            return x.clone()
        return x.cpu().numpy()


class CCQPSolverBase(ABC):
    """Abstract base class for constrained quadratic programming problems (solvers.py:11-68)."""
    _label = "?"

    @abstractmethod
    def __init__(self, desired_residual_tol, max_matrix_vector_multiplications=np.inf):
        self.desired_residual_tol = desired_residual_tol
        self.max_matrix_vector_multiplications = max_matrix_vector_multiplications
        self._solution = None
        self._solution_residual = None
        self._solution_converged = None
        self._solution_time = None
        self._solution_num_matrix_vector_mults = None
        self._executed_matrix_vector_mults = None
        # diagnostics, OFF by default: set `solver.trace = []` before solve() and the host loop appends one dict of
        # its steering scalars per completed iteration (the keys of oracle/cases.py:TRACE_COLUMNS) -- what the
        # parity tests compare with the per-iteration traces stored in tests/golden/solvers.npz.  Host-driven loops
        # only: a device-resident form (pipeline=True, the lock-step schedulers) keeps its scalars on the GPU.
        self.trace = None

    def solve(self, A, b, x0=None, convex_proj_op=None):
        """Solve; returns self.  A: (n, n); b: (n,); x0: (n,) or None (zeros); convex_proj_op:
        projection onto the feasible set (default IdentityProjOp)."""
        time_start = time.time()
        self._checkSolveInput(A, b, x0)
        print("solving " + self._label)
        run = _Run(A, b, x0, convex_proj_op)
        gen = self._iterate(run)
        with np.errstate(all="ignore"):
            try:
                req = next(gen)
                while True:
                    req = gen.send(run.eng.read(*req))      # blocking scalar read
            except StopIteration as stop:
                x, res, mv_count = stop.value
        self._finish(run, x, res, mv_count, time_start)
        return self

    def _start(self, A, b, x0=None, convex_proj_op=None, engine=None):
        """Batch-scheduler entry: stage one solve on `engine` and return (run, generator)."""
        self._checkSolveInput(A, b, x0)
        run = _Run(A, b, x0, convex_proj_op, engine)
        return run, self._iterate(run)

    def _finish(self, run, x, res, mv_count, time_start):
        self._solution = run.export(x)
        self._solution_converged = mv_count < self.max_matrix_vector_multiplications
        self._solution_residual = res
        self._solution_num_matrix_vector_mults = mv_count
        self._executed_matrix_vector_mults = run.eng.mv_executed - run.mv0
        self._solution_time = time.time() - time_start

    @abstractmethod
    def _iterate(self, run):
        """Generator; yields (slot0, count) read requests, returns (solution device vector,
        residual, reported mv_count)."""

    def _checkSolveInput(self, A, b, x0):
        pass

    def _device_resident(self, run):
        """The effective value of `pipeline` for this solve.  None (the default of all seven solvers) = ON for a plain solve() on the default engine with a dense operator, an elementwise feasible set
        and a matvec budget of at least 1000: the device-resident forms give the same bits and counts 1.5-3.1x faster at
        n <= 2048 and up to 1.4x at 4096 ... 16384 (profiles/r04_single_mid_all.jsonl); a solve MEANT to run into a small
        budget stays on the host-driven loop, whose exit paths are the reference's."""
        pipeline = getattr(self, "pipeline", False)
        if pipeline is None:
            from .operators import DenseOperator
            pipeline = isinstance(run.op, DenseOperator) and hasattr(run.P, "_dev_bb_tail") \
                and self.max_matrix_vector_multiplications >= 1000 and run.eng.own_stream is None
        return pipeline

    # (with the one-launch staging the batch of one stays ahead of the host loop up to here: PGD 1.3x at 4096, 1.05x at
    # 8192; BBPGD 1.33x / 1.06x; BBPGDf 1.8x / 1.3x, 1.1x at 16384 -- profiles/r04_single_cap_lifted.jsonl)
    _LOCKSTEP_ONE_MAX_N = 16384

    def _lockstep_one(self, run):
        """PGD / BBPGDf with a device-resident solve (pipeline): the whole solve as a lock-step batch of ONE (ccqp_bpipe_*:
        dense operator, n <= _LOCKSTEP_ONE_MAX_N, elementwise set) -- step, matvec and the fused tail queued ahead of the
        device's reports, every scalar on the device; n <= 64: one launch (one workgroup: a form for MANY problems -- for
        one it loses from n = 128 on, as APGD's).  Returns (x, residual, reported matvecs), or None where that does not
        apply or the device's own budget was reached."""
        from ._record import RecordingEngine
        from .lockstep import _bpipe_one, stats
        from .operators import DenseOperator
        op, eng = run.op, run.eng
        if not (isinstance(op, DenseOperator) and op.rows == op.n <= self._LOCKSTEP_ONE_MAX_N
                and hasattr(run.P, "_dev_bb_tail") and not isinstance(eng, RecordingEngine) and eng.gemv_events is None):
            return None
        done, k, x, res, gave_up = _bpipe_one(eng, self, op, run.b, run.x0 if run.x0_given else None, run.P, run.n)
        if gave_up:
            stats["fallback_solves"] += 1
            return None
        # (BBPGD-type: A.g0 of the start is executed, not counted -- unless x0 already meets the tolerance; PGD has none)
        extra = 0 if hasattr(self, "step_size") else (0 if (done and k == 0) else 1)
        eng.mv_executed += 1 + k + extra
        return x, res, 1 + k

    @property
    @abstractmethod
    def name(self):
        """Return the name of the solver."""

    @property
    def solution(self):
        return self._solution

    @property
    def solution_residual(self):
        return self._solution_residual

    @property
    def solution_converged(self):
        return self._solution_converged

    @property
    def solution_time(self):
        return self._solution_time

    @property
    def solution_num_matrix_vector_multiplications(self):
        return self._solution_num_matrix_vector_mults

    @property
    def executed_matrix_vector_multiplications(self):
        """Matvecs actually launched (the reference's counter skips solvers.py:635,775,1129,1163,1172)."""
        return self._executed_matrix_vector_mults


class CCQPSolverPGD(CCQPSolverBase):
    """Projected gradient descent with a constant step (solvers.py:71-194)."""
    _label = "PGD"

    def __init__(self, desired_residual_tol, max_matrix_vector_multiplications=np.inf, step_size=0.01, pipeline=None):
        super().__init__(desired_residual_tol, max_matrix_vector_multiplications)
        self.step_size = step_size
        # SAME results bit for bit: None (default) = the device-resident form where _device_resident says so (a plain
        # solve() with a dense operator of n <= 16384, an elementwise set, a budget of at least 1000): the lock-step batch
        # scheduler with a batch of one -- step, matvec, residual and exit test on the device, iterations queued ahead.
        # PGD takes thousands of iterations of ~25 us of host turn-around each; False = the host-driven loop below.
        self.pipeline = pipeline

    @property
    def name(self):
        return "PGD"

    def _iterate(self, run):
        tol, max_mv = self.desired_residual_tol, self.max_matrix_vector_multiplications
        if self._device_resident(run) and self.trace is None:
            out = self._lockstep_one(run)
            if out is not None:
                return out
        mv_count = 0
        xk = run.copy(run.x0)
        xkm1 = run.copy(run.x0)
        _, gkm1 = run.mv(xkm1, True)
        mv_count += 1
        res = yield from run.residual_now(xkm1, gkm1)
        if res >= tol:
            while True:
                xk = run.step_project(xkm1, self.step_size, gkm1)
                _, gk = run.mv(xk, True)
                mv_count += 1
                if mv_count >= max_mv:
                    break
                res = yield from run.residual_now(xk, gk)
                if self.trace is not None:
                    self.trace.append(dict(mv=mv_count, res=float(res)))
                if res < tol:
                    break
                xk, xkm1 = xkm1, xk
                gk, gkm1 = gkm1, gk
        return xk, res, mv_count


class _APGDCore(CCQPSolverBase):
    """APGD (solvers.py:197-367) and its anti-relaxation variant (:370-557)."""
    _anti_relaxation = False

    def __init__(self, desired_residual_tol, max_matrix_vector_multiplications=np.inf, reuse_matvec=False,
                 pipeline=None, pair_matvecs=None):
        super().__init__(desired_residual_tol, max_matrix_vector_multiplications)
        # SAME results bit for bit, same MV counts: the two products of an iteration share ONE pass over A.
        # y_{k+1} = (1 + beta) x_{k+1} - beta x_k (solvers.py:313-316) depends on x_{k+1}, x_k and theta only -- not on
        # the line search -- so it is formed as soon as x_{k+1} is, and A.x_{k+1} (:290) and A.y_{k+1} (:275 of the NEXT
        # iteration) come from one two-vector matvec (operator.matvec2 -> ccqp_gemv2_f64: each product with the bits of
        # its own matvec).  If the line search rejects x_{k+1} (:306-310) the second product is discarded and the retry
        # is again one pass -- never more passes than the reference has matvecs.  None (default): wherever the operator
        # reads its matrix once for the pair (dense, aligned, an even n >= 4096); True / False force it.  The
        # anti-relaxation variant's restart (:510-512) makes y_{k+1} = x_{k+1}: then the product computed ahead is dropped
        # and A.x_{k+1}, which the same pass computed, is the next iteration's A.y_k.
        self.pair_matvecs = pair_matvecs
        # SAME results bit for bit: the DEVICE-RESIDENT form of a solve (dense operator with n <= 16384, any set with a
        # device form): line search, momentum and exit test run on the device and rounds are queued ahead (ccqp_bapgd_*, the
        # lock-step batch scheduler with a batch of one: n <= 64 one launch for the whole solve; beyond, staging in one
        # launch and FUSED rounds -- one per accepted iteration: below n = 4096 its two products from one launch of the
        # batched VALU kernel, from 4096 on the persistent two-vector matvec).  Against the host-driven loop below (one
        # blocking scalar read + ~25 us of Python per iteration): 2.6-3.0x at n <= 1024, 2.0x at 2048, 1.3-1.5x at 4096,
        # 1.05-1.15x at 8192 / 16384 (profiles/r04_single_mid_all.jsonl); beyond, a pass over A hides the host.  None (default): ON for elementwise sets when the matvec budget is at
        # least 1000 (a solve that reaches its budget on the device is repeated by the host-driven loop -- the exit paths
        # are the reference's -- so a solve MEANT to run into a small budget, the Cone configs, stays on the host loop);
        # True / False force it.
        self.pipeline = pipeline
        # Extension, OFF by default: y_{k+1} = (1+beta) x_{k+1} - beta x_k is linear, so
        # A.y_{k+1} = (1+beta) A.x_{k+1} - beta A.x_k needs no matvec.  Saves one of the two matvecs per
        # iteration (half the HBM traffic of a solve) but changes the reported MV count and rounds A.y
        # differently from a direct product: an opt-in, NON-parity mode.
        self.reuse_matvec = reuse_matvec

    def _iterate(self, run):
        tol, max_mv = self.desired_residual_tol, self.max_matrix_vector_multiplications
        eng, b, AR = run.eng, run.b, self._anti_relaxation
        reuse = self.reuse_matvec
        # (None: plain solve() on the default engine only -- a lane of the host-driven batch schedulers keeps interleaving)
        pipeline = self._device_resident(run)
        if pipeline and not reuse and self.trace is None:
            out = self._lockstep(run)
            if out is not None:
                return out
        Ay_next = None                                  # A.yk obtained without a matvec (reuse mode)
        Axk = None                                      # A.xk
        # pair mode: A.yk (and A.yk + b) computed ahead, in the pass that computed the previous A.x
        pair = self.pair_matvecs
        if pair is None:
            pair = hasattr(run.op, "pair_in_one_pass") and run.op.pair_in_one_pass(eng)
        from ._record import RecordingEngine            # (the fused batch scheduler records launches: no pair form there)
        pair = bool(pair) and not reuse and hasattr(run.op, "matvec2") and not isinstance(eng, RecordingEngine)
        Ay_ahead = None
        mv_count = 0
        xk = run.copy(run.x0)
        yk = run.copy(run.x0)
        ones = run.new()
        eng.fill(1.0, ones)
        xhat = ones                                     # :432
        thk = 1.0
        thk1 = 1.0
        d = run.lin(1.0, xk, -1.0, ones)                # xk - ones
        if pair:
            (Ad, _), Ay_ahead = run.mv2(d, False, yk, True)
            eng.mv_executed -= 1                        # (the second product counts where the reference computes it)
        else:
            Ad = run.mv(d)
        s = yield from run.dots_now([(Ad, Ad), (d, d)])
        mv_count += 1
        L = np.sqrt(s[0]) / np.sqrt(s[1])               # :264-268
        t = 1.0 / L
        resmin = np.inf
        xkp1 = None
        res = None
        while True:
            if reuse and Ay_next is not None:
                Ayk = Ay_next
                gk = run.lin(1.0, Ayk, 1.0, b)
            else:
                if Ay_ahead is not None:
                    Ayk, gk = Ay_ahead                  # computed in the previous pass: counted here, as :275-278
                    Ay_ahead = None
                    eng.mv_executed += 1
                else:
                    Ayk, gk = run.mv(yk, True)
                mv_count += 1
                if mv_count >= max_mv:
                    break
                if reuse:
                    Axk = Ayk                           # first iteration: xk == yk == x0
            xkp1 = run.step_project(yk, t, gk)
            if pair:                                    # (scalars of :313-314: functions of theta_k alone)
                thk1 = 0.5 * (-thk * thk + thk * np.sqrt(4 + thk * thk))
                beta = thk * (1 - thk) / (thk * thk + thk1)
            first = True
            nback = 0
            while True:
                if pair:
                    ykp1 = run.lin(1 + beta, xkp1, -beta, xk)          # :316, ahead of the line search
                    (Ax1, g1), Ay_ahead = run.mv2(xkp1, True, ykp1, True)
                    eng.mv_executed -= 1                # (counted when the next iteration takes it, or never)
                else:
                    Ax1, g1 = run.mv(xkp1, True)
                mv_count += 1
                if mv_count >= max_mv:
                    # the reference leaves the inner loop only; the tail below still runs
                    run.residual_sq(xkp1, g1, 7)
                    if AR:
                        eng.dots([(gk, run.lin(1.0, xkp1, -1.0, xk))], 6)
                    res2 = yield from run.read(6, 2)
                    break
                fused = hasattr(run.P, "_dev_fused_dots_residual")
                d = (xkp1, yk) if fused else run.lin(1.0, xkp1, -1.0, yk)     # fused: formed on the fly
                pairs = [(xkp1, Ax1), (xkp1, b), (gk, d), (d, d)]
                if AR:
                    pairs.append((gk, (xkp1, xk) if fused else run.lin(1.0, xkp1, -1.0, xk)))
                if first:
                    pairs = [(yk, Ayk), (yk, b)] + pairs
                # speculative: the residual needs xkp1 and A.xkp1 only -> same sync as the dots; for
                # elementwise sets dots, differences and residual are ONE launch with identical bits
                if fused:
                    run.P._dev_fused_dots_residual(eng, pairs, 0 if first else 2, xkp1, g1, run.c, _GD, 7)
                else:
                    eng.dots(pairs, 0 if first else 2)
                    run.residual_sq(xkp1, g1, 7)
                if first:
                    sc = yield from run.read(0, 8)
                    r1 = sc[0] * 0.5
                    r2 = sc[1]
                    first = False
                else:
                    sc = [None, None] + (yield from run.read(2, 6))
                l1 = sc[2] * 0.5
                l2 = sc[3]
                r3 = sc[4]
                r4 = 0.5 * L * sc[5]
                res2 = sc[6:8]
                if (l1 + l2) <= (r1 + r2 + r3 + r4):    # :302
                    break
                L *= 2
                t = 1.0 / L
                xkp1 = run.step_project(yk, t, gk)
                nback += 1
            if not pair:
                thk1 = 0.5 * (-thk * thk + thk * np.sqrt(4 + thk * thk))
                beta = thk * (1 - thk) / (thk * thk + thk1)
                ykp1 = run.lin(1 + beta, xkp1, -beta, xk)   # (1+beta)*xkp1 - beta*xk
            if reuse:
                Ay_next = run.lin(1 + beta, Ax1, -beta, Axk)
            res = np.sqrt(res2[1])
            if self.trace is not None:
                self.trace.append(dict(mv=mv_count, res=float(res), L=float(L), nback=nback))
            if AR and res < resmin:                     # :501-503
                resmin = res
                xhat = run.copy(xkp1)
            if res < tol:
                break
            if AR and res2[0] > 0:                      # gk.dot(xkp1 - xk) > 0, :510-512
                ykp1 = run.copy(xkp1)
                thk1 = 1
                if reuse:
                    Ay_next = Ax1
                if pair:                                # y_{k+1} is x_{k+1} itself: its product is the pass's FIRST one
                    Ay_ahead = (Ax1, g1)
            L *= 0.9
            t = 1.0 / L
            yk, ykp1 = ykp1, yk
            xk, xkp1 = xkp1, xk                         # `xkp1` now names the OLDER iterate
            Axk = Ax1
            thk = thk1
        if res is None:
            raise UnboundLocalError("local variable 'xkp1' referenced before assignment")
        return (xhat if AR else xkp1), res, mv_count


    # largest n of a device-resident solve (pipeline=True).  Up to 2048 the whole solve is one launch or VALU rounds; from
    # 4096 on the batch of one takes the persistent multi-vector matvec (one pass for A.x_{k+1} and A.y_{k+1}, the same
    # kernel family as the host-driven loop's ccqp_gemv2_f64) and what is saved is the host's turn-around per iteration;
    # beyond 16384 a pass over A (> 0.3 ms) hides the host completely.
    _PIPELINE_MAX_N = 16384

    def _lockstep(self, run):
        """The whole solve as a lock-step batch of one (batch._bapgd_run); None if it does not apply or the matvec
        budget was reached first."""
        from ._record import RecordingEngine
        from .lockstep import _bapgd_run, _spec_from_items
        from .operators import DenseOperator
        op, eng = run.op, run.eng
        elementwise = hasattr(run.P, "_dev_bb_tail")
        if not (isinstance(op, DenseOperator) and op.rows == op.n <= self._PIPELINE_MAX_N
                and (elementwise or run.P._has_device_form())
                and not isinstance(eng, RecordingEngine) and eng.gemv_events is None):
            return None
        if elementwise:
            # ONE problem on an elementwise set: staging in one launch, then the form its size asks for (lockstep._bapgd_one)
            from .lockstep import _bapgd_one
            done, x, res, mv = _bapgd_one(eng, self, op, run.b, run.x0 if run.x0_given else None, run.P, run.n)
            if not done:
                from . import lockstep
                lockstep.stats["fallback_solves"] += 1
                return None
            eng.mv_executed += mv
            return x, res, mv
        # (Sphere / Cone / DisjointUnion sets: the segmented rounds, ccqp_bsegs_apgd_round_f64)
        plans = None if elementwise else [run.P._segplan(eng, run.n)]
        done, sols, res, mvs = _bapgd_run(eng, _spec_from_items(eng, [(self, op, run.b, run.x0, run.P)], run.n), run.n,
                                          self.max_matrix_vector_multiplications, 4, ar=self._anti_relaxation,
                                          plans=plans)
        if not done[0]:
            from . import lockstep
            lockstep.stats["fallback_solves"] += 1
            return None
        eng.mv_executed += int(mvs[0])
        return sols[0], res[0], int(mvs[0])


class CCQPSolverAPGD(_APGDCore):
    _label = "APGD"

    @property
    def name(self):
        return "APGD"


class CCQPSolverAPGDAntiRelaxation(_APGDCore):
    _label = "APGD"                                     # sic, solvers.py:422
    _anti_relaxation = True

    @property
    def name(self):
        return "Anti-relaxation APGD"


class _BBPGDCore(CCQPSolverBase):
    """BBPGD (solvers.py:560-693) and BBPGD with fallback (:696-843)."""
    _label = "BBPGDf"                                   # both print this, solvers.py:613,748
    _fallback = False

    def __init__(self, desired_residual_tol, max_matrix_vector_multiplications=np.inf, pipeline=None):
        super().__init__(desired_residual_tol, max_matrix_vector_multiplications)
        # SAME results bit for bit; None (default): ON where CCQPSolverBase._device_resident says so (dense operator,
        # elementwise set, budget >= 1000: 1.4x at n <= 2048, 1.17x at 4096, 1.03-1.06x at 8192 / 16384), True / False /
        # "graph" force it: keep the per-iteration scalars (BB step length, exit
        # test) on the device and queue iterations ahead instead of one host round trip per iteration
        # (ccqp_pipe_*; dense, sparse and sparse/diagonal composed operators, elementwise sets, plain solve()).  1.1-1.6x on 13-iteration solves with n <= 8k,
        # 1.7-1.9x on 125-iteration ones.  pipeline="graph" additionally replays three captured iterations as one
        # hipGraph launch: measured no faster than the direct launches (the host is no longer the bottleneck) and
        # 0.3 ms of capture + instantiation per solve -- kept for the record.
        self.pipeline = pipeline

    def _can_pipeline(self, run):
        from ._record import RecordingEngine
        from .operators import ComposedOperator, DenseOperator, SparseOperator
        op = run.op
        ok_op = isinstance(op, (DenseOperator, SparseOperator)) or \
            (isinstance(op, ComposedOperator) and op.can_pipeline())
        return (self._device_resident(run) and self.trace is None and not self._fallback and ok_op
                and hasattr(run.P, "_dev_bb_tail") and not isinstance(run.eng, RecordingEngine)
                and run.eng.gemv_events is None)

    def _pipelined(self, run, x0, g0, alpha, res, tol, max_mv):
        """The `while True` loop of _iterate with device-resident scalars; returns (x, res, mv_count)."""
        import ctypes
        eng, P, op, n = run.eng, run.P, run.op, run.n
        seg = P._seg(eng, 0, n)
        X = [x0, run.new(), run.new()]
        G = [g0, run.new(), run.new()]
        y = run.new()
        check = _lib_check
        # stream capture is not possible on the legacy default stream: run the loop on a side stream that is
        # ordered behind the caller's work, and hand the results back the same way (the blocking read at the end
        # means everything has completed before this function returns)
        caller = torch.cuda.current_stream(eng.device)
        side = None
        if self.pipeline == "graph" and caller.cuda_stream == 0:
            side = torch.cuda.Stream(device=eng.device)
            side.wait_stream(caller)
            torch.cuda.set_stream(side)
            eng.bind()
        try:
            return self._pipelined_loop(run, eng, seg, X, G, y, op, check, alpha, res, tol, max_mv)
        finally:
            if side is not None:
                torch.cuda.set_stream(caller)
                eng.bind()

    def _pipelined_loop(self, run, eng, seg, X, G, y, op, check, alpha, res, tol, max_mv):
        import ctypes
        lib, ctx, st = eng.lib, eng.ctx, eng.stream
        check(lib.ccqp_pipe_reset(ctx, float(alpha), st))
        state = (ctypes.c_double * 4)()
        ahead = 6                                           # iterations queued beyond the last finished one

        def step(k):
            new, cur = k % 3, (k - 1) % 3
            check(lib.ccqp_pipe_step_project_f64(ctx, seg, X[cur].data_ptr(), G[cur].data_ptr(), X[new].data_ptr(), st))
            op.pipe_apply(eng, X[new], y, run.b, G[new])    # A.x + b, a no-op once the slab says "converged"

        def tail(k):
            new, cur = k % 3, (k - 1) % 3
            check(lib.ccqp_pipe_bb_tail_f64(ctx, seg, X[new].data_ptr(), X[cur].data_ptr(), G[new].data_ptr(),
                                            G[cur].data_ptr(), run.c, _GD, float(tol), float(_EPS10), st))

        # iterations k+1, k+2, k+3 have the arguments of iterations 1, 2, 3 (ring of three buffers, scalars on the
        # device): capture them once as a hipGraph and replay it -- one launch instead of nine -- whenever the MV
        # cap is at least three full iterations away (pipeline="graph").
        graph = ctypes.c_void_p()
        # (capture + instantiation cost ~0.3 ms; dense operators only: the other forms allocate intermediates)
        from .operators import DenseOperator
        use_graph = self.pipeline == "graph" and isinstance(op, DenseOperator)
        k, capped = 0, False
        try:
            while True:
                if use_graph and 1 + k + 3 < max_mv:
                    if not graph:
                        check(lib.ccqp_pipe_capture_begin(ctx, st))
                        try:
                            for j in (1, 2, 3):
                                step(k + j)
                                tail(k + j)
                        finally:
                            check(lib.ccqp_pipe_capture_end(ctx, st, ctypes.byref(graph)))
                    check(lib.ccqp_pipe_graph_launch(ctx, graph, st))
                    k += 3
                else:
                    k += 1
                    step(k)
                    if 1 + k >= max_mv:                     # the reference breaks before the residual (:643-644)
                        capped = True
                        break
                    tail(k)
                lib.ccqp_pipe_poll(ctx, state)
                spins = 0
                while state[1] == 0.0 and k - state[0] >= ahead:
                    lib.ccqp_pipe_poll(ctx, state)
                    spins += 1
                    if spins > 200000:                      # mirror silent for ~0.1 s: ask the device itself
                        d_, it_, _ = eng.read(49, 3)        # (blocking; everything queued so far has then run)
                        state[0], state[1] = it_, d_
                        spins = 0
                if state[1] != 0.0:
                    break
            done, iters, res2 = eng.read(49, 3)             # blocking: the queue has drained
        finally:
            if graph:
                lib.ccqp_pipe_graph_destroy(graph)
        kf = int(iters)
        if done != 0.0:                                     # converged in iteration kf; later launches were no-ops
            eng.mv_executed += kf                           # (the two matvecs before the loop are already counted)
            return X[kf % 3], np.sqrt(res2), 1 + kf
        assert capped and kf == k - 1
        eng.mv_executed += k
        return X[k % 3], (np.sqrt(res2) if kf >= 1 else res), 1 + k

    def _lockstep_segments(self, run):
        """BBPGD with pipeline=True on a Sphere / Cone / DisjointUnion set: the whole solve on the device as a
        segmented lock-step batch of one (ccqp_bsegs_*; dense operator, n <= 2048); None where that does not apply."""
        from ._record import RecordingEngine
        from .lockstep import _bsegs_run, _spec_from_items
        from .operators import DenseOperator
        op, eng, P = run.op, run.eng, run.P
        if not (isinstance(op, DenseOperator) and op.rows == op.n <= 2048 and not hasattr(P, "_dev_bb_tail")
                and P._has_device_form() and not isinstance(eng, RecordingEngine) and eng.gemv_events is None):
            return None
        max_mv = self.max_matrix_vector_multiplications
        done, final, sols, res = _bsegs_run(eng, _spec_from_items(eng, [(self, op, run.b, run.x0, P)], run.n),
                                            [P._segplan(eng, run.n)], run.n, max_mv, 4)
        k = int(final[0])
        eng.mv_executed += 1 if (done[0] and k == 0) else 2 + k
        return sols[0], res[0], 1 + k

    def _iterate(self, run):
        tol, max_mv = self.desired_residual_tol, self.max_matrix_vector_multiplications
        eng = run.eng
        if self._device_resident(run) and self.trace is None and self.pipeline != "graph":
            # (dense, n <= 16384, elementwise set: the lock-step batch of one -- BBPGD too: 0.22-0.30 ms per solve at
            # n <= 2048 against 0.31-0.37 for the device-resident loop below, 0.57 against 0.61 at 4096; that loop stays for
            # larger, sparse and matrix-free operators)
            out = self._lockstep_one(run)
            if out is not None:
                return out
        if self._device_resident(run) and not self._fallback and self.trace is None:
            out = self._lockstep_segments(run)
            if out is not None:
                return out
        mv_count = 0
        xk = run.copy(run.x0)
        xkm1 = run.copy(run.x0)
        xmin = gmin = None
        if self._fallback:
            xmin = run.copy(run.x0)
            gmin = run.copy(run.x0)
        _, gkm1 = run.mv(xkm1, True)
        mv_count += 1
        res = yield from run.residual_now(xkm1, gkm1)
        resmin = np.inf
        if res >= tol:
            Ag = run.mv(gkm1)                           # executed, not counted (:635,:775)
            s = yield from run.dots_now([(gkm1, gkm1), (gkm1, Ag)])
            alpha = s[0] / s[1]
            if self._can_pipeline(run):
                return self._pipelined(run, xkm1, gkm1, alpha, res, tol, max_mv)
            while True:
                xk = run.step_project(xkm1, alpha, gkm1)
                _, gk = run.mv(xk, True)
                mv_count += 1
                if mv_count >= max_mv:
                    break
                if not self._fallback:
                    # speculative BB dots share the residual's sync; for elementwise sets residual and
                    # dots are ONE launch (no dx / dg vectors at all), with identical bits
                    if hasattr(run.P, "_dev_bb_tail"):
                        run.P._dev_bb_tail(eng, xk, xkm1, gk, gkm1, run.c, _GD, 0)
                    else:
                        run.residual_sq(xk, gk, 0)
                        dx = run.lin(1.0, xk, -1.0, xkm1)
                        dg = run.lin(1.0, gk, -1.0, gkm1)
                        eng.dots([(dx, dx), (dx, dg)], 1)
                    sc = yield from run.read(0, 3)
                    res = np.sqrt(sc[0])
                    if self.trace is not None:
                        self.trace.append(dict(mv=mv_count, res=float(res), alpha=float(alpha)))
                    if res < tol:
                        break
                else:
                    run.residual_sq(xk, gk, 0)
                    res = np.sqrt((yield from run.read(0, 1))[0])
                    if self.trace is not None:
                        self.trace.append(dict(mv=mv_count, res=float(res), alpha=float(alpha)))
                    if res < tol:
                        break
                    if res < resmin:                    # :793-796
                        resmin = np.copy(res)
                        xmin = run.copy(xk)
                        gmin = run.copy(gk)
                    if alpha < _EPS10:                  # :799-800
                        xk = run.step_project(xmin, _GD, gmin)
                    dx = run.lin(1.0, xk, -1.0, xkm1)
                    dg = run.lin(1.0, gk, -1.0, gkm1)
                    sc = [None] + (yield from run.dots_now([(dx, dx), (dx, dg)], 1))
                alpha = sc[1] / (sc[2] + _EPS10)
                xk, xkm1 = xkm1, xk
                gk, gkm1 = gkm1, gk
        return xk, res, mv_count


class CCQPSolverBBPGD(_BBPGDCore):
    @property
    def name(self):
        return "BBGPD"                                  # sic, solvers.py:673


class CCQPSolverBBPGDf(_BBPGDCore):
    _fallback = True

    @property
    def name(self):
        return "BBPDGf"                                 # sic, solvers.py:823


class CCQPSolverSPG(CCQPSolverBase):
    """SPG-QP (solvers.py:846-999).  One draw per iteration from numpy's GLOBAL legacy RNG
    (`np.random.uniform`, :959): seed with np.random.seed exactly as for the reference."""
    _label = "SPG"

    def __init__(self, desired_residual_tol, max_matrix_vector_multiplications=np.inf,
                 m=5, tau=0.5, sigma1=0.01, sigma2=0.5, pipeline=None):
        super().__init__(desired_residual_tol, max_matrix_vector_multiplications)
        # Extension, OFF by default, SAME results bit for bit and the same draws from the same generator (dense
        # operator, n <= 2048, any set built from the seven operators): the vector work of an iteration queued as four
        # launches and its scalar part done in C between them (ccqp_bspg_run_f64, the lock-step batch driver with a
        # batch of one) instead of ~10 ctypes calls and two scalar reads per iteration
        self.pipeline = pipeline
        self.m = m
        self.t = tau
        self.sigma1 = sigma1
        self.sigma2 = sigma2
        # extension (default = reference behaviour): set `solver.rng = np.random.RandomState(seed)` to
        # draw from a private stream instead of numpy's global one -- needed for reproducible results
        # when several SPG solves are interleaved by ccqppy_amd.batch
        self.rng = None

    @property
    def name(self):
        return "SPG-QP"

    # (1.5-1.6x the host-driven loop at n <= 1024, 1.4x at 2048, 1.22x at 4096, 1.08x at 8192:
    # profiles/r04_single_spg.jsonl)
    _PIPELINE_MAX_N = 16384

    def _lockstep(self, run):
        """pipeline=True: the whole solve through the lock-step driver as a batch of one; None where that does not
        apply (the host-driven loop below then runs, and raises what the reference raises)."""
        from ._record import RecordingEngine
        from .lockstep import _SPG_MAX_M, _bspg_one, _bspg_run, _spec_from_items
        from .operators import DenseOperator
        op, eng, P = run.op, run.eng, run.P
        rng = self.rng if self.rng is not None else np.random.mtrand._rand      # numpy's global legacy generator
        elementwise = hasattr(P, "_dev_bb_tail")
        if not (isinstance(op, DenseOperator) and op.rows == op.n <= self._PIPELINE_MAX_N
                and isinstance(rng, np.random.RandomState) and isinstance(self.m, (int, np.integer))
                and 1 <= self.m <= _SPG_MAX_M and (elementwise or P._has_device_form())
                and not isinstance(eng, RecordingEngine) and eng.gemv_events is None):
            return None
        if elementwise:
            host, x = _bspg_one(eng, self, op, run.b, run.x0 if run.x0_given else None, P, run.n, rng)
            sols = [x]
        else:
            host, sols = _bspg_run(eng, _spec_from_items(eng, [(self, op, run.b, run.x0, P)], run.n), [self], run.n,
                                   self.max_matrix_vector_multiplications, plans=[P._segplan(eng, run.n)], rngs=[rng])
        h = host[0]
        if h["status"] == 3:
            if h["used"]:                               # numpy's own OverflowError, after `used` successful draws (:959)
                np.random.RandomState(0).uniform(low=0.0, high=np.nan)
            return None                                 # nothing drawn: the loop below reproduces the exception
        mv = int(h["mv"])
        eng.mv_executed += mv - (1 if h["status"] == 2 else 0)
        return sols[0], np.sqrt(h["dd"]), mv

    def _iterate(self, run):
        tol, max_mv = self.desired_residual_tol, self.max_matrix_vector_multiplications
        eng = run.eng
        if self._device_resident(run) and self.trace is None:
            out = self._lockstep(run)
            if out is not None:
                return out
        mv_count = 0
        xk = run.copy(run.x0)
        _, gk = run.mv(xk, True)
        Agk = run.mv(gk)
        s = yield from run.dots_now([(gk, xk), (gk, gk), (gk, Agk)])
        fk = s[0]                                       # sic: g.x, :923
        alpha = s[1] / s[2]
        mv_count += 2
        tau, sig1, sig2 = self.t, self.sigma1, self.sigma2
        fk_queue = deque(maxlen=self.m)
        fk_queue.append(fk)
        dkdotdk = None
        while True:
            if hasattr(run.P, "_dev_step_project_minus"):
                dk = run.new()                              # P(xk - alpha*gk) - xk in one launch
                run.P._dev_step_project_minus(eng, xk, -alpha, gk, dk)
            else:
                tproj = run.step_project(xk, alpha, gk)
                dk = run.lin(1.0, tproj, -1.0, xk)
            Adk = run.mv(dk)
            mv_count += 1
            if mv_count >= max_mv:
                break
            dkdotdk, dkdotAdk, dkdotgk = yield from run.dots_now([(dk, dk), (dk, Adk), (dk, gk)])
            if np.sqrt(dkdotdk) <= tol:
                break
            fmax = max(fk_queue)
            xi = (fmax - fk) / dkdotAdk
            beta = -dkdotgk / dkdotAdk
            betahat = tau * beta + np.sqrt((tau ** 2) * (beta ** 2) + 2 * xi)
            betak = (self.rng or np.random).uniform(low=sig1, high=min(betahat, sig2))
            if self.trace is not None:
                self.trace.append(dict(mv=mv_count, dd=float(dkdotdk), alpha=float(alpha), betak=float(betak)))
            eng.axpby2(1.0, xk, betak, dk, xk, 1.0, gk, betak, Adk, gk)   # xk += betak*dk; gk += betak*Adk
            fk += betak * betak * dkdotgk + 0.5 * (betak ** 2) * dkdotAdk   # sic, :963
            fk_queue.append(fk)
            alpha = dkdotdk / dkdotAdk
        if dkdotdk is None:
            raise UnboundLocalError("local variable 'dkdotdk' referenced before assignment")
        return xk, np.sqrt(dkdotdk), mv_count


class CCQPSolverMPRGP(CCQPSolverBase):
    """MPRGP with BB proportioning/expansion steps (solvers.py:1002-1224); README's "MPRGP-BB"."""
    _label = "MPRGP"

    def __init__(self, desired_residual_tol, max_matrix_vector_multiplications=np.inf, pipeline=None, look_ahead=None):
        super().__init__(desired_residual_tol, max_matrix_vector_multiplications)
        # SAME results bit for bit, same reported and executed MV counts: at the A.dx matvec of an iteration (:1129 after
        # a CG step, :1163 after an expansion step) x_{k+1} and p_{k+1} are known, so A.x_{k+1} + b (:1086 of the next
        # iteration) and A.p_{k+1} (:1100) are computed by the SAME call (operator.matvec3 -> ccqp_gemv3_f64: one pass over
        # A, each product with the bits of its own matvec); after an expansion step A.x_{k+1} + b is the :1152 product
        # again, and the start's :1065 is the first :1086.  A CG step then costs one pass over A instead of three, an
        # expansion step two instead of four.  None (default): wherever the operator reads its matrix once for the three
        # (square dense, aligned, an even n >= 4096); True / False force it.
        self.look_ahead = look_ahead
        # SAME results bit for bit (dense operator of n <= 16384, elementwise set with array bounds): every branch decision on
        # the device, rounds queued ahead (ccqp_bmprgp_*, the lock-step batch scheduler with a batch of one, staging in one
        # launch); a solve that would reach the matvec budget falls back to the loop below.  None (default): ON where
        # _device_resident says so (a plain solve(), budget >= 1000): 2.0-2.4x the host-driven loop at n <= 512, 1.6x at
        # 1024, 1.24x at 2048 (profiles/r04_single_mprgp_small.jsonl); True / False force it.
        self.pipeline = pipeline

    @property
    def name(self):
        return "MPRGP"

    # largest n of a device-resident solve: x2.2 at 256, x1.6 at 1024, x1.24 at 2048 against the host-driven loop; from
    # 4096 on (the batch of one looks ahead: three products per pass, the persistent launch) x1.13 at 4096, x1.08 at 8192,
    # x1.04 at 16384 with the one-launch staging (profiles/r04_single_cap_lifted.jsonl; before it 0.91-0.98x)
    _PIPELINE_MAX_N = 16384

    def _lockstep(self, run):
        from ._record import RecordingEngine
        from .lockstep import _bmprgp_one
        from .operators import DenseOperator
        op, eng, P = run.op, run.eng, run.P
        if not (isinstance(op, DenseOperator) and op.rows == op.n <= self._PIPELINE_MAX_N
                and isinstance(P, ss._ElementwiseOp) and not P._scalar_bound_names()
                and not isinstance(eng, RecordingEngine) and eng.gemv_events is None):
            return None
        # (the default form gives the device 1000 matvecs: converging solves take tens; one that stalls short of its
        # tolerance -- SURVEY B.2: tol <= 1e-8 -- loses 1000 cheap rounds, not its whole budget, before the host loop)
        status, x, res, mv, execd = _bmprgp_one(eng, self, op, run.b, run.x0 if run.x0_given else None, P, run.n,
                                                device_cap=None if self.pipeline is True else 1000)
        if status != 1.0:
            from . import lockstep
            lockstep.stats["fallback_solves"] += 1
            return None
        eng.mv_executed += execd
        return x, res, mv

    FUSED = True    # use the fused elementwise-set kernels (bitwise identical to the literal sequence)

    def _free_mask_mul(self, run, x, g):
        """isclose(x, P(x)) * g"""
        out = run.new()
        if self._seg is not None:
            run.eng.free_gradient_elementwise(self._seg, x, g, out)
        else:
            run.eng.isclose_mask_mul(x, run.project(x), g, out)
        return out

    def _iterate(self, run):
        tol, max_mv = self.desired_residual_tol, self.max_matrix_vector_multiplications
        eng, P = run.eng, run.P
        # elementwise feasible set with array bounds: the proportioning test, the free gradient and
        # the CG direction run as fused kernels (ccqp_mprgp.hip); anything else takes the literal path
        if self._device_resident(run) and self.FUSED and self.trace is None:
            out = self._lockstep(run)
            if out is not None:
                return out
        self._seg = None
        if self.FUSED and isinstance(P, ss._ElementwiseOp) and not P._scalar_bound_names() \
                and run.n > ss._lib.SMALL_SEG_MAX:
            self._seg = P._seg(eng, 0, run.n)
        seg = self._seg
        from ._record import RecordingEngine
        look = self.look_ahead
        if look is None:
            look = hasattr(run.op, "triple_in_one_pass") and run.op.triple_in_one_pass(eng)
        # (any operator with a matvec3: the whole dense matrix, or a rank's row block with ONE exchange for the products)
        look = bool(look) and hasattr(run.op, "matvec3") and getattr(run.op, "n", None) == run.n \
            and not isinstance(eng, RecordingEngine)
        g_ahead = None                                  # A.x_k + b of the NEXT loop top, computed ahead (or in place)
        Ap_ahead = None                                 # A.p of the next :1100
        mv_count = 0
        xk = run.project(run.x0)                        # :1063
        xkp1 = run.copy(xk)
        _, gk = run.mv(xk, True)
        gkp1 = run.copy(gk)
        mv_count += 1
        res = yield from run.residual_now(xk, gk)
        if res >= tol:
            if look:
                p = self._free_mask_mul(run, xk, gk)    # (p_0 needs x_0 and g_0 only: A.g_0 and A.p_0 in one pass)
                (Agk, _), Ap_ahead, _ = run.mv3([gk, p, None], [False, False, False])
                eng.mv_executed -= 1                    # (A.p_0 counts where the reference computes it)
                g_ahead = (None, gk)                    # the first :1086 is the :1065 product again
            else:
                Agk = run.mv(gk)
            s = yield from run.dots_now([(gk, gk), (gk, Agk)])
            alpha_bb = s[0] / s[1]
            mv_count += 1
            if not look:
                p = self._free_mask_mul(run, xk, gk)
            while True:
                if g_ahead is not None:
                    gnew = g_ahead[1]
                    g_ahead = None
                    eng.mv_executed += 1
                else:
                    _, gnew = run.mv(xk, True)
                mv_count += 1
                if mv_count >= max_mv:
                    break
                gk = gnew
                psi = run.new()
                w = run.new()
                nv = run.new()
                if seg is not None:
                    txk = run.new()
                    eng.mprgp_split(seg, xk, gk, txk, psi, w, 10)        # P(xk), psi, 1-delta, |xk-P(xk)|^2
                    eng.mprgp_normal_dot(seg, txk, gk, nv, 10, 8)        # n(xk), n.gk          (:1095)
                    eng.mprgp_proportion(w, gk, nv, psi, 8, 0)           # beta.beta, psi.psi   (:1096-1098)
                    bb, pp = yield from run.read(0, 2)
                else:
                    txk = run.project(xk)
                    eng.isclose_mask_mul(xk, txk, gk, psi)
                    eng.isclose_mask_mul(xk, txk, None, w, invert=True)  # 1 - delta
                    P._dev_normal_vector(eng, xk, nv)                    # :1095
                    eng.dots([(nv, gk)], 8)
                    beta_x = run.new()
                    eng.chopped_gradient(w, gk, nv, 8, beta_x)           # :1096-1097
                    bb, pp = yield from run.dots_now([(beta_x, beta_x), (psi, psi)])
                self._nhalve = 0
                if bb < pp:
                    if Ap_ahead is not None:
                        Ap = Ap_ahead[0]
                        Ap_ahead = None
                        eng.mv_executed += 1
                    else:
                        Ap = run.mv(p)
                    mv_count += 1
                    if mv_count >= max_mv:
                        break
                    psip, pAp = yield from run.dots_now([(psi, p), (p, Ap)])
                    alpha_cg = psip / pAp
                    y = run.lin(1.0, xk, -alpha_cg, p)
                    alpha_f = yield from self._feasible_step(run, xk, p, alpha_cg + _EPS10)
                    branch = "cg" if alpha_cg <= alpha_f else "expand"
                    if alpha_cg <= alpha_f:
                        xkp1 = y                                         # np.copy(y): nothing writes in place
                        gkp1 = run.lin(1.0, gk, -alpha_cg, Ap)
                        dx = run.lin(1.0, xkp1, -1.0, xk)
                        psi_y = self._free_mask_mul(run, y, gkp1)
                        pn = run.new()
                        eng.cg_direction(psi_y, Ap, pAp, p, pn)          # psi_y - (psi_y*Ap/pAp)*p, :1134-1135
                        if look:                                         # A.dx, A.p_{k+1}, A.x_{k+1} + b: one pass
                            (Adx, _), Ap_ahead, g_ahead = run.mv3([dx, pn, xkp1], [False, False, True])
                            eng.mv_executed -= 2
                        else:
                            Adx = run.mv(dx)                             # executed, not counted (:1129)
                        eng.dots([(dx, dx), (dx, Adx)], 0)               # alpha_bb: read in the tail
                        p = pn
                    else:
                        xh = run.lin(1.0, xk, -alpha_f, p)
                        gh = run.lin(1.0, gk, -alpha_f, Ap)
                        dx = run.lin(1.0, xh, -1.0, xk)
                        dg = run.lin(1.0, gh, -1.0, gk)
                        s = yield from run.dots_now([(dx, dx), (dx, dg)])
                        alpha = s[0] / (s[1] + _EPS10)
                        xkp1 = run.step_project(xh, alpha, gh)
                        _, gkp1 = run.mv(xkp1, True)
                        mv_count += 1
                        if mv_count >= max_mv:
                            break
                        p = self._free_mask_mul(run, xkp1, gkp1)
                        dx = run.lin(1.0, xkp1, -1.0, xk)
                        if look:                                         # A.dx and A.p_{k+1}: one pass; the next :1086
                            (Adx, _), Ap_ahead, _ = run.mv3([dx, p, None], [False, False, False])   # is :1152 again
                            eng.mv_executed -= 1
                            g_ahead = (None, gkp1)
                        else:
                            Adx = run.mv(dx)                             # executed, not counted (:1163)
                        eng.dots([(dx, dx), (dx, Adx)], 0)
                else:
                    branch = "prop"
                    Ap_ahead = None                                      # (p is re-formed below: nothing to take over)
                    xkp1 = run.step_project(xk, alpha_bb, gk)
                    dx = run.lin(1.0, xkp1, -1.0, xk)
                    Adx = run.mv(dx)                                     # executed, not counted (:1172)
                    eng.dots([(dx, dx), (dx, Adx)], 0)
                    _, gk = run.mv(xk, True)                             # redundant but counted (:1174)
                    mv_count += 1
                    if mv_count >= max_mv:
                        break
                    p = self._free_mask_mul(run, xkp1, gkp1)             # STALE gkp1 (:1180-1182)
                # one sync for the BB step length of every branch (slots 0,1) and the residual (slot 2)
                run.residual_sq(xkp1, gkp1, 2)
                sc = yield from run.read(0, 3)
                alpha_bb = sc[0] / (sc[1] + _EPS10)
                res = np.sqrt(sc[2])
                if self.trace is not None:
                    self.trace.append(dict(mv=mv_count, res=float(res), branch=branch, nhalve=self._nhalve,
                                           alpha_bb=float(alpha_bb)))
                if res < tol:
                    break
                xk, xkp1 = xkp1, xk
                gk, gkp1 = gkp1, gk
        return xkp1, res, mv_count

    _STEPS = 8      # candidate steps per feasibility launch

    def _feasible_step(self, run, xk, p, alpha_f):
        """Largest alpha in {alpha_f * 0.5^k} with  all(isclose(y, P(y))), y = xk - alpha*p  -- the
        reference's halving loop (solvers.py:1112-1118).  For elementwise P, `_STEPS` candidates are
        evaluated per launch (the values visited are exactly the reference's: *0.5 is exact)."""
        eng, P = run.eng, run.P
        tried = 0
        if isinstance(P, ss._ElementwiseOp):
            seg = P._seg(eng, 0, run.n)
            while True:
                eng.feasible_steps_elementwise(seg, xk, p, alpha_f, self._STEPS, 16)
                counts = yield from run.read(16, self._STEPS)
                for c in counts:
                    if c == 0:
                        return alpha_f
                    alpha_f *= 0.5
                    self._nhalve += 1
                tried += self._STEPS
                if tried > 1200:   # alpha_f has underflowed to 0: the reference never exits either
                    raise RuntimeError("MPRGP feasibility bisection cannot terminate "
                                       "(solvers.py:1113-1118 loops forever on this input)")
        plan = P._segplan(eng, run.n) if isinstance(P, ss.ProjOpBase) else None
        while plan is not None:
            # any set with a device form: `_STEPS` candidates per pair of launches (ccqp_segs_feasible_steps_f64)
            eng.segs_feasible_steps(plan, xk, p, alpha_f, self._STEPS, 16)
            counts = yield from run.read(16, self._STEPS)
            for c in counts:
                if c == 0:
                    return alpha_f
                alpha_f *= 0.5
                self._nhalve += 1
            tried += self._STEPS
            if tried > 1200:
                raise RuntimeError("MPRGP feasibility bisection cannot terminate "
                                   "(solvers.py:1113-1118 loops forever on this input)")
        while True:
            yf = run.lin(1.0, xk, -alpha_f, p)
            eng.count_notclose(yf, run.project(yf), 9)
            if (yield from run.read(9, 1))[0] == 0:
                return alpha_f
            alpha_f *= 0.5
            self._nhalve += 1
            tried += 1
            if tried > 1200:
                raise RuntimeError("MPRGP feasibility bisection cannot terminate "
                                   "(solvers.py:1113-1118 loops forever on this input)")
