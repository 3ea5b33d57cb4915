"""CPU oracle for the CCQP hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product package `ccqppy_amd` never imports it and has
no CPU fallback: it fails loudly when the HIP library is missing.

This is a numpy restatement, written from the behavioural description in
SURVEY.md Appendix A, of the reference's per-iteration algorithms:

  projections  : /root/reference/src/ccqppy/solution_spaces.py
  solver loops : /root/reference/src/ccqppy/solvers.py

Every function cites the reference lines it follows.  Elementwise expressions
keep the reference's operation order (numpy rounds after every binary op, no
FMA contraction), so on the same host / same numpy the oracle reproduces the
reference bit for bit.  Parity status: PINNED -- `oracle/gen_golden.py` imports
the reference in the build container, checks this file against it and writes
the fixtures under `tests/golden/` that `tests/test_oracle_golden.py` replays.

The dense matvec / dot arithmetic itself is not in the reference tree: it is
numpy -> OpenBLAS (`dgemv`, `ddot`; unpinned, reference `setup.py:42`).  The
oracle calls the same numpy entry points (`A.dot`, `x.dot`, `np.linalg.norm`).
"""
from collections import deque

import numpy as np

EPS10 = 10 * np.finfo(float).eps  # solvers.py:656,799,806,1112,1129,1147,1163,1172
GD = 1e-6                         # solvers.py:137,321,496,629,768,1070


# --------------------------------------------------------------------------
# projection operators (solution_spaces.py)
# --------------------------------------------------------------------------
class _Op:
    kind = "?"

    def __init__(self, dim):
        self.dim = dim

    @property
    def embedded_dimension(self):
        return self.dim

    def _outside(self, x, xp):
        # `not np.isclose(np.linalg.norm(x - x_proj), 0)`  solution_spaces.py:153,229,313,396
        return not np.isclose(np.linalg.norm(x - xp), 0)


def _chop(g_i, n_i):
    # `g[i] - np.min((normal[i]*g[i], 0))*(normal[i])`  solution_spaces.py:180,256,343
    return g_i - np.min((n_i * g_i, 0)) * (n_i)


class Identity(_Op):
    """solution_spaces.py:77-125: returns the argument object itself."""
    kind = "Identity"

    def projected_gradient(self, x, g):  # :100-109: the body is a docstring only
        return None

    def __call__(self, x):
        return x

    def normal_vector(self, x):  # :92-98
        return np.zeros(self.dim)


def _need_indexable(bound):
    # the reference indexes `self.lower_bound[i]` (solution_spaces.py:158,234,318,320):
    # a scalar bound raises TypeError there.
    if np.ndim(bound) == 0:
        raise TypeError("'%s' object is not subscriptable" % type(bound).__name__)


class LowerBound(_Op):
    """solution_spaces.py:128-201 (default bound -1, :134)."""
    kind = "Lower Bound"

    def __init__(self, dim, lower_bound=None):
        super().__init__(dim)
        self.lower_bound = -np.ones(dim) if lower_bound is None else lower_bound

    def __call__(self, x):  # :200-201
        m = x < self.lower_bound
        return self.lower_bound * m + x * (1 - m)

    def normal_vector(self, x):  # :146-160
        xp = self(x)
        if self._outside(x, xp):
            return np.zeros(self.dim)
        _need_indexable(self.lower_bound)
        nv = np.zeros(self.dim)
        nv[np.isclose(xp, self.lower_bound)] = -1
        return nv

    def projected_gradient(self, x, g):  # :162-184
        free_g = np.zeros(self.dim)
        chopped_g = np.zeros(self.dim)
        normal = self.normal_vector(x)
        for i in range(self.dim):
            if np.isclose(x[i], self.lower_bound[i]):
                chopped_g[i] = _chop(g[i], normal[i])
            else:
                free_g[i] = g[i]
        return (free_g, chopped_g)


class UpperBound(_Op):
    """solution_spaces.py:204-277 (default bound +1, :210)."""
    kind = "Upper Bound"

    def __init__(self, dim, upper_bound=None):
        super().__init__(dim)
        self.upper_bound = np.ones(dim) if upper_bound is None else upper_bound

    def __call__(self, x):  # :276-277
        m = x > self.upper_bound
        return self.upper_bound * m + x * (1 - m)

    def normal_vector(self, x):  # :222-236
        xp = self(x)
        if self._outside(x, xp):
            return np.zeros(self.dim)
        _need_indexable(self.upper_bound)
        nv = np.zeros(self.dim)
        nv[np.isclose(xp, self.upper_bound)] = 1
        return nv

    def projected_gradient(self, x, g):  # :238-260
        free_g = np.zeros(self.dim)
        chopped_g = np.zeros(self.dim)
        normal = self.normal_vector(x)
        for i in range(self.dim):
            if np.isclose(x[i], self.upper_bound[i]):
                chopped_g[i] = _chop(g[i], normal[i])
            else:
                free_g[i] = g[i]
        return (free_g, chopped_g)


class Box(_Op):
    """solution_spaces.py:280-366 (defaults -1 / +1, :286,291)."""
    kind = "Box"

    def __init__(self, dim, lower_bound=None, upper_bound=None):
        super().__init__(dim)
        self.lower_bound = -np.ones(dim) if lower_bound is None else lower_bound
        self.upper_bound = np.ones(dim) if upper_bound is None else upper_bound

    def __call__(self, x):  # :363-366, literal left-to-right expression
        hi = x > self.upper_bound
        lo = x < self.lower_bound
        return self.lower_bound * lo + self.upper_bound * hi + x * (1 - hi) * (1 - lo)

    def normal_vector(self, x):  # :306-322 ("outside => zeros", upper bound tested first)
        xp = self(x)
        if self._outside(x, xp):
            return np.zeros(self.dim)
        _need_indexable(self.upper_bound)
        _need_indexable(self.lower_bound)
        nv = np.zeros(self.dim)
        at_ub = np.isclose(xp, self.upper_bound)
        at_lb = np.isclose(xp, self.lower_bound)
        nv[at_lb] = -1
        nv[at_ub] = 1  # `if ub ... elif lb`: ub wins
        return nv

    def projected_gradient(self, x, g):  # :324-347, incl. the misplaced parenthesis of :340-341
        free_g = np.zeros(self.dim)
        chopped_g = np.zeros(self.dim)
        normal = self.normal_vector(x)
        for i in range(self.dim):
            ub_i, lb_i = self.upper_bound[i], self.lower_bound[i]
            # the reference writes  isclose(x, lb or x < ub):  Python `or` picks lb unless lb == 0
            second = lb_i or x[i] < ub_i
            if np.isclose(x[i], ub_i) or x[i] > ub_i or np.isclose(x[i], second):
                chopped_g[i] = _chop(g[i], normal[i])
            else:
                free_g[i] = g[i]
        return (free_g, chopped_g)


class Sphere(_Op):
    """solution_spaces.py:369-435 (default radius 1, :375)."""
    kind = "Sphere"

    def __init__(self, dim, radius=None):
        super().__init__(dim)
        self.radius = 1 if radius is None else radius

    def __call__(self, x):  # :431-435
        r = np.linalg.norm(x)
        if r > self.radius:
            return self.radius * x / r
        return x

    def normal_vector(self, x):  # :389-403
        xp = self(x)
        if self._outside(x, xp):
            return np.zeros(self.dim)
        nv = np.zeros(self.dim)
        length = np.linalg.norm(xp)
        if np.isclose(length, self.radius):
            nv = xp / length
        return nv

    def projected_gradient(self, x, g):  # :405-415 (sic: the message says "Cone")
        raise NotImplementedError("Cone proximal gradient not implemented, yet.")


class Cone(_Op):
    """solution_spaces.py:438-492.  The reference marks this op "bugged" (:439): the
    norm runs over the WHOLE vector incl. the last entry (:484).  Reproduced as is."""
    kind = "Cone"

    def __init__(self, dim, aspect_ratio=None):
        super().__init__(dim)
        self.aspect_ratio = 1 if aspect_ratio is None else aspect_ratio

    def __call__(self, x):  # :484-492
        mu = self.aspect_ratio
        r = np.linalg.norm(x)
        if mu * x[-1] >= r:
            return x
        if -x[-1] / mu >= r:
            return np.zeros_like(x)
        return (x[-1] + mu * r) / (mu ** 2 + 1) * np.concatenate((x[:-1] / r, [-mu]))

    def normal_vector(self, x):  # :465
        raise NotImplementedError("Cone normal not implemented, yet.")


class Disjoint(_Op):
    """solution_spaces.py:495-560: consecutive slices, running offsets, fresh zeros output."""
    kind = "DisjointUnion"

    def __init__(self, *ops):
        self.proj_ops = ops
        super().__init__(sum(op.embedded_dimension for op in ops))

    def _walk(self):
        start = 0
        for op in self.proj_ops:
            d = op.embedded_dimension
            yield op, start, start + d
            start += d

    def __call__(self, x):  # :553-560
        out = np.zeros(self.dim)
        for op, s, e in self._walk():
            out[s:e] = op(x[s:e])
        return out

    def normal_vector(self, x):  # :512-525
        out = np.zeros(self.dim)
        for op, s, e in self._walk():
            out[s:e] = op.normal_vector(x[s:e])
        return out

    def projected_gradient(self, x, g):  # :527-538
        free_g = np.zeros(self.dim)
        chopped_g = np.zeros(self.dim)
        for op, s, e in self._walk():
            f_g, c_g = op.projected_gradient(x[s:e], g[s:e])
            free_g[s:e] = f_g
            chopped_g[s:e] = c_g
        return (free_g, chopped_g)


# --------------------------------------------------------------------------
# solver loops (solvers.py)
# --------------------------------------------------------------------------
class Result:
    """The six result fields of CCQPSolverBase (solvers.py:45-68)."""

    def __init__(self, x, res, mv, max_mv, executed_mv, trace):
        self.solution = np.copy(x)
        self.solution_residual = res
        self.solution_num_matrix_vector_multiplications = mv
        self.solution_converged = mv < max_mv  # nothing else, solvers.py:164
        self.executed_mv = executed_mv         # counted + uncounted call sites
        self.trace = trace


class _MV:
    """Counts EXECUTED matvecs (the reference's own counter skips some call sites)."""

    def __init__(self, A):
        self.A = A
        self.executed = 0

    def __call__(self, v):
        self.executed += 1
        return self.A.dot(v)


def _residual(P, n, x, g):
    # Eq. 25 of Mazhar 2015 as coded: solvers.py:138-139 (and every other solver)
    return np.linalg.norm(1.0 / (3 * n * GD) * (x - P(x - GD * g)))


def _setup(b, x0, P):
    n = b.shape[0]
    if P is None:
        P = Identity(n)            # solvers.py:115-116
    if x0 is None:
        x0 = np.zeros(n)           # solvers.py:125-126
    return n, x0, P


def solve_pgd(A, b, tol, max_mv=np.inf, step_size=0.01, x0=None, P=None, trace=None):
    """solvers.py:114-170."""
    n, x0, P = _setup(b, x0, P)
    mvp = _MV(A)
    mv = 0
    xk = np.copy(x0)
    xkm1 = np.copy(x0)
    gkm1 = mvp(xkm1) + b
    mv += 1
    res = _residual(P, n, xkm1, gkm1)
    if res >= tol:
        while True:
            xk = P(xkm1 - step_size * gkm1)
            gk = mvp(xk) + b
            mv += 1
            if mv >= max_mv:
                break
            res = _residual(P, n, xk, gk)
            if trace is not None:
                trace.append(dict(mv=mv, res=float(res)))
            if res < tol:
                break
            xk, xkm1 = xkm1, xk
            gk, gkm1 = gkm1, gk
    return Result(xk, res, mv, max_mv, mvp.executed, trace)


def _apgd_core(A, b, tol, max_mv, x0, P, trace, anti_relaxation):
    """solvers.py:242-343 (APGD) and :415-533 (anti-relaxation variant)."""
    n, x0, P = _setup(b, x0, P)
    mvp = _MV(A)
    mv = 0
    xk = np.copy(x0)
    yk = np.copy(x0)
    xhat = np.ones(n)              # :432 (AR only; APGD uses np.ones inline :263)
    thk = 1.0
    thk1 = 1.0
    d = xk - np.ones(n)
    L = np.linalg.norm(mvp(d)) / np.linalg.norm(d)
    mv += 1
    t = 1.0 / L
    resmin = np.inf
    xkp1 = None
    res = None
    while True:
        Ayk = mvp(yk)
        mv += 1
        if mv >= max_mv:
            break
        gk = Ayk + b
        xkp1 = P(yk - t * gk)
        r1 = yk.dot(Ayk) * 0.5
        r2 = yk.dot(b)
        nback = 0
        while True:
            Ax1 = mvp(xkp1)
            mv += 1
            if mv >= max_mv:
                break
            l1 = xkp1.dot(Ax1) * 0.5
            l2 = xkp1.dot(b)
            d = xkp1 - yk
            r3 = gk.dot(d)
            r4 = 0.5 * L * d.dot(d)
            if (l1 + l2) <= (r1 + r2 + r3 + r4):
                break
            L *= 2
            t = 1.0 / L
            xkp1 = P(yk - t * gk)
            nback += 1
        thk1 = 0.5 * (-thk * thk + thk * np.sqrt(4 + thk * thk))
        beta = thk * (1 - thk) / (thk * thk + thk1)
        ykp1 = (1 + beta) * xkp1 - beta * xk
        res = _residual(P, n, xkp1, Ax1 + b)
        if trace is not None:
            trace.append(dict(mv=mv, res=float(res), L=float(L), nback=nback))
        if anti_relaxation:
            if res < resmin:       # :501-503
                resmin = res
                xhat = np.copy(xkp1)
        if res < tol:
            break
        if anti_relaxation:
            if gk.dot(xkp1 - xk) > 0:   # :510-512
                ykp1 = np.copy(xkp1)
                thk1 = 1
        L *= 0.9
        t = 1.0 / L
        yk, ykp1 = ykp1, yk
        xk, xkp1 = xkp1, xk         # after this `xkp1` names the OLDER iterate (A.0)
        thk = thk1
    if res is None:
        # the reference dies here with UnboundLocalError (max_mv <= 2)
        raise UnboundLocalError("local variable 'res' referenced before assignment")
    sol = xhat if anti_relaxation else xkp1
    return Result(sol, res, mv, max_mv, mvp.executed, trace)


def solve_apgd(A, b, tol, max_mv=np.inf, x0=None, P=None, trace=None):
    return _apgd_core(A, b, tol, max_mv, x0, P, trace, False)


def solve_apgd_ar(A, b, tol, max_mv=np.inf, x0=None, P=None, trace=None):
    return _apgd_core(A, b, tol, max_mv, x0, P, trace, True)


def _bbpgd_core(A, b, tol, max_mv, x0, P, trace, fallback):
    """solvers.py:606-669 (BBPGD) and :741-819 (BBPGDf)."""
    n, x0, P = _setup(b, x0, P)
    mvp = _MV(A)
    mv = 0
    xk = np.copy(x0)
    xkm1 = np.copy(x0)
    xmin = np.copy(x0)
    gmin = np.copy(x0)
    gkm1 = mvp(xkm1) + b
    mv += 1
    res = _residual(P, n, xkm1, gkm1)
    resmin = np.inf
    if res >= tol:
        alpha = gkm1.dot(gkm1) / (gkm1.dot(mvp(gkm1)))   # MV not counted (:635,:775)
        while True:
            xk = P(xkm1 - alpha * gkm1)
            gk = mvp(xk) + b
            mv += 1
            if mv >= max_mv:
                break
            res = _residual(P, n, xk, gk)
            if trace is not None:
                trace.append(dict(mv=mv, res=float(res), alpha=float(alpha)))
            if res < tol:
                break
            if fallback:
                if res < resmin:                      # :793-796
                    resmin = np.copy(res)
                    xmin = np.copy(xk)
                    gmin = np.copy(gk)
                if alpha < EPS10:                     # :799-800
                    xk = P(xmin - GD * gmin)
            dx = xk - xkm1
            dg = gk - gkm1
            alpha = dx.dot(dx) / (dx.dot(dg) + EPS10)
            xk, xkm1 = xkm1, xk
            gk, gkm1 = gkm1, gk
    return Result(xk, res, mv, max_mv, mvp.executed, trace)


def solve_bbpgd(A, b, tol, max_mv=np.inf, x0=None, P=None, trace=None):
    return _bbpgd_core(A, b, tol, max_mv, x0, P, trace, False)


def solve_bbpgdf(A, b, tol, max_mv=np.inf, x0=None, P=None, trace=None):
    return _bbpgd_core(A, b, tol, max_mv, x0, P, trace, True)


def solve_spg(A, b, tol, max_mv=np.inf, m=5, tau=0.5, sigma1=0.01, sigma2=0.5,
              x0=None, P=None, trace=None):
    """solvers.py:906-975.  Draws one `np.random.uniform` per iteration from numpy's
    GLOBAL legacy RNG (:959), exactly like the reference."""
    n, x0, P = _setup(b, x0, P)
    mvp = _MV(A)
    mv = 0
    xk = np.copy(x0)
    gk = mvp(xk) + b
    fk = np.dot(gk, xk)                                 # sic (:923)
    alpha = gk.dot(gk) / (gk.dot(mvp(gk)))
    mv += 2
    queue = deque(maxlen=m)
    queue.append(fk)
    dd = None
    while True:
        dk = P(xk - alpha * gk) - xk
        Adk = mvp(dk)
        mv += 1
        if mv >= max_mv:
            break
        dd = np.dot(dk, dk)
        dAd = np.dot(dk, Adk)
        dg = np.dot(dk, gk)
        if np.sqrt(dd) <= tol:
            break
        fmax = max(queue)
        xi = (fmax - fk) / dAd
        beta = -dg / dAd
        betahat = tau * beta + np.sqrt((tau ** 2) * (beta ** 2) + 2 * xi)
        betak = np.random.uniform(low=sigma1, high=min(betahat, sigma2))
        if trace is not None:
            trace.append(dict(mv=mv, dd=float(dd), alpha=float(alpha), betak=float(betak)))
        xk += betak * dk
        gk += betak * Adk
        fk += betak * betak * dg + 0.5 * (betak ** 2) * dAd  # sic (:963)
        queue.append(fk)
        alpha = dd / dAd
    if dd is None:
        raise UnboundLocalError("local variable 'dkdotdk' referenced before assignment")
    return Result(xk, np.sqrt(dd), mv, max_mv, mvp.executed, trace)


def solve_mprgp(A, b, tol, max_mv=np.inf, x0=None, P=None, trace=None):
    """solvers.py:1048-1200 (what README/BASELINE call "MPRGP-BB")."""
    n, x0, P = _setup(b, x0, P)
    mvp = _MV(A)
    mv = 0
    xk = P(x0)
    xkp1 = np.copy(xk)
    gk = mvp(xk) + b
    gkp1 = np.copy(gk)
    mv += 1
    res = _residual(P, n, xk, gk)
    if res >= tol:
        alpha_bb = gk.dot(gk) / (gk.dot(mvp(gk)))
        mv += 1
        p = np.isclose(xk, P(xk)) * gk
        while True:
            Axk = mvp(xk)
            mv += 1
            if mv >= max_mv:
                break
            gk = Axk + b
            delta = np.isclose(xk, P(xk))
            psi = delta * gk
            nv = P.normal_vector(xk)
            beta_x = (1 - delta) * (gk - np.min([0, nv.dot(gk)]) * nv)
            if beta_x.dot(beta_x) < psi.dot(psi):
                Ap = mvp(p)
                mv += 1
                if mv >= max_mv:
                    break
                pAp = p.dot(Ap)
                alpha_cg = psi.dot(p) / pAp
                y = xk - alpha_cg * p
                alpha_f = alpha_cg + EPS10
                nhalve = 0
                while True:                             # :1113-1118
                    yf = xk - alpha_f * p
                    if np.all(np.isclose(yf, P(yf))):
                        break
                    alpha_f *= 0.5
                    nhalve += 1
                if alpha_cg <= alpha_f:
                    branch = "cg"
                    xkp1 = np.copy(y)
                    gkp1 = gk - alpha_cg * Ap
                    dx = xkp1 - xk
                    alpha_bb = dx.dot(dx) / (dx.dot(mvp(dx)) + EPS10)   # uncounted MV (:1129)
                    psi_y = np.isclose(y, P(y)) * gkp1
                    beta = psi_y * Ap / p.dot(Ap)       # elementwise VECTOR (:1134)
                    p = psi_y - beta * p
                else:
                    branch = "expand"
                    xh = xk - alpha_f * p
                    gh = gk - alpha_f * Ap
                    dx = xh - xk
                    dg = gh - gk
                    alpha = dx.dot(dx) / (dx.dot(dg) + EPS10)
                    xkp1 = P(xh - alpha * gh)
                    gkp1 = mvp(xkp1) + b
                    mv += 1
                    if mv >= max_mv:
                        break
                    p = np.copy(np.isclose(xkp1, P(xkp1)) * gkp1)
                    dx = xkp1 - xk
                    alpha_bb = dx.dot(dx) / (dx.dot(mvp(dx)) + EPS10)   # uncounted MV (:1163)
            else:
                branch = "prop"
                nhalve = 0
                xkp1 = P(xk - alpha_bb * gk)
                dx = xkp1 - xk
                alpha_bb = dx.dot(dx) / (dx.dot(mvp(dx)) + EPS10)       # uncounted MV (:1172)
                gk = mvp(xk) + b                        # redundant, counted (:1174)
                mv += 1
                if mv >= max_mv:
                    break
                p = np.copy(np.isclose(xkp1, P(xkp1)) * gkp1)           # STALE gkp1 (:1180-1182)
            res = _residual(P, n, xkp1, gkp1)
            if trace is not None:
                trace.append(dict(mv=mv, res=float(res), branch=branch, nhalve=nhalve, alpha_bb=float(alpha_bb)))
            if res < tol:
                break
            xk, xkp1 = xkp1, xk
            gk, gkp1 = gkp1, gk
    return Result(xkp1, res, mv, max_mv, mvp.executed, trace)


SOLVERS = {
    "PGD": solve_pgd,
    "APGD": solve_apgd,
    "APGDAntiRelaxation": solve_apgd_ar,
    "BBPGD": solve_bbpgd,
    "BBPGDf": solve_bbpgdf,
    "SPG": solve_spg,
    "MPRGP": solve_mprgp,
}
