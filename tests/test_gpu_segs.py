"""GPU: the segmented projection engine (csrc/ccqp_segs.hip) -- any feasible set in <= 2 launches (normal_vector: 3).

Reference: DisjointProjOp.__call__ / normal_vector (solution_spaces.py:540-560, 512-525), Sphere (:417-435, 389-403),
Cone (:470-492), the Eq. 25 residual (solvers.py:137-139).  Bars:
  * projection / fused step / normal_vector: BITWISE equal to the round-1 per-block kernels (which the 431 operator
    goldens pin against the reference: tests/test_gpu_ops.py runs those through this engine now);
  * residual: a new (documented) summation order -> against the oracle's `np.linalg.norm` with an fp64 bound."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import cpu_ref, gen_ref
from tests import helpers as H

pytestmark = pytest.mark.gpu

# seeds of the randomised union tests: 12 / 8 / 6 in the suite; CCQP_SEGS_FUZZ_CASES=N for a one-off soak
import os  # noqa: E402
_SOAK = int(os.environ.get("CCQP_SEGS_FUZZ_CASES", "0"))


@pytest.fixture(scope="module")
def eng():
    from ccqppy_amd._engine import get_engine
    return get_engine(0)


def _legacy_project(P, eng, x):
    """The per-block path of round 1 (one thread-per-block launch + two launches per large Sphere / Cone block)."""
    from ccqppy_amd import solution_spaces as ss
    out = torch.zeros_like(x)
    if isinstance(P, ss.DisjointProjOp):
        P._block_project(eng, x, out, 0, x.numel())
    else:
        P._block_project(eng, x, out, 0, x.numel(), False)
    return out


def _count_launch_calls(eng, names):
    calls = {k: 0 for k in names}
    originals = {k: getattr(eng, k) for k in names}

    def wrap(k):
        def f(*a, **kw):
            calls[k] += 1
            return originals[k](*a, **kw)
        return f
    for k in names:
        setattr(eng, k, wrap(k))

    def restore():
        for k in names:
            delattr(eng, k)
    return calls, restore


def test_thousand_spheres_of_100_project_in_two_launches(eng):
    from ccqppy_amd import solution_spaces as ss
    P = ss.DisjointProjOp(*[ss.SphereProjOp(100)] * 1000)
    n = P.embedded_dimension
    assert n == 100000
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=torch.Generator("cuda").manual_seed(1)) * 0.2
    x[:50000] *= 0.3                                          # half of the blocks inside their sphere
    legacy = _legacy_project(P, eng, x)
    calls, restore = _count_launch_calls(eng, ["segs_project", "dots", "proj_normscale", "proj_small"])
    try:
        out = torch.zeros_like(x)
        P._dev_project(eng, x, out)
        got = P(x)
    finally:
        restore()
    assert calls == {"segs_project": 2, "dots": 0, "proj_normscale": 0, "proj_small": 0}
    plan = P._segplan(eng, n)
    assert (plan.nwork, plan.nnorm, plan.npartials) == (1000, 1000, 1000)   # 1 norm launch + 1 apply launch
    assert torch.equal(out.view(torch.int64), legacy.view(torch.int64))
    assert torch.equal(got.view(torch.int64), legacy.view(torch.int64))
    # and against the oracle (block norms differ from OpenBLAS' order by ulps)
    want = cpu_ref.Disjoint(*[cpu_ref.Sphere(100, 1)] * 1000)(x.cpu().numpy())
    assert H.ulp_diff(out.cpu().numpy(), want) <= 8 + 2 * np.sqrt(100)
    inside = np.linalg.norm(want.reshape(1000, 100) - x.cpu().numpy().reshape(1000, 100), axis=1) == 0
    assert 100 < inside.sum() < 900


def _random_union(rng, with_cone=True, arrays_only=False):
    """(product op, oracle op, n): blocks of every size class in random order."""
    from ccqppy_amd import solution_spaces as ss
    fac, ofac = C.product_factory(), C.oracle_factory()
    sizes = [1, 2, 3, 15, 16, 63, 64, 65, 100, 255, 257, 2047, 2048, 2049, 4100, 9000]
    specs, arrays = [], {}
    for i in range(int(rng.integers(3, 40))):
        d = int(rng.choice(sizes))
        kind = str(rng.choice(["Box", "Lower", "Upper", "Identity", "Sphere", "Sphere"] + (["Cone"] * 2 if with_cone else [])))
        sp = {"kind": kind, "dim": d}
        if kind in ("Box", "Lower", "Upper"):
            form = int(rng.integers(0, 3 if not arrays_only else 1))
            if form == 0:
                arrays["lb%d" % i] = -np.abs(rng.normal(size=d)) - 0.01
                arrays["ub%d" % i] = np.abs(rng.normal(size=d)) + 0.01
                sp["lb"], sp["ub"] = {"arr": "lb%d" % i}, {"arr": "ub%d" % i}
            elif form == 1:
                sp["lb"], sp["ub"] = {"scalar": -0.3}, {"scalar": 0.4}
            # form 2: defaults
        elif kind == "Sphere":
            sp["radius"] = float(rng.choice([0.05, 1.0, 7.5])) if rng.random() < 0.7 else None
        elif kind == "Cone":
            sp["aspect_ratio"] = float(rng.choice([0.5, 1.0, 3.0])) if rng.random() < 0.7 else None
        if rng.random() < 0.15:
            sp = {"kind": "Disjoint", "ops": [sp, {"kind": "Sphere", "dim": int(rng.choice([3, 70]))}]}
        specs.append(sp)
    spec = {"kind": "Disjoint", "ops": specs}
    return C.build_op(spec, fac, arrays), C.build_op(spec, ofac, arrays), C.spec_dim(spec), ss


@pytest.mark.parametrize("seed", range(_SOAK or 12))
def test_random_unions_bitwise_equal_to_the_per_block_kernels(seed, eng):
    rng = np.random.default_rng(1000 + seed)
    P, Po, n, ss = _random_union(rng)
    scale = [0.02, 1.0, 5.0][seed % 3]
    xh = rng.normal(size=n) * scale
    x = eng.to_device(xh)
    legacy = _legacy_project(P, eng, x)
    out = torch.zeros_like(x)
    P._dev_project(eng, x, out)
    assert torch.equal(out.view(torch.int64), legacy.view(torch.int64))
    want = Po(xh)
    maxd = max(op.embedded_dimension for op, _ in P._leaves())
    assert H.ulp_diff(out.cpu().numpy(), want) <= 8 + 2 * np.sqrt(maxd)
    # fused step P(a*x + b*y): the axpby is formed on the fly with the roundings of the separate kernel
    yh = rng.normal(size=n)
    y = eng.to_device(yh)
    for a, b in ((1.0, -0.37), (1.25, 0.5)):
        t = eng.empty(n)
        eng.axpby(a, x, b, y, t)
        ref = _legacy_project(P, eng, t)
        got = torch.zeros_like(x)
        P._dev_step_project(eng, a, x, b, y, got)
        assert torch.equal(got.view(torch.int64), ref.view(torch.int64))
    # SPG's direction P(x + b*y) - x
    t = eng.empty(n)
    eng.axpby(1.0, x, -0.2, y, t)
    ref = _legacy_project(P, eng, t)
    d_ref = eng.empty(n)
    eng.axpby(1.0, ref, -1.0, x, d_ref)
    d_got = eng.empty(n)
    P._dev_step_project_minus(eng, x, -0.2, y, d_got)
    assert torch.equal(d_got.view(torch.int64), d_ref.view(torch.int64))
    # Eq. 25 residual: against the oracle's norm (new summation order: per work item, items in index order)
    c, gd = 1.0 / (3 * n * 1e-6), 1e-6
    P._dev_residual_sq(eng, x, y, c, gd, 5)
    got_r2 = float(eng.read(5, 1)[0])
    v = c * (xh - Po(xh - gd * yh))
    want_r2 = float(v.dot(v))
    assert abs(got_r2 - want_r2) <= 1e-12 * want_r2 + 1e-300
    # determinism: the same bits again
    P._dev_residual_sq(eng, x, y, c, gd, 6)
    assert float(eng.read(6, 1)[0]) == got_r2


@pytest.mark.parametrize("seed", range(_SOAK or 8))
def test_random_unions_normal_vector_bitwise_equal_to_the_per_block_kernels(seed, eng):
    rng = np.random.default_rng(2000 + seed)
    P, Po, n, ss = _random_union(rng, with_cone=False, arrays_only=True)
    xh = rng.normal(size=n) * 0.3
    # put whole blocks ON their sets so that non-zero normals occur: project first, then nudge within isclose
    xh = Po(xh)
    xh[::7] += 2e-9
    x = eng.to_device(xh)
    legacy = torch.zeros_like(x)
    P._legacy_normal_vector(eng, x, legacy)
    got = torch.zeros_like(x)
    calls, restore = _count_launch_calls(eng, ["segs_normal", "proj_small", "normal_elementwise", "normal_sphere"])
    try:
        P._dev_normal_vector(eng, x, got)
    finally:
        restore()
    assert calls["segs_normal"] == 1 and calls["proj_small"] == calls["normal_sphere"] == 0
    assert torch.equal(got.view(torch.int64), legacy.view(torch.int64))
    want = Po.normal_vector(xh)
    maxd = max(op.embedded_dimension for op, _ in P._leaves())
    assert H.ulp_diff(got.cpu().numpy(), want) <= 8 + 2 * np.sqrt(maxd)
    assert np.count_nonzero(want) > 0
    # public entry point, numpy in / numpy out
    assert H.same_bits(P.normal_vector(xh), got.cpu().numpy())


@pytest.mark.parametrize("n", [1, 2, 15, 64, 65, 2048, 2049, 5000, 131072, 2048 * 1024 + 5])
@pytest.mark.parametrize("kind", ["Sphere", "Cone"])
def test_standalone_norm_sets_match_the_two_launch_path(kind, n, eng):
    """Sphere / Cone over a whole vector (config 3's set at n = 131072; beyond 1024 chunks the chunk size grows)."""
    from ccqppy_amd import solution_spaces as ss
    rng = np.random.default_rng(n)
    P = ss.SphereProjOp(n, 0.5) if kind == "Sphere" else ss.ConeProjOp(n, 0.5)
    for scale in (1e-4, 1.0):
        x = eng.to_device(rng.normal(size=n) * scale)
        legacy = _legacy_project(P, eng, x)
        out = torch.zeros_like(x)
        P._dev_project(eng, x, out)
        assert torch.equal(out.view(torch.int64), legacy.view(torch.int64))
        g = eng.to_device(rng.normal(size=n))
        t = eng.empty(n)
        eng.axpby(1.0, x, -0.01, g, t)
        ref = _legacy_project(P, eng, t)
        got = torch.zeros_like(x)
        P._dev_step_project(eng, 1.0, x, -0.01, g, got)
        assert torch.equal(got.view(torch.int64), ref.view(torch.int64))
    if kind == "Sphere":
        xs = eng.to_device(rng.normal(size=n))
        xs = xs * (0.5 / float(xs.norm()))
        a, b = torch.zeros_like(xs), torch.zeros_like(xs)
        P._legacy_normal_vector(eng, xs, a)
        P._dev_normal_vector(eng, xs, b)
        assert torch.equal(a.view(torch.int64), b.view(torch.int64))


def test_union_with_a_user_callable_block_takes_the_per_block_path(eng):
    """Blocks may be anything callable with an `embedded_dimension` (solution_spaces.py:555-558)."""
    from ccqppy_amd import solution_spaces as ss

    class Clamp:
        embedded_dimension = 70

        def __call__(self, x):
            return np.clip(x, -0.1, 0.1)

    P = ss.DisjointProjOp(ss.SphereProjOp(100), Clamp(), ss.BoxProjOp(5))
    rng = np.random.default_rng(0)
    x = rng.normal(size=175)
    want = np.concatenate([cpu_ref.Sphere(100, 1)(x[:100]), np.clip(x[100:170], -0.1, 0.1),
                           cpu_ref.Box(5, -np.ones(5), np.ones(5))(x[170:])])
    assert H.ulp_diff(P(x), want) <= 8 + 2 * np.sqrt(100)
    assert P._segplan(eng, 175) is None


@pytest.mark.parametrize("sname", ["BBPGD", "SPG", "APGD", "MPRGP"])
def test_solvers_on_a_union_of_large_blocks_against_the_oracle(sname, eng):
    """n = 5000: 30 x Sphere(100) + Box(2000) -- every projection, fused step and residual of the solve goes through
    the segmented engine; MPRGP also through its normal_vector.  Same MV counts, x within 1e-10."""
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    n = 5000
    A = gen_ref.spd_rows(n, 77)
    b = -A.dot(gen_ref.xstar(n, 77))
    lb, ub = -0.5 * np.ones(2000), 0.5 * np.ones(2000)
    P = ss.DisjointProjOp(*([ss.SphereProjOp(100, 2.0)] * 30 + [ss.BoxProjOp(2000, lb, ub)]))
    Po = cpu_ref.Disjoint(*([cpu_ref.Sphere(100, 2.0)] * 30 + [cpu_ref.Box(2000, lb, ub)]))
    tol, mx = (1e-6, 300) if sname == "MPRGP" else (1e-8, 3000)
    np.random.seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        r = getattr(solvers, "CCQPSolver" + sname)(tol, mx).solve(problems.spd_rows(n, 77), b, convex_proj_op=P)
    np.random.seed(11)
    with np.errstate(all="ignore"):
        o = cpu_ref.SOLVERS[sname](A, b, tol, mx, P=Po)
    assert r.solution_num_matrix_vector_multiplications == o.solution_num_matrix_vector_multiplications
    assert r.executed_matrix_vector_multiplications == o.executed_mv
    assert bool(r.solution_converged) == bool(o.solution_converged)
    assert np.linalg.norm(r.solution - o.solution) <= 1e-10 * np.linalg.norm(o.solution)


@pytest.mark.parametrize("seed", range(_SOAK or 6))
def test_feasible_step_candidates_in_two_launches_equal_one_by_one_evaluation(seed, eng):
    """MPRGP's halving loop (solvers.py:1112-1118) on a non-elementwise set: K candidates per call of
    ccqp_segs_feasible_steps_f64 against the reference's form -- one projection and one count per candidate."""
    rng = np.random.default_rng(3000 + seed)
    P, Po, n, ss = _random_union(rng, with_cone=False)
    xh = Po(rng.normal(size=n) * 0.5)                     # a feasible point
    ph = rng.normal(size=n) * [1e-3, 1.0, 50.0][seed % 3]
    x, p = eng.to_device(xh), eng.to_device(ph)
    plan = P._segplan(eng, n)
    alpha0 = 0.7
    K = 16
    eng.segs_feasible_steps(plan, x, p, alpha0, K, 16)
    got = [int(v) for v in eng.read(16, K)]
    want, a = [], alpha0
    for k in range(K):
        yf = xh - a * ph
        want.append(int(np.count_nonzero(~np.isclose(yf, Po(yf)))))
        a *= 0.5
    # (a Sphere block's norm differs from OpenBLAS' by ulps: an entry exactly on the isclose threshold may flip; the
    # counts agree wherever no entry sits within 1e-12 of that threshold -- all of them, in practice)
    assert got == want
    # and against the device's own one-by-one path
    a = alpha0
    for k in range(K):
        yf = eng.empty(n)
        eng.axpby(1.0, x, -a, p, yf)
        pr = torch.zeros_like(yf)
        P._dev_project(eng, yf, pr)
        eng.count_notclose(yf, pr, 9)
        assert int(eng.read(9, 1)[0]) == got[k]
        a *= 0.5
