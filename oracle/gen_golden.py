#!/usr/bin/env python3
"""Golden-vector generator  --  TEST INFRASTRUCTURE; runs ONLY in the build container.

Imports the reference read-only from /root/reference (never copied, never shipped):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python3 oracle/gen_golden.py

For every case of `oracle/cases.py` it
  1. runs the REFERENCE (`ccqppy.solution_spaces` / `ccqppy.solvers`),
  2. runs the CPU oracle `oracle/cpu_ref.py` on the same inputs,
  3. insists the two agree BIT FOR BIT (values, MV counts, residual, converged flag,
     exception type), i.e. pins the oracle,
  4. writes inputs + the reference's outputs to tests/golden/*.npz|json.
The fixtures are data only (inputs and expected outputs).
"""
import contextlib
import zlib
import io
import json
import os
import sys
import time

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import warnings  # noqa: E402

warnings.filterwarnings("ignore", category=SyntaxWarning)
import numpy as np  # noqa: E402

from ccqppy import solution_spaces as ref_ss  # noqa: E402  (the reference)
from ccqppy import solvers as ref_solvers  # noqa: E402

from oracle import cases as C  # noqa: E402
from oracle import cpu_ref, gen_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

REF_FACTORY = {"Identity": ref_ss.IdentityProjOp, "Lower": ref_ss.LowerBoundProjOp,
               "Upper": ref_ss.UpperBoundProjOp, "Box": ref_ss.BoxProjOp,
               "Sphere": ref_ss.SphereProjOp, "Cone": ref_ss.ConeProjOp,
               "Disjoint": ref_ss.DisjointProjOp}
REF_SOLVERS = {"PGD": ref_solvers.CCQPSolverPGD, "APGD": ref_solvers.CCQPSolverAPGD,
               "APGDAntiRelaxation": ref_solvers.CCQPSolverAPGDAntiRelaxation,
               "BBPGD": ref_solvers.CCQPSolverBBPGD, "BBPGDf": ref_solvers.CCQPSolverBBPGDf,
               "SPG": ref_solvers.CCQPSolverSPG, "MPRGP": ref_solvers.CCQPSolverMPRGP}


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.uint64)


def same_bits(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def call_or_error(fn):
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                return fn(), None
    except Exception as e:  # noqa: BLE001
        return None, type(e).__name__


def gen_ops():
    cases, arrays = C.op_cases()
    ofac = C.oracle_factory()
    out_arrays = dict(arrays)
    meta = []
    for c in cases:
        x = arrays[c["x"]]
        rop = C.build_op(c["spec"], REF_FACTORY, arrays)
        oop = C.build_op(c["spec"], ofac, arrays)
        assert rop.embedded_dimension == oop.embedded_dimension == C.spec_dim(c["spec"])
        ry, rerr = call_or_error(lambda: rop(x.copy()))
        oy, oerr = call_or_error(lambda: oop(x.copy()))
        assert rerr == oerr, (c["id"], rerr, oerr)
        entry = dict(c)
        entry["err"] = rerr
        if rerr is None:
            assert same_bits(ry, oy), ("oracle != reference (projection)", c["id"])
            out_arrays[c["id"] + ".y"] = np.asarray(ry, dtype=np.float64)
        if c["nv"]:
            rn, rnerr = call_or_error(lambda: rop.normal_vector(x.copy()))
            on, onerr = call_or_error(lambda: oop.normal_vector(x.copy()))
            assert rnerr == onerr, (c["id"], "normal_vector", rnerr, onerr)
            entry["nv_err"] = rnerr
            if rnerr is None:
                assert same_bits(rn, on), ("oracle != reference (normal_vector)", c["id"])
                out_arrays[c["id"] + ".nv"] = np.asarray(rn, dtype=np.float64)
        meta.append(entry)
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out_arrays)
    with open(os.path.join(GOLD, "ops_cases.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("ops: %d cases pinned" % len(meta))


def gen_pg():
    cases, arrays = C.pg_cases()
    ofac = C.oracle_factory()
    out = dict(arrays)
    meta = []
    for c in cases:
        x, g = arrays[c["x"]], arrays[c["g"]]
        rop = C.build_op(c["spec"], REF_FACTORY, arrays)
        oop = C.build_op(c["spec"], ofac, arrays)
        r, rerr = call_or_error(lambda: rop.projected_gradient(x.copy(), g.copy()))
        o, oerr = call_or_error(lambda: oop.projected_gradient(x.copy(), g.copy()))
        assert rerr == oerr, (c["id"], rerr, oerr)
        entry = dict(c)
        entry["err"] = rerr
        entry["none"] = rerr is None and r is None
        if rerr is None and r is not None:
            assert o is not None and same_bits(r[0], o[0]) and same_bits(r[1], o[1]), c["id"]
            out[c["id"] + ".free"] = np.asarray(r[0], dtype=np.float64)
            out[c["id"] + ".chop"] = np.asarray(r[1], dtype=np.float64)
        else:
            assert (r is None) == (o is None)
        meta.append(entry)
    np.savez_compressed(os.path.join(GOLD, "pg.npz"), **out)
    with open(os.path.join(GOLD, "pg_cases.json"), "w") as f:
        json.dump(meta, f, indent=0)
    from collections import Counter
    print("projected_gradient: %d cases pinned" % len(meta), Counter(e["err"] for e in meta))


_PROBLEMS = {}


def get_problem(pid):
    if pid in _PROBLEMS:
        return _PROBLEMS[pid]
    if pid.startswith("suite:"):
        A, b = C.suite_problem(pid[6:])
    elif pid == "readme":
        A = np.array(C.TRIDIAG)
        b = -A.dot(np.array([1, 0, 1]))
    elif pid.startswith("sgen:"):
        _, n, k, seed = pid.split(":")
        A, b, _ = gen_ref.sparse_problem(int(n), int(k), int(seed))
    elif pid.startswith("sstencil:"):
        _, n, seed = pid.split(":")
        A, b, _ = gen_ref.sparse_problem(int(n), 26, int(seed), pattern="stencil27")
    elif pid.startswith("contact:"):
        _, m, n, seed = pid.split(":")
        A, b, _ = gen_ref.contact_problem(int(m), int(n), int(seed))
    elif pid.startswith("sirr:"):
        _, n, seed = pid.split(":")
        A, b, _ = gen_ref.irregular_problem(int(n), int(seed))
    else:
        _, n, seed = pid.split(":")
        A, b, _ = gen_ref.problem(int(n), int(seed))
    _PROBLEMS[pid] = (A, b)
    return A, b


def gen_solvers(sparse=False, contact=False, big=False):
    """sparse=False: the dense cases -> solvers.npz; sparse=True: the reference run with A =
    scipy.sparse.csr_matrix (it only calls `A.dot`) -> sparse.npz, which also holds the CSR arrays of the
    irregular matrices and of one small generated matrix (known-answer vectors for the product generator)."""
    cases = C.big_solver_cases() if big else C.contact_solver_cases() if contact else \
        C.sparse_solver_cases() if sparse else C.solver_cases()
    stem = "big" if big else "contact" if contact else "sparse" if sparse else "solvers"
    sparse = sparse or contact                   # both: A is not a dense array (no entrywise perturbation of A)
    ofac = C.oracle_factory()
    arrays = dict(C.SUITE_ARRAYS)
    out = {}
    meta = []
    t0 = time.time()
    if sparse and not contact:
        K = gen_ref.sparse_spd(64, 4, 5)
        out["kat.sgen:64:4:5.indptr"], out["kat.sgen:64:4:5.indices"], out["kat.sgen:64:4:5.data"] = \
            K.indptr, K.indices, K.data
        K = gen_ref.sparse_spd(1000, 32, 7, 100, 333)
        out["kat.sgen:1000:32:7.rows100_333.indices"], out["kat.sgen:1000:32:7.rows100_333.data"] = K.indices, K.data
    for c in cases:
        A, b = get_problem(c["problem"])
        n = b.shape[0]
        out[c["problem"] + ".b"] = b
        if c["problem"].startswith("contact:"):
            pid = c["problem"]
            out[pid + ".D.indptr"], out[pid + ".D.indices"], out[pid + ".D.data"] = A.D.indptr, A.D.indices, A.D.data
            out[pid + ".D.shape"] = np.array(A.D.shape)
            out[pid + ".minv"], out[pid + ".eps"] = A.minv, np.array(A.eps)
        if c["problem"].startswith("sirr:"):
            out[c["problem"] + ".indptr"], out[c["problem"] + ".indices"], out[c["problem"] + ".data"] = \
                A.indptr, A.indices, A.data
        rop = None if c["op"] is None else C.build_op(c["op"], REF_FACTORY, arrays)
        oop = None if c["op"] is None else C.build_op(c["op"], ofac, arrays)
        max_mv = np.inf if c["max_mv"] is None else c["max_mv"]
        x0 = C.case_x0(c, n)

        def run_ref():
            np.random.seed(c["rng_seed"])
            s = REF_SOLVERS[c["solver"]](c["tol"], max_mv, **c["kwargs"])
            with contextlib.redirect_stdout(io.StringIO()):
                s.solve(A, b, x0=None if x0 is None else x0.copy(), convex_proj_op=rop)
            return s

        def run_oracle():
            np.random.seed(c["rng_seed"])
            return cpu_ref.SOLVERS[c["solver"]](A, b, c["tol"], max_mv, x0=None if x0 is None else x0.copy(),
                                                P=oop, **c["kwargs"])

        r, rerr = call_or_error(run_ref)
        draws_ref = np.random.uniform()
        o, oerr = call_or_error(run_oracle)
        draws_orc = np.random.uniform()
        assert rerr == oerr, (c["id"], rerr, oerr)
        entry = dict(c)
        entry["err"] = rerr
        if rerr is None:
            assert draws_ref == draws_orc, ("RNG stream position differs", c["id"])
            assert same_bits(r.solution, o.solution), ("oracle != reference (x)", c["id"])
            assert r.solution_num_matrix_vector_multiplications == \
                o.solution_num_matrix_vector_multiplications, c["id"]
            assert bool(r.solution_converged) == bool(o.solution_converged), c["id"]
            assert same_bits(r.solution_residual, o.solution_residual), ("oracle != reference (res)", c["id"])
            out[c["id"] + ".x"] = np.asarray(r.solution, dtype=np.float64)
            entry["mv"] = int(r.solution_num_matrix_vector_multiplications)
            entry["converged"] = bool(r.solution_converged)
            entry["res"] = float(r.solution_residual)
            entry["executed_mv"] = int(o.executed_mv)
            if True:
                # per-iteration host scalars (first C.TRACE_ROWS iterations) from the oracle's `trace=` hook, taken from
                # a re-run that must reproduce the pinned result bit for bit
                def run_traced():
                    np.random.seed(c["rng_seed"])
                    return cpu_ref.SOLVERS[c["solver"]](A, b, c["tol"], max_mv, x0=None if x0 is None else x0.copy(),
                                                        P=oop, trace=[], **c["kwargs"])
                t, terr = call_or_error(run_traced)
                assert terr is None and same_bits(t.solution, r.solution) and \
                    t.solution_num_matrix_vector_multiplications == entry["mv"], ("traced re-run differs", c["id"])
                out[c["id"] + ".trace"] = C.trace_array(c["solver"], t.trace)
                entry["trace_len"] = len(t.trace)
            # Control-flow stability probe (SURVEY.md hard part 1): re-run the oracle on inputs
            # perturbed by <= 1 ulp.  Where the REFERENCE algorithm's own MV count moves under such
            # a perturbation, "identical MV count" is not a meaningful bar for any implementation
            # whose matvec rounds differently from OpenBLAS; the band and the spread of x are
            # recorded and the GPU test checks against those instead.
            rng = np.random.default_rng(zlib.crc32(c["id"].encode()))
            mvs, spread = [entry["mv"]], 0.0
            for _ in range(2 if big else 8):          # (BASELINE-size cases: two probes, each is a full-size solve)
                bp = np.asarray(b, dtype=np.float64) * (1 + rng.integers(-1, 2, size=n) * 2.0 ** -52)
                Ap = A
                if n <= 64 and not sparse:
                    Ap = np.asarray(A, dtype=np.float64) * (1 + rng.integers(-1, 2, size=A.shape) * 2.0 ** -52)
                np.random.seed(c["rng_seed"])
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    with np.errstate(all="ignore"):
                        po = cpu_ref.SOLVERS[c["solver"]](Ap, bp, c["tol"], max_mv,
                                                          x0=None if x0 is None else x0.copy(), P=oop, **c["kwargs"])
                mvs.append(int(po.solution_num_matrix_vector_multiplications))
                den = max(np.linalg.norm(o.solution), 1e-300)
                spread = max(spread, float(np.linalg.norm(po.solution - o.solution) / den))
            entry["mv_band"] = [min(mvs), max(mvs)]
            entry["x_spread"] = spread
        meta.append(entry)
        print("  %-40s mv=%-6s band=%-12s spread=%-9.2e res=%-12s err=%s  [%.1fs]" % (
            c["id"], entry.get("mv"), entry.get("mv_band"), entry.get("x_spread", 0.0), entry.get("res"), rerr,
            time.time() - t0), flush=True)
    for k, v in arrays.items():
        out[k] = v
    np.savez_compressed(os.path.join(GOLD, stem + ".npz"), **out)
    with open(os.path.join(GOLD, stem + "_cases.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("%s: %d cases pinned" % (stem, len(meta)))


def gen_generator():
    """Known-answer vectors for the synthetic generator (oracle/gen_ref.py): the device
    generator must reproduce these bits."""
    out = {}
    for n, seed in ((8, 1234), (33, 5), (256, 1234)):
        A = gen_ref.spd_rows(n, seed)
        assert np.array_equal(A, A.T)
        out["A.%d.%d" % (n, seed)] = A
        out["xstar.%d.%d" % (n, seed)] = gen_ref.xstar(n, seed)
    A = gen_ref.spd_rows(1024, 1234)
    w = np.linalg.eigvalsh(A)
    out["eig.1024.1234"] = np.array([w[0], w[-1]])
    out["rows.1024.1234.100_4"] = gen_ref.spd_rows(1024, 1234, 100, 4)
    assert np.array_equal(out["rows.1024.1234.100_4"], A[100:104])
    np.savez_compressed(os.path.join(GOLD, "generator.npz"), **out)
    print("generator: eig(n=1024) in [%.4f, %.4f]" % (w[0], w[-1]))


def main():
    os.makedirs(GOLD, exist_ok=True)
    only = sys.argv[1:]          # e.g. `gen_golden.py sparse` regenerates just that fixture set
    steps = {"generator": gen_generator, "ops": gen_ops, "pg": gen_pg, "solvers": gen_solvers,
             "sparse": lambda: gen_solvers(sparse=True), "contact": lambda: gen_solvers(contact=True),
             "big": lambda: gen_solvers(big=True)}
    for name, fn in steps.items():
        if not only or name in only:
            fn()
    if only:
        return
    import platform
    meta = {"numpy": np.__version__, "python": platform.python_version(), "machine": platform.machine(),
            "note": "expected outputs are the reference's (palmerb4/CCQPpy @ /root/reference) outputs; "
                    "oracle/cpu_ref.py reproduced every one bit for bit when this was generated"}
    try:
        from threadpoolctl import threadpool_info
        meta["blas"] = [{k: d.get(k) for k in ("internal_api", "version", "num_threads", "architecture")}
                        for d in threadpool_info()]
    except Exception:  # noqa: BLE001
        pass
    with open(os.path.join(GOLD, "META.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
