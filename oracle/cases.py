"""Shared parity-case catalogue  --  TEST INFRASTRUCTURE.

One JSON-able description per case, from which the SAME operator / solver call can be
built for three implementations:
  * the reference (only inside `oracle/gen_golden.py`, in the build container),
  * the CPU oracle `oracle/cpu_ref.py`,
  * the product `ccqppy_amd` (HIP).
Inputs are either seeded here or stored beside the expected outputs in tests/golden/.
"""
import numpy as np

# ------------------------------------------------------------------ op specs
# bound := None | {"scalar": v, "int": bool} | {"fill": v, "int": bool} | {"arr": "<npz key>"}


def _bound(b, dim, arrays):
    if b is None:
        return None
    if "arr" in b:
        return arrays[b["arr"]]
    v = int(b["scalar" if "scalar" in b else "fill"]) if b.get("int") else float(
        b["scalar" if "scalar" in b else "fill"])
    if "scalar" in b:
        return v
    return np.full(dim, v, dtype=np.int64 if b.get("int") else np.float64)


def build_op(spec, factory, arrays=None):
    """factory: dict kind -> class taking the reference's positional arguments."""
    arrays = arrays or {}
    k = spec["kind"]
    if k == "Identity":
        return factory[k](spec["dim"])
    if k == "Lower":
        return factory[k](spec["dim"], _bound(spec.get("lb"), spec["dim"], arrays))
    if k == "Upper":
        return factory[k](spec["dim"], _bound(spec.get("ub"), spec["dim"], arrays))
    if k == "Box":
        return factory[k](spec["dim"], _bound(spec.get("lb"), spec["dim"], arrays),
                          _bound(spec.get("ub"), spec["dim"], arrays))
    if k == "Sphere":
        return factory[k](spec["dim"], spec.get("radius"))
    if k == "Cone":
        return factory[k](spec["dim"], spec.get("aspect_ratio"))
    if k == "Disjoint":
        subs = []
        if "repeat" in spec:
            subs += [spec["repeat"]["op"]] * spec["repeat"]["count"]
        subs += spec.get("ops", [])
        return factory[k](*[build_op(s, factory, arrays) for s in subs])
    raise KeyError(k)


def spec_dim(spec):
    if spec["kind"] != "Disjoint":
        return spec["dim"]
    d = 0
    if "repeat" in spec:
        d += spec_dim(spec["repeat"]["op"]) * spec["repeat"]["count"]
    return d + sum(spec_dim(s) for s in spec.get("ops", []))


def spec_uses_norm(spec):
    """True if the op's result depends on a floating-point REDUCTION (Sphere/Cone)."""
    if spec["kind"] in ("Sphere", "Cone"):
        return True
    if spec["kind"] == "Disjoint":
        subs = ([spec["repeat"]["op"]] if "repeat" in spec else []) + spec.get("ops", [])
        return any(spec_uses_norm(s) for s in subs)
    return False


def spec_max_norm_dim(spec):
    """Largest dimension over which a Sphere/Cone norm is taken (0 if none)."""
    if spec["kind"] in ("Sphere", "Cone"):
        return spec["dim"]
    if spec["kind"] == "Disjoint":
        subs = ([spec["repeat"]["op"]] if "repeat" in spec else []) + spec.get("ops", [])
        return max([spec_max_norm_dim(s) for s in subs] + [0])
    return 0


def oracle_factory():
    from oracle import cpu_ref as r
    return {"Identity": r.Identity, "Lower": r.LowerBound, "Upper": r.UpperBound, "Box": r.Box,
            "Sphere": r.Sphere, "Cone": r.Cone, "Disjoint": r.Disjoint}


def product_factory():
    from ccqppy_amd import solution_spaces as ss
    return {"Identity": ss.IdentityProjOp, "Lower": ss.LowerBoundProjOp,
            "Upper": ss.UpperBoundProjOp, "Box": ss.BoxProjOp, "Sphere": ss.SphereProjOp,
            "Cone": ss.ConeProjOp, "Disjoint": ss.DisjointProjOp}


# ------------------------------------------------------------------ op cases
DIMS = [1, 2, 3, 8, 15, 16, 17, 64, 1000]


def op_cases():
    """Returns (cases, arrays): cases = list of dicts {id, spec, x (npz key), nv: bool}."""
    rng = np.random.default_rng(20260101)
    arrays = {}
    cases = []

    def add(cid, spec, x, nv=True):
        arrays[cid + ".x"] = np.asarray(x, dtype=np.float64)
        cases.append({"id": cid, "spec": spec, "x": cid + ".x", "nv": nv})

    for d in DIMS:
        lbv = np.sort(rng.normal(size=(2, d)), axis=0)
        arrays["lb%d" % d] = lbv[0] - 0.05
        arrays["ub%d" % d] = lbv[1] + 0.05
        specs = {
            "identity": {"kind": "Identity", "dim": d},
            "lower_def": {"kind": "Lower", "dim": d},
            "lower_arr": {"kind": "Lower", "dim": d, "lb": {"arr": "lb%d" % d}},
            "upper_def": {"kind": "Upper", "dim": d},
            "upper_arr": {"kind": "Upper", "dim": d, "ub": {"arr": "ub%d" % d}},
            "box_def": {"kind": "Box", "dim": d},
            "box_arr": {"kind": "Box", "dim": d, "lb": {"arr": "lb%d" % d}, "ub": {"arr": "ub%d" % d}},
            "box_int": {"kind": "Box", "dim": d, "lb": {"fill": -2, "int": True}, "ub": {"fill": 1, "int": True}},
            "box_scalar": {"kind": "Box", "dim": d, "lb": {"scalar": -0.5}, "ub": {"scalar": 0.5}},
            "box_half": {"kind": "Box", "dim": d, "lb": {"fill": -0.5}, "ub": {"fill": 0.5}},
            "sphere_def": {"kind": "Sphere", "dim": d},
            "sphere_r07": {"kind": "Sphere", "dim": d, "radius": 0.7},
            "sphere_r25": {"kind": "Sphere", "dim": d, "radius": 2.5},
            "cone_def": {"kind": "Cone", "dim": d},
            "cone_mu05": {"kind": "Cone", "dim": d, "aspect_ratio": 0.5},
            "cone_mu2": {"kind": "Cone", "dim": d, "aspect_ratio": 2},
        }
        for name, spec in specs.items():
            for si, scale in enumerate((0.3, 3.0)):
                x = rng.normal(size=d) * scale
                # scalar bounds / Cone have no usable normal_vector -> error-type parity only
                add("%s.d%d.s%d" % (name, d, si), spec, x)

    # crafted elementwise edge values (-0.0 -> +0.0, +-inf -> NaN, NaN passes)
    edge = np.array([-1.0, 1.0, -0.0, 0.0, np.inf, -np.inf, np.nan, 0.5, -1.0000000001,
                     1.0000000001, 1e-300, -1e-300, 1e300, -1e300, 0.99999999, -0.99999999])
    d = edge.size
    add("edge.box_def", {"kind": "Box", "dim": d}, edge, nv=False)
    add("edge.lower_def", {"kind": "Lower", "dim": d}, edge, nv=False)
    add("edge.upper_def", {"kind": "Upper", "dim": d}, edge, nv=False)
    add("edge.box_negint", {"kind": "Box", "dim": d, "lb": {"fill": -9, "int": True},
                            "ub": {"fill": -3, "int": True}}, edge * 4 - 5, nv=False)
    add("edge.box_negflt", {"kind": "Box", "dim": d, "lb": {"fill": -9.0}, "ub": {"fill": -3.0}},
        np.where(np.isfinite(edge), edge * 4 - 5, edge), nv=False)
    add("edge.box_scalar_int", {"kind": "Box", "dim": d, "lb": {"scalar": -1, "int": True},
                                "ub": {"scalar": 1, "int": True}}, edge, nv=False)
    add("edge.lower_posflt", {"kind": "Lower", "dim": d, "lb": {"fill": 2.0}}, edge, nv=False)
    add("edge.upper_negint", {"kind": "Upper", "dim": d, "ub": {"fill": -2, "int": True}}, edge, nv=False)

    # on-boundary / near-boundary points for normal_vector and np.isclose masks
    for d in (3, 8, 40):
        lb = arrays.setdefault("nlb%d" % d, -np.abs(rng.normal(size=d)) - 0.1)
        ub = arrays.setdefault("nub%d" % d, np.abs(rng.normal(size=d)) + 0.1)
        x = rng.uniform(lb, ub)
        x[::3] = lb[::3]
        x[1::3] = ub[1::3]
        x_near = x.copy()
        x_near[::3] += 3e-9           # inside isclose tolerance of the bound
        x_near[1::3] -= 1.5e-5 * np.abs(ub[1::3])   # outside rtol band
        for tag, xx in (("on", x), ("near", x_near), ("out", x + 1e-3), ("in", 0.5 * (lb + ub))):
            add("nv.box.d%d.%s" % (d, tag), {"kind": "Box", "dim": d, "lb": {"arr": "nlb%d" % d},
                                             "ub": {"arr": "nub%d" % d}}, xx)
            add("nv.lower.d%d.%s" % (d, tag), {"kind": "Lower", "dim": d, "lb": {"arr": "nlb%d" % d}}, xx)
            add("nv.upper.d%d.%s" % (d, tag), {"kind": "Upper", "dim": d, "ub": {"arr": "nub%d" % d}}, xx)
        v = rng.normal(size=d)
        v /= np.linalg.norm(v)
        for tag, xx in (("on", v * 1.3), ("near", v * (1.3 + 2e-9)), ("out", v * 1.31), ("in", v * 0.4),
                        ("zero", np.zeros(d))):
            add("nv.sphere.d%d.%s" % (d, tag), {"kind": "Sphere", "dim": d, "radius": 1.3}, xx)

    # cone branches: inside (mu*z >= r), polar (-z/mu >= r), generic
    for d in (1, 2, 3, 8, 15):
        for mu in (None, 0.5, 3.0):
            base = rng.normal(size=d) * 0.05
            ins = base.copy(); ins[-1] = 5.0
            pol = base.copy(); pol[-1] = -5.0
            gen = rng.normal(size=d)
            spec = {"kind": "Cone", "dim": d, "aspect_ratio": mu}
            tag = "mu%s" % ("def" if mu is None else str(mu))
            add("cone.%s.d%d.inside" % (tag, d), spec, ins)
            add("cone.%s.d%d.polar" % (tag, d), spec, pol)
            add("cone.%s.d%d.generic" % (tag, d), spec, gen)

    # Disjoint: mixed, nested, many small blocks
    mixed = {"kind": "Disjoint", "ops": [
        {"kind": "Box", "dim": 5}, {"kind": "Cone", "dim": 3}, {"kind": "Sphere", "dim": 3},
        {"kind": "Identity", "dim": 2}, {"kind": "Lower", "dim": 4}, {"kind": "Upper", "dim": 4}]}
    nested = {"kind": "Disjoint", "ops": [
        {"kind": "Sphere", "dim": 2, "radius": 0.5}, mixed,
        {"kind": "Disjoint", "ops": [{"kind": "Identity", "dim": 1}, {"kind": "Box", "dim": 3}]},
        {"kind": "Cone", "dim": 7, "aspect_ratio": 0.25}]}
    many = {"kind": "Disjoint", "repeat": {"op": {"kind": "Sphere", "dim": 3}, "count": 341},
            "ops": [{"kind": "Identity", "dim": 1}]}
    big = {"kind": "Disjoint", "ops": [
        {"kind": "Box", "dim": 500, "lb": {"fill": -0.5}, "ub": {"fill": 0.5}},
        {"kind": "Cone", "dim": 500}]}
    nvmix = {"kind": "Disjoint", "ops": [
        {"kind": "Box", "dim": 5}, {"kind": "Sphere", "dim": 3},
        {"kind": "Identity", "dim": 2}, {"kind": "Lower", "dim": 4}, {"kind": "Upper", "dim": 4}]}
    for name, spec in (("mixed", mixed), ("nested", nested), ("many", many), ("big", big), ("nvmix", nvmix)):
        dd = spec_dim(spec)
        for si, scale in enumerate((0.3, 1.0, 3.0)):
            add("disjoint.%s.s%d" % (name, si), spec, rng.normal(size=dd) * scale)
    xb = np.clip(rng.normal(size=spec_dim(nvmix)), -1, 1)
    add("disjoint.nvmix.onbound", nvmix, xb)

    # round 2: unions with MANY blocks beyond the thread-per-block size and blocks spanning several 2048-element
    # reduction chunks (the segmented engine, csrc/ccqp_segs.hip; reference loop solution_spaces.py:540-560, Sphere
    # :417-435, Cone :470-492).  Own generator so that the cases above keep their inputs.
    rng2 = np.random.default_rng(20260202)
    arrays["l2lb"] = -np.abs(rng2.normal(size=2500)) - 0.05
    arrays["l2ub"] = np.abs(rng2.normal(size=2500)) + 0.05
    sph100 = {"kind": "Disjoint", "repeat": {"op": {"kind": "Sphere", "dim": 100}, "count": 40}}
    largemix = {"kind": "Disjoint", "ops": [
        {"kind": "Box", "dim": 5}, {"kind": "Cone", "dim": 3}, {"kind": "Sphere", "dim": 300, "radius": 2.0},
        {"kind": "Identity", "dim": 70}, {"kind": "Box", "dim": 2500, "lb": {"arr": "l2lb"}, "ub": {"arr": "l2ub"}},
        {"kind": "Cone", "dim": 2100, "aspect_ratio": 0.5}, {"kind": "Sphere", "dim": 64}, {"kind": "Upper", "dim": 65},
        {"kind": "Disjoint", "ops": [{"kind": "Sphere", "dim": 65, "radius": 0.25}, {"kind": "Lower", "dim": 130}]},
        {"kind": "Sphere", "dim": 4100, "radius": 30.0}]}
    largenv = {"kind": "Disjoint", "ops": [
        {"kind": "Sphere", "dim": 300, "radius": 1.3}, {"kind": "Box", "dim": 2500, "lb": {"arr": "l2lb"},
                                                         "ub": {"arr": "l2ub"}},
        {"kind": "Lower", "dim": 70}, {"kind": "Sphere", "dim": 100}, {"kind": "Identity", "dim": 66},
        {"kind": "Sphere", "dim": 2200, "radius": 2.0}, {"kind": "Upper", "dim": 3}]}
    for name, spec in (("sph100x40", sph100), ("largemix", largemix)):
        dd = spec_dim(spec)
        for si, scale in enumerate((0.05, 1.0)):
            add("disjoint.%s.s%d" % (name, si), spec, rng2.normal(size=dd) * scale, nv=(name == "sph100x40"))
    # normal_vector of large blocks: on / inside / outside each set
    dd = spec_dim(largenv)
    on = rng2.uniform(-0.04, 0.04, size=dd)
    v = rng2.normal(size=300); on[:300] = 1.3 * v / np.linalg.norm(v)                 # on the first sphere
    xb = rng2.uniform(arrays["l2lb"], arrays["l2ub"])
    xb[::3] = arrays["l2lb"][::3]; xb[1::3] = arrays["l2ub"][1::3]
    on[300:2800] = xb                                                                  # on the box's faces
    on[2800:2870] = -1.0                                                               # on the default lower bound
    v = rng2.normal(size=2200); on[3036:5236] = 2.0 * v / np.linalg.norm(v)            # on the large sphere
    on[5236:] = 1.0
    add("disjoint.largenv.on", largenv, on)
    add("disjoint.largenv.in", largenv, 0.01 * rng2.normal(size=dd))
    out = on.copy(); out[:300] *= 1.5; out[400] += 3.0; out[3036:5236] *= 1.01
    add("disjoint.largenv.out", largenv, out)
    for d, r in ((65, 1.0), (100, 0.5), (2049, 3.0), (5000, 1.0)):
        for si, scale in enumerate((0.01, 1.0)):
            add("large.sphere.d%d.s%d" % (d, si), {"kind": "Sphere", "dim": d, "radius": r}, rng2.normal(size=d) * scale)
            add("large.cone.d%d.s%d" % (d, si), {"kind": "Cone", "dim": d, "aspect_ratio": r},
                rng2.normal(size=d) * scale)
    return cases, arrays


def pg_cases():
    """projected_gradient cases: (cases, arrays); each case has x and g."""
    rng = np.random.default_rng(424242)
    arrays, cases = {}, []

    def add(cid, spec, x, g):
        arrays[cid + ".x"] = np.asarray(x, dtype=np.float64)
        arrays[cid + ".g"] = np.asarray(g, dtype=np.float64)
        cases.append({"id": cid, "spec": spec, "x": cid + ".x", "g": cid + ".g"})

    for d in (1, 3, 8, 40, 200):
        lb = arrays.setdefault("pglb%d" % d, np.where(rng.random(d) < 0.3, 0.0, -np.abs(rng.normal(size=d)) - 0.1))
        ub = arrays.setdefault("pgub%d" % d, np.abs(rng.normal(size=d)) + 1.1)
        x = rng.uniform(lb, ub)
        x[::3] = lb[::3]
        x[1::3] = ub[1::3]
        x[2::7] = 1.0          # hits the `isclose(x, True)` arm of the Box condition where lb == 0
        x[4::9] = 0.0
        for tag, xx in (("on", x), ("near", x + 2e-9), ("out", x + 1e-3 * (rng.random(d) < 0.2)),
                        ("in", 0.5 * (lb + ub))):
            g = rng.normal(size=d)
            add("pg.box.d%d.%s" % (d, tag), {"kind": "Box", "dim": d, "lb": {"arr": "pglb%d" % d},
                                             "ub": {"arr": "pgub%d" % d}}, xx, g)
            add("pg.lower.d%d.%s" % (d, tag), {"kind": "Lower", "dim": d, "lb": {"arr": "pglb%d" % d}}, xx, g)
            add("pg.upper.d%d.%s" % (d, tag), {"kind": "Upper", "dim": d, "ub": {"arr": "pgub%d" % d}}, xx, g)
    d = 6
    x, g = rng.normal(size=d), rng.normal(size=d)
    add("pg.identity", {"kind": "Identity", "dim": d}, x, g)
    add("pg.sphere", {"kind": "Sphere", "dim": d}, x, g)
    add("pg.cone", {"kind": "Cone", "dim": d}, x, g)
    add("pg.box_scalar", {"kind": "Box", "dim": d, "lb": {"scalar": -0.5}, "ub": {"scalar": 0.5}}, x, g)
    add("pg.box_default", {"kind": "Box", "dim": d}, np.clip(x, -1, 1), g)
    add("pg.box_int", {"kind": "Box", "dim": d, "lb": {"fill": 0, "int": True}, "ub": {"fill": 1, "int": True}},
        np.array([0.0, 1.0, 0.5, 1.0 + 3e-9, -1e-9, 2.0]), g)
    dj = {"kind": "Disjoint", "ops": [{"kind": "Box", "dim": 5}, {"kind": "Lower", "dim": 4},
                                      {"kind": "Disjoint", "ops": [{"kind": "Upper", "dim": 3}, {"kind": "Box", "dim": 70}]}]}
    dd = spec_dim(dj)
    for si, scale in enumerate((0.5, 1.0)):
        add("pg.disjoint.s%d" % si, dj, np.clip(rng.normal(size=dd) * scale, -1, 1), rng.normal(size=dd))
    dj_id = {"kind": "Disjoint", "ops": [{"kind": "Box", "dim": 2}, {"kind": "Identity", "dim": 2}]}
    add("pg.disjoint_identity", dj_id, rng.normal(size=4), rng.normal(size=4))
    dj_sp = {"kind": "Disjoint", "ops": [{"kind": "Sphere", "dim": 2}, {"kind": "Box", "dim": 2}]}
    add("pg.disjoint_sphere", dj_sp, rng.normal(size=4), rng.normal(size=4))
    return cases, arrays


# ------------------------------------------------------------------ solver cases
TRIDIAG = [[2, -1, 0], [-1, 2, -1], [0, -1, 2]]

SUITE = {  # /root/reference/src/ccqppy/problem_suite.py:42-177 (data only)
    "UnconstrainedSPD1": dict(xexact=[1, 0, 1], bfrom=[1, 0, 1], op={"kind": "Identity", "dim": 3}),
    "UnconstrainedSPD2": dict(xexact=[1, 0, 1], bfrom=[1, 0, 1], op={"kind": "Disjoint", "ops": [
        {"kind": "Identity", "dim": 1}, {"kind": "Identity", "dim": 1}, {"kind": "Identity", "dim": 1}]}),
    "BoxConstrainedSPD": dict(xexact=[1, 0, 1], bfrom=[1, 0, 1], op={
        "kind": "Box", "dim": 3, "lb": {"arr": "suite.box.lb"}, "ub": {"arr": "suite.box.ub"}}),
    "ThinBoxConstrainedSPD": dict(xexact=[1, 0, 1], bfrom=[1, 0, 1], op={
        "kind": "Box", "dim": 3, "lb": {"arr": "suite.thin.lb"}, "ub": {"arr": "suite.thin.ub"}}),
    "ActiveBoxConstrainedSPD": dict(xexact=[9, 9, 9], bfrom=[1, 1, 1], op={
        "kind": "Box", "dim": 3, "lb": {"arr": "suite.active.lb"}, "ub": {"arr": "suite.active.ub"}}),
}

SUITE_ARRAYS = {
    "suite.box.lb": np.array([0, 0, 0]), "suite.box.ub": np.array([2, 2, 2]),
    "suite.thin.lb": np.array([-10, -0.1, 0.9]), "suite.thin.ub": np.array([10, 0.1, 1.1]),
    "suite.active.lb": np.array([9, 9, 9]), "suite.active.ub": np.array([10, 10, 10]),
    "readme.lb": np.array([-2, -2, -4]), "readme.ub": np.array([2, 2, 5]),
}

SOLVER_NAMES = ["PGD", "APGD", "APGDAntiRelaxation", "BBPGD", "BBPGDf", "SPG", "MPRGP"]


def suite_problem(name):
    """(A int64, b int64) exactly as problem_suite.py builds them."""
    A = np.array(TRIDIAG)
    return A, -A.dot(np.array(SUITE[name]["bfrom"]))


def _box_half(n):
    return {"kind": "Box", "dim": n, "lb": {"fill": -0.5}, "ub": {"fill": 0.5}}


def solver_cases():
    """List of dicts: id, problem ('suite:<name>' | 'readme' | 'gen:<n>:<seed>'), solver, tol,
    max_mv (None = inf), kwargs, op spec, rng_seed."""
    cases = []
    for pname in SUITE:
        for s in SOLVER_NAMES:
            cases.append(dict(id="suite.%s.%s" % (pname, s), problem="suite:" + pname, solver=s,
                              tol=1e-8, max_mv=10000, kwargs={"step_size": 0.1} if s == "PGD" else {},
                              op=SUITE[pname]["op"], rng_seed=1234))
    cases.append(dict(id="readme.SPG", problem="readme", solver="SPG", tol=1e-10, max_mv=5000, kwargs={},
                      op={"kind": "Box", "dim": 3, "lb": {"arr": "readme.lb"}, "ub": {"arr": "readme.ub"}},
                      rng_seed=0))
    for n in (64, 256, 1024, 4096):
        g = "gen:%d:1234" % n
        half = n // 2
        nsph = n // 3
        rest = n - 3 * nsph
        many = {"kind": "Disjoint", "repeat": {"op": {"kind": "Sphere", "dim": 3}, "count": nsph}}
        if rest:
            many["ops"] = [{"kind": "Identity", "dim": rest}]
        table = [
            ("APGD.box", "APGD", 1e-8, 5000, _box_half(n)),
            ("BBPGD.box", "BBPGD", 1e-8, 5000, _box_half(n)),
            ("SPG.box", "SPG", 1e-8, 5000, _box_half(n)),
            ("MPRGP.box", "MPRGP", 1e-6, 300, _box_half(n)),
            ("BBPGD.cone", "BBPGD", 1e-8, 64, {"kind": "Cone", "dim": n}),
            ("SPG.boxcone", "SPG", 1e-8, 64, {"kind": "Disjoint", "ops": [_box_half(half),
                                                                         {"kind": "Cone", "dim": n - half}]}),
            ("BBPGD.spheres", "BBPGD", 1e-8, 5000, many),
        ]
        if n <= 1024:
            table += [
                ("PGD.box", "PGD", 1e-8, 5000, _box_half(n)),
                ("APGDAntiRelaxation.box", "APGDAntiRelaxation", 1e-8, 5000, _box_half(n)),
                ("BBPGDf.box", "BBPGDf", 1e-8, 5000, _box_half(n)),
                ("APGD.identity", "APGD", 1e-8, 5000, None),
                ("BBPGD.sphere", "BBPGD", 1e-8, 5000, {"kind": "Sphere", "dim": n, "radius": 4.0}),
                ("MPRGP.lower", "MPRGP", 1e-6, 300, {"kind": "Lower", "dim": n, "lb": {"fill": -0.25}}),
            ]
        for tag, s, tol, mx, op in table:
            cases.append(dict(id="gen%d.%s" % (n, tag), problem=g, solver=s, tol=tol, max_mv=mx,
                              kwargs={"step_size": 0.3} if s == "PGD" else {}, op=op, rng_seed=1234))
    # exit paths: MV-cap behaviour, stale iterate, overshoot, UnboundLocalError
    for s in SOLVER_NAMES:
        for mx in (1, 2, 3, 4, 11, 12):
            cases.append(dict(id="exit.%s.max%d" % (s, mx), problem="gen:64:1234", solver=s, tol=1e-12,
                              max_mv=mx, kwargs={"step_size": 0.3} if s == "PGD" else {},
                              op=_box_half(64), rng_seed=1234))
    # solver x operator matrix at n = 256 (every pair the fixtures above do not already hold)
    n = 256
    mixed = {"kind": "Disjoint", "ops": [
        {"kind": "Box", "dim": 100, "lb": {"fill": -0.5}, "ub": {"fill": 0.5}},
        {"kind": "Sphere", "dim": 56, "radius": 1.5},
        {"kind": "Disjoint", "repeat": {"op": {"kind": "Sphere", "dim": 3}, "count": 20}},
        {"kind": "Lower", "dim": 20, "lb": {"fill": -0.25}}, {"kind": "Identity", "dim": 20}]}
    opsx = {"lower": {"kind": "Lower", "dim": n, "lb": {"fill": -0.25}},
            "upper": {"kind": "Upper", "dim": n, "ub": {"fill": 0.25}},
            "sphere": {"kind": "Sphere", "dim": n, "radius": 4.0},
            "cone": {"kind": "Cone", "dim": n},
            "mixed": mixed}
    for s in SOLVER_NAMES:
        for oname, op in opsx.items():
            mx = 40 if oname == "cone" else 3000
            cases.append(dict(id="matrix256.%s.%s" % (s, oname), problem="gen:256:1234", solver=s,
                              tol=1e-6 if s == "MPRGP" else 1e-8, max_mv=mx,
                              kwargs={"step_size": 0.3} if s == "PGD" else {}, op=op, rng_seed=1234))
    # non-default SPG parameters (m, tau, sigma1, sigma2), PGD's default step, NaN path of APGD (x0 == ones
    # makes the Lipschitz estimate 0/0, solvers.py:263-268)
    cases.append(dict(id="params.SPG.a", problem="gen:256:1234", solver="SPG", tol=1e-8, max_mv=5000,
                      kwargs={"m": 3, "tau": 0.3, "sigma1": 0.05, "sigma2": 0.8}, op=_box_half(256), rng_seed=7))
    cases.append(dict(id="params.SPG.b", problem="gen:256:1234", solver="SPG", tol=1e-8, max_mv=5000,
                      kwargs={"m": 1, "tau": 0.9, "sigma1": 0.2, "sigma2": 0.25}, op=_box_half(256), rng_seed=8))
    cases.append(dict(id="params.PGD.default_step", problem="gen:64:1234", solver="PGD", tol=1e-6, max_mv=5000,
                      kwargs={}, op=_box_half(64), rng_seed=1234))
    for s in ("APGD", "APGDAntiRelaxation"):
        cases.append(dict(id="nan.%s.x0_ones" % s, problem="gen:64:1234", solver=s, tol=1e-8, max_mv=14, kwargs={},
                          op=_box_half(64), rng_seed=1234, x0="ones"))
    # SPG-QP beyond the Box: every kind of set, non-default parameters, budgets that end mid-run, a warm start, other
    # sizes and generator seeds -- pins the lock-step SPG driver (ccqp_bspg.hip) directly against the reference
    for n, seed in ((96, 21), (257, 22), (1024, 23)):
        g = "gen:%d:%d" % (n, seed)
        nsph = n // 3
        many = {"kind": "Disjoint", "repeat": {"op": {"kind": "Sphere", "dim": 3, "radius": 0.3}, "count": nsph}}
        if n - 3 * nsph:
            many["ops"] = [{"kind": "Identity", "dim": n - 3 * nsph}]
        q = n // 4
        mix = {"kind": "Disjoint", "ops": [
            {"kind": "Box", "dim": q, "lb": {"fill": -0.3}, "ub": {"fill": 0.4}},
            {"kind": "Sphere", "dim": q, "radius": 0.7},
            {"kind": "Upper", "dim": q, "ub": {"fill": 0.2}},
            {"kind": "Sphere", "dim": n - 3 * q, "radius": 2.0}]}
        tab = [("lower", {"kind": "Lower", "dim": n, "lb": {"fill": -0.2}}, {}, 5000, None, 31),
               ("identity", None, {"m": 2}, 5000, None, 32),
               ("sphere", {"kind": "Sphere", "dim": n, "radius": 2.0}, {"tau": 0.7}, 5000, None, 33),
               ("spheres3", many, {"m": 8, "sigma2": 0.9}, 5000, None, 34),
               ("mix", mix, {"sigma1": 0.1}, 5000, "ramp", 35),
               ("mix.max23", mix, {}, 23, None, 36),
               ("sphere.max7", {"kind": "Sphere", "dim": n, "radius": 2.0}, {}, 7, None, 37)]
        for tag, op, kw, mx, x0, rs in tab:
            c = dict(id="spgx%d.%s" % (n, tag), problem=g, solver="SPG", tol=1e-8, max_mv=mx, kwargs=kw, op=op,
                     rng_seed=rs)
            if x0:
                c["x0"] = x0
            cases.append(c)
    # warm start + default (infinite) MV budget
    cases.append(dict(id="warm.BBPGD", problem="gen:256:7", solver="BBPGD", tol=1e-9, max_mv=None, kwargs={},
                      op=_box_half(256), rng_seed=1234, x0="ramp"))
    cases.append(dict(id="warm.APGD", problem="gen:256:7", solver="APGD", tol=1e-8, max_mv=None, kwargs={},
                      op=_box_half(256), rng_seed=1234, x0="ramp"))
    cases.append(dict(id="warm.MPRGP.identity", problem="gen:64:7", solver="MPRGP", tol=1e-8, max_mv=500,
                      kwargs={}, op=None, rng_seed=1234, x0="ramp"))
    return cases


# ------------------------------------------------------------------ per-iteration host-scalar traces
# One row per completed iteration ("the trace is what localises a control-flow divergence"): the
# scalars that steer the reference's branches -- the sufficient-decrease test solvers.py:302, the exit tests
# :324,649,949,1186, the BB steps :655-656, MPRGP's three-way branch :1098,1121 and its halving loop :1113-1118.
TRACE_ROWS = 32
TRACE_COLUMNS = {
    "PGD": ("mv", "res"),
    "APGD": ("mv", "res", "L", "nback"),
    "APGDAntiRelaxation": ("mv", "res", "L", "nback"),
    "BBPGD": ("mv", "res", "alpha"),
    "BBPGDf": ("mv", "res", "alpha"),
    "SPG": ("mv", "dd", "alpha", "betak"),
    "MPRGP": ("mv", "res", "branch", "nhalve", "alpha_bb"),
}
TRACE_EXACT = ("mv", "nback", "branch", "nhalve")          # integer-valued columns: compared exactly
MPRGP_BRANCH = {"cg": 0.0, "expand": 1.0, "prop": 2.0}


def trace_array(solver, trace):
    """First TRACE_ROWS entries of a trace (list of dicts, oracle/cpu_ref.py `trace=` or the product's
    `solver.trace`) as a float array with TRACE_COLUMNS[solver] columns."""
    cols = TRACE_COLUMNS[solver]
    rows = []
    for e in trace[:TRACE_ROWS]:
        rows.append([MPRGP_BRANCH[e[c]] if c == "branch" else float(e[c]) for c in cols])
    return np.array(rows, dtype=np.float64).reshape(len(rows), len(cols))


def case_x0(case, n):
    if case.get("x0") == "ramp":
        return np.linspace(-0.75, 0.75, n)
    if case.get("x0") == "ones":
        return np.ones(n)
    return None


# ------------------------------------------------------------------ sparse-operator solver cases
def sparse_solver_cases():
    """Same record layout as solver_cases(); problems are 'sgen:<n>:<k>:<seed>' (circulant-pattern sparse
    SPD matrix, oracle/gen_ref.py:sparse_spd) or 'sirr:<n>:<seed>' (irregular, gen_ref.irregular_spd).
    The reference solves them with A = scipy.sparse.csr_matrix (it only calls `A.dot`)."""
    cases = []

    def add(cid, problem, solver, tol, mx, op, **extra):
        cases.append(dict(id=cid, problem=problem, solver=solver, tol=tol, max_mv=mx,
                          kwargs={"step_size": 0.3} if solver == "PGD" else {}, op=op, rng_seed=1234, **extra))

    for s in SOLVER_NAMES:
        add("sgen256k8.%s.box" % s, "sgen:256:8:1234", s, 1e-6 if s == "MPRGP" else 1e-8, 5000, _box_half(256))
    for n, k in ((1024, 8), (4096, 32), (4096, 100)):
        p = "sgen:%d:%d:1234" % (n, k)
        half = n // 2
        tag = "sgen%dk%d" % (n, k)
        add(tag + ".APGD.box", p, "APGD", 1e-8, 5000, _box_half(n))
        add(tag + ".BBPGD.box", p, "BBPGD", 1e-8, 5000, _box_half(n))
        add(tag + ".SPG.box", p, "SPG", 1e-8, 5000, _box_half(n))
        add(tag + ".MPRGP.box", p, "MPRGP", 1e-6, 300, _box_half(n))
        add(tag + ".BBPGD.cone", p, "BBPGD", 1e-8, 64, {"kind": "Cone", "dim": n})
        add(tag + ".SPG.boxcone", p, "SPG", 1e-8, 64,
            {"kind": "Disjoint", "ops": [_box_half(half), {"kind": "Cone", "dim": n - half}]})
    for n in (300, 2000):
        p = "sirr:%d:5" % n
        for s in ("APGD", "BBPGD", "SPG", "MPRGP"):
            add("sirr%d.%s.box" % (n, s), p, s, 1e-6 if s == "MPRGP" else 1e-8, 3000, _box_half(n))
        add("sirr%d.BBPGD.identity" % n, p, "BBPGD", 1e-8, 3000, None)
        add("sirr%d.BBPGDf.lower" % n, p, "BBPGDf", 1e-8, 3000, {"kind": "Lower", "dim": n, "lb": {"fill": -0.25}})
    add("sgen256k8.warm.BBPGD", "sgen:256:8:7", "BBPGD", 1e-9, None, _box_half(256), x0="ramp")
    for n in (4096, 32768):                      # 27-point stencil (side 16 / 32): local-x windows on the GPU
        p = "sstencil:%d:1234" % n
        for s in ("APGD", "BBPGD", "SPG", "MPRGP"):
            add("sstencil%d.%s.box" % (n, s), p, s, 1e-6 if s == "MPRGP" else 1e-8, 3000, _box_half(n))
    return cases


# ------------------------------------------------------------------ BASELINE-size cases, pinned by the reference itself
def big_solver_cases():
    """The BASELINE configs that fit the build container (64 GiB, 8 cores) at their OWN sizes, run by the REFERENCE:
    config 2 (n = 32768 APGD + Box), config 3's solver / set pair at n = 32768 (BBPGD + Cone, 64-matvec budget), and
    two problems of config 5 (n = 8192 MPRGP + Box, tol 1e-6 / 300 matvecs, seeds 1234 + p).  Same record layout as
    solver_cases(); fixtures store b, x, the scalars and the per-iteration trace (no matrices: both sides regenerate
    them from the seed)."""
    n2 = 32768
    cases = [
        dict(id="config2.APGD.box", problem="gen:%d:1234" % n2, solver="APGD", tol=1e-8, max_mv=5000, kwargs={},
             op=_box_half(n2), rng_seed=1234),
        dict(id="config3pair.BBPGD.cone", problem="gen:%d:1234" % n2, solver="BBPGD", tol=1e-8, max_mv=64, kwargs={},
             op={"kind": "Cone", "dim": n2}, rng_seed=1234),
        dict(id="config2size.SPG.boxcone", problem="gen:%d:1234" % n2, solver="SPG", tol=1e-8, max_mv=64, kwargs={},
             op={"kind": "Disjoint", "ops": [_box_half(n2 // 2), {"kind": "Cone", "dim": n2 - n2 // 2}]},
             rng_seed=1234),
    ]
    for p in (0, 1):
        cases.append(dict(id="config5.MPRGP.box.p%d" % p, problem="gen:8192:%d" % (1234 + p), solver="MPRGP", tol=1e-6,
                          max_mv=300, kwargs={}, op=_box_half(8192), rng_seed=1234))
    return cases


# ------------------------------------------------------------------ matrix-free (composed) operator cases
def contact_solver_cases():
    """Problems 'contact:<m>:<n>:<seed>': A = D diag(minv) D^T + eps I handed to the reference as an object
    with `.dot` only (oracle/gen_ref.py:ContactHessian).  m constraints, n body coordinates."""
    cases = []

    def add(cid, problem, solver, tol, mx, op):
        cases.append(dict(id=cid, problem=problem, solver=solver, tol=tol, max_mv=mx,
                          kwargs={"step_size": 0.05} if solver == "PGD" else {}, op=op, rng_seed=1234))

    def nonneg(m):
        return {"kind": "Lower", "dim": m, "lb": {"fill": 0.0}}

    for m, n, seed in ((300, 200, 3), (2000, 3000, 5), (4096, 1024, 7)):
        p = "contact:%d:%d:%d" % (m, n, seed)
        tag = "contact%dx%d" % (m, n)
        add(tag + ".BBPGD.nonneg", p, "BBPGD", 1e-8, 3000, nonneg(m))
        add(tag + ".APGD.nonneg", p, "APGD", 1e-8, 3000, nonneg(m))
        add(tag + ".SPG.box", p, "SPG", 1e-8, 3000, _box_half(m))
        add(tag + ".MPRGP.box", p, "MPRGP", 1e-6, 1500, _box_half(m))
        add(tag + ".BBPGDf.identity", p, "BBPGDf", 1e-8, 3000, None)
    for s in ("PGD", "APGDAntiRelaxation"):
        add("contact300x200.%s.nonneg" % s, "contact:300:200:3", s, 1e-8, 3000, nonneg(300))
    m = 2000
    add("contact2000x3000.BBPGD.friction", "contact:2000:3000:5", "BBPGD", 1e-8, 3000,
        {"kind": "Disjoint", "ops": [{"kind": "Disjoint", "repeat": {"op": {"kind": "Sphere", "dim": 2, "radius": 0.3},
                                                                    "count": 500}},
                                     nonneg(m - 1000)]})
    return cases
