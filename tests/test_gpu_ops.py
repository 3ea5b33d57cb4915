"""GPU parity of the kernels behind the C ABI (called through ccqppy_amd, i.e. through ctypes ->
libccqp_hip.so) against the reference's golden vectors and the CPU oracle.

Bar: bit-exact for everything elementwise / index-like (projections, masks, normal vectors,
generator, axpy family) and for Sphere/Cone blocks of dimension <= 15; reductions over longer
vectors (dot, norm, matvec) cannot reproduce OpenBLAS' summation order and are checked against
fp64 error bounds stated in each test."""
import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import gen_ref
from tests import helpers as H

pytestmark = pytest.mark.gpu

OPS, OP_ARR = H.load_ops()


@pytest.fixture(scope="module")
def eng():
    from ccqppy_amd._engine import get_engine
    e = get_engine(0)
    assert e.arch.startswith("gfx950"), e.arch
    return e


def _call(fn):
    try:
        return fn(), None
    except Exception as e:  # noqa: BLE001
        return None, type(e).__name__


@pytest.mark.parametrize("case", OPS, ids=[c["id"] for c in OPS])
def test_projection_golden(case, eng):
    op = C.build_op(case["spec"], C.product_factory(), OP_ARR)
    x = OP_ARR[case["x"]]
    y, err = _call(lambda: op(x.copy()))
    assert err == case["err"]
    maxd = C.spec_max_norm_dim(case["spec"])
    want = OP_ARR[case["id"] + ".y"]
    if maxd <= 15:
        assert H.same_bits(y, want), (y, want)
    else:
        # the norm's summation order differs from OpenBLAS: r agrees to ~sqrt(dim) ulp, and every
        # output element inherits that relative perturbation
        assert H.ulp_diff(y, want) <= 8 + 2 * np.sqrt(maxd)
    if case["nv"]:
        nv, nerr = _call(lambda: op.normal_vector(x.copy()))
        assert nerr == case["nv_err"]
        if nerr is None:
            wantn = OP_ARR[case["id"] + ".nv"]
            if maxd <= 15:
                assert H.same_bits(nv, wantn)
            else:
                assert H.ulp_diff(nv, wantn) <= 8 + 2 * np.sqrt(maxd)


def test_projection_aliasing_and_tensor_io(eng):
    from ccqppy_amd import solution_spaces as ss
    x = np.array([0.1, -0.2, 0.3])
    assert ss.SphereProjOp(3)(x) is x                    # inside: same object (solution_spaces.py:435)
    xin = np.array([0.0, 0.0, 5.0])
    assert ss.ConeProjOp(3)(xin) is xin                  # mu*z >= |x| (whole-vector norm!): same object (:487)
    big = np.zeros(200)
    big[-1] = 2.0
    assert ss.ConeProjOp(200)(big) is big                # same for a block-reduced norm
    assert ss.SphereProjOp(200)(big * 0.25) is not None and ss.SphereProjOp(200, 3.0)(big) is big
    xt = torch.tensor([2.0, -3.0, 0.5], dtype=torch.float64, device="cuda")
    yt = ss.BoxProjOp(3)(xt)
    assert isinstance(yt, torch.Tensor) and yt.is_cuda
    assert yt.cpu().tolist() == [1.0, -1.0, 0.5]
    with pytest.raises(ValueError):
        ss.BoxProjOp(3)(np.zeros(4))


def test_device_sqrt_and_div_are_correctly_rounded(eng):
    """Sphere/Cone take sqrt and divide on the device: both must be IEEE correctly rounded for the
    bit-exact claims above to hold for every input, not just the fixtures."""
    from ccqppy_amd import solution_spaces as ss
    rng = np.random.default_rng(5)
    bad = 0
    for _ in range(400):
        d = int(rng.integers(1, 16))
        x = rng.normal(size=d) * 10.0 ** rng.uniform(-3, 3)
        r = float(10.0 ** rng.uniform(-3, 3))
        want = C.build_op({"kind": "Sphere", "dim": d, "radius": r}, C.oracle_factory())(x)
        got = ss.SphereProjOp(d, r)(x)
        bad += not H.same_bits(got, want)
        want = C.build_op({"kind": "Cone", "dim": d, "aspect_ratio": r}, C.oracle_factory())(x)
        got = ss.ConeProjOp(d, r)(x)
        bad += not H.same_bits(got, want)
    assert bad == 0


def test_generator_bits(eng):
    from ccqppy_amd import problems
    g = dict(np.load(H.GOLD + "/generator.npz"))
    for n, seed in ((8, 1234), (33, 5), (256, 1234)):
        A = problems.spd_rows(n, seed).cpu().numpy()
        assert H.same_bits(A, g["A.%d.%d" % (n, seed)])
        assert H.same_bits(problems.xstar(n, seed).cpu().numpy(), g["xstar.%d.%d" % (n, seed)])
    assert H.same_bits(problems.spd_rows(1024, 1234, 100, 4).cpu().numpy(), g["rows.1024.1234.100_4"])
    # a size the fixtures do not hold: compare with the CPU statement directly, odd n + row offset
    assert H.same_bits(problems.spd_rows(1537, 99, 700, 129).cpu().numpy(), gen_ref.spd_rows(1537, 99, 700, 129))
    A = problems.spd_rows(2048, 7).cpu().numpy()
    assert np.array_equal(A, A.T) and np.all(np.diag(A) == 2.0)


def test_axpy_family_bit_exact(eng):
    rng = np.random.default_rng(1)
    n = 10007
    x, y, z = (rng.normal(size=n) for _ in range(3))
    x[:5] = [0.0, -0.0, np.inf, -np.inf, np.nan]
    dx, dy, dz = (eng.to_device(v) for v in (x, y, z))
    out = eng.empty(n)
    with np.errstate(all="ignore"):
        for a, b in ((1.0, 1.0), (1.0, -1.0), (1.0, -0.37), (1.25, -0.25), (0.3, 0.7)):
            eng.axpby(a, dx, b, dy, out)
            assert H.same_bits(out.cpu().numpy(), a * x + b * y)
        eng.axpby(1.0, dx, -0.37, dy, out)
        assert H.same_bits(out.cpu().numpy(), x - 0.37 * y)          # x - alpha*g
        eng.scale(3.3e5, dx, out)
        assert H.same_bits(out.cpu().numpy(), 3.3e5 * x)
        eng.mul(dx, dy, out)
        assert H.same_bits(out.cpu().numpy(), x * y)
        eng.muldiv(dx, dy, 0.77, out)
        assert H.same_bits(out.cpu().numpy(), x * y / 0.77)
        eng.submul(dx, dy, dz, out)
        assert H.same_bits(out.cpu().numpy(), x - y * z)
        o2 = eng.to_device(x.copy())
        eng.axpby(1.0, o2, 0.5, dy, o2)                               # in place: xk += beta*dk
        xx = x.copy()
        xx += 0.5 * y
        assert H.same_bits(o2.cpu().numpy(), xx)


def test_isclose_masks_bit_exact(eng):
    rng = np.random.default_rng(2)
    n = 5000
    b = rng.normal(size=n)
    a = b.copy()
    a[::4] += 1e-8 * rng.normal(size=a[::4].size)
    a[1::4] *= 1 + 1e-5 * rng.normal(size=a[1::4].size)
    a[2::4] += rng.normal(size=a[2::4].size)
    a[:6] = [np.inf, -np.inf, np.nan, 1.0, np.inf, 0.0]
    b[:6] = [np.inf, np.inf, np.nan, np.inf, 1.0, -0.0]
    g = rng.normal(size=n)
    da, db, dg = eng.to_device(a), eng.to_device(b), eng.to_device(g)
    out = eng.empty(n)
    with np.errstate(all="ignore"):
        m = np.isclose(a, b)
        eng.isclose_mask_mul(da, db, None, out)
        assert np.array_equal(out.cpu().numpy(), m.astype(np.float64))
        eng.isclose_mask_mul(da, db, dg, out)
        assert H.same_bits(out.cpu().numpy(), m * g)
        eng.isclose_mask_mul(da, db, dg, out, invert=True)
        assert H.same_bits(out.cpu().numpy(), (1 - m) * g)
    eng.count_notclose(da, db, 3)
    assert eng.read(3, 1)[0] == float(np.count_nonzero(~m))


@pytest.mark.parametrize("n", [0, 1, 3, 64, 255, 256, 8192, 8193, 100003, 1 << 20])
def test_dots_accuracy_and_determinism(n, eng):
    rng = np.random.default_rng(n + 1)
    x, y = rng.normal(size=n), rng.normal(size=n)
    dx, dy = eng.to_device(x), eng.to_device(y)
    got = eng.dots_now([(dx, dy), (dx, dx)], 4)
    import math
    exact = [math.fsum(x * y), math.fsum(x * x)]
    bound = [4 * np.finfo(float).eps * max(1.0, np.sqrt(n)) * float(np.sum(np.abs(x * y))) + 1e-300,
             4 * np.finfo(float).eps * max(1.0, np.sqrt(n)) * float(np.sum(x * x)) + 1e-300]
    for gval, e, bnd in zip(got, exact, bound):
        assert abs(float(gval) - e) <= bnd
    again = eng.dots_now([(dx, dy), (dx, dx)], 10)
    assert [float(v) for v in again] == [float(v) for v in got]        # run-to-run bitwise


def _gemv_ref_bound(A, x):
    n = A.shape[1]
    return 4 * np.finfo(float).eps * np.sqrt(max(n, 1)) * (np.abs(A) @ np.abs(x)) + 1e-300


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 3), (5, 7), (64, 64), (130, 257), (257, 130), (1000, 1000),
                                       (512, 4096), (4096, 513)])
def test_gemv_matches_fp64_reference(rows, cols, eng):
    rng = np.random.default_rng(rows * 31 + cols)
    A = rng.normal(size=(rows, cols))
    x = rng.normal(size=cols)
    b = rng.normal(size=rows)
    dA, dx, db = eng.to_device(A), eng.to_device(x), eng.to_device(b)
    y, g = eng.empty(rows), eng.empty(rows)
    eng.gemv(dA, dx, y, db, g)
    import math
    exact = np.array([math.fsum(A[i] * x) for i in range(rows)])
    yy = y.cpu().numpy()
    assert np.all(np.abs(yy - exact) <= _gemv_ref_bound(A, x))          # 4*eps*sqrt(n)*sum|a x|
    assert H.same_bits(g.cpu().numpy(), yy + b)                         # epilogue = one numpy add
    y2 = eng.empty(rows)
    eng.gemv(dA, dx, y2)
    assert H.same_bits(y2.cpu().numpy(), yy)


def test_gemv_bits_do_not_depend_on_tiling_alignment_or_sharding(eng):
    """The determinism contract of include/ccqp_hip.h: y[i] depends on (row i, cols) only."""
    rng = np.random.default_rng(11)
    rows, cols = 777, 2050
    A = rng.normal(size=(rows, cols))
    x = rng.normal(size=cols)
    dA, dx = eng.to_device(A), eng.to_device(x)
    y0 = eng.empty(rows)
    eng.gemv(dA, dx, y0)
    base = y0.cpu().numpy()
    for R in (1, 2, 4, 8):
        for U in (1, 2, 4):
            for nt in (0, 1):
                for bpc in (0, 1, 3):
                    y = eng.empty(rows)
                    eng.gemv(dA, dx, y, variant=R | (U << 4) | (nt << 8) | (bpc << 12))
                    assert H.same_bits(y.cpu().numpy(), base), (R, U, nt, bpc)
    # launch rounds (ccqp_gemv.hip: what a >= 32 GiB matvec does by default), forced on a small matrix: bpc = 1 and
    # R = 1 make a round 1024 rows on a 256-CU device, so a 5000-row matrix takes five launches; with and without
    # the fused epilogue; 2 tasks per wave and round; explicitly off
    rows2 = 5000
    A2 = rng.normal(size=(rows2, 300))
    b2 = rng.normal(size=rows2)
    dA2, dx2, db2 = eng.to_device(A2), eng.to_device(x[:300]), eng.to_device(b2)
    y1, g1 = eng.empty(rows2), eng.empty(rows2)
    eng.gemv(dA2, dx2, y1, db2, g1, variant=0x800)
    assert eng.gemv_plan(rows2, 300, 0x800) == (rows2, 1)
    for v in (0x1411, 0x2412, 0x201411, 0x1418):
        per, launches = eng.gemv_plan(rows2, 300, v)
        assert launches == -(-rows2 // per) and (launches > 1 or per >= rows2)
        y2, g2 = eng.empty(rows2), eng.empty(rows2)
        eng.gemv(dA2, dx2, y2, db2, g2, variant=v)
        assert H.same_bits(y2.cpu().numpy(), y1.cpu().numpy()) and H.same_bits(g2.cpu().numpy(), g1.cpu().numpy()), hex(v)
        y3 = eng.empty(rows2)
        eng.gemv(dA2, dx2, y3, variant=v)
        assert H.same_bits(y3.cpu().numpy(), y1.cpu().numpy()), hex(v)
    assert eng.gemv_plan(rows2, 300, 0x1411)[1] >= 2
    # the default: small matrices in one launch of the VALU form; from 16 GiB on ONE persistent launch of the LDS-staged
    # form (where it applies: columns a multiple of 128); the VALU form (forced: 0xF) cuts the headline matrix into rounds
    assert eng.gemv_plan(4096, 4096)[1] == 1 and eng.gemv_form(4096, 4096) == "gemv_stream_kernel"
    assert eng.gemv_plan(131072, 131072) == (131072, 1) and eng.gemv_form(131072, 131072) == "gemv_lds_kernel"
    assert eng.gemv_form(16384, 131072) == "gemv_lds_kernel" and eng.gemv_form(32768, 32768) == "gemv_lds_kernel"
    assert eng.gemv_form(24576, 24576) == "gemv_stream_kernel" and eng.gemv_form(32768, 32768, 0xF) == "gemv_stream_kernel"
    assert eng.gemv_form(131072, 131000) == "gemv_lds_kernel" and eng.gemv_form(131072, 131001) == "gemv_stream_kernel"
    per, launches = eng.gemv_plan(131072, 131072, 0xF)
    assert launches > 1 and per * launches >= 131072 and eng.gemv_form(131072, 131072, 0xF) == "gemv_stream_kernel"
    # the LDS-staged form (variant bit 24: rows through an LDS ring by LDS-DMA loads): the SAME bits as the VALU form, on
    # shapes it covers (an even number of columns >= 128: 1, 2, 3 and many 128-column steps, with and without a column
    # tail; an odd row count; epilogue; row blocks; more tasks per wave than one flush group), and a silent fall-through
    # to the VALU form on the others (odd or fewer than 128 columns)
    for rows3, cols3 in ((5, 128), (333, 256), (64, 384), (1001, 4096), (777, 2050), (300001, 128), (50, 300), (9, 130),
                         (33, 254), (7, 129), (11, 126)):
        A3 = rng.normal(size=(rows3, cols3))
        x3, b3 = rng.normal(size=cols3), rng.normal(size=rows3)
        dA3, dx3, db3 = eng.to_device(A3), eng.to_device(x3), eng.to_device(b3)
        ya, ga, yb, gb = eng.empty(rows3), eng.empty(rows3), eng.empty(rows3), eng.empty(rows3)
        eng.gemv(dA3, dx3, ya, db3, ga)
        # (bits 25-27: ring slots 4 (code + 1); bits 28-29: rows per wave 2 / 4 / 1; bit 30: 2 waves per workgroup;
        #  bits 10-11 = 2: one persistent launch; bits 12-19: workgroups per CU)
        for v in (1 << 24, (1 << 24) | (1 << 8), (1 << 24) | (1 << 12), (1 << 24) | (1 << 25) | (2 << 10),
                  (1 << 24) | (2 << 25) | (1 << 12), (1 << 24) | (1 << 28) | (2 << 10) | (1 << 12),
                  (1 << 24) | (2 << 28) | (3 << 25), (1 << 24) | (1 << 30) | (2 << 10) | (1 << 12),
                  (1 << 24) | (1 << 30) | (1 << 25) | (2 << 12)):
            eng.gemv(dA3, dx3, yb, db3, gb, variant=v)
            assert H.same_bits(yb.cpu().numpy(), ya.cpu().numpy()) and H.same_bits(gb.cpu().numpy(), ga.cpu().numpy())
        yc = eng.empty(rows3 - rows3 // 2)
        eng.gemv(dA3[rows3 // 2:], dx3, yc, variant=1 << 24)
        assert H.same_bits(yc.cpu().numpy(), ya.cpu().numpy()[rows3 // 2:])
    # row blocks (what each rank of a row-sharded run computes)
    for r0, r1 in ((0, 100), (100, 389), (389, 777)):
        y = eng.empty(r1 - r0)
        eng.gemv(dA[r0:r1], dx, y)
        assert H.same_bits(y.cpu().numpy(), base[r0:r1])
    # unaligned views: odd leading dimension and odd offsets take the 8-byte load path
    big = eng.empty(rows * (cols + 1) + 3)
    Av = big[1:1 + rows * (cols + 1)].view(rows, cols + 1)[:, :cols]
    Av.copy_(dA)
    xb = eng.empty(cols + 1)
    xv = xb[1:]
    xv.copy_(dx)
    y = eng.empty(rows)
    eng.gemv(Av, xv, y)
    assert H.same_bits(y.cpu().numpy(), base)


@pytest.mark.parametrize("rows,cols", [(16, 64), (17, 128), (100, 1024), (777, 4096), (5, 64), (4096, 8192)])
def test_gemv_mfma_variant_against_fp64_reference(rows, cols, eng):
    """The `v_mfma_f64_16x16x4_f64` form of the matvec (variant bit 9; BASELINE config 3's "MFMA GEMV", kept as an
    A/B variant -- profiles/r02_gemv_mfma_ab.md).  Asymmetric data: a swapped row / column map cannot pass.  Same
    fp64 bound as the VALU form; the two forms sum a row's columns in different orders, so their last bits may
    differ -- by no more than twice that bound.  Row blocks of the MFMA form are bitwise invariant too."""
    import math
    rng = np.random.default_rng(rows * 7 + cols)
    A = rng.normal(size=(rows, cols)) + np.arange(rows)[:, None] * 0.01 - np.arange(cols)[None, :] * 0.003
    x = rng.normal(size=cols)
    b = rng.normal(size=rows)
    dA, dx, db = eng.to_device(A), eng.to_device(x), eng.to_device(b)
    exact = np.array([math.fsum(A[i] * x) for i in range(rows)])
    bound = _gemv_ref_bound(A, x)
    base = eng.empty(rows)
    eng.gemv(dA, dx, base)
    for U in (0, 1, 2, 4):
        y, g = eng.empty(rows), eng.empty(rows)
        eng.gemv(dA, dx, y, db, g, variant=0x200 | (U << 4))
        yy = y.cpu().numpy()
        assert np.all(np.abs(yy - exact) <= bound), U
        assert H.same_bits(g.cpu().numpy(), yy + b)
        assert np.all(np.abs(yy - base.cpu().numpy()) <= 2 * bound)
        if U == 0:
            first = yy
        else:
            assert H.same_bits(yy, first)                      # the unroll does not change the order
    if rows > 40:
        yb = eng.empty(rows - 32)
        eng.gemv(dA[32:], dx, yb, variant=0x200)
        assert H.same_bits(yb.cpu().numpy(), first[32:])
    # a shape the MFMA form does not cover (cols % 64 != 0) falls through to the VALU form: its bits
    y2 = eng.empty(rows)
    eng.gemv(dA[:, :cols - 2], dx, y2, variant=0x200)
    y3 = eng.empty(rows)
    eng.gemv(dA[:, :cols - 2], dx, y3)
    assert H.same_bits(y2.cpu().numpy(), y3.cpu().numpy())


def test_abi_error_reporting(eng):
    import ctypes
    from ccqppy_amd import _lib
    lib = _lib.load()
    rc = lib.ccqp_gemv_f64(eng.ctx, None, 4, None, None, 4, 8, 0, None)
    assert rc == 1 and b"lda" in lib.ccqp_last_error()
    rc = lib.ccqp_dots_f64(eng.ctx, 9, None, None, 4, 0, None)
    assert rc == 1
    x = eng.empty(8)
    rc = lib.ccqp_gemv_f64(eng.ctx, ctypes.c_void_p(x.data_ptr()), 8, ctypes.c_void_p(x.data_ptr()),
                           ctypes.c_void_p(x.data_ptr()), 1, 8, 0x3, None)
    assert rc != 0
    with pytest.raises(_lib.CCQPHipError):
        _lib.check(rc)


def test_abi_error_reporting_of_the_fused_and_pipelined_entry_points(eng):
    """Refused calls return CCQP_ERR_INVALID with a message and launch nothing."""
    import ctypes
    from ccqppy_amd import _lib, solution_spaces as ss
    lib = _lib.load()
    n = 100
    eng.bind()
    v = [eng.to_device(np.arange(n, dtype=np.float64) + i) for i in range(4)]
    box = ss.BoxProjOp(n, -np.ones(n), np.ones(n))
    seg = box._seg(eng, 0, n)
    P = ctypes.c_void_p
    one = (P * 1)(v[0].data_ptr())
    none = (P * 1)(None)
    # residual slot inside the dot slots
    rc = lib.ccqp_fused_dots_f64(eng.ctx, 1, one, none, one, none, n, 5, seg, P(v[0].data_ptr()), P(v[1].data_ptr()),
                                 1.0, 1e-6, 5, eng.stream)
    assert rc == 1 and b"overlaps" in lib.ccqp_last_error()
    # k out of range, residual on a sub-vector
    assert lib.ccqp_fused_dots_f64(eng.ctx, 9, one, none, one, none, n, 0, None, None, None, 0.0, 0.0, 0, eng.stream) == 1
    part = box._seg(eng, 10, 50)
    rc = lib.ccqp_fused_dots_f64(eng.ctx, 1, one, none, one, none, n, 0, part, P(v[0].data_ptr()), P(v[1].data_ptr()),
                                 1.0, 1e-6, 9, eng.stream)
    assert rc == 1 and b"whole vectors" in lib.ccqp_last_error()
    rc = lib.ccqp_bb_tail_elementwise_f64(eng.ctx, part, P(v[0].data_ptr()), P(v[1].data_ptr()), P(v[2].data_ptr()),
                                          P(v[3].data_ptr()), 1.0, 1e-6, 0, eng.stream)
    assert rc == 1 and b"offset must be 0" in lib.ccqp_last_error()
    # the two updates of axpby2 must be independent
    rc = lib.ccqp_axpby2_f64(eng.ctx, 1.0, P(v[0].data_ptr()), 1.0, P(v[1].data_ptr()), P(v[2].data_ptr()), 1.0,
                             P(v[2].data_ptr()), 1.0, P(v[3].data_ptr()), P(v[3].data_ptr()), n, eng.stream)
    assert rc == 1 and b"depend" in lib.ccqp_last_error()
    # the pipelined iteration is for elementwise sets on whole vectors
    sseg = _lib.Seg()
    sseg.kind, sseg.dim, sseg.offset = _lib.OP_SPHERE, n, 0
    rc = lib.ccqp_pipe_step_project_f64(eng.ctx, sseg, P(v[0].data_ptr()), P(v[1].data_ptr()), P(v[2].data_ptr()),
                                        eng.stream)
    assert rc == 1 and b"elementwise" in lib.ccqp_last_error()
    rc = lib.ccqp_pipe_bb_tail_f64(eng.ctx, part, P(v[0].data_ptr()), P(v[1].data_ptr()), P(v[2].data_ptr()),
                                   P(v[3].data_ptr()), 1.0, 1e-6, 1e-8, 1e-15, eng.stream)
    assert rc == 1 and b"whole vectors" in lib.ccqp_last_error()


import json as _json

with open(H.GOLD + "/pg_cases.json") as _f:
    PG = _json.load(_f)
PG_ARR = dict(np.load(H.GOLD + "/pg.npz"))


@pytest.mark.parametrize("case", PG, ids=[c["id"] for c in PG])
def test_projected_gradient_golden(case, eng):
    """8(f)2: free/chopped gradient split, bit-exact against the reference (incl. the Box condition's
    misplaced parenthesis and the None / NotImplementedError / TypeError / AttributeError paths)."""
    op = C.build_op(case["spec"], C.product_factory(), PG_ARR)
    r, err = _call(lambda: op.projected_gradient(PG_ARR[case["x"]].copy(), PG_ARR[case["g"]].copy()))
    assert err == case["err"]
    if err is None:
        assert (r is None) == case["none"]
        if r is not None:
            assert isinstance(r, tuple) and len(r) == 2
            assert H.same_bits(r[0], PG_ARR[case["id"] + ".free"])
            assert H.same_bits(r[1], PG_ARR[case["id"] + ".chop"])


@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 8192, 100003, (1 << 21) + 5])
def test_fused_iteration_kernels_have_the_bits_of_the_separate_launches(n, eng):
    """ccqp_fused_dots_f64, ccqp_bb_tail_elementwise_f64, ccqp_axpby2_f64 and ccqp_proj_elementwise_sub_f64 exist
    only to save launches: every result must be bit for bit what the sequence of plain kernels gives (one and
    several reduction blocks, with and without the last-block finish)."""
    from ccqppy_amd import solution_spaces as ss
    rng = np.random.default_rng(n)
    eng.bind()
    v = [eng.to_device(rng.normal(size=n)) for _ in range(6)]
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    lb[::3] = -0.1
    op = ss.BoxProjOp(n, lb, ub)
    seg = op._seg(eng, 0, n)
    c, gd = 1.0 / (3 * n * 1e-6), 1e-6

    def bits(slot0, count):
        return H.bits(np.array(eng.read(slot0, count)))

    # --- fused dots with differences + residual vs axpby / dots / residual
    d01 = eng.empty(n)
    eng.axpby(1.0, v[0], -1.0, v[1], d01)
    d23 = eng.empty(n)
    eng.axpby(1.0, v[2], -1.0, v[3], d23)
    eng.dots([(v[4], v[5]), (v[4], d01), (d01, d01), (d01, d23), (v[0], v[0])], 10)
    eng.residual_sq_elementwise(seg, v[0], v[2], c, gd, 20)
    want = np.concatenate([bits(10, 5), bits(20, 1)])
    eng.fused_dots([(v[4], v[5]), (v[4], (v[0], v[1])), ((v[0], v[1]), (v[0], v[1])), ((v[0], v[1]), (v[2], v[3])),
                    (v[0], v[0])], 30, (seg, v[0], v[2], c, gd, 40))
    assert np.array_equal(np.concatenate([bits(30, 5), bits(40, 1)]), want)
    eng.fused_dots([(v[4], v[5]), (v[4], (v[0], v[1]))], 50)          # without the residual row
    assert np.array_equal(bits(50, 2), want[:2])
    # --- BB tail vs residual + two dots on materialised differences
    eng.residual_sq_elementwise(seg, v[0], v[2], c, gd, 0)
    eng.dots([(d01, d01), (d01, d23)], 1)
    want = bits(0, 3)
    eng.bb_tail_elementwise(seg, v[0], v[1], v[2], v[3], c, gd, 5)
    assert np.array_equal(bits(5, 3), want)
    # --- two updates in one launch, in place like SPG's
    a, b_ = v[0].clone(), v[2].clone()
    a2, b2 = v[0].clone(), v[2].clone()
    eng.axpby(1.0, a, 0.37, v[1], a)
    eng.axpby(1.0, b_, 0.37, v[3], b_)
    eng.axpby2(1.0, a2, 0.37, v[1], a2, 1.0, b2, 0.37, v[3], b2)
    assert torch.equal(a, a2) and torch.equal(b_, b2)
    # --- P(x - alpha g) - x
    t, want_d, got_d = eng.empty(n), eng.empty(n), eng.empty(n)
    eng.proj_elementwise(seg, v[0], t, 1.0, -0.3, v[2])
    eng.axpby(1.0, t, -1.0, v[0], want_d)
    eng.proj_elementwise_sub(seg, v[0], got_d, 1.0, -0.3, v[2], v[0])
    assert H.same_bits(got_d.cpu().numpy(), want_d.cpu().numpy())
