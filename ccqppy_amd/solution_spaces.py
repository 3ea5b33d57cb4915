"""Projection operators onto convex feasible sets -- MI355X (HIP) implementation.

Drop-in for `ccqppy.solution_spaces` (reference: /root/reference/src/ccqppy/solution_spaces.py):
same class names, constructor arguments, attributes (`dim`, `lower_bound`, `upper_bound`,
`radius`, `aspect_ratio`, `proj_ops`), `name` strings, `embedded_dimension`, `__call__` and
`normal_vector`.  `op(x)` accepts a numpy array (copied to the GPU and back) or a torch tensor
on a ROCm device (zero copy) and returns the same kind; every FLOP runs in libccqp_hip.so.
There is no CPU fallback.  `projected_gradient` (dead code in the reference) and the
host-side `plot` helper are provided for API completeness.

Device protocol used by `ccqppy_amd.solvers` (all arguments are fp64 device tensors):
  _dev_project(eng, x, out)                  out = P(x)            (x is not out)
  _dev_step_project(eng, a, x, b, y, out)    out = P(a*x + b*y)
  _dev_residual_sq(eng, x, g, c, gd, slot)   slab[slot] = |c*(x - P(x - gd*g))|^2
  _dev_normal_vector(eng, x, out)            out = normal_vector(x)
"""
from abc import ABC, abstractmethod
import ctypes

import numpy as np
import torch

from . import _lib
from ._engine import get_engine

__all__ = ["ProjOpBase", "IdentityProjOp", "LowerBoundProjOp", "UpperBoundProjOp", "BoxProjOp",
           "SphereProjOp", "ConeProjOp", "DisjointProjOp", "HostCallableProjOp"]

# scalar-slab slots reserved for operator internals (solvers use 0..31)
_SLOT_NORM2 = 40   # |x_block|^2 of a Sphere/Cone block
_SLOT_DIST2 = 41   # |x - P(x)|^2 of a block (normal_vector)
_SLOT_LEN2 = 42    # |P(x)|^2 of a Sphere block (normal_vector)


def _is_scalar(b):
    return np.ndim(b) == 0


def _is_int(b):
    if isinstance(b, (bool, np.bool_)):
        return False
    if isinstance(b, (int, np.integer)):
        return True
    return isinstance(b, np.ndarray) and np.issubdtype(b.dtype, np.integer)


class _SegPlan:
    """A feasible set flattened for the segmented engine (csrc/ccqp_segs.hip): device table of block descriptors,
    work lists (one workgroup per item), and one zero-initialised workspace per engine that uses it."""

    def __init__(self, eng, segs):
        lib = eng.lib
        n = len(segs)
        host = (_lib.Seg * max(n, 1))()
        self._keep = []
        for i, sg in enumerate(segs):
            ctypes.memmove(ctypes.byref(host[i]), ctypes.byref(sg), ctypes.sizeof(_lib.Seg))
            self._keep.append(getattr(sg, "_keep", None))      # device copies of bounds stay alive with the plan
        nwork, nnorm, npart = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.ccqp_segs_plan(host, n, None, 0, ctypes.byref(nwork), None, 0, ctypes.byref(nnorm),
                                      ctypes.byref(npart)))
        work = (_lib.SegWork * max(nwork.value, 1))()
        normwork = (_lib.SegWork * max(nnorm.value, 1))()
        _lib.check(lib.ccqp_segs_plan(host, n, work, nwork.value, ctypes.byref(nwork), normwork, nnorm.value,
                                      ctypes.byref(nnorm), ctypes.byref(npart)))
        self.nseg, self.nwork, self.nnorm, self.npartials = n, nwork.value, nnorm.value, npart.value

        def up(buf):
            return torch.from_numpy(np.frombuffer(bytes(buf), dtype=np.uint8).copy()).to(eng.device)
        self.table, self.work, self.normwork = up(host), up(work), up(normwork)
        self.table_ptr, self.work_ptr, self.normwork_ptr = (self.table.data_ptr(), self.work.data_ptr(),
                                                            self.normwork.data_ptr())
        self.ws_doubles = int(lib.ccqp_segs_workspace_doubles(self.npartials, self.nwork))
        self._ws = {}

    def workspace(self, eng):
        # per ENGINE (= per stream): lane engines of one device share the read-only tables, never a workspace
        ws = self._ws.get(id(eng))
        if ws is None or ws[0]() is not eng:
            import weakref
            ws = (weakref.ref(eng), torch.zeros(self.ws_doubles, dtype=torch.float64, device=eng.device))
            if len(self._ws) > 128:
                self._ws = {k: v for k, v in self._ws.items() if v[0]() is not None}
            self._ws[id(eng)] = ws
        return ws[1]


class ProjOpBase(ABC):
    """Mirror of the reference ABC (solution_spaces.py:9-74)."""
    _kind = None            # CCQP_OP_* for leaf operators
    _version = 0            # bumped by _refresh() whenever a public parameter was found changed
    _fixed_dim = True       # False: the reference never uses `dim` inside __call__ (Identity/Sphere/Cone)

    def __init__(self, embedded_dimension):
        self.dim = embedded_dimension

    @property
    @abstractmethod
    def name(self):
        """Return the name of the projection operator."""

    @property
    def embedded_dimension(self):
        """Return the embedded dimension of the projection operator."""
        return self.dim

    # -------------------------------------------------------------- public entry points
    def __call__(self, x):
        eng, xd, n = self._stage_in(x)
        out = eng.empty(n)
        same = self._dev_project(eng, xd, out, want_alias=True)
        if same:
            return x          # the reference returns the argument object itself
        return self._stage_out(x, out)

    def normal_vector(self, x):
        eng, xd, n = self._stage_in(x)
        out = eng.empty(n)
        self._dev_normal_vector(eng, xd, out)
        return self._stage_out(x, out)

    def plot(self, num_random_samples, lower_bound, upper_bound):
        """Scatter-plot the projections of random samples of the box [lower_bound, upper_bound] (1-3 D).
        Host-side visualisation only (solution_spaces.py:48-74); matplotlib is imported on demand."""
        import matplotlib.pyplot as plt
        dim = self.embedded_dimension
        fig = plt.figure()
        ax = fig.add_subplot(projection='3d')
        for _ in range(num_random_samples):
            sample = np.random.rand(dim) * (upper_bound - lower_bound) + lower_bound
            p = np.asarray(self.__call__(sample))
            ax.scatter(p[0], p[1] if dim > 1 else 0.0, p[2] if dim > 2 else 0.0)
        plt.show()

    def _refresh(self):
        """Re-read the operator's public parameters (the reference reads `self.lower_bound`, `self.radius`, ... on
        every call, solution_spaces.py:200-201,276-277,363-366,430-433,484-492): if they were reassigned or edited
        IN PLACE since the device copies / descriptors were built, drop those caches.  Called at every public entry
        point (`op(x)`, `normal_vector`, `projected_gradient`, the start of a solve or of a batch); returns True if
        something changed.  Nothing can change in between -- one Python thread drives a solve."""
        return False

    def _stage_in(self, x):
        self._refresh()
        if isinstance(x, torch.Tensor):
            eng = get_engine(x.device if x.is_cuda else None)
        else:
            eng = get_engine()
        xd = eng.to_device(x).reshape(-1)
        n = xd.numel()
        if self._fixed_dim and n != self.dim:
            raise ValueError("operands could not be broadcast together with shapes (%d,) (%d,)" % (n, self.dim))
        return eng, xd, n

    @staticmethod
    def _stage_out(x, out):
        if isinstance(x, torch.Tensor):
            return out
        return out.cpu().numpy()

    # -------------------------------------------------------------- device protocol (generic parts)
    def _leaves(self, offset=0):
        return [(self, offset)]

    def _segplan(self, eng, n):
        """The set as a segment plan for vectors of length n (None: some block has no device descriptor, e.g. a host
        callable inside a DisjointProjOp -- the per-block path is taken)."""
        cache = self.__dict__.setdefault("_segplans", {})
        key = (eng.index, n)
        if key not in cache:
            segs = []
            for op, off in self._leaves():
                d = op.embedded_dimension if op is not self else n
                if getattr(op, "_kind", None) is None:
                    segs = None
                    break
                if d > 0:
                    segs.append(op._seg(eng, off, d))
            if len(cache) > 16:
                cache.clear()
            cache[key] = None if segs is None else _SegPlan(eng, segs)
        return cache[key]

    def _dev_project(self, eng, x, out, want_alias=False):
        if not want_alias:
            plan = self._segplan(eng, x.numel())
            if plan is not None:
                if x.data_ptr() == out.data_ptr():
                    raise ValueError("%s cannot project in place" % type(self).__name__)
                eng.segs_project(plan, 1.0, x, 0.0, None, None, out)
                return False
        return self._block_project(eng, x, out, 0, x.numel(), want_alias)

    def _dev_step_project(self, eng, a, x, b, y, out):
        plan = self._segplan(eng, x.numel())
        if plan is not None:
            eng.segs_project(plan, a, x, b, y, None, out)      # a*x + b*y formed on the fly, rounded like axpby
            return
        t = eng.scratch(x.numel(), "op.step")
        eng.axpby(a, x, b, y, t)
        self._dev_project(eng, t, out)

    def _dev_step_project_minus(self, eng, x, b, y, out):
        """out = P(x + b*y) - x  (SPG's direction, solvers.py:937)."""
        plan = self._segplan(eng, x.numel())
        if plan is not None:
            eng.segs_project(plan, 1.0, x, b, y, x, out)
            return
        t = eng.scratch(x.numel(), "op.stepm")
        self._dev_step_project(eng, 1.0, x, b, y, t)
        eng.axpby(1.0, t, -1.0, x, out)

    def _dev_residual_sq(self, eng, x, g, c, gd, slot):
        plan = self._segplan(eng, x.numel())
        if plan is not None:
            eng.segs_residual_sq(plan, x, g, c, gd, slot)
            return
        n = x.numel()
        t1 = eng.scratch(n, "op.res1")
        t2 = eng.scratch(n, "op.res2")
        eng.axpby(1.0, x, -gd, g, t1)      # x - gd*g
        self._dev_project(eng, t1, t2)     # P(.)
        eng.axpby(1.0, x, -1.0, t2, t1)    # x - P(.)
        eng.scale(c, t1, t1)               # c * (...)
        eng.dots([(t1, t1)], slot)

    def _has_device_form(self):
        """True if every block has a device descriptor (no user callable inside): cached until a refresh finds a change."""
        c = self.__dict__.get("_devform")
        if c is None:
            c = self.__dict__["_devform"] = all(getattr(op, "_kind", None) is not None for op, _ in self._leaves())
        return c

    def _normal_on_device(self):
        """False where normal_vector needs the host per block: a Cone block raises (solution_spaces.py:465), scalar
        bounds raise TypeError depending on the data (:318), a host callable is called."""
        return self._kind is not None

    def _normal_everywhere_on_device(self):
        """True if normal_vector of every block runs on the device (cached: a union may hold thousands of blocks)."""
        ok = self.__dict__.get("_normal_dev")
        if ok is None:
            ok = self.__dict__["_normal_dev"] = all(op._normal_on_device() for op, _ in self._leaves())
        return ok

    def _dev_normal_vector(self, eng, x, out):
        ok = self._normal_everywhere_on_device()
        if ok:
            plan = self._segplan(eng, x.numel())
            if plan is not None:
                eng.segs_normal(plan, x, out)
                return
        self._legacy_normal_vector(eng, x, out)

    def _legacy_normal_vector(self, eng, x, out):
        self._block_normal(eng, x, out, 0, x.numel())

    def _block_project(self, eng, x, out, offset, n, want_alias=False):
        raise NotImplementedError

    def _block_normal(self, eng, x, out, offset, n):
        raise NotImplementedError

    def _single_table(self, eng, offset, n):
        """Device table holding this one block, for the thread-per-block kernel (blocks of
        dimension <= CCQP_SMALL_SEG_MAX use its sequential-FMA norm: bit-exact vs numpy for <= 15)."""
        cache = self.__dict__.setdefault("_single_tables", {})
        key = (eng.index, offset, n)
        ent = cache.get(key)
        if ent is None:
            seg = self._seg(eng, offset, n)
            raw = np.frombuffer(bytes(seg), dtype=np.uint8).copy()
            ent = (torch.from_numpy(raw).to(eng.device), getattr(seg, "_keep", None))
            if len(cache) > 64:
                cache.clear()
            cache[key] = ent
        return ent[0]


# ------------------------------------------------------------------ elementwise family
class _ElementwiseOp(ProjOpBase):
    """Identity / Lower / Upper / Box: one fused kernel per use, literal numpy expression order."""

    def _bounds(self):
        return None, None

    @staticmethod
    def _snapshot(b):
        """Private copy of a bound for change detection: None / scalars as they are; arrays as (dtype, shape, bytes) --
        or (dtype, shape, array copy) beyond 16 KiB, where an integer-view comparison beats `tobytes`."""
        if b is None or _is_scalar(b):
            return b
        a = np.asarray(b)
        return (a.dtype, a.shape, a.tobytes() if a.nbytes <= 16384 else np.array(a, copy=True))

    @staticmethod
    def _unchanged(b, snap):
        """Bit-for-bit comparison of a bound with its snapshot (-0.0 vs +0.0 and int vs float matter to the mask
        arithmetic `bound * mask`).  ~0.3 us for the small arrays of a batch of problems, ~70 us for n = 131072."""
        if b is None or _is_scalar(b):
            if type(snap) is tuple:
                return False
            if b is None or snap is None:
                return b is snap
            return type(b) is type(snap) and (b == snap or (b != b and snap != snap)) \
                and np.signbit(b) == np.signbit(snap)
        if type(snap) is not tuple:
            return False
        a = b if type(b) is np.ndarray else np.asarray(b)
        dt, shp, data = snap
        if a.dtype != dt or a.shape != shp:
            return False
        if type(data) is bytes:
            return a.tobytes() == data
        if dt.itemsize == 8 and a.flags.c_contiguous:
            return bool(np.array_equal(a.view(np.int64), data.view(np.int64)))
        return a.tobytes() == data.tobytes()

    @staticmethod
    def _same_snapshot(a, b):
        """Two snapshots (of _snapshot) holding the same bits."""
        if a is b:
            return True
        if type(a) is not tuple or type(b) is not tuple:
            return False                                 # (None / scalars: re-checked against the bound itself, cheaply)
        if a[0] != b[0] or a[1] != b[1]:
            return False
        da = a[2] if type(a[2]) is bytes else a[2].tobytes()
        db = b[2] if type(b[2]) is bytes else b[2].tobytes()
        return da == db

    def _refresh(self, memo=None):
        """memo (a dict owned by ONE staging pass of a batch, during which no user code runs): operators built over the
        same bound OBJECTS share their snapshots, so the bit-for-bit comparison of a bound with its snapshot is made once
        per pass and bound object instead of once per operator (128 problems x two 64 KiB bounds: 2 ms otherwise)."""
        lb, ub = self._bounds()
        st = self.__dict__.get("_bounds_state")
        if memo is not None:
            hit = memo.get((id(lb), id(ub)))
            if hit is not None and hit[0] is lb and hit[1] is ub:
                # `hit` = the state of an operator just checked in this pass: lb / ub hold the bits of hit[2] / hit[3]
                if st is hit:
                    return False
                if st is not None and st[0] is lb and st[1] is ub and self._same_snapshot(st[2], hit[2]) \
                        and self._same_snapshot(st[3], hit[3]):
                    self.__dict__["_bounds_state"] = hit         # (share from now on)
                    return False
                if st is None:
                    self.__dict__["_bounds_state"] = hit
                    for k in ("_seg_cache", "_dev_bounds", "_single_tables", "_segplans", "_normal_dev"):
                        self.__dict__.pop(k, None)
                    return False
            changed = self._refresh()
            memo[(id(lb), id(ub))] = self.__dict__["_bounds_state"]
            return changed
        # the state holds references to the bound objects themselves, so the ids the caches are keyed on cannot be
        # recycled while the caches live; contents are compared bit for bit against the snapshot taken at upload
        if st is not None and st[0] is lb and st[1] is ub:
            s2, s3 = st[2], st[3]
            # fast path (a batch of many small problems calls this once per problem): small arrays, bytes snapshots
            if (lb is None or (type(lb) is np.ndarray and type(s2) is tuple and type(s2[2]) is bytes
                               and lb.dtype == s2[0] and lb.shape == s2[1] and lb.tobytes() == s2[2])) and \
                    (ub is None or (type(ub) is np.ndarray and type(s3) is tuple and type(s3[2]) is bytes
                                    and ub.dtype == s3[0] and ub.shape == s3[1] and ub.tobytes() == s3[2])):
                return False
            if self._unchanged(lb, s2) and self._unchanged(ub, s3):
                return False
        self.__dict__["_bounds_state"] = (lb, ub, self._snapshot(lb), self._snapshot(ub))
        for k in ("_seg_cache", "_dev_bounds", "_single_tables", "_segplans", "_normal_dev"):
            self.__dict__.pop(k, None)
        if st is not None:
            self._version = self._version + 1
        return st is not None

    def _seg(self, eng, offset, n):
        if "_bounds_state" not in self.__dict__:
            self._refresh()
        lb, ub = self._bounds()
        segs = self.__dict__.setdefault("_seg_cache", {})
        skey = (eng.index, offset, n, id(lb), id(ub))
        hit = segs.get(skey)
        if hit is not None:
            return hit                      # descriptors are read-only: built once per (device, block)
        seg = self._build_seg(eng, offset, n, lb, ub)
        if len(segs) > 256:
            segs.clear()
        segs[skey] = seg
        return seg

    def _build_seg(self, eng, offset, n, lb, ub):
        cache = self.__dict__.setdefault("_dev_bounds", {})
        key = (eng.index, id(lb), id(ub))
        ent = cache.get(key)
        if ent is None:
            def stage(b):
                if b is None or _is_scalar(b):
                    return None
                return eng.to_device(np.asarray(b)).reshape(-1)
            ent = (stage(lb), stage(ub))
            cache.clear()
            cache[key] = ent
        dlb, dub = ent
        seg = _lib.Seg()
        seg.kind = self._kind
        seg.flags = (_lib.SEG_LB_INT if _is_int(lb) else 0) | (_lib.SEG_UB_INT if _is_int(ub) else 0)
        seg.offset = offset
        seg.dim = n
        seg.lb = dlb.data_ptr() if dlb is not None else None
        seg.ub = dub.data_ptr() if dub is not None else None
        seg.lb_s = float(lb) if (lb is not None and _is_scalar(lb)) else 0.0
        seg.ub_s = float(ub) if (ub is not None and _is_scalar(ub)) else 0.0
        seg._keep = ent
        return seg

    def _block_project(self, eng, x, out, offset, n, want_alias=False):
        eng.proj_elementwise(self._seg(eng, offset, n), x, out)
        return False

    def _dev_project(self, eng, x, out, want_alias=False):
        return self._block_project(eng, x, out, 0, x.numel(), want_alias)

    def _normal_on_device(self):
        return not self._scalar_bound_names()

    def _dev_normal_vector(self, eng, x, out):
        self._block_normal(eng, x, out, 0, x.numel())

    def _dev_step_project(self, eng, a, x, b, y, out):
        eng.proj_elementwise(self._seg(eng, 0, x.numel()), x, out, a, b, y)

    def _dev_residual_sq(self, eng, x, g, c, gd, slot):
        eng.residual_sq_elementwise(self._seg(eng, 0, x.numel()), x, g, c, gd, slot)

    def _dev_step_project_minus(self, eng, x, b, y, out):
        """out = P(x + b*y) - x in one launch."""
        eng.proj_elementwise_sub(self._seg(eng, 0, x.numel()), x, out, 1.0, b, y, x)

    def _dev_fused_dots_residual(self, eng, pairs, slot0, x, g, c, gd, res_slot):
        """dot products (operands may be (u, v) = u - v) + residual^2(x, g) in one launch."""
        eng.fused_dots(pairs, slot0, (self._seg(eng, 0, x.numel()), x, g, c, gd, res_slot))

    def _dev_bb_tail(self, eng, x_new, x_old, g_new, g_old, c, gd, slot0):
        """residual^2 + the two Barzilai-Borwein dots in one launch (elementwise sets only)."""
        eng.bb_tail_elementwise(self._seg(eng, 0, x_new.numel()), x_new, x_old, g_new, g_old, c, gd, slot0)

    def _scalar_bound_names(self):
        lb, ub = self._bounds()
        return [type(b).__name__ for b in (ub, lb) if b is not None and _is_scalar(b)]

    def projected_gradient(self, x, g):
        """(free gradient, chopped gradient) at x; their sum is the projected gradient.
        solution_spaces.py:162-184, 238-260, 324-347 (unused by any live solver of the reference)."""
        eng, xd, n = self._stage_in(x)
        gd = eng.to_device(g).reshape(-1)
        free_g, chopped_g = eng.empty(n), eng.empty(n)
        self._block_projected_gradient(eng, xd, gd, free_g, chopped_g, 0, n)
        return (self._stage_out(x, free_g), self._stage_out(x, chopped_g))

    def _block_projected_gradient(self, eng, x, g, free_g, chopped_g, offset, n):
        scalars = self._scalar_bound_names()
        if scalars and n > 0:
            # `self.lower_bound[i]` inside the loop (or inside normal_vector) on a scalar bound
            raise TypeError("'%s' object is not subscriptable" % scalars[0])
        nv = eng.scratch(x.numel(), "op.pg.nv")
        self._block_normal(eng, x, nv, offset, n)
        eng.projected_gradient_elementwise(self._seg(eng, offset, n), x, g, nv, free_g, chopped_g)

    def _block_normal(self, eng, x, out, offset, n):
        # solution_spaces.py:146-160, 222-236, 306-322
        if 0 < n <= _lib.SMALL_SEG_MAX and not self._scalar_bound_names():
            eng.proj_small(self._single_table(eng, offset, n), 1, x, out, 1)
            return
        xp = eng.scratch(x.numel(), "op.nv.xp")
        d = eng.scratch(x.numel(), "op.nv.d")
        sl = slice(offset, offset + n)
        self._block_project(eng, x, xp, offset, n)
        eng.axpby(1.0, x[sl], -1.0, xp[sl], d[sl])
        eng.dots([(d[sl], d[sl])], _SLOT_DIST2)
        scalars = self._scalar_bound_names()
        if scalars and n > 0:
            # the reference indexes `self.upper_bound[i]`: TypeError for a scalar bound, but only
            # when x is not "outside" (it returns zeros before reaching the loop otherwise)
            dist2 = eng.read(_SLOT_DIST2, 1)[0]
            if np.isclose(np.sqrt(dist2), 0):
                raise TypeError("'%s' object is not subscriptable" % scalars[0])
        eng.normal_elementwise(self._seg(eng, offset, n), xp, out, _SLOT_DIST2)


class IdentityProjOp(_ElementwiseOp):
    """{x in R^n}.  solution_spaces.py:77-125."""
    _kind = _lib.OP_IDENTITY
    _fixed_dim = False

    def __init__(self, embedded_dimension):
        self.dim = embedded_dimension

    @property
    def name(self):
        return "Identity"

    def __call__(self, x):
        return x                       # same object, :125

    def normal_vector(self, x):
        # :92-98 returns np.zeros(self.dim) whatever x is
        if isinstance(x, torch.Tensor):
            return torch.zeros(self.dim, dtype=torch.float64, device=x.device)
        return np.zeros(self.dim)

    def _block_normal(self, eng, x, out, offset, n):
        eng.fill(0.0, out[offset:offset + n])

    def projected_gradient(self, x, g):
        return None                    # the reference's method body is a docstring only (:100-109)

    def _block_projected_gradient(self, eng, x, g, free_g, chopped_g, offset, n):
        # inside a DisjointProjOp the reference unpacks this None (:533)
        raise TypeError("cannot unpack non-iterable NoneType object")


class LowerBoundProjOp(_ElementwiseOp):
    """{x : x >= lb}.  solution_spaces.py:128-201; default lb = -1 (:134)."""
    _kind = _lib.OP_LOWER

    def __init__(self, embedded_dimension, lower_bound=None):
        self.dim = embedded_dimension
        self.lower_bound = lower_bound if lower_bound is not None else -np.ones(embedded_dimension)

    @property
    def name(self):
        return "Lower Bound"

    def _bounds(self):
        return self.lower_bound, None


class UpperBoundProjOp(_ElementwiseOp):
    """{x : x <= ub}.  solution_spaces.py:204-277; default ub = +1 (:210)."""
    _kind = _lib.OP_UPPER

    def __init__(self, embedded_dimension, upper_bound=None):
        self.dim = embedded_dimension
        self.upper_bound = upper_bound if upper_bound is not None else np.ones(embedded_dimension)

    @property
    def name(self):
        return "Upper Bound"

    def _bounds(self):
        return None, self.upper_bound


class BoxProjOp(_ElementwiseOp):
    """{x : lb <= x <= ub}.  solution_spaces.py:280-366; defaults -1 / +1 (:286,291)."""
    _kind = _lib.OP_BOX

    def __init__(self, embedded_dimension, lower_bound=None, upper_bound=None):
        self.dim = embedded_dimension
        self.lower_bound = lower_bound if lower_bound is not None else -np.ones(embedded_dimension)
        self.upper_bound = upper_bound if upper_bound is not None else np.ones(embedded_dimension)

    @property
    def name(self):
        return "Box"

    def _bounds(self):
        return self.lower_bound, self.upper_bound


# ------------------------------------------------------------------ norm-based family
class _NormOp(ProjOpBase):
    _fixed_dim = False

    def _param(self):
        raise NotImplementedError

    def _refresh(self):
        cur = self._param()
        st = self.__dict__.get("_param_state")
        if st is not None and st[0] is cur:
            return False
        if st is not None and type(st[0]) is type(cur) and _is_scalar(cur) and st[0] == cur:
            self.__dict__["_param_state"] = (cur,)
            return False
        self.__dict__["_param_state"] = (cur,)
        self.__dict__.pop("_single_tables", None)
        self.__dict__.pop("_segplans", None)
        if st is not None:
            self._version = self._version + 1
        return st is not None

    def _seg(self, eng, offset, n):
        raise NotImplementedError

    def _alias_test(self, r, x_last):
        raise NotImplementedError

    def _block_project(self, eng, x, out, offset, n, want_alias=False):
        if n == 0:
            return False
        if n <= _lib.SMALL_SEG_MAX:
            eng.proj_small(self._single_table(eng, offset, n), 1, x, out, 0)
            if want_alias:
                # the `return x` branches hand back the argument object (:435, :487); a block this
                # small is cheapest to compare after the fact
                a = x[offset:offset + n].view(torch.int64)
                return bool(torch.equal(a, out[offset:offset + n].view(torch.int64)))
            return False
        sl = x[offset:offset + n]
        eng.dots([(sl, sl)], _SLOT_NORM2)
        if want_alias:
            # `return x` branches of the reference hand back the argument object (:435, :487)
            r = np.sqrt(eng.read(_SLOT_NORM2, 1)[0])
            if self._alias_test(r, x, offset, n):
                return True
        eng.proj_normscale(self._seg(eng, offset, n), x, out, _SLOT_NORM2)
        return False


class SphereProjOp(_NormOp):
    """{x : |x| <= r}.  solution_spaces.py:369-435; default radius 1 (:375)."""
    _kind = _lib.OP_SPHERE

    def __init__(self, embedded_dimension, radius=None):
        self.dim = embedded_dimension
        self.radius = radius if radius is not None else 1

    @property
    def name(self):
        return "Sphere"

    def _param(self):
        return self.radius

    def _seg(self, eng, offset, n):
        seg = _lib.Seg()
        seg.kind = self._kind
        seg.flags = _lib.SEG_P0_INT if _is_int(self.radius) else 0
        seg.offset = offset
        seg.dim = n
        seg.lb_s = float(self.radius)
        seg.ub_s = 0.0
        return seg

    def _alias_test(self, r, x, offset, n):
        return not (r > self.radius)

    def projected_gradient(self, x, g):
        raise NotImplementedError("Cone proximal gradient not implemented, yet.")   # sic, :415

    def _block_projected_gradient(self, eng, x, g, free_g, chopped_g, offset, n):
        raise NotImplementedError("Cone proximal gradient not implemented, yet.")

    def _block_normal(self, eng, x, out, offset, n):
        # solution_spaces.py:389-403
        if n == 0:
            return
        if n <= _lib.SMALL_SEG_MAX:
            eng.proj_small(self._single_table(eng, offset, n), 1, x, out, 1)
            return
        xp = eng.scratch(x.numel(), "op.nv.xp")
        d = eng.scratch(x.numel(), "op.nv.d")
        sl = slice(offset, offset + n)
        self._block_project(eng, x, xp, offset, n)
        eng.axpby(1.0, x[sl], -1.0, xp[sl], d[sl])
        eng.dots([(d[sl], d[sl]), (xp[sl], xp[sl])], _SLOT_DIST2)   # DIST2, LEN2 are adjacent slots
        eng.normal_sphere(self._seg(eng, offset, n), xp, out, _SLOT_DIST2, _SLOT_LEN2)


class ConeProjOp(_NormOp):
    """{[x, z] : x.x <= mu^2 z^2} as coded in solution_spaces.py:438-492.  The reference flags this
    operator "bugged" (:439: the norm at :484 includes the last entry); it is reproduced as is."""
    _kind = _lib.OP_CONE

    def __init__(self, embedded_dimension, aspect_ratio=None):
        self.dim = embedded_dimension
        self.aspect_ratio = aspect_ratio if aspect_ratio is not None else 1

    @property
    def name(self):
        return "Cone"

    def _param(self):
        return self.aspect_ratio

    def _seg(self, eng, offset, n):
        mu = self.aspect_ratio
        seg = _lib.Seg()
        seg.kind = self._kind
        seg.flags = _lib.SEG_P0_INT if _is_int(mu) else 0
        seg.offset = offset
        seg.dim = n
        seg.lb_s = float(mu)
        seg.ub_s = float(mu ** 2 + 1)     # evaluated here with Python/numpy semantics (:491)
        return seg

    def _alias_test(self, r, x, offset, n):
        z = np.float64(x[offset + n - 1].item())
        return bool(self.aspect_ratio * z >= r)

    def normal_vector(self, x):
        raise NotImplementedError("Cone normal not implemented, yet.")   # :465

    def proximal_gradient(self, x, g):
        raise NotImplementedError("Cone proximal gradient not implemented, yet.")   # :468

    def _normal_on_device(self):
        return False

    def _block_normal(self, eng, x, out, offset, n):
        raise NotImplementedError("Cone normal not implemented, yet.")

    def _block_project(self, eng, x, out, offset, n, want_alias=False):
        if x.data_ptr() == out.data_ptr():
            raise ValueError("ConeProjOp cannot project in place")
        return super()._block_project(eng, x, out, offset, n, want_alias)


# ------------------------------------------------------------------ disjoint union
class DisjointProjOp(ProjOpBase):
    """Cartesian product of sub-operators on consecutive slices.  solution_spaces.py:495-560.

    Device plan: the union is flattened (recursively) to a table of block descriptors and runs through the
    segmented engine (csrc/ccqp_segs.hip) -- projection, fused step, residual in at most 2 launches, normal_vector in
    3, for any number of blocks of any size (the reference spends ~85 % of a solve in its Python loop over the
    blocks).  Only a union that holds a block without a device form (a user callable; a Cone or a
    scalar-bounded block under normal_vector, whose errors depend on the data) takes the per-block path."""

    def __init__(self, *convex_proj_ops):
        self.proj_ops = convex_proj_ops
        self.dim = 0
        for proj_op in self.proj_ops:
            self.dim += proj_op.embedded_dimension

    @property
    def name(self):
        return "DisjointUnion"

    def _wrapped(self, op):
        """Blocks may be anything callable with an `embedded_dimension` (solution_spaces.py:555-558 only calls them)."""
        if isinstance(op, ProjOpBase):
            return op
        cache = self.__dict__.setdefault("_foreign", {})
        w = cache.get(id(op))
        if w is None or w.fn is not op:
            w = cache[id(op)] = HostCallableProjOp(op.embedded_dimension, op)
        return w

    def _leaves(self, offset=0):
        out = []
        for op in self.proj_ops:
            out.extend(self._wrapped(op)._leaves(offset))
            offset += op.embedded_dimension
        return out

    def _refresh(self):
        ops = self.proj_ops
        st = self.__dict__.get("_ops_state")
        if st is None or st[0] is not ops:
            # the DISTINCT children (a union is often one block object repeated: `[SphereProjOp(3)] * 1000`)
            uniq = list({id(o): o for o in ops if isinstance(o, ProjOpBase)}.values())
            structural = st is not None
        else:
            uniq, structural = st[2], False
        for op in uniq:
            op._refresh()
        # a child may have been refreshed on its own in the meantime (it was called directly): compare versions
        vers = tuple(op._version for op in uniq)
        changed = structural or (st is not None and st[1] != vers)
        self.__dict__["_ops_state"] = (ops, vers, uniq)
        if changed:
            for k in ("_plans", "_segplans", "_devform", "_normal_dev"):
                self.__dict__.pop(k, None)
            self._version = self._version + 1
        return changed

    def _plan(self, eng, for_normal):
        cache = self.__dict__.setdefault("_plans", {})
        key = (eng.index, for_normal)
        plan = cache.get(key)
        if plan is not None:
            return plan
        small, large = [], []
        for op, off in self._leaves():
            d = op.embedded_dimension
            if d == 0:
                continue
            tabled = d <= _lib.SMALL_SEG_MAX and op._kind is not None
            if for_normal and tabled:
                if isinstance(op, ConeProjOp):
                    tabled = False         # raises when reached, like the reference
                elif isinstance(op, _ElementwiseOp) and op._scalar_bound_names():
                    tabled = False         # TypeError path needs the host
            (small if tabled else large).append((op, off))
        table = None
        keep = []
        if small:
            segs = (_lib.Seg * len(small))()
            for i, (op, off) in enumerate(small):
                s = op._seg(eng, off, op.embedded_dimension)
                keep.append(getattr(s, "_keep", None))
                ctypes.memmove(ctypes.byref(segs[i]), ctypes.byref(s), ctypes.sizeof(_lib.Seg))
            raw = np.frombuffer(bytes(segs), dtype=np.uint8).copy()
            table = torch.from_numpy(raw).to(eng.device)
        plan = (table, len(small), large, keep)
        cache[key] = plan
        return plan

    def __call__(self, x):
        eng, xd, n = self._stage_in(x)
        out = torch.zeros(n, dtype=torch.float64, device=eng.device)   # `np.zeros(self.dim)` :553
        self._dev_project(eng, xd, out)
        return self._stage_out(x, out)

    def _block_project(self, eng, x, out, offset, n, want_alias=False):
        # per-block path (a block without a device descriptor is present)
        if x.data_ptr() == out.data_ptr():
            raise ValueError("DisjointProjOp cannot project in place")
        table, nsmall, large, _ = self._plan(eng, False)
        if nsmall:
            eng.proj_small(table, nsmall, x, out, 0)
        for op, off in large:
            op._block_project(eng, x, out, off, op.embedded_dimension)
        return False

    def _legacy_normal_vector(self, eng, x, out):
        table, nsmall, large, _ = self._plan(eng, True)
        if nsmall:
            eng.proj_small(table, nsmall, x, out, 1)
        for op, off in large:
            op._block_normal(eng, x, out, off, op.embedded_dimension)

    def projected_gradient(self, x, g):
        """Concatenation of the sub-operators' (free, chopped) gradients, solution_spaces.py:527-538."""
        eng, xd, n = self._stage_in(x)
        gd = eng.to_device(g).reshape(-1)
        free_g = torch.zeros(n, dtype=torch.float64, device=eng.device)
        chopped_g = torch.zeros(n, dtype=torch.float64, device=eng.device)
        for op, off in self._leaves():
            if not hasattr(op, "_block_projected_gradient"):
                raise AttributeError("'%s' object has no attribute 'projected_gradient'" % type(op).__name__)
            op._block_projected_gradient(eng, xd, gd, free_g, chopped_g, off, op.embedded_dimension)
        return (self._stage_out(x, free_g), self._stage_out(x, chopped_g))


# ------------------------------------------------------------------ arbitrary host callables
class HostCallableProjOp(ProjOpBase):
    """Any Python callable `x_proj = fn(x)` on numpy arrays as the feasible set of a solve.

    The reference accepts whatever is callable as `convex_proj_op` (solvers.py:115-116 and the same lines of every
    solver: the object is only ever called, plus `.normal_vector(x)` under MPRGP).  `CCQPSolver*.solve` wraps such an
    object in this class: every projection copies the vector to the host, calls `fn` there and copies the result
    back -- two PCIe transfers and a stream synchronisation per projection, i.e. SLOW, and documented as such.  It is
    what makes a user-defined set run at all; it is not a fallback of any of the seven built-in operators, which
    never take this route."""
    _fixed_dim = False

    def __init__(self, embedded_dimension, fn):
        if not callable(fn):
            raise TypeError("'%s' object is not callable" % type(fn).__name__)
        self.dim = embedded_dimension
        self.fn = fn

    @property
    def name(self):
        return str(getattr(self.fn, "name", getattr(self.fn, "__name__", type(self.fn).__name__)))

    def __call__(self, x):
        if isinstance(x, torch.Tensor):
            out = self._host_call(self.fn, x)
            return torch.from_numpy(out).to(x.device)
        return self.fn(x)

    def normal_vector(self, x):
        nv = self._normal_fn()
        if isinstance(x, torch.Tensor):
            return torch.from_numpy(self._host_call(nv, x)).to(x.device)
        return nv(x)

    def _normal_fn(self):
        nv = getattr(self.fn, "normal_vector", None)
        if nv is None:          # what the reference's MPRGP raises on `convex_proj_op.normal_vector(xk)` (:1095)
            raise AttributeError("'%s' object has no attribute 'normal_vector'" % type(self.fn).__name__)
        return nv

    @staticmethod
    def _host_call(fn, x):
        xh = x.detach().cpu().numpy()          # (synchronises the stream x was produced on)
        out = np.ascontiguousarray(np.asarray(fn(xh), dtype=np.float64)).reshape(-1)
        if out.shape[0] != xh.shape[0]:
            raise ValueError("operands could not be broadcast together with shapes (%d,) (%d,)"
                             % (xh.shape[0], out.shape[0]))
        return out

    def _round_trip(self, fn, eng, x, out):
        flush = getattr(eng, "flush_self", None)
        if flush is not None:
            flush()                            # recording engines: queued work first, in program order
        with torch.cuda.stream(eng.torch_stream):
            res = self._host_call(fn, x)
            out.copy_(torch.from_numpy(res), non_blocking=False)

    def _dev_project(self, eng, x, out, want_alias=False):
        self._round_trip(self.fn, eng, x, out)
        return False

    def _dev_normal_vector(self, eng, x, out):
        self._round_trip(self._normal_fn(), eng, x, out)

    # as a block of a DisjointProjOp
    def _block_project(self, eng, x, out, offset, n, want_alias=False):
        self._round_trip(self.fn, eng, x[offset:offset + n], out[offset:offset + n])
        return False

    def _block_normal(self, eng, x, out, offset, n):
        self._round_trip(self._normal_fn(), eng, x[offset:offset + n], out[offset:offset + n])
