"""Per-device engine: thin, typed Python front of the C ABI (include/ccqp_hip.h).

PyTorch is plumbing only: device memory (torch tensors), HIP streams and, for the row-sharded
operator, torch.distributed (RCCL).  All arithmetic runs in libccqp_hip.so.  Scalars that steer
the solvers' control flow are read back into numpy float64 scalars so the host logic uses exactly
the arithmetic types of the reference's Python loops.

An Engine owns one C-ABI context (scalar slab + reduction workspace) and launches on ONE stream:
  * the default engine of a device follows torch's current stream (re-bound at every public entry);
  * "lane" engines (batch-sharded mode, ccqppy_amd/batch.py) own a private stream each, so several
    independent solves can be in flight on one GPU while the host round-robins between them.
"""
import ctypes

import numpy as np
import torch

from . import _lib

import weakref

_engines = {}
_live = weakref.WeakSet()  # every engine alive (lane engines are cached by their schedulers): set_default_option reaches them
_default_options = {}      # device index -> {option name: value}: what every NEW engine (lane engines too) of the device starts with


def _require_gpu():
    if not torch.cuda.is_available():
        raise _lib.CCQPHipUnavailable(
            "ccqppy_amd: no ROCm GPU visible to PyTorch.  The CCQP hot path only runs on the HIP "
            "engine (gfx950); there is no CPU fallback.")


def _index_of(device):
    if device is None:
        return torch.cuda.current_device()
    if isinstance(device, int):
        return device
    dev = torch.device(device)
    return torch.cuda.current_device() if dev.index is None else dev.index


def get_engine(device=None):
    """Default engine of a torch device / index (default: the current device), bound to torch's
    current stream on that device."""
    _require_gpu()
    idx = _index_of(device)
    eng = _engines.get(idx)
    if eng is None:
        eng = Engine(idx)
        _engines[idx] = eng
    eng.bind()
    return eng


def set_default_option(name, value, device=None):
    """Tuning option `name` (ccqppy_amd._lib.OPTIONS; 0 = the tuned default) for the default engine of `device` and for
    every engine created on that device afterwards (the lanes of the batch schedulers).  The A/B switches of tools/ and
    tests/: no option changes a result bit, and nothing in the library reads the environment."""
    _require_gpu()
    idx = _index_of(device)
    if name not in _lib.OPTIONS:
        raise KeyError("no such option %r (known: %s)" % (name, ", ".join(sorted(_lib.OPTIONS))))
    _default_options.setdefault(idx, {})[name] = int(value)
    get_engine(idx)
    for eng in list(_live):
        if eng.index == idx:
            eng.set_option(name, value)


def make_lanes(device, count):
    """`count` engines on `device`, each with its own context and its own HIP stream."""
    _require_gpu()
    idx = _index_of(device)
    return [Engine(idx, torch.cuda.Stream(device=idx)) for _ in range(count)]


class Engine:
    def __init__(self, index, stream=None):
        self.lib = lib = _lib.load()
        self.index = index
        self.device = torch.device("cuda", index)
        h = ctypes.c_void_p()
        with torch.cuda.device(index):
            _lib.check(lib.ccqp_ctx_create(index, ctypes.byref(h)))
        self.ctx = h
        dev = ctypes.c_int()
        ncu = ctypes.c_int()
        buf = ctypes.create_string_buffer(64)
        _lib.check(lib.ccqp_ctx_info(self.ctx, ctypes.byref(dev), ctypes.byref(ncu), buf, 64))
        self.num_cu = ncu.value
        self.arch = buf.value.decode()
        self._host = (ctypes.c_double * _lib.NUM_SLOTS)()
        self._scratch = {}
        self.gemv_variant = 0
        self.mv_executed = 0      # every matvec launched through this engine (counted or not)
        self.products_launched = 0  # every A.v a launch computed (never corrected: a product computed ahead and dropped stays)
        self.gemv_events = None   # list of (start, end) HIP events per matvec when profiling is on
        for name, value in _default_options.get(index, {}).items():
            self.set_option(name, value)
        _live.add(self)
        self.own_stream = stream  # torch.cuda.Stream of a lane engine, None for the default engine
        self.torch_stream = stream
        self.stream = None        # raw hipStream_t (int) every launch goes to
        self.bind()

    def __del__(self):
        try:
            self.lib.ccqp_ctx_destroy(self.ctx)
        except Exception:  # noqa: BLE001
            pass

    # ------------------------------------------------------------------ tuning options (per context, never the environment)
    def set_option(self, name, value):
        _lib.check(self.lib.ccqp_ctx_set_option(self.ctx, _lib.OPTIONS[name], int(value)))

    def get_option(self, name):
        v = ctypes.c_int64()
        _lib.check(self.lib.ccqp_ctx_get_option(self.ctx, _lib.OPTIONS[name], ctypes.byref(v)))
        return v.value

    def options(self):
        """{name: value} of the options that are not at their default -- what makes a measurement self-describing."""
        return {name: v for name in _lib.OPTIONS for v in [self.get_option(name)] if v}

    # ------------------------------------------------------------------ stream / memory plumbing
    def bind(self):
        """(Re)bind the launch stream: the lane's own stream, else torch's current stream."""
        if self.own_stream is None:
            self.torch_stream = torch.cuda.current_stream(self.device)
        self.stream = self.torch_stream.cuda_stream

    def empty(self, n):
        return torch.empty(int(n), dtype=torch.float64, device=self.device)

    def scratch(self, n, tag):
        key = (int(n), tag)
        t = self._scratch.get(key)
        if t is None:
            t = self.empty(n)
            self._scratch[key] = t
        return t

    def to_device(self, a):
        """numpy array / torch tensor -> contiguous fp64 tensor on this device (ints upcast once,
        as numpy does inside `A.dot(x) + b`)."""
        if isinstance(a, torch.Tensor):
            if a.device != self.device or a.dtype != torch.float64 or not a.is_contiguous():
                a = a.to(device=self.device, dtype=torch.float64).contiguous()
            return a
        arr = np.ascontiguousarray(a, dtype=np.float64)
        return torch.from_numpy(arr).to(self.device)

    # ------------------------------------------------------------------ scalars
    def read(self, slot0, count):
        """Blocking read of `count` slab slots -> list of np.float64 (a synchronisation point)."""
        rc = self.lib.ccqp_scalars_read(self.ctx, slot0, count, self._host, self.stream)
        if rc:
            _lib.check(rc)
        h = self._host
        return [np.float64(h[i]) for i in range(count)]

    def read_async(self, slot0, count):
        """Enqueue the device->pinned copy of the slots (+ the context's event); fetch() completes it."""
        rc = self.lib.ccqp_scalars_read_async(self.ctx, slot0, count, self.stream)
        if rc:
            _lib.check(rc)

    def fetch(self, slot0, count):
        """Wait for the pending read_async (no-op if it has landed) and return the values."""
        rc = self.lib.ccqp_scalars_host(self.ctx, slot0, count, self._host)
        if rc:
            _lib.check(rc)
        h = self._host
        return [np.float64(h[i]) for i in range(count)]

    # ------------------------------------------------------------------ a1 matvec
    def gemv_plan(self, rows, cols, variant=None):
        """(rows per kernel launch, launches) of a rows x cols matvec."""
        rpl, nl = ctypes.c_int64(), ctypes.c_int64()
        v = self.gemv_variant if variant is None else variant
        _lib.check(self.lib.ccqp_gemv_plan(self.ctx, rows, cols, v, ctypes.byref(rpl), ctypes.byref(nl)))
        return rpl.value, nl.value

    def gemv2(self, A, x1, x2, y1, y2, b=None, g1=None, g2=None, variant=None):
        """y1 = A.x1 and y2 = A.x2 (g_i = y_i + b where given) in ONE pass over A where the LDS-staged form applies
        (ccqp_gemv2_f64), each with the bits of its own gemv(); otherwise two single matvecs.  Counts as TWO executed
        matvecs; one (start, end) event pair when profiling is on."""
        rows, cols = A.shape
        assert A.stride(1) == 1 and x1.numel() >= cols and x2.numel() >= cols and y1.numel() >= rows and y2.numel() >= rows
        ev = None
        if self.gemv_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(self.torch_stream)
        _lib.check(self.lib.ccqp_gemv2_f64(self.ctx, A.data_ptr(), A.stride(0), x1.data_ptr(), x2.data_ptr(),
                                          0 if b is None else b.data_ptr(), y1.data_ptr(),
                                          0 if g1 is None else g1.data_ptr(), y2.data_ptr(),
                                          0 if g2 is None else g2.data_ptr(), rows, cols,
                                          self.gemv_variant if variant is None else variant, self.stream))
        if ev is not None:
            ev[1].record(self.torch_stream)
            self.gemv_events.append(ev)
        self.mv_executed += 2
        self.products_launched += 2

    def gemv3(self, A, xs, ys, b=None, gs=(None, None, None), variant=None):
        """ys[j] = A.xs[j] (gs[j] = ys[j] + b where given) for the j with ys[j] not None -- up to three products of a
        SQUARE matrix in ONE pass over it where the LDS-staged form applies (ccqp_gemv3_f64), each with the bits of its
        own gemv().  Counts one executed matvec per product; one (start, end) event pair when profiling is on."""
        n = A.shape[0]
        assert A.shape[1] == n and A.stride(1) == 1
        arr = ctypes.c_void_p * 3
        px = arr(*[None if x is None else x.data_ptr() for x in xs])
        py = arr(*[None if y is None else y.data_ptr() for y in ys])
        pg = arr(*[None if g is None else g.data_ptr() for g in gs])
        ev = None
        if self.gemv_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(self.torch_stream)
        _lib.check(self.lib.ccqp_gemv3_f64(self.ctx, A.data_ptr(), A.stride(0), px, 0 if b is None else b.data_ptr(), py, pg,
                                          n, self.gemv_variant if variant is None else variant, self.stream))
        if ev is not None:
            ev[1].record(self.torch_stream)
            self.gemv_events.append(ev)
        self.mv_executed += sum(y is not None for y in ys)
        self.products_launched += sum(y is not None for y in ys)

    def gemvk(self, A, xs, ys, b=None, gs=None, variant=None):
        """ys[j] = A.xs[j] (gs[j] = ys[j] + b where given) for k = len(xs) <= 3 products of a rows x cols BLOCK in one
        pass where the pair kernel would take one (ccqp_gemvk_f64), each with the bits of its own gemv().  Counts k
        executed matvecs; one (start, end) event pair when profiling is on."""
        k = len(xs)
        rows, cols = A.shape
        gs = [None] * k if gs is None else gs
        assert 1 <= k <= 3 and len(ys) == k and len(gs) == k and A.stride(1) == 1
        assert all(x.numel() >= cols for x in xs) and all(y.numel() >= rows for y in ys)
        arr = ctypes.c_void_p * 3
        px = arr(*[x.data_ptr() for x in xs])
        py = arr(*[y.data_ptr() for y in ys])
        pg = arr(*[None if g is None else g.data_ptr() for g in gs])
        ev = None
        if self.gemv_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(self.torch_stream)
        _lib.check(self.lib.ccqp_gemvk_f64(self.ctx, A.data_ptr(), A.stride(0), k, px, 0 if b is None else b.data_ptr(), py,
                                          pg, rows, cols, self.gemv_variant if variant is None else variant, self.stream))
        if ev is not None:
            ev[1].record(self.torch_stream)
            self.gemv_events.append(ev)
        self.mv_executed += k
        self.products_launched += k

    def gemvk_form(self, k, rows, cols, variant=None):
        buf = ctypes.create_string_buffer(64)
        _lib.check(self.lib.ccqp_gemvk_form(self.ctx, k, rows, cols, self.gemv_variant if variant is None else variant, buf, 64))
        return buf.value.decode()

    def unpack_gathered(self, gathered, k, per, ys, b=None, gs=None):
        """ys[j][i] = gathered[(r k + j) per + i - r per], r = i // per (the slabs of k row-sharded products after their
        ONE all-gather), gs[j] = ys[j] + b where given -- one launch (ccqp_unpack_gathered_f64)."""
        n = ys[0].numel()
        gs = [None] * k if gs is None else gs
        arr = ctypes.c_void_p * 3
        py = arr(*[y.data_ptr() for y in ys])
        pg = arr(*[None if g is None else g.data_ptr() for g in gs])
        _lib.check(self.lib.ccqp_unpack_gathered_f64(self.ctx, gathered.data_ptr(), k, per, n, py,
                                                    0 if b is None else b.data_ptr(), pg, self.stream))

    def gemv3_form(self, n, variant=None):
        """"bp_gemv3_pers_kernel" (one pass for up to three products of an n x n matrix) or "single matvecs"."""
        buf = ctypes.create_string_buffer(64)
        _lib.check(self.lib.ccqp_gemv3_form(self.ctx, n, self.gemv_variant if variant is None else variant, buf, 64))
        return buf.value.decode()

    def gemv2_form(self, rows, cols, variant=None):
        """"gemv_lds2_kernel" (one pass for two products) or "two single matvecs"."""
        buf = ctypes.create_string_buffer(64)
        _lib.check(self.lib.ccqp_gemv2_form(self.ctx, rows, cols, self.gemv_variant if variant is None else variant, buf, 64))
        return buf.value.decode()

    def gemv_form(self, rows, cols, variant=None):
        """Name of the kernel a rows x cols matvec runs as (aligned operands assumed)."""
        buf = ctypes.create_string_buffer(64)
        v = self.gemv_variant if variant is None else variant
        _lib.check(self.lib.ccqp_gemv_form(self.ctx, rows, cols, v, buf, 64))
        return buf.value.decode()

    def gemv(self, A, x, y, b=None, g=None, variant=None):
        """y = A.x for a (rows, cols) row-major device matrix; optionally g = y + b."""
        rows, cols = A.shape
        assert A.stride(1) == 1 and x.numel() >= cols and y.numel() >= rows
        v = self.gemv_variant if variant is None else variant
        self.mv_executed += 1
        self.products_launched += 1
        ev = None
        if self.gemv_events is not None:
            # HIP events on the stream the kernel is launched on
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(self.torch_stream)
        if g is None:
            rc = self.lib.ccqp_gemv_f64(self.ctx, A.data_ptr(), A.stride(0), x.data_ptr(), y.data_ptr(), rows,
                                        cols, v, self.stream)
        else:
            rc = self.lib.ccqp_gemv_epilogue_f64(self.ctx, A.data_ptr(), A.stride(0), x.data_ptr(), b.data_ptr(),
                                                 y.data_ptr(), g.data_ptr(), rows, cols, v, self.stream)
        if rc:
            _lib.check(rc)
        if ev is not None:
            ev[1].record(self.torch_stream)
            self.gemv_events.append(ev)

    def spmv_plan(self, rowptr, col, nnz, local_x=True):
        """One-time analysis of a CSR matrix -> (plan tensor, host summary {rows on the wave-per-row path,
        windows in transposed gather order, local-x windows, diagonal windows} as a ctypes int64[4])."""
        rows = rowptr.numel() - 1
        words = self.lib.ccqp_spmv_plan_words(rows, nnz)
        plan = torch.empty(words, dtype=torch.int64, device=self.device)
        summary = (ctypes.c_int64 * 4)()
        rc = self.lib.ccqp_spmv_plan_csr(self.ctx, rowptr.data_ptr(), col.data_ptr() if (nnz and local_x) else None, rows, nnz,
                                         plan.data_ptr(), summary, self.stream)
        if rc:
            _lib.check(rc)
        return plan, summary

    def spmv_csr(self, rowptr, col, val, plan, summary, cols, x, y, b=None, g=None, width=0):
        """y = A.x for a CSR matrix (rowptr int64[rows+1], col int32[nnz], val float64[nnz]); g = y + b."""
        rows = rowptr.numel() - 1
        assert x.numel() >= cols and y.numel() >= rows
        self.mv_executed += 1
        self.products_launched += 1
        ev = None
        if self.gemv_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(self.torch_stream)
        rc = self.lib.ccqp_spmv_csr_f64(self.ctx, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), plan.data_ptr(),
                                        summary, x.data_ptr(), None if g is None else b.data_ptr(), y.data_ptr(),
                                        None if g is None else g.data_ptr(), rows, cols, val.numel(), width,
                                        self.stream)
        if rc:
            _lib.check(rc)
        if ev is not None:
            ev[1].record(self.torch_stream)
            self.gemv_events.append(ev)

    def csr_violations(self, rowptr, col, cols):
        """Number of violated CSR invariants (blocking read of one scalar)."""
        rc = self.lib.ccqp_csr_validate(self.ctx, rowptr.data_ptr(), col.data_ptr(), rowptr.numel() - 1, cols,
                                        col.numel(), 0, self.stream)
        if rc:
            _lib.check(rc)
        return int(self.read(0, 1)[0])

    # ------------------------------------------------------------------ a2 elementwise
    def axpby(self, a, x, b, y, out):
        rc = self.lib.ccqp_axpby_f64(self.ctx, a, x.data_ptr(), b, y.data_ptr(), out.data_ptr(), out.numel(),
                                     self.stream)
        if rc:
            _lib.check(rc)

    def axpby2(self, a1, x1, b1, y1, out1, a2, x2, b2, y2, out2):
        """out1 = a1*x1 + b1*y1 and out2 = a2*x2 + b2*y2 in one launch."""
        rc = self.lib.ccqp_axpby2_f64(self.ctx, a1, x1.data_ptr(), b1, y1.data_ptr(), out1.data_ptr(), a2,
                                      x2.data_ptr(), b2, y2.data_ptr(), out2.data_ptr(), out1.numel(), self.stream)
        if rc:
            _lib.check(rc)

    def proj_elementwise_sub(self, seg, x, out, a, b, y, z):
        """out = P(a*x + b*y) - z for an elementwise set."""
        rc = self.lib.ccqp_proj_elementwise_sub_f64(self.ctx, seg, a, x.data_ptr(), b, y.data_ptr(), z.data_ptr(),
                                                    out.data_ptr(), self.stream)
        if rc:
            _lib.check(rc)

    def scale(self, a, x, out):
        rc = self.lib.ccqp_scale_f64(self.ctx, a, x.data_ptr(), out.data_ptr(), out.numel(), self.stream)
        if rc:
            _lib.check(rc)

    def copy(self, x, out):
        rc = self.lib.ccqp_copy_f64(self.ctx, x.data_ptr(), out.data_ptr(), out.numel(), self.stream)
        if rc:
            _lib.check(rc)

    def fill(self, value, out):
        rc = self.lib.ccqp_fill_f64(self.ctx, value, out.data_ptr(), out.numel(), self.stream)
        if rc:
            _lib.check(rc)

    def mul(self, x, y, out):
        rc = self.lib.ccqp_mul_f64(self.ctx, x.data_ptr(), y.data_ptr(), out.data_ptr(), out.numel(), self.stream)
        if rc:
            _lib.check(rc)

    def muldiv(self, x, y, s, out):
        rc = self.lib.ccqp_muldiv_f64(self.ctx, x.data_ptr(), y.data_ptr(), s, out.data_ptr(), out.numel(),
                                      self.stream)
        if rc:
            _lib.check(rc)

    def submul(self, x, y, z, out):
        rc = self.lib.ccqp_submul_f64(self.ctx, x.data_ptr(), y.data_ptr(), z.data_ptr(), out.data_ptr(),
                                      out.numel(), self.stream)
        if rc:
            _lib.check(rc)

    # ------------------------------------------------------------------ a3 reductions
    def dots(self, pairs, slot0=0):
        """slab[slot0 + j] = pairs[j][0] . pairs[j][1]   (asynchronous)."""
        k = len(pairs)
        n = pairs[0][0].numel()
        xs = (ctypes.c_void_p * k)(*[p[0].data_ptr() for p in pairs])
        ys = (ctypes.c_void_p * k)(*[p[1].data_ptr() for p in pairs])
        rc = self.lib.ccqp_dots_f64(self.ctx, k, xs, ys, n, slot0, self.stream)
        if rc:
            _lib.check(rc)

    def dots_now(self, pairs, slot0=0):
        self.dots(pairs, slot0)
        return self.read(slot0, len(pairs))

    # ------------------------------------------------------------------ projections
    def proj_elementwise(self, seg, x, out, a=1.0, b=0.0, y=None):
        rc = self.lib.ccqp_proj_elementwise_f64(self.ctx, seg, a, x.data_ptr(), b,
                                                None if y is None else y.data_ptr(), out.data_ptr(), self.stream)
        if rc:
            _lib.check(rc)

    def proj_normscale(self, seg, x, out, slot):
        rc = self.lib.ccqp_proj_normscale_f64(self.ctx, seg, x.data_ptr(), out.data_ptr(), slot, self.stream)
        if rc:
            _lib.check(rc)

    def proj_small(self, table, nseg, x, out, mode):
        rc = self.lib.ccqp_proj_small_segments_f64(self.ctx, table.data_ptr(), nseg, x.data_ptr(), out.data_ptr(),
                                                   mode, self.stream)
        if rc:
            _lib.check(rc)

    # segmented engine (ccqp_segs.hip): any feasible set in <= 2 launches; `plan` is a solution_spaces._SegPlan
    def segs_project(self, plan, a, x, b, y, z, out):
        """out = P(a*x + b*y) - z   (y None: P(x); z None: no subtraction)."""
        rc = self.lib.ccqp_segs_project_f64(self.ctx, plan.table_ptr, plan.work_ptr, plan.nwork, plan.normwork_ptr,
                                            plan.nnorm, plan.npartials, plan.workspace(self).data_ptr(), a,
                                            x.data_ptr(), b, None if y is None else y.data_ptr(),
                                            None if z is None else z.data_ptr(), out.data_ptr(), self.stream)
        if rc:
            _lib.check(rc)

    def segs_residual_sq(self, plan, x, g, c, gd, slot):
        rc = self.lib.ccqp_segs_residual_sq_f64(self.ctx, plan.table_ptr, plan.work_ptr, plan.nwork,
                                                plan.normwork_ptr, plan.nnorm, plan.npartials,
                                                plan.workspace(self).data_ptr(), x.data_ptr(), g.data_ptr(), c, gd,
                                                slot, self.stream)
        if rc:
            _lib.check(rc)

    def segs_normal(self, plan, x, out):
        rc = self.lib.ccqp_segs_normal_f64(self.ctx, plan.table_ptr, plan.work_ptr, plan.nwork, plan.normwork_ptr,
                                           plan.nnorm, plan.npartials, plan.workspace(self).data_ptr(), x.data_ptr(),
                                           out.data_ptr(), self.stream)
        if rc:
            _lib.check(rc)

    def segs_feasible_steps(self, plan, x, p, alpha0, K, slot0):
        """slab[slot0 + k] = #{i: not isclose(yf_i, P(yf)_i)}, yf = x - (alpha0 * 0.5^k) * p, k < K -- any feasible set."""
        rc = self.lib.ccqp_segs_feasible_steps_f64(self.ctx, plan.table_ptr, plan.work_ptr, plan.nwork,
                                                   plan.normwork_ptr, plan.nnorm, plan.npartials,
                                                   plan.workspace(self).data_ptr(), x.data_ptr(), p.data_ptr(), alpha0,
                                                   K, slot0, self.stream)
        if rc:
            _lib.check(rc)

    def residual_sq_elementwise(self, seg, x, g, c, gd, slot):
        rc = self.lib.ccqp_residual_sq_elementwise_f64(self.ctx, seg, x.data_ptr(), g.data_ptr(), c, gd, slot,
                                                       self.stream)
        if rc:
            _lib.check(rc)

    def fused_dots(self, pairs, slot0=0, residual=None):
        """pairs: [(a, b)] where a / b is a vector or a tuple (u, v) standing for u - v; slots slot0.. get the dot
        products, and with residual=(seg, x, g, c, gd, slot) that slot gets the Eq. 25 residual^2 -- one launch."""
        k = len(pairs)
        tabs = [(ctypes.c_void_p * k)() for _ in range(4)]
        n = None
        for j, (a, b) in enumerate(pairs):
            for base, opnd in ((0, a), (2, b)):
                if isinstance(opnd, tuple):
                    tabs[base][j], tabs[base + 1][j] = opnd[0].data_ptr(), opnd[1].data_ptr()
                    n = opnd[0].numel()
                else:
                    tabs[base][j], tabs[base + 1][j] = opnd.data_ptr(), None
                    n = opnd.numel()
        if residual is None:
            rc = self.lib.ccqp_fused_dots_f64(self.ctx, k, tabs[0], tabs[1], tabs[2], tabs[3], n, slot0, None, None,
                                              None, 0.0, 0.0, 0, self.stream)
        else:
            seg, rx, rg, c, gd, rslot = residual
            rc = self.lib.ccqp_fused_dots_f64(self.ctx, k, tabs[0], tabs[1], tabs[2], tabs[3], n, slot0, seg,
                                              rx.data_ptr(), rg.data_ptr(), c, gd, rslot, self.stream)
        if rc:
            _lib.check(rc)

    def bb_tail_elementwise(self, seg, x_new, x_old, g_new, g_old, c, gd, slot0):
        """slots [slot0, slot0+3) = residual^2(x_new, g_new), dx.dx, dx.dg in one launch."""
        rc = self.lib.ccqp_bb_tail_elementwise_f64(self.ctx, seg, x_new.data_ptr(), x_old.data_ptr(),
                                                   g_new.data_ptr(), g_old.data_ptr(), c, gd, slot0, self.stream)
        if rc:
            _lib.check(rc)

    # ------------------------------------------------------------------ MPRGP helpers
    def isclose_mask_mul(self, a, b, g, out, invert=False):
        rc = self.lib.ccqp_isclose_mask_mul_f64(self.ctx, a.data_ptr(), b.data_ptr(),
                                                None if g is None else g.data_ptr(), 1 if invert else 0,
                                                out.data_ptr(), out.numel(), self.stream)
        if rc:
            _lib.check(rc)

    def count_notclose(self, a, b, slot):
        rc = self.lib.ccqp_count_notclose_f64(self.ctx, a.data_ptr(), b.data_ptr(), a.numel(), slot, self.stream)
        if rc:
            _lib.check(rc)

    def feasible_steps_elementwise(self, seg, x, p, alpha0, K, slot0):
        rc = self.lib.ccqp_feasible_steps_elementwise_f64(self.ctx, seg, x.data_ptr(), p.data_ptr(), alpha0, K,
                                                          slot0, self.stream)
        if rc:
            _lib.check(rc)

    def normal_elementwise(self, seg, xp, out, dist2_slot):
        rc = self.lib.ccqp_normal_elementwise_f64(self.ctx, seg, xp.data_ptr(), out.data_ptr(), dist2_slot,
                                                  self.stream)
        if rc:
            _lib.check(rc)

    def normal_sphere(self, seg, xp, out, dist2_slot, len2_slot):
        rc = self.lib.ccqp_normal_sphere_f64(self.ctx, seg, xp.data_ptr(), out.data_ptr(), dist2_slot, len2_slot,
                                             self.stream)
        if rc:
            _lib.check(rc)

    def chopped_gradient(self, w, g, nv, ng_slot, out):
        rc = self.lib.ccqp_chopped_gradient_f64(self.ctx, w.data_ptr(), g.data_ptr(), nv.data_ptr(), ng_slot,
                                                out.data_ptr(), out.numel(), self.stream)
        if rc:
            _lib.check(rc)

    def projected_gradient_elementwise(self, seg, x, g, nv, free_g, chopped_g):
        rc = self.lib.ccqp_projected_gradient_elementwise_f64(self.ctx, seg, x.data_ptr(), g.data_ptr(),
                                                              nv.data_ptr(), free_g.data_ptr(),
                                                              chopped_g.data_ptr(), self.stream)
        if rc:
            _lib.check(rc)

    # fused MPRGP steps (elementwise feasible sets)
    def mprgp_split(self, seg, x, g, xp, psi, w, dist2_slot):
        rc = self.lib.ccqp_mprgp_split_f64(self.ctx, seg, x.data_ptr(), g.data_ptr(), xp.data_ptr(),
                                           psi.data_ptr(), w.data_ptr(), dist2_slot, self.stream)
        if rc:
            _lib.check(rc)

    def mprgp_normal_dot(self, seg, xp, g, nv, dist2_slot, ng_slot):
        rc = self.lib.ccqp_mprgp_normal_dot_f64(self.ctx, seg, xp.data_ptr(), g.data_ptr(), nv.data_ptr(),
                                                dist2_slot, ng_slot, self.stream)
        if rc:
            _lib.check(rc)

    def mprgp_proportion(self, w, g, nv, psi, ng_slot, slot0):
        rc = self.lib.ccqp_mprgp_proportion_f64(self.ctx, w.data_ptr(), g.data_ptr(), nv.data_ptr(),
                                                psi.data_ptr(), ng_slot, w.numel(), slot0, self.stream)
        if rc:
            _lib.check(rc)

    def free_gradient_elementwise(self, seg, x, g, out):
        rc = self.lib.ccqp_free_gradient_elementwise_f64(self.ctx, seg, x.data_ptr(), g.data_ptr(),
                                                         out.data_ptr(), self.stream)
        if rc:
            _lib.check(rc)

    def cg_direction(self, psi, Ap, pAp, p, out):
        rc = self.lib.ccqp_cg_direction_f64(self.ctx, psi.data_ptr(), Ap.data_ptr(), pAp, p.data_ptr(),
                                            out.data_ptr(), out.numel(), self.stream)
        if rc:
            _lib.check(rc)

    # ------------------------------------------------------------------ generator
    def gen_spd_rows(self, A, row0, n, seed):
        rows = A.shape[0]
        _lib.check(self.lib.ccqp_gen_spd_rows_f64(self.ctx, A.data_ptr(), A.stride(0), row0, rows, n,
                                                  seed & 0xFFFFFFFFFFFFFFFF, self.stream))

    def gen_vec(self, out, seed):
        _lib.check(self.lib.ccqp_gen_vec_f64(self.ctx, out.data_ptr(), out.numel(), seed & 0xFFFFFFFFFFFFFFFF,
                                             self.stream))
