"""Per-device engine: thin, typed Python front of the C ABI (include/ccqp_hip.h).

PyTorch is plumbing only: device memory (torch tensors), the current HIP stream and, for the
row-sharded operator, torch.distributed (RCCL).  All arithmetic runs in libccqp_hip.so.
Scalars that steer the solvers' control flow are read back into numpy float64 scalars so the
host logic uses exactly the arithmetic types of the reference's Python loops.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_engines = {}


def _require_gpu():
    if not torch.cuda.is_available():
        raise _lib.CCQPHipUnavailable(
            "ccqppy_amd: no ROCm GPU visible to PyTorch.  The CCQP hot path only runs on the HIP "
            "engine (gfx950); there is no CPU fallback.")


def get_engine(device=None):
    """Engine for a torch device / index (default: the current device)."""
    _require_gpu()
    if device is None:
        idx = torch.cuda.current_device()
    elif isinstance(device, int):
        idx = device
    else:
        dev = torch.device(device)
        idx = torch.cuda.current_device() if dev.index is None else dev.index
    eng = _engines.get(idx)
    if eng is None:
        eng = Engine(idx)
        _engines[idx] = eng
    return eng


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class Engine:
    def __init__(self, index):
        self.lib = _lib.load()
        self.index = index
        self.device = torch.device("cuda", index)
        h = ctypes.c_void_p()
        with torch.cuda.device(index):
            _lib.check(self.lib.ccqp_ctx_create(index, ctypes.byref(h)))
        self.ctx = h
        dev = ctypes.c_int()
        ncu = ctypes.c_int()
        buf = ctypes.create_string_buffer(64)
        _lib.check(self.lib.ccqp_ctx_info(self.ctx, ctypes.byref(dev), ctypes.byref(ncu), buf, 64))
        self.num_cu = ncu.value
        self.arch = buf.value.decode()
        self._host = (ctypes.c_double * _lib.NUM_SLOTS)()
        self._scratch = {}
        self.gemv_variant = 0
        self.mv_executed = 0   # every matvec launched through this engine (counted or not)
        self.gemv_events = None   # list of (start, end) HIP events per matvec when profiling is on

    # ------------------------------------------------------------------ memory plumbing
    @property
    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

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
        """Blocking read of `count` slab slots -> list of np.float64 (the only sync point)."""
        _lib.check(self.lib.ccqp_scalars_read(self.ctx, slot0, count, self._host, self.stream))
        return [np.float64(self._host[i]) for i in range(count)]

    # ------------------------------------------------------------------ a1 matvec
    def gemv(self, A, x, y, b=None, g=None, variant=None):
        """y = A.x for a (rows, cols) row-major device matrix; optionally g = y + b."""
        rows, cols = A.shape
        assert A.stride(1) == 1 and x.numel() >= cols and y.numel() >= rows
        v = self.gemv_variant if variant is None else variant
        self.mv_executed += 1
        ev = None
        if self.gemv_events is not None:
            # HIP events on the stream the kernel is launched on (torch's current stream)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(self.device))
        if g is None:
            _lib.check(self.lib.ccqp_gemv_f64(self.ctx, ptr(A), A.stride(0), ptr(x), ptr(y), rows, cols, v,
                                              self.stream))
        else:
            _lib.check(self.lib.ccqp_gemv_epilogue_f64(self.ctx, ptr(A), A.stride(0), ptr(x), ptr(b), ptr(y),
                                                       ptr(g), rows, cols, v, self.stream))
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(self.device))
            self.gemv_events.append(ev)

    # ------------------------------------------------------------------ a2 elementwise
    def axpby(self, a, x, b, y, out):
        _lib.check(self.lib.ccqp_axpby_f64(self.ctx, float(a), ptr(x), float(b), ptr(y), ptr(out),
                                           out.numel(), self.stream))

    def scale(self, a, x, out):
        _lib.check(self.lib.ccqp_scale_f64(self.ctx, float(a), ptr(x), ptr(out), out.numel(), self.stream))

    def copy(self, x, out):
        _lib.check(self.lib.ccqp_copy_f64(self.ctx, ptr(x), ptr(out), out.numel(), self.stream))

    def fill(self, value, out):
        _lib.check(self.lib.ccqp_fill_f64(self.ctx, float(value), ptr(out), out.numel(), self.stream))

    def mul(self, x, y, out):
        _lib.check(self.lib.ccqp_mul_f64(self.ctx, ptr(x), ptr(y), ptr(out), out.numel(), self.stream))

    def muldiv(self, x, y, s, out):
        _lib.check(self.lib.ccqp_muldiv_f64(self.ctx, ptr(x), ptr(y), float(s), ptr(out), out.numel(),
                                            self.stream))

    def submul(self, x, y, z, out):
        _lib.check(self.lib.ccqp_submul_f64(self.ctx, ptr(x), ptr(y), ptr(z), ptr(out), out.numel(),
                                            self.stream))

    # ------------------------------------------------------------------ a3 reductions
    def dots(self, pairs, slot0=0):
        """slab[slot0 + j] = pairs[j][0] . pairs[j][1]   (asynchronous)."""
        k = len(pairs)
        n = pairs[0][0].numel()
        xs = (ctypes.c_void_p * k)(*[p[0].data_ptr() for p in pairs])
        ys = (ctypes.c_void_p * k)(*[p[1].data_ptr() for p in pairs])
        _lib.check(self.lib.ccqp_dots_f64(self.ctx, k, xs, ys, n, slot0, self.stream))

    def dots_now(self, pairs, slot0=0):
        self.dots(pairs, slot0)
        return self.read(slot0, len(pairs))

    # ------------------------------------------------------------------ projections
    def proj_elementwise(self, seg, x, out, a=1.0, b=0.0, y=None):
        _lib.check(self.lib.ccqp_proj_elementwise_f64(self.ctx, ctypes.byref(seg), float(a), ptr(x), float(b),
                                                      ptr(y), ptr(out), self.stream))

    def proj_normscale(self, seg, x, out, slot):
        _lib.check(self.lib.ccqp_proj_normscale_f64(self.ctx, ctypes.byref(seg), ptr(x), ptr(out), slot,
                                                    self.stream))

    def proj_small(self, table, nseg, x, out, mode):
        _lib.check(self.lib.ccqp_proj_small_segments_f64(self.ctx, ptr(table), nseg, ptr(x), ptr(out), mode,
                                                         self.stream))

    def residual_sq_elementwise(self, seg, x, g, c, gd, slot):
        _lib.check(self.lib.ccqp_residual_sq_elementwise_f64(self.ctx, ctypes.byref(seg), ptr(x), ptr(g),
                                                             float(c), float(gd), slot, self.stream))

    # ------------------------------------------------------------------ MPRGP helpers
    def isclose_mask_mul(self, a, b, g, out, invert=False):
        _lib.check(self.lib.ccqp_isclose_mask_mul_f64(self.ctx, ptr(a), ptr(b), ptr(g), 1 if invert else 0,
                                                      ptr(out), out.numel(), self.stream))

    def count_notclose(self, a, b, slot):
        _lib.check(self.lib.ccqp_count_notclose_f64(self.ctx, ptr(a), ptr(b), a.numel(), slot, self.stream))

    def normal_elementwise(self, seg, xp, out, dist2_slot):
        _lib.check(self.lib.ccqp_normal_elementwise_f64(self.ctx, ctypes.byref(seg), ptr(xp), ptr(out),
                                                        dist2_slot, self.stream))

    def normal_sphere(self, seg, xp, out, dist2_slot, len2_slot):
        _lib.check(self.lib.ccqp_normal_sphere_f64(self.ctx, ctypes.byref(seg), ptr(xp), ptr(out), dist2_slot,
                                                   len2_slot, self.stream))

    def chopped_gradient(self, w, g, nv, ng_slot, out):
        _lib.check(self.lib.ccqp_chopped_gradient_f64(self.ctx, ptr(w), ptr(g), ptr(nv), ng_slot, ptr(out),
                                                      out.numel(), self.stream))

    # ------------------------------------------------------------------ generator
    def gen_spd_rows(self, A, row0, n, seed):
        rows = A.shape[0]
        _lib.check(self.lib.ccqp_gen_spd_rows_f64(self.ctx, ptr(A), A.stride(0), row0, rows, n,
                                                  ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), self.stream))

    def gen_vec(self, out, seed):
        _lib.check(self.lib.ccqp_gen_vec_f64(self.ctx, ptr(out), out.numel(),
                                             ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), self.stream))
