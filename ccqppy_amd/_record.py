"""Recording engine: an Engine whose launches are DEFERRED so that the batch scheduler can merge the
same operation of many in-flight solves into one batched launch (ccqp_batch_run_f64).

A solve runs against a RecordingEngine exactly as against a normal Engine; the operations that have
a batched form are appended to `queue` instead of being launched.  Anything else (non-elementwise
projections, generic MPRGP helpers, blocking reads) first flushes this lane's own queue in order and
then launches immediately, so per-lane program order is always preserved.
"""
import ctypes

from . import _lib
from ._engine import Engine


class RecordingEngine(Engine):
    def __init__(self, index, stream):
        super().__init__(index, stream)
        slab, part = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(self.lib.ccqp_ctx_workspace(self.ctx, ctypes.byref(slab), ctypes.byref(part)))
        self.slab_ptr = slab.value
        self.partials_ptr = part.value
        self.queue = []      # (opcode, signature, ptrs, scalars, n, n2, lda, slots, k, seg)
        self.keep = []       # tensors referenced by queued ops (kept alive until the flush is enqueued)

    # ------------------------------------------------------------------ recording helpers
    def _rec(self, opcode, sig, tensors, scalars=(), n=0, n2=0, lda=0, slots=(), k=0, seg=None):
        ptrs = []
        keep = self.keep
        for t in tensors:
            if t is None:
                ptrs.append(None)
            else:
                ptrs.append(t.data_ptr())
                keep.append(t)
        if seg is not None:
            keep.append(seg)
        self.queue.append((opcode, sig, ptrs, scalars, n, n2, lda, slots, k, seg))

    def write_bop(self, bop, op):
        """Write queued op `op` of this lane into the descriptor `bop`."""
        opcode, sig, ptrs, scalars, n, n2, lda, slots, k, seg = op
        p = bop.p
        for i, v in enumerate(ptrs):
            p[i] = v
        s = bop.s
        for i, v in enumerate(scalars):
            s[i] = v
        bop.n = n
        bop.n2 = n2
        bop.lda = lda
        bop.slab = self.slab_ptr
        bop.partials = self.partials_ptr
        sl = bop.slot
        for i, v in enumerate(slots):
            sl[i] = v
        bop.k = k
        if seg is not None:
            bop.seg = seg

    def take(self):
        q, self.queue = self.queue, []
        return q

    def release(self):
        self.keep = []

    def flush_self(self):
        """Launch this lane's queued operations now, one group per operation, in order."""
        q = self.take()
        if not q:
            return
        n = len(q)
        ops = (_lib.Bop * n)()
        codes = (ctypes.c_int * n)()
        counts = (ctypes.c_int * n)()
        for i, op in enumerate(q):
            self.write_bop(ops[i], op)
            codes[i] = op[0]
            counts[i] = 1
        _lib.check(self.lib.ccqp_batch_run_f64(self.ctx, n, codes, counts, ops, self.stream))
        self.release()

    # ------------------------------------------------------------------ deferred operations
    def gemv(self, A, x, y, b=None, g=None, variant=None):
        rows, cols = A.shape
        assert A.stride(1) == 1 and x.numel() >= cols and y.numel() >= rows
        self.mv_executed += 1
        lda = A.stride(0)
        aligned = ((A.data_ptr() | x.data_ptr()) & 15) == 0 and lda % 2 == 0
        self._rec(_lib.BOP_GEMV, (_lib.BOP_GEMV, rows, cols, lda, aligned), (A, x, b if g is not None else None, y, g),
                  n=rows, n2=cols, lda=lda)

    def axpby(self, a, x, b, y, out):
        n = out.numel()
        self._rec(_lib.BOP_AXPBY, (_lib.BOP_AXPBY, n), (x, y, out), (a, b), n=n)

    def axpby2(self, a1, x1, b1, y1, out1, a2, x2, b2, y2, out2):
        self.axpby(a1, x1, b1, y1, out1)
        self.axpby(a2, x2, b2, y2, out2)

    def proj_elementwise_sub(self, seg, x, out, a, b, y, z):
        t = self.empty(out.numel())
        self.proj_elementwise(seg, x, t, a, b, y)
        self.axpby(1.0, t, -1.0, z, out)

    def scale(self, a, x, out):
        n = out.numel()
        self._rec(_lib.BOP_SCALE, (_lib.BOP_SCALE, n), (x, out), (a,), n=n)

    def copy(self, x, out):
        n = out.numel()
        self._rec(_lib.BOP_COPY, (_lib.BOP_COPY, n), (x, out), n=n)

    def fill_(self, value, out):
        n = out.numel()
        self._rec(_lib.BOP_FILL, (_lib.BOP_FILL, n), (out,), (value,), n=n)

    def cg_direction(self, psi, Ap, pAp, p, out):
        n = out.numel()
        self._rec(_lib.BOP_CG_DIRECTION, (_lib.BOP_CG_DIRECTION, n), (psi, Ap, p, out), (pAp,), n=n)

    def dots(self, pairs, slot0=0):
        k = len(pairs)
        n = pairs[0][0].numel()
        flat = []
        for a, b in pairs:
            flat.append(a)
            flat.append(b)
        self._rec(_lib.BOP_DOTS, (_lib.BOP_DOTS, n, k), flat, n=n, slots=(slot0,), k=k)

    def proj_elementwise(self, seg, x, out, a=1.0, b=0.0, y=None):
        self._rec(_lib.BOP_PROJ_ELEM, (_lib.BOP_PROJ_ELEM, seg.dim), (x, y, out), (a, b), seg=seg)

    def residual_sq_elementwise(self, seg, x, g, c, gd, slot):
        self._rec(_lib.BOP_RESIDUAL_ELEM, (_lib.BOP_RESIDUAL_ELEM, seg.dim), (x, g), (c, gd), slots=(slot,), k=1,
                  seg=seg)

    def fused_dots(self, pairs, slot0=0, residual=None):
        # batched form: materialise the differences, then the plain batched dots / residual (same bits)
        flat = []
        for a, b in pairs:
            ops = []
            for opnd in (a, b):
                if isinstance(opnd, tuple):
                    d = self.empty(opnd[0].numel())
                    self.axpby(1.0, opnd[0], -1.0, opnd[1], d)
                    opnd = d
                ops.append(opnd)
            flat.append(tuple(ops))
        self.dots(flat, slot0)
        if residual is not None:
            seg, rx, rg, c, gd, rslot = residual
            self.residual_sq_elementwise(seg, rx, rg, c, gd, rslot)

    def bb_tail_elementwise(self, seg, x_new, x_old, g_new, g_old, c, gd, slot0):
        # the batched form keeps the four operations the fused kernel stands for (bitwise the same results)
        self.residual_sq_elementwise(seg, x_new, g_new, c, gd, slot0)
        dx, dg = self.empty(x_new.numel()), self.empty(x_new.numel())
        self.axpby(1.0, x_new, -1.0, x_old, dx)
        self.axpby(1.0, g_new, -1.0, g_old, dg)
        self.dots([(dx, dx), (dx, dg)], slot0 + 1)

    def feasible_steps_elementwise(self, seg, x, p, alpha0, K, slot0):
        self._rec(_lib.BOP_FEASIBLE_ELEM, (_lib.BOP_FEASIBLE_ELEM, seg.dim, K), (x, p), (alpha0,), slots=(slot0,),
                  k=K, seg=seg)

    def mprgp_split(self, seg, x, g, xp, psi, w, dist2_slot):
        self._rec(_lib.BOP_MPRGP_SPLIT, (_lib.BOP_MPRGP_SPLIT, seg.dim), (x, g, xp, psi, w), slots=(dist2_slot,),
                  k=1, seg=seg)

    def mprgp_normal_dot(self, seg, xp, g, nv, dist2_slot, ng_slot):
        self._rec(_lib.BOP_MPRGP_NORMAL_DOT, (_lib.BOP_MPRGP_NORMAL_DOT, seg.dim), (xp, g, nv),
                  slots=(ng_slot, dist2_slot), k=1, seg=seg)

    def mprgp_proportion(self, w, g, nv, psi, ng_slot, slot0):
        n = w.numel()
        self._rec(_lib.BOP_MPRGP_PROPORTION, (_lib.BOP_MPRGP_PROPORTION, n), (w, g, nv, psi), n=n,
                  slots=(slot0, ng_slot), k=2)

    def free_gradient_elementwise(self, seg, x, g, out):
        self._rec(_lib.BOP_FREE_GRADIENT_ELEM, (_lib.BOP_FREE_GRADIENT_ELEM, seg.dim), (x, g, out), seg=seg)

    # ------------------------------------------------------------------ everything else: in order, now
    def read(self, slot0, count):
        self.flush_self()
        return super().read(slot0, count)

    def read_async(self, slot0, count):
        # the scheduler flushes merged groups before it calls this; a lone caller must flush itself
        if self.queue:
            self.flush_self()
        return super().read_async(slot0, count)


def _immediate(name):
    base = getattr(Engine, name)

    def method(self, *args, **kwargs):
        self.flush_self()
        return base(self, *args, **kwargs)
    method.__name__ = name
    return method


for _name in ("mul", "muldiv", "submul", "proj_normscale", "proj_small", "isclose_mask_mul", "count_notclose",
              "normal_elementwise", "normal_sphere", "chopped_gradient", "projected_gradient_elementwise",
              "gen_spd_rows", "gen_vec", "spmv_csr", "spmv_plan", "csr_violations", "segs_project", "segs_residual_sq",
              "segs_normal", "segs_feasible_steps"):
    setattr(RecordingEngine, _name, _immediate(_name))

# Engine.fill(value, out) is the public name; keep it, recorded
RecordingEngine.fill = RecordingEngine.fill_
