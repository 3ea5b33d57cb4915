"""The `A` of a solve as a device operator.

The reference only ever calls `A.dot(v)` (solvers.py:25,133,...).  Here that call is one of
  * DenseOperator        -- the whole matrix on one GPU, `ccqp_gemv_f64`;
  * SparseOperator       -- a CSR matrix on one GPU (scipy.sparse / torch sparse input), `ccqp_spmv_csr_f64`;
  * ComposedOperator     -- matrix-free product of dense / sparse / diagonal factors (+ shift * I), e.g. the
                            Delassus operator D M^-1 D^T of a contact problem;
  * RowShardedOperator   -- one process per GPU (torch.distributed, backend "nccl" = RCCL over
                            xGMI): rank r owns the contiguous row block [r*n/G, (r+1)*n/G), all
                            n-vectors are replicated, every rank runs the same host loop.  One
                            matvec = local GEMV on the block -> ONE exchange of the n-vector
                            (all-gather by default; zero-padded all-reduce on request).
Because every row sum is produced by exactly one wave in an order that depends on n only
(include/ccqp_hip.h), A.x is bitwise identical for every GPU count, hence so are all scalars
and therefore the control flow and MV counts on every rank.
"""
import torch

from ._engine import get_engine
from ._lib import check as _check


def row_partition(n, world, rank):
    """Contiguous row block of `rank`: (row0, rows, rows_per) with rows_per = ceil(n/world)."""
    per = -(-n // world)
    r0 = min(rank * per, n)
    r1 = min(r0 + per, n)
    return r0, r1 - r0, per


def batch_partition(nproblems, world, rank):
    """Problems of the batch-sharded mode handled by `rank`: p with p % world == rank."""
    return list(range(rank, nproblems, world))


class DenseOperator:
    def __init__(self, A, engine=None):
        if isinstance(A, torch.Tensor) and A.is_cuda:
            self.engine = engine or get_engine(A.device)
        else:
            self.engine = engine or get_engine()
        self.A = self.engine.to_device(A)
        assert self.A.dim() == 2
        self.n = self.A.shape[1]
        self.rows = self.A.shape[0]

    def matvec(self, x, y, b=None, g=None, engine=None):
        """y = A.x (and g = y + b); `engine` = the lane to launch on (default: the operator's)."""
        (engine or self.engine).gemv(self.A, x, y, b, g)

    def apply(self, eng, x, y, b=None, g=None):
        eng.gemv(self.A, x, y, b, g)

    def matvec2(self, x1, y1, g1, x2, y2, g2, b=None, engine=None):
        """y1 = A.x1, y2 = A.x2 (g_i = y_i + b where g_i is given) -- ONE pass over A where the LDS-staged form applies
        (Engine.gemv2), each product with the bits of its own matvec()."""
        (engine or self.engine).gemv2(self.A, x1, x2, y1, y2, b, g1, g2)

    def matvec3(self, xs, ys, gs, b=None, engine=None):
        """ys[j] = A.xs[j] (gs[j] = ys[j] + b where given) for the j with ys[j] not None: up to three products in ONE pass
        over a square A where the LDS-staged form applies (Engine.gemv3), each with the bits of its own matvec()."""
        (engine or self.engine).gemv3(self.A, xs, ys, b, gs)

    def _rows_aligned(self):
        """What the form queries assume and the launches check: 16-byte aligned rows (the solvers' vectors are fresh
        allocations, always aligned).  A view with an odd leading dimension or an 8-byte offset runs its pairs / triples
        as single passes -- then speculating on a product costs a whole pass, so the solvers must not."""
        return self.A.data_ptr() % 16 == 0 and self.A.stride(0) % 2 == 0

    def triple_in_one_pass(self, engine=None):
        eng = engine or self.engine
        return self.rows == self.n and self._rows_aligned() and hasattr(eng, "gemv3_form") and \
            eng.gemv3_form(self.n) == "bp_gemv3_pers_kernel"

    def pair_in_one_pass(self, engine=None):
        """True if matvec2 reads the matrix once (what makes speculating on the second product worth it)."""
        eng = engine or self.engine                     # (a recording engine of the fused batch scheduler has no pair form)
        return self._rows_aligned() and hasattr(eng, "gemv2_form") and eng.gemv2_form(self.rows, self.n) == "gemv_lds2_kernel"

    def pipe_apply(self, eng, x, y, b=None, g=None):
        """apply() behind the pipeline's "converged" word (solvers.CCQPSolverBBPGD(pipeline=True))."""
        A = self.A
        if g is None:
            raise NotImplementedError("the pipelined dense matvec is the whole operator (it needs b and g)")
        _check(eng.lib.ccqp_pipe_gemv_epilogue_f64(eng.ctx, A.data_ptr(), A.stride(0), x.data_ptr(), b.data_ptr(),
                                                   y.data_ptr(), g.data_ptr(), A.shape[0], A.shape[1],
                                                   eng.gemv_variant, eng.stream))


class VectorExchange:
    """Reassembles the replicated n-vector from per-rank row slabs.  Backend-agnostic on purpose:
    the same code runs over RCCL (device tensors) and, in the CPU test-suite, over gloo.

    Per matvec the device side issues ONE copy of the own slab into a persistent staging buffer (n/G doubles;
    send and receive buffers then never alias, whatever the backend's rules about in-place all-gather are) and
    ONE collective; `events`, when set to a list, receives a (start, end) pair of HIP events per call, recorded
    on the stream the engine launches on (bench.py: exchange time per matvec)."""

    def __init__(self, n, group=None, mode="allgather"):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n = n
        self.row0, self.rows, self.per = row_partition(n, self.world, self.rank)
        assert mode in ("allgather", "allreduce", "direct")
        # "direct": the all-gather is issued by RCCL itself through the C ABI (ccqp_comm_allgather_f64) on the engine's
        # own stream, in place -- no staging copy, no second stream, no event hand-off; opt-in (default: torch's binding)
        self.direct = mode == "direct"
        self._comm = None
        self.mode = "allgather" if self.direct else mode
        self.even = (n % self.world) == 0
        self._pad = None
        self._send = None
        backend = dist.get_backend(group)
        self.backend = backend
        self.stage_cpu = backend != "nccl"   # gloo: stage device tensors through the host (tests only)
        if self.direct and backend != "nccl":
            raise ValueError("VectorExchange(mode='direct') needs the nccl (RCCL) backend, got %r" % backend)
        self.events = None
        self.calls = 0
        self._many = {}                      # k -> (send, gathered) buffers of exchange_many

    def _run(self, fn, t):
        """Run collective `fn` on tensor t, staging through the host when the backend cannot
        take device tensors (gloo; test path only)."""
        if self.stage_cpu and t.is_cuda:
            host = t.cpu()
            fn(host)
            t.copy_(host)
        else:
            fn(t)

    def _staged(self, mine, eng):
        """The own slab in the persistent send buffer (engine copy kernel; plain torch for CPU tensors)."""
        if eng is None or not mine.is_cuda:
            return mine.clone()
        if self._send is None or self._send.numel() != mine.numel() or self._send.device != mine.device:
            self._send = torch.empty(mine.numel(), dtype=mine.dtype, device=mine.device)
        eng.copy(mine, self._send)
        return self._send

    def _gather_into(self, full, mine, eng):
        src = self._staged(mine, eng) if mine.device == full.device else mine.clone()
        if self.backend == "nccl":
            # RCCL: an error here is an error of the run -- no second attempt through another call, no other backend
            self.dist.all_gather_into_tensor(full, src, group=self.group)
            return
        try:
            self.dist.all_gather_into_tensor(full, src, group=self.group)
        except (RuntimeError, NotImplementedError):      # a CPU backend without the flat form (tests only)
            self.dist.all_gather(list(full.chunk(self.world)), src, group=self.group)

    def __call__(self, y, eng=None):
        """y holds this rank's slab at [row0, row0+rows); on return every rank holds all of y."""
        if self.world == 1:
            return
        ev = None
        st = None
        if y.is_cuda:
            st = eng.torch_stream if eng is not None else torch.cuda.current_stream(y.device)
        if self.events is not None and st is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(st)
        if st is not None and eng is not None and not self.direct:
            # torch.distributed orders a collective against torch's CURRENT stream only: make that the stream the
            # engine launched the staging copy and the matvec on (a lane engine, or a caller inside another
            # `with torch.cuda.stream(...)` who has not re-bound), so copy, matvec and collective share one order
            with torch.cuda.stream(st):
                self._exchange(y, eng)
        else:
            self._exchange(y, eng)
        self.calls += 1
        if ev is not None:
            ev[1].record(st)
            self.events.append(ev)

    # ---- k products, ONE exchange (APGD's pair, MPRGP's look-ahead triple on a row-sharded matrix) -----------------
    def slabs(self, k, like):
        """k views of `rows` doubles each for this rank's slabs of k products: the local k-vector matvec writes them and
        exchange_many ships them as they lie -- no staging copy.  Layout of the gathered buffer: rank r's slab of
        product j at [(r k + j) per, + rows_r); all-gather: the views are a separate send buffer of k per doubles (send
        and receive never alias); direct / all-reduce: they lie inside the gathered buffer at this rank's offset (RCCL's
        in-place all-gather; the zero-padded sum)."""
        key = (k, like.device)
        bufs = self._many.get(key)
        if bufs is None:
            gathered = torch.zeros(self.world * k * self.per, dtype=torch.float64, device=like.device)
            inplace = self.direct or self.mode == "allreduce"
            lo = self.rank * k * self.per
            send = gathered[lo:lo + k * self.per] if inplace else torch.zeros(k * self.per, dtype=torch.float64,
                                                                               device=like.device)
            bufs = self._many[key] = (send, gathered)
        send = bufs[0]
        return [send[j * self.per:j * self.per + self.rows] for j in range(k)]

    def exchange_many(self, ys, b=None, gs=None, eng=None):
        """After the local matvec has filled slabs(k, ...): ONE collective of k per doubles per rank, then ONE unpack
        launch that scatters the slabs into the replicated vectors ys[j] and forms gs[j] = ys[j] + b where given (the same
        single add as the fused epilogue: the same bits).  k = len(ys) <= 3."""
        k = len(ys)
        gs = [None] * k if gs is None else gs
        send, gathered = self._many[(k, ys[0].device)]
        dev = ys[0].is_cuda
        ev = st = None
        if dev:
            st = eng.torch_stream if eng is not None else torch.cuda.current_stream(ys[0].device)
        if self.events is not None and st is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(st)
        if self.world > 1:
            if st is not None and eng is not None and not self.direct:
                with torch.cuda.stream(st):              # (one ordering domain with the matvec: see __call__)
                    self._collective_many(k, send, gathered, eng)
            else:
                self._collective_many(k, send, gathered, eng)
        elif send.data_ptr() != gathered.data_ptr():
            gathered.copy_(send)
        if dev and eng is not None:
            eng.unpack_gathered(gathered, k, self.per, ys, b, gs)
        else:                                            # CPU tensors (the gloo tests): the same index map, the same add
            view = gathered.view(self.world, k, self.per)
            for j in range(k):
                ys[j].copy_(view[:, j, :].reshape(-1)[:self.n])
                if gs[j] is not None:
                    torch.add(ys[j], b, out=gs[j])
        self.calls += 1
        if ev is not None:
            ev[1].record(st)
            self.events.append(ev)

    def _collective_many(self, k, send, gathered, eng):
        dist, cnt = self.dist, k * self.per
        if self.direct:
            from . import _lib
            _lib.check(eng.lib.ccqp_comm_allgather_f64(self._direct_comm(eng), gathered.data_ptr(), cnt, eng.stream))
            return
        if self.mode == "allreduce":
            lo = self.rank * cnt
            for a, z in ((0, lo), (lo + cnt, self.world * cnt)):
                if z > a:
                    if eng is not None and gathered.is_cuda:
                        eng.fill(0.0, gathered[a:z])
                    else:
                        gathered[a:z].zero_()
            self._run(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group), gathered)
            return
        if self.stage_cpu and gathered.is_cuda:          # gloo with device tensors (tests only): through the host
            hs, hg = send.cpu(), torch.empty(gathered.numel(), dtype=gathered.dtype)
            self._gather_flat(hg, hs)
            gathered.copy_(hg)
        else:
            self._gather_flat(gathered, send)

    def _gather_flat(self, full, src):
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(full, src, group=self.group)      # (an RCCL error is an error of the run)
            return
        try:
            self.dist.all_gather_into_tensor(full, src, group=self.group)
        except (RuntimeError, NotImplementedError):      # a CPU backend without the flat form (tests only)
            self.dist.all_gather(list(full.chunk(self.world)), src, group=self.group)

    def open(self, eng):
        """Create the C-ABI RCCL communicator of mode="direct" now (collective: every rank of the group must call it;
        a barrier of the process group follows, so no rank runs ahead into a matvec while another is still inside
        ncclCommInitRank).  RowShardedOperator calls this at construction.  While this exchange is in use no
        torch.distributed collective of the same ranks may be IN FLIGHT on another stream (two communicators with
        concurrent collectives is the known RCCL/NCCL deadlock pattern): the solvers issue none."""
        if self.direct and self._comm is None and self.world > 1:
            self._direct_comm(eng)
            self.dist.barrier(group=self.group)

    def close(self):
        """Destroy the C-ABI communicator (idempotent; also run when the exchange is garbage-collected)."""
        comm, self._comm = getattr(self, "_comm", None), None
        if comm is not None:
            try:
                self._comm_lib.ccqp_comm_destroy(comm)
            except Exception:  # noqa: BLE001
                pass

    def __del__(self):
        self.close()

    def _direct_comm(self, eng):
        """The C-ABI communicator of this exchange (created once: rank 0's unique id travels over the process group)."""
        if self._comm is None:
            import ctypes
            from . import _lib
            lib = eng.lib
            idbuf = (ctypes.c_ubyte * 128)()
            if self.rank == 0:
                _lib.check(lib.ccqp_comm_unique_id(idbuf))
            box = [bytes(idbuf)]
            src = 0 if self.group is None else self.dist.get_global_rank(self.group, 0)
            self.dist.broadcast_object_list(box, src=src, group=self.group)
            idbuf = (ctypes.c_ubyte * 128).from_buffer_copy(box[0])
            comm = ctypes.c_void_p()
            _lib.check(lib.ccqp_comm_create(eng.ctx, self.world, self.rank, idbuf, ctypes.byref(comm)))
            self._comm = comm
            self._comm_lib = lib
        return self._comm

    def _exchange_direct(self, y, eng):
        from . import _lib
        comm = self._direct_comm(eng)
        if self.even:
            _lib.check(eng.lib.ccqp_comm_allgather_f64(comm, y.data_ptr(), self.rows, eng.stream))
            return
        if self._pad is None or self._pad.device != y.device:
            self._pad = torch.zeros(self.per * self.world, dtype=y.dtype, device=y.device)
        full, lo = self._pad, self.rank * self.per
        if self.rows:
            eng.copy(y[self.row0:self.row0 + self.rows], full[lo:lo + self.rows])
        _lib.check(eng.lib.ccqp_comm_allgather_f64(comm, full.data_ptr(), self.per, eng.stream))
        eng.copy(full[:self.n], y)

    def _exchange(self, y, eng):
        dist = self.dist
        if self.direct:
            if eng is None or not y.is_cuda:
                raise ValueError("VectorExchange(mode='direct') works on device vectors through an engine")
            return self._exchange_direct(y, eng)
        if self.mode == "allreduce":
            # zero everything but the own slab, then sum: adding zeros is exact
            for lo, hi in ((0, self.row0), (self.row0 + self.rows, self.n)):
                if hi > lo:
                    if eng is not None and y.is_cuda:
                        eng.fill(0.0, y[lo:hi])
                    else:
                        y[lo:hi].zero_()
            self._run(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group), y)
            return
        if self.even:
            self._run(lambda t: self._gather_into(t, t[self.row0:self.row0 + self.rows],
                                                  eng if t.is_cuda else None), y)
            return
        # ragged: gather fixed-size (padded) slabs, then drop the padding
        if self._pad is None or self._pad.device != y.device:
            self._pad = torch.zeros(self.per * self.world, dtype=y.dtype, device=y.device)
        full = self._pad
        lo = self.rank * self.per
        if self.rows:
            if eng is not None and y.is_cuda:
                eng.copy(y[self.row0:self.row0 + self.rows], full[lo:lo + self.rows])
            else:
                full[lo:lo + self.rows].copy_(y[self.row0:self.row0 + self.rows])
        self._run(lambda t: self._gather_into(t, t[lo:lo + self.per], eng if t.is_cuda else None), full)
        # rank r's rows sit at [r*per, r*per + rows_r); only the last non-empty rank is short
        if eng is not None and y.is_cuda:
            eng.copy(full[:self.n], y)
        else:
            y.copy_(full[:self.n])


def _is_sparse_input(A):
    """scipy.sparse matrix/array or torch sparse tensor (the reference accepts anything with `.dot`)."""
    if isinstance(A, torch.Tensor):
        return A.layout != torch.strided
    return hasattr(A, "tocsr") and hasattr(A, "nnz")


class SparseOperator:
    """`A.dot(v)` for a SPARSE matrix held in HBM as CSR (int64 row pointers, int32 columns, fp64
    values): one `ccqp_spmv_csr_f64` launch per matvec, 12 B per stored entry.  Accepts a
    scipy.sparse matrix (any format), a torch sparse tensor (CSR or COO), or the CSR triple
    `(rowptr, col, val)` with `shape=`.  Stored entries are kept exactly as given (explicit zeros and
    the storage order within a row included); the bits of A.x depend on them and on x only
    (include/ccqp_hip.h), so row-sharded and single-GPU runs agree bitwise.
    `check=True` validates the index arrays on the device once (the matvec does not bounds-check)."""

    def __init__(self, A, shape=None, engine=None, check=True, width=0, local_x=True, diagonal_windows=True):
        if isinstance(A, torch.Tensor) and A.is_cuda:
            self.engine = engine or get_engine(A.device)
        elif isinstance(A, (tuple, list)) and isinstance(A[2], torch.Tensor) and A[2].is_cuda:
            self.engine = engine or get_engine(A[2].device)
        else:
            self.engine = engine or get_engine()
        dev = self.engine.device
        if isinstance(A, (tuple, list)):
            if shape is None:
                raise ValueError("SparseOperator: shape=(rows, cols) is required with a (rowptr, col, val) triple")
            rowptr, col, val = A
        elif isinstance(A, torch.Tensor):
            if A.layout == torch.sparse_coo:
                A = A.coalesce().to_sparse_csr()
            if A.layout != torch.sparse_csr:
                A = A.to_sparse_csr()
            shape = tuple(A.shape)
            rowptr, col, val = A.crow_indices(), A.col_indices(), A.values()
        else:
            A = A.tocsr()
            shape = tuple(A.shape)
            rowptr, col, val = A.indptr, A.indices, A.data
        self.rows, self.n = int(shape[0]), int(shape[1])
        if self.n > 2 ** 31 - 1:
            raise ValueError("SparseOperator: more than 2^31-1 columns")

        def dev_tensor(a, dtype):
            if not isinstance(a, torch.Tensor):
                import numpy as np
                a = torch.from_numpy(np.ascontiguousarray(a))
            return a.to(device=dev, dtype=dtype).contiguous()

        self.rowptr = dev_tensor(rowptr, torch.int64)
        self.col = dev_tensor(col, torch.int32)
        self.val = dev_tensor(val, torch.float64)
        if self.rowptr.numel() != self.rows + 1 or self.col.numel() != self.val.numel():
            raise ValueError("SparseOperator: inconsistent CSR array lengths")
        self.nnz = self.val.numel()
        self.width = width
        self.A = self                      # what `op.A` means for a dense operator: the stored matrix
        self.engine.bind()
        if check:
            bad = self.engine.csr_violations(self.rowptr, self.col, self.n)
            if bad:
                raise ValueError("SparseOperator: %d violated CSR invariants (row pointers not monotone / "
                                 "not ending at nnz, or column index outside [0, %d))" % (bad, self.n))
        # one-time analysis: which rows each workgroup streams, which rows are too long for the LDS path
        # (local_x=False keeps every window on the global-gather paths: a tuning / testing switch, same bits)
        # (diagonal_windows=False likewise keeps windows with one repeated column-offset pattern on the generic paths)
        self.local_x = local_x
        self.diagonal_windows = diagonal_windows
        self.plan, self.summary = self.engine.spmv_plan(self.rowptr, self.col, self.nnz, local_x)
        if not diagonal_windows and self.summary[3]:
            _check(self.engine.lib.ccqp_spmv_plan_set(self.engine.ctx, self.plan.data_ptr(), self.nnz, 1, self.summary,
                                                      self.engine.stream))

    @property
    def shape(self):
        return (self.rows, self.n)

    def row_block(self, row0, rows):
        """Rows [row0, row0+rows) as a SparseOperator of their own (device-side slicing, no copy of the
        entries: `col` / `val` are views, only the row pointers are rebased)."""
        lo = int(self.rowptr[row0].item())
        hi = int(self.rowptr[row0 + rows].item())
        rp = (self.rowptr[row0:row0 + rows + 1] - lo).contiguous()
        return SparseOperator((rp, self.col[lo:hi], self.val[lo:hi]), shape=(rows, self.n), engine=self.engine,
                              check=False, width=self.width, local_x=self.local_x,
                              diagonal_windows=self.diagonal_windows)

    def apply(self, eng, x, y, b=None, g=None):
        eng.spmv_csr(self.rowptr, self.col, self.val, self.plan, self.summary, self.n, x, y, b, g, self.width)

    def pipe_apply(self, eng, x, y, b=None, g=None):
        """apply() behind the pipeline's "converged" word."""
        _check(eng.lib.ccqp_pipe_spmv_csr_f64(eng.ctx, self.rowptr.data_ptr(), self.col.data_ptr(), self.val.data_ptr(),
                                              self.plan.data_ptr(), self.summary, x.data_ptr(),
                                              None if g is None else b.data_ptr(), y.data_ptr(),
                                              None if g is None else g.data_ptr(), self.rows, self.n, self.nnz,
                                              self.width, eng.stream))

    @property
    def T(self):
        """The transposed matrix as a SparseOperator of its own (built once on the device: a stable sort of
        the entries by column keeps the rows of A in order inside every row of A^T)."""
        if getattr(self, "_transposed", None) is None:
            counts = self.rowptr[1:] - self.rowptr[:-1]
            row_of = torch.repeat_interleave(torch.arange(self.rows, device=self.val.device), counts)
            order = torch.argsort(self.col.long(), stable=True)
            rp = torch.zeros(self.n + 1, dtype=torch.int64, device=self.val.device)
            rp[1:] = torch.cumsum(torch.bincount(self.col.long(), minlength=self.n), 0)
            self._transposed = SparseOperator((rp, row_of[order].to(torch.int32), self.val[order]),
                                              shape=(self.n, self.rows), engine=self.engine, check=False,
                                              width=self.width, local_x=self.local_x,
                                              diagonal_windows=self.diagonal_windows)
        return self._transposed

    def matvec(self, x, y, b=None, g=None, engine=None):
        self.apply(engine or self.engine, x, y, b, g)

    def dot(self, x):
        """numpy-style convenience: A.x as a new device vector."""
        eng = self.engine
        eng.bind()
        y = torch.empty(self.rows, dtype=torch.float64, device=eng.device)
        self.apply(eng, eng.to_device(x), y)
        return y


class DiagonalOperator:
    """y = d * x for a diagonal factor (e.g. an inverse mass matrix): one elementwise multiply, rounded once
    like numpy's `d * x`."""

    def __init__(self, d, engine=None):
        self.engine = engine or get_engine(d.device if isinstance(d, torch.Tensor) and d.is_cuda else None)
        self.d = self.engine.to_device(d).reshape(-1)
        self.rows = self.n = self.d.numel()

    @property
    def shape(self):
        return (self.rows, self.n)

    def apply(self, eng, x, y, b=None, g=None):
        eng.mul(self.d, x, y)
        if g is not None:
            eng.axpby(1.0, y, 1.0, b, g)

    def matvec(self, x, y, b=None, g=None, engine=None):
        self.apply(engine or self.engine, x, y, b, g)

    def pipe_apply(self, eng, x, y, b=None, g=None):
        _check(eng.lib.ccqp_pipe_mul_f64(eng.ctx, self.d.data_ptr(), x.data_ptr(), y.data_ptr(), y.numel(), eng.stream))
        if g is not None:
            _check(eng.lib.ccqp_pipe_axpby_f64(eng.ctx, 1.0, y.data_ptr(), 1.0, b.data_ptr(), g.data_ptr(), g.numel(),
                                               eng.stream))


class ComposedOperator:
    """Matrix-free A = F_1 . F_2 ... F_k (+ shift * I): the product is never formed, one matvec applies the
    factors right to left.  This is the shape of the Hessians CCQPs come from -- e.g. the
    Delassus operator D M^-1 D^T of a contact problem: `ComposedOperator(D, Minv, D.T, shift=eps)` with D a
    SparseOperator and Minv a DiagonalOperator.  Factors are DenseOperator / SparseOperator /
    DiagonalOperator (anything with `apply(engine, x, y)`, `rows`, `n`).  The reference needs only
    `A.dot(v)`, so the same object-with-a-dot can be handed to it; every intermediate is rounded where
    numpy rounds (`shift * v` and the final sum are separate operations).  Counts as ONE matvec."""

    def __init__(self, *factors, shift=0.0, engine=None):
        if not factors:
            raise ValueError("ComposedOperator needs at least one factor")
        for left, right in zip(factors[:-1], factors[1:]):
            if left.n != right.rows:
                raise ValueError("ComposedOperator: inner dimensions do not match (%d vs %d)" % (left.n, right.rows))
        if shift != 0.0 and factors[0].rows != factors[-1].n:
            raise ValueError("ComposedOperator: a shift needs a square product")
        self.factors = factors
        self.shift = float(shift)
        self.engine = engine or factors[-1].engine
        self.rows, self.n = factors[0].rows, factors[-1].n
        self.A = self

    @property
    def shape(self):
        return (self.rows, self.n)

    def matvec(self, x, y, b=None, g=None, engine=None):
        eng = engine or self.engine
        mv0 = eng.mv_executed
        cur = x
        last = len(self.factors) - 1
        for i, f in enumerate(reversed(self.factors)):
            out = y if (i == last and self.shift == 0.0) else eng.empty(f.rows)
            f.apply(eng, cur, out)
            cur = out
        if self.shift != 0.0:
            eng.axpby(1.0, cur, self.shift, x, y)          # t + shift * x, `shift * x` rounded first
        eng.mv_executed = mv0 + 1
        if g is not None:
            eng.axpby(1.0, y, 1.0, b, g)

    def dot(self, x):
        eng = self.engine
        eng.bind()
        y = torch.empty(self.rows, dtype=torch.float64, device=eng.device)
        self.matvec(eng.to_device(x), y)
        return y

    def pipe_apply(self, eng, x, y, b=None, g=None):
        """matvec() with every kernel behind the pipeline's "converged" word; sparse and diagonal factors only."""
        cur = x
        last = len(self.factors) - 1
        for i, f in enumerate(reversed(self.factors)):
            out = y if (i == last and self.shift == 0.0) else eng.empty(f.rows)
            f.pipe_apply(eng, cur, out)
            cur = out
        if self.shift != 0.0:
            _check(eng.lib.ccqp_pipe_axpby_f64(eng.ctx, 1.0, cur.data_ptr(), self.shift, x.data_ptr(), y.data_ptr(),
                                               y.numel(), eng.stream))
        if g is not None:
            _check(eng.lib.ccqp_pipe_axpby_f64(eng.ctx, 1.0, y.data_ptr(), 1.0, b.data_ptr(), g.data_ptr(), g.numel(),
                                               eng.stream))

    def can_pipeline(self):
        return all(isinstance(f, (SparseOperator, DiagonalOperator)) for f in self.factors)


class RowShardedOperator:
    """A row block of A per rank + one vector exchange per matvec.  The local block is a dense
    (rows, n) device matrix or a SparseOperator holding the same rows in CSR."""

    def __init__(self, A_local, n, group=None, mode="allgather", engine=None):
        self.sparse = isinstance(A_local, SparseOperator)
        self.engine = engine or (A_local.engine if self.sparse else get_engine(A_local.device))
        self.A = A_local
        self.n = n
        self.exchange = VectorExchange(n, group, mode)
        self.exchange.open(self.engine)
        self.row0, self.rows = self.exchange.row0, self.exchange.rows
        assert tuple(A_local.shape) == (self.rows, n), "local block must be rows [row0, row0+rows) x n"

    def matvec(self, x, y, b=None, g=None, engine=None):
        eng = engine or self.engine
        sl = slice(self.row0, self.row0 + self.rows)
        if self.sparse:
            self.A.apply(eng, x, y[sl])
        else:
            eng.gemv(self.A, x, y[sl])
        self.exchange(y, eng)
        if g is not None:
            eng.axpby(1.0, y, 1.0, b, g)   # same bits as the fused epilogue: one add

    def matvec2(self, x1, y1, g1, x2, y2, g2, b=None, engine=None):
        """Two products: ONE pass over the local row block (Engine.gemv2) straight into the exchange's send slabs, ONE
        exchange, ONE unpack launch that also forms g_j = y_j + b -- per APGD iteration 1 matvec launch + 1 collective +
        1 small launch (round 3: 1 + 2 copies + 2 collectives + 2 axpby)."""
        self.matvec_many([x1, x2], [y1, y2], [g1, g2], b, engine)

    def matvec3(self, xs, ys, gs, b=None, engine=None):
        """ys[j] = A.xs[j] (gs[j] = ys[j] + b where given) for the j with ys[j] not None -- MPRGP's look-ahead on a
        row-sharded matrix: one pass over the local block and one exchange for all of them."""
        act = [j for j in range(len(xs)) if ys[j] is not None]
        self.matvec_many([xs[j] for j in act], [ys[j] for j in act], [gs[j] for j in act], b, engine)

    def matvec_many(self, xs, ys, gs, b=None, engine=None):
        eng = engine or self.engine
        k = len(xs)
        if k == 1 or self.sparse:
            for x, y, g in zip(xs, ys, gs):
                self.matvec(x, y, b, g, eng)
            return
        slabs = self.exchange.slabs(k, ys[0])
        if self.rows:
            eng.gemvk(self.A, xs, slabs)
        else:
            eng.mv_executed += k                        # (a rank without rows still counts the products it takes part in)
            eng.products_launched += k
        self.exchange.exchange_many(ys, b, gs, eng)

    def _block_aligned(self):
        return (not self.sparse) and self.A.data_ptr() % 16 == 0 and self.A.stride(0) % 2 == 0

    def pair_in_one_pass(self, engine=None):
        eng = engine or self.engine
        return self._block_aligned() and hasattr(eng, "gemv2_form") and eng.gemv2_form(self.rows, self.n) == "gemv_lds2_kernel"

    def triple_in_one_pass(self, engine=None):
        eng = engine or self.engine
        return self._block_aligned() and hasattr(eng, "gemvk_form") and eng.gemvk_form(3, self.rows, self.n) == "gemv_lds3_kernel"


def as_operator(A, engine=None):
    if hasattr(A, "matvec") and hasattr(A, "engine"):
        return A
    if _is_sparse_input(A):
        return SparseOperator(A, engine=engine)
    return DenseOperator(A, engine)


class RegeneratedRowBlocksOperator:
    """Matrix-free form of the synthetic generator matrix (operator generality): A is
    never stored; each matvec visits `blocks` row blocks one after the other, regenerating each block
    in one reusable buffer (`ccqp_gen_spd_rows_f64`, 64 GiB in ~20 ms) and multiplying it with the
    production GEMV.  The row sums are those of any row-sharded run (GEMV bits depend on (row, n) only),
    so this reproduces an 8-rank n = 262144 solve on ONE GPU, where 512 GiB could not be resident."""

    def __init__(self, n, seed, blocks, engine=None):
        assert n % blocks == 0
        self.engine = engine or get_engine()
        self.n, self.seed, self.blocks = n, seed, blocks
        self.rows = n // blocks
        self.buf = torch.empty((self.rows, n), dtype=torch.float64, device=self.engine.device)
        self.A = self.buf

    def matvec(self, x, y, b=None, g=None, engine=None):
        eng = engine or self.engine
        for r in range(self.blocks):
            eng.gen_spd_rows(self.buf, r * self.rows, self.n, self.seed)
            eng.gemv(self.buf, x, y[r * self.rows:(r + 1) * self.rows])
        eng.mv_executed -= self.blocks - 1          # one matvec per call, however many blocks
        if g is not None:
            eng.axpby(1.0, y, 1.0, b, g)


class SymmetricOperator:
    """OPT-IN, NON-parity operator for a fully stored SYMMETRIC matrix (the Hessian of a convex QP always
    is): `ccqp_symv_upper_f64` reads only the upper triangle, 4n^2 bytes per matvec instead of 8n^2
    (SURVEY.md 8(f)3).  The reference's `A.dot(v)` never assumes symmetry, so the summation order -- and
    hence the last bits of A.x -- differ from DenseOperator; MV counts are those of the reference
    wherever its control flow is stable under 1-ulp perturbations.  Requires n % 128 == 0."""

    def __init__(self, A, engine=None, check=True, use_general=None):
        self.engine = engine or get_engine(A.device if isinstance(A, torch.Tensor) and A.is_cuda else None)
        self.A = self.engine.to_device(A)
        assert self.A.dim() == 2 and self.A.shape[0] == self.A.shape[1]
        self.n = self.rows = self.A.shape[0]
        if self.n % 128:
            raise ValueError("SymmetricOperator needs n % 128 == 0 (got n = %d)" % self.n)
        if check:
            k = min(self.n, 512)
            blk = self.A[:k, -k:]
            if not torch.equal(blk, self.A[-k:, :k].t()):
                raise ValueError("SymmetricOperator: the matrix is not symmetric")
        # below ~16k the triangle kernel's two launches + combine cost more than they save
        # (n = 8192: 0.17 ms vs 0.08 ms for the general matvec; n = 32768: 0.76 vs 1.23 ms; n = 131072:
        # 12.1 vs 20.0 ms): small problems go through the general kernel
        self.use_general = self.n < 16384 if use_general is None else use_general
        self.workspace = None
        if not self.use_general:
            words = self.engine.lib.ccqp_symv_workspace_doubles(self.n)
            self.workspace = torch.empty(words, dtype=torch.float64, device=self.engine.device)

    def matvec(self, x, y, b=None, g=None, engine=None):
        eng = engine or self.engine
        if self.use_general:
            eng.gemv(self.A, x, y, b, g)
            return
        flush = getattr(eng, "flush_self", None)
        if flush is not None:
            flush()                                   # recording engines: keep program order
        eng.mv_executed += 1
        rc = eng.lib.ccqp_symv_upper_f64(eng.ctx, self.A.data_ptr(), self.A.stride(0), x.data_ptr(),
                                         None if g is None else b.data_ptr(), y.data_ptr(),
                                         None if g is None else g.data_ptr(), self.n, self.workspace.data_ptr(),
                                         eng.stream)
        if rc:
            from . import _lib
            _lib.check(rc)
