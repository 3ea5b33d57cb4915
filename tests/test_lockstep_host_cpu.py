"""CPU: the host side of the lock-step schedulers that needs no GPU -- the run-ahead / polling loop against a scripted
device, and the descriptor records (ctypes structs vs. the numpy dtypes the tables are built with)."""
import ctypes
import types

import numpy as np
import pytest


class _FakeDevice:
    """Plays the device: `lag` rounds behind the host's queue, all `B` problems converged at round `done_at`."""

    def __init__(self, B, lag, done_at):
        self.B, self.lag, self.done_at = B, lag, done_at
        self.queued = 0
        self.polls = 0
        self.max_ahead = 0
        self.seen = -1.0

    def launch(self, k):
        assert k == self.queued + 1                      # rounds are queued in order, each once
        self.queued = k

    def ccqp_bpipe_poll(self, ctx, state):
        self.polls += 1
        if self.polls % 3 == 0:                          # the device finishes one more round every third poll
            self.seen = min(self.seen + 1 if self.seen >= 0 else 0.0, float(self.queued - self.lag if self.lag else self.queued))
            self.seen = max(self.seen, -1.0)
        rep = max(self.seen, -1.0)
        state[1] = rep
        state[0] = float(self.B) if rep >= self.done_at else 0.0
        self.max_ahead = max(self.max_ahead, self.queued - max(rep, 0.0))
        return 0


def _eng(dev):
    return types.SimpleNamespace(lib=dev, ctx=None, device=None)


@pytest.mark.parametrize("ahead", [1, 2, 4, 9])
def test_poll_loop_bounds_the_run_ahead_and_stops_on_convergence(ahead):
    from ccqppy_amd.lockstep import _bpipe_poll_loop
    dev = _FakeDevice(B=5, lag=0, done_at=17)
    k = _bpipe_poll_loop(_eng(dev), 5, dev.launch, np.inf, ahead)
    assert 17 <= k <= 17 + ahead                          # stops within `ahead` rounds of the converging one
    assert dev.max_ahead <= ahead


def test_poll_loop_stops_at_the_budget_without_waiting_for_a_report():
    from ccqppy_amd.lockstep import _bpipe_poll_loop
    dev = _FakeDevice(B=3, lag=0, done_at=10 ** 9)
    k = _bpipe_poll_loop(_eng(dev), 3, dev.launch, 12, 4)
    assert k == 12 and dev.queued == 12
    dev = _FakeDevice(B=3, lag=0, done_at=10 ** 9)
    assert _bpipe_poll_loop(_eng(dev), 3, dev.launch, 0, 4) == 0 and dev.queued == 0      # no round at all
    dev = _FakeDevice(B=3, lag=0, done_at=10 ** 9)
    assert _bpipe_poll_loop(_eng(dev), 3, dev.launch, 1, 4) == 1 and dev.polls == 0       # last round: no wait


@pytest.mark.parametrize("ahead", [1, 4])
def test_poll_many_keeps_every_run_within_its_own_run_ahead(ahead):
    """Two lock-step runs advanced together (the two halves of a large MPRGP batch, each on its own engine): rounds of
    each are queued in order, never more than `ahead` beyond that run's own last report, and each run stops on its own
    convergence or budget."""
    from ccqppy_amd.lockstep import _bpipe_poll_many
    devs = [_FakeDevice(B=5, lag=0, done_at=17), _FakeDevice(B=3, lag=0, done_at=9), _FakeDevice(B=2, lag=0, done_at=10 ** 9)]
    lasts = [np.inf, np.inf, 12]
    runs = []
    for d, last in zip(devs, lasts):
        eng = types.SimpleNamespace(lib=d, ctx=None, device=None, torch_stream=None)
        runs.append(types.SimpleNamespace(eng=eng, B=d.B, launch=d.launch, last=last))
    _bpipe_poll_many(runs, ahead)
    assert 17 <= runs[0].k <= 17 + ahead and 9 <= runs[1].k <= 9 + ahead and runs[2].k == 12
    for d, r in zip(devs, runs):
        assert d.queued == r.k and d.max_ahead <= ahead
    # no round at all / a single round: nothing is waited for
    d = _FakeDevice(B=1, lag=0, done_at=10 ** 9)
    eng = types.SimpleNamespace(lib=d, ctx=None, device=None, torch_stream=None)
    r0 = types.SimpleNamespace(eng=eng, B=1, launch=d.launch, last=0)
    _bpipe_poll_many([r0], ahead)
    assert r0.k == 0 and d.queued == 0
    r1 = types.SimpleNamespace(eng=eng, B=1, launch=d.launch, last=1)
    _bpipe_poll_many([r1], ahead)
    assert r1.k == 1 and d.polls == 0


def test_spec_part_slices_every_array():
    import torch
    from ccqppy_amd.lockstep import _Spec
    sp = _Spec()
    sp.B, sp.n = 5, 3
    sp.A = np.arange(5, dtype=np.uint64)
    sp.lda = np.arange(5, dtype=np.int64) + 10
    sp.b = np.arange(5, dtype=np.uint64) + 20
    sp.seg = np.zeros(5, dtype=np.int8)
    sp.tol = np.linspace(0.1, 0.5, 5)
    sp.x0 = [None, torch.ones(3), None, None, torch.zeros(3)]
    sp.keep, sp.step, sp.fallback = None, None, False
    h = sp.part(3, 5)
    assert len(h) == 2 and h.n == 3 and list(h.A) == [3, 4] and list(h.lda) == [13, 14] and list(h.b) == [23, 24]
    assert h.x0[0] is None and h.x0[1] is sp.x0[4] and h.tol[1] == sp.tol[4]
    sp.x0 = None
    assert sp.part(0, 2).x0 is None


def test_descriptor_dtypes_match_the_c_structs():
    from ccqppy_amd import _lib
    for struct, size in ((_lib.Seg, 56), (_lib.PProb, 192), (_lib.PApgd, 208), (_lib.PMprgp, 224)):
        dt = np.dtype(struct)
        assert dt.itemsize == ctypes.sizeof(struct) == size
        for name, _ in struct._fields_:
            assert dt.fields[name][1] == getattr(struct, name).offset
    # the records embed struct ccqp_seg at the same offset the C header gives it (after the vector pointers)
    assert np.dtype(_lib.PProb).fields["seg"][1] == 80
    assert np.dtype(_lib.PApgd).fields["seg"][1] == 96
    assert np.dtype(_lib.PMprgp).fields["seg"][1] == 128


def test_spec_staging_of_start_vectors():
    import torch
    from ccqppy_amd.lockstep import _Spec
    sp = _Spec()
    sp.B, sp.n = 3, 4
    plane = torch.full((3, 4), 7.0, dtype=torch.float64)
    sp.x0 = None
    sp.stage_x0(plane)
    assert torch.equal(plane, torch.zeros(3, 4, dtype=torch.float64))
    sp.x0 = [None, torch.arange(4, dtype=torch.float64), None]
    plane.fill_(7.0)
    sp.stage_x0(plane)
    assert torch.equal(plane[1], torch.arange(4, dtype=torch.float64)) and float(plane[0].abs().sum() + plane[2].abs().sum()) == 0.0
    sp.x0 = torch.ones(3, 4, dtype=torch.float64)
    sp.stage_x0(plane)
    assert torch.equal(plane, torch.ones(3, 4, dtype=torch.float64))


def test_spg_descriptor_layouts_and_draw_source():
    """struct ccqp_pspg / ccqp_spg_host mirrors, and the address handed to the C driver as a draw source: the numpy
    bitgen_t record of a RandomState, whose next_double is the double `uniform` / `random_sample` consume."""
    import ctypes
    from ccqppy_amd import _lib
    from ccqppy_amd.lockstep import _draw_source
    for struct, size in ((_lib.PSpg, 128), (_lib.SpgHost, 120)):
        assert np.dtype(struct).itemsize == ctypes.sizeof(struct) == size
    assert np.dtype(_lib.PSpg).fields["seg"][1] == 56 and np.dtype(_lib.SpgHost).fields["mv"][1] == 72

    class BitGen(ctypes.Structure):                      # numpy/random/bitgen.h
        _fields_ = [("state", ctypes.c_void_p), ("next_uint64", ctypes.c_void_p), ("next_uint32", ctypes.c_void_p),
                    ("next_double", ctypes.c_void_p), ("next_raw", ctypes.c_void_p)]
    rng, twin = np.random.RandomState(99), np.random.RandomState(99)
    rec = BitGen.from_address(_draw_source(rng))
    nxt = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p)(rec.next_double)
    got = [nxt(rec.state) for _ in range(5)]
    assert got == list(twin.random_sample(5))
    assert rng.uniform(2.0, 3.0) == twin.uniform(2.0, 3.0)       # ... and the Python object saw the draws
    # numpy's GLOBAL generator has such a record too (what pipeline=True uses without solver.rng)
    np.random.seed(5)
    rec = BitGen.from_address(_draw_source(np.random.mtrand._rand))
    nxt = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p)(rec.next_double)
    first = nxt(rec.state)
    np.random.seed(5)
    assert first == np.random.random_sample()


def test_spg_argument_rules_of_the_lockstep_scheduler():
    """SPG in lock-step: one RandomState per solver, no trace, a sane queue length -- decided without any GPU work."""
    import torch
    from ccqppy_amd import lockstep, solvers
    A = torch.eye(8, dtype=torch.float64)
    b = torch.ones(8, dtype=torch.float64)
    probs = [(A, b, None, None), (A, b, None, None)]

    def made(**kw):
        out = []
        for k in range(2):
            s = solvers.CCQPSolverSPG(1e-8, 100, **kw.get("ctor", {}))
            s.rng = kw["rng"](k) if "rng" in kw else np.random.RandomState(k)
            if kw.get("trace"):
                s.trace = []
            out.append(s)
        return out
    dev = torch.device("cpu")
    assert lockstep.why_not_pipelined(made(), probs, dev) is None
    assert isinstance(lockstep.why_not_pipelined(made(rng=lambda k: None), probs, dev), TypeError)
    shared = np.random.RandomState(0)
    assert "share" in str(lockstep.why_not_pipelined(made(rng=lambda k: shared), probs, dev))
    assert isinstance(lockstep.why_not_pipelined(made(rng=lambda k: np.random.default_rng(k)), probs, dev), TypeError)
    assert "traced" in str(lockstep.why_not_pipelined(made(trace=True), probs, dev))
    assert isinstance(lockstep.why_not_pipelined(made(ctor={"m": 0}), probs, dev), ValueError)
    assert isinstance(lockstep.why_not_pipelined(made(ctor={"m": 65}), probs, dev), ValueError)
    assert isinstance(lockstep.why_not_pipelined(made(ctor={"m": 2.5}), probs, dev), ValueError)
    mixed = made()
    mixed[1] = solvers.CCQPSolverBBPGD(1e-8, 100)
    assert isinstance(lockstep.why_not_pipelined(mixed, probs, dev), TypeError)


@pytest.mark.parametrize("n", [64, 5000])          # (bytes snapshots up to 16 KiB, array snapshots beyond)
def test_refresh_memo_shares_snapshots_between_operators_and_still_sees_edits(n):
    """solve_batch_pipelined's staging pass compares a bound array with its snapshot once per pass and bound OBJECT:
    operators over the same arrays share one state; an in-place edit, a reassignment, or an operator refreshed on its
    own in between are still seen by every operator (the reference re-reads its bounds at every call,
    solution_spaces.py:363-366)."""
    from ccqppy_amd import solution_spaces as ss
    lb, ub = -np.ones(n), np.ones(n)
    ops = [ss.BoxProjOp(n, lb, ub) for _ in range(4)]
    ops[3]._refresh()                                      # one of them already has a state (and snapshots) of its own
    memo = {}
    assert [P._refresh(memo) for P in ops] == [False] * 4
    assert all(P._bounds_state is ops[0]._bounds_state for P in ops)
    assert [P._refresh({}) for P in ops] == [False] * 4    # a new pass: nothing changed
    v0 = [P._version for P in ops]
    lb[3] = -2.0                                           # in-place edit: every operator's caches are stale
    memo = {}
    assert [P._refresh(memo) for P in ops] == [True] * 4
    assert [P._version for P in ops] == [v + 1 for v in v0]
    assert [P._refresh({}) for P in ops] == [False] * 4
    ub[5] = 3.0
    assert ops[1]._refresh() is True                       # refreshed on its own (e.g. called directly) ...
    memo = {}
    assert [P._refresh(memo) for P in ops] == [True, False, True, True]    # ... the others still notice in the pass
    ops[2].lower_bound = lb.copy()                         # reassignment: another object, the same bits
    memo = {}
    assert ops[2]._refresh(memo) is True and [P._refresh(memo) for P in ops[:2]] == [False, False]
    ident = [ss.IdentityProjOp(n) for _ in range(3)]
    memo = {}
    assert [P._refresh(memo) for P in ident] == [False] * 3
