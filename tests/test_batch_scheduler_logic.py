"""CPU: the merging logic of the fused batch scheduler (ccqppy_amd.batch._flush_group) with fake
recording engines -- no GPU, no library calls.  Property under test: the groups handed to
ccqp_batch_run_f64 (a) contain every recorded operation exactly once, (b) only merge operations with
the same signature, (c) keep every lane's operations in their recorded order (groups are issued in
queue-position order)."""
import ctypes

from ccqppy_amd import batch


class _FakeLib:
    def __init__(self):
        self.calls = []

    def ccqp_batch_run_f64(self, ctx, ngroups, codes, counts, ops, stream):
        self.calls.append(([codes[i] for i in range(ngroups)], [counts[i] for i in range(ngroups)]))
        return 0


class _FakeEngine:
    def __init__(self, name, ops, lib=None):
        self.name, self.queue, self.lib, self.ctx, self.stream = name, list(ops), lib, None, None
        self.written = []
        self.released = False

    def take(self):
        q, self.queue = self.queue, []
        return q

    def write_bop(self, bop, op):
        bop.append((self.name, op))

    def release(self):
        self.released = True


class _Lane:
    def __init__(self, eng):
        self.eng = eng


def _op(code, *shape):
    return (code, (code,) + shape, [], (), 0, 0, 0, (), 0, None)


def test_flush_groups_by_position_and_signature():
    lib = _FakeLib()
    grp = _FakeEngine("grp", [], lib)
    a = _FakeEngine("a", [_op(1, 8), _op(2, 8), _op(6, 8, 2)])
    b = _FakeEngine("b", [_op(1, 8), _op(2, 8), _op(6, 8, 3), _op(2, 8)])
    c = _FakeEngine("c", [_op(1, 16), _op(7, 8)])
    ops_buf = _ListOfLists(32)
    codes = (ctypes.c_int * 32)()
    counts = (ctypes.c_int * 32)()
    nops, ngroups = batch._flush_group([_Lane(a), _Lane(b), _Lane(c)], grp, ops_buf, codes, counts)
    assert nops == 9 and a.released and b.released and c.released
    assert len(lib.calls) == 1
    got_codes, got_counts = lib.calls[0]
    # position 0: gemv(8) x2 merged, gemv(16) alone; position 1: axpby x2 merged, op 7 alone;
    # position 2: dots k=2 and dots k=3 must NOT merge; position 3: the trailing axpby of b
    assert got_codes == [1, 1, 2, 7, 6, 6, 2]
    assert got_counts == [2, 1, 2, 1, 1, 1, 1]
    assert ngroups == 7 and sum(got_counts) == nops
    flat = [e[0] for e in ops_buf.items[:nops]]
    # every lane's own operations appear in recorded order
    for eng, want in ((a, [1, 2, 6]), (b, [1, 2, 6, 2]), (c, [1, 7])):
        mine = [op[0] for name, op in flat if name == eng.name]
        assert mine == want


def test_flush_with_nothing_recorded_launches_nothing():
    lib = _FakeLib()
    grp = _FakeEngine("grp", [], lib)
    a = _FakeEngine("a", [])
    nops, ngroups = batch._flush_group([_Lane(a)], grp, _ListOfLists(4), (ctypes.c_int * 4)(), (ctypes.c_int * 4)())
    assert (nops, ngroups) == (0, 0) and lib.calls == [] and a.released


class _ListOfLists:
    """Stands in for the ctypes array of descriptors: element i collects what write_bop stored."""

    def __init__(self, n):
        self.items = [[] for _ in range(n)]

    def __getitem__(self, i):
        return self.items[i]

    def __len__(self):
        return len(self.items)
