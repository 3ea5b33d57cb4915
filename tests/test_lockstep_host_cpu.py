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


def test_descriptor_dtypes_match_the_c_structs():
    from ccqppy_amd import _lib
    for struct, size in ((_lib.Seg, 56), (_lib.PProb, 192), (_lib.PApgd, 192), (_lib.PMprgp, 224)):
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
