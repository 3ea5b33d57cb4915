"""CPU: the host-side planner of the segmented projection engine (`ccqp_segs_plan`, csrc/ccqp_segs.hip) -- pure C, no
GPU work.  A feasible set = table of blocks; the plan = one work item per workgroup (runs of small blocks, 2048-element
chunks of large ones) + the chunk-sum layout.  Counterpart of the Python loop of DisjointProjOp.__call__
(solution_spaces.py:540-560)."""
import ctypes

import pytest

from ccqppy_amd import _lib


def _plan(dims_kinds):
    lib = _lib.load()
    n = len(dims_kinds)
    segs = (_lib.Seg * max(n, 1))()
    off = 0
    for i, (d, k) in enumerate(dims_kinds):
        segs[i].kind, segs[i].offset, segs[i].dim = k, off, d
        off += d
    nw, nn, npart = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.ccqp_segs_plan(segs, n, None, 0, ctypes.byref(nw), None, 0, ctypes.byref(nn), ctypes.byref(npart)))
    work = (_lib.SegWork * max(nw.value, 1))()
    norm = (_lib.SegWork * max(nn.value, 1))()
    _lib.check(lib.ccqp_segs_plan(segs, n, work, nw.value, ctypes.byref(nw), norm, nn.value, ctypes.byref(nn),
                                  ctypes.byref(npart)))
    items = [(w.type, w.seg, w.chunk, w.pbase) for w in work[:nw.value]]
    nitems = [(w.type, w.seg, w.chunk, w.pbase) for w in norm[:nn.value]]
    return items, nitems, npart.value


def test_thousand_spheres_of_100_are_a_thousand_items_and_two_launches_worth_of_lists():
    items, norm, npart = _plan([(100, _lib.OP_SPHERE)] * 1000)
    assert len(items) == 1000 and len(norm) == 1000 and npart == 1000
    assert items[7] == (1, 7, 0, 7) and norm == items


def test_small_blocks_are_grouped_in_runs_of_up_to_256():
    items, norm, npart = _plan([(3, _lib.OP_SPHERE)] * 1365)
    assert norm == [] and npart == 0
    assert items == [(0, 256 * r, min(256, 1365 - 256 * r), 0) for r in range(6)]
    # empty blocks ride along in a run; a large block closes it
    items, norm, npart = _plan([(5, _lib.OP_BOX), (0, _lib.OP_CONE), (64, _lib.OP_CONE), (65, _lib.OP_CONE),
                                (2, _lib.OP_LOWER)])
    assert items == [(0, 0, 3, 0), (1, 3, 0, 0), (0, 4, 1, 0)]
    assert norm == [(1, 3, 0, 0)] and npart == 1


def test_large_blocks_are_chunked_with_the_reduction_geometry():
    # config 4's set: Box(n/2) + Cone(n/2), n = 262144 -> 64 chunks each; only the Cone's go to the norm launch
    items, norm, npart = _plan([(131072, _lib.OP_BOX), (131072, _lib.OP_CONE)])
    assert len(items) == 128 and npart == 128
    assert items[:2] == [(1, 0, 0, 0), (1, 0, 1, 0)] and items[64] == (1, 1, 0, 64) and items[-1] == (1, 1, 63, 64)
    assert norm == items[64:]
    # beyond 1024 chunks of 2048 the chunk grows (ccqp_red_geometry): never more than 1024 items per block
    items, norm, npart = _plan([(2048 * 1024 + 1, _lib.OP_SPHERE)])
    assert len(items) == npart == len(norm) and npart <= 1024


def test_errors_and_workspace_size():
    lib = _lib.load()
    segs = (_lib.Seg * 1)()
    segs[0].kind, segs[0].dim = 17, 4
    out = [ctypes.c_int64() for _ in range(3)]
    rc = lib.ccqp_segs_plan(segs, 1, None, 0, ctypes.byref(out[0]), None, 0, ctypes.byref(out[1]), ctypes.byref(out[2]))
    assert rc != 0 and b"kind" in lib.ccqp_last_error()
    segs[0].kind, segs[0].dim = _lib.OP_BOX, -1
    assert lib.ccqp_segs_plan(segs, 1, None, 0, ctypes.byref(out[0]), None, 0, ctypes.byref(out[1]),
                              ctypes.byref(out[2])) != 0
    assert lib.ccqp_segs_workspace_doubles(10, 20) == max(3 * 10 + 20, 16 * (10 + 20)) + 8
    assert lib.ccqp_segs_workspace_doubles(-1, 0) == -1
    with pytest.raises(_lib.CCQPHipError):
        _lib.check(1)
