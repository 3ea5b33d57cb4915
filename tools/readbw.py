#!/usr/bin/env python3
"""What read bandwidth does this MI355X give a kernel that does NOTHING but stream the 128 GiB of the headline matrix?
(tools/readbw.hip: each wave sums 1 MiB segments with 16-byte non-temporal loads, the matvec's access pattern without
x, fma chains or the butterfly.)  Launch shapes: one persistent grid, one wave per segment, rounds that fill the machine
once.  The best of them is the practical ceiling the matvec's 7.04 TB/s is to be read against.
Compiles the kernel with hipcc into $TMPDIR at run time.      python tools/readbw.py"""
import ctypes, json, os, subprocess, sys, tempfile
import numpy as np, torch
_here = os.path.dirname(os.path.abspath(__file__))
_so = os.path.join(tempfile.gettempdir(), "libreadbw_%d.so" % os.getpid())
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", _so, os.path.join(_here, "readbw.hip")])
lib = ctypes.CDLL(_so)
lib.readbw_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n = 131072
A = torch.empty((n, n), dtype=torch.float64, device="cuda")
A.fill_(1.0)
out = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(seg, grid_per_launch, rows_per_launch, U, nt):
    nseg_total = n * n // seg
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    done = 0
    while done < nseg_total:
        k = min(rows_per_launch, nseg_total - done)
        assert lib.readbw_launch(A.data_ptr() + done * seg * 8, seg, k, out.data_ptr(), grid_per_launch, U, nt, st) == 0
        done += k
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
cfgs = []
for U in (2, 4, 8):
    cfgs += [("persistent grid 2048", n, 2048, n, U, 1), ("rounds 8192 segs/launch, 2048 wg", n, 2048, 8192, U, 1),
             ("one wave per segment, 32768 wg", n, 32768, n, U, 1), ("rounds 4096 segs, 1024 wg", n, 1024, 4096, U, 1)]
cfgs += [("seg 32768 doubles, one wave per seg, rounds 32768 segs, 8192 wg", 32768, 8192, 32768, 4, 1),
         ("seg 524288, rounds 8192 segs, 2048 wg", 524288, 2048, 8192, 4, 1),
         ("cached loads, rounds 8192", n, 2048, 8192, 2, 0)]
res = {i: [] for i in range(len(cfgs))}
for i, c in enumerate(cfgs): run(*c[1:])
for rep in range(4):
    for i, c in enumerate(cfgs):
        res[i].append(run(*c[1:]))
for i, c in enumerate(cfgs):
    t = float(np.median(res[i]))
    print(json.dumps({"config": c[0], "U": c[4], "ms": round(t, 3), "TBps": round(8.0 * n * n / t / 1e9, 3)}), flush=True)

# ---- LDS-DMA loads (round 3): persistent grids, R segments per wave at a time, ring of S slots, nothing consumed
lib.readbw_lds_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
def run_lds(seg, grid, wpb, R, S, blocked=0):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.readbw_lds_launch(A.data_ptr(), seg, n * n // seg, grid, wpb, R, S, blocked, st)
    assert rc == 0, rc
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
lcfgs = [(n, 256, 2, 4, 4), (n, 256, 4, 4, 4), (n, 256, 2, 4, 8), (n, 256, 4, 2, 8), (n, 256, 2, 8, 4), (n, 512, 1, 4, 4),
         (n, 512, 2, 4, 4), (n, 256, 4, 1, 16), (n, 256, 4, 4, 2), (n, 256, 8, 4, 2), (n, 256, 1, 4, 8), (n, 256, 1, 8, 4),
         (8192, 256, 2, 4, 4), (8192, 256, 4, 4, 4), (8192, 256, 2, 4, 8), (32768, 256, 2, 4, 4),
         (8192, 256, 2, 4, 4, 1), (8192, 256, 4, 4, 4, 1), (8192, 256, 2, 4, 8, 1), (8192, 256, 4, 1, 16, 1), (8192, 256, 4, 1, 8, 1),
         (8192, 256, 2, 1, 16, 1), (8192, 256, 4, 1, 16, 0), (8192, 256, 4, 2, 8, 1), (8192, 256, 4, 2, 4, 1), (8192, 256, 8, 1, 8, 1),
         (131072, 256, 2, 4, 4, 1), (32768, 256, 2, 4, 4, 1), (8192, 512, 2, 4, 4, 1), (8192, 512, 1, 4, 4, 1), (8192, 256, 1, 4, 8, 1),
         (4096, 256, 2, 4, 4, 0), (4096, 256, 2, 4, 4, 1), (4096, 256, 4, 1, 16, 1)]
lres = {i: [] for i in range(len(lcfgs))}
for i, c in enumerate(lcfgs): run_lds(*c)
for rep in range(4):
    for i, c in enumerate(lcfgs):
        lres[i].append(run_lds(*c))
for i, c in enumerate(lcfgs):
    t = float(np.median(lres[i]))
    print(json.dumps({"config": "LDS-DMA ring, segments of %d doubles, %d workgroups x %d waves, R %d segments per wave, S %d slots, %s "
                      "(%d KiB in flight per workgroup)" % (c[0], c[1], c[2], c[3], c[4],
                                                             "each wave a contiguous run of tasks" if len(c) > 5 and c[5] else "moving band",
                                                             c[2] * c[3] * (c[4] - 1)),
                      "ms": round(t, 3), "TBps": round(8.0 * n * n / t / 1e9, 3)}), flush=True)

# ---- static against DYNAMIC task assignment (a global counter, drawn one task ahead), on the whole matrix and on the
# first 65536 / 32768 / 16384 rows (the row slabs one rank holds at 2 / 4 / 8 GPUs)
lib.readbw_lds_dyn_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
counter = torch.zeros(4, dtype=torch.int32, device="cuda")
def run_pair(rows, dyn, grid=256, wpb=2, R=4, S=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if dyn:
        rc = lib.readbw_lds_dyn_launch(A.data_ptr(), n, rows, grid, wpb, R, S, counter.data_ptr(), st)
    else:
        rc = lib.readbw_lds_launch(A.data_ptr(), n, rows, grid, wpb, R, S, 0, st)
    assert rc == 0, rc
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for rows in (n, n // 2, n // 4, n // 8):
    t = {0: [], 1: []}
    for rep in range(7):
        for dyn in (0, 1):
            ms = run_pair(rows, dyn)
            if rep:
                t[dyn].append(ms)
    for dyn in (0, 1):
        m = float(np.median(t[dyn]))
        print(json.dumps({"config": "LDS-DMA ring, %d rows of 1 MiB, 256 workgroups x 2 waves, R 4, S 4, %s task assignment"
                          % (rows, "DYNAMIC" if dyn else "static"), "ms": round(m, 4), "min_ms": round(min(t[dyn]), 4),
                          "TBps": round(8.0 * rows * n / m / 1e9, 3)}), flush=True)

# ---- the static ring with the matvec's x stream beside it (one shared vector, 1 KiB per step and wave, out of L2), and with
# the waves' starts staggered by up to 63 * k * 64 clocks
lib.readbw_lds_x_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_void_p]
xv = torch.ones(n, dtype=torch.float64, device="cuda")
def run_x(rows, stagger):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.readbw_lds_x_launch(A.data_ptr(), xv.data_ptr(), n, rows, 256, 2, stagger, st)
    assert rc == 0, rc
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for rows in (n, n // 2, n // 4, n // 8):
    modes = [("no x", None), ("x", 0), ("x, stagger 1", 1), ("x, stagger 4", 4), ("x, stagger 16", 16), ("x, stagger 64", 64)]
    t = {m[0]: [] for m in modes}
    for rep in range(7):
        for name, sg in modes:
            ms = run_pair(rows, 0) if sg is None else run_x(rows, sg)
            if rep:
                t[name].append(ms)
    for name, sg in modes:
        m = float(np.median(t[name]))
        print(json.dumps({"config": "LDS-DMA ring, %d rows of 1 MiB, 256 workgroups x 2 waves, R 4, S 4, static: %s" % (rows, name),
                          "ms": round(m, 4), "min_ms": round(min(t[name]), 4), "TBps": round(8.0 * rows * n / m / 1e9, 3)}), flush=True)

# ---- spacing of the R segments a wave reads side by side (64 KiB and 32 KiB segments: config 5's rows and n = 4096)
lib.readbw_lds_rs_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
def run_rs(seg, wpb, R, S, rstride):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.readbw_lds_rs_launch(A.data_ptr(), seg, n * n // seg, 256, wpb, R, S, rstride, st)
    assert rc == 0, rc
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
rcfgs = [(8192, 2, 4, 4, rs) for rs in (1, 2, 4, 8, 16, 32, 64, 256, 2048)] + [(8192, 4, 2, 8, rs) for rs in (1, 16, 64)] + \
        [(4096, 2, 4, 4, rs) for rs in (1, 8, 32, 128)] + \
        [(131072, 2, 4, 4, rs) for rs in (1, 2, 3, 4, 5, 8, 16, 32, 64, 128, 256, 512, 1024)] + \
        [(262144, 2, 4, 4, rs) for rs in (1, 2, 4, 8, 16, 32, 64, 128, 256)] + \
        [(32768, 2, 4, 4, rs) for rs in (1, 4, 8)] + [(65536, 2, 4, 4, rs) for rs in (1, 2, 4)] + \
        [(16384, 2, 4, 4, rs) for rs in (1, 8, 16)] + [(8192, 2, 4, 4, rs) for rs in (12, 20, 24)]
rres = {i: [] for i in range(len(rcfgs))}
for rep in range(5):
    for i, c in enumerate(rcfgs):
        ms = run_rs(*c)
        if rep:
            rres[i].append(ms)
for i, c in enumerate(rcfgs):
    t = float(np.median(rres[i]))
    nseg = n * n // c[0]
    nseg -= nseg % (c[2] * c[4])
    print(json.dumps({"config": "LDS-DMA ring, segments of %d doubles, 256 workgroups x %d waves, R %d, S %d, side-by-side segments "
                      "%d apart (%d KiB)" % (c[0], c[1], c[2], c[3], c[4], c[4] * c[0] * 8 // 1024),
                      "ms": round(t, 4), "TBps": round(8.0 * nseg * c[0] / t / 1e9, 3)}), flush=True)
