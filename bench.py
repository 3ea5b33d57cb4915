#!/usr/bin/env python3
"""Headline benchmark of the CCQP hot path on MI355X (contract: see the task's bench.py section).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W       (N > 1: starts its own N ranks, launch_ranks_if_asked below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`): n = 131072 dense random-SPD fp64 box-constrained QP, APGD
(tol 1e-8, max 5000 MVs), synthetic inputs generated on the GPU.  One STEP = one complete APGD solve
from x0 = 0 (35 matvecs + all vector work + every host<->device scalar sync).
`value` = matvecs executed per second over the whole job; `ms_per_step` = solve wall-clock.
The solver is the package's default CCQPSolverAPGD: the two matvecs of an iteration (A.x_{k+1} and A.y_{k+1}; the second
does not depend on the line search) are computed by ONE two-vector launch, ccqp_gemv2_f64 -- one pass over the matrix
for both, each product with the bits of its own matvec, the reference's MV counts (35 = 35 against the oracle in this
very line, `cpu_baseline.gpu_vs_oracle`).  Every matvec is computed; what is saved is reading A twice.  `--no-pair` runs
one launch per matvec (the A/B).
For N > 1 the matrix is row-block sharded (one process per GPU, RCCL all-gather of the n-vector per
matvec) and the SAME problem is solved: strong scaling.

What a step launched is in the line itself: `matvec_launches_per_step`, `pair_launches_per_step`,
`products_computed_per_step` and `products_discarded_per_step` (headline: 19 pair launches = 1 start + 16 iterations + 2
rejected line-search trials; 38 products computed, 3 never taken over, 35 executed = the reference's count).
Extra objects on the JSON line:
  roofline     -- dominant kernel (fp64 streaming GEMV; `kernel` names the form this shape runs as, ccqp_gemv_form):
                  algorithmic bytes / duration measured with HIP events on the launch stream inside the timed region,
                  against the 8 TB/s HBM3E peak.  Paired launches (`matvecs_per_launch` = 2): the algorithmic bytes of a
                  launch are A once + two vectors in + two out, `solve_level_frac` prices the launches' bytes over the
                  wall-clock, and `solve_level_frac_at_one_pass_per_matvec` is the figure comparable with unpaired
                  runs (> 1 = passes shared).  Unpaired: a matvec is `launches_per_matvec` launches of the kernel, each over
                  `rows_per_launch` rows (ccqp_gemv_plan; the headline size: ONE persistent launch of the LDS-staged
                  form): `achieved` = bytes of a MATVEC / the events' time around all of its launches (gaps between
                  them included); `bytes_per_launch`, `avg_launch_ms` (= that time / launches) and `traffic` are per
                  KERNEL launch, the unit of the rocprofv3 tables under profiles/.
  cpu_baseline -- the numpy oracle (oracle/cpu_ref.py, a port of the reference; OpenBLAS threads =
                  host cores) solving THE SAME problem (n = 131072, ~60 s of host work) on the GPU box's host, on
                  rank 0, at every N: at N = 1 the device matrix is copied to the host, at N > 1 rank 0 regenerates the
                  whole matrix on its own GPU in 8192-row chunks (the generator is row-addressable, so these are the
                  bits the ranks hold) while the other ranks wait at the final barrier; if the host has less than
                  ~160 GiB of free RAM the sample is n = 65536 and the figure is scaled by bytes (stated in `sample`).
                  A failure inside this leg never loses the bench line: `cpu_baseline` is then null and
                  `cpu_baseline_error` says why.
  box          -- host name, GPU name / unique id of rank 0's device: ties the line to the rocprofv3 tables taken
                  on the same box (profiles/*: `box`).
For N > 1 the roofline object also carries `exchange_ms_per_mv` (HIP events around each vector exchange: one per matvec
LAUNCH -- a pair launch ships both slabs in one collective -- `exchanges_per_step` of them), `vector_work_ms_per_step` (step wall-clock minus matvec and exchange time) and `rccl_ranks`, and the line
an `exchange_ab` object: after the timed steps, untimed, 50 exchanges of the n-vector by each of the three forms
(torch all_gather_into_tensor | the C ABI's in-place RCCL all-gather on the compute stream | zero-padded
all-reduce), HIP-event timed on the launch stream, with a check that all three leave the same bits in y on every
rank -- the line decides by itself which exchange should be the default on the box it ran on.  It runs LAST (after
the CPU baseline), under a watchdog, and a failure or a hang inside it -- it includes a form no multi-rank run has
exercised yet -- puts an `error` into `exchange_ab`, sets the top-level `exchange_ab_ok` to false and ends the run with
the complete line and exit code 0.
Any RCCL error in the measured part ends the run with a non-zero exit code and RCCL's message on stderr: no retry, no
other backend.
"""
import argparse
from datetime import timedelta
import contextlib
import io
import json
import os
import sys
import time


def launch_ranks_if_asked(argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process is only the LAUNCHER.  It
    starts ONE child -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port <free port> bench.py <the same arguments>` -- lets the child's stdout / stderr through, and leaves with
    the child's exit code.  It runs before numpy / torch are imported: the launcher never touches a GPU, never retries and
    has no other backend to fall back to.  Returns only if there is nothing to launch."""
    if "WORLD_SIZE" in os.environ:
        return
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    known, _ = ap.parse_known_args(argv)
    if known.gpus <= 1:
        return
    import signal
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(known.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] launcher: %s" % " ".join(cmd), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env)

    def forward(signum, _frame):                        # a signal for the launcher is a signal for the job
        try:
            child.send_signal(signum)
        except OSError:
            pass

    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, forward)
    rc = child.wait()
    sys.exit(rc if rc >= 0 else 128 - rc)


if __name__ == "__main__":
    launch_ranks_if_asked(sys.argv[1:])

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def mv_bytes(rows, n):
    """Algorithmic bytes of one (rows x n) fp64 matvec: A once + x read + y written."""
    return 8.0 * (rows * n + n + rows)


def pmc_traffic(bytes_per_launch, kernel_prefix="gemv_stream_kernel"):
    """HBM bytes per GEMV launch from a committed rocprofv3 PMC summary of this same launch
    (profiles/*_pmc.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc passes).
    PMC counters cannot be read from inside the timed run, so this is a LOOK-UP, reported as
    `traffic_from_profile`, and only if the profile was taken of the same kernel source: the summary records
    the sha1 of the matvec sources (tools/summarize_prof.py) and a mismatch -- the kernel changed after the
    profile was taken -- yields None."""
    import glob
    cur = gemv_source_hash()
    # newest round first; within a round the profile of the bench itself before other drivers' profiles
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True)
    files.sort(key=lambda f: (os.path.basename(f).split("_")[0], "bench" in os.path.basename(f)), reverse=True)
    for f in files:
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        for k, e in d.items():
            if not isinstance(e, dict):
                continue
            if kernel_prefix in k and e.get("algorithmic_bytes_per_launch") == bytes_per_launch \
                    and "hbm_traffic_bytes_per_launch" in e:
                if e.get("kernel_source_sha1") not in (None, cur):
                    continue                                   # stale: the kernel was edited since
                return e["hbm_traffic_bytes_per_launch"], {
                    "file": os.path.relpath(f, ROOT), "kernel": k,
                    "kernel_source_sha1": e.get("kernel_source_sha1"), "current_source_sha1": cur,
                    "verified_same_source": e.get("kernel_source_sha1") == cur}
    return None, None


def gemv_source_hash():
    """sha1 over the sources the dense matvec kernel is built from."""
    import hashlib
    h = hashlib.sha1()
    for rel in ("ccqppy_amd/csrc/ccqp_gemv.hip", "ccqppy_amd/csrc/ccqp_bodies.h", "ccqppy_amd/csrc/ccqp_internal.h"):
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def host_available_gib():
    try:
        import psutil
        return psutil.virtual_memory().available / 2.0 ** 30
    except Exception:  # noqa: BLE001
        try:
            for ln in open("/proc/meminfo"):
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) / 2.0 ** 20
        except Exception:  # noqa: BLE001
            pass
    return 0.0


def cpu_baseline(holder, b, x_gpu, mv_gpu, n_cpu, seed, tol, max_mv, n_metric, device):
    """Time the oracle on the host cores.  n_cpu == n_metric: the SAME problem as the GPU solved (the matrix reaches
    the host in 8192-row chunks -- slices of the resident device matrix at N = 1, regenerated rows at N > 1 where this
    rank holds only its block; copy time reported separately, never part of the rate); otherwise a smaller sample of
    the same generator / settings, scaled by bytes.  `holder` = [operator]: the ONLY reference to the device matrix;
    it is cleared here as soon as the matrix is not needed any more, so the HBM really is released."""
    from ccqppy_amd import problems
    from ccqppy_amd.operators import DenseOperator
    from oracle import cpu_ref
    try:
        # torch.distributed.run exports OMP_NUM_THREADS=1 to every rank: the baseline is "the host cores", so the BLAS
        # pool is raised to them here (OpenBLAS caps it at the thread count it was built for), then counted
        from threadpoolctl import threadpool_info, threadpool_limits
        threadpool_limits(limits=os.cpu_count(), user_api="blas")
        cores = max([d.get("num_threads", 1) for d in threadpool_info() if d.get("user_api") == "blas"] + [1])
    except Exception:  # noqa: BLE001
        cores = os.cpu_count()
    same = n_cpu == n_metric
    resident = same and isinstance(holder[0], DenseOperator)
    if not same:
        holder[0] = None                                   # free the metric's matrix BEFORE the sample's is built
        torch.cuda.empty_cache()
    step = 8192
    if resident:
        Adev = holder[0].A
        rows_of = lambda r0, cnt: Adev[r0:r0 + cnt]        # noqa: E731
        how = "device matrix copied to the host"
    else:
        if same:
            holder[0] = None                               # a row block of a sharded run: not what the oracle needs
            torch.cuda.empty_cache()
        rows_of = lambda r0, cnt: problems.spd_rows(n_cpu, seed, r0, cnt, device)   # noqa: E731
        how = "matrix regenerated on this rank's GPU in %d-row chunks and copied to the host" % step
    t0 = time.perf_counter()
    A = np.empty((n_cpu, n_cpu), dtype=np.float64)
    for r0 in range(0, n_cpu, step):
        cnt = min(step, n_cpu - r0)
        A[r0:r0 + cnt] = rows_of(r0, cnt).cpu().numpy()
        if (r0 // step) % 4 == 3:
            print("[bench] cpu_baseline: rows %d / %d on the host" % (r0 + cnt, n_cpu), file=sys.stderr, flush=True)
    t_copy = time.perf_counter() - t0
    if same:
        bb = b.cpu().numpy()
    else:
        # b = -A.xstar of the SAMPLE problem, by the host (the sample is only timed, never compared with the GPU)
        from ccqppy_amd.problems import xstar
        bb = -A.dot(xstar(n_cpu, seed, device).cpu().numpy())
    rows_of = Adev = None
    holder[0] = None
    torch.cuda.empty_cache()
    lb, ub = -0.5 * np.ones(n_cpu), 0.5 * np.ones(n_cpu)
    for _ in range(2):
        A[:min(n_cpu, 16384)].dot(bb)                      # warm the BLAS thread pool (threads are created on first use)
    print("[bench] cpu_baseline: oracle solve at n=%d on %d BLAS threads ..." % (n_cpu, cores), file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    r = cpu_ref.solve_apgd(A, bb, tol, max_mv, P=cpu_ref.Box(n_cpu, lb, ub))
    dt = time.perf_counter() - t0
    gbs = r.executed_mv * mv_bytes(n_cpu, n_cpu) / dt / 1e9
    out = {
        "value": gbs * 1e9 / mv_bytes(n_metric, n_metric),
        "unit": "matvecs/s" if same else
                "matvecs/s (n=%d-equivalent: measured solve bytes/s / bytes per n=%d matvec)" % (n_metric, n_metric),
        "cores": int(cores),
        "kind": "port",
        "sample": "oracle/cpu_ref.solve_apgd (numpy+OpenBLAS), %s n=%d box-QP, tol=%g: %d matvecs in %.2f s = "
                  "%.2f matvecs/s at n=%d (%.1f GB/s)" % ("the metric's own problem," if same else "smaller sample,",
                                                         n_cpu, tol, r.executed_mv, dt, r.executed_mv / dt, n_cpu, gbs),
        "n": n_cpu,
        "reported_mv": int(r.solution_num_matrix_vector_multiplications),
        "solve_s": dt,
        "matrix_copy_to_host_s": t_copy,
        "matrix_source": how,
        "host_cpus": os.cpu_count(),
    }
    if same and x_gpu is not None:
        # free parity evidence: the GPU solve of the timed region against the oracle on the same matrix
        out["gpu_vs_oracle"] = {
            "mv_equal": int(r.solution_num_matrix_vector_multiplications) == int(mv_gpu),
            "rel_diff_x": float(np.linalg.norm(x_gpu - r.solution) / np.linalg.norm(r.solution)),
        }
    return out


def box_id(device_index):
    """Identity of the box a number was measured on: host name + the GPU's name and unique id (torch's device
    properties; the KFD topology's `unique_id` when torch does not expose one)."""
    import socket
    out = {"host": socket.gethostname()}
    try:
        pr = torch.cuda.get_device_properties(device_index)
        out["gpu"] = pr.name
        out["arch"] = getattr(pr, "gcnArchName", None)
        uid = getattr(pr, "uuid", None)
        if uid is not None:
            out["gpu_uuid"] = str(uid)
    except Exception:  # noqa: BLE001
        pass
    try:
        import glob
        ids = []
        for f in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
            kv = dict(ln.split(None, 1) for ln in open(f).read().splitlines() if " " in ln)
            if int(kv.get("simd_count", "0")) > 0:
                ids.append("%x" % int(kv.get("unique_id", "0")))
        out["kfd_unique_ids"] = ids
    except Exception:  # noqa: BLE001
        pass
    return out


# watchdog of the untimed exchange A/B (normally a few seconds); the two environment variables are the hooks of
# tests/test_gpu_bench_multirank.py, which rehearses a failing / hanging form
AB_LIMIT_S = int(os.environ.get("CCQP_BENCH_AB_LIMIT_S", "180"))
AB_FAULT = os.environ.get("CCQP_BENCH_AB_FAULT", "")


def exchange_ab(n, row0, rows, slab, eng, group, backend, reps=50, state=None):
    """N > 1, untimed: the three forms of the per-matvec vector exchange, `reps` calls each, HIP events on the launch
    stream (first call of each form untimed: communicator / buffer set-up), and a check that all three leave the same
    bits in y.  The vector exchanged is a real A.x slab (this rank's row block times x*, computed by the caller while
    the matrix was resident); before every form the rest of y is filled with NaN so that an entry a form failed to
    deliver cannot pass.  (Zero-padded all-reduce: x + 0 + ... + 0 is exact except that it turns -0.0 into +0.0; the
    slab of a random matvec holds no zero.)  `state` (a dict) receives the form being run and the results so far, for
    the caller's watchdog: this A/B is informational and must never cost the timed line above it."""
    import torch.distributed as dist
    from ccqppy_amd.operators import VectorExchange
    world = dist.get_world_size(group)
    sl = slice(row0, row0 + rows)
    state = {} if state is None else state
    out, ys = state.setdefault("out", {}), {}
    for mode in ("allgather", "allreduce", "direct"):   # (the experimental form last: the other two are kept if it fails)
        state["mode"] = mode
        if mode == "allreduce" and AB_FAULT == "raise" and dist.get_rank(group) == world - 1:
            raise RuntimeError("injected failure (CCQP_BENCH_AB_FAULT)")
        if mode == "allreduce" and AB_FAULT == "hang":
            time.sleep(10 ** 6)
        if mode == "direct" and backend != "nccl":
            out[mode] = "unavailable: %s" % backend
            continue
        ex = VectorExchange(n, group, mode)
        ex.open(eng)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device=eng.device)
        y[sl] = slab
        ex(y, eng)                                      # set-up call, untimed
        torch.cuda.synchronize()
        dist.barrier(group=group)
        ex.events = []
        for _ in range(reps):
            ex(y, eng)
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b_) for a, b_ in ex.events]
        ex.events = None
        # the timed calls re-exchange an already complete y; the bit check is on a fresh NaN-filled one
        y.fill_(float("nan"))
        y[sl] = slab
        ex(y, eng)
        torch.cuda.synchronize()
        ys[mode] = y.view(torch.int64).clone()
        out[mode] = {"mean_ms": float(np.mean(ms)), "median_ms": float(np.median(ms)), "min_ms": float(np.min(ms)),
                     "max_ms": float(np.max(ms)), "calls": reps, "bytes_moved_per_rank": 8 * (n if mode == "allreduce" else rows),
                     "launches_besides_the_collective": {"allgather": 1, "direct": 0, "allreduce": 2}[mode] if n % world == 0 else None}
        ex.close()
    # the exchange of a PAIR of products (APGD's two matvecs per pass): round 3's two exchanges + two `+ b` launches against
    # round 4's one exchange of both slabs + one unpack launch that forms g = y + b (RowShardedOperator.matvec2)
    state["mode"] = "pair"
    pair = {}
    bvec = torch.ones(n, dtype=torch.float64, device=eng.device)
    for mode in ("allgather", "direct"):
        if mode == "direct" and backend != "nccl":
            continue
        state["mode"] = "pair/" + mode
        ex = VectorExchange(n, group, mode)
        ex.open(eng)
        y1, y2, g1, g2 = (torch.full((n,), float("nan"), dtype=torch.float64, device=eng.device) for _ in range(4))

        def unfused():
            y1[sl] = slab
            y2[sl] = slab
            ex(y1, eng)
            ex(y2, eng)
            eng.axpby(1.0, y1, 1.0, bvec, g1)
            eng.axpby(1.0, y2, 1.0, bvec, g2)

        def fused():
            s1, s2 = ex.slabs(2, y1)
            s1.copy_(slab)
            s2.copy_(slab)
            ex.exchange_many([y1, y2], bvec, [g1, g2], eng)

        res = {}
        for name, fn in (("two_exchanges_two_axpby", unfused), ("one_exchange_one_unpack", fused)):
            fn()
            torch.cuda.synchronize()
            dist.barrier(group=group)
            ms = []
            for _ in range(reps):
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(eng.torch_stream)
                fn()
                b_.record(eng.torch_stream)
                ms.append((a, b_))
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b_) for a, b_ in ms]
            res[name] = {"median_ms": float(np.median(ms)), "min_ms": float(np.min(ms)), "calls": reps}
            if name == "two_exchanges_two_axpby":
                keep = (y1.clone(), g2.clone())
                for t in (y1, y2, g1, g2):
                    t.fill_(float("nan"))
        res["bitwise_equal"] = bool(torch.equal(keep[0].view(torch.int64), y1.view(torch.int64)) and
                                    torch.equal(keep[1].view(torch.int64), g2.view(torch.int64)) and
                                    torch.equal(y1.view(torch.int64), ys["allgather"]))
        pair[mode] = res
        ex.close()
    out["pair_of_products"] = pair
    state["mode"] = "comparison"
    names = list(ys)
    same = all(bool(torch.equal(ys[names[0]], ys[m])) for m in names[1:])
    # the same bits on every rank: xor-fold of the words, compared across ranks on the host
    fold = int(torch.bitwise_xor(ys[names[0]][0::2], ys[names[0]][1::2]).sum().item()) if n % 2 == 0 else \
        int(ys[names[0]].sum().item())
    folds = [None] * world
    dist.all_gather_object(folds, (fold, same), group=group)
    out["bitwise_equal_forms"] = bool(all(f[1] for f in folds))
    out["bitwise_equal_ranks"] = bool(all(f[0] == folds[0][0] for f in folds))
    out["no_nan_left"] = not bool(torch.isnan(ys[names[0]].view(torch.float64)).any().item())
    timed = {m: out[m]["median_ms"] for m in names}
    out["fastest"] = min(timed, key=timed.get)
    out["note"] = "HIP events on the launch stream around each call (rank 0's clock; includes waiting for the slowest rank)"
    return out


def exchange_ab_ok(ab):
    """The top-level verdict on the untimed exchange A/B: every form answered, same bits across forms and ranks, no NaN
    left.  The process still ends with exit code 0 when this is False (the timed line is complete by then) -- so the
    failure has a field of its own that nothing reading the line can miss."""
    return bool(ab) and "error" not in ab and bool(ab.get("bitwise_equal_forms")) and \
        bool(ab.get("bitwise_equal_ranks")) and bool(ab.get("no_nan_left")) and \
        all(v.get("bitwise_equal") for v in ab.get("pair_of_products", {}).values())


def main():
    try:
        _main()
    except BaseException as e:  # noqa: BLE001
        if isinstance(e, SystemExit):
            raise
        # a failed collective (RCCL) or kernel: the message, a non-zero exit code, nothing else -- no second attempt and
        # no other backend in this process; os._exit because a process group with a failed collective may hang in
        # its destructor, and torchrun ends the other ranks when one exits non-zero
        import traceback
        traceback.print_exc()
        print("[bench] rank %s: FATAL %s: %s" % (os.environ.get("RANK", "0"), type(e).__name__, e), file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(1)


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", dest="n", type=int, default=131072, help="problem size n (default: the metric's)")
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--max-mv", type=int, default=5000)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--exchange", dest="mode", default="allgather", choices=["allgather", "allreduce", "direct"],
                    help="N > 1: torch.distributed all_gather_into_tensor (default) | zero-padded all_reduce | 'direct' = the "
                         "in-place RCCL all-gather of the C ABI on the compute stream (ccqp_comm_allgather_f64)")
    ap.add_argument("--variant", type=lambda v: int(v, 0), default=0, help="GEMV tuning variant word")
    ap.add_argument("--no-exchange-ab", action="store_true", help="N > 1: skip the untimed A/B of the three exchanges")
    ap.add_argument("--no-pair", action="store_true",
                    help="one launch (one pass over A) per matvec instead of one per pair of matvecs (A/B; same bits and counts)")
    ap.add_argument("--cpu-n", type=int, default=-1,
                    help="size of the CPU-baseline problem (0 = skip); default: the metric's own n when the host has "
                         ">= 160 GiB of free RAM (~30 s copy + ~60 s oracle solve), else 65536")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # (`python bench.py --gpus N` without a launcher never gets here: launch_ranks_if_asked started the ranks)
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: one rank per GPU, started by bench.py itself or by "
                 "torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    # CCQP_DIST_BACKEND=gloo lets two ranks share one GPU for a rehearsal on a 1-GPU box
    backend = os.environ.get("CCQP_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=timedelta(minutes=30))
        else:
            dist.init_process_group(backend, timeout=timedelta(minutes=30))
        group = dist.group.WORLD

    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine

    eng = get_engine(local_rank)
    eng.gemv_variant = args.variant
    n = args.n
    t_gen = time.perf_counter()
    op, b, _ = problems.random_problem(n, args.seed, device=local_rank, group=group, mode=args.mode)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    P = ss.BoxProjOp(n, -0.5, 0.5)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step():
        np.random.seed(args.seed)
        s = solvers.CCQPSolverAPGD(args.tol, args.max_mv, pair_matvecs=False if args.no_pair else None)
        with contextlib.redirect_stdout(io.StringIO()):
            s.solve(op, b, convex_proj_op=P)
        return s

    # Do the two matvecs of an APGD iteration share ONE pass over the matrix?  (CCQPSolverAPGD's default wherever the
    # dense matvec is the LDS-staged form: A.x_{k+1} and A.y_{k+1} from one ccqp_gemv2_f64 launch, each product with the
    # bits of its own matvec, same MV counts; --no-pair: one launch per matvec, the A/B.)
    paired = (not args.no_pair) and hasattr(op, "pair_in_one_pass") and op.pair_in_one_pass(eng)

    for _ in range(args.warmup):
        step()
    barrier()
    eng.gemv_events = []
    exch = getattr(op, "exchange", None)
    if exch is not None:
        exch.events = []
        exch.calls = 0
    mv0 = eng.mv_executed
    prod0 = eng.products_launched
    t0 = time.perf_counter()
    last = None
    step_end = []                                       # host clock after each step's solve() returned (its last scalar
    for _ in range(args.steps):                         # read is blocking, so the step's device work is done by then)
        last = step()
        step_end.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    mv_exec = eng.mv_executed - mv0
    products = eng.products_launched - prod0            # every A.v a launch of the timed region computed, taken over or not
    events = eng.gemv_events
    eng.gemv_events = None
    gemv_ms = [a.elapsed_time(b_) for a, b_ in events]
    exch_ms = []
    if exch is not None:
        exch_ms = [a.elapsed_time(b_) for a, b_ in exch.events]
        exch.events = None
    rccl_ranks = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
        # ranks that actually took part in a device collective (one all_gather_into_tensor of the rank ids)
        ids = torch.full((world,), -1, dtype=torch.int64, device="cuda")
        me = torch.tensor([rank], dtype=torch.int64, device="cuda")
        if backend == "nccl":
            torch.distributed.all_gather_into_tensor(ids, me)
        else:
            parts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            torch.distributed.all_gather(parts, me.cpu())
            ids = torch.cat(parts)
        rccl_ranks = int((ids.cpu() == torch.arange(world)).sum()) if backend == "nccl" else 0

    # outside the timed region, informational: the opt-in NON-parity APGD mode that obtains A.y_{k+1} from
    # A.x_{k+1} and A.x_k (one matvec per iteration instead of two) -- never part of `value`
    torch.cuda.synchronize()
    t_fast = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        fast = solvers.CCQPSolverAPGD(args.tol, args.max_mv, reuse_matvec=True).solve(op, b, convex_proj_op=P)
    torch.cuda.synchronize()
    t_fast = time.perf_counter() - t_fast
    # second informational extension (1 GPU only): the same solve on operators.SymmetricOperator, which
    # reads only the upper triangle of the (symmetric) Hessian -- 4n^2 bytes per matvec instead of 8n^2
    sym_info = None
    if world == 1 and n % 128 == 0:
        from ccqppy_amd.operators import SymmetricOperator
        try:
            sop = SymmetricOperator(op.A)
            with contextlib.redirect_stdout(io.StringIO()):
                solvers.CCQPSolverAPGD(args.tol, args.max_mv, reuse_matvec=True).solve(sop, b, convex_proj_op=P)
                torch.cuda.synchronize()
                t_sym = time.perf_counter()
                sres = solvers.CCQPSolverAPGD(args.tol, args.max_mv, reuse_matvec=True).solve(sop, b, convex_proj_op=P)
                torch.cuda.synchronize()
                t_sym = time.perf_counter() - t_sym
            sym_info = {"note": "opt-in non-parity: SymmetricOperator (upper-triangle matvec) + reuse_matvec",
                        "wall_s": t_sym, "executed_mv": int(sres.executed_matrix_vector_multiplications),
                        "converged": bool(sres.solution_converged), "residual": float(sres.solution_residual)}
            del sop
        except torch.cuda.OutOfMemoryError:
            sym_info = {"note": "skipped: no room for the n^2/32 workspace"}

    ab_in = None
    if world > 1 and not args.no_exchange_ab:
        # the vector the untimed exchange A/B moves: this rank's slab of A.x*, computed while the matrix is resident
        from ccqppy_amd.problems import xstar
        slab = eng.empty(op.rows)
        eng.gemv(op.A, xstar(n, 99, eng.device), slab)
        torch.cuda.synchronize()
        ab_in = (n, op.row0, op.rows, slab)

    rows_local = op.rows
    holder = [op]
    del op                      # from here on `holder` owns the matrix (cpu_baseline releases it)
    if rank == 0:
        avg_ms = float(np.mean(gemv_ms)) if gemv_ms else float("nan")
        steps = max(args.steps, 1)
        rows_per_launch, launches = eng.gemv_plan(rows_local, n)
        form = eng.gemv_form(rows_local, n)              # the kernel this shape runs as: its name in a rocprofv3 table
        per_launch = 1
        launch_bytes = mv_bytes(rows_per_launch, n)
        if paired:
            # one launch = TWO products in one pass: A once, two vectors in, two out -- the algorithmic bytes of the pair
            form, per_launch, launches = eng.gemv2_form(rows_local, n), 2, 1
            rows_per_launch = rows_local
            launch_bytes = 8.0 * (rows_local * n + 2 * n + 2 * rows_local)
        achieved = launch_bytes * launches / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(launch_bytes, form)
        line = {
            "metric": "A.x matvecs/sec (+ solve wall-clock), n=%d random-SPD box-CCQP, APGD" % n,
            "value": mv_exec / elapsed,
            "unit": "matvecs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "step_ms": [round(1e3 * (b_ - a), 3) for a, b_ in zip([t0] + step_end[:-1], step_end)],
            # what a step launched: matvec launches (paired: two products each), the products they computed, and those
            # of them the solver never took over (the product computed ahead of a REJECTED line-search trial; they cost
            # no extra pass over A).  `value` counts the reference's executed matvecs: computed - discarded.
            "matvec_launches_per_step": len(gemv_ms) / steps,
            "pair_launches_per_step": len(gemv_ms) / steps if paired else 0,
            "products_computed_per_step": products / steps,
            "products_discarded_per_step": (products - mv_exec) / steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE metric: n=%d dense random-SPD fp64 + BoxProjOp(+-0.5), CCQPSolverAPGD(tol=%g, "
                            "max_mv=%d); 1 step = 1 full solve from x0=0" % (n, args.tol, args.max_mv),
                "n": n, "solver": "APGD", "projection": "Box", "tol": args.tol,
                "sharding": "none" if world == 1 else "row-block x%d, %s %s of the n-vector per matvec" % (
                    world, "RCCL" if backend == "nccl" else backend, args.mode),
                "matrix_bytes_per_gpu": 8 * rows_local * n,
            },
            "solve": {
                "wall_s": elapsed / max(args.steps, 1),
                "reported_mv": int(last.solution_num_matrix_vector_multiplications),
                "executed_mv": int(last.executed_matrix_vector_multiplications),
                "converged": bool(last.solution_converged),
                "residual": float(last.solution_residual),
                "generate_s": t_gen,
            },
            "extension_reuse_matvec": {
                "note": "opt-in non-parity mode CCQPSolverAPGD(reuse_matvec=True), measured outside the timed region",
                "wall_s": t_fast, "executed_mv": int(fast.executed_matrix_vector_multiplications),
                "converged": bool(fast.solution_converged), "residual": float(fast.solution_residual),
            },
            "extension_symmetric": sym_info,
            "roofline": {
                "bound": "hbm",
                "kernel": "%s (%s%s)" % (
                    form, "ccqp_gemv2_f64: the two matvecs of an APGD iteration in ONE pass over A" if paired
                    else "ccqp_gemv_f64 / ccqp_gemv_epilogue_f64",
                    "; rows through per-wave LDS rings by LDS-DMA loads, one persistent launch"
                    if form in ("gemv_lds_kernel", "gemv_lds2_kernel") else ""),
                "matvecs_per_launch": per_launch,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic if (traffic_src or {}).get("verified_same_source") else None,
                "traffic_from_profile": traffic,
                "traffic_source": traffic_src,
                "bytes_per_launch": launch_bytes,
                "avg_launch_ms": avg_ms / launches,
                "launches_per_matvec": launches,
                "rows_per_launch": rows_per_launch,
                "bytes_per_matvec": mv_bytes(rows_local, n),
                "avg_matvec_ms": avg_ms / per_launch,       # (a paired launch delivers two products)
                "matvecs_timed": len(gemv_ms) * per_launch,
                # the bytes the solve's matvec launches have to read, over the whole wall-clock of the timed region
                "solve_level_frac": len(gemv_ms) * launch_bytes * launches / elapsed / 1e9 / HBM_PEAK_GBS,
                # for comparison with unpaired runs: every matvec priced at a full pass over A (> 1 means passes shared)
                "solve_level_frac_at_one_pass_per_matvec": mv_exec * mv_bytes(rows_local, n) / elapsed / 1e9 / HBM_PEAK_GBS,
                "launches_timed": len(gemv_ms),
                "gemv_ms_per_step": float(np.sum(gemv_ms)) / steps if gemv_ms else None,
                # N > 1: where the rest of a step goes (rank 0's own clock; the exchange includes waiting for the
                # slowest rank's slab)
                "exchange_ms_per_mv": float(np.mean(exch_ms)) if exch_ms else None,
                "exchange_ms_per_step": float(np.sum(exch_ms)) / steps if exch_ms else None,
                # (one exchange per matvec LAUNCH: a pair launch ships both slabs in one collective + one unpack launch)
                "exchanges_per_step": len(exch_ms) / steps if exch_ms else None,
                "vector_work_ms_per_step": (1e3 * elapsed - float(np.sum(gemv_ms)) - float(np.sum(exch_ms))) / steps
                                           if gemv_ms else None,
                "rccl_ranks": rccl_ranks,
                "exchange_backend": None if exch is None else exch.backend,
            },
        }
        line["box"] = box_id(local_rank)
        # tuning options of the engine that are not at their default (bench.py sets none: {} -- a measurement says so itself)
        line["box"]["options"] = eng.options()
        line["box"]["gemv_variant"] = hex(args.variant)
        cpu_n = args.cpu_n
        if cpu_n < 0:
            need = 8.0 * n * n / 2.0 ** 30 + 24.0
            cpu_n = n if host_available_gib() >= need else min(n, 65536)
        line["cpu_baseline"] = None
        if cpu_n > 0:
            x_gpu = last.solution.cpu().numpy() if isinstance(last.solution, torch.Tensor) else np.asarray(last.solution)
            try:
                line["cpu_baseline"] = cpu_baseline(holder, b, x_gpu, last.solution_num_matrix_vector_multiplications,
                                                    cpu_n, args.seed, args.tol, args.max_mv, n, local_rank)
            except Exception as e:  # noqa: BLE001  (the timed region succeeded: keep the line)
                import traceback
                traceback.print_exc()
                line["cpu_baseline_error"] = "%s: %s" % (type(e).__name__, e)
    else:
        line = None
        holder[0] = None
    ab_failed = False
    if ab_in is not None:
        # LAST, after the line is complete (ranks > 0 wait here while rank 0 times the oracle): the A/B of the three
        # exchange forms is informational and includes a form no multi-rank run has exercised yet ("direct"), so a
        # failure or a hang in it costs the `exchange_ab` object, never the measured line -- rank 0 prints the line
        # with the error in it and every rank leaves without another collective
        import threading
        barrier()
        state = {"out": {}, "mode": None}

        def give_up():
            if line is not None:
                line["exchange_ab"] = dict(state["out"], error="no answer within %d s in form %r (untimed A/B; the timed "
                                           "region above is complete)" % (AB_LIMIT_S, state["mode"]))
                line["exchange_ab_ok"] = False
                print(json.dumps(line), flush=True)
            print("[bench] rank %d: exchange A/B gave no answer in form %r; leaving" % (rank, state["mode"]),
                  file=sys.stderr, flush=True)
            os._exit(0)

        watchdog = threading.Timer(AB_LIMIT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            ab = exchange_ab(*ab_in, eng, group, backend, state=state)
            barrier()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            ab = dict(state["out"], error="%s in form %r: %s" % (type(e).__name__, state["mode"], e))
            ab_failed = True
        watchdog.cancel()
        if line is not None:
            line["exchange_ab"] = ab
            line["exchange_ab_ok"] = exchange_ab_ok(ab)
    if line is not None:
        print(json.dumps(line), flush=True)
    if ab_failed:
        sys.stdout.flush()
        os._exit(0)                     # (the other ranks may sit in the failed form's collective: their watchdogs end them)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
