#!/usr/bin/env python3
"""One BASELINE config per call, as a program that can stand directly behind `rocprofv3 ... --` and that prints a
JSON line with a `roofline` object measured LIVE (HIP events on the launch stream around every matvec launch):

    python3 tools/config_profile_driver.py 2      # n = 32768  APGD  + Box          (1 solve per repetition)
    python3 tools/config_profile_driver.py 3      # n = 131072 BBPGD + Cone, 64 MVs (1 solve per repetition)
    python3 tools/config_profile_driver.py 5      # 128 x n = 8192 MPRGP + Box through solve_batch_pipelined
                                                  # (the 1-GPU share of the 1024-problem batch)
    options: --reps R (timed repetitions, default 2)  --warm W (default 1)  --ahead K (config 5: rounds queued ahead)
             --cpu  also time the oracle (oracle/cpu_ref.py: the reference's algorithm on numpy + OpenBLAS) on the host
                    cores beside it -> a `cpu_baseline` object: config 2 on the SAME problem (and x compared), config 3
                    on a bounded sample (n = 32768, same generator / solver / set; the full size is compared in
                    tests/test_gpu_baseline_configs.py), config 5 on the first 8 problems of the batch one after the
                    other (the reference has no batch form) with x compared.  Not for runs under rocprofv3.

roofline.achieved = algorithmic bytes of all matvecs executed in the timed repetitions / the summed event time of
their launches (configs 2, 3: Engine.gemv_events, one pair per matvec = `launches_per_matvec` kernel launches;
config 5: the C ABI's ccqp_mv_timing_* hook, one pair per batched launch -- launches of rounds in which no problem
multiplies are part of the sum, so the figure is the conservative one); `full_launch` is the same for the launches
that carry the whole batch.  `traffic` is looked up in a committed PMC summary of the same kernel and reported only
when that summary was taken of the same sources (sha1)."""
import argparse
import contextlib
import ctypes
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._opts import add_option_args, apply_options  # noqa: E402

SOURCES = {
    "dense": ("ccqppy_amd/csrc/ccqp_gemv.hip", "ccqppy_amd/csrc/ccqp_bodies.h", "ccqppy_amd/csrc/ccqp_internal.h"),
    "bmprgp": ("ccqppy_amd/csrc/ccqp_bmprgp.hip", "ccqppy_amd/csrc/ccqp_lockstep.h", "ccqppy_amd/csrc/ccqp_bodies.h",
               "ccqppy_amd/csrc/ccqp_internal.h"),
}


def mv_bytes(rows, n):
    return 8.0 * (rows * n + n + rows)


def source_hash(which):
    import hashlib
    h = hashlib.sha1()
    for rel in SOURCES[which]:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def traffic_lookup(kernel_prefix, bytes_per_launch, which, tag_hint):
    """(traffic or None, source record) from the newest profiles/*_pmc.json that holds this kernel at this size."""
    import glob
    cur = source_hash(which)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True)
    files.sort(key=lambda f: (os.path.basename(f).split("_")[0], tag_hint in os.path.basename(f)), reverse=True)
    for f in files:
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        for k, e in d.items():
            if not isinstance(e, dict) or kernel_prefix not in k or "hbm_traffic_bytes_per_launch" not in e:
                continue
            if e.get("algorithmic_bytes_per_launch") != bytes_per_launch:
                continue
            same = e.get("kernel_source_sha1") == cur
            return (e["hbm_traffic_bytes_per_launch"] if same else None), {
                "file": os.path.relpath(f, ROOT), "kernel": k, "kernel_source_sha1": e.get("kernel_source_sha1"),
                "current_source_sha1": cur, "verified_same_source": same,
                "traffic_from_profile": e["hbm_traffic_bytes_per_launch"], "box": e.get("box")}
    return None, None


def blas_threads():
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threadpool_limits(limits=os.cpu_count(), user_api="blas")
        return int(max([d.get("num_threads", 1) for d in threadpool_info() if d.get("user_api") == "blas"] + [1]))
    except Exception:  # noqa: BLE001
        return int(os.cpu_count())


def to_host(A, step=8192):
    out = np.empty(tuple(A.shape), dtype=np.float64)
    for r0 in range(0, A.shape[0], step):
        out[r0:r0 + step] = A[r0:r0 + step].cpu().numpy()
    return out


def cpu_dense(which, op, b, x_gpu, gpu_mv_per_s, n):
    """The oracle beside configs 2 / 3 (tools may call the oracle; the product never does)."""
    from ccqppy_amd import problems
    from oracle import cpu_ref
    cores = blas_threads()
    if which == "2":
        A, bb, same, ns = to_host(op.A), b.cpu().numpy(), True, n
        run = lambda: cpu_ref.solve_apgd(A, bb, 1e-8, 5000, P=cpu_ref.Box(ns, -0.5, 0.5))       # noqa: E731
        what = "oracle/cpu_ref.solve_apgd, the SAME n=32768 problem"
    else:
        ns, same = 32768, False
        op2, b2, _ = problems.random_problem(ns, 1234)
        A, bb = to_host(op2.A), b2.cpu().numpy()
        del op2
        torch.cuda.empty_cache()
        run = lambda: cpu_ref.solve_bbpgd(A, bb, 1e-8, 64, P=cpu_ref.Cone(ns))                  # noqa: E731
        what = "oracle/cpu_ref.solve_bbpgd + Cone, 64-MV budget, bounded sample n=32768 of the same generator"
    A[:8192].dot(bb)
    t0 = time.perf_counter()
    r = run()
    dt = time.perf_counter() - t0
    gbs = r.executed_mv * mv_bytes(ns, ns) / dt / 1e9
    out = {"value": gbs * 1e9 / mv_bytes(n, n), "unit": "matvecs/s at n=%d%s" % (n, "" if same else " (bytes-equivalent)"),
           "cores": cores, "kind": "port", "solve_s": dt, "executed_mv": int(r.executed_mv),
           "reported_mv": int(r.solution_num_matrix_vector_multiplications), "GBps": gbs,
           "sample": "%s: %d matvecs in %.2f s on %d BLAS threads" % (what, r.executed_mv, dt, cores),
           "gpu_over_cpu": gpu_mv_per_s / (gbs * 1e9 / mv_bytes(n, n))}
    if same and x_gpu is not None:
        xc = r.solution
        out["gpu_vs_oracle"] = {"rel_dx": float(np.linalg.norm(x_gpu - xc) / max(np.linalg.norm(xc), 1e-300))}
    return out


def dense_config(which, args):
    from bench import box_id, HBM_PEAK_GBS
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    if which == "2":
        n, label = 32768, "config 2: n=32768 dense random-SPD fp64 + BoxProjOp(+-0.5), CCQPSolverAPGD(1e-8, 5000)"
        make, P = (lambda: solvers.CCQPSolverAPGD(1e-8, 5000)), ss.BoxProjOp(n, -0.5, 0.5)
    else:
        n, label = 131072, "config 3: n=131072 dense random-SPD fp64 + ConeProjOp, CCQPSolverBBPGD(1e-8, max_mv=64)"
        make, P = (lambda: solvers.CCQPSolverBBPGD(1e-8, 64)), ss.ConeProjOp(n)
    op, b, _ = problems.random_problem(n, 1234)

    def solve():
        np.random.seed(1234)
        with contextlib.redirect_stdout(io.StringIO()):
            return make().solve(op, b, convex_proj_op=P)

    for _ in range(args.warm):
        solve()
    torch.cuda.synchronize()
    # wall-clock first, WITHOUT the event pairs (creating and recording two timing events per matvec costs the host
    # loop ~0.2 ms per matvec -- a third of a solve at n = 32768), then the same repetitions with them for the kernel
    mv0 = eng.mv_executed
    t0 = time.perf_counter()
    for _ in range(args.reps):
        r = solve()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mv = eng.mv_executed - mv0
    eng.gemv_events = []
    t0 = time.perf_counter()
    for _ in range(args.reps):
        solve()
    torch.cuda.synchronize()
    dt_events = time.perf_counter() - t0
    ms = [a.elapsed_time(b_) for a, b_ in eng.gemv_events]
    eng.gemv_events = None
    rpl, launches = eng.gemv_plan(n, n)
    form = eng.gemv_form(n, n)
    # APGD (config 2) computes the two matvecs of an iteration with ONE two-vector launch wherever the dense matvec is
    # the LDS-staged form (CCQPSolverAPGD's default: ccqp_gemv2_f64, one pass over A, same bits and counts)
    paired = which == "2" and op.pair_in_one_pass(eng)
    per_launch, launch_bytes = 1, mv_bytes(rpl, n)
    if paired:
        form, per_launch, launches, rpl = eng.gemv2_form(n, n), 2, 1, n
        launch_bytes = 8.0 * (n * n + 4 * n)
    achieved = launch_bytes * launches / (np.mean(ms) * 1e-3) / 1e9
    traffic, src = traffic_lookup(form, launch_bytes, "dense", "config" + which)
    cpu = None
    if args.cpu:
        xg = r.solution
        xg = xg.cpu().numpy() if isinstance(xg, torch.Tensor) else np.asarray(xg)
        if which == "3":
            del op
            torch.cuda.empty_cache()
            op = None
        cpu = cpu_dense(which, op, b, xg, mv / dt, n)
        if cpu.get("reported_mv") is not None and which == "2":
            cpu["gpu_vs_oracle"]["reported_mv"] = [int(r.solution_num_matrix_vector_multiplications), cpu["reported_mv"]]
    return {
        "cpu_baseline": cpu,
        "config": label, "n": n, "repetitions": args.reps, "wall_s_per_solve": dt / args.reps,
        "executed_mv_per_solve": mv // args.reps, "reported_mv": int(r.solution_num_matrix_vector_multiplications),
        "converged": bool(r.solution_converged), "residual": float(r.solution_residual),
        "mv_per_s": mv / dt,
        "roofline": {
            "bound": "hbm", "kernel": "%s (%s)" % (form, "ccqp_gemv2_f64: two matvecs per pass over A" if paired
                                                   else "ccqp_gemv_f64 / ccqp_gemv_epilogue_f64"),
            "matvecs_per_launch": per_launch,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src,
            "bytes_per_matvec": mv_bytes(n, n), "avg_matvec_ms": float(np.mean(ms)) / per_launch,
            "matvecs_timed": len(ms) * per_launch,
            "launches_per_matvec": launches, "rows_per_launch": rpl, "bytes_per_launch": launch_bytes,
            "avg_launch_ms": float(np.mean(ms)) / launches,
            # the bytes the matvec launches have to read over the wall-clock (events and wall-clock are taken in
            # separate repetitions of the same solve: same number of launches)
            "solve_level_frac": (len(ms) / args.reps) * launch_bytes * launches / (dt / args.reps) / 1e9 / HBM_PEAK_GBS,
            "solve_level_frac_at_one_pass_per_matvec": mv * mv_bytes(n, n) / dt / 1e9 / HBM_PEAK_GBS,
            "gemv_ms_per_solve": float(np.sum(ms)) / args.reps,
            "other_ms_per_solve": 1e3 * dt / args.reps - float(np.sum(ms)) / args.reps,
            "wall_s_per_solve_with_event_pairs": dt_events / args.reps,
        },
        "kernel_source_sha1": source_hash("dense"), "box": box_id(0),
    }


def config5(args):
    from bench import box_id, HBM_PEAK_GBS
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    lib = eng.lib
    n, B = args.size, args.problems
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    probs = []
    for p in range(B):
        op, b, _ = problems.random_problem(n, 1234 + p, device=0)
        probs.append((op, b, None, ss.BoxProjOp(n, lb, ub)))
    make = lambda: solvers.CCQPSolverMPRGP(1e-6, 300)       # noqa: E731
    fb0 = lockstep.stats["fallback_solves"]
    halved0 = lockstep.stats["halved_runs"]
    for _ in range(args.warm + (1 if args.warm else 0)):     # (the first call also creates the half-batch lanes)
        lockstep.solve_batch_pipelined(make, probs, device=0, ahead=args.ahead)
    torch.cuda.synchronize()
    # untimed-hook repetitions first (the wall-clock figure), then ONE repetition with the event pairs
    best, timing = None, {}
    for _ in range(args.reps + (2 if args.warm else 0)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tm = {}
        res = lockstep.solve_batch_pipelined(make, probs, device=0, ahead=args.ahead, timing=tm)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, timing = dt, tm
    cap = 4096
    lib.ccqp_mv_timing_begin(cap)
    t0 = time.perf_counter()
    res = lockstep.solve_batch_pipelined(make, probs, device=0, ahead=args.ahead)
    torch.cuda.synchronize()
    dt_hook = time.perf_counter() - t0
    buf = (ctypes.c_double * cap)()
    cnt = ctypes.c_int()
    lib.ccqp_mv_timing_end(buf, cap, ctypes.byref(cnt))
    ms = np.array(buf[:min(cnt.value, cap)])
    mv = sum(r.executed_matrix_vector_multiplications for r in res)
    total_bytes = mv * mv_bytes(n, n)                        # every matvec priced at a full pass over its matrix
    # The run LOOKS AHEAD (ccqp_bmprgp.hip): a launch multiplies up to three vectors per matrix in one pass, and the
    # rounds whose products were computed ahead launch nothing worth timing (~0.01 ms).  The launches that carry a whole
    # (half-)batch: at least half the longest.  A large batch advances as two alternating halves (lockstep._bmprgp_run):
    # a launch then carries half the problems.
    full = ms[ms >= 0.5 * ms.max()]
    per_launch = (B + 1) // 2 if lockstep.stats["halved_runs"] > halved0 else B
    full_bytes = per_launch * mv_bytes(n, n)                 # (A once per matrix; the vectors are 0.05 % of it)
    read_bytes = len(full) * full_bytes                      # what the matvec launches of a solve have to read
    achieved = read_bytes / (ms.sum() * 1e-3) / 1e9
    traffic, src = traffic_lookup("bp_gemv3_pers", full_bytes, "bmprgp", "config5")
    cpu = None
    if args.cpu:
        from oracle import cpu_ref
        cores = blas_threads()
        k = min(8, B)
        t_cpu, mv_cpu, worst, same_mv = 0.0, 0, 0.0, True
        for p in range(k):
            A, bb = to_host(probs[p][0].A), probs[p][1].cpu().numpy()
            if p == 0:
                A.dot(bb)
            t0 = time.perf_counter()
            rc = cpu_ref.solve_mprgp(A, bb, 1e-6, 300, P=cpu_ref.Box(n, lb, ub))
            t_cpu += time.perf_counter() - t0
            mv_cpu += rc.executed_mv
            xg = res[p].solution
            xg = xg.cpu().numpy() if isinstance(xg, torch.Tensor) else np.asarray(xg)
            worst = max(worst, float(np.linalg.norm(xg - rc.solution) / max(np.linalg.norm(rc.solution), 1e-300)))
            same_mv = same_mv and int(rc.solution_num_matrix_vector_multiplications) == \
                int(res[p].solution_num_matrix_vector_multiplications)
        cpu = {"value": k / t_cpu, "unit": "problems/s", "cores": cores, "kind": "port", "problems": k, "solve_s": t_cpu,
               "executed_mv": int(mv_cpu), "GBps": mv_cpu * mv_bytes(n, n) / t_cpu / 1e9,
               "sample": "oracle/cpu_ref.solve_mprgp on the first %d problems of the batch, one after the other (the "
                         "reference has no batch form), %d BLAS threads: %.2f s" % (k, cores, t_cpu),
               "gpu_over_cpu": (B / best) / (k / t_cpu),
               "gpu_vs_oracle": {"worst_rel_dx": worst, "same_reported_mv": bool(same_mv)}}
    return {
        "cpu_baseline": cpu,
        "config": "config 5 (1-GPU share): %d x (n=%d dense random-SPD fp64 + BoxProjOp(array bounds +-0.5), "
                  "CCQPSolverMPRGP(1e-6, 300)), solve_batch_pipelined (lock-step rounds, ahead=%d)" % (B, n, args.ahead),
        "n": n, "problems": B, "wall_s": best, "problems_per_s": B / best, "executed_mv": int(mv),
        "mv_per_s": mv / best, "converged": int(sum(bool(r.solution_converged) for r in res)),
        "fallback_solves": lockstep.stats["fallback_solves"] - fb0,
        "reported_mv_first8": [int(r.solution_num_matrix_vector_multiplications) for r in res[:8]],
        "host_split": timing, "wall_s_with_event_pairs": dt_hook,
        "roofline": {
            "bound": "hbm", "kernel": "bp_gemv3_pers_kernel (the A.dx launch of an iteration with A.x_{k+1} + b and A.p_{k+1} in "
                                      "the same pass) and bp_gemv_pers_kernel (ccqp_bmprgp_round_f64 / _start_f64; "
                                      "persistent LDS-DMA forms, rows about 1 MiB apart)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src,
            "launches_timed": int(len(ms)), "launch_ms_sum": float(ms.sum()), "launch_ms_max": float(ms.max()),
            "bytes_all_matvecs_at_one_pass_each": total_bytes, "bytes_read_by_the_launches": read_bytes,
            "passes_per_problem": len(full) * per_launch / B, "executed_matvecs_per_problem": mv / B,
            "full_launch": {"launches": int(len(full)), "problems_per_launch": int(per_launch), "bytes_per_launch": full_bytes,
                            "avg_launch_ms": float(full.mean()),
                            "achieved": full_bytes / (full.mean() * 1e-3) / 1e9,
                            "frac": full_bytes / (full.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "solve_level_frac": read_bytes / best / 1e9 / HBM_PEAK_GBS,
            "solve_level_frac_at_one_pass_per_matvec": total_bytes / best / 1e9 / HBM_PEAK_GBS,
            "not_matvec_ms": 1e3 * best - float(ms.sum()),
        },
        "kernel_source_sha1": source_hash("bmprgp"), "box": box_id(0),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["2", "3", "5"])
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--warm", type=int, default=1)
    ap.add_argument("--ahead", type=int, default=4)
    ap.add_argument("--problems", type=int, default=128)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--cpu", action="store_true")
    add_option_args(ap)
    args = ap.parse_args()
    options = apply_options(args)                  # (per-context tuning options: the A/B switches, tools/README.md)
    torch.cuda.set_device(0)
    line = config5(args) if args.which == "5" else dense_config(args.which, args)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
