"""bench.py at N > 1, rehearsed inside the GPU suite: 2 gloo ranks on ONE device (RCCL refuses two ranks on one GPU,
profiles/r02_rccl_two_ranks_one_gpu_probe.txt), a small matrix.  The line must be complete (cpu_baseline on rank 0,
exchange_ab with the bitwise checks), and a failing or hanging form of the untimed exchange A/B must cost the
`exchange_ab` object only -- never the measured line or the exit code."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(port, extra_env, timeout=420, size=4096, own_launcher=False):
    env = dict(os.environ, CCQP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("OMP_NUM_THREADS", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--size", str(size),
            "--cpu-n", str(size)]
    if own_launcher:
        cmd = [sys.executable] + args           # the driver's form of the command: bench.py starts its own two ranks
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    return p, lines


def _check_measured_part(d, size=4096):
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["unit"] == "matvecs/s" and d["value"] > 0
    assert d["scaling"] == "strong" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["solve"]["converged"] and d["solve"]["reported_mv"] == d["solve"]["executed_mv"] > 5
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["exchange_ms_per_mv"] > 0
    c = d["cpu_baseline"]
    assert c is not None and c["kind"] == "port" and c["n"] == size and c["value"] > 0 and c["cores"] >= 1
    # the same problem on both sides: the oracle's count and iterate against the 2-rank GPU solve
    assert c["reported_mv"] == d["solve"]["reported_mv"]
    assert c["gpu_vs_oracle"]["mv_equal"] and c["gpu_vs_oracle"]["rel_diff_x"] < 1e-10
    assert d["box"]["gpu_uuid"]


def test_two_rank_line_is_complete():
    """`python bench.py --gpus 2 ...` by itself (no torch.distributed.run in the command): the launcher process starts
    the two ranks, passes their output through and leaves with their exit code.  n = 16384: each rank's 8192 x 16384 block
    is big enough for the two matvecs of an APGD iteration to share one pass (ccqp_gemv2_f64) and one exchange."""
    p, lines = _run(0, {}, size=16384, own_launcher=True)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "[bench] launcher:" in p.stderr and "torch.distributed.run" in p.stderr
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    _check_measured_part(d, 16384)
    assert d["exchange_ab_ok"] is True
    assert d["roofline"]["matvecs_per_launch"] == 2 and d["pair_launches_per_step"] == d["matvec_launches_per_step"] > 5
    assert d["products_computed_per_step"] == 2 * d["pair_launches_per_step"]
    assert d["products_computed_per_step"] - d["products_discarded_per_step"] == d["solve"]["executed_mv"]
    ab = d["exchange_ab"]
    assert "error" not in ab
    assert ab["direct"] == "unavailable: gloo"
    for form in ("allgather", "allreduce"):
        assert ab[form]["calls"] == 50 and ab[form]["median_ms"] > 0
    assert ab["bitwise_equal_forms"] and ab["bitwise_equal_ranks"] and ab["no_nan_left"]
    # the exchange of a pair of products: one collective + one unpack launch against two + two, same bits
    pp = ab["pair_of_products"]["allgather"]
    assert pp["bitwise_equal"] and pp["one_exchange_one_unpack"]["median_ms"] > 0 and pp["two_exchanges_two_axpby"]["calls"] == 50
    # and the timed solve itself exchanged once per PAIR launch (exchange events = matvec launches)
    assert d["roofline"]["exchange_ms_per_mv"] > 0 and d["roofline"]["exchanges_per_step"] == d["matvec_launches_per_step"]


@pytest.mark.parametrize("fault", ["raise", "hang"])
def test_a_failing_exchange_form_costs_only_the_ab_object(fault):
    p, lines = _run(29613 if fault == "raise" else 29615, {"CCQP_BENCH_AB_FAULT": fault, "CCQP_BENCH_AB_LIMIT_S": "20"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    _check_measured_part(d)
    ab = d["exchange_ab"]
    assert d["exchange_ab_ok"] is False
    assert "error" in ab and "allreduce" in ab["error"]
    assert ab["allgather"]["calls"] == 50          # the form that ran before the failing one is kept
