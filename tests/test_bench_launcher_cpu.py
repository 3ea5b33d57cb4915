"""bench.py as its own launcher (`python bench.py --gpus N`, the form of the driver's command), on a box without a GPU:
the parent must start ONE torch.distributed.run child with N ranks, never touch the GPU itself, and pass the child's exit
code on.  Without a GPU the ranks end in bench.py's FATAL path -- which is exactly what shows that ranks were started
with their own RANK and that the code came back through the launcher."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""             # (a GPU box runs this test too: no device for the ranks either way)
    env["CUDA_VISIBLE_DEVICES"] = ""
    return env


def test_gpus_2_starts_two_ranks_and_passes_their_exit_code_on():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--size", "1024", "--cpu-n", "0"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert "must be launched with" not in p.stderr
    assert "[bench] launcher:" in p.stderr and "--nproc-per-node 2" in p.stderr and "--master-addr 127.0.0.1" in p.stderr
    # (torch.distributed.run ends the other rank as soon as one has failed: at least one of them got to say why)
    assert "[bench] rank 0: FATAL" in p.stderr or "[bench] rank 1: FATAL" in p.stderr
    assert p.returncode != 0                    # the ranks failed (no GPU): the launcher says so
    assert not [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]


def test_the_launcher_does_not_import_torch_or_numpy():
    code = ("import sys, runpy\n"
            "sys.argv = ['bench.py', '--gpus', '2']\n"
            "import subprocess\n"
            "class Stop(Exception): pass\n"
            "def fake(cmd, env=None):\n"
            "    assert 'torch' not in sys.modules and 'numpy' not in sys.modules\n"
            "    print('CMD', ' '.join(cmd)); raise Stop\n"
            "subprocess.Popen = fake\n"
            "try:\n"
            "    runpy.run_path(%r, run_name='__main__')\n"
            "except Stop:\n"
            "    print('STOPPED-BEFORE-IMPORTS')\n" % os.path.join(ROOT, "bench.py"))
    p = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "STOPPED-BEFORE-IMPORTS" in p.stdout, p.stderr[-2000:]
    assert "-m torch.distributed.run --nnodes=1 --nproc-per-node 2" in p.stdout


def test_one_gpu_and_ranks_of_a_launcher_do_not_launch():
    sys.path.insert(0, ROOT)
    code = ("import os, sys\n"
            "sys.path.insert(0, %r)\n"
            "import importlib.util\n"
            "spec = importlib.util.spec_from_file_location('bench_mod', %r)\n"
            "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
            "m.launch_ranks_if_asked(['--gpus', '1'])\n"
            "os.environ['WORLD_SIZE'] = '4'\n"
            "m.launch_ranks_if_asked(['--gpus', '4'])\n"
            "print('NO-LAUNCH')\n" % (ROOT, os.path.join(ROOT, "bench.py")))
    p = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "NO-LAUNCH" in p.stdout, p.stderr[-2000:]
