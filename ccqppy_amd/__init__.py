"""ccqppy_amd -- MI355X-native convex-constrained QP engine with the CCQPpy API.

    from ccqppy_amd import solvers
    from ccqppy_amd import solution_spaces as ss

mirrors `from ccqppy import solvers, solution_spaces` (reference: src/ccqppy/__init__.py:1-4).
The hot path runs in libccqp_hip.so (hand-written gfx950 HIP kernels, C ABI in
include/ccqp_hip.h); there is no CPU fallback.

Beyond the reference's API: `operators` (dense / sparse / matrix-free / row-sharded operators), `batch` (schedulers
for many independent problems, incl. the device-resident lock-step ones of `lockstep`), `ensemble` (the benchmark
harness), `problems` / `problem_suite` (generators).
"""
__version__ = "0.1.0"

from .solution_spaces import *  # noqa: F401,F403
from .solvers import *  # noqa: F401,F403
from . import operators, problems, solution_spaces, solvers  # noqa: F401
