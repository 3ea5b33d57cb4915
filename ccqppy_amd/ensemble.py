"""Ensemble benchmark over random CCQPs: solver x projection family x problem size x trial.

Counterpart of `BenchmarkRandomCCQP` (/root/reference/benchmarks/benchmark_random_ccqp.py:15-152):
same constructor, same `run()`, same four result tensors of shape
[solver, projection family, size, trial] (`_problem_residual`, `_problem_converged`, `_problem_time`,
`_problem_num_matrix_vector_mults`; dtypes as :71-77).  Differences, all forced by scale:
  * problems come from the device generator (`problems.random_problem(n, seed=trial)`)
    instead of `scipy.stats.wishart.rvs` + an unseeded `np.random.rand` (:59-63, O(n^3));
  * `process_results()` returns / prints tables (mean and the 95 % band computed with the reference's
    own formula :121-127) instead of opening matplotlib windows; `to_json()` stores everything;
  * with `lanes > 1` the trials of one (solver, operator) cell run concurrently: in lock-step with every scalar on
    the device where that applies (batch.solve_batch_pipelined: PGD / BBPGD / BBPGDf / APGD / AR / MPRGP on elementwise
    sets, PGD / BBPGD on Sphere / Cone / DisjointUnion sets too; the per-trial time is then the cell's wall-clock /
    trials), else on `lanes` streams (batch.solve_batch).  Results per trial are bitwise those of sequential solves
    either way; `cell_scheduler` records which one ran.
"""
import contextlib
import copy
import io
import json

import numpy as np

from . import problems as _problems
from .batch import solve_batch, solve_batch_pipelined
from .lockstep import why_not_pipelined


class BenchmarkRandomCCQP:
    """Benchmark each solver using an ensemble of randomly generated CCQPs."""

    def __init__(self, num_random_trials, solvers_to_benchmark, convex_proj_ops_to_benchmark, lanes=1,
                 device=None, lockstep=True, spg_seed=None):
        self.num_trials = num_random_trials
        self.solvers_to_benchmark = solvers_to_benchmark
        self.convex_proj_ops_to_benchmark = convex_proj_ops_to_benchmark
        self.lanes = lanes
        self.device = device
        self.lockstep = lockstep
        # SPG draws from numpy's global stream (the reference's harness never seeds it, so its SPG cells differ from
        # run to run).  spg_seed = an integer: trial t of every SPG cell draws from its own RandomState(spg_seed + t)
        # instead -- reproducible under every scheduler, and what lets the SPG cells run in lock-step
        self.spg_seed = spg_seed
        self.problem_sizes = np.zeros(len(convex_proj_ops_to_benchmark[0]), dtype=int)
        for i, proj_op in enumerate(self.convex_proj_ops_to_benchmark[0]):
            self.problem_sizes[i] = proj_op.embedded_dimension
        self._problem_residual = None
        self._problem_l2norm_error = None
        self._problem_converged = None
        self._problem_time = None
        self._problem_num_matrix_vector_mults = None
        self.cell_scheduler = {}        # (solver, family, size) -> "lock-step" | "stream lanes" | "sequential"

    def generate_random_convex_quadratic_func(self, problem_size, seed=1234):
        """(A, b) on the device: A = counter-hashed SPD matrix (kappa ~ 3.7), b = -A x*, x* = 1 - 2u."""
        op, b, _ = _problems.random_problem(int(problem_size), int(seed), device=self.device)
        return op, b

    def _copies(self, solver, count):
        """One solver object per trial (the reference reuses one: results are read before the next solve)."""
        made = [copy.copy(solver) for _ in range(count)]
        if self.spg_seed is not None and hasattr(solver, "rng") and solver.rng is None:
            for t, s in enumerate(made):
                s.rng = np.random.RandomState(self.spg_seed + t)
        return made

    def run(self):
        shape = [len(self.solvers_to_benchmark), len(self.convex_proj_ops_to_benchmark),
                 len(self.convex_proj_ops_to_benchmark[0]), self.num_trials]
        self._problem_residual = np.zeros(shape, dtype=float)
        self._problem_converged = np.zeros(shape, dtype=int)
        self._problem_time = np.zeros(shape, dtype=np.float32)
        self._problem_num_matrix_vector_mults = np.zeros(shape, dtype=int)
        cache = {}
        for proj_type_id, proj_ops in enumerate(self.convex_proj_ops_to_benchmark):
            for proj_id, proj_op in enumerate(proj_ops):
                n = int(self.problem_sizes[proj_id])
                for trial_id in range(self.num_trials):
                    if (n, trial_id) not in cache:
                        cache[(n, trial_id)] = self.generate_random_convex_quadratic_func(n, trial_id)
                for solver_id, solver in enumerate(self.solvers_to_benchmark):
                    batch = [(cache[(n, t)][0], cache[(n, t)][1], None, proj_op) for t in range(self.num_trials)]
                    results = None
                    how = "sequential"
                    if self.lanes > 1 and self.lockstep:
                        # decided on the argument check alone: whatever a lock-step run itself raises propagates
                        made = self._copies(solver, len(batch))
                        from ._engine import get_engine
                        if why_not_pipelined(made, batch, get_engine(self.device).device) is None:
                            results = solve_batch_pipelined(iter(made).__next__, batch, device=self.device)
                            for r in results:
                                r._solution_time = r.solution_time / max(len(results), 1)
                            how = "lock-step"
                    if results is not None:
                        pass
                    elif self.lanes > 1:
                        how = "stream lanes"
                        results = solve_batch(iter(self._copies(solver, len(batch))).__next__, batch, lanes=self.lanes,
                                              device=self.device)
                    else:
                        results = []
                        for s, (A, b, x0, op) in zip(self._copies(solver, len(batch)), batch):
                            with contextlib.redirect_stdout(io.StringIO()):
                                results.append(s.solve(A, b, convex_proj_op=op))
                    self.cell_scheduler[(solver_id, proj_type_id, proj_id)] = how
                    for trial_id, r in enumerate(results):
                        idx = (solver_id, proj_type_id, proj_id, trial_id)
                        self._problem_residual[idx] = r.solution_residual
                        self._problem_converged[idx] = r.solution_converged
                        self._problem_time[idx] = r.solution_time
                        self._problem_num_matrix_vector_mults[idx] = r.solution_num_matrix_vector_multiplications
        return self

    def summary(self, data):
        """mean / lower / upper 95 % band per (solver, family, size) -- the arithmetic of plot() :113-127
        (including its `num_problems - 1` degrees of freedom)."""
        import scipy.stats as st
        num_problems = len(self.problem_sizes)
        mean = np.mean(data, axis=3)
        h = np.zeros_like(mean, dtype=float)
        for s in range(data.shape[0]):
            for p in range(data.shape[1]):
                for i in range(num_problems):
                    sem = st.sem(data[s, p, i, :]) if data.shape[3] > 1 else 0.0
                    h[s, p, i] = sem * st.t.ppf((1 + 0.95) / 2., num_problems - 1) if num_problems > 1 else 0.0
        return mean, mean - h, mean + h

    def process_results(self, stream=None):
        out = {}
        for name, data in (("wall-clock time [s]", self._problem_time),
                           ("number of matrix-vector multiplications", self._problem_num_matrix_vector_mults),
                           ("solution residual", self._problem_residual)):
            mean, lo, hi = self.summary(np.asarray(data, dtype=float))
            out[name] = {"mean": mean, "lower_95": lo, "upper_95": hi}
            print(name, file=stream)
            for p, ops in enumerate(self.convex_proj_ops_to_benchmark):
                for s, solver in enumerate(self.solvers_to_benchmark):
                    row = " ".join("%10.4g" % v for v in mean[s, p])
                    print("  %-14s %-22s %s" % (ops[0].name, solver.name, row), file=stream)
        print("", file=stream)
        return out

    def to_json(self, path):
        doc = {"problem_sizes": self.problem_sizes.tolist(), "num_trials": self.num_trials,
               "solvers": [s.name for s in self.solvers_to_benchmark],
               "projection_families": [ops[0].name for ops in self.convex_proj_ops_to_benchmark],
               "residual": self._problem_residual.tolist(), "converged": self._problem_converged.tolist(),
               "time_s": self._problem_time.astype(float).tolist(),
               "matrix_vector_mults": self._problem_num_matrix_vector_mults.tolist(),
               "scheduler": [[[self.cell_scheduler.get((s, p, i), "?") for i in range(len(self.problem_sizes))]
                              for p in range(len(self.convex_proj_ops_to_benchmark))]
                             for s in range(len(self.solvers_to_benchmark))]}
        with open(path, "w") as f:
            json.dump(doc, f)
        return doc
