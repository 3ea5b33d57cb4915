"""Analytic 3x3 test problems with known solutions -- the data of the reference's
`ccqppy.problem_suite` (/root/reference/src/ccqppy/problem_suite.py:42-177) behind the same five
class names and properties (`number_of_unknowns`, `A`, `b`, `convex_proj_op`, `exact_solution`),
with the projection operators of this package."""
import numpy as np

from . import solution_spaces as ss

__all__ = ["TestProblemBase", "UnconstrainedSPD1", "UnconstrainedSPD2", "BoxConstrainedSPD",
           "ThinBoxConstrainedSPD", "ActiveBoxConstrainedSPD"]

_TRIDIAGONAL = ((2, -1, 0), (-1, 2, -1), (0, -1, 2))


class TestProblemBase:
    """min 1/2 x'Ax + b'x over a convex set, unique solution known in closed form."""
    __test__ = False          # not a pytest class
    _exact = (1, 0, 1)
    _b_from = (1, 0, 1)       # b = -A . _b_from  (problem_suite.py:58,85,113,141,169)

    @property
    def number_of_unknowns(self):
        return 3

    @property
    def A(self):
        return np.array(_TRIDIAGONAL)

    @property
    def b(self):
        return -self.A.dot(np.array(self._b_from))

    @property
    def exact_solution(self):
        return np.array(self._exact)

    @property
    def convex_proj_op(self):
        raise NotImplementedError


class UnconstrainedSPD1(TestProblemBase):
    @property
    def convex_proj_op(self):
        return ss.IdentityProjOp(3)


class UnconstrainedSPD2(TestProblemBase):
    @property
    def convex_proj_op(self):
        return ss.DisjointProjOp(ss.IdentityProjOp(1), ss.IdentityProjOp(1), ss.IdentityProjOp(1))


class BoxConstrainedSPD(TestProblemBase):
    """unconstrained minimiser inside the box"""
    @property
    def convex_proj_op(self):
        return ss.BoxProjOp(3, lower_bound=np.array([0, 0, 0]), upper_bound=np.array([2, 2, 2]))


class ThinBoxConstrainedSPD(TestProblemBase):
    """unconstrained minimiser inside a thin box"""
    @property
    def convex_proj_op(self):
        return ss.BoxProjOp(3, lower_bound=np.array([-10, -0.1, 0.9]), upper_bound=np.array([10, 0.1, 1.1]))


class ActiveBoxConstrainedSPD(TestProblemBase):
    """unconstrained minimiser (1,1,1) outside the box [9,10]^3"""
    _exact = (9, 9, 9)
    _b_from = (1, 1, 1)

    @property
    def convex_proj_op(self):
        return ss.BoxProjOp(3, lower_bound=np.array([9, 9, 9]), upper_bound=np.array([10, 10, 10]))
