import sys
sys.path.insert(0, "/root/repo")
import torch
from ccqppy_amd import problems
from ccqppy_amd._engine import get_engine
from ccqppy_amd.operators import SymmetricOperator
eng = get_engine(0)
n = 131072
A = problems.spd_rows(n, 1); x = problems.xstar(n, 1); y2 = eng.empty(n)
op = SymmetricOperator(A)
for _ in range(6): op.matvec(x, y2)
torch.cuda.synchronize()
