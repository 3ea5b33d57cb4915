"""Driver for rocprofv3 runs of the CSR matvec: `python3 tools/spmv_profile_driver.py n k width [scramble]`."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccqppy_amd import problems
from ccqppy_amd._engine import get_engine
from ccqppy_amd.operators import SparseOperator
n, k, width = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3], 0)
eng = get_engine(0)
eng.bind()
rowptr, col, val = problems.sparse_spd_csr(n, k, 1234)
if len(sys.argv) > 4:
    perm = torch.randperm(n, device=col.device, dtype=torch.int64)
    col = perm[col.long()].to(torch.int32)
op = SparseOperator((rowptr, col, val), shape=(n, n), width=width, check=False)
x = problems.xstar(n, 1, eng.device)
y = eng.empty(n)
for _ in range(6):
    op.matvec(x, y)
torch.cuda.synchronize()
