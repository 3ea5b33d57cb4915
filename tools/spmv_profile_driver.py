"""Driver for rocprofv3 runs of the CSR matvec: `python3 tools/spmv_profile_driver.py n k width [scramble|stencil|nodiag]`
(k = 26 with `stencil`: 27-point-stencil pattern, local-x windows; `nodiag`: the circulant pattern with the diagonal
windows switched off -- the generic paths of the same plan)."""
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
if len(sys.argv) > 4 and sys.argv[4] == "stencil":
    side = int(round(n ** (1.0 / 3.0)))
    offs = sorted({dz * side * side + dy * side + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)})
    assert len(offs) == k + 1
    i = torch.arange(n, device=col.device, dtype=torch.int64)[:, None]
    d = torch.tensor(offs, device=col.device, dtype=torch.int64)[None, :]
    col = torch.sort(torch.remainder(i + d, n), dim=1).values.reshape(-1).to(torch.int32)
elif len(sys.argv) > 4 and sys.argv[4] == "nodiag":
    pass
elif len(sys.argv) > 4:
    perm = torch.randperm(n, device=col.device, dtype=torch.int64)
    col = perm[col.long()].to(torch.int32)
op = SparseOperator((rowptr, col, val), shape=(n, n), width=width, check=False,
                    diagonal_windows=not (len(sys.argv) > 4 and sys.argv[4] == "nodiag"))
x = problems.xstar(n, 1, eng.device)
y = eng.empty(n)
for _ in range(6):
    op.matvec(x, y)
torch.cuda.synchronize()
