"""GPU: the ctypes stub printed in INTEGRATION.md section 2 is executed as written (only the library path is
made absolute) and its two classes -- DeviceMatrix (dense `A.dot`) and DeviceCSR (sparse `A.dot`) -- are
checked against numpy / scipy, so the binding shown to maintainers cannot rot."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def _stub_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 2."):text.index("## 3.")]
    return re.findall(r"```python\n(.*?)```", section, flags=re.S)


def test_documented_ctypes_stub_runs_and_matches_numpy(built_lib):
    blocks = _stub_blocks()
    assert len(blocks) == 2 and "class DeviceMatrix" in blocks[0] and "class DeviceCSR" in blocks[1]
    ns = {}
    code = "\n".join(blocks).replace('ctypes.CDLL("libccqp_hip.so")', "ctypes.CDLL(%r)" % built_lib)
    exec(compile(code, "INTEGRATION.md", "exec"), ns)          # noqa: S102 - our own documentation
    rng = np.random.default_rng(0)
    A = rng.normal(size=(300, 200))
    v = rng.normal(size=200)
    vd = torch.as_tensor(v, device="cuda")
    y = ns["DeviceMatrix"](A).dot(vd).cpu().numpy()
    np.testing.assert_allclose(y, A.dot(v), rtol=1e-12, atol=1e-12)
    S = sp.random(300, 200, density=0.05, format="csr", random_state=1)
    ys = ns["DeviceCSR"](S).dot(vd).cpu().numpy()
    np.testing.assert_allclose(ys, S.dot(v), rtol=1e-12, atol=1e-13)
    # the error contract of the stub: a refused call raises with the library's message
    bad = ns["DeviceMatrix"](A)
    bad.A = bad.A.t()                                # (200, 300) view with row stride 1 < 300 columns
    with pytest.raises(RuntimeError, match="lda < cols"):
        bad.dot(torch.zeros(300, dtype=torch.float64, device="cuda"))
