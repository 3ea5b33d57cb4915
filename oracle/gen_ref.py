"""Host (numpy, uint64) statement of the synthetic problem generator  --  TEST
INFRASTRUCTURE (oracle side).  The product generator is the HIP kernel behind
`ccqp_gen_spd_rows_f64` / `ccqp_gen_vec_f64` (include/ccqp_hip.h); this file is the
independent CPU statement the GPU tests compare it with, bit for bit.

Why a new generator: the reference's own random problems come from
`scipy.stats.wishart.rvs` + an unseeded `np.random.rand`
(/root/reference/benchmarks/benchmark_random_ccqp.py:59-63), O(n^3) and only used
for n <= 12.  SURVEY.md 8(d) defines the replacement below: a counter-hashed
symmetric matrix that any rank can fill for its own row range.

  u(k)      = (splitmix64(key + k) >> 11) * 2^-53                in [0, 1)
  A[i][j]   = (2*u_A(min(i,j)*n + max(i,j)) - 1) / sqrt(n)       i != j
  A[i][i]   = 2.0
  xstar[i]  = 1 - 2*u_x(i)          (same law as benchmark_random_ccqp.py:61)
  b         = -A . xstar            (benchmark_random_ccqp.py:62)

Spectrum ~ Wigner semicircle shifted by 2: [2 - 2/sqrt(3), 2 + 2/sqrt(3)], kappa ~ 3.7.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_VEC_STREAM = np.uint64(0xD1B54A32D192ED03)
_TWO_M53 = 2.0 ** -53


def splitmix64(z):
    """One SplitMix64 output step on uint64 arrays (wrapping arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _key(seed, stream):
    s = np.array([np.uint64(seed & 0xFFFFFFFFFFFFFFFF)], dtype=np.uint64)
    return splitmix64(s ^ stream)[0]


def matrix_key(seed):
    return _key(seed, np.uint64(0))


def vector_key(seed):
    return _key(seed, _VEC_STREAM)


def uniform01(key, k):
    with np.errstate(over="ignore"):
        h = splitmix64(np.asarray(k, dtype=np.uint64) + key)
    return (h >> np.uint64(11)).astype(np.float64) * _TWO_M53


def spd_rows(n, seed, row0=0, rows=None):
    """Rows [row0, row0+rows) of the n x n matrix, C-contiguous float64."""
    if rows is None:
        rows = n - row0
    key = matrix_key(seed)
    sq = np.sqrt(np.float64(n))
    out = np.empty((rows, n), dtype=np.float64)
    j = np.arange(n, dtype=np.uint64)
    nn = np.uint64(n)
    step = max(1, (1 << 22) // max(n, 1))
    for r in range(0, rows, step):
        i = np.arange(row0 + r, min(row0 + r + step, row0 + rows), dtype=np.uint64)[:, None]
        lo = np.minimum(i, j[None, :])
        hi = np.maximum(i, j[None, :])
        with np.errstate(over="ignore"):
            k = lo * nn + hi
        u = uniform01(key, k)
        blk = (2.0 * u - 1.0) / sq
        ii = (i[:, 0] - np.uint64(row0 + r)).astype(np.int64)
        blk[ii, i[:, 0].astype(np.int64)] = 2.0
        out[r:r + blk.shape[0]] = blk
    return out


def xstar(n, seed):
    return 1.0 - 2.0 * uniform01(vector_key(seed), np.arange(n, dtype=np.uint64))


def problem(n, seed):
    """(A, b, xstar) with b = -A.xstar computed by numpy (used for n <= 32768 fixtures)."""
    A = spd_rows(n, seed)
    xs = xstar(n, seed)
    return A, -A.dot(xs), xs


# ------------------------------------------------------------------ sparse problems
# The reference's solvers only call `A.dot(v)`, so a scipy.sparse matrix is a legal A for them.  Two
# families are used by the sparse fixtures:
#   * "sgen": the dense generator matrix above restricted to k symmetric off-diagonals (a circulant
#     pattern: columns (i +- d_t) mod n for k/2 hashed distinct offsets d_t in [1, n/2 - 1]) with the
#     off-diagonal entries rescaled to (2u - 1)/k and the diagonal kept at 2 -- Gershgorin puts the
#     spectrum inside [1, 3].  Any row range can be produced independently (row-sharded runs).
#   * "sirr": an irregular symmetric, strictly diagonally dominant matrix from scipy.sparse.random, with
#     row lengths from 1 (diagonal only) to dozens of entries.
_OFF_STREAM = np.uint64(0xA0761D6478BD642F)


def sparse_offsets(n, k, seed):
    """k/2 distinct positive offsets in [1, n/2 - 1] (hashed; collisions resolved by linear probing)."""
    m = k // 2
    span = n // 2 - 1
    assert k % 2 == 0 and 1 <= m <= span
    key = _key(seed, _OFF_STREAM)
    with np.errstate(over="ignore"):
        h = splitmix64(np.arange(m, dtype=np.uint64) + key)
    used, out = set(), []
    for t in range(m):
        d = 1 + int(h[t] % np.uint64(span))
        while d in used:
            d = d % span + 1
        used.add(d)
        out.append(d)
    return out


def stencil_offsets(n):
    """The 13 positive offsets of a periodic 27-point stencil on a side^3 grid, side = round(n^(1/3)) >= 4
    (dz*side^2 + dy*side + dx over {-1,0,1}^3): a banded pattern whose neighbouring rows share columns."""
    side = int(round(n ** (1.0 / 3.0)))
    assert side >= 4 and 2 * (side * side + side + 1) < n
    offs = sorted({dz * side * side + dy * side + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)})
    return [d for d in offs if d > 0]


def sparse_spd(n, k, seed, row0=0, rows=None, pattern="hashed"):
    """Rows [row0, row0+rows) of the n x n "sgen" matrix as scipy CSR (sorted columns, k+1 entries per row).
    pattern "hashed": k/2 hashed offsets spread over the whole vector; "stencil27": k = 26, the 27-point stencil."""
    import scipy.sparse as sp
    if rows is None:
        rows = n - row0
    if pattern == "stencil27":
        assert k == 26
        offs = stencil_offsets(n)
    else:
        offs = sparse_offsets(n, k, seed)
    delta = np.array(sorted([-d for d in offs] + [0] + offs), dtype=np.int64)
    i = np.arange(row0, row0 + rows, dtype=np.int64)[:, None]
    cols = np.sort((i + delta[None, :]) % n, axis=1)
    lo = np.minimum(i, cols).astype(np.uint64)
    hi = np.maximum(i, cols).astype(np.uint64)
    with np.errstate(over="ignore"):
        kk = lo * np.uint64(n) + hi
    vals = (2.0 * uniform01(matrix_key(seed), kk) - 1.0) / np.float64(k)
    vals[cols == i] = 2.0
    w = delta.shape[0]
    indptr = np.arange(rows + 1, dtype=np.int64) * w
    return sp.csr_matrix((vals.reshape(-1), cols.reshape(-1).astype(np.int32), indptr), shape=(rows, n))


def sparse_problem(n, k, seed, pattern="hashed"):
    """(A csr, b, xstar) with b = -A.xstar computed by scipy."""
    A = sparse_spd(n, k, seed, pattern=pattern)
    xs = xstar(n, seed)
    return A, -A.dot(xs), xs


def irregular_spd(n, seed, density=0.02):
    """Irregular symmetric strictly diagonally dominant CSR matrix (row lengths vary; every 7th row and
    column is cleared except for its diagonal)."""
    import scipy.sparse as sp
    rng = np.random.RandomState(seed)
    R = sp.random(n, n, density=density, format="csr", random_state=rng, data_rvs=lambda s: rng.uniform(-1, 1, s))
    S = (R + R.T).tolil()
    for r in range(0, n, 7):
        S[r, :] = 0
        S[:, r] = 0
    S = S.tocsr()
    S.setdiag(0)
    S.eliminate_zeros()
    d = 1.0 + np.asarray(abs(S).sum(axis=1)).reshape(-1)
    A = (S + sp.diags(d)).tocsr()
    A.sort_indices()
    return A


def irregular_problem(n, seed):
    A = irregular_spd(n, seed)
    xs = xstar(n, seed)
    return A, -A.dot(xs), xs


# ------------------------------------------------------------------ matrix-free "contact" Hessians
class ContactHessian:
    """A = D . diag(minv) . D^T + eps * I, never formed: the only thing the reference's solvers need from A is
    `A.dot(v)` (solvers.py:25), so this object is a legal A for them.  Evaluated exactly as written below --
    the product-side ComposedOperator(D, Minv, D.T, shift=eps) performs the same sequence of roundings."""

    def __init__(self, D, minv, eps):
        self.D = D.tocsr()
        self.Dt = D.T.tocsr()
        self.minv = np.asarray(minv, dtype=np.float64)
        self.eps = float(eps)
        self.shape = (D.shape[0], D.shape[0])

    def dot(self, v):
        t = self.Dt.dot(v)
        t = self.minv * t
        t = self.D.dot(t)
        return t + self.eps * v

    def toarray(self):
        return (self.D @ np.diag(self.minv) @ self.D.T.toarray()) + self.eps * np.eye(self.shape[0])


def contact_problem(m, n, seed, per_row=6, eps=0.05):
    """(ContactHessian, b, xstar): D is m x n with ~per_row entries per row (a constraint touches a few
    bodies), minv in [0.5, 1.5); b = -A.xstar."""
    import scipy.sparse as sp
    rng = np.random.RandomState(seed)
    D = sp.random(m, n, density=min(1.0, per_row / n), format="csr", random_state=rng,
                  data_rvs=lambda s: rng.uniform(-1, 1, s))
    D.sort_indices()
    minv = 0.5 + uniform01(vector_key(seed + 1), np.arange(n, dtype=np.uint64))
    A = ContactHessian(D, minv, eps)
    xs = xstar(m, seed)
    return A, -A.dot(xs), xs
