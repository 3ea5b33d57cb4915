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
