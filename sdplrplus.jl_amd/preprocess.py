"""One-time symbolic preprocessing of the sparse constraints — host side, stays on the host.

Produces exactly the arrays ``preprocess_sparsecons`` builds (src/preprocess.jl:24-169) and
``SolverAuxiliary`` keeps (src/structs.jl:278-288), 0-based, from the batched COO form of
``structs.SparseBatch``.  The reference walks the matrices one by one with a binary search per
entry; this is the same result from counting sorts (scipy's COO → CSC / CSC → CSR conversions) and one
sampled lookup, so that the north-star instance (m = 1e5 one-entry matrices, 2.2 M entries) is laid
out in ≈ 50 ms.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import scipy.sparse as sp

from .structs import SparseBatch


@dataclasses.dataclass
class AggregatedLayout:
    n: int
    n_sparse: int
    matptr: np.ndarray         # triu_agg_sparse_A_matptr      [n_sparse+1]
    nzind: np.ndarray          # triu_agg_sparse_A_nzind       [nnzAgg]
    nzval_one: np.ndarray      # triu_agg_sparse_A_nzval_one   [nnzAgg]
    nzval_two: np.ndarray      # triu_agg_sparse_A_nzval_two   [nnzAgg]
    global_inds: np.ndarray    # sparse_As_global_inds         [n_sparse]
    triu_colptr: np.ndarray    # triu_sparse_S.colptr          [n+1]
    triu_rowval: np.ndarray    # triu_sparse_S.rowval          [nnzT]
    full_colptr: np.ndarray    # sparse_S.colptr               [n+1]
    full_rowval: np.ndarray    # sparse_S.rowval               [nnzS]
    mappedto_triu: np.ndarray  # agg_sparse_A_mappedto_triu    [nnzS]

    @property
    def nnzT(self) -> int:
        return int(self.triu_rowval.size)

    @property
    def nnzS(self) -> int:
        return int(self.full_rowval.size)

    @property
    def nnzAgg(self) -> int:
        return int(self.nzind.size)


def _positions_in_pattern(colptr: np.ndarray, rowval: np.ndarray, n: int, I: np.ndarray, J: np.ndarray) -> np.ndarray:
    """Position of every (I[k], J[k]) in the CSC pattern (colptr, rowval) — the binary search of
    src/preprocess.jl:110-119, run in C by scipy's sampler (the CSC arrays of a matrix are the CSR arrays of its
    transpose: sample (row = J, col = I)); −1 where the pair is not stored."""
    nnz = rowval.size
    if I.size == 0 or nnz == 0:
        return np.full(I.size, -1, dtype=np.int64)
    try:
        from scipy.sparse import _sparsetools
        out = np.zeros(I.size, dtype=np.int64)
        _sparsetools.csr_sample_values(n, n, colptr, rowval, np.arange(1, nnz + 1, dtype=np.int64), I.size,
                                       np.ascontiguousarray(J), np.ascontiguousarray(I), out)
        return out - 1
    except (ImportError, AttributeError, TypeError, ValueError):       # another scipy: sorted keys instead
        keys = np.repeat(np.arange(n, dtype=np.int64), np.diff(colptr)) * n + rowval
        want = J * n + I
        pos = np.minimum(np.searchsorted(keys, want), nnz - 1)
        return np.where(keys[pos] == want, pos, -1)


def preprocess_sparsecons(batch: SparseBatch) -> AggregatedLayout:
    n = int(batch.n)
    I = batch.I.astype(np.int64, copy=False)
    J = batch.J.astype(np.int64, copy=False)
    V = batch.V.astype(np.float64, copy=False)
    if I.size and (I.min() < 0 or I.max() >= n or J.min() < 0 or J.max() >= n):
        raise ValueError("constraint entry outside the n×n matrix")
    upper = I <= J  # triu keeps i <= j (src/preprocess.jl:9)

    # aggregated full pattern sparse(I, J, ones) (src/preprocess.jl:87): scipy's COO → CSC is a counting sort in C
    F = sp.csc_matrix((np.ones(I.size, dtype=np.int8), (I, J)), shape=(n, n))
    F.sum_duplicates()
    F.sort_indices()
    full_colptr = F.indptr.astype(np.int64)
    full_rowval = F.indices.astype(np.int64)
    fcols = np.repeat(np.arange(n, dtype=np.int64), np.diff(full_colptr))
    # its upper triangle (:90) — the triu of the union is the union of the trius
    in_upper = full_rowval <= fcols
    triu_rowval = full_rowval[in_upper]
    triu_colptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(fcols[in_upper], minlength=n), out=triu_colptr[1:])
    nnzT = int(triu_rowval.size)

    # per-matrix segments of the upper-triangular entries (src/preprocess.jl:97-132)
    csum = np.concatenate(([0], np.cumsum(upper, dtype=np.int64)))
    matptr = csum[batch.ent_ptr]
    Iu, Ju, Vu = I[upper], J[upper], V[upper]
    nzind = _positions_in_pattern(triu_colptr, triu_rowval, n, Iu, Ju)
    nzval_one = Vu.copy()
    nzval_two = np.where(Iu == Ju, Vu, 2.0 * Vu)  # off-diagonal entries count twice (:121-128)

    # full pattern → position of (min, max) in the upper-triangular pattern (:135-156), without a search: walking the
    # full pattern column by column, its upper entries ARE the triu pattern in order, and its lower entries (i > j in
    # column j) are row j of the triu pattern right of the diagonal — the CSR form of the triu pattern, in order
    mapped = np.empty(full_rowval.size, dtype=np.int64)
    mapped[in_upper] = np.arange(nnzT, dtype=np.int64)
    T = sp.csc_matrix((np.arange(1, nnzT + 1, dtype=np.int64), triu_rowval, triu_colptr), shape=(n, n)).tocsr()
    trows = np.repeat(np.arange(n, dtype=np.int64), np.diff(T.indptr))
    strict = T.indices > trows
    lower = ~in_upper
    if (int(lower.sum()) == int(strict.sum()) and np.array_equal(full_rowval[lower], T.indices[strict])
            and np.array_equal(fcols[lower], trows[strict])):
        mapped[lower] = T.data[strict] - 1
    else:
        # not every strictly-upper entry has its lower mirror (e.g. upper-triangular-only input, which the reference's
        # search accepts): look the lower entries up one by one, as src/preprocess.jl:143-156 does
        pos = _positions_in_pattern(triu_colptr, triu_rowval, n, fcols[lower], full_rowval[lower])
        if pos.size and pos.min() < 0:
            raise ValueError("a constraint matrix is not symmetric: a lower-triangular entry has no "
                             "upper-triangular mirror in the aggregated pattern")
        mapped[lower] = pos
    return AggregatedLayout(n, batch.n_matrices, matptr.astype(np.int64), nzind, nzval_one,
                            nzval_two, batch.global_inds.astype(np.int64), triu_colptr,
                            triu_rowval, full_colptr, full_rowval, mapped)
