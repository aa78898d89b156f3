"""One-time symbolic preprocessing of the sparse constraints — host side, stays on the host.

Produces exactly the arrays ``preprocess_sparsecons`` builds (src/preprocess.jl:24-169) and
``SolverAuxiliary`` keeps (src/structs.jl:278-288), 0-based, from the batched COO form of
``structs.SparseBatch``.  The reference walks the matrices one by one with a binary search per
entry; this is the same result computed with sorted keys (np.unique / np.searchsorted), so that the
north-star instance (m = 1e5 one-entry matrices) is laid out in well under a second.
"""
from __future__ import annotations

import dataclasses

import numpy as np

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


def _csc_pattern(keys: np.ndarray, n: int):
    """Pattern of ``sparse(I, J, ones, n, n)`` from keys = J·n + I: unique sorted keys ⇒ column-major,
    rows ascending inside a column (src/preprocess.jl:87,90)."""
    ukeys = np.unique(keys)
    cols = ukeys // n
    rowval = (ukeys - cols * n).astype(np.int64)
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(cols, minlength=n), out=colptr[1:])
    return ukeys, colptr, rowval


def preprocess_sparsecons(batch: SparseBatch) -> AggregatedLayout:
    n = int(batch.n)
    I = batch.I.astype(np.int64, copy=False)
    J = batch.J.astype(np.int64, copy=False)
    V = batch.V.astype(np.float64, copy=False)
    if I.size and (I.min() < 0 or I.max() >= n or J.min() < 0 or J.max() >= n):
        raise ValueError("constraint entry outside the n×n matrix")
    upper = I <= J  # triu keeps i <= j (src/preprocess.jl:9)

    full_keys, full_colptr, full_rowval = _csc_pattern(J * n + I, n)
    tri_keys, triu_colptr, triu_rowval = _csc_pattern(J[upper] * n + I[upper], n)

    # per-matrix segments of the upper-triangular entries (src/preprocess.jl:97-132)
    csum = np.concatenate(([0], np.cumsum(upper, dtype=np.int64)))
    matptr = csum[batch.ent_ptr]
    Iu, Ju, Vu = I[upper], J[upper], V[upper]
    nzind = np.searchsorted(tri_keys, Ju * n + Iu).astype(np.int64)
    nzval_one = Vu.copy()
    nzval_two = np.where(Iu == Ju, Vu, 2.0 * Vu)  # off-diagonal entries count twice (:121-128)

    # full pattern → position of (min, max) in the upper-triangular pattern (:135-156)
    fcols = np.repeat(np.arange(n, dtype=np.int64), np.diff(full_colptr))
    rr = np.minimum(full_rowval, fcols)
    cc = np.maximum(full_rowval, fcols)
    want = cc * n + rr
    mapped = np.searchsorted(tri_keys, want).astype(np.int64)
    bad = (mapped >= tri_keys.size)
    if tri_keys.size:
        bad |= tri_keys[np.minimum(mapped, tri_keys.size - 1)] != want
    if bad.any():
        raise ValueError("a constraint matrix is not symmetric: a lower-triangular entry has no "
                         "upper-triangular mirror in the aggregated pattern")
    return AggregatedLayout(n, batch.n_matrices, matptr.astype(np.int64), nzind, nzval_one,
                            nzval_two, batch.global_inds.astype(np.int64), triu_colptr,
                            triu_rowval, full_colptr, full_rowval, mapped)
