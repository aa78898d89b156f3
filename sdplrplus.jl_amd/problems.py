"""SDP problem constructors and synthetic graph generators (the data format just before the hot path).

``maxcut`` … ``mu_conductance_ineq`` return ``(C, As, bs[, constraint_types])`` exactly like the
reference's builders (test/problem.jl:16-236 = exps/problems.jl:14-341), with 0-based indices.
The ``*_data`` variants return an ``SDPData`` built straight from batched arrays — same matrices,
no per-constraint Python objects — for the n = 1e5 configurations of BASELINE.json.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import scipy.sparse as sp

from .structs import SDPData, SparseBatch, SparseMatrixCOO, SymLowRankMatrix

# ------------------------------------------------------------------------------------------------
# graphs
# ------------------------------------------------------------------------------------------------


def _check_undirected(A: sp.spmatrix) -> sp.csc_matrix:
    A = sp.csc_matrix(A, dtype=np.float64)
    if (abs(A - A.T) > 0).nnz != 0:
        raise ValueError("Only undirected graphs supported now.")
    A.sort_indices()
    return A


def make_random_graph(n: int, p: float, rng: np.random.Generator) -> sp.csc_matrix:
    """test/runtests.jl:30-36: threshold the average of two uniform matrices at p, zero diagonal."""
    M = rng.random((n, n))
    M = (M + M.T) / 2
    M = (M > p).astype(np.float64)
    np.fill_diagonal(M, 0.0)
    return sp.csc_matrix(M)


def gnp_graph(n: int, p: float, seed: int) -> sp.csc_matrix:
    """G(n, p), unit weights, no self-loops: upper-triangular pairs sampled by geometric skipping
    (SURVEY.md §8d config 2), then symmetrised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    total = n * (n - 1) // 2
    picks = []
    pos = -1
    expect = max(int(total * p * 1.1) + 1024, 1024)
    while True:
        gaps = rng.geometric(p, size=expect).astype(np.int64)
        idx = pos + np.cumsum(gaps)
        keep = idx[idx < total]
        picks.append(keep)
        if keep.size < idx.size:
            break
        pos = int(idx[-1])
    t = np.concatenate(picks)
    # linear index t of the strict upper triangle (row-major) → (i, j), i < j
    tf = t.astype(np.float64)
    i = np.floor(((2 * n - 1) - np.sqrt((2 * n - 1) ** 2 - 8 * tf)) / 2).astype(np.int64)
    start = lambda ii: ii * (2 * n - ii - 1) // 2
    i = np.where(start(i) > t, i - 1, i)
    i = np.where(start(i + 1) <= t, i + 1, i)
    j = t - start(i) + i + 1
    assert (i >= 0).all() and (j > i).all() and (j < n).all()
    w = np.ones(i.size)
    A = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([i, j]), np.concatenate([j, i]))),
                      shape=(n, n)).tocsc()
    A.sort_indices()
    return A


def chung_lu_graph(n: int, avg_degree: float, exponent: float, seed: int) -> sp.csc_matrix:
    """Power-law expected-degree (Chung–Lu) graph, largest connected component, self-loops removed —
    the offline stand-in for the SNAP graph of BASELINE.json config 3, mirroring the
    clean-up of exps/data_preprocess.jl:118-134."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = (np.arange(1, n + 1, dtype=np.float64)) ** (-1.0 / (exponent - 1.0))
    w *= avg_degree * n / w.sum()
    n_edges = int(round(avg_degree * n / 2))
    prob = w / w.sum()
    u = rng.choice(n, size=n_edges, p=prob)
    v = rng.choice(n, size=n_edges, p=prob)
    keep = u != v
    u, v = u[keep], v[keep]
    A = sp.coo_matrix((np.ones(u.size), (u, v)), shape=(n, n)).tocsr()
    A = ((A + A.T) > 0).astype(np.float64)
    ncomp, labels = sp.csgraph.connected_components(A, directed=False)
    big = np.argmax(np.bincount(labels))
    idx = np.flatnonzero(labels == big)
    A = sp.csc_matrix(A[idx][:, idx])
    A.sort_indices()
    return A


def graph_from_edges(n: int, edges: np.ndarray, weights=None) -> sp.csc_matrix:
    """Symmetric adjacency from an (E, 2) array of 0-based undirected edges."""
    edges = np.asarray(edges, dtype=np.int64)
    w = np.ones(edges.shape[0]) if weights is None else np.asarray(weights, dtype=np.float64)
    i, j = edges[:, 0], edges[:, 1]
    A = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([i, j]), np.concatenate([j, i]))),
                      shape=(n, n)).tocsc()
    A.sort_indices()
    return A


def _laplacian(A: sp.csc_matrix, scale: float) -> sp.csc_matrix:
    d = np.asarray(A.sum(axis=1)).ravel()
    L = sp.csc_matrix(sp.diags(d) - A)
    L = sp.csc_matrix(L * scale)
    L.sort_indices()
    return L


# ------------------------------------------------------------------------------------------------
# list-returning builders (reference API)
# ------------------------------------------------------------------------------------------------


def maxcut(A) -> Tuple[sp.csc_matrix, list, np.ndarray]:
    """minimize −¼⟨L, X⟩ s.t. Diag(X) = 1  (test/problem.jl:16-30)."""
    A = _check_undirected(A)
    n = A.shape[0]
    L = _laplacian(A, -0.25)
    As = [SparseMatrixCOO([i], [i], [1.0], n, n) for i in range(n)]
    return L, As, np.ones(n)


def lovasz_theta(A) -> Tuple[SymLowRankMatrix, list, np.ndarray]:
    """minimize −⟨11ᵀ, X⟩ s.t. Tr(X) = 1, X_ij = 0 ∀(i,j) ∈ E  (test/problem.jl:43-65)."""
    A = _check_undirected(A)
    n = A.shape[0]
    C = SymLowRankMatrix(-np.ones(1), np.ones((n, 1)))
    As, bs = [], []
    coo = A.tocoo()  # column-major walk like zip(findnz(A)...)
    order = np.lexsort((coo.row, coo.col))
    for i, j in zip(coo.row[order], coo.col[order]):
        if i < j:
            As.append(SparseMatrixCOO([i, j], [j, i], [1.0, 1.0], n, n))
            bs.append(0.0)
        elif i == j:
            As.append(SparseMatrixCOO([i], [i], [1.0], n, n))
            bs.append(0.0)
    As.append(sp.identity(n, format="csc"))
    bs.append(1.0)
    return C, As, np.asarray(bs)


def minimum_bisection(A) -> Tuple[sp.csc_matrix, list, np.ndarray]:
    """minimize ¼⟨L, X⟩ s.t. Diag(X) = 1, 1ᵀX1 = 0  (test/problem.jl:78-94)."""
    A = _check_undirected(A)
    n = A.shape[0]
    L = _laplacian(A, 0.25)
    As = [SparseMatrixCOO([i], [i], [1.0], n, n) for i in range(n)]
    As.append(SymLowRankMatrix(np.ones(1), np.ones((n, 1))))
    return L, As, np.concatenate([np.ones(n), [0.0]])


def cutnorm(A) -> Tuple[sp.csc_matrix, list, np.ndarray]:
    """−½·[0 A; Aᵀ 0] with unit diagonal constraints  (test/problem.jl:96-112)."""
    A = sp.csc_matrix(A, dtype=np.float64)
    m_, n_ = A.shape
    C = sp.bmat([[None, A], [A.T, None]], format="csc") / 2
    N = m_ + n_
    As = [SparseMatrixCOO([i], [i], [1.0], N, N) for i in range(N)]
    C = sp.csc_matrix(-C)
    C.sort_indices()
    return C, As, np.ones(N)


def _mu_bounds(volG: float, mu: float):
    return (1 - mu) / (mu * volG), mu / ((1 - mu) * volG)  # ub, lb (test/problem.jl:119-120)


def mu_conductance(A, mu: float):
    """μ-conductance lifted to 3n×3n with slack diagonals  (test/problem.jl:137-179)."""
    A = _check_undirected(A)
    n = A.shape[0]
    d = np.asarray(A.sum(axis=1)).ravel()
    volG = d.sum()
    L = sp.csc_matrix(sp.diags(d) - A)
    N = 3 * n
    pad = lambda M: sp.csc_matrix((M.tocoo().data, (M.tocoo().row, M.tocoo().col)), shape=(N, N))
    padded_d = np.concatenate([d, np.zeros(2 * n)])
    As, bs = [], []
    As.append(pad(sp.csc_matrix(sp.diags(d))))
    bs.append(1.0)
    As.append(SymLowRankMatrix(np.ones(1), padded_d.reshape(-1, 1)))
    bs.append(0.0)
    ub, lb = _mu_bounds(volG, mu)
    for i in range(n):
        As.append(SparseMatrixCOO([i, i + n], [i, i + n], [1.0, 1.0], N, N))
        bs.append(ub)
    for i in range(n):
        As.append(SparseMatrixCOO([i, i + 2 * n], [i, i + 2 * n], [1.0, -1.0], N, N))
        bs.append(lb)
    C = pad(L)
    C.sort_indices()
    return C, As, np.asarray(bs)


def mu_conductance_ineq(A, mu: float):
    """μ-conductance with native inequality rows  (test/problem.jl:196-236)."""
    A = _check_undirected(A)
    n = A.shape[0]
    d = np.asarray(A.sum(axis=1)).ravel()
    volG = d.sum()
    L = sp.csc_matrix(sp.diags(d) - A)
    L.sort_indices()
    ub, lb = _mu_bounds(volG, mu)
    As, bs, ct = [], [], []
    As.append(sp.csc_matrix(sp.diags(d))); bs.append(1.0); ct.append(False)
    As.append(SymLowRankMatrix(np.ones(1), d.reshape(-1, 1))); bs.append(0.0); ct.append(False)
    for i in range(n):
        As.append(SparseMatrixCOO([i], [i], [1.0], n, n)); bs.append(ub); ct.append(True)
    for i in range(n):
        As.append(SparseMatrixCOO([i], [i], [-1.0], n, n)); bs.append(-lb); ct.append(True)
    return L, As, np.asarray(bs), np.asarray(ct, dtype=bool)


def _pad3(M, n):
    M = sp.coo_matrix(M)
    out = sp.csc_matrix((M.data, (M.row, M.col)), shape=(3 * n, 3 * n))
    out.sort_indices()
    return out


def relaxed_maxcut(A):
    """−¼⟨L,X⟩ with 0.99 ≤ Diag(X) ≤ 1 through slack diagonals (exps/problems.jl:188-216)."""
    A = _check_undirected(A)
    n = A.shape[0]
    N = 3 * n
    L = _pad3(_laplacian(A, -0.25), n)
    As, bs = [], []
    for i in range(n):
        As.append(SparseMatrixCOO([i, i + n], [i, i + n], [1.0, 1.0], N, N)); bs.append(1.0)
    for i in range(n):
        As.append(SparseMatrixCOO([i, i + 2 * n], [i, i + 2 * n], [1.0, -1.0], N, N)); bs.append(0.99)
    return L, As, np.asarray(bs)


def mu_conductance_reformulated(A, mu: float):
    """exps/problems.jl:233-281: the second slack block bounds the first slack (ub − lb)."""
    A = _check_undirected(A)
    n = A.shape[0]
    N = 3 * n
    d = np.asarray(A.sum(axis=1)).ravel()
    volG = d.sum()
    L = sp.csc_matrix(sp.diags(d) - A)
    ub, lb = _mu_bounds(volG, mu)
    As = [_pad3(sp.diags(d), n), SymLowRankMatrix(np.ones(1), np.concatenate([d, np.zeros(2 * n)]).reshape(-1, 1))]
    bs = [1.0, 0.0]
    for i in range(n):
        As.append(SparseMatrixCOO([i, i + n], [i, i + n], [1.0, 1.0], N, N)); bs.append(ub)
    for i in range(n):
        As.append(SparseMatrixCOO([i + n, i + 2 * n], [i + n, i + 2 * n], [1.0, 1.0], N, N)); bs.append(ub - lb)
    return _pad3(L, n), As, np.asarray(bs)


def mu_conductance_native(A, mu: float):
    """exps/problems.jl:295-341: n×n, inequality rows scaled by ‖D‖_F, rank-1 row scaled by ‖D‖_F/‖d‖²."""
    A = _check_undirected(A)
    n = A.shape[0]
    d = np.asarray(A.sum(axis=1)).ravel()
    volG = d.sum()
    L = sp.csc_matrix(sp.diags(d) - A)
    L.sort_indices()
    D_norm = float(np.linalg.norm(d))          # norm(D, 2) of the diagonal matrix = ‖d‖₂
    dd_norm = float(np.linalg.norm(d) ** 2)
    ub, lb = _mu_bounds(volG, mu)
    As = [sp.csc_matrix(sp.diags(d)), SymLowRankMatrix(np.array([D_norm / dd_norm]), d.reshape(-1, 1))]
    bs, ct = [1.0, 0.0], [False, False]
    for i in range(n):
        As.append(SparseMatrixCOO([i], [i], [D_norm], n, n)); bs.append(ub * D_norm); ct.append(True)
    for i in range(n):
        As.append(SparseMatrixCOO([i], [i], [-D_norm], n, n)); bs.append(-lb * D_norm); ct.append(True)
    return L, As, np.asarray(bs), np.asarray(ct, dtype=bool)


# ------------------------------------------------------------------------------------------------
# batched builders (same matrices, no per-constraint objects)
# ------------------------------------------------------------------------------------------------


def _csc_entries(M: sp.csc_matrix):
    M = sp.csc_matrix(M)
    M.sort_indices()
    cols = np.repeat(np.arange(M.shape[1], dtype=np.int64), np.diff(M.indptr))
    return M.indices.astype(np.int64), cols, M.data.astype(np.float64)


def _batch(n, groups, gids) -> SparseBatch:
    """groups: list of (ptr_increments, I, J, V) blocks appended in order."""
    ptrs, Is, Js, Vs = [np.zeros(1, dtype=np.int64)], [], [], []
    off = 0
    for counts, I, J, V in groups:
        ptrs.append(off + np.cumsum(counts, dtype=np.int64))
        off += int(np.sum(counts))
        Is.append(I); Js.append(J); Vs.append(V)
    return SparseBatch(n, np.concatenate(ptrs), np.concatenate(Is).astype(np.int64),
                       np.concatenate(Js).astype(np.int64), np.concatenate(Vs).astype(np.float64),
                       np.asarray(gids, dtype=np.int64))


def maxcut_data(A) -> SDPData:
    A = _check_undirected(A)
    n = A.shape[0]
    L = _laplacian(A, -0.25)
    idx = np.arange(n, dtype=np.int64)
    LI, LJ, LV = _csc_entries(L)
    batch = _batch(n, [(np.ones(n, dtype=np.int64), idx, idx, np.ones(n)),
                       (np.array([LI.size]), LI, LJ, LV)], np.arange(n + 1))
    return SDPData.from_batch(L, np.ones(n), batch, [])


def cutnorm_data(A) -> SDPData:
    """``cutnorm`` (test/problem.jl:96-112) straight from batched arrays: C = −½·[0 A; Aᵀ 0] on 2n vertices, unit diagonal
    constraints — MaxCut-shaped (the reference's batch runs it on the Gset graphs: exps/gen_batch_test.jl:3)."""
    A = sp.csc_matrix(A, dtype=np.float64)
    m_, n_ = A.shape
    C = sp.csc_matrix(-(sp.bmat([[None, A], [A.T, None]], format="csc") / 2))
    C.sort_indices()
    N = m_ + n_
    idx = np.arange(N, dtype=np.int64)
    CI, CJ, CV = _csc_entries(C)
    batch = _batch(N, [(np.ones(N, dtype=np.int64), idx, idx, np.ones(N)),
                       (np.array([CI.size]), CI, CJ, CV)], np.arange(N + 1))
    return SDPData.from_batch(C, np.ones(N), batch, [])


def minimum_bisection_data(A) -> SDPData:
    A = _check_undirected(A)
    n = A.shape[0]
    L = _laplacian(A, 0.25)
    idx = np.arange(n, dtype=np.int64)
    LI, LJ, LV = _csc_entries(L)
    gids = np.concatenate([np.arange(n), [n + 1]])  # constraints 0..n-1, C = index m = n+1
    batch = _batch(n, [(np.ones(n, dtype=np.int64), idx, idx, np.ones(n)),
                       (np.array([LI.size]), LI, LJ, LV)], gids)
    lowrank = [(n, SymLowRankMatrix(np.ones(1), np.ones((n, 1))))]
    return SDPData.from_batch(L, np.concatenate([np.ones(n), [0.0]]), batch, lowrank)


def lovasz_theta_data(A) -> SDPData:
    A = _check_undirected(A)
    n = A.shape[0]
    coo = A.tocoo()
    order = np.lexsort((coo.row, coo.col))
    ri, ci = coo.row[order].astype(np.int64), coo.col[order].astype(np.int64)
    keep = ri <= ci
    ri, ci = ri[keep], ci[keep]
    offd = ri < ci
    counts = np.where(offd, 2, 1).astype(np.int64)
    # entry lists: (i,j),(j,i) for an edge, (i,i) for a self-loop (test/problem.jl:51-60)
    I = np.empty(int(counts.sum()), dtype=np.int64)
    J = np.empty_like(I)
    starts = np.concatenate(([0], np.cumsum(counts)[:-1]))
    I[starts] = ri; J[starts] = ci
    second = starts[offd] + 1
    I[second] = ci[offd]; J[second] = ri[offd]
    ne = ri.size
    idx = np.arange(n, dtype=np.int64)
    batch = _batch(n, [(counts, I, J, np.ones(I.size)),
                       (np.array([n]), idx, idx, np.ones(n))], np.arange(ne + 1))
    C = SymLowRankMatrix(-np.ones(1), np.ones((n, 1)))
    b = np.concatenate([np.zeros(ne), [1.0]])
    return SDPData.from_batch(C, b, batch, [(ne + 1, C)])
