"""Host-side data structures mirroring the reference's (src/structs.jl, src/options.jl).

Only what the hot path's boundary needs: the matrix container types the public ``sdplr`` accepts
(src/sdplr.jl:29-34), ``SDPData`` (src/structs.jl:150-180) normalised into the batched COO form
the preprocessing consumes, and ``BurerMonteiroConfig`` (src/options.jl:1-24).
"""
from __future__ import annotations

import dataclasses
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp


class SparseMatrixCOO:
    """LuxurySparse.SparseMatrixCOO stand-in (fields ``is, js, vs, m, n``; src/preprocess.jl:8,15).

    Indices are 0-based here (numpy convention); entries keep their stored order, which is the
    order ``findnz`` yields them in the reference.
    """

    __slots__ = ("is_", "js", "vs", "m", "n")

    def __init__(self, is_, js, vs, m: int, n: int):
        self.is_ = np.asarray(is_, dtype=np.int64)
        self.js = np.asarray(js, dtype=np.int64)
        self.vs = np.asarray(vs, dtype=np.float64)
        self.m, self.n = int(m), int(n)

    @property
    def shape(self):
        return (self.m, self.n)

    def toarray(self) -> np.ndarray:
        out = np.zeros((self.m, self.n))
        np.add.at(out, (self.is_, self.js), self.vs)
        return out


class Diagonal:
    """LinearAlgebra.Diagonal stand-in; converted with ``sparse(A)`` (src/structs.jl:307-309)."""

    __slots__ = ("diag",)

    def __init__(self, diag):
        self.diag = np.asarray(diag, dtype=np.float64)

    @property
    def shape(self):
        return (self.diag.size, self.diag.size)

    def toarray(self) -> np.ndarray:
        return np.diag(self.diag)


class SymLowRankMatrix:
    """Symmetric low-rank matrix B·D·Bᵀ (src/structs.jl:11-24).  ``B`` is n×s, ``D`` the s diagonal
    entries."""

    __slots__ = ("D", "B")

    def __init__(self, D, B):
        self.D = np.asarray(D, dtype=np.float64).reshape(-1)
        B = np.asarray(B, dtype=np.float64)
        if B.ndim == 1:
            B = B.reshape(-1, 1)
        if B.shape[1] != self.D.size:
            raise ValueError("SymLowRankMatrix: B must be n×s with s = length(D)")
        self.B = np.ascontiguousarray(B)

    @property
    def shape(self):
        return (self.B.shape[0], self.B.shape[0])

    def toarray(self) -> np.ndarray:
        return (self.B * self.D) @ self.B.T

    def norm(self, p=2) -> float:
        """``norm(A, p)`` for p ∈ {2 (Frobenius), Inf} (src/structs.jl:61-82).

        The reference forms every column of B·D·Bᵀ (O(n²s)); the Frobenius value is computed here from
        the s×s Gram matrix, ‖BDBᵀ‖_F² = tr((D·BᵀB)²), which is the same number in O(n s²).
        """
        if p == 2:
            G = self.B.T @ self.B
            M = (self.D[:, None] * G)
            return float(np.sqrt(max(np.trace(M @ M), 0.0)))
        if p == np.inf:
            # max_i ‖(BDBᵀ)[:, i]‖_∞ = largest |entry|; done column-block-wise to stay O(n·chunk)
            U = self.B * self.D
            res = 0.0
            for lo in range(0, self.B.shape[0], 4096):
                res = max(res, float(np.abs(U @ self.B[lo:lo + 4096].T).max()))
            return res
        raise ValueError("undefined norm for Constraint")  # src/structs.jl:80


def _entries_findnz_order(A) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(I, J, V) of a sparse constraint in the order the reference's ``findnz`` yields them."""
    if isinstance(A, SparseMatrixCOO):
        return A.is_, A.js, A.vs
    if isinstance(A, Diagonal):
        idx = np.arange(A.diag.size, dtype=np.int64)
        return idx, idx, A.diag
    if sp.issparse(A):
        A = sp.csc_matrix(A)
        A.sort_indices()
        cols = np.repeat(np.arange(A.shape[1], dtype=np.int64), np.diff(A.indptr))
        return A.indices.astype(np.int64), cols, A.data.astype(np.float64)
    raise TypeError("Currently only sparse/symmetric low-rank/diagonal constraints are supported.")


@dataclasses.dataclass
class SparseBatch:
    """A list of sparse n×n matrices as concatenated COO entry lists (the form
    ``preprocess_sparsecons`` walks matrix by matrix, src/preprocess.jl:63-79,97-131)."""

    n: int
    ent_ptr: np.ndarray      # [nA+1] entry range of matrix k
    I: np.ndarray            # 0-based rows
    J: np.ndarray            # 0-based cols
    V: np.ndarray
    global_inds: np.ndarray  # [nA] 0-based index into the (m+1)-vectors (m ⇒ the cost matrix)

    @property
    def n_matrices(self) -> int:
        return int(self.ent_ptr.size - 1)


class SDPData:
    """``SDPData`` (src/structs.jl:150-180) plus the classification done at the top of
    ``SolverAuxiliary`` (src/structs.jl:296-332): sparse/diagonal matrices (C last, global index
    m) go to ``sparse``; SymLowRankMatrix ones to ``lowrank`` as (global index, matrix)."""

    def __init__(self, C, As, b, constraint_types: Optional[Sequence[bool]] = None,
                 _prebuilt: Optional[Tuple[SparseBatch, list]] = None):
        self.C = C
        self.As = As
        self.b = np.ascontiguousarray(np.asarray(b, dtype=np.float64))
        self.n = int(C.shape[0])
        self.m = int(self.b.size)
        if constraint_types is None:
            self.constraint_types = np.zeros(self.m, dtype=bool)
        else:
            self.constraint_types = np.asarray(constraint_types, dtype=bool)
            if self.constraint_types.size != self.m:
                raise ValueError("constraint_types must have length m")
        self.has_inequalities = bool(self.constraint_types.any())
        if _prebuilt is not None:
            self.sparse, self.lowrank = _prebuilt
        else:
            if len(As) != self.m:
                raise ValueError("length(As) must equal length(b)")
            self.sparse, self.lowrank = self._classify(C, As)

    def _classify(self, C, As):
        ptr, Is, Js, Vs, gids, lowrank = [0], [], [], [], [], []
        for gid, A in list(enumerate(As)) + [(self.m, C)]:
            if isinstance(A, SymLowRankMatrix):
                lowrank.append((gid, A))
                continue
            I, J, V = _entries_findnz_order(A)
            Is.append(I); Js.append(J); Vs.append(V)
            ptr.append(ptr[-1] + I.size)
            gids.append(gid)
        cat = lambda xs, dt: (np.concatenate(xs).astype(dt) if xs else np.zeros(0, dtype=dt))
        batch = SparseBatch(self.n, np.asarray(ptr, dtype=np.int64), cat(Is, np.int64),
                            cat(Js, np.int64), cat(Vs, np.float64),
                            np.asarray(gids, dtype=np.int64))
        return batch, lowrank

    @classmethod
    def from_batch(cls, C, b, sparse: SparseBatch, lowrank: list, constraint_types=None):
        """Build directly from an already-batched constraint set (large instances: avoids creating
        m Python matrix objects).  ``sparse`` must already contain C (global index m) if C is sparse."""
        return cls(C, None, b, constraint_types, _prebuilt=(sparse, lowrank))

    def normC(self) -> float:
        """``norm(C, 2)`` (src/sdplr.jl:160): Frobenius norm (computed once per SDPData: C does not change)."""
        cached = getattr(self, "_normC", None)
        if cached is None:
            cached = self._normC = self._compute_normC()
        return cached

    def _compute_normC(self) -> float:
        if isinstance(self.C, SymLowRankMatrix):
            return self.C.norm(2)
        if isinstance(self.C, Diagonal):
            return float(np.linalg.norm(self.C.diag))
        if isinstance(self.C, SparseMatrixCOO):
            return float(np.linalg.norm(self.C.toarray())) if self.n <= 4096 else float(
                np.sqrt((sp.coo_matrix((self.C.vs, (self.C.is_, self.C.js)), shape=self.C.shape).tocsc().data ** 2).sum()))
        return float(np.sqrt((sp.csc_matrix(self.C).data ** 2).sum()))


def barvinok_pataki(n: int, m: int) -> int:
    """min(n, ⌊√(2m) + 1⌋)  (src/utils.jl:7-11)."""
    return int(min(n, int(np.floor(np.sqrt(2 * m) + 1))))


@dataclasses.dataclass
class BurerMonteiroConfig:
    """Field names and defaults of the reference's config (src/options.jl:1-24)."""

    ptol: float = 1e-2
    gtol: float = 0.0
    objtol: float = 1e-2
    σ_0: float = 2.0
    σfac: float = 2.0
    maxtime: float = 3600.0
    printlevel: int = 1
    printfreq: float = 60.0
    numlbfgsvecs: int = 4
    maxmajoriter: int = 10 ** 5
    maxiter: int = 10 ** 7
    fprec: float = 1e8
    rankupd_tol: int = 4
    prior_trace_bound: float = 1e18
    dataset: str = ""
    eval_DIMACS_errs: bool = False
    eigval_highprecision: bool = False
    init_func: Optional[Callable[..., Any]] = None
    init_args: tuple = ()
    gtol_mode: str = "relative"
    ptol_mode: str = "relative"
    objtol_mode: str = "relative"
    # not in the reference: seed of the host RNG that replaces Julia's global RNG for
    # Rt0 (src/structs.jl:236) and the Lanczos start vector (src/coreop.jl:473)
    seed: int = 0
