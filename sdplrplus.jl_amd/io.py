"""On-disk formats either side of the path: the SDPA and SDPLR-1.03 problem writers and
the SDPLR initial-solution writer of the reference's exps/data_utils.jl:21-153, plus plain edge-list / Gset
readers.  Indices in the files are 1-based, as the formats require."""
from __future__ import annotations

import os
from typing import Sequence

import numpy as np
import scipy.sparse as sp

from .structs import Diagonal, SparseMatrixCOO, SymLowRankMatrix


def _triu_entries(A):
    """(i, j, v) of triu(A) in the order Julia's findnz(triu(A)) yields them (column-major for CSC, stored
    order for COO), 1-based."""
    if isinstance(A, SparseMatrixCOO):
        keep = A.is_ <= A.js
        return A.is_[keep] + 1, A.js[keep] + 1, A.vs[keep]
    if isinstance(A, Diagonal):
        idx = np.arange(1, A.diag.size + 1)
        return idx, idx, A.diag
    T = sp.csc_matrix(sp.triu(sp.csc_matrix(A)))
    T.sort_indices()
    cols = np.repeat(np.arange(T.shape[1]), np.diff(T.indptr))
    return T.indices + 1, cols + 1, T.data


def write_problem_sdpa(path: str, C, As: Sequence, bs) -> None:
    """``write_problem_sdpa`` (exps/data_utils.jl:21-51): SDPA sparse format, one block, cost negated."""
    n, m = C.shape[0], len(As)
    with open(path, "w") as f:
        f.write(f"{m}\n1\n{n}\n")
        f.write("".join(f"{float(b)!r} " for b in bs) + "\n")
        for i, j, v in zip(*_triu_entries(C)):
            f.write(f"0 1 {i} {j} {-float(v)!r}\n")
        for k, A in enumerate(As, start=1):
            for i, j, v in zip(*_triu_entries(A)):
                f.write(f"{k} 1 {i} {j} {float(v)!r}\n")


def _write_matrix_sdplr(A, idx: int, f) -> None:
    """``write_matrix_sdplr`` (exps/data_utils.jl:53-86): 's' sparse blocks, 'l' low-rank blocks."""
    if isinstance(A, SymLowRankMatrix):
        f.write(f"{idx} 1 l {A.B.shape[1]}\n")
        for d in A.D:
            f.write(f"{float(d)!r}\n")
        for col in range(A.B.shape[1]):      # B in column-major order
            for v in A.B[:, col]:
                f.write(f"{float(v)!r}\n")
        return
    i, j, v = _triu_entries(A)
    f.write(f"{idx} 1 s {len(v)}\n")
    for a, b, c in zip(i, j, v):
        f.write(f"{a} {b} {float(c)!r}\n")


def write_problem_sdplr(path: str, C, As: Sequence, bs) -> None:
    """``write_problem_sdplr`` (exps/data_utils.jl:88-120)."""
    n, m = C.shape[0], len(As)
    with open(path, "w") as f:
        f.write(f"{m}\n1\n{n}\n")
        f.write("".join(f"{float(b)!r} " for b in bs) + "\n")
        f.write("1\n")                       # ignored by SDPLR
        _write_matrix_sdplr(C, 0, f)
        for k, A in enumerate(As, start=1):
            _write_matrix_sdplr(A, k, f)


def write_initial_solution(path: str, R: np.ndarray, lam: np.ndarray) -> None:
    """``write_initial_solution`` (exps/data_utils.jl:122-153): R is n×r, written column-major."""
    n, r = R.shape
    with open(path, "w") as f:
        f.write(f"dual variable {len(lam)}\n")
        for v in lam:
            f.write(f"{float(v)!r}\n")
        f.write(f"primal variable 1 s {n} {r} {r}\n")
        for j in range(r):
            for v in R[:, j]:
                f.write(f"{float(v)!r}\n")
        f.write("special majiter 0\nspecial iter 0\nspecial lambdaupdate 0")   # (sic: no newline in the reference)
        f.write("special CG 0\nspecial curr_CG 0\nspecial totaltime 0\n")
        f.write(f"special sigma {1.0 / n!r}\nspecial scale 1.0\n")


def read_sdpa(path: str):
    """Reader for the files ``write_problem_sdpa`` produces → (C, As, bs) with scipy CSC matrices."""
    with open(path) as f:
        m = int(f.readline()); nblocks = int(f.readline()); n = int(f.readline().split()[0])
        assert nblocks == 1
        bs = np.array([float(t) for t in f.readline().split()])
        rows = [[] for _ in range(m + 1)]; cols = [[] for _ in range(m + 1)]; vals = [[] for _ in range(m + 1)]
        for line in f:
            t = line.split()
            if not t:
                continue
            k, _, i, j, v = int(t[0]), int(t[1]), int(t[2]) - 1, int(t[3]) - 1, float(t[4])
            rows[k] += [i] if i == j else [i, j]
            cols[k] += [j] if i == j else [j, i]
            vals[k] += [v] if i == j else [v, v]
    mats = [sp.csc_matrix((vals[k], (rows[k], cols[k])), shape=(n, n)) for k in range(m + 1)]
    return -mats[0], mats[1:], bs


def read_edge_list(path: str, one_based: bool = True, gset_header: bool = False) -> sp.csc_matrix:
    """Undirected graph from a whitespace edge list ``i j [w]`` (SNAP style: '#' comments; Gset style: a
    ``n m`` header line).  Self-loops are dropped and the adjacency is symmetrised
    (exps/data_preprocess.jl:118-134)."""
    I, J, W = [], [], []
    n = 0
    with open(path) as f:
        first = True
        for line in f:
            if line.startswith(("#", "%")) or not line.strip():
                continue
            t = line.split()
            if first and gset_header:
                n = int(t[0]); first = False
                continue
            first = False
            i, j = int(t[0]) - one_based, int(t[1]) - one_based
            if i == j:
                continue
            I.append(i); J.append(j); W.append(float(t[2]) if len(t) > 2 else 1.0)
    n = max(n, max(I + J) + 1 if I else 0)
    A = sp.coo_matrix((W + W, (I + J, J + I)), shape=(n, n)).tocsc()
    A.data[:] = np.where(A.data != 0, A.data, 0)
    A.sort_indices()
    return A
