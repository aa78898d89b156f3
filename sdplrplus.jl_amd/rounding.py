"""Randomized rounding callbacks of the reference's experiment runner (exps/test.jl:67-105): they turn the factor R of an
SDP solution into a cut / a bisection.  Host-side post-processing of one downloaded n×r factor (100 GEMVs), not part
of the device path."""
from __future__ import annotations

import numpy as np

from . import problems


def eval_cut(L, x) -> float:
    """Cut value ¼·xᵀLx of a ±1 labelling (exps/test.jl:67-69)."""
    return 0.25 * float(x @ (L @ x))


def maxcut_rounding(A, R: np.ndarray, rng: np.random.Generator, trials: int = 100) -> float:
    """Goemans–Williamson hyperplane rounding, best of `trials` (exps/test.jl:71-81).  R is n×r."""
    L = problems._laplacian(A, 1.0)
    best = -np.inf
    for _ in range(trials):
        x = np.sign(R @ rng.standard_normal(R.shape[1]))
        x[x == 0] = 1.0
        best = max(best, eval_cut(L, x))
    return best


def minimumbisection_rounding(A, R: np.ndarray, rng: np.random.Generator, trials: int = 100) -> float:
    """Sort a random projection and split it in halves, best of `trials` (exps/test.jl:83-98)."""
    L = problems._laplacian(A, 1.0)
    n = R.shape[0]
    best = np.inf
    for _ in range(trials):
        perm = np.argsort(R @ rng.standard_normal(R.shape[1]), kind="stable")
        part = np.zeros(n)
        part[perm] = np.where((np.arange(1, n + 1) * 2) <= n, 1.0, -1.0)
        best = min(best, eval_cut(L, part))
    return best
