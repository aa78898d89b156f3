"""Shared helpers of the parity tests: dense recomputation of everything the operators produce
(the reference's own checking style, test/coreop.jl:8-16,122-127) and solver construction."""
import itertools

import numpy as np
import scipy.sparse as sp

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import cabi, problems

# the reference's unit-test grid, test/coreop.jl:46-47: seed = position in product([5,8,12],[.4,.7],[2,3])
GRID = [(seed + 1, n, p, r) for seed, (r, p, n) in
        enumerate(itertools.product([2, 3], [0.4, 0.7], [5, 8, 12]))]

FAMILIES_EQ = {
    "maxcut": problems.maxcut,
    "lovasz_theta": problems.lovasz_theta,
    "minimum_bisection": problems.minimum_bisection,
    "cutnorm": problems.cutnorm,
    "mu_conductance_0.01": lambda A: problems.mu_conductance(A, 0.01),
    "mu_conductance_0.05": lambda A: problems.mu_conductance(A, 0.05),
    "mu_conductance_0.1": lambda A: problems.mu_conductance(A, 0.1),
}


def dense(M) -> np.ndarray:
    if sp.issparse(M):
        return M.toarray()
    return M.toarray()


def make_data(family: str, seed: int, n: int, p: float):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    A = problems.make_random_graph(n, p, rng)
    while A.nnz == 0:  # an empty graph has Vol(G) = 0 (μ-conductance bounds become infinite)
        A = problems.make_random_graph(n, p, rng)
    if family.startswith("ineq_"):
        mu = float(family.split("_")[1])
        C, As, bs, ct = problems.mu_conductance_ineq(A, mu)
        return sj.SDPData(C, As, bs, ct), C, As, bs
    C, As, bs = FAMILIES_EQ[family](A)
    return sj.SDPData(C, As, bs), C, As, bs


def primal_vio_dense(C, As, bs, R) -> np.ndarray:
    """[⟨Aᵢ, RRᵀ⟩ − bᵢ ; ⟨C, RRᵀ⟩]  (test/coreop.jl:8-16); R is n×r."""
    X = R @ R.T
    out = np.zeros(len(bs) + 1)
    for i, A in enumerate(As):
        out[i] = np.sum(dense(A) * X) - bs[i]
    out[-1] = np.sum(dense(C) * X)
    return out


def S_dense(C, As, y) -> np.ndarray:
    """Σ yᵢ·Aᵢ + y_{m+1}·C  (test/coreop.jl:122-127)."""
    S = y[-1] * dense(C)
    for i, A in enumerate(As):
        S = S + y[i] * dense(A)
    return S


def make_solver(abi, data, r, seed=0, sigma0=2.0, h=4):
    cfg = sj.BurerMonteiroConfig(σ_0=sigma0, numlbfgsvecs=h, seed=seed, printlevel=0)
    return sj.build_solver(abi, data, r, cfg), cfg


def lagrangian_dense(C, As, bs, R, lam, lam_ub, sigma) -> float:
    """ℒ of src/coreop.jl:6-9 from dense matrices."""
    pv = primal_vio_dense(C, As, bs, R)
    v = pv[:-1]
    yt = np.minimum(lam_ub, lam - sigma * v)
    return pv[-1] + np.sum(yt * yt - lam * lam) / (2 * sigma)
