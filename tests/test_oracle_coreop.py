"""Pins the CPU oracle with the reference's own unit-test identities (test/coreop.jl), restated
against dense numpy recomputation.  CPU only."""
import numpy as np
import pytest

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import cabi
from helpers import (FAMILIES_EQ, GRID, S_dense, lagrangian_dense, make_data, make_solver,
                     primal_vio_dense)


def fd_gradient(fun, x, h=1e-6):
    g = np.zeros_like(x)
    for i in range(x.size):
        e = np.zeros_like(x); e[i] = h
        g[i] = (fun(x + e) - fun(x - e)) / (2 * h)
    return g


@pytest.mark.parametrize("family", list(FAMILIES_EQ))
@pytest.mark.parametrize("seed,n,p,r", GRID)
def test_f_g_linesearch(oracle_abi, family, seed, n, p, r):
    """test/coreop.jl:35-77."""
    data, C, As, bs = make_data(family, seed, n, p)
    var, _ = make_solver(oracle_abi, data, r, seed=seed)
    R = var.Rt
    L = var.f()
    assert np.max(np.abs(var.primal_vio_raw - primal_vio_dense(C, As, bs, R))) < 1e-10   # :58-61
    lam, lam_ub = var.λ, var.λ_ub
    assert abs(L - lagrangian_dense(C, As, bs, R, lam, lam_ub, 2.0)) < 1e-9 * max(1, abs(L))

    # test_gradient_fd!, test/coreop.jl:19-32 (central differences of the dense Lagrangian)
    var.g()
    G = var.Gt
    fun = lambda x: lagrangian_dense(C, As, bs, x.reshape(R.shape), lam, lam_ub, 2.0)
    Gnum = fd_gradient(fun, R.ravel().copy()).reshape(R.shape)
    assert np.max(np.abs(Gnum - G)) / (1 + np.max(np.abs(G))) < 1e-6
    # and against the closed form 2·S·R with S = C − Σ ỹᵢAᵢ (the same identity at 1e-10)
    S = S_dense(C, As, var.y)
    assert np.max(np.abs(G - 2 * S @ R)) < 1e-10 * (1 + np.max(np.abs(G)))

    # line search along −G keeps primal_vio_raw in sync, test/coreop.jl:65-72
    var.dirt = -G
    α, Lval = var.linesearch(1.0)
    var.axpy_R(α)
    Rn = var.Rt
    assert np.allclose(Rn, R + α * (-G), rtol=0, atol=1e-13)
    assert np.max(np.abs(var.primal_vio_raw - primal_vio_dense(C, As, bs, Rn))) < 1e-10
    # the returned value is the Lagrangian at the new point and does not exceed ℒ(0)
    assert abs(Lval - lagrangian_dense(C, As, bs, Rn, lam, lam_ub, 2.0)) < 1e-9 * max(1, abs(Lval))
    assert Lval <= L + 1e-12 * max(1, abs(L))
    var.close()


@pytest.mark.parametrize("mu", [0.01, 0.05, 0.1])
@pytest.mark.parametrize("seed,n,p,r", GRID)
def test_inequalities(oracle_abi, mu, seed, n, p, r):
    """test/coreop.jl:92-119."""
    data, C, As, bs = make_data(f"ineq_{mu}", seed, n, p)
    var, _ = make_solver(oracle_abi, data, r, seed=seed)
    R = var.Rt
    var.f()
    pv = primal_vio_dense(C, As, bs, R)
    assert np.max(np.abs(var.primal_vio_raw - pv)) < 1e-10
    cap = np.where(data.constraint_types, np.maximum(pv[:-1], 0.0), pv[:-1])     # :80-88
    assert np.max(np.abs(var.primal_vio - cap)) < 1e-10
    var.g()
    G = var.Gt
    lam, lam_ub = var.λ, var.λ_ub
    fun = lambda x: lagrangian_dense(C, As, bs, x.reshape(R.shape), lam, lam_ub, 2.0)
    Gnum = fd_gradient(fun, R.ravel().copy()).reshape(R.shape)
    assert np.max(np.abs(Gnum - G)) / (1 + np.max(np.abs(G))) < 1e-6
    # Armijo line search bookkeeping (src/linesearch.jl:139-191)
    var.dirt = -G
    α, Lα = var.linesearch_armijo(1.0)
    var.axpy_R(α)
    Rn = var.Rt
    assert np.max(np.abs(var.primal_vio_raw - primal_vio_dense(C, As, bs, Rn))) < 1e-10
    assert abs(Lα - lagrangian_dense(C, As, bs, Rn, lam, lam_ub, 2.0)) < 1e-9 * max(1, abs(Lα))
    var.close()


AT_FAMILIES = ["maxcut", "lovasz_theta", "minimum_bisection", "mu_conductance_0.01",
               "mu_conductance_0.05", "mu_conductance_0.1", "ineq_0.01", "ineq_0.05", "ineq_0.1"]


@pytest.mark.parametrize("family", AT_FAMILIES)
@pytest.mark.parametrize("seed,n,p,r", GRID)
def test_At(oracle_abi, family, seed, n, p, r):
    """test/coreop.jl:130-214: both orientations of 𝒜t! against the dense S."""
    data, C, As, bs = make_data(family, seed, n, p)
    var, _ = make_solver(oracle_abi, data, r, seed=seed)
    N = data.n
    var.f()
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    y = rng.standard_normal(data.m + 1)
    var.y = y
    var.At_preprocess()
    S = S_dense(C, As, y)
    R = var.Rt
    var.At_left(cabi.F_GT, cabi.F_RT)
    assert np.max(np.abs(var.Gt - S @ R)) < 1e-10          # y_left = Rt·S  ⇔  (S·R) in n×r view
    x = rng.standard_normal((N, r))
    assert np.max(np.abs(var.At_right(x) - S @ x)) < 1e-10
    xv = rng.standard_normal(N)
    assert np.max(np.abs(var.At_right(xv) - S @ xv)) < 1e-10
    var.close()
