"""Known-answer tests of the reference (test/maxcut.jl, test/minimumbisection.jl) run through the
Python restatement of _sdplr on the CPU oracle; plus preprocessing, SymLowRank norm, L-BFGS,
Lanczos and quartic checks that pin the remaining oracle pieces.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import cabi, problems
from helpers import S_dense, make_data, make_solver
from oracle import oracle

K2 = sp.csc_matrix(np.array([[0.0, 1.0], [1.0, 0.0]]))


@pytest.mark.parametrize("native", [False, True])
def test_maxcut_k2(oracle_abi, native):
    """test/maxcut.jl:5-25: obj ≈ −1."""
    C, As, bs = problems.maxcut(K2)
    res = sj.sdplr(C, As, bs, 1, fprec=0.0, gtol=1e-8, objtol=1e-8, ptol=1e-8,
                   prior_trace_bound=2.0, abi=oracle_abi, printlevel=0, native_inner_loop=native)
    assert res["obj"] == pytest.approx(-1, rel=1.5e-8)


def test_maxcut_k2_sigma0(oracle_abi):
    """test/maxcut.jl:27-48."""
    C, As, bs = problems.maxcut(K2)
    res = sj.sdplr(C, As, bs, 1, σ_0=10.0, fprec=0.0, gtol=1e-8, objtol=1e-8, ptol=1e-8,
                   prior_trace_bound=2.0, abi=oracle_abi, printlevel=0)
    assert res["obj"] == pytest.approx(-1, rel=1.5e-8)


def test_maxcut_k2_init_func(oracle_abi):
    """test/maxcut.jl:50-76."""
    C, As, bs = problems.maxcut(K2)
    rng = np.random.Generator(np.random.PCG64(7))

    def init_func(data, r, sigma):
        return rng.standard_normal((data.n, r)) * np.sqrt(sigma), np.zeros(data.m)

    res = sj.sdplr(C, As, bs, 1, init_func=init_func, init_args=(10.0,), fprec=0.0, gtol=1e-8,
                   objtol=1e-8, ptol=1e-8, prior_trace_bound=2.0, abi=oracle_abi, printlevel=0)
    assert res["obj"] == pytest.approx(-1, rel=1.5e-8)


def test_minimum_bisection_k2(oracle_abi):
    """test/minimumbisection.jl:3-24."""
    C, As, bs = problems.minimum_bisection(K2)
    res = sj.sdplr(C, As, bs, 1, fprec=0.0, objtol=1e-4, ptol=1e-4, prior_trace_bound=2.0,
                   abi=oracle_abi, printlevel=0)
    assert (res["obj"] - 1) / (1 + abs(res["obj"])) < 1e-4
    assert set(res) >= {"Rt", "lambda", "Rt0", "lambda0", "sigma", "grad_norm", "primal_vio", "obj",
                        "max_dual_value", "min_duality_gap", "totaltime", "dual_time", "primaltime",
                        "iter", "majoriter", "DIMACS_errs", "ptol", "objtol", "fprec",
                        "rankupd_tol", "r", "preprocess_time"}   # src/sdplr.jl:426-448,130


def test_config1_maxcut_n100_r2(oracle_abi):
    """BASELINE.json configs[0]: MaxCut on G(100, 0.1), r = 2, CPU plumbing run to ptol = objtol = 1e-2."""
    A = problems.gnp_graph(100, 0.1, 1)
    C, As, bs = problems.maxcut(A)
    res = sj.sdplr(C, As, bs, 2, prior_trace_bound=100.0, abi=oracle_abi, printlevel=0, seed=1)
    assert res["primal_vio"] <= 1e-2
    # weak duality: the dual bound never exceeds the primal SDP value, which is ≤ the rank-r value
    assert res["max_dual_value"] <= res["obj"] + 1e-6 * abs(res["obj"])
    # MaxCut SDP value lies between |E|/2 and |E|
    nE = A.nnz / 2
    assert nE / 2 <= -res["obj"] <= nE * 1.001


@pytest.mark.parametrize("family", ["maxcut", "lovasz_theta", "minimum_bisection", "cutnorm",
                                    "mu_conductance_0.05", "ineq_0.05"])
def test_preprocess_matches_oracle(family):
    """product preprocessing (vectorised) ≡ the literal restatement of src/preprocess.jl:24-169."""
    for seed, n, p in [(1, 5, 0.4), (2, 12, 0.7), (3, 30, 0.5)]:
        data, *_ = make_data(family, seed, n, p)
        a = sj.preprocess_sparsecons(data.sparse)
        b = oracle.preprocess(data.sparse)
        for k in ("matptr", "nzind", "nzval_one", "nzval_two", "triu_colptr", "triu_rowval",
                  "full_colptr", "full_rowval", "mappedto_triu", "global_inds"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), (family, seed, k)


def test_preprocess_edge_cases():
    """What the vectorised preprocessing must still do at the edges: repeated positions across matrices and unsorted
    entries inside one (≡ the literal restatement), a lower-triangular entry without its mirror (the reference's
    binary search at src/preprocess.jl:143-156 would fail: reported as an error), entries outside the matrix, and a
    problem without sparse matrices."""
    from sdplrplus_jl_amd.structs import SparseBatch
    n = 6
    # matrix 0: unsorted symmetric entries incl. a repeated diagonal position shared with matrix 1; matrix 2: one off-diagonal pair
    I = np.array([4, 1, 1, 0, 3, 3, 1, 2, 5], dtype=np.int64)
    J = np.array([1, 4, 1, 0, 3, 3, 1, 5, 2], dtype=np.int64)
    V = np.array([2.0, 2.0, 1.0, -1.0, 0.5, 0.25, 3.0, 7.0, 7.0])
    ent_ptr = np.array([0, 4, 7, 9], dtype=np.int64)
    batch = SparseBatch(n, ent_ptr, I, J, V, np.array([0, 1, 2], dtype=np.int64))
    a, b = sj.preprocess_sparsecons(batch), oracle.preprocess(batch)
    for k in ("matptr", "nzind", "nzval_one", "nzval_two", "triu_colptr", "triu_rowval",
              "full_colptr", "full_rowval", "mappedto_triu", "global_inds"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    # every full-pattern entry maps to the position of (min, max) in the upper-triangular pattern
    cols = np.repeat(np.arange(n), np.diff(a.full_colptr))
    tcols = np.repeat(np.arange(n), np.diff(a.triu_colptr))
    assert np.array_equal(a.triu_rowval[a.mappedto_triu], np.minimum(a.full_rowval, cols))
    assert np.array_equal(tcols[a.mappedto_triu], np.maximum(a.full_rowval, cols))
    bad = SparseBatch(n, np.array([0, 1], dtype=np.int64), np.array([4]), np.array([1]), np.array([1.0]), np.array([0]))
    with pytest.raises(ValueError, match="not symmetric"):
        sj.preprocess_sparsecons(bad)
    out = SparseBatch(n, np.array([0, 1], dtype=np.int64), np.array([6]), np.array([1]), np.array([1.0]), np.array([0]))
    with pytest.raises(ValueError, match="outside"):
        sj.preprocess_sparsecons(out)
    empty = SparseBatch(n, np.array([0], dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64),
                        np.zeros(0), np.zeros(0, dtype=np.int64))
    e = sj.preprocess_sparsecons(empty)
    assert e.nnzT == 0 and e.nnzS == 0 and e.nnzAgg == 0 and np.array_equal(e.triu_colptr, np.zeros(n + 1, dtype=np.int64))


def test_set_sparse_coo_builds_the_same_layout(oracle_abi):
    """The ABI's set_sparse_coo (preprocess_sparsecons inside the library; here the oracle's literal restatement behind
    the same entry point) hands back, through get_layout, the arrays of the host-side mirror."""
    from sdplrplus_jl_amd import cabi
    for family in ("maxcut", "lovasz_theta", "minimum_bisection", "cutnorm", "mu_conductance_0.05", "ineq_0.05"):
        data, *_ = make_data(family, 2, 12, 0.7)
        a = sj.preprocess_sparsecons(data.sparse)
        s = cabi.DeviceSolver(oracle_abi, data.n, data.m, 2, 4)
        s.set_sparse_coo(data.sparse)
        b = s.get_layout()
        for k in ("matptr", "nzind", "nzval_one", "nzval_two", "triu_colptr", "triu_rowval",
                  "full_colptr", "full_rowval", "mappedto_triu", "global_inds"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), (family, k)
        s.close()


def test_preprocess_accepts_upper_triangular_only_input():
    """A matrix given by its upper triangle alone: the reference's search (src/preprocess.jl:135-156) maps every stored
    entry and never misses a mirror, so this is accepted; a LOWER entry without its upper mirror is the error."""
    from sdplrplus_jl_amd.structs import SparseBatch
    n = 6
    up = SparseBatch(n, np.array([0, 3], dtype=np.int64), np.array([1, 0, 2]), np.array([4, 0, 5]), np.array([1.0, 2.0, 3.0]),
                     np.array([0]))
    a, b = sj.preprocess_sparsecons(up), oracle.preprocess(up)
    for k in ("matptr", "nzind", "nzval_one", "nzval_two", "triu_colptr", "triu_rowval", "full_colptr", "full_rowval",
              "mappedto_triu"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert a.nnzS == a.nnzT == 3 and np.array_equal(a.mappedto_triu, np.arange(3))


def test_batched_builders_match_lists():
    """*_data builders produce the very same batch as the list-based reference builders."""
    A = problems.gnp_graph(60, 0.1, 3)
    for lst, bat in [(problems.maxcut, problems.maxcut_data),
                     (problems.minimum_bisection, problems.minimum_bisection_data),
                     (problems.lovasz_theta, problems.lovasz_theta_data)]:
        C, As, bs = lst(A)
        d1, d2 = sj.SDPData(C, As, bs), bat(A)
        for k in ("ent_ptr", "I", "J", "V", "global_inds"):
            assert np.array_equal(getattr(d1.sparse, k), getattr(d2.sparse, k)), (lst.__name__, k)
        assert np.array_equal(d1.b, d2.b)
        assert [g for g, _ in d1.lowrank] == [g for g, _ in d2.lowrank]
        assert d1.normC() == pytest.approx(d2.normC(), rel=1e-14)


def test_symlowrank_norm():
    """test/symlowrank.jl:6-15."""
    rng = np.random.Generator(np.random.PCG64(11))
    for _ in range(100):
        n, s = int(rng.integers(50, 101)), int(rng.integers(1, 21))
        D, B = rng.standard_normal(s), rng.standard_normal((n, s))
        dense = (B * D) @ B.T
        A = sj.SymLowRankMatrix(D, B)
        for p, ref in [(2, np.linalg.norm(dense)), (np.inf, np.abs(dense).max())]:
            assert oracle.symlowrank_norm(B, D, p) == pytest.approx(ref, rel=1.5e-8)
            assert A.norm(p) == pytest.approx(ref, rel=1.5e-8)


def test_lbfgs_two_loop_is_bfgs_inverse(oracle_abi):
    """src/lbfgs.jl:77-149 against the dense BFGS inverse-Hessian recurrence with H₀ = I."""
    data, *_ = make_data("maxcut", 1, 8, 0.4)
    r, h = 3, 4
    var, _ = make_solver(oracle_abi, data, r, h=h)
    N = data.n * r
    rng = np.random.Generator(np.random.PCG64(5))
    H = np.eye(N)
    g = rng.standard_normal(N)
    for it in range(7):                         # wraps the cyclic buffer (h = 4)
        var.Gt = g.reshape(data.n, r)
        descent = var.lbfgs_dir(negate=True)
        d = var.dirt.ravel()
        assert np.allclose(d, -H @ g, rtol=1e-10, atol=1e-12)
        assert descent == pytest.approx(d @ g, rel=1e-12)
        α = 0.3 + 0.1 * it
        g_new = g + rng.standard_normal(N) * 0.1 + 0.5 * α * d   # keeps yᵀs away from 0
        var.Gt = g_new.reshape(data.n, r)
        var.lbfgs_update(α)
        s_, y_ = α * d, g_new - g
        # dense L-BFGS with memory h: rebuild H from the last ≤ h pairs
        if it == 0:
            pairs = []
        pairs.append((s_, y_))
        pairs = pairs[-h:]
        H = np.eye(N)
        for (ss, yy) in pairs:
            ρ = 1.0 / (yy @ ss)
            V = np.eye(N) - ρ * np.outer(yy, ss)
            H = V.T @ H @ V + ρ * np.outer(ss, ss)
        g = g_new
    var.lbfgs_clear()
    var.Gt = g.reshape(data.n, r)
    var.lbfgs_dir(negate=True)
    assert np.allclose(var.dirt.ravel(), -g)    # cleared history ⇒ steepest descent
    var.close()


def test_lanczos_and_dual_obj(oracle_abi):
    """src/coreop.jl:376-415,461-514 against dense eigenvalues."""
    for family in ["maxcut", "minimum_bisection", "lovasz_theta"]:
        data, C, As, bs = make_data(family, 3, 12, 0.4)
        var, _ = make_solver(oracle_abi, data, 3)
        n = data.n
        var.f()
        rng = np.random.Generator(np.random.PCG64(9))
        v0 = rng.standard_normal(n)
        dual, ev = var.dual_obj(float(n), 0, v0)
        y = var.y
        S = S_dense(C, As, y)
        lam_min = np.linalg.eigvalsh(S)[0]
        # q = n − 1 full Lanczos steps on an n×n matrix resolves the extreme eigenvalue
        assert ev == pytest.approx(lam_min, abs=1e-6 * max(1, abs(lam_min)))
        assert dual == pytest.approx(-(y[:-1] @ bs) + n * min(lam_min, 0.0), rel=1e-6)
        # raw recurrence: T = Qᵀ S Q, first coefficient is the Rayleigh quotient
        al, be, k = var.lanczos(n - 1, v0)
        v = v0 / np.linalg.norm(v0)
        assert al[0] == pytest.approx(v @ S @ v, rel=1e-12)
        assert be[0] == pytest.approx(np.linalg.norm(S @ v - al[0] * v), rel=1e-12)
        T = np.diag(al + 1) + np.diag(be[:k - 1], 1) + np.diag(be[:k - 1], -1)
        assert var.tridiag_mineig(al, be) == pytest.approx(np.linalg.eigvalsh(T)[0] - 1, abs=1e-12)
        var.close()


def test_quartic_argmin_is_global_min_on_grid():
    """scalar stage of linesearch! (src/linesearch.jl:58-112): the returned α minimises the quartic
    over [0, α_max] (checked on a fine grid), and a non-descent slope is refused (:60-62)."""
    rng = np.random.Generator(np.random.PCG64(21))
    grid = np.linspace(0, 1, 20001)
    for _ in range(300):
        bq = rng.standard_normal(5)
        bq[1] = -abs(bq[1])
        bq[4] = abs(bq[4]) * rng.choice([1.0, 1e-3, 0.0])
        if bq[4] == 0.0:
            bq[3] = 0.0 if rng.random() < 0.5 else bq[3]
        rc, a, f = oracle.quartic_argmin(bq, 1.0)
        assert rc == 0 and 0 <= a <= 1
        vals = np.polyval(bq[::-1], grid)
        assert f <= vals.min() + 1e-9 * max(1, abs(vals.min()))
        assert f == pytest.approx(np.polyval(bq[::-1], a), rel=1e-12, abs=1e-12)
    rc, _, _ = oracle.quartic_argmin(np.array([0.0, 1.0, 1.0, 0.0, 1.0]), 1.0)
    assert rc == cabi.ERR_NOT_DESCENT


def _check_eigval_and_dimacs(abi):
    """SDP_S_eigval (src/coreop.jl:351-374) and DIMACS_errors (:426-453) against dense recomputation."""
    for family in ["maxcut", "minimum_bisection", "lovasz_theta"]:
        data, C, As, bs = make_data(family, 4, 30, 0.3)
        var, _ = make_solver(abi, data, 4, seed=1)
        n, m = data.n, data.m
        normC, normb = data.normC(), float(np.linalg.norm(data.b))
        st = var.fg(normC, normb)
        var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 5, 0.0, *st)
        rng = np.random.Generator(np.random.PCG64(3))
        var.λ = rng.standard_normal(m)
        var.f()
        R, λ = var.Rt, var.λ
        S = S_dense(C, As, np.concatenate([-λ, [1.0]]))
        w = np.linalg.eigvalsh(S)
        errs = sj.DIMACS_errors(data, var)
        pv = var.primal_vio_raw
        obj = float(np.sum((C.toarray() @ R) * R))
        λb = float(λ @ bs)
        ref = [np.linalg.norm(pv[:m]) / (1 + normb), 0.0, 0.0, max(0.0, -w[0]) / (1 + normC),
               (obj - λb) / (1 + abs(obj) + abs(λb)), float(np.sum(R * (S @ R))) / (1 + abs(obj) + abs(λb))]
        assert np.allclose(errs, ref, rtol=1e-7, atol=1e-10), (family, errs, ref)
        ev = sj.SDP_S_eigval(var, 2, True, which="SA", ncv=min(20, n), tol=1e-10)
        assert np.allclose(ev, w[:2], rtol=1e-8, atol=1e-9)
        var.close()


def test_eigval_and_dimacs_on_oracle(oracle_abi):
    _check_eigval_and_dimacs(oracle_abi)


def test_highprecision_and_dimacs_options(oracle_abi):
    """config.eigval_highprecision / eval_DIMACS_errs (src/options.jl:17-18) through the driver."""
    A = problems.gnp_graph(40, 0.2, 5)
    C, As, bs = problems.maxcut(A)
    res = sj.sdplr(C, As, bs, 6, abi=oracle_abi, printlevel=0, seed=2, prior_trace_bound=40.0,
                   ptol=1e-3, objtol=1e-3, eigval_highprecision=True, eval_DIMACS_errs=True)
    assert res["primal_vio"] <= 1e-3 and res["min_duality_gap"] <= 1e-3
    e = res["DIMACS_errs"]
    assert e.shape == (6,) and e[1] == 0 and e[2] == 0
    assert abs(e[0]) < 1e-2 and e[3] < 1e-2 and abs(e[4]) < 1e-2 and abs(e[5]) < 1e-2
    # the dual bound from the ARPACK eigenvalue is a valid lower bound on the SDP value
    assert res["max_dual_value"] <= res["obj"] + 1e-6 * abs(res["obj"])


@pytest.mark.parametrize("name", ["relaxed_maxcut", "mu_conductance_reformulated", "mu_conductance_native"])
def test_experiment_builders(oracle_abi, name):
    """the remaining builders of exps/problems.jl (:188-341): the operator identities hold on them too."""
    from helpers import primal_vio_dense
    A = problems.gnp_graph(9, 0.5, 3)
    ct = None
    if name == "relaxed_maxcut":
        C, As, bs = problems.relaxed_maxcut(A)
    elif name == "mu_conductance_reformulated":
        C, As, bs = problems.mu_conductance_reformulated(A, 0.05)
    else:
        C, As, bs, ct = problems.mu_conductance_native(A, 0.05)
    data = sj.SDPData(C, As, bs, ct)
    var, _ = make_solver(oracle_abi, data, 3, seed=4)
    var.f()
    assert np.max(np.abs(var.primal_vio_raw - primal_vio_dense(C, As, bs, var.Rt))) < 1e-10
    var.g()
    assert np.max(np.abs(var.Gt - 2 * S_dense(C, As, var.y) @ var.Rt)) < 1e-10 * (1 + np.max(np.abs(var.Gt)))
    var.close()
