"""GPU parity tests (run on the MI355X box with -m gpu): the HIP library against the CPU oracle on
identical seeded inputs, through the C ABI, and against dense / scipy recomputation.

Tolerances (FP64): operator identities 1e-10 (the reference's own bar, test/coreop.jl:59-72);
‖grad‖ and objective 1e-8 relative (BASELINE.json north_star)."""
import numpy as np
import pytest
import scipy.sparse as sp

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import cabi, problems
from helpers import (FAMILIES_EQ, GRID, S_dense, lagrangian_dense, make_data, make_solver,
                     primal_vio_dense)

pytestmark = pytest.mark.gpu

ALL_FAMILIES = list(FAMILIES_EQ) + ["ineq_0.01", "ineq_0.05", "ineq_0.1"]
# a slice of the reference grid (test/coreop.jl:46-47) plus ranks that exercise every sub-wave shape
CASES = [(1, 5, 0.4, 2), (4, 5, 0.7, 2), (8, 8, 0.4, 3), (12, 12, 0.7, 3), (13, 12, 0.4, 1),
         (14, 12, 0.4, 10), (15, 9, 0.4, 32), (16, 7, 0.4, 33)]


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b)))) if a.size else 0.0


def pair(hip_abi, oracle_abi, data, r, seed, h=4):
    g, _ = make_solver(hip_abi, data, r, seed=seed, h=h)
    o, _ = make_solver(oracle_abi, data, r, seed=seed, h=h)
    assert np.array_equal(g.Rt, o.Rt)
    return g, o


@pytest.mark.parametrize("family", ALL_FAMILIES)
@pytest.mark.parametrize("seed,n,p,r", CASES)
def test_operators_match_oracle_and_dense(hip_abi, oracle_abi, family, seed, n, p, r):
    data, C, As, bs = make_data(family, seed, n, p)
    g, o = pair(hip_abi, oracle_abi, data, r, seed)
    R = g.Rt
    # ---- f! ------------------------------------------------------------------------------------
    Lg, Lo = g.f(), o.f()
    assert np.max(np.abs(g.primal_vio_raw - primal_vio_dense(C, As, bs, R))) < 1e-10
    assert rel(g.primal_vio_raw, o.primal_vio_raw) < 1e-12
    assert rel(g.primal_vio, o.primal_vio) < 1e-12
    assert abs(Lg - Lo) <= 1e-12 * max(1, abs(Lo)) and abs(g.obj - o.obj) <= 1e-12 * max(1, abs(o.obj))
    # ---- g! ------------------------------------------------------------------------------------
    g.g(); o.g()
    assert rel(g.y, o.y) < 1e-13
    assert rel(g.get_vec(cabi.V_TRIU_S_NZVAL), o.get_vec(cabi.V_TRIU_S_NZVAL)) < 1e-13
    assert rel(g.get_vec(cabi.V_S_NZVAL), o.get_vec(cabi.V_S_NZVAL)) < 1e-13
    G = g.Gt
    assert rel(G, o.Gt) < 1e-12
    assert np.max(np.abs(G - 2 * S_dense(C, As, g.y) @ R)) < 1e-10 * (1 + np.max(np.abs(G)))
    gn_g, pn_g = g.norms(3.0, 2.0, True, True)
    gn_o, pn_o = o.norms(3.0, 2.0, True, True)
    assert gn_g == pytest.approx(gn_o, rel=1e-12) and pn_g == pytest.approx(pn_o, rel=1e-12, abs=1e-300)
    assert gn_g == pytest.approx(np.linalg.norm(G) / 3.0, rel=1e-12)
    # ---- 𝒜 one- and two-argument on arbitrary slots ------------------------------------------------
    g.dirt = -G; o.dirt = -G
    for out in (cabi.V_A_RD, cabi.V_A_DD):
        g.A(cabi.F_RT, cabi.F_DIRT, out); o.A(cabi.F_RT, cabi.F_DIRT, out)
        assert rel(g.get_vec(out), o.get_vec(out)) < 1e-12
    X = (R @ (-G).T + (-G) @ R.T) / 2
    ref = np.array([np.sum(A_.toarray() * X) for A_ in As] + [np.sum(C.toarray() * X)])
    assert np.max(np.abs(g.get_vec(cabi.V_A_DD) - ref)) < 1e-10 * (1 + np.max(np.abs(ref)))
    # ---- line search ------------------------------------------------------------------------------
    ineq = family.startswith("ineq")
    if ineq:
        (ag, Lg2), (ao, Lo2) = g.linesearch_armijo(1.0), o.linesearch_armijo(1.0)
        assert ag == ao
    else:
        (ag, Lg2), (ao, Lo2) = g.linesearch(1.0), o.linesearch(1.0)
        assert abs(ag - ao) <= 1e-7 * max(1.0, abs(ao))
    assert abs(Lg2 - Lo2) <= 1e-10 * max(1, abs(Lo2))
    assert rel(g.A_RD, o.A_RD) < 1e-12 and rel(g.A_DD, o.A_DD) < 1e-12
    g.axpy_R(ag)
    Rn = g.Rt
    assert np.max(np.abs(g.primal_vio_raw - primal_vio_dense(C, As, bs, Rn))) < 1e-10   # test/coreop.jl:70-72
    assert abs(Lg2 - lagrangian_dense(C, As, bs, Rn, g.λ, g.λ_ub, 2.0)) < 1e-9 * max(1, abs(Lg2))
    # ---- 𝒜t! both orientations with a random y (test/coreop.jl:156-172) ---------------------------
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    y = rng.standard_normal(data.m + 1)
    g.y = y
    g.At_preprocess()
    S = S_dense(C, As, y)
    g.At_left(cabi.F_GT, cabi.F_RT)
    assert np.max(np.abs(g.Gt - S @ Rn)) < 1e-10
    x = rng.standard_normal((data.n, 3))
    assert np.max(np.abs(g.At_right(x) - S @ x)) < 1e-10
    g.close(); o.close()


@pytest.mark.parametrize("h", [0, 1, 2, 3, 4, 6, 9, 17, 20])     # (> 16: the literal two-loop route, k_lit_*)
@pytest.mark.parametrize("r", [2, 3, 32])
def test_lbfgs_matches_oracle(hip_abi, oracle_abi, h, r):
    """lbfgs_dir!/lbfgs_update!/lbfgs_clear! (src/lbfgs.jl) incl. the cyclic wrap and host-written slots."""
    data, *_ = make_data("maxcut", 1, 9, 0.4)
    g, o = pair(hip_abi, oracle_abi, data, r, 3, h=h)
    rng = np.random.Generator(np.random.PCG64(5))
    N = (data.n, r)
    grad = rng.standard_normal(N)
    for it in range(2 * h + 3):
        g.Gt = grad; o.Gt = grad
        dg, do = g.lbfgs_dir(True), o.lbfgs_dir(True)
        assert rel(g.dirt, o.dirt) < 1e-10
        assert dg == pytest.approx(do, rel=1e-9)
        α = 0.3 + 0.05 * it
        grad = grad + 0.1 * rng.standard_normal(N) + 0.5 * α * o.dirt
        g.Gt = grad; o.Gt = grad
        g.lbfgs_update(α); o.lbfgs_update(α)
        assert rel(g.dirt, o.dirt) < 1e-12
        for j in range(h):
            assert rel(g.get_factor(cabi.F_LBFGS_S + j), o.get_factor(cabi.F_LBFGS_S + j)) < 1e-12
            assert rel(g.get_factor(cabi.F_LBFGS_Y + j), o.get_factor(cabi.F_LBFGS_Y + j)) < 1e-12
        if h:
            assert np.allclose(g.get_vec(cabi.V_LBFGS_RHO), o.get_vec(cabi.V_LBFGS_RHO), rtol=1e-10)
            assert g.get_scalar(cabi.S_LBFGS_LATEST) == o.get_scalar(cabi.S_LBFGS_LATEST)
    if h:
        # history written from the host behind the library's back: Gram data must be rebuilt
        Snew = rng.standard_normal(N)
        g.set_factor(cabi.F_LBFGS_S + 0, Snew); o.set_factor(cabi.F_LBFGS_S + 0, Snew)
        g.lbfgs_dir(False); o.lbfgs_dir(False)
        assert rel(g.dirt, o.dirt) < 1e-10
        assert np.allclose(g.get_vec(cabi.V_LBFGS_A), o.get_vec(cabi.V_LBFGS_A), rtol=1e-9, atol=1e-12)
    g.lbfgs_clear(); o.lbfgs_clear()
    g.Gt = grad; o.Gt = grad
    g.lbfgs_dir(True)
    # h = 0: the reference returns before negating (src/lbfgs.jl:88-91) and relies on the fallback
    assert np.array_equal(g.dirt, -grad if h else grad)
    g.descent_fallback()
    assert np.array_equal(g.Gt, -grad) and np.array_equal(g.dirt, -grad)
    g.close(); o.close()


@pytest.mark.parametrize("family", ["maxcut", "minimum_bisection", "lovasz_theta", "cutnorm",
                                    "mu_conductance_0.05", "ineq_0.05"])
def test_inner_loop_trajectory(hip_abi, oracle_abi, family):
    """The native inner loop (src/sdplr.jl:190-278) against the oracle's, iteration by iteration, and
    against the op-by-op path: same ℒ, ‖grad‖, ‖pv‖ to 1e-8 over a short fixed-length run."""
    data, C, As, bs = make_data(family, 2, 12, 0.4)
    r = 3
    g, o = pair(hip_abi, oracle_abi, data, r, 11)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    armijo = data.has_inequalities
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    assert np.allclose(sg, so, rtol=1e-11)
    for it in range(8):
        rg = g.inner_loop(normC, normb, True, True, armijo, 0.0, -1e300, 1, 0.0, *sg)
        ro = o.inner_loop(normC, normb, True, True, armijo, 0.0, -1e300, 1, 0.0, *so)
        assert rg[4] == ro[4] == 1 and rg[5] == ro[5] == 2
        assert np.allclose(rg[:3], ro[:3], rtol=1e-8, atol=1e-12), (it, rg, ro)
        assert rel(g.Rt, o.Rt) < 1e-8
        # dirt left as lbfgs_update! leaves it (dirt *= α, src/lbfgs.jl:142) — the fused step kernel skips that
        # store inside the loop and the library restores it from s_latest on the way out
        assert rel(g.dirt, o.dirt) < 1e-7
        sg, so = rg[:3], ro[:3]
    # many iterations in one call, early exit by the gradient test
    g2, o2 = pair(hip_abi, oracle_abi, data, r, 11)
    sg, so = g2.fg(normC, normb), o2.fg(normC, normb)
    rg = g2.inner_loop(normC, normb, True, True, armijo, 0.3 * sg[1], -1e300, 500, 0.0, *sg)
    ro = o2.inner_loop(normC, normb, True, True, armijo, 0.3 * so[1], -1e300, 500, 0.0, *so)
    assert rg[4] == ro[4] and rg[5] == ro[5] == 0
    assert np.allclose(rg[:3], ro[:3], rtol=1e-6)
    assert rel(g2.dirt, o2.dirt) < 1e-5
    # iteration budget exit
    rg = g2.inner_loop(normC, normb, True, True, armijo, 0.0, -1e300, 5, 0.0, *rg[:3])
    assert rg[4] == 5 and rg[5] == 2
    # relative-decrease exit with a huge fprec (always fires after the first step)
    rg = g2.inner_loop(normC, normb, True, True, armijo, 0.0, 1e300, 50, 0.0, *rg[:3])
    assert rg[4] == 1 and rg[5] == 1
    ro = o2.inner_loop(normC, normb, True, True, armijo, 0.0, -1e300, 5, 0.0, *ro[:3])
    ro = o2.inner_loop(normC, normb, True, True, armijo, 0.0, 1e300, 50, 0.0, *ro[:3])
    assert ro[4] == 1 and ro[5] == 1 and rel(g2.dirt, o2.dirt) < 1e-5   # no lbfgs_update! before this exit: dirt unscaled
    # the state that exit leaves behind (history with y_next = −G_old parked, G rewritten by g!) must carry on
    rg2 = g2.inner_loop(normC, normb, True, True, armijo, 0.0, -1e300, 4, 0.0, *rg[:3])
    ro2 = o2.inner_loop(normC, normb, True, True, armijo, 0.0, -1e300, 4, 0.0, *ro[:3])
    assert rg2[4] == ro2[4] == 4 and np.allclose(rg2[:3], ro2[:3], rtol=1e-5) and rel(g2.Rt, o2.Rt) < 1e-5
    for s_ in (g, o, g2, o2):
        s_.close()


@pytest.mark.parametrize("name", ["relaxed_maxcut", "mu_conductance_reformulated", "mu_conductance_native"])
@pytest.mark.parametrize("r", [3, 32])
def test_experiment_builders_on_gpu(hip_abi, oracle_abi, name, r):
    """The remaining builders of exps/problems.jl (:188-341) on the device: two-entry diagonal matrices over a 3n×3n
    slack layout, a full diagonal matrix beside the general one, two singleton constraints per row with inequalities
    (other classifications of the structured path than the five BASELINE families take).  Operators against dense
    recomputation and the oracle, then the native inner loop against the oracle's."""
    A = problems.gnp_graph(14, 0.4, 3)
    ct = None
    if name == "relaxed_maxcut":
        C, As, bs = problems.relaxed_maxcut(A)
    elif name == "mu_conductance_reformulated":
        C, As, bs = problems.mu_conductance_reformulated(A, 0.05)
    else:
        C, As, bs, ct = problems.mu_conductance_native(A, 0.05)
    data = sj.SDPData(C, As, bs, ct)
    g, o = pair(hip_abi, oracle_abi, data, r, 4)
    Lg, Lo = g.f(), o.f()
    assert np.max(np.abs(g.primal_vio_raw - primal_vio_dense(C, As, bs, g.Rt))) < 1e-10
    assert rel(g.primal_vio_raw, o.primal_vio_raw) < 1e-12 and abs(Lg - Lo) <= 1e-12 * max(1, abs(Lo))
    g.g(); o.g()
    assert rel(g.y, o.y) < 1e-13 and rel(g.Gt, o.Gt) < 1e-12
    assert np.max(np.abs(g.Gt - 2 * S_dense(C, As, g.y) @ g.Rt)) < 1e-10 * (1 + np.max(np.abs(g.Gt)))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    armijo = data.has_inequalities
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    for it in range(6):
        rg = g.inner_loop(normC, normb, True, True, armijo, 0.0, -1e300, 1, 0.0, *sg)
        ro = o.inner_loop(normC, normb, True, True, armijo, 0.0, -1e300, 1, 0.0, *so)
        assert rg[4] == ro[4] == 1
        assert np.allclose(rg[:3], ro[:3], rtol=1e-7, atol=1e-11), (it, rg, ro)
        assert rel(g.Rt, o.Rt) < 1e-7
        sg, so = rg[:3], ro[:3]
    g.close(); o.close()


def test_armijo_decision_at_the_bound(hip_abi, oracle_abi):
    """Which step the backtracking accepts when ℒ(α) sits at the Armijo bound (src/linesearch.jl:173-181).
    (1) an exact tie — dirt = 0 makes ℒ(α) = ℒ(0) for every α and the slope 0 — is accepted at once (`≤`, :177),
    by construction of the sums and not by luck of rounding; (2) the direction is scaled until the oracle's decision
    flips between two neighbouring scales (bisection to 1e-12), and 1e-7 to either side of that flip the device picks
    the oracle's step: the windows in which a different summation order may legitimately decide otherwise are
    ~1e-13 wide."""
    data, *_ = make_data("ineq_0.05", 3, 12, 0.4)
    g, o = pair(hip_abi, oracle_abi, data, 3, 5)
    R0 = g.Rt.copy()
    for s_ in (g, o):
        s_.f(); s_.g()
    D0 = -o.Gt

    def arm(s_, t):
        s_.Rt = R0
        s_.f(); s_.g()            # primal_vio_raw, obj and the y the slope uses (:171)
        s_.dirt = t * D0
        return s_.linesearch_armijo(1.0)

    (ag, Lg), (ao, Lo) = arm(g, 0.0), arm(o, 0.0)
    assert ag == ao == 1.0 and Lg == pytest.approx(Lo, rel=1e-13)
    ts = [2.0 ** k for k in range(-12, 13)]
    al = [arm(o, t)[0] for t in ts]
    flips = [i for i in range(len(ts) - 1) if al[i] != al[i + 1]]
    assert len(flips) >= 2, al    # the sweep crosses several backtracking counts
    checked = 0
    for i in flips[:3]:
        lo, hi = ts[i], ts[i + 1]
        a_lo = al[i]
        while hi / lo - 1.0 > 1e-12:
            mid = 0.5 * (lo + hi)
            if arm(o, mid)[0] == a_lo:
                lo = mid
            else:
                hi = mid
        for t in (lo * (1 - 1e-7), hi * (1 + 1e-7)):
            (ag, Lg), (ao, Lo) = arm(g, t), arm(o, t)
            assert ag == ao, (t, ag, ao)
            assert Lg == pytest.approx(Lo, rel=1e-11)
        assert arm(o, lo * (1 - 1e-7))[0] != arm(o, hi * (1 + 1e-7))[0]
        checked += 1
    assert checked >= 2
    g.close(); o.close()


@pytest.mark.parametrize("family", ["maxcut", "lovasz_theta"])
@pytest.mark.parametrize("h", [1, 2, 3, 5, 8])
def test_inner_loop_trajectory_over_history_lengths(hip_abi, oracle_abi, family, h):
    """numlbfgsvecs ≠ 4 inside the device-driven loop: the seam kernel's register form (h ≤ 4) and its general form,
    through several wraps of the cyclic history (src/lbfgs.jl:77-149), one call of 3h + 2 iterations."""
    data, *_ = make_data(family, 3, 12, 0.4)
    g, o = pair(hip_abi, oracle_abi, data, 3, 7, h=h)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    k = 3 * h + 2
    arm = data.has_inequalities
    if arm:
        k = h + 3          # (Armijo: a backtracking decision can flip on round-off further out)
    rg = g.inner_loop(normC, normb, True, True, arm, 0.0, -1e300, k, 0.0, *sg)
    ro = o.inner_loop(normC, normb, True, True, arm, 0.0, -1e300, k, 0.0, *so)
    assert rg[4] == ro[4] == k
    assert np.allclose(rg[:3], ro[:3], rtol=1e-6 if arm else 1e-7, atol=1e-12), (rg, ro)
    assert rel(g.Rt, o.Rt) < (1e-5 if arm else 1e-7)
    assert g.get_scalar(cabi.S_LBFGS_LATEST) == o.get_scalar(cabi.S_LBFGS_LATEST)
    assert np.allclose(g.get_vec(cabi.V_LBFGS_RHO), o.get_vec(cabi.V_LBFGS_RHO), rtol=1e-6)
    g.close(); o.close()


@pytest.mark.parametrize("family", ["maxcut", "minimum_bisection", "lovasz_theta", "mu_conductance_0.05", "ineq_0.05"])
def test_iteration_paths_are_run_to_run_deterministic(hip_abi, family):
    """No float atomics on global memory, LDS adds in issue order on lane-private addresses, fixed-order block and
    grid reductions: three runs of the same 20 iterations give bit-identical R, G and ℒ on every iteration path
    (singleton / rank-1 / non-singleton fast paths, generic path, Armijo)."""
    data, C, As, bs = make_data(family, 5, 120, 0.08)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    seen = set()
    for rep in range(3):
        g, _ = make_solver(hip_abi, data, 16, seed=2)
        st = g.fg(normC, normb)
        out = g.inner_loop(normC, normb, True, True, data.has_inequalities, 0.0, -1e300, 20, 0.0, *st)
        seen.add((g.Rt.tobytes(), g.Gt.tobytes(), out[0]))
        g.close()
    assert len(seen) == 1


@pytest.mark.parametrize("family", ["maxcut", "minimum_bisection", "lovasz_theta"])
def test_inner_loop_time_budget_exit_leaves_a_consistent_state(hip_abi, oracle_abi, family):
    """The host's time budget (src/sdplr.jl:299) stops the loop between two batches, i.e. after a whole number
    of iterations, with the last lbfgs_update!'s Gram partials still unfolded on the device: the library folds
    them on the way out.  However many iterations it ran, an oracle run of the same count must be at the same
    point, and both must carry on identically."""
    from sdplrplus_jl_amd import cabi
    big = family == "maxcut"    # large enough for the update's and the fused step's grids to differ
    if big:   # sparse enough to stay on the singleton fast path (no hub rows)
        data = problems.maxcut_data(problems.gnp_graph(1500, 0.01, 8))
    else:
        data, C, As, bs = make_data(family, 8, 16, 0.4)
    g, o = pair(hip_abi, oracle_abi, data, 8 if big else 3, 31)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    # a first call captures the hipGraph; rewriting G from the host then makes the next call rebuild a Gram row
    # through the stand-alone update kernel (another grid) before it REPLAYS the graph
    sg = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 8, 0.0, *sg)[:3]
    so = o.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 8, 0.0, *so)[:3]
    g.set_factor(cabi.F_GT, g.Gt)
    o.set_factor(cabi.F_GT, o.Gt)
    rg = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 10_000, 1e-9, *sg)   # budget: stop at once
    assert rg[5] == 3 and 0 < rg[4] < 10_000, rg          # EXIT_TIME after at least one batch
    ro = o.inner_loop(normC, normb, True, True, False, 0.0, -1e300, rg[4], 0.0, *so)
    assert np.allclose(rg[:3], ro[:3], rtol=1e-6) and rel(g.Rt, o.Rt) < 1e-6
    rg2 = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 5, 0.0, *rg[:3])
    ro2 = o.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 5, 0.0, *ro[:3])
    assert rg2[4] == ro2[4] == 5 and np.allclose(rg2[:3], ro2[:3], rtol=1e-5) and rel(g.Rt, o.Rt) < 1e-5
    g.close(); o.close()


@pytest.mark.parametrize("family,h", [("maxcut", 0), ("maxcut", 1), ("maxcut", 2), ("maxcut", 4), ("maxcut", 6),
                                      ("cutnorm", 4), ("minimum_bisection", 3), ("lovasz_theta", 2)])
def test_inner_loop_takes_the_steepest_descent_fallback(hip_abi, oracle_abi, family, h):
    """src/sdplr.jl:201-205: when ⟨dir, G⟩ is NaN or ≥ 0 the loop replaces the direction by steepest descent
    (G ← −G; dir ← G).  Inside the device loop that test is evaluated by the seam kernel from the Gram data
    and applied by k_lbfgs_dir itself; the oracle reduces the dot product as the reference does.  Cases: no
    history at all (h = 0: lbfgs_dir! returns +G, every iteration falls back) and a history whose ρ have
    newest pair the host has replaced by (s = G, y = 0, ρ ≪ 0), which makes the two-loop direction an ascent one."""
    from sdplrplus_jl_amd import cabi
    data, C, As, bs = make_data(family, 4, 14, 0.4)
    g, o = pair(hip_abi, oracle_abi, data, 3, 13, h=h)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    if h > 0:
        sg = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, h + 1, 0.0, *sg)[:3]
        so = o.inner_loop(normC, normb, True, True, False, 0.0, -1e300, h + 1, 0.0, *so)[:3]
        for s_ in (g, o):   # newest pair: s = G with a large negative ρ ⇒ gᵀHg < 0
            j = int(s_.get_scalar(cabi.S_LBFGS_LATEST)) - 1
            G = s_.Gt
            s_.set_factor(cabi.F_LBFGS_S + j, G)
            s_.set_factor(cabi.F_LBFGS_Y + j, np.zeros_like(G))
            rho = s_.get_vec(cabi.V_LBFGS_RHO)
            rho[j] = -1e6 / float(np.sum(G * G))
            s_.set_vec(cabi.V_LBFGS_RHO, rho)
        dg, do = g.lbfgs_dir(True), o.lbfgs_dir(True)
        assert do > 0 and dg == pytest.approx(do, rel=1e-9)
    for it in range(3):
        rg = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 1, 0.0, *sg)
        ro = o.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 1, 0.0, *so)
        assert rg[4] == ro[4] == 1
        assert np.allclose(rg[:3], ro[:3], rtol=1e-8, atol=1e-12), (it, rg, ro)
        assert rel(g.Rt, o.Rt) < 1e-8 and rel(g.Gt, o.Gt) < 1e-7
        sg, so = rg[:3], ro[:3]
    g.close(); o.close()


@pytest.mark.parametrize("family", ["maxcut", "minimum_bisection", "lovasz_theta"])
def test_lanczos_and_dual_obj(hip_abi, oracle_abi, family):
    data, C, As, bs = make_data(family, 3, 12, 0.4)
    g, o = pair(hip_abi, oracle_abi, data, 3, 4)
    n = data.n
    g.f(); o.f()
    v0 = np.random.Generator(np.random.PCG64(9)).standard_normal(n)
    (dg, eg), (do, eo) = g.dual_obj(float(n), 0, v0), o.dual_obj(float(n), 0, v0)
    w = np.linalg.eigvalsh(S_dense(C, As, g.y))
    # n − 1 plain Lanczos steps give a Ritz value: never below λ_min, and close to it
    assert w[0] - 1e-9 <= eg <= w[0] + 1e-2 * (w[-1] - w[0])
    assert eg == pytest.approx(eo, abs=1e-8 * max(1, abs(eo))) and dg == pytest.approx(do, rel=1e-8)
    ag, bg, kg = g.lanczos(5, v0)
    ao, bo, ko = o.lanczos(5, v0)
    assert kg == ko == 5
    assert np.allclose(ag, ao, rtol=1e-9, atol=1e-12) and np.allclose(bg, bo, rtol=1e-9, atol=1e-12)
    assert g.tridiag_mineig(ag, bg) == pytest.approx(o.tridiag_mineig(ao, bo), abs=1e-12)
    g.close(); o.close()


def test_lanczos_early_exit(hip_abi, oracle_abi):
    """exact invariant subspace ⇒ β = 0 ⇒ break after one step (src/coreop.jl:494-496)"""
    data, C, As, bs = make_data("maxcut", 3, 12, 0.4)
    g, o = pair(hip_abi, oracle_abi, data, 3, 4)
    y = np.concatenate([np.arange(1.0, data.m + 1), [0.0]])   # S = Diag(1..n): C switched off
    e3 = np.zeros(data.n); e3[3] = 2.0
    for s_ in (g, o):
        s_.y = y
        s_.At_preprocess()
        a1, b1, k1 = s_.lanczos(6, e3)
        assert k1 == 1 and a1[0] == 4.0 and b1[0] == 0.0
        assert s_.approx_mineigval_lanczos(6, e3) == 4.0
    g.close(); o.close()


K2 = sp.csc_matrix(np.array([[0.0, 1.0], [1.0, 0.0]]))


@pytest.mark.parametrize("native", [True, False])
def test_known_answers_on_gpu(hip_abi, native):
    """test/maxcut.jl:24,47 and test/minimumbisection.jl:22 through the device library."""
    C, As, bs = problems.maxcut(K2)
    res = sj.sdplr(C, As, bs, 1, fprec=0.0, gtol=1e-8, objtol=1e-8, ptol=1e-8, prior_trace_bound=2.0,
                   printlevel=0, native_inner_loop=native)
    assert res["obj"] == pytest.approx(-1, rel=1.5e-8)
    res = sj.sdplr(C, As, bs, 1, σ_0=10.0, fprec=0.0, gtol=1e-8, objtol=1e-8, ptol=1e-8,
                   prior_trace_bound=2.0, printlevel=0, native_inner_loop=native)
    assert res["obj"] == pytest.approx(-1, rel=1.5e-8)
    C, As, bs = problems.minimum_bisection(K2)
    res = sj.sdplr(C, As, bs, 1, fprec=0.0, objtol=1e-4, ptol=1e-4, prior_trace_bound=2.0,
                   printlevel=0, native_inner_loop=native)
    assert (res["obj"] - 1) / (1 + abs(res["obj"])) < 1e-4


def test_config1_and_rank_doubling_on_gpu(hip_abi, oracle_abi):
    """BASELINE.json configs[0] on the device, next to the CPU oracle run of the same instance."""
    A = problems.gnp_graph(100, 0.1, 1)
    C, As, bs = problems.maxcut(A)
    kw = dict(prior_trace_bound=100.0, printlevel=0, seed=1)
    rg = sj.sdplr(C, As, bs, 2, **kw)
    ro = sj.sdplr(C, As, bs, 2, abi=oracle_abi, **kw)
    assert rg["primal_vio"] <= 1e-2 and ro["primal_vio"] <= 1e-2
    assert rg["obj"] == pytest.approx(ro["obj"], rel=2e-2)       # both within objtol of the SDP value
    assert rg["max_dual_value"] <= rg["obj"] + 1e-6 * abs(rg["obj"])
    # rank doubling path (src/coreop.jl:518-526): r = 1 cannot close the gap, rankupd_tol = 1 doubles at once
    res = sj.sdplr(C, As, bs, 1, prior_trace_bound=100.0, printlevel=0, seed=2, rankupd_tol=1,
                   objtol=1e-3, maxmajoriter=40)
    assert res["r"] >= 2


def test_north_star_size_properties(hip_abi):
    """MaxCut G(n = 1e5, p = 2e-4), r = 32 (BASELINE.json configs[1]) checked through size-independent
    identities computed with scipy: 𝒜(RRᵀ)ᵢ = ‖Rᵢ‖², ⟨C,RRᵀ⟩, G = 2(C + Diag(y))R, and the line-search
    bookkeeping identity of test/coreop.jl:65-72."""
    n, r = 100_000, 32
    A = problems.gnp_graph(n, 2e-4, 20240610)
    data = problems.maxcut_data(A)
    g, _ = make_solver(hip_abi, data, r, seed=0)
    R = g.Rt
    C = data.C
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    L, gn, pn = g.fg(normC, normb)
    pv = g.primal_vio_raw
    rows = np.einsum("ij,ij->i", R, R)
    assert np.max(np.abs(pv[:-1] - (rows - 1.0))) < 1e-10
    CR = C @ R
    obj = float(np.sum(CR * R))
    assert abs(pv[-1] - obj) < 1e-10 * abs(obj)
    y = g.y
    Gref = 2 * (CR + y[:-1, None] * R)
    G = g.Gt
    assert np.max(np.abs(G - Gref)) < 1e-10 * np.max(np.abs(Gref))
    assert gn == pytest.approx(np.linalg.norm(Gref) / normC, rel=1e-12)
    # a few inner iterations, then the bookkeeping identity at the moved point
    res = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 5, 0.0, L, gn, pn)
    assert res[4] == 5 and res[0] < L
    R2 = g.Rt
    pv2 = g.primal_vio_raw
    assert np.max(np.abs(pv2[:-1] - (np.einsum("ij,ij->i", R2, R2) - 1.0))) < 1e-9
    obj2 = float(np.sum((C @ R2) * R2))
    assert abs(pv2[-1] - obj2) < 1e-10 * abs(obj2)
    G2ref = 2 * (C @ R2 + g.y[:-1, None] * R2)
    assert np.max(np.abs(g.Gt - G2ref)) < 1e-10 * np.max(np.abs(G2ref))
    assert res[1] == pytest.approx(np.linalg.norm(G2ref) / normC, rel=1e-10)
    g.close()


def test_config3_lovasz_theta_properties(hip_abi):
    """BASELINE.json configs[2] stand-in: Lovász-θ on a Chung–Lu power-law graph (n ≈ 5e4, |E| ≈ 2.5e5), r = 32:
    one 2-entry COO constraint per edge + identity + low-rank cost.  Checked with scipy identities:
    𝒜(RRᵀ)_e = 2⟨R_i,R_j⟩, trace row = ‖R‖²_F, ⟨C,RRᵀ⟩ = −‖Rᵀ1‖², G = 2·S·R."""
    A = problems.chung_lu_graph(50_000, 10.0, 2.5, 3)
    data = problems.lovasz_theta_data(A)
    n, m, r = data.n, data.m, 32
    g, _ = make_solver(hip_abi, data, r, seed=1)
    R = g.Rt
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    assert normC == pytest.approx(float(n), rel=1e-12)          # ‖−11ᵀ‖_F = n
    L, gn, pn = g.fg(normC, normb)
    pv = g.primal_vio_raw
    coo = sp.triu(A, k=1).tocoo()
    order = np.lexsort((coo.row, coo.col))                     # the builder walks A column-major
    ei, ej = coo.row[order], coo.col[order]
    assert ei.size + 1 == m
    assert np.max(np.abs(pv[:m - 1] - 2 * np.einsum("ij,ij->i", R[ei], R[ej]))) < 1e-10
    assert abs(pv[m - 1] - (np.sum(R * R) - 1.0)) < 1e-9 * np.sum(R * R)
    w = R.sum(axis=0)
    assert abs(pv[m] + w @ w) < 1e-10 * (w @ w)
    y = g.y
    S = sp.coo_matrix((np.concatenate([y[:m - 1], y[:m - 1]]), (np.concatenate([ei, ej]), np.concatenate([ej, ei]))),
                      shape=(n, n)).tocsr() + y[m - 1] * sp.identity(n, format="csr")
    Gref = 2 * (S @ R - y[m] * np.outer(np.ones(n), w))
    G = g.Gt
    assert np.max(np.abs(G - Gref)) < 1e-10 * np.max(np.abs(Gref))
    res = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 10, 0.0, L, gn, pn)
    assert res[4] == 10 and res[0] < L
    R2 = g.Rt
    pv2 = g.primal_vio_raw
    assert np.max(np.abs(pv2[:m - 1] - 2 * np.einsum("ij,ij->i", R2[ei], R2[ej]))) < 1e-9
    g.close()


def test_config4_minimum_bisection_with_lanczos(hip_abi):
    """BASELINE.json configs[3]: MinBisection n = 1e5 (diag constraints + rank-1 constraint 11ᵀ), r = 32, and the
    Lanczos dual-bound path with q = 2⌈√100·ln n⌉ = 232 steps (src/coreop.jl:402), checked against scipy:
    S·v products, the Ritz value bound λ_min(T) ≥ λ_min(S), and dual = −yᵀb + n·min(λ,0)."""
    n, r = 100_000, 32
    A = problems.gnp_graph(n, 2e-4, 4)
    data = problems.minimum_bisection_data(A)
    g, _ = make_solver(hip_abi, data, r, seed=2)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    L, gn, pn = g.fg(normC, normb)
    R = g.Rt
    pv = g.primal_vio_raw
    w = R.sum(axis=0)
    assert abs(pv[n] - w @ w) < 1e-10 * (w @ w)                 # ⟨11ᵀ, RRᵀ⟩ − 0
    assert np.max(np.abs(pv[:n] - (np.einsum("ij,ij->i", R, R) - 1.0))) < 1e-10
    y = g.y
    C = data.C
    Gref = 2 * (C @ R + y[:n, None] * R + y[n] * np.outer(np.ones(n), w))
    assert np.max(np.abs(g.Gt - Gref)) < 1e-10 * np.max(np.abs(Gref))
    res = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 20, 0.0, L, gn, pn)
    assert res[4] == 20
    v0 = np.random.Generator(np.random.PCG64(5)).standard_normal(n)
    dual, ev = g.dual_obj(float(n), 0, v0)
    y = g.y
    S = sp.csr_matrix(C) + sp.diags(y[:n])
    Sop = lambda x: S @ x + y[n] * np.ones(n) * x.sum()
    x = np.random.Generator(np.random.PCG64(6)).standard_normal(n)
    assert np.max(np.abs(g.At_right(x) - Sop(x))) < 1e-10 * np.max(np.abs(Sop(x)))
    al, be, k = g.lanczos(232, v0)
    assert k == 232
    v = v0 / np.linalg.norm(v0)
    assert al[0] == pytest.approx(v @ Sop(v), rel=1e-11)
    assert ev == pytest.approx(g.tridiag_mineig(al, be), abs=1e-12)
    from scipy.sparse.linalg import LinearOperator, eigsh
    lam = eigsh(LinearOperator((n, n), matvec=Sop, dtype=np.float64), k=1, which="SA", tol=1e-6)[0][0]
    assert ev >= lam - 1e-6 * abs(lam) and ev <= lam + 0.05 * abs(lam)      # Ritz value: above λ_min, close to it
    assert dual == pytest.approx(-(y[:n + 1] @ data.b) + n * min(ev, 0.0), rel=1e-10)
    g.close()


def test_config5_gset_batch_slice(hip_abi):
    """BASELINE.json configs[4] in miniature: G1 and G2 of the reference's batch (exps/batch_test.txt: rank 10,
    ptol = objtol = 0.01) solved concurrently on one GPU; weak duality and the known Gset MaxCut bounds."""
    import os
    from sdplrplus_jl_amd import batch
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gset_G1_G9.npz"))
    graphs = [problems.graph_from_edges(int(z[f"G{k}_n"]), z[f"G{k}"]) for k in (1, 2)]
    assert graphs[0].shape == (800, 800) and graphs[0].nnz == 38352
    local = batch.solve_local(graphs, 0, 1, 10, concurrency=2, make_data=problems.maxcut_data, ptol=0.01,
                              objtol=0.01, seed=0, prior_trace_bound=800.0)
    res = batch.gather(local, 2)
    for k in range(2):
        obj, dual = res[k, 1], res[k, 2]
        assert dual <= obj + 1e-6 * abs(obj)
        # best known cuts of G1/G2 are 11624/11620 and the SDP bound is ≈ 12083/12089: −obj lies in between ±1 %
        assert 11600 * 0.99 <= -obj <= 12100 * 1.01


def test_concurrent_handles_match_serial(hip_abi):
    """Different handles driven from different host threads (SURVEY §8b 'Threading'): 24 full MaxCut solves,
    8 in flight, must reproduce the serial results bit for bit (regression: a legacy-stream zero-fill used to
    race with the factor upload on the solver's own stream; hipGraph capture used to break other threads)."""
    from concurrent.futures import ThreadPoolExecutor
    datas = [problems.maxcut_data(problems.gnp_graph(300, 0.1, 200 + k)) for k in range(24)]

    def one(k):
        res = sj.sdplr(data=datas[k], r=8, printlevel=0, ptol=0.01, objtol=0.01, seed=0,
                       prior_trace_bound=300.0, maxmajoriter=60)
        return res["obj"], res["iter"], res["majoriter"]

    serial = [one(k) for k in range(24)]
    with ThreadPoolExecutor(max_workers=8) as ex:
        threaded = list(ex.map(one, range(24)))
    assert all(o < -100 for o, _, _ in serial)
    # Lanczos start vectors are drawn per solve from the same seed ⇒ identical trajectories
    assert threaded == serial


def test_recycled_device_blocks_do_not_leak_state(hip_abi):
    """The library's device-memory pool hands the blocks of a destroyed handle to the next one of the same sizes
    (BASELINE config 5 creates and destroys a handle per instance).  Solves of instance A, then B (same sizes, other
    data), then A again on recycled blocks must reproduce A's first result bit for bit — for the structured path and
    for a hub-row instance, whose extra arrays have other sizes."""
    def solve(data, r):
        res = sj.sdplr(data=data, r=r, printlevel=0, ptol=0.01, objtol=0.01, seed=0,
                       prior_trace_bound=float(data.n), maxmajoriter=40)
        return res["obj"], res["max_dual_value"], res["iter"], res["majoriter"]

    A = problems.maxcut_data(problems.gnp_graph(400, 0.05, 7))
    B = problems.maxcut_data(problems.gnp_graph(400, 0.05, 8))
    H = problems.lovasz_theta_data(problems.chung_lu_graph(400, 6.0, 2.2, 5))
    first = solve(A, 6)
    other = solve(B, 6)
    hub = solve(H, 6)
    assert solve(A, 6) == first and solve(B, 6) == other and solve(H, 6) == hub
    assert first != other


@pytest.mark.parametrize("family,toggles", [
    ("maxcut", ["SDPLR_HIP_NO_FAST"]), ("maxcut", ["SDPLR_HIP_NO_FAST2"]), ("maxcut", ["SDPLR_HIP_NO_GRAPH"]),
    ("maxcut", ["SDPLR_HIP_NO_PDROP"]), ("cutnorm", ["SDPLR_HIP_NO_PDROP"]), ("maxcut", ["SDPLR_HIP_NO_PDROP", "SDPLR_HIP_NO_GRAPH"]),
    ("maxcut", ["SDPLR_HIP_NO_LSHEAD"]), ("cutnorm", ["SDPLR_HIP_NO_LSHEAD", "SDPLR_HIP_NO_GRAPH"]),
    ("maxcut", ["SDPLR_HIP_NO_UPDFUSE"]), ("minimum_bisection", ["SDPLR_HIP_NO_UPDFUSE"]),
    ("mu_conductance_0.05", ["SDPLR_HIP_NO_TILE"]), ("mu_conductance_0.05", ["SDPLR_HIP_NO_UPDFUSE"]),
    ("ineq_0.05", ["SDPLR_HIP_NO_TILE", "SDPLR_HIP_NO_UPDFUSE"]),
    ("lovasz_theta", ["SDPLR_HIP_NO_UPDFUSE"]), ("maxcut", ["SDPLR_HIP_NO_FAST", "SDPLR_HIP_NO_UPDFUSE"]),
    ("ineq_0.05", ["SDPLR_HIP_NO_FAST", "SDPLR_HIP_NO_UPDFUSE"]),
    ("maxcut", ["SDPLR_HIP_DOT_DESCENT"]), ("cutnorm", ["SDPLR_HIP_DOT_DESCENT"]), ("ineq_0.05", ["SDPLR_HIP_DOT_DESCENT"]),
    ("minimum_bisection", ["SDPLR_HIP_NO_FAST"]),
    ("minimum_bisection", ["SDPLR_HIP_NO_FAST2"]), ("minimum_bisection", ["SDPLR_HIP_NO_LRFUSE"]),
    ("minimum_bisection", ["SDPLR_HIP_NO_TILE"]), ("cutnorm", ["SDPLR_HIP_NO_FAST"]),
    ("lovasz_theta", ["SDPLR_HIP_NO_LRFUSE"]), ("minimum_bisection", ["SDPLR_HIP_NO_FAST", "SDPLR_HIP_NO_LRFUSE"]),
    ("mu_conductance_0.05", ["SDPLR_HIP_NO_FAST"]), ("ineq_0.05", ["SDPLR_HIP_NO_FAST"]),
    # the eager route (what instances below n·r = 2¹⁷ take by default; the suite forces graphs otherwise) for EVERY family
    ("lovasz_theta", ["SDPLR_HIP_NO_EDGE"]), ("lovasz_theta", ["SDPLR_HIP_NO_EDGE", "SDPLR_HIP_NO_UPDFUSE"]),
    ("minimum_bisection", ["SDPLR_HIP_NO_GRAPH"]), ("lovasz_theta", ["SDPLR_HIP_NO_GRAPH"]),
    ("cutnorm", ["SDPLR_HIP_NO_GRAPH"]), ("mu_conductance_0.01", ["SDPLR_HIP_NO_GRAPH"]),
    ("mu_conductance_0.05", ["SDPLR_HIP_NO_GRAPH"]), ("mu_conductance_0.1", ["SDPLR_HIP_NO_GRAPH"]),
    ("ineq_0.01", ["SDPLR_HIP_NO_GRAPH"]), ("ineq_0.05", ["SDPLR_HIP_NO_GRAPH"]), ("ineq_0.1", ["SDPLR_HIP_NO_GRAPH"]),
])
def test_code_paths_agree(hip_abi, oracle_abi, family, toggles, monkeypatch):
    """The structured fast paths, the hipGraph batches and the eager launches are the same algorithm:
    with a path switched off (environment toggles read at solver construction / loop entry) a 25-iteration
    run gives the same ℒ, ‖grad‖, ‖pv‖ and R to 1e-9, and both agree with the CPU oracle to 1e-8."""
    data, C, As, bs = make_data(family, 5, 40, 0.3)
    r = 4
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    armijo = data.has_inequalities

    iters = 6 if armijo else 25     # see the note on the μ-conductance-ineq start below
    tolR = 1e-4 if armijo else 1e-9
    st0 = []

    def run(abi):
        s_, _ = make_solver(abi, data, r, seed=21)
        st = s_.fg(normC, normb)
        st0[:] = st
        out = s_.inner_loop(normC, normb, True, True, armijo, 0.0, -1e300, iters, 0.0, *st)
        R = s_.Rt
        s_.close()
        return out, R

    base, Rb = run(hip_abi)
    for t in toggles:
        monkeypatch.setenv(t, "1")
    alt, Ra = run(hip_abi)
    ora, Ro = run(oracle_abi)
    assert base[4] == alt[4] == ora[4] == iters
    # the μ-conductance-ineq start is violently infeasible (ℒ falls from 1e9 to 1e2 in 25 steps): absolute
    # round-off of the first steps is carried along, so ℒ is compared on the scale it started from
    scale = np.array([max(abs(st0[0]), abs(base[0])), max(st0[1], base[1]), max(st0[2], base[2])])
    assert np.all(np.abs(np.array(base[:3]) - np.array(alt[:3])) <= (1e-6 if armijo else 1e-9) * scale) and rel(Rb, Ra) < tolR
    assert np.all(np.abs(np.array(base[:3]) - np.array(ora[:3])) <= (1e-6 if armijo else 1e-8) * scale) and rel(Rb, Ro) < 10 * tolR


def test_step_kernel_without_P_carries_G_forward(hip_abi, oracle_abi, monkeypatch):
    """The singleton fast path with A_g = C and no low-rank matrix runs the step kernel that neither reads nor writes
    P = A_g·R: G_new = G_old + 2(αW + d_new∘R_new − d_old∘R_old) (k_fast_step2<…, PDROP>).  Checked here: (1) G after
    every call of a chain of loops equals the gradient formed from scratch at that point (g! on the same handle) to
    round-off and the oracle's to 1e-8 — across the refresh (SDPLR_HIP_P_REFRESH_ITERS=7: every third call of 3
    iterations rebuilds G), (2) a host write between two loops (here: λ) sends the next loop through the P-based kernel,
    which rebuilds G from the new multipliers exactly as the reference's g! would, (3) R and the history agree with the
    P-based route (SDPLR_HIP_NO_PDROP) to 1e-9 after the chain."""
    monkeypatch.setenv("SDPLR_HIP_P_REFRESH_ITERS", "7")
    data = problems.maxcut_data(problems.gnp_graph(500, 0.03, 11))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    r = 6

    def chain(abi, touch):
        s_, _ = make_solver(abi, data, r, seed=4)
        st = s_.fg(normC, normb)
        gs = []
        for call in range(5):
            out = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 3, 0.0, *st)
            st = out[:3]
            gs.append(s_.Gt.copy())
            if touch and call == 2:
                s_.λ = s_.λ + 0.25                         # G is no longer known to be the gradient at the device's state
        R, S0, Y0 = s_.Rt.copy(), s_.get_factor(cabi.F_LBFGS_S).copy(), s_.get_factor(cabi.F_LBFGS_Y).copy()
        s_.g()                                               # the gradient from scratch at the final point
        Gfresh = s_.Gt.copy()
        s_.close()
        return gs, R, S0, Y0, Gfresh, st

    for touch in (False, True):
        gh, Rh, Sh, Yh, Gf, sth = chain(hip_abi, touch)
        go, Ro, So, Yo, Gfo, sto = chain(oracle_abi, touch)
        for a, b in zip(gh, go):
            assert rel(a, b) < 1e-8
        assert rel(gh[-1], Gf) < 1e-12                       # carried forward ≡ from scratch
        assert rel(Rh, Ro) < 1e-8 and rel(Sh, So) < 1e-7 and rel(Yh, Yo) < 1e-7
        monkeypatch.setenv("SDPLR_HIP_NO_PDROP", "1")
        gp, Rp, Sp, Yp, _, stp = chain(hip_abi, touch)
        monkeypatch.delenv("SDPLR_HIP_NO_PDROP")
        assert rel(Rh, Rp) < 1e-9 and rel(gh[-1], gp[-1]) < 1e-9 and rel(Yh, Yp) < 1e-8
        assert np.allclose(sth, stp, rtol=1e-9)


@pytest.mark.parametrize("r,n,p,tile_k", [
    (1, 90, 0.2, None), (2, 90, 0.2, None), (3, 64, 0.3, 4), (5, 70, 0.3, None), (8, 120, 0.1, 8),
    (16, 150, 0.1, 3), (27, 150, 0.1, None), (32, 400, 0.05, 8), (32, 1000, 0.02, None), (48, 200, 0.1, 5),
    (64, 130, 0.2, 8), (100, 90, 0.3, 2), (128, 70, 0.3, None), (200, 60, 0.4, 7),
    (32, 600, 0.04, 16), (32, 500, 0.05, 13), (10, 300, 0.05, 16), (33, 200, 0.1, 11),     # tiles taller than 8 rows
])
def test_column_sweep_tiles_match_row_kernel(hip_abi, oracle_abi, r, n, p, tile_k, monkeypatch):
    """W = A_g·D on the singleton fast path has two forms: one sub-wave group per row (k_spmm_fast) and the
    column-sweep tiles (k_spmm_tile: K-row tiles, column-sorted padded lists, LDS-resident partial rows).
    Both fold a row's products in increasing column order, so R after 12 iterations must agree to round-off of
    the line-search scalars only — for every sub-wave shape (LPR, VEC), ragged ranks, multi-chunk rows
    (r > 128), forced tile heights, tiles with empty lists (isolated vertices) — and match the CPU oracle."""
    rng = np.random.Generator(np.random.PCG64(1000 + r))
    A = problems.make_random_graph(n, p, rng)
    while A.nnz == 0:
        A = problems.make_random_graph(n, p, rng)
    data = problems.maxcut_data(A)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))

    def run(abi):
        s_, _ = make_solver(abi, data, r, seed=5)
        st = s_.fg(normC, normb)
        out = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 12, 0.0, *st)
        R = s_.Rt
        s_.close()
        return out, R

    if tile_k is not None:
        monkeypatch.setenv("SDPLR_HIP_TILE_K", str(tile_k))
    tile, Rt_ = run(hip_abi)
    monkeypatch.setenv("SDPLR_HIP_NO_TILE", "1")
    rowk, Rr = run(hip_abi)
    ora, Ro = run(oracle_abi)
    assert tile[4] == rowk[4] == ora[4] == 12
    assert np.allclose(tile[:3], rowk[:3], rtol=1e-10, atol=1e-12) and rel(Rt_, Rr) < 1e-10
    assert np.allclose(tile[:3], ora[:3], rtol=1e-8, atol=1e-10) and rel(Rt_, Ro) < 1e-8


def test_column_sweep_tiles_survive_rank_update(hip_abi, oracle_abi):
    """rank_update! (src/coreop.jl:518-526) changes the sub-wave width the tile lists are padded for: the
    library rebuilds them in reset_rank, and the iterations after the rank change still match the oracle."""
    from sdplrplus_jl_amd import cabi
    A = problems.gnp_graph(300, 0.05, 77)
    data = problems.maxcut_data(A)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    outs = []
    for abi in (hip_abi, oracle_abi):
        s_, _ = make_solver(abi, data, 8, seed=5)
        st = s_.fg(normC, normb)
        s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 5, 0.0, *st)
        s_.reset_rank(24)                      # LPR 4 → 16
        R0 = 2.0 * np.random.Generator(np.random.PCG64(9)).random((data.n, 24)) - 1.0
        s_.set_vec(cabi.V_B, data.b)
        s_.set_factor(cabi.F_RT, R0)
        s_.lbfgs_clear()
        st = s_.fg(normC, normb)
        out = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 10, 0.0, *st)
        outs.append((out, s_.Rt))
        s_.close()
    (g, Rg), (o, Ro) = outs
    assert g[4] == o[4] == 10
    assert np.allclose(g[:3], o[:3], rtol=1e-8, atol=1e-10) and rel(Rg, Ro) < 1e-8


def test_lanczos_paths_agree(hip_abi, monkeypatch):
    data, C, As, bs = make_data("minimum_bisection", 6, 60, 0.2)
    g, _ = make_solver(hip_abi, data, 4, seed=3)
    g.f()
    v0 = np.random.Generator(np.random.PCG64(2)).standard_normal(data.n)
    g.dual_obj(float(data.n), 0, v0)
    a1, b1, k1 = g.lanczos(40, v0)
    monkeypatch.setenv("SDPLR_HIP_CLASSIC_LANCZOS", "1")
    a2, b2, k2 = g.lanczos(40, v0)
    assert k1 == k2 == 40
    # unnormalised two-kernel recurrence vs the reference's normalise-every-step form: same coefficients
    # until round-off accumulates through the (unorthogonalised) recurrence
    # (here an extreme Ritz value converges after ≈4 steps, after which plain Lanczos amplifies round-off)
    assert np.allclose(a1[:4], a2[:4], rtol=1e-8) and np.allclose(b1[:4], b2[:4], rtol=1e-8)
    assert g.tridiag_mineig(a1, b1) == pytest.approx(g.tridiag_mineig(a2, b2), abs=1e-8)
    g.close()


def _random_problem(kind: str, n: int, rng):
    """Hand-made constraint structures that hit the classification edges of the device layout."""
    def sym_sparse(density, with_diag=True):
        M = sp.random(n, n, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csr")
        M = sp.triu(M, k=1)
        M = M + M.T
        if with_diag:
            M = M + sp.diags(rng.standard_normal(n))
        return sp.csc_matrix(M)

    def coo_sym(pairs, vals):
        I, J, V = [], [], []
        for (i, j), v in zip(pairs, vals):
            I.append(i); J.append(j); V.append(v)
            if i != j:
                I.append(j); J.append(i); V.append(v)
        return sj.SparseMatrixCOO(I, J, V, n, n)

    lowrank = lambda s: sj.SymLowRankMatrix(rng.standard_normal(s), rng.standard_normal((n, s)))
    if kind == "general_constraint_diag_cost":       # the one off-diagonal matrix is a constraint, C is diagonal
        C = sj.Diagonal(rng.standard_normal(n))
        As = [sym_sparse(0.2)] + [coo_sym([(i, i)], [1.0 + i]) for i in range(n)]
    elif kind == "two_general":                       # two off-diagonal matrices ⇒ generic path
        C = sym_sparse(0.2)
        As = [sym_sparse(0.1), sj.Diagonal(np.ones(n))] + [coo_sym([(i, i)], [1.0]) for i in range(0, n, 2)]
    elif kind == "duplicates_and_missing_diag":       # duplicate COO entries; rows without any diagonal entry
        C = sym_sparse(0.15, with_diag=False)
        As = [sj.SparseMatrixCOO([0, 0, 2], [0, 0, 2], [0.5, 0.25, -1.0], n, n),
              coo_sym([(1, 1), (3, 3), (5, 5)], [1.0, 2.0, 3.0]), lowrank(2)]
    elif kind == "all_lowrank":                       # no sparse matrix at all
        C = lowrank(1)
        As = [lowrank(2), lowrank(1)]
    elif kind == "lowrank_cost_multi_column":         # s > 1 columns, several low-rank matrices + general constraint
        C = lowrank(3)
        As = [sym_sparse(0.2), lowrank(2)] + [coo_sym([(i, i)], [1.0]) for i in range(n)]
    elif kind == "edge_constraints_sparse_cost":      # Lovász-like edge rows AND a sparse cost ⇒ generic path
        C = sym_sparse(0.1)
        As = [coo_sym([(i, (i + 1) % n)], [1.0]) for i in range(n)] + [sp.identity(n, format="csc")]
    else:
        raise KeyError(kind)
    bs = rng.standard_normal(len(As))
    return C, As, bs


@pytest.mark.parametrize("kind", ["general_constraint_diag_cost", "two_general", "duplicates_and_missing_diag",
                                  "all_lowrank", "lowrank_cost_multi_column", "edge_constraints_sparse_cost"])
@pytest.mark.parametrize("r", [2, 5, 32])
def test_unusual_structures(hip_abi, oracle_abi, kind, r):
    rng = np.random.Generator(np.random.PCG64(77))
    n = 14
    C, As, bs = _random_problem(kind, n, rng)
    data = sj.SDPData(C, As, bs)
    g, o = pair(hip_abi, oracle_abi, data, r, 9)
    normC, normb = 3.0, 2.0
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    assert np.allclose(sg, so, rtol=1e-11)
    R = g.Rt
    assert np.max(np.abs(g.primal_vio_raw - primal_vio_dense(C, As, bs, R))) < 1e-9 * (1 + np.max(np.abs(g.primal_vio_raw)))
    assert rel(g.Gt, o.Gt) < 1e-12
    assert np.max(np.abs(g.Gt - 2 * S_dense(C, As, g.y) @ R)) < 1e-10 * (1 + np.max(np.abs(g.Gt)))
    for it in range(6):
        rg = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 1, 0.0, *sg)
        ro = o.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 1, 0.0, *so)
        scale = np.maximum(np.abs(so), np.abs(ro[:3]))
        assert np.all(np.abs(np.array(rg[:3]) - np.array(ro[:3])) <= 1e-8 * scale), (kind, it, rg, ro)
        sg, so = rg[:3], ro[:3]
    assert rel(g.Rt, o.Rt) < 1e-8
    assert np.max(np.abs(g.primal_vio_raw - primal_vio_dense(C, As, bs, g.Rt))) < 1e-8 * (1 + np.max(np.abs(g.primal_vio_raw)))
    v0 = rng.standard_normal(n)
    (dg, eg), (do, eo) = g.dual_obj(5.0, 0, v0), o.dual_obj(5.0, 0, v0)
    # λ_min inherits the 1e-8 agreement of R on the scale of ‖S‖ = O(1), not on its own (it can be ≈ 1e-2)
    assert eg == pytest.approx(eo, rel=1e-7, abs=2e-7) and dg == pytest.approx(do, rel=1e-7, abs=1e-6)
    g.close(); o.close()


def test_eigval_and_dimacs_on_gpu(hip_abi):
    from test_oracle_endtoend import _check_eigval_and_dimacs
    _check_eigval_and_dimacs(hip_abi)


@pytest.mark.parametrize("n_weights", [1, 3, 255, 400])
def test_lanczos_palette_form_matches_value_form(hip_abi, monkeypatch, n_weights):
    """Band plan with 3-byte entries (2-byte column + 1-byte code into the ≤ 255 distinct off-diagonal values of the one
    general matrix, S[i,j] = y_g·A_g[i,j]; the diagonal apart) against the 12-byte value form (SDPLR_HIP_NO_LZPAL=1) on
    weighted MaxCut instances: same α, β to 1e-12 of ‖S‖ — and NOT bit-identical while the palette is in use (another
    summation order), bit-identical once the number of distinct weights exceeds 255 and the library falls back."""
    rng = np.random.Generator(np.random.PCG64(9))
    A = problems.gnp_graph(300, 0.08, 21).tocoo()
    up = A.row < A.col
    w = rng.integers(1, n_weights + 1, size=int(up.sum())).astype(float) * (0.5 if n_weights > 1 else 1.0)
    if n_weights >= 255:
        w[:n_weights] = 0.5 * np.arange(1, n_weights + 1)    # every weight present
    W = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([A.row[up], A.col[up]]), np.concatenate([A.col[up], A.row[up]]))),
                      shape=A.shape).tocsc()
    data = problems.maxcut_data(W)
    v0 = rng.standard_normal(data.n)
    y = rng.standard_normal(data.m + 1)
    monkeypatch.setenv("SDPLR_HIP_LZBAND_MIN_N", "1")
    monkeypatch.setenv("SDPLR_HIP_LZBAND_BW", "128")
    monkeypatch.setenv("SDPLR_HIP_LZBAND_CH", "40")
    runs = []
    for no_pal in (False, True):
        if no_pal:
            monkeypatch.setenv("SDPLR_HIP_NO_LZPAL", "1")
        s_, _ = make_solver(hip_abi, data, 4, seed=3)
        s_.y = y
        s_.At_preprocess()
        al, be, k = s_.lanczos(40, v0)
        # S·x through the operator API (gather form) for a direct check of the first step: α₁ = v'Sv
        v = v0 / np.linalg.norm(v0)
        assert al[0] == pytest.approx(float(v @ s_.At_right(v.reshape(-1, 1)).ravel()), rel=1e-12)
        runs.append((al, be, k))
        s_.close()
    (a1, b1, k1), (a2, b2, k2) = runs
    assert k1 == k2 == 40
    scale = np.max(np.abs(a2))
    assert np.allclose(a1[:8], a2[:8], rtol=0, atol=1e-11 * scale) and np.allclose(b1[:8], b2[:8], rtol=0, atol=1e-11 * scale)
    same = np.array_equal(a1, a2) and np.array_equal(b1, b2)
    assert same == (n_weights > 255)


@pytest.mark.parametrize("family,bw,ch,hub", [("maxcut", 64, 16, None), ("minimum_bisection", 64, 7, None),
                                              ("lovasz_theta", 128, 32, 8), ("mu_conductance_0.05", 64, 1000, None),
                                              ("maxcut", 16384, 4096, 8)])
def test_lanczos_band_form_matches_gather_form_and_oracle(hip_abi, oracle_abi, monkeypatch, family, bw, ch, hub):
    """The Lanczos SpMV through LDS-resident bands of x (k_lz_band + k_lz_step_band) against the gather form
    (k_lz_spmv + k_lz_step) and the oracle: small instances with the plan forced on (several narrow bands, several
    row chunks, ragged last band / chunk, rows without entries in a band, a rank-one term, hub rows left to
    k_spmv_long), on the S of a seeded y.  Coefficients to 1e-9 over the well-conditioned first steps, the Ritz
    value of the whole run to 1e-6 of ‖S‖."""
    data, C, As, bs = make_data(family, 7, 150, 0.08)
    n = data.n
    v0 = np.random.Generator(np.random.PCG64(12)).standard_normal(n)
    if hub is not None:
        monkeypatch.setenv("SDPLR_HIP_HUB_THRESH", str(hub))
    runs = {}
    for name in ("band", "gather", "oracle"):
        if name == "band":
            monkeypatch.setenv("SDPLR_HIP_LZBAND_MIN_N", "1")
            monkeypatch.setenv("SDPLR_HIP_LZBAND_BW", str(bw))
            monkeypatch.setenv("SDPLR_HIP_LZBAND_CH", str(ch))
        else:
            monkeypatch.setenv("SDPLR_HIP_NO_LZBAND", "1")
        s_, _ = make_solver(oracle_abi if name == "oracle" else hip_abi, data, 4, seed=3)
        # the same S on all three: a seeded y (an iterated state differs between the libraries in the 7th digit on
        # the ill-conditioned families, which is about the solver, not about the SpMV)
        s_.y = np.random.Generator(np.random.PCG64(44)).standard_normal(data.m + 1)
        s_.At_preprocess()
        ev = s_.approx_mineigval_lanczos(60, v0)
        dual = ev
        al, be, k = s_.lanczos(60, v0)
        runs[name] = (dual, ev, al, be, k, s_.tridiag_mineig(al, be))
        s_.close()
    for other in ("gather", "oracle"):
        a, b = runs["band"], runs[other]
        assert a[4] == b[4] == 60
        scale = np.max(np.abs(b[2]))
        # comparable steps: while ε·Π(‖S‖/β_j) < 1e-10 (see test_first_iterations_match_oracle_at_baseline_size)
        amp, K = np.finfo(float).eps, 0
        while K < 10 and amp < 1e-10:
            K += 1
            amp *= scale / b[3][K - 1]
        assert np.allclose(a[2][:K], b[2][:K], rtol=1e-9, atol=1e-12 * scale), (K, a[2][:K], b[2][:K])
        assert np.allclose(a[3][:K], b[3][:K], rtol=1e-9, atol=1e-12 * scale)
        # the Ritz value of the whole (unorthogonalised) run: agreement on the scale of ‖S‖, looser where the run
        # has long lost orthogonality (the μ-conductance S has λ_max/λ_min ≈ 1e7)
        assert abs(a[5] - b[5]) <= 1e-6 * scale and abs(a[1] - b[1]) <= 1e-6 * scale and a[1] == a[5]


@pytest.mark.parametrize("family,r,h", [("maxcut", 128, 12), ("maxcut", 64, 16), ("minimum_bisection", 96, 7), ("lovasz_theta", 128, 5),
                                        # numlbfgsvecs beyond the Gram form's 16 (src/lbfgs.jl:35-47 allocates any m): the recursion as
                                        # written, on the structured, the rank-one and the edge path, graph replay and eager
                                        ("maxcut", 8, 20), ("minimum_bisection", 6, 33), ("lovasz_theta", 4, 18), ("ineq_0.05", 4, 24)])
def test_wide_ranks_and_long_histories(hip_abi, oracle_abi, family, r, h):
    """Ranks beyond one DPP row (r = 64 … 128: whole-wave row groups, the tile kernel's wide instantiations — built without
    scratch since round 3) together with histories longer than the fused kernels' four pairs (numlbfgsvecs up to the
    library's 16: lbfgs_update! then runs as one launch per window of four slots): 2h + 3 inner iterations against the
    oracle, through several wraps of the cyclic history (src/lbfgs.jl:77-149)."""
    data, *_ = make_data(family, 6, 40, 0.2)
    g, o = pair(hip_abi, oracle_abi, data, r, 17, h=h)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    assert np.allclose(sg, so, rtol=1e-11)
    k = 2 * h + 3
    rg = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *sg)
    ro = o.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *so)
    assert rg[4] == ro[4] == k
    assert np.allclose(rg[:3], ro[:3], rtol=1e-7, atol=1e-12), (rg, ro)
    assert rel(g.Rt, o.Rt) < 1e-7
    assert g.get_scalar(cabi.S_LBFGS_LATEST) == o.get_scalar(cabi.S_LBFGS_LATEST)
    assert np.allclose(g.get_vec(cabi.V_LBFGS_RHO), o.get_vec(cabi.V_LBFGS_RHO), rtol=1e-5)
    g.close(); o.close()
