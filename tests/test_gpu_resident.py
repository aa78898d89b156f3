"""GPU parity tests of the RESIDENT route (sdplrplus.jl_amd/csrc/k_resident.h): on a small instance one workgroup
owns the instance for a whole inner `while` (src/sdplr.jl:190-278) — ONE launch per call of the loop — and one launch
runs a whole Lanczos recurrence (src/coreop.jl:461-500).  BASELINE config 5 (n = 800, rank 10: exps/batch_test.txt:1-9)
takes this route.  The rest of the GPU suite runs with SDPLR_HIP_FORCE_GRAPH=1 ("treat the instance as a large one"),
which keeps its tiny instances on the multi-launch routes; the tests here remove it.

Every comparison is the same call sequence on the HIP library and on the CPU oracle (tolerances in the asserts), plus
the multi-launch route of the HIP library itself (SDPLR_HIP_NO_RESIDENT=1)."""
import os

import numpy as np
import pytest

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import cabi, problems

from helpers import make_data, make_solver

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, float(np.max(np.abs(b)))))


@pytest.fixture(autouse=True)
def _small_instances_take_their_own_route(monkeypatch):
    monkeypatch.delenv("SDPLR_HIP_FORCE_GRAPH", raising=False)
    monkeypatch.delenv("SDPLR_HIP_NO_RESIDENT", raising=False)


def pair(hip_abi, oracle_abi, data, r, seed, h=4):
    g, _ = make_solver(hip_abi, data, r, seed=seed, h=h)
    o, _ = make_solver(oracle_abi, data, r, seed=seed, h=h)
    return g, o


def gset(name="G1"):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "gset_G1_G9.npz"))
    return problems.graph_from_edges(int(z[f"{name}_n"]), z[name])


LOOP = dict(gtol=0.0, fprec=-1e300)


def run(s, normC, normb, k, st, gtol=0.0, fprec=-1e300, budget=0.0):
    return s.inner_loop(normC, normb, True, True, False, gtol, fprec, k, budget, *st)


@pytest.mark.parametrize("family,n,r,h", [("maxcut", 12, 3, 4), ("maxcut", 40, 8, 1), ("maxcut", 40, 10, 2),
                                          ("cutnorm", 14, 5, 4), ("maxcut", 30, 2, 3),
                                          ("maxcut", 15, 3, 2),      # n·r odd: the 16-byte units of DIR leave one element over
                                          ("maxcut", 67, 1, 4),      # rank one, one row more than a slice
                                          # a rank-one matrix among the constraints (1ᵀX1 = 0, test/problem.jl:78-94): its
                                          # r-vectors Rᵀb, Dᵀb live in LDS, its slot is committed by the scalar stage
                                          ("minimum_bisection", 30, 4, 4), ("minimum_bisection", 41, 3, 2),
                                          ("minimum_bisection", 64, 10, 4)])
def test_resident_loop_iteration_by_iteration(hip_abi, oracle_abi, family, n, r, h):
    """One iteration per call, eight calls: ℒ, ‖grad‖, ‖pv‖ to 1e-8 (north_star's tolerance), R, G, y, primal_vio_raw,
    dirt (restored from s_latest on the way out), the history bookkeeping; every call must have been ONE resident
    launch."""
    data, *_ = make_data(family, 2, n, 0.4)
    g, o = pair(hip_abi, oracle_abi, data, r, 11, h=h)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    assert np.allclose(sg, so, rtol=1e-11)
    # fg! itself was one launch (k_rs_fg): G, y, primal_vio_raw, primal_vio and the objective it leaves behind
    assert g.stats()["resident_fg"] == 1
    assert rel(g.Gt, o.Gt) < 1e-12 and rel(g.y, o.y) < 1e-12 and rel(g.primal_vio_raw, o.primal_vio_raw) < 1e-12
    assert rel(g.get_vec(cabi.V_PV), o.get_vec(cabi.V_PV)) < 1e-12 and g.obj == pytest.approx(o.obj, rel=1e-13)
    for it in range(8):
        rg, ro = run(g, normC, normb, 1, sg), run(o, normC, normb, 1, so)
        assert rg[4] == ro[4] == 1 and rg[5] == ro[5] == 2
        assert np.allclose(rg[:3], ro[:3], rtol=1e-8, atol=1e-12), (it, rg, ro)
        assert rg[3] == pytest.approx(ro[3], rel=1e-6)                       # α*
        assert rel(g.Rt, o.Rt) < 1e-8 and rel(g.Gt, o.Gt) < 1e-7
        assert rel(g.dirt, o.dirt) < 1e-7
        assert rel(g.y, o.y) < 1e-8 and rel(g.primal_vio_raw, o.primal_vio_raw) < 1e-8
        assert g.obj == pytest.approx(o.obj, rel=1e-9, abs=1e-9)   # (abs: MinBisection's objective passes near zero on the way)
        assert g.get_scalar(cabi.S_LBFGS_LATEST) == o.get_scalar(cabi.S_LBFGS_LATEST)
        sg, so = rg[:3], ro[:3]
    st = g.stats()
    assert st["resident_loops"] == 8 and st["graph_batches"] == 0 and st["eager_batches"] == 0
    assert np.allclose(g.get_vec(cabi.V_LBFGS_RHO), o.get_vec(cabi.V_LBFGS_RHO), rtol=1e-6)
    for j in range(h):
        assert rel(g.get_factor(cabi.F_LBFGS_S + j), o.get_factor(cabi.F_LBFGS_S + j)) < 1e-6
        assert rel(g.get_factor(cabi.F_LBFGS_Y + j), o.get_factor(cabi.F_LBFGS_Y + j)) < 1e-6
    g.close(); o.close()


def test_resident_loop_exits_and_continuations(hip_abi, oracle_abi):
    """All four exits of src/sdplr.jl:190-278 taken ON THE DEVICE inside one launch, and the state each leaves behind
    carries on like the oracle's: gradient test (:190), iteration budget (:272-277), relative decrease (:238-241,
    which skips lbfgs_update! and leaves dirt unscaled and y_next = −G_old parked), time budget."""
    data, *_ = make_data("maxcut", 2, 60, 0.2)
    g, o = pair(hip_abi, oracle_abi, data, 6, 11)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    rg = run(g, normC, normb, 500, sg, gtol=0.02 * sg[1])
    ro = run(o, normC, normb, 500, so, gtol=0.02 * so[1])
    assert rg[4] == ro[4] > 3 and rg[5] == ro[5] == 0
    assert np.allclose(rg[:3], ro[:3], rtol=1e-6) and rel(g.dirt, o.dirt) < 1e-5
    rg, ro = run(g, normC, normb, 5, rg[:3]), run(o, normC, normb, 5, ro[:3])
    assert rg[4] == ro[4] == 5 and rg[5] == ro[5] == 2 and np.allclose(rg[:3], ro[:3], rtol=1e-6)
    rg, ro = run(g, normC, normb, 50, rg[:3], fprec=1e300), run(o, normC, normb, 50, ro[:3], fprec=1e300)
    assert rg[4] == ro[4] == 1 and rg[5] == ro[5] == 1
    assert rel(g.dirt, o.dirt) < 1e-5 and rel(g.Gt, o.Gt) < 1e-5
    rg, ro = run(g, normC, normb, 4, rg[:3]), run(o, normC, normb, 4, ro[:3])
    assert rg[4] == ro[4] == 4 and np.allclose(rg[:3], ro[:3], rtol=1e-5) and rel(g.Rt, o.Rt) < 1e-5
    # already converged for the gradient tolerance asked for: no iteration, nothing touched
    R_before, D_before = g.Rt, g.dirt
    r0 = run(g, normC, normb, 10, rg[:3], gtol=10.0 * rg[1])
    assert r0[4] == 0 and r0[5] == 0 and np.array_equal(g.Rt, R_before) and np.array_equal(g.dirt, D_before)
    # time budget: the device watches its own clock; whatever count it reached, the oracle at that count agrees
    rt = run(g, normC, normb, 1_000_000, rg[:3], budget=2e-3)
    assert rt[5] == 3 and 0 < rt[4] < 1_000_000, rt
    ro = run(o, normC, normb, int(rt[4]), ro[:3])
    assert ro[4] == rt[4]
    assert g.stats()["resident_loops"] == 6
    g.close(); o.close()


@pytest.mark.parametrize("r", [1, 2, 5, 16, 33, 64])
def test_resident_loop_over_ranks(hip_abi, oracle_abi, r):
    """Every sub-wave shape (odd ranks: 8-byte lanes; r = 33: a whole wave per row with 31 idle lanes)."""
    data, *_ = make_data("maxcut", 4, 50, 0.2)
    g, o = pair(hip_abi, oracle_abi, data, r, 5)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    rg, ro = run(g, normC, normb, 6, sg), run(o, normC, normb, 6, so)
    assert rg[4] == ro[4] == 6 and g.stats()["resident_loops"] == 1
    assert np.allclose(rg[:3], ro[:3], rtol=1e-8, atol=1e-12)
    assert rel(g.Rt, o.Rt) < 1e-8 and rel(g.Gt, o.Gt) < 1e-7
    g.close(); o.close()


def test_resident_loop_agrees_with_the_multi_launch_route(hip_abi, monkeypatch):
    """The same library on its two routes for a config-5 instance (Gset G1, rank 10): 30 iterations, P refreshed at the
    entry of the second call as well (SDPLR_HIP_P_REFRESH_ITERS is process-wide, so R is written from the host)."""
    data = problems.maxcut_data(gset("G1"))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    out = {}
    for route in ("resident", "launches"):
        if route == "launches":
            monkeypatch.setenv("SDPLR_HIP_NO_RESIDENT", "1")
        g, _ = make_solver(hip_abi, data, 10, seed=0)
        st = g.fg(normC, normb)
        res = run(g, normC, normb, 20, st)
        g.set_factor(cabi.F_RT, g.Rt)                 # R "written outside the loop": P = A_g·R is rebuilt at entry
        st2 = g.fg(normC, normb)
        res2 = run(g, normC, normb, 10, st2)
        out[route] = (res, res2, g.Rt, g.Gt, g.y, g.stats())
        g.close()
    a, b = out["resident"], out["launches"]
    assert a[5]["resident_loops"] == 2 and b[5]["resident_loops"] == 0
    assert a[0][4] == b[0][4] == 20 and a[1][4] == b[1][4] == 10
    assert np.allclose(a[0][:3], b[0][:3], rtol=1e-9) and np.allclose(a[1][:3], b[1][:3], rtol=1e-8)
    assert rel(a[2], b[2]) < 1e-8 and rel(a[3], b[3]) < 1e-7 and rel(a[4], b[4]) < 1e-8


def test_resident_loop_on_a_config5_instance_vs_oracle(hip_abi, oracle_abi):
    """Gset G1 (n = 800, nnz = 38 352, rank 10 — exps/batch_test.txt:1): 25 iterations in one launch against the oracle,
    north_star's 1e-8 on ℒ / ‖grad‖ / objective."""
    data = problems.maxcut_data(gset("G1"))
    g, o = pair(hip_abi, oracle_abi, data, 10, 0)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    rg, ro = run(g, normC, normb, 25, sg), run(o, normC, normb, 25, so)
    assert rg[4] == ro[4] == 25
    assert np.allclose(rg[:3], ro[:3], rtol=1e-8)
    assert g.obj == pytest.approx(o.obj, rel=1e-8)
    assert rel(g.Rt, o.Rt) < 1e-7
    g.close(); o.close()


def test_resident_loop_rows_beyond_the_register_window_and_without_constraints(hip_abi, oracle_abi):
    """n = 1500 > 2 × 512: the third row of a thread takes its constraint data from memory; and a problem whose
    constraints cover only some rows (the others have d_j = 0)."""
    data = problems.maxcut_data(problems.gnp_graph(1500, 0.01, 8))
    g, o = pair(hip_abi, oracle_abi, data, 4, 3)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    rg, ro = run(g, normC, normb, 6, sg), run(o, normC, normb, 6, so)
    assert g.stats()["resident_loops"] == 1
    assert rg[4] == ro[4] == 6 and np.allclose(rg[:3], ro[:3], rtol=1e-8) and rel(g.Rt, o.Rt) < 1e-8
    g.close(); o.close()
    import scipy.sparse as sp
    rng = np.random.Generator(np.random.PCG64(5))
    n = 40
    A = problems.make_random_graph(n, 0.3, rng)
    C, As, bs = problems.maxcut(A)
    keep = [i for i in range(n) if i % 3 != 1]          # rows 1, 4, 7, … carry no constraint
    data = sj.SDPData(C, [As[i] for i in keep], np.asarray([bs[i] for i in keep]))
    g, o = pair(hip_abi, oracle_abi, data, 4, 3)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    rg, ro = run(g, normC, normb, 4, sg), run(o, normC, normb, 4, so)
    assert g.stats()["resident_loops"] == 1
    assert rg[4] == ro[4] == 4 and np.allclose(rg[:3], ro[:3], rtol=1e-8) and rel(g.Rt, o.Rt) < 1e-8
    g.close(); o.close()


@pytest.mark.parametrize("h", [1, 2, 4])
def test_resident_loop_takes_the_steepest_descent_fallback(hip_abi, oracle_abi, h):
    """src/sdplr.jl:201-205 inside the resident launch (the descent test comes from the Gram data, the DIR phase applies
    G ← −G; dir ← G): a history whose newest pair the host has replaced by (s = G, y = 0, ρ ≪ 0)."""
    data, *_ = make_data("maxcut", 4, 14, 0.4)
    g, o = pair(hip_abi, oracle_abi, data, 3, 13, h=h)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    sg, so = run(g, normC, normb, h + 1, sg)[:3], run(o, normC, normb, h + 1, so)[:3]
    for s_ in (g, o):
        j = int(s_.get_scalar(cabi.S_LBFGS_LATEST)) - 1
        G = s_.Gt
        s_.set_factor(cabi.F_LBFGS_S + j, G)
        s_.set_factor(cabi.F_LBFGS_Y + j, np.zeros_like(G))
        rho = s_.get_vec(cabi.V_LBFGS_RHO)
        rho[j] = -1e6 / float(np.sum(G * G))
        s_.set_vec(cabi.V_LBFGS_RHO, rho)
    for it in range(3):
        rg, ro = run(g, normC, normb, 1, sg), run(o, normC, normb, 1, so)
        assert rg[4] == ro[4] == 1
        assert np.allclose(rg[:3], ro[:3], rtol=1e-8, atol=1e-12), (it, rg, ro)
        assert rel(g.Rt, o.Rt) < 1e-8 and rel(g.Gt, o.Gt) < 1e-7
        sg, so = rg[:3], ro[:3]
    assert g.stats()["resident_loops"] >= 4
    g.close(); o.close()


def test_resident_loop_is_run_to_run_deterministic_and_reports_not_descent(hip_abi):
    data = problems.maxcut_data(problems.gnp_graph(300, 0.05, 7))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    seen = set()
    for rep in range(3):
        g, _ = make_solver(hip_abi, data, 8, seed=2)
        st = g.fg(normC, normb)
        out = run(g, normC, normb, 40, st)
        seen.add((g.Rt.tobytes(), g.Gt.tobytes(), out[0]))
        assert g.stats()["resident_loops"] == 1
        g.close()
    assert len(seen) == 1


@pytest.mark.parametrize("family", ["maxcut", "minimum_bisection", "lovasz_theta"])
def test_resident_lanczos_and_dual_obj(hip_abi, oracle_abi, family, monkeypatch):
    """approx_mineigval_lanczos (src/coreop.jl:461-514) as one launch: the raw α, β of every step against the oracle
    (plain normalised recurrence on both sides), the Ritz value, dual_obj; the low-rank term of MinBisection / Lovász-θ
    rides along; the early exit on an invariant subspace (:494-496)."""
    data, C, As, bs = make_data(family, 3, 30, 0.3)
    g, o = pair(hip_abi, oracle_abi, data, 3, 4)
    n = data.n
    g.f(); o.f()
    v0 = np.random.Generator(np.random.PCG64(9)).standard_normal(n)
    (dg, eg), (do, eo) = g.dual_obj(float(n), 0, v0), o.dual_obj(float(n), 0, v0)
    assert g.stats()["resident_lanczos"] == 1
    assert eg == pytest.approx(eo, abs=1e-8 * max(1, abs(eo))) and dg == pytest.approx(do, rel=1e-8)
    if family in ("maxcut", "minimum_bisection"):
        # on the instances of the resident loop dual_obj is ONE launch (copy2y, the recurrence, the tridiagonal's smallest
        # eigenvalue by 64-point multisection, ⟨y, b⟩); piece by piece — host bisection — it must give the very same numbers
        monkeypatch.setenv("SDPLR_HIP_NO_FUSED_DUAL", "1")
        d2, e2 = g.dual_obj(float(n), 0, v0)
        monkeypatch.delenv("SDPLR_HIP_NO_FUSED_DUAL")
        assert (d2, e2) == (dg, eg)
        d3, e3 = g.dual_obj(float(n), 40000, v0)      # q = 2⌈200·ln n⌉ > n − 1: clipped (src/coreop.jl:465)
        d4, e4 = o.dual_obj(float(n), 40000, v0)
        assert e3 == pytest.approx(e4, abs=1e-8 * max(1, abs(e4))) and d3 == pytest.approx(d4, rel=1e-8)
    ag, bg, kg = g.lanczos(5, v0)
    ao, bo, ko = o.lanczos(5, v0)
    assert kg == ko == 5   # (plain Lanczos: later steps of the rank-one cases are ill-conditioned on both sides)
    tol = 1e-9 if family == "maxcut" else 1e-7
    assert np.allclose(ag, ao, rtol=tol, atol=1e-11) and np.allclose(bg, bo, rtol=tol, atol=1e-11)
    if family == "maxcut":
        y = np.concatenate([np.arange(1.0, data.m + 1), [0.0]])   # S = Diag(1..n)
        e3 = np.zeros(n); e3[3] = 2.0
        for s_ in (g, o):
            s_.y = y
            s_.At_preprocess()
            a1, b1, k1 = s_.lanczos(6, e3)
            assert k1 == 1 and a1[0] == 4.0 and b1[0] == 0.0
    g.close(); o.close()


def test_resident_rank_one_constraint_agrees_with_the_multi_launch_route(hip_abi, oracle_abi, monkeypatch):
    """MinBisection on a Gset graph (the reference's batch generator names it: exps/gen_batch_test.jl:3; n = 800, rank 10)
    on the resident route — the rank-one constraint's projections in LDS — against the same library's multi-launch route
    (SDPLR_HIP_NO_RESIDENT_LR=1 keeps instances with a low-rank matrix off the resident route) and the oracle: fg!, 10
    iterations in three calls (a continuation without fg!: the P-less kernel carries G and the low-rank share forward),
    the dual bound.  (Ten, not more: on this family round-off differences — here the order in which Rᵀb and Dᵀb are
    summed — grow by ≈ 1.7× per iteration, 4e-10 on R after 10 iterations and 5e-7 after 25 against the oracle, where
    the multi-launch route, which sums in the oracle's row order, stays 15× closer: scripts/probes/diag_mb.py.)"""
    data = problems.minimum_bisection_data(gset("G2"))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    v0 = np.random.Generator(np.random.PCG64(3)).standard_normal(data.n)
    out = {}
    for route in ("resident", "launches", "oracle"):
        if route == "launches":
            monkeypatch.setenv("SDPLR_HIP_NO_RESIDENT_LR", "1")
        s_, _ = make_solver(oracle_abi if route == "oracle" else hip_abi, data, 10, seed=2)
        st = s_.fg(normC, normb)
        fg0 = st
        for k in (4, 4, 2):
            st = run(s_, normC, normb, k, st)[:3]
        dual = s_.dual_obj(float(data.n), 0, v0)
        stats = s_.stats() if route != "oracle" else None
        out[route] = (fg0, st, s_.Rt.copy(), s_.Gt.copy(), s_.y.copy(), dual, stats)
        s_.close()
        monkeypatch.delenv("SDPLR_HIP_NO_RESIDENT_LR", raising=False)
    assert out["resident"][6]["resident_loops"] == 3 and out["resident"][6]["resident_fg"] == 1
    assert out["resident"][6]["resident_lanczos"] == 1 and out["launches"][6]["resident_loops"] == 0
    for other in ("launches", "oracle"):
        a, b = out["resident"], out[other]
        assert np.allclose(a[0], b[0], rtol=1e-11) and np.allclose(a[1], b[1], rtol=1e-8, atol=1e-10)
        assert rel(a[2], b[2]) < 1e-8 and rel(a[3], b[3]) < 1e-6 and rel(a[4], b[4]) < 1e-7
        assert a[5][0] == pytest.approx(b[5][0], rel=1e-6) and a[5][1] == pytest.approx(b[5][1], abs=1e-6 * max(1.0, abs(b[5][1])))


def test_resident_cutnorm_on_a_gset_graph(hip_abi, oracle_abi):
    """CutNorm on a Gset graph (exps/gen_batch_test.jl:3): 2·800 = 1600 vertices at rank 10 — the direction alone fills the
    CU's LDS (128 KB), the per-row vectors ⟨R_j,D_j⟩, ‖D_j‖², d_j go through global memory (RsLoopArgs::rowvec).  fg!, 12
    iterations in three calls, the dual bound, against the oracle; and the whole solve inside the oracle's tolerance window."""
    data = problems.cutnorm_data(gset("G1"))
    assert data.n == 1600
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    g, o = pair(hip_abi, oracle_abi, data, 10, 6)
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    assert np.allclose(sg, so, rtol=1e-11) and rel(g.Gt, o.Gt) < 1e-12
    for k in (4, 4, 4):
        rg, ro = run(g, normC, normb, k, sg), run(o, normC, normb, k, so)
        assert rg[4] == ro[4] == k and np.allclose(rg[:3], ro[:3], rtol=1e-8, atol=1e-12)
        sg, so = rg[:3], ro[:3]
    assert rel(g.Rt, o.Rt) < 1e-8 and rel(g.Gt, o.Gt) < 1e-7 and rel(g.y, o.y) < 1e-8
    st = g.stats()
    assert st["resident_loops"] == 3 and st["resident_fg"] == 1 and st["eager_batches"] == 0 and st["graph_batches"] == 0
    v0 = np.random.Generator(np.random.PCG64(1)).standard_normal(data.n)
    (dg, eg), (do, eo) = g.dual_obj(800.0, 0, v0), o.dual_obj(800.0, 0, v0)
    assert g.stats()["resident_lanczos"] == 1
    assert eg == pytest.approx(eo, abs=1e-7 * max(1, abs(eo))) and dg == pytest.approx(do, rel=1e-7)
    g.close(); o.close()
    kw = dict(ptol=0.01, objtol=0.01, seed=0, prior_trace_bound=800.0, printlevel=0, maxtime=120.0)
    a = sj.sdplr(data=data, r=10, **kw)
    b = sj.sdplr(data=data, r=10, **kw)
    assert a["obj"] == b["obj"] and a["iter"] == b["iter"]
    gap = (a["obj"] - a["max_dual_value"]) / min(abs(a["obj"]), abs(a["max_dual_value"]))
    assert -1e-2 <= gap <= 1e-2


def test_resident_solve_of_a_minimum_bisection_instance(hip_abi, oracle_abi):
    """sdplr() end to end on MinBisection of Gset G1 (rank 10, ptol = objtol = 0.01): every inner loop and dual bound one
    launch, the result inside the tolerance window of the oracle's solve, a second run bit-identical, and the lockstep
    driver (MaxCut and MinBisection instances side by side: two P-less / P-based groups per round) equal to one by one."""
    from sdplrplus_jl_amd import batch
    data = problems.minimum_bisection_data(gset("G1"))
    kw = dict(ptol=0.01, objtol=0.01, seed=0, prior_trace_bound=800.0, printlevel=0, maxtime=120.0)
    a = sj.sdplr(data=data, r=10, **kw)
    b = sj.sdplr(data=data, r=10, **kw)
    o = sj.sdplr(data=data, r=10, abi=oracle_abi, **kw)
    assert a["obj"] == b["obj"] and a["max_dual_value"] == b["max_dual_value"] and a["iter"] == b["iter"]
    assert abs(a["obj"] - o["obj"]) <= 2e-2 * abs(o["obj"])
    assert abs(a["max_dual_value"] - o["max_dual_value"]) <= 3e-2 * abs(o["max_dual_value"])
    datas = [data, problems.maxcut_data(gset("G3")), problems.minimum_bisection_data(gset("G4")), problems.maxcut_data(gset("G5"))]
    many = batch.solve_lockstep(datas, 10, **kw)
    assert many[0]["obj"] == a["obj"] and many[0]["iter"] == a["iter"] and np.array_equal(many[0]["Rt"], a["Rt"])
    one = sj.sdplr(data=datas[2], r=10, **kw)
    assert many[2]["obj"] == one["obj"] and many[2]["iter"] == one["iter"]


def test_resident_solve_of_a_gset_instance(hip_abi, oracle_abi):
    """sdplr() end to end on Gset G1 at the reference's batch settings (rank 10, ptol = objtol = 0.01): every inner loop
    and every dual bound is one launch; the result lies in the objtol window of the oracle's solve; a second run is
    bit-identical."""
    data = problems.maxcut_data(gset("G1"))
    kw = dict(ptol=0.01, objtol=0.01, seed=0, prior_trace_bound=800.0, printlevel=0)
    a = sj.sdplr(data=data, r=10, **kw)
    b = sj.sdplr(data=data, r=10, **kw)
    o = sj.sdplr(data=data, r=10, abi=oracle_abi, **kw)
    assert a["obj"] == b["obj"] and a["max_dual_value"] == b["max_dual_value"] and a["iter"] == b["iter"]
    assert abs(a["obj"] - o["obj"]) <= 1e-2 * abs(o["obj"])
    assert abs(a["max_dual_value"] - o["max_dual_value"]) <= 2e-2 * abs(o["max_dual_value"])
    gap = (a["obj"] - a["max_dual_value"]) / min(abs(a["obj"]), abs(a["max_dual_value"]))
    assert -1e-2 <= gap <= 1e-2


@pytest.mark.parametrize("family", ["maxcut", "lovasz_theta", "minimum_bisection", "cutnorm", "mu_conductance_0.05", "ineq_0.05"])
def test_native_preprocessing_matches_the_host_mirror(hip_abi, oracle_abi, family):
    """sdplr_hip_set_sparse_coo — preprocess_sparsecons (src/preprocess.jl:24-169) inside the library — against the
    vectorised host mirror and the oracle's literal restatement: every array of the aggregated layout, bit for bit."""
    from oracle import oracle
    for seed, n, p in [(1, 5, 0.4), (2, 12, 0.7), (3, 60, 0.3)]:
        data, *_ = make_data(family, seed, n, p)
        a, o = sj.preprocess_sparsecons(data.sparse), oracle.preprocess(data.sparse)
        s = cabi.DeviceSolver(hip_abi, data.n, data.m, 2, 4)
        s.set_sparse_coo(data.sparse)
        b = s.get_layout()
        for k in ("matptr", "nzind", "nzval_one", "nzval_two", "triu_colptr", "triu_rowval",
                  "full_colptr", "full_rowval", "mappedto_triu", "global_inds"):
            assert np.array_equal(getattr(a, k), getattr(b, k)) and np.array_equal(getattr(o, k), getattr(b, k)), (family, seed, k)
        s.close()


def test_native_preprocessing_edge_cases(hip_abi):
    """Unsorted entries, positions shared by several matrices, upper-triangular-only input (accepted, as by the
    reference's search), a lower entry without its mirror / an entry outside the matrix (SDPLR_ERR_INVALID_ARG with the
    message), no entries at all."""
    from sdplrplus_jl_amd.structs import SparseBatch
    n = 6

    def native(batch):
        s = cabi.DeviceSolver(hip_abi, n, 3, 2, 4)
        try:
            s.set_sparse_coo(batch)
            return s.get_layout()
        finally:
            s.close()

    I = np.array([4, 1, 1, 0, 3, 3, 1, 2, 5], dtype=np.int64)
    J = np.array([1, 4, 1, 0, 3, 3, 1, 5, 2], dtype=np.int64)
    V = np.array([2.0, 2.0, 1.0, -1.0, 0.5, 0.25, 3.0, 7.0, 7.0])
    batch = SparseBatch(n, np.array([0, 4, 7, 9], dtype=np.int64), I, J, V, np.array([0, 1, 2], dtype=np.int64))
    up = SparseBatch(n, np.array([0, 3], dtype=np.int64), np.array([1, 0, 2]), np.array([4, 0, 5]), np.array([1.0, 2.0, 3.0]),
                     np.array([0]))
    for bt in (batch, up):
        a, b = sj.preprocess_sparsecons(bt), native(bt)
        for k in ("matptr", "nzind", "nzval_one", "nzval_two", "triu_colptr", "triu_rowval", "full_colptr", "full_rowval",
                  "mappedto_triu", "global_inds"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), k
    bad = SparseBatch(n, np.array([0, 1], dtype=np.int64), np.array([4]), np.array([1]), np.array([1.0]), np.array([0]))
    with pytest.raises(cabi.SdplrError, match="not symmetric"):
        native(bad)
    out = SparseBatch(n, np.array([0, 1], dtype=np.int64), np.array([6]), np.array([1]), np.array([1.0]), np.array([0]))
    with pytest.raises(cabi.SdplrError, match="outside"):
        native(out)
    empty = SparseBatch(n, np.array([0], dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64),
                        np.zeros(0), np.zeros(0, dtype=np.int64))
    e = native(empty)
    assert e.nnzT == 0 and e.nnzS == 0 and e.nnzAgg == 0


@pytest.mark.parametrize("update_lambda", [True, False])
def test_major_iteration_is_the_sequence_it_stands_for(hip_abi, oracle_abi, update_lambda, monkeypatch):
    """sdplr_hip_major_iteration — [λ update] → σ → lbfgs_clear! → fg! → while loop (src/sdplr.jl:358-369, :384, :389,
    :190-278) as ONE resident launch — against the same five calls made one by one on the HIP library (its multi-launch
    route) and on the oracle: state after the call, and the continuation."""
    data = problems.maxcut_data(problems.gnp_graph(200, 0.05, 3))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    g, o = pair(hip_abi, oracle_abi, data, 6, 2)
    monkeypatch.setenv("SDPLR_HIP_NO_RESIDENT", "1")
    g2, _ = make_solver(hip_abi, data, 6, seed=2)      # the same library, one call at a time on the multi-launch route
    monkeypatch.delenv("SDPLR_HIP_NO_RESIDENT")
    outs = []
    for s_ in (g, o, g2):
        st = s_.fg(normC, normb)
        st = run(s_, normC, normb, 12, st)[:3]
        sigma = 2.0 if update_lambda else 4.0
        if s_ is g2:
            if update_lambda:
                s_.update_lambda()
            s_.σ = sigma
            s_.lbfgs_clear()
            st = s_.fg(normC, normb)
            res = run(s_, normC, normb, 9, st, gtol=1e-3)
        else:
            res = s_.major_iteration(normC, normb, True, True, False, update_lambda, sigma, 1e-3, -1e300, 9, 0.0)
        outs.append(res)
    assert g.stats()["resident_loops"] == 2 and g.stats()["resident_fg"] == 2 and g2.stats()["resident_loops"] == 0
    for res in outs[1:]:
        assert res[4] == outs[0][4] == 9 and np.allclose(res[:3], outs[0][:3], rtol=1e-8)
    for other in (o, g2):
        assert rel(g.Rt, other.Rt) < 1e-8 and rel(g.λ, other.λ) < 1e-12 and g.σ == other.σ
        assert rel(g.y, other.y) < 1e-8 and rel(g.Gt, other.Gt) < 1e-7 and rel(g.dirt, other.dirt) < 1e-6
        assert g.get_scalar(cabi.S_LBFGS_LATEST) == other.get_scalar(cabi.S_LBFGS_LATEST)
        assert np.allclose(g.get_vec(cabi.V_LBFGS_RHO), other.get_vec(cabi.V_LBFGS_RHO), rtol=1e-6)
    # already converged for fg!'s gradient: no iteration, exit 0, and fg!'s values come back
    r0 = [s_.major_iteration(normC, normb, True, True, False, False, 8.0, 1e9, -1e300, 5, 0.0) for s_ in (g, o)]
    assert r0[0][4] == r0[1][4] == 0 and r0[0][5] == r0[1][5] == 0 and np.allclose(r0[0][:3], r0[1][:3], rtol=1e-9)
    rg, ro = run(g, normC, normb, 5, r0[0][:3]), run(o, normC, normb, 5, r0[1][:3])
    assert rg[4] == ro[4] == 5 and np.allclose(rg[:3], ro[:3], rtol=1e-7)
    for s_ in (g, o, g2):
        s_.close()


@pytest.mark.parametrize("W", [2, 3, 4])
@pytest.mark.parametrize("r", [10, 7])
def test_resident_team_matches_the_single_workgroup_loop(hip_abi, oracle_abi, monkeypatch, W, r):
    """SDPLR_HIP_TEAM=W (default 3; 1: none): W workgroups of one XCD share an instance inside the resident loop (k_resident.h, TEAM) — every
    member forms the direction, each takes its slices of the SpMM and its rows of the line search, the commit and STEP;
    Gram / norm partials, the rows of W with their dots and the ten line-search sums cross between them behind three team
    barriers per iteration.  Same iterates as the one-workgroup loop up to the order of those sums, and the oracle's to
    north_star's tolerance; even rank (16-byte pieces) and odd."""
    data = problems.maxcut_data(gset("G1"))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    monkeypatch.setenv("SDPLR_HIP_TEAM", "1")
    one, o = pair(hip_abi, oracle_abi, data, r, 0)
    s1 = one.fg(normC, normb)
    r1 = run(one, normC, normb, 25, s1)
    monkeypatch.setenv("SDPLR_HIP_TEAM", str(W))
    team = make_solver(hip_abi, data, r, seed=0)[0]
    st = team.fg(normC, normb)
    assert np.allclose(st, s1, rtol=1e-13)     # (the sliced ELL is cut for the team: fg!'s sums run over the rows in another order)
    rt = run(team, normC, normb, 25, st)
    ro = run(o, normC, normb, 25, o.fg(normC, normb))
    assert rt[4] == r1[4] == ro[4] == 25 and rt[5] == r1[5]
    assert np.allclose(rt[:3], r1[:3], rtol=1e-10) and rel(team.Rt, one.Rt) < 1e-9 and rel(team.Gt, one.Gt) < 1e-8
    assert np.allclose(rt[:3], ro[:3], rtol=1e-8) and rel(team.Rt, o.Rt) < 1e-7
    assert team.obj == pytest.approx(o.obj, rel=1e-8)
    # the loop continues from the state a team left (and a team from the state one workgroup left)
    a, b = run(team, normC, normb, 7, rt[:3]), run(one, normC, normb, 7, r1[:3])
    assert a[4] == b[4] == 7 and np.allclose(a[:3], b[:3], rtol=1e-9)
    monkeypatch.setenv("SDPLR_HIP_TEAM", "1")
    a2, b2 = run(team, normC, normb, 5, a[:3]), run(one, normC, normb, 5, b[:3])
    assert np.allclose(a2[:3], b2[:3], rtol=1e-9) and rel(team.Rt, one.Rt) < 1e-8
    for s_ in (one, team, o):
        s_.close()


def test_resident_team_in_batch_calls_and_whole_solves(hip_abi, monkeypatch):
    """Teams inside the lockstep batch calls (grid 8·W·⌈B/8⌉, block b = rank (b/8) mod W of instance ((b/8)/W)·8 + b mod 8):
    the team size is a property of the instance, so a batch call and the single calls run the same kernel — bit for bit —
    and a whole solve ends where the one-workgroup solve ends (to the solve's own tolerance)."""
    datas = [problems.maxcut_data(gset(g)) for g in ("G1", "G2", "G3")] + [make_data("maxcut", 5, 60, 0.2)[0]]
    kw = dict(ptol=1e-2, objtol=1e-2, maxtime=120.0, printlevel=0, prior_trace_bound=800.0)
    monkeypatch.setenv("SDPLR_HIP_TEAM", "1")
    plain = sj.sdplr(data=datas[0], r=10, **kw)
    monkeypatch.setenv("SDPLR_HIP_TEAM", "4")
    A = [make_solver(hip_abi, d, 10, seed=3)[0] for d in datas]
    B = [make_solver(hip_abi, d, 10, seed=3)[0] for d in datas]
    norms = [(d.normC(), float(np.linalg.norm(d.b))) for d in datas]
    assert cabi.batch_fg(hip_abi, A, [(nc, nb, 1, 1) for nc, nb in norms]) == \
        [s.fg(nc, nb, True, True) + (s.obj,) for s, (nc, nb) in zip(B, norms)]
    for rnd, upd in enumerate((0, 1, 1)):
        args = [(nc, nb, 1, 1, 0, upd, 2.0 * (rnd + 1), 1e-3, 1e-30, 9 + 3 * k + rnd, 0.0) for k, (nc, nb) in enumerate(norms)]
        assert cabi.batch_major_iteration(hip_abi, A, args) == [s.major_iteration(*a) + (s.obj,) for s, a in zip(B, args)], rnd
    for a, b in zip(A, B):
        assert np.array_equal(a.Rt, b.Rt) and np.array_equal(a.Gt, b.Gt) and np.array_equal(a.dirt, b.dirt)
        a.close()
        b.close()
    team = sj.sdplr(data=datas[0], r=10, **kw)
    assert team["obj"] == pytest.approx(plain["obj"], rel=1e-5) and team["max_dual_value"] == pytest.approx(plain["max_dual_value"], rel=1e-5)
    assert abs(team["iter"] - plain["iter"]) <= max(5, plain["iter"] // 10)


def test_resident_team_with_a_rank_one_constraint(hip_abi, oracle_abi, monkeypatch):
    """MinBisection on Gset G2 in a team: every member keeps Rᵀb, Dᵀb and the slot's primal_vio_raw for itself (same values in
    every member), rank 0 stores them.  Against the one-workgroup loop and the oracle after 10 iterations (on this family
    differences in the order of a sum grow ≈ 1.7× per iteration: scripts/probes/diag_mb.py)."""
    data = problems.minimum_bisection_data(gset("G2"))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    monkeypatch.setenv("SDPLR_HIP_TEAM", "1")
    one, o = pair(hip_abi, oracle_abi, data, 10, 0)
    r1 = run(one, normC, normb, 10, one.fg(normC, normb))
    ro = run(o, normC, normb, 10, o.fg(normC, normb))
    for W in (2, 3, 4):
        monkeypatch.setenv("SDPLR_HIP_TEAM", str(W))
        team = make_solver(hip_abi, data, 10, seed=0)[0]
        rt = run(team, normC, normb, 10, team.fg(normC, normb))
        assert rt[4] == r1[4] == 10 and team.stats()["resident_loops"] == 1
        assert np.allclose(rt[:3], r1[:3], rtol=1e-8) and rel(team.Rt, one.Rt) < 1e-7
        assert np.allclose(rt[:3], ro[:3], rtol=1e-7) and rel(team.Rt, o.Rt) < 1e-6
        assert np.array_equal(team.get_vec(cabi.V_PV_RAW)[[-2, -1]] != 0, [True, True])
        team.close()
    one.close(); o.close()


def test_resident_team_that_is_not_on_one_xcd_falls_back(hip_abi, monkeypatch):
    """The members check their placement before anything is touched (HW_REG_XCC_ID; dispatch order is observed, never
    promised).  SDPLR_HIP_TEAM_TEST_FAIL makes the check answer "not one XCD": the launch returns untouched, the host runs the
    instance without a team — for good — and the results are those of the one-workgroup loop on the same handle layout;
    through the single entry point and through a batch call."""
    datas = [problems.maxcut_data(gset(g)) for g in ("G1", "G2")]
    norms = [(d.normC(), float(np.linalg.norm(d.b))) for d in datas]
    args = [(nc, nb, 1, 1, 0, 0, 2.0, 1e-3, 1e-30, 12, 0.0) for nc, nb in norms]
    A = [make_solver(hip_abi, d, 10, seed=3)[0] for d in datas]       # teams of two
    ref = [s.major_iteration(*a) + (s.obj,) for s, a in zip(A, args)]
    monkeypatch.setenv("SDPLR_HIP_TEAM_TEST_FAIL", "1")
    B = [make_solver(hip_abi, d, 10, seed=3)[0] for d in datas]       # single calls: fall back inside the call
    got = [s.major_iteration(*a) + (s.obj,) for s, a in zip(B, args)]
    C_ = [make_solver(hip_abi, d, 10, seed=3)[0] for d in datas]      # a batch call: the items come back through the single route
    gotb = cabi.batch_major_iteration(hip_abi, C_, args)
    monkeypatch.delenv("SDPLR_HIP_TEAM_TEST_FAIL")
    for r0, r1, r2 in zip(ref, got, gotb):
        assert r1 == r2                                               # (both ran the one-workgroup loop on the same layout)
        assert r0[4] == r1[4] and np.allclose(r0[:3], r1[:3], rtol=1e-9)
    for a, b, c in zip(A, B, C_):
        assert np.array_equal(b.Rt, c.Rt) and rel(a.Rt, b.Rt) < 1e-8
    # … and they stay without a team: the next call (flag gone) still equals the one-workgroup result of the twin
    nxt = [(nc, nb, 1, 1, 0, 1, 4.0, 1e-3, 1e-30, 9, 0.0) for nc, nb in norms]
    assert [s.major_iteration(*a) for s, a in zip(B, nxt)] == [s.major_iteration(*a) for s, a in zip(C_, nxt)]
    for s_ in A + B + C_:
        s_.close()
