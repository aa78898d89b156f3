"""GPU tests of the lockstep batch calls (include/sdplr_hip.h: sdplr_hip_batch_fg / _major_iteration / _dual_obj): the
resident instances of a batch go out as ONE launch, one workgroup per instance (k_rs_*_batch in k_resident.h).  The bar:
bit-identical to the single-instance entry points — the same kernel body runs, only the way its arguments arrive differs —
and the oracle's results to the tolerances of tests/test_gpu_resident.py."""
import os

import numpy as np
import pytest

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, cabi, problems

from helpers import make_data, make_solver

pytestmark = pytest.mark.gpu

# (the trace bound of the experiments, exps/test.jl:166-176: with the default — 1e18 — the dual bound hinges on the sign
# of a Lanczos estimate at rounding level and such a solve is not a reproducible test case)
KW = dict(ptol=1e-2, objtol=1e-2, maxtime=120.0, printlevel=0, prior_trace_bound=800.0)


@pytest.fixture(autouse=True)
def _small_instances_take_their_own_route(monkeypatch):
    monkeypatch.delenv("SDPLR_HIP_FORCE_GRAPH", raising=False)
    monkeypatch.delenv("SDPLR_HIP_NO_RESIDENT", raising=False)


def gset(name):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "gset_G1_G9.npz"))
    return problems.maxcut_data(problems.graph_from_edges(int(z[f"{name}_n"]), z[name]))


def same(a, b):
    assert a["iter"] == b["iter"] and a["majoriter"] == b["majoriter"]
    assert a["obj"] == b["obj"] and a["max_dual_value"] == b["max_dual_value"]
    assert np.array_equal(a["Rt"], b["Rt"]) and np.array_equal(a["lambda"], b["lambda"])
    assert a["schedule"] == b["schedule"]


def test_batch_calls_equal_the_single_calls_bit_for_bit(hip_abi):
    """fg → major_iteration (λ update on / off) → dual_obj on six instances of different sizes, once through the batch
    calls and once through the single-instance calls: every output and the whole device state afterwards are equal bit
    for bit, and the batch made one shared launch per call."""
    specs = [("maxcut", 1, 40), ("maxcut", 2, 64), ("maxcut", 3, 25), ("cutnorm", 4, 14), ("maxcut", 5, 90), ("maxcut", 6, 33)]
    datas = [make_data(f, s, n, 0.3)[0] for f, s, n in specs]
    A = [make_solver(hip_abi, d, 6, seed=7)[0] for d in datas]
    B = [make_solver(hip_abi, d, 6, seed=7)[0] for d in datas]
    norms = [(d.normC(), float(np.linalg.norm(d.b))) for d in datas]
    rng = np.random.default_rng(0)
    fa = cabi.batch_fg(hip_abi, A, [(nc, nb, 1, 1) for nc, nb in norms])
    fb = [s.fg(nc, nb, True, True) + (s.obj,) for s, (nc, nb) in zip(B, norms)]
    assert fa == fb
    for rnd, upd in enumerate((0, 1, 1, 0)):
        sig = 2.0 * (rnd + 1)
        args = [(nc, nb, 1, 1, 0, upd, sig, 1e-3, 1e-30, 7 + k, 0.0) for k, (nc, nb) in enumerate(norms)]
        ma = cabi.batch_major_iteration(hip_abi, A, args)
        mb = [s.major_iteration(*a) + (s.obj,) for s, a in zip(B, args)]
        assert ma == mb, rnd
        v0s = [rng.standard_normal(d.n) for d in datas]
        da = cabi.batch_dual_obj(hip_abi, A, [(float(d.n), 50 * (rnd + 1), v) for d, v in zip(datas, v0s)])
        db = [s.dual_obj(float(d.n), 50 * (rnd + 1), v) for s, d, v in zip(B, datas, v0s)]
        assert [x[:2] for x in da] == db, rnd
        for x, s in zip(da, B):               # (var.y comes back with the batch's results)
            assert np.array_equal(x[2], s.y)
    for a, b in zip(A, B):
        for name in ("Rt", "Gt", "dirt", "y", "λ", "primal_vio_raw"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
        for j in range(4):
            assert np.array_equal(a.get_factor(cabi.F_LBFGS_S + j), b.get_factor(cabi.F_LBFGS_S + j))
            assert np.array_equal(a.get_factor(cabi.F_LBFGS_Y + j), b.get_factor(cabi.F_LBFGS_Y + j))
        assert np.array_equal(a.get_vec(cabi.V_LBFGS_RHO), b.get_vec(cabi.V_LBFGS_RHO))
        assert a.σ == b.σ and a.obj == b.obj
        st = a.stats()
        assert st["resident_shared_launches"] == 1 + 4 + 4 and st["graph_batches"] == 0 and st["eager_batches"] == 0
        assert b.stats()["resident_shared_launches"] == 0
    # the state the batch left is a state the single calls continue from
    for a, b, (nc, nb) in zip(A, B, norms):
        assert a.inner_loop(nc, nb, True, True, False, 0.0, -1e300, 3, 0.0, *a.fg(nc, nb)) == \
            b.inner_loop(nc, nb, True, True, False, 0.0, -1e300, 3, 0.0, *b.fg(nc, nb))
        a.close()
        b.close()


def test_batch_of_mixed_routes_and_shapes(hip_abi, oracle_abi, monkeypatch):
    """One call with: resident instances of both kernel shapes (even ranks 6 and 20: 16-byte pieces of a row; odd ranks 5, 7
    and 3: 8-byte ones — one launch per shape), among them one with a rank-one constraint (MinBisection, rank 6), and an
    instance of the edge path (Lovász-θ: multi-launch route) — the call is total, every item equals its single-instance twin."""
    specs = [("maxcut", 1, 40, 6), ("maxcut", 2, 50, 20), ("maxcut", 3, 30, 5), ("maxcut", 4, 36, 7),
             ("lovasz_theta", 5, 24, 6), ("maxcut", 6, 28, 3), ("minimum_bisection", 7, 26, 6)]
    datas = [make_data(f, s, n, 0.3)[0] for f, s, n, _ in specs]
    A = [make_solver(hip_abi, d, sp[3], seed=3)[0] for d, sp in zip(datas, specs)]
    B = [make_solver(hip_abi, d, sp[3], seed=3)[0] for d, sp in zip(datas, specs)]
    norms = [(d.normC(), float(np.linalg.norm(d.b))) for d in datas]
    assert cabi.batch_fg(hip_abi, A, [(nc, nb, 1, 1) for nc, nb in norms]) == \
        [s.fg(nc, nb, True, True) + (s.obj,) for s, (nc, nb) in zip(B, norms)]
    args = [(nc, nb, 1, 1, int(d.has_inequalities), 1, 3.0, 1e-2, 1e-30, 12, 0.0) for d, (nc, nb) in zip(datas, norms)]
    assert cabi.batch_major_iteration(hip_abi, A, args) == [s.major_iteration(*a) + (s.obj,) for s, a in zip(B, args)]
    v0s = [np.random.default_rng(k).standard_normal(d.n) for k, d in enumerate(datas)]
    da = cabi.batch_dual_obj(hip_abi, A, [(float(d.n), 100, v) for d, v in zip(datas, v0s)])
    assert [x[:2] for x in da] == [s.dual_obj(float(d.n), 100, v) for s, d, v in zip(B, datas, v0s)]
    for x, s in zip(da, B):
        assert np.array_equal(x[2], s.y)
    shared = [s.stats()["resident_shared_launches"] for s in A]
    assert shared[:4] == [3, 3, 3, 3] and shared[4] == 0 and shared[5] == 3 and shared[6] == 3
    # … and against the oracle, to the resident route's tolerances
    for k in (1, 2):
        o = make_solver(oracle_abi, datas[k], specs[k][3], seed=3)[0]
        o.fg(*norms[k], True, True)
        ro = o.major_iteration(*args[k])
        assert np.allclose([A[k].get_scalar(cabi.S_OBJ)], [o.obj], rtol=1e-9)
        assert np.allclose(A[k].Rt, o.Rt, rtol=1e-7, atol=1e-9) and ro[4] > 0
        o.close()
    for s in A + B:
        s.close()


def test_solve_lockstep_equals_sdplr_one_by_one(hip_abi):
    """Nine whole solves (two Gset graphs of BASELINE config 5 among them) side by side = the nine solves one by one: same
    iterations, objective, dual bound, R, λ and schedule, bit for bit."""
    datas = [gset("G1"), gset("G5")]
    for seed, n in ((1, 120), (2, 200), (3, 64), (4, 333), (5, 150), (6, 90)):
        datas.append(make_data("maxcut", seed, n, 0.08)[0])
    datas.append(make_data("minimum_bisection", 7, 40, 0.2)[0])
    one = [sj.sdplr(data=d, r=10, **KW) for d in datas]
    many = batch.solve_lockstep(datas, 10, **KW)
    for a, b in zip(one, many):
        assert not isinstance(b, Exception), b
        same(a, b)
    assert one[0]["iter"] > 50 and all(x["majoriter"] < 40 for x in one)


@pytest.mark.parametrize("cap", [5, 64])
def test_capped_rounds_resume_bit_for_bit(hip_abi, cap):
    """The lockstep driver caps the inner iterations per launch and resumes the loops that hit the cap (MAJOR_RESUME: the
    resident kernel without its prologue, on the control block, the history and — without P — the G it left): the solves
    equal the uncapped driver's and sdplr() one by one bit for bit, Gset instances (their longest inner loops run for
    hundreds of iterations) and small random ones (even and odd ranks are different kernel shapes) alike."""
    datas = [gset("G1"), gset("G2")] + [make_data("maxcut", s, n, 0.3)[0] for s, n in ((1, 40), (2, 64), (3, 25))]
    for r in (10, 5):
        uncapped = batch.solve_lockstep(datas, r, abi=hip_abi, setup_workers=2, iteration_cap=0, **KW)
        capped = batch.solve_lockstep(datas, r, abi=hip_abi, setup_workers=2, iteration_cap=cap, **KW)
        for a, b in zip(uncapped, capped):
            same(a, b)
        same(sj.sdplr(data=datas[0], r=r, abi=hip_abi, **KW), capped[0])


def test_resume_on_the_single_entry_point(hip_abi):
    """major_iteration(update_lambda = MAJOR_RESUME) ≡ the continuation of the loop: 12 iterations in one call equal
    5 + 5 + 2 through two resumes (ℒ, norms, R, G, the history), on the resident route with and without P."""
    data = gset("G3")
    normC, normb = data.normC(), float(np.linalg.norm(data.b))

    def run(chunks):
        s_ = make_solver(hip_abi, data, 10, seed=3)[0]
        out = s_.major_iteration(normC, normb, 1, 1, 0, 0, 2.0, 0.0, -1e300, chunks[0], 0.0)
        total = out[4]
        for c in chunks[1:]:
            out = s_.major_iteration(normC, normb, 1, 1, 0, cabi.MAJOR_RESUME, 2.0, 0.0, -1e300, c, 0.0, *out[:3])
            total += out[4]
        state = (out[:4], total, s_.Rt.copy(), s_.Gt.copy(), s_.get_factor(cabi.F_LBFGS_Y + 1).copy())
        s_.close()
        return state

    a, b = run([12]), run([5, 5, 2])
    assert a[0] == b[0] and a[1] == b[1] == 12
    for x, y in zip(a[2:], b[2:]):
        assert np.array_equal(x, y)


def test_lockstep_through_rank_doublings(hip_abi):
    """Rank doublings inside a lockstep batch (reset_rank, a fresh point, then single-instance fg! / inner loop for that
    instance while the others keep sharing launches): bit-identical to the solves one by one."""
    datas = [make_data("maxcut", seed, n, 0.3)[0] for seed, n in ((1, 40), (2, 60), (4, 50), (5, 44))]
    kw = dict(ptol=1e-3, objtol=1e-4, maxtime=60.0, printlevel=0, prior_trace_bound=60.0, rankupd_tol=2)
    one = [sj.sdplr(data=d, r=2, **kw) for d in datas]
    many = batch.solve_lockstep(datas, 2, **kw)
    assert any(x["r"] > 2 for x in one)
    for a, b in zip(one, many):
        assert not isinstance(b, Exception), b
        same(a, b)


def test_lockstep_on_the_multi_launch_routes(hip_abi, monkeypatch):
    """Instances that are not on the resident route (switched off here; in production: larger ones, low-rank or inequality
    constraints) are served by the single-instance entry points on a few host threads inside each batch call — same
    results as one by one, no shared launches."""
    monkeypatch.setenv("SDPLR_HIP_NO_RESIDENT", "1")
    datas = [make_data("maxcut", seed, n, 0.1)[0] for seed, n in ((1, 150), (2, 220), (3, 90), (4, 300), (5, 180))]
    datas.append(make_data("minimum_bisection", 7, 60, 0.2)[0])
    kw = dict(KW, prior_trace_bound=300.0)
    one = [sj.sdplr(data=d, r=8, **kw) for d in datas]
    many = batch.solve_lockstep(datas, 8, **kw)
    for a, b in zip(one, many):
        assert not isinstance(b, Exception), b
        same(a, b)


def test_batch_time_budget_and_iteration_budget(hip_abi):
    """Per-item budgets: an item with max_local_iters = 2 stops after 2 iterations (exit 2) while its neighbours run on."""
    datas = [make_data("maxcut", s, 50, 0.2)[0] for s in (1, 2, 3)]
    A = [make_solver(hip_abi, d, 5, seed=1)[0] for d in datas]
    norms = [(d.normC(), float(np.linalg.norm(d.b))) for d in datas]
    cabi.batch_fg(hip_abi, A, [(nc, nb, 1, 1) for nc, nb in norms])
    args = [(nc, nb, 1, 1, 0, 0, 2.0, 0.0, -1e300, (2, 30, 9)[k], 0.0) for k, (nc, nb) in enumerate(norms)]
    res = cabi.batch_major_iteration(hip_abi, A, args)
    assert [r[4] for r in res] == [2, 30, 9] and [r[5] for r in res] == [2, 2, 2]
    for s in A:
        s.close()


def test_edge_path_group_shares_its_launches_bit_for_bit(hip_abi):
    """Lovász-θ instances (multi-launch edge path: too many constraints for the resident route) in one batch call: those of
    one launch shape run their while loops behind SHARED launches (k_group.h: blockIdx.y ↔ instance, the single-instance
    kernel bodies) — every output and the whole device state equal the single-instance calls bit for bit, with per-item
    iteration budgets, through four rounds (λ update on / off), next to an instance of another route."""
    specs = [("lovasz_theta", 1, 40), ("lovasz_theta", 2, 40), ("lovasz_theta", 3, 44), ("maxcut", 4, 30), ("lovasz_theta", 5, 38)]
    datas = [make_data(f, s, n, 0.3)[0] for f, s, n in specs]
    A = [make_solver(hip_abi, d, 6, seed=7)[0] for d in datas]
    B = [make_solver(hip_abi, d, 6, seed=7)[0] for d in datas]
    norms = [(d.normC(), float(np.linalg.norm(d.b))) for d in datas]
    assert cabi.batch_fg(hip_abi, A, [(nc, nb, 1, 1) for nc, nb in norms]) == \
        [s.fg(nc, nb, True, True) + (s.obj,) for s, (nc, nb) in zip(B, norms)]
    for rnd, upd in enumerate((0, 1, 1, 0)):
        sig = 2.0 * (rnd + 1)
        args = [(nc, nb, 1, 1, 0, upd, sig, 1e-3, 1e-30, (7, 30, 12, 9, 50)[k] + rnd, 0.0) for k, (nc, nb) in enumerate(norms)]
        ma = cabi.batch_major_iteration(hip_abi, A, args)
        mb = [s.major_iteration(*a) + (s.obj,) for s, a in zip(B, args)]
        assert ma == mb, rnd
    for a, b in zip(A, B):
        for name in ("Rt", "Gt", "dirt", "y", "λ", "primal_vio_raw"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
        for j in range(4):
            assert np.array_equal(a.get_factor(cabi.F_LBFGS_S + j), b.get_factor(cabi.F_LBFGS_S + j))
            assert np.array_equal(a.get_factor(cabi.F_LBFGS_Y + j), b.get_factor(cabi.F_LBFGS_Y + j))
        assert np.array_equal(a.get_vec(cabi.V_LBFGS_RHO), b.get_vec(cabi.V_LBFGS_RHO))
        assert a.σ == b.σ and a.obj == b.obj
    grouped = [s.stats()["group_launch_loops"] for s in A]
    assert grouped[3] == 0 and sum(1 for g in grouped if g == 4) >= 2, grouped
    assert all(s.stats()["group_launch_loops"] == 0 for s in B)
    # the state the group left is a state the single calls continue from
    for a, b, (nc, nb) in zip(A, B, norms):
        assert a.inner_loop(nc, nb, True, True, False, 0.0, -1e300, 3, 0.0, *a.fg(nc, nb)) == \
            b.inner_loop(nc, nb, True, True, False, 0.0, -1e300, 3, 0.0, *b.fg(nc, nb))
        a.close()
        b.close()


def test_lockstep_solves_of_an_edge_path_group(hip_abi, monkeypatch):
    """Whole Lovász-θ solves side by side (shared launches for the while loops) = the solves one by one, bit for bit; and
    with the shared launches switched off."""
    datas = [make_data("lovasz_theta", s, 36, 0.25)[0] for s in (1, 2, 3, 4)]
    kw = dict(KW, prior_trace_bound=1.0, maxtime=60.0)   # (tr X = 1 is a constraint of this family)
    one = [sj.sdplr(data=d, r=7, **kw) for d in datas]
    many = batch.solve_lockstep(datas, 7, **kw)
    for a, b in zip(one, many):
        assert not isinstance(b, Exception), b
        same(a, b)
    monkeypatch.setenv("SDPLR_HIP_NO_GROUP_LAUNCH", "1")
    for a, b in zip(one, batch.solve_lockstep(datas, 7, **kw)):
        same(a, b)


def test_handles_follow_the_device_set_for_the_process(hip_abi):
    """`sdplr_hip_set_device` is sticky (a new host thread starts on device 0, hipSetDevice is per thread): handles created
    and driven from fresh threads live on the device set once from the main thread, a device that does not exist is refused
    and leaves the setting alone, and the pools give their cached blocks back on request."""
    import threading
    assert hip_abi.set_device(0) == cabi.OK
    assert hip_abi.set_device(4096) != cabi.OK
    data = make_data("maxcut", 1, 40, 0.3)[0]
    nc, nb = data.normC(), float(np.linalg.norm(data.b))
    ref = make_solver(hip_abi, data, 6, seed=7)[0]
    want = ref.fg(nc, nb, True, True)
    out = {}

    def build():
        out["s"] = make_solver(hip_abi, data, 6, seed=7)[0]

    def drive():
        out["fg"] = out["s"].fg(nc, nb, True, True)
    for fn in (build, drive):
        t = threading.Thread(target=fn)
        t.start()
        t.join()
    assert out["fg"] == want
    assert cabi.batch_fg(hip_abi, [ref, out["s"]], [(nc, nb, 1, 1)] * 2) == [want + (ref.obj,)] * 2
    ref.close()
    out["s"].close()
    assert hip_abi.trim_pools() == cabi.OK
    again = make_solver(hip_abi, data, 6, seed=7)[0]      # (the pools refill)
    assert again.fg(nc, nb, True, True) == want
    again.close()
