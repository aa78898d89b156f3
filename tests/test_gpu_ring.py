"""The ring form of the L-BFGS history (csrc/k_dense.h): inside the device-driven loop of the P-less singleton fast path
lbfgshis.vecs[j].s / .y (src/lbfgs.jl:4-12) are kept as (α_j, dir_j) and (G_j, G_{j+1}) — the step kernel stores neither
s_j nor y_j.  fl(α·d) and fl(g' − g) are the values lbfgs_update! (src/lbfgs.jl:142-145) would have stored, so the
iterates must be those of the stored form BIT FOR BIT, and whatever looks at the arena outside the loop must find the stored
form.  SDPLR_HIP_NO_RING (read at solver construction) keeps the stored form inside the loop: the reference of these tests;
the CPU oracle is the second one."""
import numpy as np
import pytest

from helpers import make_solver
from sdplrplus_jl_amd import cabi, problems

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def _state(s_, h):
    S = [s_.get_factor(cabi.F_LBFGS_S + j).copy() for j in range(h)]
    Y = [s_.get_factor(cabi.F_LBFGS_Y + j).copy() for j in range(h)]
    return dict(R=s_.Rt.copy(), G=s_.Gt.copy(), D=s_.get_factor(cabi.F_DIRT).copy(), S=S, Y=Y,
                rho=s_.get_vec(cabi.V_LBFGS_RHO).copy(), y=s_.y.copy())


def _same(a, b):
    for k in ("R", "G", "D", "rho", "y"):
        assert np.array_equal(a[k], b[k]), k
    for j, (x, z) in enumerate(zip(a["S"], b["S"])):
        assert np.array_equal(x, z), ("s", j)
    for j, (x, z) in enumerate(zip(a["Y"], b["Y"])):
        assert np.array_equal(x, z), ("y", j)


@pytest.mark.parametrize("r,h,n", [(32, 4, 600), (6, 4, 500), (8, 3, 400), (16, 2, 300), (10, 1, 300), (64, 4, 200), (128, 4, 150)])
def test_ring_form_is_the_stored_form_bit_for_bit(hip_abi, oracle_abi, monkeypatch, r, h, n):
    """fg! → 3 iterations → 9 more (the ring wraps: h + 1 positions) → look → 4 more from the stored pairs (for its first h
    iterations such a ring reads the pairs older than itself from their slots) → look: R, G, dirt, every s_j, y_j, ρ and y
    equal the stored-form run bitwise both times; the oracle agrees to 1e-8."""
    data = problems.maxcut_data(problems.gnp_graph(n, 8.0 / n, 100 + r))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))

    def run(abi, ring):
        if ring:
            monkeypatch.delenv("SDPLR_HIP_NO_RING", raising=False)
        else:
            monkeypatch.setenv("SDPLR_HIP_NO_RING", "1")
        s_, _ = make_solver(abi, data, r, seed=3, h=h)
        st = s_.fg(normC, normb)
        out1 = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 3, 0.0, *st)
        out2 = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 9, 0.0, *out1[:3])
        stats_before = s_.stats() if abi is hip_abi else None
        a = _state(s_, h)
        out3 = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 4, 0.0, *out2[:3])   # a new ring, entered on the stored pairs
        b = _state(s_, h)
        stats = s_.stats() if abi is hip_abi else None
        s_.close()
        return (out1, out2, out3), a, b, stats_before, stats

    outs_r, a_r, b_r, sb, st = run(hip_abi, True)
    outs_s, a_s, b_s, _, st_s = run(hip_abi, False)
    assert sb["ring_history_loops"] == 2 and sb["ring_materializations"] == 0   # the two calls shared one ring
    assert st["ring_history_loops"] == 3 and st["ring_materializations"] == 2
    assert st_s["ring_history_loops"] == 0 and st_s["p_less_loops"] == 3
    assert outs_r == outs_s
    _same(a_r, a_s)
    _same(b_r, b_s)
    outs_o, a_o, b_o, _, _ = run(oracle_abi, False)
    for x, z in zip(outs_r, outs_o):
        assert x[4] == z[4] and np.allclose(x[:3], z[:3], rtol=1e-8)
    assert rel(b_r["R"], b_o["R"]) < 1e-8 and rel(b_r["G"], b_o["G"]) < 1e-7


def test_ring_after_the_relative_decrease_exit(hip_abi, monkeypatch):
    """A step that is not followed by lbfgs_update! (src/sdplr.jl:238-241) leaves the slot lbfgs_dir! had claimed with
    y = −G_old and dirt unscaled: the ring is turned back into exactly that, and the next loop carries on from it."""
    data = problems.maxcut_data(problems.gnp_graph(500, 0.02, 5))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    r, h = 8, 4

    def run(ring):
        if ring:
            monkeypatch.delenv("SDPLR_HIP_NO_RING", raising=False)
        else:
            monkeypatch.setenv("SDPLR_HIP_NO_RING", "1")
        s_, _ = make_solver(hip_abi, data, r, seed=9, h=h)
        st = s_.fg(normC, normb)
        out1 = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 6, 0.0, *st)
        # fprec·eps so large that the next step's relative decrease is "too small": the loop leaves before the update
        out2 = s_.inner_loop(normC, normb, True, True, False, 0.0, 1e300, 6, 0.0, *out1[:3])
        a = _state(s_, h)
        out3 = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 5, 0.0, *out2[:3])
        b = _state(s_, h)
        s_.close()
        return (out1, out2, out3), a, b

    o_r, a_r, b_r = run(True)
    o_s, a_s, b_s = run(False)
    assert o_r[1][5] == 1 and o_r[1][4] == 1   # EXIT_RELDELTA after one step
    assert o_r == o_s
    _same(a_r, a_s)
    _same(b_r, b_s)


def test_ring_and_lbfgs_clear(hip_abi, monkeypatch):
    """lbfgs_clear! on a ring keeps what the stored form keeps — G and dirt = s_latest — and the next loop starts a new
    ring; major_iteration (clear → fg! → loop in one call) goes the same way.  The dual bound between loop and clear (the
    order of src/sdplr.jl:280-389) reads nothing of the history and leaves the ring alone."""
    data = problems.maxcut_data(problems.gnp_graph(450, 0.03, 6))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    r, h = 12, 4

    def run(ring):
        if ring:
            monkeypatch.delenv("SDPLR_HIP_NO_RING", raising=False)
        else:
            monkeypatch.setenv("SDPLR_HIP_NO_RING", "1")
        s_, _ = make_solver(hip_abi, data, r, seed=2, h=h)
        st = s_.fg(normC, normb)
        out1 = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 7, 0.0, *st)
        v0 = np.random.Generator(np.random.PCG64(1)).standard_normal(data.n)
        dual = s_.dual_obj(float(data.n), 0, v0)        # (works on S(y): a ring stays a ring — counted below)
        s_.lbfgs_clear()
        a = _state(s_, h)
        st = s_.fg(normC, normb)
        out2 = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 6, 0.0, *st)
        out3 = s_.major_iteration(normC, normb, True, True, False, True, 4.0, 0.0, -1e300, 8, 0.0)
        b = _state(s_, h)
        stats = s_.stats()
        s_.close()
        return (out1, out2, out3, dual), a, b, stats

    o_r, a_r, b_r, st = run(True)
    o_s, a_s, b_s, _ = run(False)
    assert st["ring_history_loops"] == 3 and st["ring_materializations"] == 1   # (only the closing look)
    assert o_r == o_s
    _same(a_r, a_s)
    _same(b_r, b_s)


@pytest.mark.parametrize("family,toggle,r", [("minbis", None, 8), ("minbis", None, 32), ("cutnorm", None, 8),
                                              ("maxcut", "SDPLR_HIP_NO_PDROP", 16), ("maxcut", "SDPLR_HIP_NO_LSHEAD", 16),
                                              ("lovasz", None, 8)])
def test_ring_form_on_the_other_singleton_loops(hip_abi, oracle_abi, monkeypatch, family, toggle, r):
    """The ring form also carries the P-based step kernel (MinBisection's rank-one constraint with its projections out of the
    tile kernel; a MaxCut loop told to keep P) and CutNorm's P-less loop: bitwise the stored form after a chain of calls that
    wraps the ring, materialises it, and enters a new one on the stored pairs; 1e-8 against the oracle.  A shape the ring
    form does not take (the P-less kernel without its line-search head) must simply stay on the stored form."""
    g = problems.gnp_graph(360, 0.03, 21)
    data = {"minbis": problems.minimum_bisection_data, "cutnorm": problems.cutnorm_data, "maxcut": problems.maxcut_data,
            "lovasz": problems.lovasz_theta_data}[family](g)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    h = 4
    if toggle:
        monkeypatch.setenv(toggle, "1")

    def run(abi, ring):
        if ring:
            monkeypatch.delenv("SDPLR_HIP_NO_RING", raising=False)
        else:
            monkeypatch.setenv("SDPLR_HIP_NO_RING", "1")
        s_, _ = make_solver(abi, data, r, seed=13, h=h)
        st = s_.fg(normC, normb)
        out1 = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 2, 0.0, *st)
        out2 = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 7, 0.0, *out1[:3])
        a = _state(s_, h)
        out3 = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 3, 0.0, *out2[:3])
        b = _state(s_, h)
        stats = s_.stats() if abi is hip_abi else None
        s_.close()
        return (out1, out2, out3), a, b, stats

    o_r, a_r, b_r, st = run(hip_abi, True)
    o_s, a_s, b_s, st_s = run(hip_abi, False)
    expect_ring = toggle != "SDPLR_HIP_NO_LSHEAD" and family != "lovasz"   # (the edge path: measured, no gain — stored form)
    assert st["ring_history_loops"] == (3 if expect_ring else 0) and st_s["ring_history_loops"] == 0
    assert o_r == o_s
    _same(a_r, a_s)
    _same(b_r, b_s)
    o_o, a_o, b_o, _ = run(oracle_abi, False)
    for x, z in zip(o_r, o_o):
        assert x[4] == z[4] and np.allclose(x[:3], z[:3], rtol=1e-8)
    assert rel(b_r["R"], b_o["R"]) < 1e-8


def test_ring_on_the_eager_route_and_under_the_profiler(hip_abi, monkeypatch):
    """hipGraph batches (the suite's default), eager launches (SDPLR_HIP_NO_GRAPH) and the per-kernel event timing all enqueue
    the same ring kernels: the same state bitwise.  A budget of one and of two iterations (shorter than a batch) too."""
    data = problems.maxcut_data(problems.gnp_graph(420, 0.03, 33))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    r, h = 16, 4

    def run(mode, ring=True):
        monkeypatch.delenv("SDPLR_HIP_NO_GRAPH", raising=False)
        monkeypatch.delenv("SDPLR_HIP_NO_RING", raising=False)
        if mode == "eager":
            monkeypatch.setenv("SDPLR_HIP_NO_GRAPH", "1")
        if not ring:
            monkeypatch.setenv("SDPLR_HIP_NO_RING", "1")
        s_, _ = make_solver(hip_abi, data, r, seed=8, h=h)
        st = s_.fg(normC, normb)
        outs = []
        for k in (1, 2, 9, 1, 5):
            if mode == "profile" and k == 9:
                s_.profile_enable(True)
            out = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *st)
            st = out[:3]
            outs.append(out)
        if mode == "profile":
            prof = s_.profile()
            s_.profile_enable(False)
            assert prof["fast_step"][0] >= 15 and prof["lbfgs_dir"][0] >= 15
        a = _state(s_, h)
        stats = s_.stats()
        s_.close()
        return outs, a, stats

    o_g, a_g, st_g = run("graph")
    assert st_g["ring_history_loops"] == 5 and st_g["ring_materializations"] == 1
    for mode in ("eager", "profile"):
        o_m, a_m, st_m = run(mode)
        assert st_m["ring_history_loops"] == 5
        assert o_m == o_g
        _same(a_m, a_g)
    o_s, a_s, _ = run("graph", ring=False)
    assert o_s == o_g
    _same(a_s, a_g)


def test_ring_left_by_the_time_budget_and_by_a_loop_that_does_not_start(hip_abi):
    """The time budget ends a loop wherever the device happens to be (no two runs alike): whatever the ring held, the stored
    form it is turned back into must be consistent — ρ_j = 1/⟨s_j, y_j⟩ for every pair, the newest pair's s = dirt,
    G the gradient from scratch at R — and a loop that leaves before its first iteration (‖G‖ already below the tolerance)
    leaves a ring of no pairs, i.e. the history as it found it."""
    data = problems.maxcut_data(problems.gnp_graph(2000, 0.01, 3))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    r, h = 32, 4
    s_, _ = make_solver(hip_abi, data, r, seed=1, h=h)
    st = s_.fg(normC, normb)
    out = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 100000, 2e-3, *st)
    assert out[5] == 3 and out[4] >= h + 2            # EXIT_TIME, after the ring has wrapped
    a = _state(s_, h)
    rho = s_.get_vec(cabi.V_LBFGS_RHO)
    latest = int(s_.get_scalar(cabi.S_LBFGS_LATEST))
    for j in range(h):
        assert abs(rho[j] * float(np.sum(a["S"][j] * a["Y"][j])) - 1.0) < 1e-10, j
    assert np.array_equal(a["D"], a["S"][latest - 1])
    s_.g()
    assert rel(a["G"], s_.Gt) < 1e-10
    assert s_.stats()["ring_history_loops"] == 1 and s_.stats()["ring_materializations"] == 1
    # a loop that does not start: tolerance above ‖G‖
    st = s_.fg(normC, normb)
    out2 = s_.inner_loop(normC, normb, True, True, False, 10.0 * st[1], -1e300, 50, 0.0, *st)
    assert out2[4] == 0 and out2[5] == 0
    b = _state(s_, h)
    for j in range(h):
        assert np.array_equal(a["S"][j], b["S"][j]) and np.array_equal(a["Y"][j], b["Y"][j])
    assert np.array_equal(b["D"], a["D"])
    s_.close()


@pytest.mark.parametrize("fresh_g", [False, True])
def test_ring_takes_the_steepest_descent_fallback(hip_abi, oracle_abi, monkeypatch, fresh_g):
    """src/sdplr.jl:201-205 on the ring form: the seam kernel finds ⟨dir, G⟩ ≥ 0 from the Gram data and k_lbfgs_dir_ring writes
    −G — leaving G itself unflipped (the flip of :203 is undone by y_j = G_new − G_old in the stored form; on a ring nothing
    reads G in between).  The newest stored pair is replaced by (s = G, y = 0, ρ ≪ 0) as in
    test_inner_loop_takes_the_steepest_descent_fallback; the next loop enters a ring on that history and falls back in its
    first iteration — on the P-based step kernel (the host wrote behind G's back) and, after a fresh g!, on the P-less one.
    Bitwise the stored form, 1e-8 the oracle."""
    data = problems.maxcut_data(problems.gnp_graph(200, 0.06, 17))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    r, h = 8, 4

    def run(abi, ring):
        if ring:
            monkeypatch.delenv("SDPLR_HIP_NO_RING", raising=False)
        else:
            monkeypatch.setenv("SDPLR_HIP_NO_RING", "1")
        s_, _ = make_solver(abi, data, r, seed=5, h=h)
        st = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, h + 2, 0.0, *s_.fg(normC, normb))[:3]
        j = int(s_.get_scalar(cabi.S_LBFGS_LATEST)) - 1
        G = s_.Gt
        s_.set_factor(cabi.F_LBFGS_S + j, G)
        s_.set_factor(cabi.F_LBFGS_Y + j, np.zeros_like(G))
        rho = s_.get_vec(cabi.V_LBFGS_RHO)
        rho[j] = -1e6 / float(np.sum(G * G))
        s_.set_vec(cabi.V_LBFGS_RHO, rho)
        if fresh_g:
            s_.g()
        outs = []
        for k in (1, 1, 6):
            out = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *st)
            st = out[:3]
            outs.append(out)
        a = _state(s_, h)
        stats = s_.stats() if abi is hip_abi else None
        s_.close()
        return outs, a, stats

    o_r, a_r, st = run(hip_abi, True)
    o_s, a_s, _ = run(hip_abi, False)
    assert st["ring_history_loops"] == 4
    assert o_r == o_s
    _same(a_r, a_s)
    o_o, a_o, _ = run(oracle_abi, False)
    # the first step is a steepest-descent one: α·‖G‖² = −⟨dir, G⟩·α, far from the L-BFGS step it replaces
    for x, z in zip(o_r, o_o):
        assert x[4] == z[4] and np.allclose(x[:3], z[:3], rtol=1e-8)
    assert rel(a_r["R"], a_o["R"]) < 1e-8


def test_random_call_sequences_on_a_ring_handle_and_a_stored_form_handle(hip_abi):
    """scripts/stress_ring.py, 25 sequences: random interleavings of loops (every kind of exit), lbfgs_clear!, fg!, g!, the
    dual bound, the λ update, major_iteration, looks at and writes to G / the history / λ, stand-alone lbfgs_dir! and
    lbfgs_update!, a line search and a rank reset — on two handles of the same instance, one keeping the history in ring
    form inside its loops, one storing it: equal bit for bit after every call and in full at the end."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "stress_ring.py")
    spec = importlib.util.spec_from_file_location("_stress_ring", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, degenerate, ring_loops, mats = mod.run(7, 25, hip=hip_abi, nmax=500)
    assert bad == 0
    assert ring_loops >= 25 and mats >= 10      # (the sweep did exercise the hand-overs)
