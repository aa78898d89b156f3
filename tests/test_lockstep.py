"""The lockstep batch driver (sdplrplus.jl_amd/batch.py: solve_lockstep, serve_batch) and the stepper it drives
(sdplr.py: sdplr_steps): B solves side by side must be B times `sdplr(data=…)`, result for result.  On the CPU the oracle's
ABI stands in for the library (its batch calls are loops over the single-instance functions), which pins the driver's
control flow; tests/test_gpu_lockstep.py repeats the comparison on the HIP library, where a batch call is one launch."""
import numpy as np
import pytest

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, cabi, problems
from sdplrplus_jl_amd.sdplr import REQ_FG, sdplr_steps, serve, build_solver
from sdplrplus_jl_amd.structs import BurerMonteiroConfig

from helpers import make_data

KW = dict(ptol=1e-2, objtol=1e-2, maxtime=60.0, printlevel=0, prior_trace_bound=30.0)


def instances():
    out = []
    for seed, n in ((1, 24), (2, 30), (3, 18), (4, 30)):
        data, *_ = make_data("maxcut", seed, n, 0.3)
        out.append(data)
    data, *_ = make_data("minimum_bisection", 5, 20, 0.3)      # a low-rank constraint: another route, same batch
    out.append(data)
    return out


def same(a, b):
    assert a["iter"] == b["iter"] and a["majoriter"] == b["majoriter"]
    assert a["obj"] == b["obj"] and a["max_dual_value"] == b["max_dual_value"]
    assert np.array_equal(a["Rt"], b["Rt"]) and np.array_equal(a["lambda"], b["lambda"])
    assert a["schedule"] == b["schedule"]


def test_lockstep_equals_one_by_one_on_the_oracle(oracle_abi):
    datas = instances()
    one = [sj.sdplr(data=d, r=4, abi=oracle_abi, **KW) for d in datas]
    many = batch.solve_lockstep(datas, 4, abi=oracle_abi, setup_workers=2, **KW)
    for a, b in zip(one, many):
        same(a, b)


@pytest.mark.parametrize("cap", [1, 3, 7])
def test_capped_launches_resume_to_the_same_solve(oracle_abi, cap):
    """A lockstep round caps the inner iterations per launch (SDPLR_LOCKSTEP_CAP) and RESUMES the loops that ran into the
    cap (update_lambda = MAJOR_RESUME: no λ update, no lbfgs_clear!, no fg!) while the instances that have left theirs
    move on: the solves must be those of the uncapped driver and of sdplr() one by one, result for result."""
    datas = instances()
    one = [sj.sdplr(data=d, r=4, abi=oracle_abi, **KW) for d in datas]
    assert max(max(row[1] for row in x["schedule"]) for x in one) > 7      # (some inner loop is longer than every cap tried)
    uncapped = batch.solve_lockstep(datas, 4, abi=oracle_abi, setup_workers=1, iteration_cap=0, **KW)
    capped = batch.solve_lockstep(datas, 4, abi=oracle_abi, setup_workers=1, iteration_cap=cap, **KW)
    for a, b, c in zip(one, uncapped, capped):
        same(a, b)
        same(a, c)


def test_cap_bookkeeping():
    """_Cap alone: a budget of 10 at cap 4 goes out as 4 (major), 4 (resume), 2 (resume); the stepper sees one answer with
    all ten iterations; an early exit inside a launch ends the sequence there."""
    cap = batch._Cap(4)
    req = ("major_iteration", 1.0, 2.0, True, True, False, 1, 2.0, 1e-3, 1e-9, 10, 60.0)
    out = cap.outgoing(0, req)
    assert out[10] == 4 and out[6] == 1
    assert cap.incoming(0, (5.0, 0.5, 0.1, 0.3, 4, batch.EXIT_ITERS, -1.0)) is None
    out = cap.outgoing(0, req)
    assert out[6] == cabi.MAJOR_RESUME and out[10] == 4 and out[12:] == (5.0, 0.5, 0.1)
    assert cap.incoming(0, (4.0, 0.4, 0.1, 0.3, 4, batch.EXIT_ITERS, -1.5)) is None
    out = cap.outgoing(0, req)
    assert out[10] == 2
    assert cap.incoming(0, (3.0, 0.3, 0.1, 0.2, 2, batch.EXIT_ITERS, -2.0)) == (3.0, 0.3, 0.1, 0.2, 10, batch.EXIT_ITERS, -2.0)
    assert not cap.open
    out = cap.outgoing(1, req)
    assert cap.incoming(1, (3.0, 0.3, 0.1, 0.2, 3, 0, -2.0)) == (3.0, 0.3, 0.1, 0.2, 3, 0, -2.0) and not cap.open
    small = req[:10] + (3, 60.0)
    assert cap.outgoing(2, small) == small and cap.incoming(2, (1.0, 0.1, 0.1, 0.1, 3, batch.EXIT_ITERS, 0.0))[4] == 3


def test_lockstep_through_rank_doublings(oracle_abi):
    """Instances that double their rank on the way (rank_update!: reset_rank + a fresh point, then fg! and an inner loop that
    are NOT batch calls) stay in step with the others."""
    datas = [make_data("maxcut", seed, n, 0.3)[0] for seed, n in ((1, 40), (2, 60), (4, 50))]
    kw = dict(ptol=1e-3, objtol=1e-4, maxtime=60.0, printlevel=0, prior_trace_bound=60.0, rankupd_tol=2)
    one = [sj.sdplr(data=d, r=2, abi=oracle_abi, **kw) for d in datas]
    many = batch.solve_lockstep(datas, 2, abi=oracle_abi, setup_workers=1, **kw)
    assert [x["r"] for x in one] == [4, 8, 4] or all(x["r"] > 2 for x in one)
    for a, b in zip(one, many):
        same(a, b)


def test_stepper_yields_the_documented_requests(oracle_abi):
    data = instances()[0]
    config = BurerMonteiroConfig()
    for k, v in KW.items():
        setattr(config, k, v)
    var = build_solver(oracle_abi, data, 4, config)
    steps = sdplr_steps(data, var, config)
    kinds = []
    try:
        req = next(steps)
        while True:
            kinds.append(req[0])
            req = steps.send(serve(var, req))
    except StopIteration as done:
        ans = done.value
    var.close()
    # the fg! of src/sdplr.jl:170 rides the first major_iteration; the one of :396 closes the solve
    assert kinds[0] == "major_iteration" and kinds[-1] == REQ_FG
    assert set(kinds) <= {"fg", "major_iteration", "inner_loop", "dual_obj"} and "dual_obj" in kinds
    assert kinds.count("major_iteration") + kinds.count("inner_loop") == ans["majoriter"]


def test_an_instance_that_fails_does_not_take_the_batch_down(oracle_abi, monkeypatch):
    """The first dual_obj request of instance 1 raises; its stepper gets the exception (and ends with it), the other
    instances finish with the results they have on their own."""
    datas = instances()[:3]

    class Boom(RuntimeError):
        pass

    victim = []
    real_build, real_dual = batch.build_solver, cabi.batch_dual_obj

    def build(abi, data, r, config):
        s = real_build(abi, data, r, config)
        if data is datas[1]:
            victim.append(s)
        return s

    def failing(abi, solvers, args):
        out = real_dual(abi, solvers, args)
        return [Boom("injected") if sv is victim[0] else o for sv, o in zip(solvers, out)]

    monkeypatch.setattr(batch, "build_solver", build)
    monkeypatch.setattr(cabi, "batch_dual_obj", failing)
    res = batch.solve_lockstep(datas, 4, abi=oracle_abi, setup_workers=1, **KW)
    monkeypatch.undo()
    assert isinstance(res[1], Boom)
    same(sj.sdplr(data=datas[0], r=4, abi=oracle_abi, **KW), res[0])
    same(sj.sdplr(data=datas[2], r=4, abi=oracle_abi, **KW), res[2])


def test_batch_calls_reject_a_handle_listed_twice(hip_abi):
    arr = (cabi.FgItem * 2)()
    arr[0].s = arr[1].s = 12345          # (never dereferenced: the duplicate check comes first)
    assert hip_abi.batch_fg(2, arr) == cabi.ERR_INVALID_ARG
    assert hip_abi.batch_fg(0, None) == cabi.OK
    assert hip_abi.batch_major_iteration(-1, None) == cabi.ERR_INVALID_ARG


def test_a_refused_batch_call_leaves_no_item_looking_served(hip_abi):
    """A batch call that returns before it has served its items (here: the duplicate-handle check) leaves every item at
    ERR_UNSERVED — never OK beside zeroed outputs — and the Python wrapper raises on the call's own return code."""
    for Item, fn in ((cabi.FgItem, hip_abi.batch_fg), (cabi.MajorItem, hip_abi.batch_major_iteration), (cabi.DualItem, hip_abi.batch_dual_obj)):
        arr = (Item * 3)()
        arr[0].s = arr[2].s = 12345
        arr[1].s = 54321
        for k in range(3):
            arr[k].status = cabi.OK
        assert fn(3, arr) == cabi.ERR_INVALID_ARG
        assert [int(arr[k].status) for k in range(3)] == [cabi.ERR_UNSERVED] * 3

    class Fake:
        def __init__(self, h):
            import ctypes
            self._h = ctypes.c_void_p(h)
    with pytest.raises(cabi.SdplrError):
        cabi.batch_fg(hip_abi, [Fake(12345), Fake(12345)], [(1.0, 1.0, 1, 1)] * 2)
