"""The all-cores timing build of the oracle (libsdplr_oracle_omp.so: the same source with -fopenmp; what
bench.py's cpu_baseline.all_cores times) computes the same thing as the one-thread checker: parallel reductions
change the summation order, nothing else."""
import ctypes as C

import numpy as np

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
from helpers import make_solver


def test_openmp_build_matches_the_one_thread_checker(oracle_abi):
    from oracle import oracle
    oracle.build()
    omp = sj.CABI(oracle.LIB_OMP, "sdplr_oracle_")
    gomp = C.CDLL("libgomp.so.1")
    gomp.omp_set_num_threads(4)
    data = problems.maxcut_data(problems.gnp_graph(6000, 4e-3, 3))      # N = 6000·12 > the 65 536-element threshold
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    outs = []
    for abi in (oracle_abi, omp):
        s, _ = make_solver(abi, data, 12, seed=1)
        st = s.fg(normC, normb)
        res = s.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 6, 0.0, *st)
        outs.append((st, res, s.Rt, s.Gt))
        s.close()
    (st1, r1, R1, G1), (st2, r2, R2, G2) = outs
    assert np.allclose(st1, st2, rtol=1e-12)
    assert r1[4] == r2[4] == 6 and np.allclose(r1[:3], r2[:3], rtol=1e-9)
    assert np.max(np.abs(R1 - R2)) < 1e-9 * (1 + np.max(np.abs(R1)))
    assert np.max(np.abs(G1 - G2)) < 1e-8 * (1 + np.max(np.abs(G1)))


def test_openmp_build_is_reproducible_run_to_run():
    """The all-cores build's only cross-thread sum (ddot) adds its per-thread range sums in thread order: two runs with the
    same thread count give the same bits (an OpenMP `reduction` would combine in arrival order — on MinBisection n = 1e5 two
    runs of the same solve once took different numbers of major iterations)."""
    from oracle import oracle
    oracle.build()
    omp = sj.CABI(oracle.LIB_OMP, "sdplr_oracle_")
    gomp = C.CDLL("libgomp.so.1")
    gomp.omp_set_num_threads(4)
    data = problems.minimum_bisection_data(problems.gnp_graph(6000, 4e-3, 5))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    runs = []
    for _ in range(3):
        s, _cfg = make_solver(omp, data, 12, seed=2)
        st = s.fg(normC, normb)
        res = s.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 15, 0.0, *st)
        runs.append((st, res, s.Rt.copy(), s.Gt.copy()))
        s.close()
    for st, res, R, G in runs[1:]:
        assert st == runs[0][0] and res == runs[0][1]
        assert np.array_equal(R, runs[0][2]) and np.array_equal(G, runs[0][3])
