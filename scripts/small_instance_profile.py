"""dev tool: where the wall time of ONE small solve goes (BASELINE config 5: Gset G1, n = 800, rank 10, tol 1e-2)"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gset_G1_G9.npz"))
A = problems.graph_from_edges(int(z["G1_n"]), z["G1"])
abi = sj.load_hip()
def solve():
    data = problems.maxcut_data(A)
    return sj.sdplr(data=data, r=10, ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=float(A.shape[0]), printlevel=0) if False else None
C, As, bs = problems.maxcut(A)
def run():
    return sj.sdplr(C, As, bs, 10, ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=float(A.shape[0]), printlevel=0)
run()
t0 = time.perf_counter(); res = run(); dt = time.perf_counter() - t0
print("one solve", round(1e3 * dt, 2), "ms; iterations", res["iter"], "majors", res.get("majoriter"), "primal_time", res.get("primaltime"), "dual_time", res.get("dual_time"))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
