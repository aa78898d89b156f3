"""dev tool: does one small solve slow down when K of them run side by side (K host threads, K handles, K streams)?
Per-thread wall time of sdplr() on Gset G1 (rank 10, tol 1e-2) and of its pieces."""
import os, sys, time, json, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems, cabi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
A = problems.graph_from_edges(int(z["G1_n"]), z["G1"])
abi = sj.load_hip(); abi.device_synchronize()
data = problems.maxcut_data(A)
kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=800.0, printlevel=0)
sj.sdplr(data=data, r=10, **kw)
# time spent inside the library per entry point (monkey-patched wrappers)
acc = {}
lock = threading.Lock()
def wrap(name):
    f = getattr(cabi.DeviceSolver, name)
    def g(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return f(self, *a, **k)
        finally:
            dt = time.perf_counter() - t0
            with lock:
                acc[name] = acc.get(name, 0.0) + dt
    setattr(cabi.DeviceSolver, name, g)
for nm in ("major_iteration", "inner_loop", "dual_obj", "fg", "finalize", "set_sparse_coo", "close", "lbfgs_clear", "update_lambda", "set_vec", "set_factor", "get_vec", "get_factor", "set_scalar", "get_scalar"):
    wrap(nm)
for K in (1, 2, 4, 8, 16):
    acc.clear()
    times = []
    def work():
        for _ in range(4):
            t0 = time.perf_counter()
            sj.sdplr(data=data, r=10, **kw)
            times.append(time.perf_counter() - t0)
    th = [threading.Thread(target=work) for _ in range(K)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    n = 4 * K
    print(json.dumps({"threads": K, "solve_ms_mean": round(1e3 * sum(times) / n, 2), "aggregate_solves_per_s": round(n / wall, 1),
                      "per_solve_ms": {k: round(1e3 * v / n, 3) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}}))
