"""dev tool: Lanczos steps/s on the MaxCut north-star instance (`minbis`: MinBisection n = 1e5; `lovasz`: the Chung–Lu
Lovász-θ stand-in, value-form bands + hub rows): q steps as src/coreop.jl:402 gives them, 5 runs"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
which = sys.argv[1] if len(sys.argv) > 1 else "maxcut"
if which == "minbis":
    data = problems.minimum_bisection_data(problems.gnp_graph(100_000, 2e-4, 4))
elif which == "lovasz":
    data = problems.lovasz_theta_data(problems.chung_lu_graph(50_000, 10.0, 2.5, 3))
else:
    data = problems.maxcut_data(problems.gnp_graph(100_000, 2e-4, 20240610))
var = sj.build_solver(abi, data, 32, sj.BurerMonteiroConfig(seed=1, printlevel=0))
var.f()
v0 = np.random.Generator(np.random.PCG64(5)).standard_normal(data.n)
var.dual_obj(float(data.n), 0, v0)
Q = int(2 * np.ceil(np.sqrt(100.0) * np.log(data.n)))
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); al, be, k = var.lanczos(Q, v0); dt = time.perf_counter() - t0
    best = min(best, dt)
    print(k, f"{1e6 * dt / k:.2f} us/step  {k / dt:.0f} steps/s")
print("best", f"{Q / best:.0f} steps/s", "ritz", var.tridiag_mineig(al, be), var.stats())
