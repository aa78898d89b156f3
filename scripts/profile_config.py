"""dev tool: per-kernel device time of the inner iteration for a BASELINE config (eager, event-timed)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
which = sys.argv[1]
if which == "lovasz":
    data = problems.lovasz_theta_data(problems.chung_lu_graph(50_000, 10.0, 2.5, 3))
elif which == "minbis":
    data = problems.minimum_bisection_data(problems.gnp_graph(100_000, 2e-4, 4))
else:
    data = problems.maxcut_data(problems.gnp_graph(100_000, 2e-4, 20240610))
var = sj.build_solver(abi, data, 32, sj.BurerMonteiroConfig(seed=1, printlevel=0))
normC, normb = data.normC(), float(np.linalg.norm(data.b))
st = var.fg(normC, normb)
run = lambda s, k: var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *s)[:3]
st = run(st, 40)
var.profile_enable(True)
st = run(st, 40)
p = var.profile()
tot = 0.0
for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:20s} {1e3*ms/40:8.1f} us/iter  ({c/40:.2f} launches/iter)")
    tot += 1e3 * ms / 40
print("sum", tot, "dims", var.dims())
if which in ("minbis", "maxcut", "lovasz"):
    var.dual_obj(float(data.n), 0, np.ones(data.n))
    var.profile_enable(True)
    v0 = np.random.Generator(np.random.PCG64(5)).standard_normal(data.n)
    var.lanczos(232, v0)
    for k, (c, ms) in sorted(var.profile().items(), key=lambda kv: -kv[1][1]):
        print(f"LZ {k:20s} {1e3*ms/232:8.2f} us/step  ({c/232:.2f} launches/step)")
