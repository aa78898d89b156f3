"""dev tool: a plain run of one BASELINE configuration for rocprofv3 (scripts/pmc_config.sh): fg!, 40 + 100 inner iterations
on the route the library picks, one dual bound and one 232-step Lanczos run.  which ∈ lovasz | minbis | maxcut | wide64 |
wide128 (the MaxCut instance at rank 64 / 128: where rank doubling sends a solve)."""
import sys, os
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
rank = {"wide64": 64, "wide128": 128}.get(which, 32)
var = sj.build_solver(abi, data, rank, sj.BurerMonteiroConfig(seed=1, printlevel=0))
normC, normb = data.normC(), float(np.linalg.norm(data.b))
st = var.fg(normC, normb)
run = lambda s, k: var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *s)[:3]
st = run(st, 40)
st = run(st, 96)
v0 = np.random.Generator(np.random.PCG64(5)).standard_normal(data.n)
var.dual_obj(float(data.n), 0, v0)
var.lanczos(232, v0)
print("dims", var.dims(), "state", st)
