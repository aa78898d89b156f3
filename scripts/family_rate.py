"""dev tool: inner-iteration rate of one BASELINE configuration (maxcut | minbis | lovasz), 3 × 200 iterations.
    python scripts/family_rate.py lovasz        (SDPLR_HIP_LIBRARY=… selects another build, e.g. the -DSDPLR_STAMPS one)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
fam = sys.argv[1] if len(sys.argv) > 1 else "maxcut"
abi = sj.load_hip()
if fam == "lovasz":
    data = problems.lovasz_theta_data(problems.chung_lu_graph(50_000, 10.0, 2.5, 3))
elif fam == "minbis":
    data = problems.minimum_bisection_data(problems.gnp_graph(100_000, 2e-4, 4))
else:
    data = problems.maxcut_data(problems.gnp_graph(100_000, 2e-4, 20240610))
var = sj.build_solver(abi, data, 32, sj.BurerMonteiroConfig(seed=1, printlevel=0))
normC, normb = data.normC(), float(np.linalg.norm(data.b))
st = var.fg(normC, normb)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    st = var.fg(normC, normb)
run = lambda s, k: var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *s)[:3]
st = run(st, 24)
best = 0
for _ in range(3):
    abi.device_synchronize(); t0 = time.perf_counter(); st = run(st, 200); abi.device_synchronize()
    best = max(best, 200 / (time.perf_counter() - t0))
print(fam, round(best), "it/s", flush=True)
