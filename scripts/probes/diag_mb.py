import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
from oracle import oracle
from helpers import make_solver
z = np.load('/root/repo/tests/golden/gset_G1_G9.npz')
g = problems.graph_from_edges(int(z["G2_n"]), z["G2"])
data = problems.minimum_bisection_data(g)
normC, normb = data.normC(), float(np.linalg.norm(data.b))
def rel(a, b): return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
sol = {}
for route in ("resident", "launches", "oracle"):
    if route == "launches": os.environ["SDPLR_HIP_NO_RESIDENT_LR"] = "1"
    s_, _ = make_solver(oracle.abi() if route == "oracle" else sj.load_hip(), data, 10, seed=2)
    os.environ.pop("SDPLR_HIP_NO_RESIDENT_LR", None)
    sol[route] = [s_, s_.fg(normC, normb)]
for it in range(1, 26):
    for route in sol:
        s_, st = sol[route]
        sol[route][1] = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 1, 0.0, *st)[:3]
    R = {k: v[0].Rt for k, v in sol.items()}
    print(it, "res-launch %.2e  res-oracle %.2e  launch-oracle %.2e" % (rel(R["resident"], R["launches"]), rel(R["resident"], R["oracle"]), rel(R["launches"], R["oracle"])), "L", sol["resident"][1][0], sol["oracle"][1][0])
print(sol["resident"][0].stats())
