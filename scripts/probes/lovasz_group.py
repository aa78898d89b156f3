"""dev probe: Lovász-θ on Gset G1–G9 (rank 10) through the lockstep driver, wall and iterations; SDPLR_HIP_NO_GROUP_LAUNCH=1
for the single-instance launches.  argv[1] = number of instances (default 9)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, problems
abi = sj.load_hip()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 9
z = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "gset_G1_G9.npz"))
graphs = [problems.graph_from_edges(int(z[f"G{i}_n"]), z[f"G{i}"]) for i in range(1, k + 1)]
ds = [problems.lovasz_theta_data(g) for g in graphs]
kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=1.0, maxtime=120.0)
t0 = time.perf_counter()
r = batch.solve_local(ds, 0, 1, 10, concurrency=8, lockstep=True, **kw)
w = time.perf_counter() - t0
print(f"instances {k} wall {w:.3f} s iterations {int(r[:, 3].sum())} obj {r[:, 1]}")
