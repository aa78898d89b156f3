"""dev tool: how much faster is the gather kernel when the graph has locality?  Same n, same degree as the
north-star instance, edges restricted to |i − j| ≤ band (band = n/2 is the unrestricted case)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems

abi = sj.load_hip()
n, deg = bench.N_NODES, bench.N_NODES * bench.P_EDGE
rng = np.random.default_rng(1)
for band in (n // 2, 25000, 12500, 6000, 3000, 1000):
    m = int(n * deg / 2)
    i = rng.integers(0, n, m)
    j = (i + rng.integers(1, band + 1, m)) % n
    A = problems.graph_from_edges(n, np.stack([i, j], 1))
    data = problems.maxcut_data(A)
    var = sj.build_solver(abi, data, bench.RANK_R, sj.BurerMonteiroConfig(seed=0, printlevel=0))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    st = var.fg(normC, normb)
    st = bench.run_fixed(var, normC, normb, st, 40)
    var.profile_enable(True)
    bench.run_fixed(var, normC, normb, st, 40)
    p = var.profile()
    row = {k: 1e3 * v[1] / v[0] for k, v in p.items() if v[0] >= 40}
    print(f"band {band:6d} ({band*2*256/1e6:6.1f} MB window)  nnz {A.nnz}  " +
          "  ".join(f"{k} {v:6.1f}" for k, v in sorted(row.items(), key=lambda kv: -kv[1])[:4]), flush=True)
    var.close()
