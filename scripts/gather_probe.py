"""dev tool: how fast are the gather kernels when the factor fits the per-XCD L2?  (r sweep at n = 1e5)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
A = problems.gnp_graph(bench.N_NODES, bench.P_EDGE, bench.GRAPH_SEED)
data = problems.maxcut_data(A)
lay = sj.preprocess_sparsecons(data.sparse)
normC, normb = data.normC(), float(np.linalg.norm(data.b))
for r in (2, 4, 8, 16, 32, 64):
    var = sj.build_solver(abi, data, r, sj.BurerMonteiroConfig(seed=0, printlevel=0), layout=lay)
    st = var.fg(normC, normb)
    st = var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 30, 0.0, *st)[:3]
    var.profile_enable(True)
    var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 40, 0.0, *st)
    p = var.profile()
    N = 8 * bench.N_NODES * r
    row = {k: 1e3 * p[k][1] / p[k][0] for k in ("spmm", "sddmm_linesearch", "lbfgs_dir", "lbfgs_update", "axpy_R")}
    gath_spmm = lay.nnzS * 8 * r / row["spmm"] / 1e6
    gath_sddmm = lay.nnzT * 2 * 8 * r * 2 / row["sddmm_linesearch"] / 1e6
    print(f"r={r:3d} factor={N/1e6:6.1f} MB  spmm {row['spmm']:7.1f} us ({gath_spmm:6.2f} TB/s gathered)  "
          f"sddmm {row['sddmm_linesearch']:7.1f} us ({gath_sddmm:6.2f} TB/s)  dir {row['lbfgs_dir']:6.1f}  upd {row['lbfgs_update']:6.1f}  axpy {row['axpy_R']:5.1f}", flush=True)
    var.close()
