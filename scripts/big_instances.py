import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
for n, p, r in ((1_000_000, 2e-5, 16), (300_000, 1e-4, 64), (2_000_000, 4e-6, 8)):
    t0 = time.time()
    A = problems.gnp_graph(n, p, 11)
    data = problems.maxcut_data(A)
    var = sj.build_solver(abi, data, r, sj.BurerMonteiroConfig(seed=0, printlevel=0))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    st = var.fg(normC, normb)
    t1 = time.time()
    out = var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 40, 0.0, *st)
    abi.device_synchronize(); t2 = time.time()
    out = var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 100, 0.0, *out[:3])
    abi.device_synchronize(); t3 = time.time()
    R = var.Rt
    pv = var.primal_vio_raw
    e1 = np.max(np.abs(pv[:-1] - (np.einsum("ij,ij->i", R, R) - 1.0)))
    CR = data.C @ R
    e2 = abs(pv[-1] - float(np.sum(CR * R))) / abs(pv[-1])
    G = var.Gt
    Gref = 2 * (CR + var.y[:-1, None] * R)
    e3 = np.max(np.abs(G - Gref)) / np.max(np.abs(Gref))
    print(f"n={n} r={r} nnz={A.nnz} setup {t1-t0:.1f}s  {1e3*(t3-t2)/100:.3f} ms/iter  L {st[0]:.6e} -> {out[0]:.6e}  errs {e1:.2e} {e2:.2e} {e3:.2e}", flush=True)
    var.close()
