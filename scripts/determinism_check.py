import sys
sys.path.insert(0, "/root/repo")
import numpy as np, hashlib
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
cases = {
 "maxcut": problems.maxcut_data(problems.gnp_graph(20000, 1e-3, 3)),
 "minbis": problems.minimum_bisection_data(problems.gnp_graph(20000, 1e-3, 4)),
 "lovasz": problems.lovasz_theta_data(problems.chung_lu_graph(8000, 10.0, 2.5, 3)),
}
mu = problems.mu_conductance(problems.gnp_graph(6000, 3e-3, 5), 0.05)
cases["mucond"] = sj.SDPData(*mu)
for name, data in cases.items():
    hs = []
    for rep in range(3):
        var = sj.build_solver(abi, data, 16, sj.BurerMonteiroConfig(seed=1, printlevel=0))
        normC, normb = data.normC(), float(np.linalg.norm(data.b))
        st = var.fg(normC, normb)
        out = var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 60, 0.0, *st)
        hs.append(hashlib.sha1(var.Rt.tobytes() + var.Gt.tobytes()).hexdigest()[:12] + f" L={out[0]:.15e}")
        var.close()
    print(name, "deterministic" if len(set(hs)) == 1 else "NON-DETERMINISTIC", hs[0] if len(set(hs)) == 1 else hs)
