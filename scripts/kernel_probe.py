"""dev tool: device time of the individual dense kernels at the north-star size"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, sdplrplus_jl_amd as sj
abi = sj.load_hip()
data, var = bench.build_instance(abi, bench.GRAPH_SEED)
normC, normb = data.normC(), float(np.linalg.norm(data.b))
st = var.fg(normC, normb)
st = bench.run_fixed(var, normC, normb, st, 30)
var.profile_enable(True)
for _ in range(30):
    var.lbfgs_dir(True)
    var.lbfgs_update(0.5)
p = var.profile()
for k in ("lbfgs_dir", "lbfgs_update", "lbfgs_boundary", "descent"):
    print(f"{k:16s} {1e3*p[k][1]/p[k][0]:8.2f} us  x{p[k][0]}")
