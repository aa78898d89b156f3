"""dev tool: time of the W = A_g·D kernel under another build of the library (SDPLR_HIP_LIBRARY=…, e.g. one compiled
with -DSDPLR_TILE_WIN=8, or with parts of the kernel stubbed out for an ablation — results may then be garbage)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, sdplrplus_jl_amd as sj
abi = sj.load_hip()
data, var = bench.build_instance(abi, bench.GRAPH_SEED)
normC, normb = data.normC(), float(np.linalg.norm(data.b))
st = var.fg(normC, normb)
var.profile_enable(True)
for it in range(6):
    try:
        var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 1, 0.0, *st)
    except Exception as e:
        print("stopped:", e)
p = var.profile()
print(os.environ.get("SDPLR_HIP_LIBRARY", "default"), {k: round(1e3 * v[1] / v[0], 1) for k, v in p.items() if k in ("spmm_W", "fast_step")}, p.get("spmm_W"))
