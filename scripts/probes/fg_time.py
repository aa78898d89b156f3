"""dev tool: device time of fg! at the north-star size (event-timed kernels of 20 calls) on the structured route and, with
SDPLR_HIP_NO_FAST_FG=1, on the generic one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
data = problems.maxcut_data(problems.gnp_graph(100_000, 2e-4, 20240610))
var = sj.build_solver(abi, data, 32, sj.BurerMonteiroConfig(seed=0, printlevel=0))
normC, normb = data.normC(), float(np.linalg.norm(data.b))
for _ in range(20): st = var.fg(normC, normb)
abi.device_synchronize(); t0 = time.perf_counter()
for _ in range(50): st = var.fg(normC, normb)
abi.device_synchronize(); t1 = time.perf_counter()
print(f"fg! {1e6 * (t1 - t0) / 50:.1f} us per call (host-paired, incl. the scalar read-back)", st)
var.profile_enable(True)
for _ in range(20): var.fg(normC, normb)
p = var.profile()
tot = 0.0
for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k:18s} {1e3 * ms / 20:7.1f} us/call ({c / 20:.2f} launches)"); tot += 1e3 * ms / 20
print(f"   sum {tot:.1f} us")
