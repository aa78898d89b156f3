import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
data = problems.maxcut_data(problems.gnp_graph(100_000, 2e-4, 20240610))
var = sj.build_solver(abi, data, 32, sj.BurerMonteiroConfig(seed=1, printlevel=0))
var.f()
v0 = np.random.Generator(np.random.PCG64(5)).standard_normal(data.n)
var.dual_obj(float(data.n), 0, v0)
for _ in range(3):
    t0 = time.perf_counter(); al, be, k = var.lanczos(232, v0); dt = time.perf_counter() - t0
    print(k, 1e6 * dt / k, "us/step")
