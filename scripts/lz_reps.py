import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
A = problems.gnp_graph(100_000, 2e-4, 4)
data = problems.maxcut_data(A)
var = sj.build_solver(abi, data, 32, sj.BurerMonteiroConfig(seed=2, printlevel=0))
normC, normb = data.normC(), float(np.linalg.norm(data.b))
var.fg(normC, normb)
v0 = np.random.Generator(np.random.PCG64(5)).standard_normal(data.n)
var.dual_obj(float(data.n), 0, v0)
for reps in (1, 2, 4, 8, 16):
    os.environ["SDPLR_HIP_LZ_REPS"] = str(reps)
    var.lanczos(232, v0)
    abi.device_synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        al, be, k = var.lanczos(232, v0)
    abi.device_synchronize(); dt = (time.perf_counter() - t0) / 5
    print(reps, k, f"{1e3*dt:.2f} ms  {1e6*dt/k:.1f} us/step", flush=True)
