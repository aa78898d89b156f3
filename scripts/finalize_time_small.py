"""Set-up and tear-down of one config-5 instance (Gset G1, rank 10), lap by lap, single thread, warm pools:
`SDPLR_HIP_TIMING=1` in the last repetition prints finalize's own laps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems, cabi
from sdplrplus_jl_amd.sdplr import _load_point
z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
A = problems.graph_from_edges(int(z["G1_n"]), z["G1"])
abi = sj.load_hip(); abi.device_synchronize()
cfg = sj.BurerMonteiroConfig()
data = problems.maxcut_data(A)
N = 20
for rep in range(4):
    laps = np.zeros(5)
    for _ in range(N):
        t0 = time.perf_counter()
        s = cabi.DeviceSolver(abi, data.n, data.m, 10, 4); t1 = time.perf_counter()
        s.set_sparse_coo(data.sparse); t2 = time.perf_counter()
        if rep == 3 and _ == N - 1: os.environ["SDPLR_HIP_TIMING"] = "1"
        s.finalize(); t3 = time.perf_counter()
        os.environ.pop("SDPLR_HIP_TIMING", None)
        _load_point(s, data, 10, cfg); t4 = time.perf_counter()
        s.close(); t5 = time.perf_counter()
        laps += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
    laps *= 1e3 / N
    print(f"create {laps[0]:.3f}  set_sparse_coo {laps[1]:.3f}  finalize {laps[2]:.3f}  load_point {laps[3]:.3f}  close {laps[4]:.3f}  total {laps.sum():.3f} ms")
