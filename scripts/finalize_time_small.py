import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems, cabi
z = np.load('/root/repo/tests/golden/gset_G1_G9.npz')
A = problems.graph_from_edges(int(z["G1_n"]), z["G1"])
abi = sj.load_hip(); abi.device_synchronize()
for rep in range(3):
    t0 = time.perf_counter(); data = problems.maxcut_data(A); t1 = time.perf_counter()
    lay = sj.preprocess_sparsecons(data.sparse); t2 = time.perf_counter()
    s = cabi.DeviceSolver(abi, data.n, data.m, 10, 4); s.set_sparse(lay); t3 = time.perf_counter()
    if rep == 2: os.environ["SDPLR_HIP_TIMING"] = "1"
    s.finalize(); t4 = time.perf_counter()
    s.close(); t5 = time.perf_counter()
    print(f"maxcut_data {1e3*(t1-t0):.2f}  preprocess {1e3*(t2-t1):.2f}  create+set_sparse {1e3*(t3-t2):.2f}  finalize {1e3*(t4-t3):.2f}  close {1e3*(t5-t4):.2f} ms")
