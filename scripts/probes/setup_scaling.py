"""Set-up of the 64 config-5 instances (build_solver: create, set_sparse_coo, finalize, first point) against the number of
host threads doing it, pools warm."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
from sdplrplus_jl_amd.sdplr import build_solver
abi = sj.load_hip(); abi.device_synchronize(); abi.warmup(64)
z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
graphs = [problems.graph_from_edges(int(z[f"G{k}_n"]), z[f"G{k}"]) for k in range(1, 10)]
graphs += [problems.gnp_graph(800, 0.06, seed) for seed in range(10, 65)]
datas = [problems.maxcut_data(g) for g in graphs]
cfg = sj.BurerMonteiroConfig(seed=0, printlevel=0)
def build_all(nt):
    if nt == 1:
        return [build_solver(abi, d, 10, cfg) for d in datas]
    with ThreadPoolExecutor(nt) as ex:
        return list(ex.map(lambda d: build_solver(abi, d, 10, cfg), datas))
for s in build_all(16): s.close()
for s in build_all(16): s.close()
for nt in (1, 2, 4, 8, 12, 16, 24):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); ss = build_all(nt); dt = time.perf_counter() - t0
        for s in ss: s.close()
        best = min(best, dt)
    print(f"threads {nt:2d}: {1e3 * best:7.2f} ms for 64 instances ({1e3 * best / 64:.3f} ms each)", flush=True)
