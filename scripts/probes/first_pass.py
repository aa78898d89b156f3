"""dev probe: what the FIRST lockstep batch of a process costs over the later ones — set-up (threads) and rounds apart."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, problems
from sdplrplus_jl_amd.sdplr import build_solver
abi = sj.load_hip(); abi.device_synchronize(); abi.warmup(64)
graphs = [problems.gnp_graph(800, 0.06, seed) for seed in range(64)]
datas = [problems.maxcut_data(g) for g in graphs]
kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=800.0, printlevel=0)
sj.sdplr(data=problems.maxcut_data(problems.gnp_graph(300, 0.1, 1)), r=6, **kw)
cfg = sj.BurerMonteiroConfig(seed=0, printlevel=0)
for rep in range(2):
    t0 = time.perf_counter()
    with ThreadPoolExecutor(8) as ex:
        ss = list(ex.map(lambda d: build_solver(abi, d, 10, cfg), datas))
    t1 = time.perf_counter()
    for s in ss: s.close()
    print(f"set-up alone, pass {rep}: {1e3 * (t1 - t0):.1f} ms", flush=True)
for rep in range(3):
    t0 = time.perf_counter(); out = batch.solve_lockstep(datas, 10, **kw); print(f"whole batch, pass {rep}: {time.perf_counter() - t0:.4f} s", flush=True)
