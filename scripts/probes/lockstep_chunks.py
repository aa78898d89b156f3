import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, problems
abi = sj.load_hip(); abi.device_synchronize(); abi.warmup(64)
datas = [problems.maxcut_data(problems.gnp_graph(800, 0.06, 1000 + k)) for k in range(256)]
kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=800.0)
print("queues", os.environ.get("GPU_MAX_HW_QUEUES"), "threads_first", len(sys.argv) > 1)
if len(sys.argv) > 1:
    batch.solve_local(datas, 0, 1, 10, concurrency=16, **kw)
for rep in range(3):
    for k in range(0, 256, 64):
        t0 = time.perf_counter()
        batch.solve_local(datas[k:k + 64], 0, 1, 10, concurrency=16, lockstep=True, **kw)
        print(rep, k, round(time.perf_counter() - t0, 4), flush=True)
