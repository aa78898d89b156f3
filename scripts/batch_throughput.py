"""dev tool: steady-state throughput of the small-instance route — 256 MaxCut instances (G(800, 0.06), rank 10, tol 1e-2)
on one GPU against the number of instances in flight."""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, problems
abi = sj.load_hip(); abi.device_synchronize()
datas = [problems.maxcut_data(problems.gnp_graph(800, 0.06, 1000 + k)) for k in range(256)]
kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=800.0)
abi.warmup(32)
batch.solve_local(datas[:4], 0, 1, 10, concurrency=4, **kw)
for conc in (8, 16, 24, 32):
    abi.device_synchronize(); t0 = time.perf_counter()
    rows = batch.solve_local(datas, 0, 1, 10, concurrency=conc, **kw)
    dt = time.perf_counter() - t0
    print(json.dumps({"in_flight": conc, "instances": 256, "wall_s": round(dt, 4), "instances_per_s": round(256 / dt, 1),
                      "inner_iterations_per_s": round(float(rows[:, 3].sum()) / dt)}), flush=True)
