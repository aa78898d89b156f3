"""dev tool: steady-state throughput of the small-instance route — 256 MaxCut instances (G(800, 0.06), rank 10, tol 1e-2)
on one GPU: as independent driver threads against the number of instances in flight, and in lockstep (one launch per step
for a whole chunk) against the chunk size."""
import os, sys, time, json
# usage: batch_throughput.py threads|lockstep — one driver per process: the threaded one wants a hardware queue per instance
# in flight (GPU_MAX_HW_QUEUES, read when HIP starts), the lockstep one is 3–6× slower in such a process
MODE = sys.argv[1] if len(sys.argv) > 1 else "lockstep"
if MODE == "threads":
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
ref = None
for conc in ((16, 24) if MODE == "threads" else ()):
    abi.device_synchronize(); t0 = time.perf_counter()
    rows = batch.solve_local(datas, 0, 1, 10, concurrency=conc, **kw)
    dt = time.perf_counter() - t0
    ref = rows
    print(json.dumps({"driver": "threads", "in_flight": conc, "instances": 256, "wall_s": round(dt, 4),
                      "instances_per_s": round(256 / dt, 1), "inner_iterations_per_s": round(float(rows[:, 3].sum()) / dt)}), flush=True)
for chunk in ((64, 128, 256) if MODE == "lockstep" else ()):
    for rep in range(2):                      # (the first pass at a chunk size fills the pools for that many live handles)
        abi.device_synchronize(); t0 = time.perf_counter()
        parts = [batch.solve_local(datas[k:k + chunk], 0, 1, 10, concurrency=16, lockstep=True, **kw) for k in range(0, 256, chunk)]
        dt = time.perf_counter() - t0
    rows = np.concatenate(parts)
    print(json.dumps({"driver": "lockstep", "chunk": chunk, "instances": 256, "wall_s": round(dt, 4),
                      "instances_per_s": round(256 / dt, 1), "inner_iterations_per_s": round(float(rows[:, 3].sum()) / dt)}), flush=True)
