"""Per-round wall times of a lockstep batch over the throughput set (seeds 1000+), and the iteration spread."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, cabi, problems
abi = sj.load_hip(); abi.device_synchronize(); abi.warmup(64)
datas = [problems.maxcut_data(problems.gnp_graph(800, 0.06, 1000 + k)) for k in range(64)]
kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=800.0, printlevel=0)
real = batch.serve_batch
log = []
def timed(abi_, solvers, reqs):
    t0 = time.perf_counter()
    out = real(abi_, solvers, reqs)
    kinds = {}
    for q in reqs:
        kinds[q[0]] = kinds.get(q[0], 0) + 1
    its = [o[4] for o, q in zip(out, reqs) if q[0] in ("major_iteration", "inner_loop") and not isinstance(o, Exception)]
    log.append((kinds, round(1e3 * (time.perf_counter() - t0), 2), (max(its), int(np.median(its))) if its else None))
    return out
batch.serve_batch = timed
for rep in range(2):
    log.clear()
    t0 = time.perf_counter()
    res = batch.solve_lockstep(datas, 10, **kw)
    print("wall", round(time.perf_counter() - t0, 4), "iters", sorted(int(x["iter"]) for x in res)[-8:], "majors", sorted(int(x["majoriter"]) for x in res)[-5:])
for l in log:
    print("  ", l)
