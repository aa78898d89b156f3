"""Where a lockstep batch spends its wall time: set-up, each round's library calls, the steppers' own Python."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, cabi, problems
from sdplrplus_jl_amd.structs import BurerMonteiroConfig
from sdplrplus_jl_amd.sdplr import build_solver, sdplr_steps

abi = sj.load_hip()
abi.device_synchronize(); abi.warmup(16)
z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
graphs = [problems.graph_from_edges(int(z[f"G{k}_n"]), z[f"G{k}"]) for k in range(1, 10)]
graphs += [problems.gnp_graph(800, 0.06, seed) for seed in range(10, 65)]
datas = [problems.maxcut_data(g) for g in graphs]
kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=800.0, printlevel=0)
if os.environ.get("PROBE_FAMILY") == "lovasz":      # Lovász-θ on G1–G9 (multi-launch edge path; shared launches per group)
    datas = [problems.lovasz_theta_data(g) for g in graphs[:9]]
    kw["prior_trace_bound"] = 1.0
batch.solve_lockstep(datas[:4], 10, **kw)
for rep in range(1 if os.environ.get('PROBE_FAMILY') else 2):
    laps = {"setup": 0.0, "calls": 0.0, "python": 0.0}
    t00 = time.perf_counter()
    cfgs = []
    for _ in datas:
        c = BurerMonteiroConfig()
        for k, v in kw.items():
            setattr(c, k, v)
        cfgs.append(c)
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    with ThreadPoolExecutor(16) as ex:
        solvers = list(ex.map(lambda dc: build_solver(abi, dc[0], 10, dc[1]), zip(datas, cfgs)))
    laps["setup"] = time.perf_counter() - t0
    steppers = [sdplr_steps(d, v, c) for d, v, c in zip(datas, solvers, cfgs)]
    pending = {}
    t0 = time.perf_counter()
    for k, g in enumerate(steppers):
        pending[k] = next(g)
    laps["python"] += time.perf_counter() - t0
    rounds = []
    results = {}
    cap = batch._Cap(int(os.environ.get("SDPLR_LOCKSTEP_CAP", "0")))
    while pending:
        ks = sorted(pending)
        t0 = time.perf_counter()
        reqs = [cap.outgoing(k, pending[k]) for k in ks]
        resp = batch.serve_batch(abi, [solvers[k] for k in ks], reqs)
        t1 = time.perf_counter()
        kinds = {}
        for k, q in zip(ks, reqs):
            name = q[0] + ("+" if q[0] == "major_iteration" and q[6] == cabi.MAJOR_RESUME else "")
            kinds[name] = kinds.get(name, 0) + 1
        for k, r in zip(ks, resp):
            r = cap.incoming(k, r)
            if r is None:
                continue
            try:
                pending[k] = steppers[k].send(r)
            except StopIteration as done:
                results[k] = done.value
                del pending[k]
        t2 = time.perf_counter()
        laps["calls"] += t1 - t0
        laps["python"] += t2 - t1
        its = [r[4] for q, r in zip(reqs, resp) if q[0] == "major_iteration" and not isinstance(r, Exception)]
        rounds.append((kinds, round(1e3 * (t1 - t0), 2), round(1e3 * (t2 - t1), 2), (max(its), sum(its)) if its else ()))
    t0 = time.perf_counter()
    for v in solvers:
        v.close()
    laps["close"] = time.perf_counter() - t0
    print("total", round(time.perf_counter() - t00, 4), {k: round(v, 4) for k, v in laps.items()})
    for r in rounds:
        print("   ", r)
