"""dev tool: timing of the resident route on a config-5 instance (Gset G1, n = 800, rank 10): µs per inner iteration of the
one-launch loop, µs per Lanczos step of the one-launch recurrence, one whole solve, and the same on the multi-launch
route (SDPLR_HIP_NO_RESIDENT=1 in a second process, see --route)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
A = problems.graph_from_edges(int(z["G1_n"]), z["G1"])
abi = sj.load_hip()
data = problems.maxcut_data(A)
normC, normb = data.normC(), float(np.linalg.norm(data.b))
out = {"route": "launches" if os.environ.get("SDPLR_HIP_NO_RESIDENT") else "resident"}
for r in (10, 20):
    cfg = sj.BurerMonteiroConfig(seed=0, printlevel=0)
    g = sj.build_solver(abi, data, r, cfg)
    st = g.fg(normC, normb)
    st = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 50, 0.0, *st)[:3]     # warm
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        res = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 400, 0.0, *st)
        ts.append(time.perf_counter() - t0)
        st = res[:3]
    out[f"loop_us_per_iter_r{r}"] = round(1e6 * min(ts) / 400, 2)
    v0 = np.random.Generator(np.random.PCG64(1)).standard_normal(data.n)
    g.dual_obj(800.0, 100, v0)
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        a, b, k = g.lanczos(134, v0)
        ts.append(time.perf_counter() - t0)
    out[f"lanczos_us_per_step_r{r}"] = round(1e6 * min(ts) / 134, 2)
    out[f"stats_r{r}"] = {k: v for k, v in g.stats().items() if v}
    g.close()
kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=800.0, printlevel=0)
sj.sdplr(data=data, r=10, **kw)
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    res = sj.sdplr(data=data, r=10, **kw)
    ts.append(time.perf_counter() - t0)
out["solve_ms"] = round(1e3 * min(ts), 2)
out["solve_iters"] = int(res["iter"])
out["solve_majoriter"] = int(res["majoriter"])
print(json.dumps(out))
