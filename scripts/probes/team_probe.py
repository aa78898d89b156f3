"""dev probe: one Gset G1 MaxCut solve (rank 10) and the 64-instance lockstep batch, with SDPLR_HIP_TEAM workgroups per instance."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, problems
abi = sj.load_hip(); abi.device_synchronize(); abi.warmup(64)
z = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "gset_G1_G9.npz"))
graphs = [problems.graph_from_edges(int(z[f"G{k}_n"]), z[f"G{k}"]) for k in range(1, 10)]
graphs += [problems.gnp_graph(800, 0.06, seed) for seed in range(10, 65)]
datas = [problems.maxcut_data(g) for g in graphs]
kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=800.0, printlevel=0)
# inner-loop rate of one instance
v = sj.build_solver(abi, datas[0], 10, sj.BurerMonteiroConfig(seed=0, printlevel=0))
nC, nB = datas[0].normC(), float(np.linalg.norm(datas[0].b))
st = v.fg(nC, nB)
st = v.inner_loop(nC, nB, True, True, False, 0.0, -1e300, 50, 0.0, *st)[:3]
abi.device_synchronize(); t0 = time.perf_counter()
res = v.inner_loop(nC, nB, True, True, False, 0.0, -1e300, 400, 0.0, *st)
abi.device_synchronize(); dt = time.perf_counter() - t0
print(f"TEAM={os.environ.get('SDPLR_HIP_TEAM', '1')}: {1e6 * dt / 400:.1f} us per inner iteration; L {res[0]!r} gn {res[1]!r}")
v.close()
r1 = sj.sdplr(data=datas[0], r=10, **kw)
print("G1 solve: iters", r1["iter"], "obj", r1["obj"], "dual", r1["max_dual_value"])
batch.solve_lockstep(datas[:8], 10, **kw)
for rep in range(3):
    t0 = time.perf_counter()
    out = batch.solve_lockstep(datas, 10, **kw)
    w = time.perf_counter() - t0
    bad = [o for o in out if isinstance(o, Exception)]
    print(f"batch of 64: {w:.4f} s, iterations {sum(o['iter'] for o in out if not isinstance(o, Exception))}, errors {len(bad)} {bad[:1]}")
