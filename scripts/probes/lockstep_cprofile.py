import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, problems
abi = sj.load_hip(); abi.device_synchronize(); abi.warmup(64)
z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
graphs = [problems.graph_from_edges(int(z[f"G{k}_n"]), z[f"G{k}"]) for k in range(1, 10)]
graphs += [problems.gnp_graph(800, 0.06, seed) for seed in range(10, 65)]
datas = [problems.maxcut_data(g) for g in graphs]
kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=800.0, printlevel=0)
batch.solve_lockstep(datas, 10, **kw); batch.solve_lockstep(datas, 10, **kw)
pr = cProfile.Profile(); pr.enable()
batch.solve_lockstep(datas, 10, **kw)
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:5000])
