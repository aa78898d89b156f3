import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
for k in (22, 26, 61, 62, 23):
    A = problems.gnp_graph(800, 0.06, 10 + (k - 9))
    data = problems.maxcut_data(A)
    t = time.time()
    res = sj.sdplr(data=data, r=10, printlevel=int(os.environ.get("PL", "0")), ptol=0.01, objtol=0.01, seed=0, prior_trace_bound=800.0, maxmajoriter=30)
    print(k, res["obj"], res["max_dual_value"], res["iter"], res["majoriter"], res["primal_vio"], res["r"], time.time() - t, flush=True)
