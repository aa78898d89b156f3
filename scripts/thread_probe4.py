import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems, cabi
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import run_batch
abi = sj.load_hip()
graphs = run_batch.instances()
def one(k):
    data = problems.maxcut_data(graphs[k])
    res = sj.sdplr(data=data, r=10, printlevel=0, ptol=0.01, objtol=0.01, seed=0, prior_trace_bound=800.0, maxmajoriter=40)
    return (k, res["obj"], res["iter"], res["majoriter"], res["primal_vio"], res["sigma"], float(np.abs(res["Rt"]).max()), float(np.abs(res["Rt0"]).max()), res["grad_norm"])
for rep in range(3):
    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(one, range(64)))
    bad = [r for r in res if not (r[1] < -1000)]
    print("rep", rep, "bad:", bad, flush=True)
