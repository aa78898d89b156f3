import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems, cabi
abi = sj.load_hip()
nth = int(sys.argv[1])
datas = [problems.maxcut_data(problems.gnp_graph(800, 0.06, 10 + k)) for k in range(48)]
def one(k):
    res = sj.sdplr(data=datas[k], r=10, printlevel=0, ptol=0.01, objtol=0.01, seed=0, prior_trace_bound=800.0, maxmajoriter=40)
    return (res["obj"], res["iter"], res["majoriter"], res["primal_vio"], res["sigma"], float(np.abs(res["Rt"]).max()), float(np.abs(res["Rt0"]).max()))
ref = [one(k) for k in range(48)]
with ThreadPoolExecutor(max_workers=nth) as ex:
    res = list(ex.map(one, range(48)))
nbad = 0
for k, (a, b) in enumerate(zip(ref, res)):
    if abs(a[0] - b[0]) > 1e-3 * abs(a[0]) or a[2] != b[2]:
        nbad += 1
        print("MISMATCH", k, a, b)
print("threads", nth, "mismatches", nbad)
