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
    data = datas[k]
    var = sj.build_solver(abi, data, 10, sj.BurerMonteiroConfig(seed=0, printlevel=0))
    bad = []
    if not np.array_equal(var.Rt, var.Rt0): bad.append("Rt upload")
    if not np.array_equal(var.b, data.b): bad.append("b upload")
    if not np.array_equal(var.λ_ub, np.full(data.m, np.inf)): bad.append("lub upload")
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    st = var.fg(normC, normb)
    out = var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 60, 0.0, *st)
    var.close()
    return (st[0], out[0], bad)
ref = [one(k) for k in range(48)]
with ThreadPoolExecutor(max_workers=nth) as ex:
    res = list(ex.map(one, range(48)))
nbad = 0
for k, (a, b) in enumerate(zip(ref, res)):
    if a[0] != b[0] or abs(a[1] - b[1]) > 1e-6 * abs(a[1]) or b[2]:
        nbad += 1
        print("MISMATCH", k, a, b)
print("threads", nth, "mismatches", nbad)
