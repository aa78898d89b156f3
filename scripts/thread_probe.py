import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
nth = int(sys.argv[1])
def one(k):
    data = problems.maxcut_data(problems.gnp_graph(800, 0.06, 10 + k))
    var = sj.build_solver(abi, data, 10, sj.BurerMonteiroConfig(seed=0, printlevel=0))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    st = var.fg(normC, normb)
    out = var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 50, 0.0, *st)
    var.close()
    return out[0]
t0 = time.perf_counter()
with ThreadPoolExecutor(max_workers=nth) as ex:
    res = list(ex.map(one, range(16)))
print(nth, "threads ok", time.perf_counter() - t0, res[:3])
