"""Where the set-up of ONE config-5 instance goes (one thread, pools warm): every library call of build_solver timed."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems, cabi
import importlib
drv = importlib.import_module("sdplrplus_jl_amd.sdplr")
abi = sj.load_hip(); abi.device_synchronize(); abi.warmup(64)
z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
graphs = [problems.graph_from_edges(int(z[f"G{k}_n"]), z[f"G{k}"]) for k in range(1, 10)]
graphs += [problems.gnp_graph(800, 0.06, seed) for seed in range(10, 65)]
datas = [problems.maxcut_data(g) for g in graphs]
cfg = sj.BurerMonteiroConfig(seed=0, printlevel=0)
laps = collections.defaultdict(float)
def timed(cls, name):
    f = getattr(cls, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            laps[name] += time.perf_counter() - t0
    setattr(cls, name, w)
for nm in ("set_sparse_coo", "finalize", "set_vec", "set_factor", "set_scalar", "close"):
    timed(cabi.DeviceSolver, nm)
_ip = drv.initial_point
def ip(*a, **k):
    t0 = time.perf_counter()
    try:
        return _ip(*a, **k)
    finally:
        laps["initial_point (numpy)"] += time.perf_counter() - t0
drv.initial_point = ip
_init = cabi.DeviceSolver.__init__
def init(self, *a, **k):
    t0 = time.perf_counter(); _init(self, *a, **k); laps["create"] += time.perf_counter() - t0
cabi.DeviceSolver.__init__ = init
for rep in range(3):
    laps.clear()
    t0 = time.perf_counter()
    ss = [drv.build_solver(abi, d, 10, cfg) for d in datas]
    tot = time.perf_counter() - t0
    for s in ss: s.close()
    print(f"rep {rep}: {1e3 * tot:.1f} ms for 64 | per instance (µs): " + ", ".join(f"{k} {1e6 * v / 64:.0f}" for k, v in sorted(laps.items(), key=lambda kv: -kv[1])), flush=True)
