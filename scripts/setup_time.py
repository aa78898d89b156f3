"""dev tool: where the set-up time of one solve goes (north-star instance)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems, cabi
from sdplrplus_jl_amd.preprocess import preprocess_sparsecons
abi = sj.load_hip()
t = [time.perf_counter()]
def lap(name):
    t.append(time.perf_counter()); print(f"{name:28s} {1e3*(t[-1]-t[-2]):8.1f} ms", flush=True)
A = problems.gnp_graph(100_000, 2e-4, 20240610); lap("gnp_graph")
data = problems.maxcut_data(A); lap("maxcut_data")
abi.device_synchronize(); lap("HIP context (first call)")
cfg = sj.BurerMonteiroConfig(seed=0, printlevel=0)
if os.environ.get("SETUP_HOST_PREPROCESS"):
    lay = preprocess_sparsecons(data.sparse); lap("preprocess_sparsecons (host mirror)")
    s = sj.DeviceSolver(abi, data.n, data.m, 32, 4); lap("create")
    s.set_sparse(lay); lap("set_sparse")
else:
    s = sj.DeviceSolver(abi, data.n, data.m, 32, 4); lap("create")
    s.set_sparse_coo(data.sparse); lap("set_sparse_coo (native preprocess)")
os.environ["SDPLR_HIP_TIMING"] = "1"
s.finalize(); lap("finalize")
del os.environ["SDPLR_HIP_TIMING"]
Rt0, l0, lub = sj.initial_point(data, 32, cfg); lap("initial_point")
s.set_vec(cabi.V_B, data.b); s.set_vec(cabi.V_LAMBDA_UB, lub); s.set_vec(cabi.V_LAMBDA, l0); s.set_factor(cabi.F_RT, Rt0); lap("uploads")
normC, normb = data.normC(), float(np.linalg.norm(data.b)); lap("norms")
st = s.fg(normC, normb); lap("first fg!")
out = s.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 8, 0.0, *st); lap("first inner_loop (capture)")
out = s.inner_loop(normC, normb, True, True, False, 0.0, -1e300, 8, 0.0, *out[:3]); lap("second inner_loop")
v0 = np.ones(data.n); s.dual_obj(1e5, 0, v0); lap("first dual_obj (capture)")
s.dual_obj(1e5, 0, v0); lap("second dual_obj")
