import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import sdplrplus_jl_amd as sj
from helpers import make_data, make_solver
from oracle import oracle
hip = sj.load_hip(); ora = oracle.abi()
data, C, As, bs = make_data("ineq_0.05", 5, 40, 0.3)
normC, normb = data.normC(), float(np.linalg.norm(data.b))
def mk(abi):
    s_, _ = make_solver(abi, data, 4, seed=21); return s_, s_.fg(normC, normb)
a, sa = mk(hip)
os.environ["SDPLR_HIP_NO_FAST"] = "1"
b, sb = mk(hip)
del os.environ["SDPLR_HIP_NO_FAST"]
o, so = mk(ora)
for it in range(25):
    ra = a.inner_loop(normC, normb, True, True, True, 0.0, -1e300, 1, 0.0, *sa)
    rb = b.inner_loop(normC, normb, True, True, True, 0.0, -1e300, 1, 0.0, *sb)
    ro = o.inner_loop(normC, normb, True, True, True, 0.0, -1e300, 1, 0.0, *so)
    sa, sb, so = ra[:3], rb[:3], ro[:3]
    print(it, "alpha fast/gen/oracle", ra[3], rb[3], ro[3], " L", ra[0], rb[0], ro[0])
