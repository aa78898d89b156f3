"""dev tool: how fast the HIP library's and the oracle's L-BFGS trajectories drift apart on the north-star instance
(relative differences of ℒ, ‖grad‖, ‖pv‖, obj and R after 5, 25 and 60 inner iterations from the same start)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
from oracle import oracle
from helpers import make_solver
hip, ora = sj.load_hip(), oracle.abi()
data = problems.maxcut_data(problems.gnp_graph(100_000, 2e-4, 20240610))
normC, normb = data.normC(), float(np.linalg.norm(data.b))
g,_ = make_solver(hip, data, 32, seed=0); o,_ = make_solver(ora, data, 32, seed=0)
sg, so = g.fg(normC, normb), o.fg(normC, normb)
tot = 0
for k in (5, 20, 35):
    rg = g.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *sg)
    ro = o.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *so)
    tot += k
    sg, so = rg[:3], ro[:3]
    R, Ro = g.Rt, o.Rt
    print(tot, "rel L %.2e gnorm %.2e pv %.2e obj %.2e  R %.2e" % (abs(rg[0]-ro[0])/abs(ro[0]), abs(rg[1]-ro[1])/abs(ro[1]), abs(rg[2]-ro[2])/abs(ro[2]), abs(g.obj-o.obj)/abs(o.obj), np.max(np.abs(R-Ro))/(1+np.max(np.abs(Ro)))), flush=True)
