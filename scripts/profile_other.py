import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
A = problems.gnp_graph(100_000, 2e-4, 5)
for name, (C, As, bs) in (("mu_conductance", problems.mu_conductance(A, 0.05)), ("cutnorm", problems.cutnorm(problems.gnp_graph(50_000, 4e-4, 6)))):
    data = sj.SDPData(C, As, bs)
    var = sj.build_solver(abi, data, 32, sj.BurerMonteiroConfig(seed=1, printlevel=0))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    st = var.fg(normC, normb)
    run = lambda s, k: var.inner_loop(normC, normb, True, True, data.has_inequalities, 0.0, -1e300, k, 0.0, *s)[:3]
    st = run(st, 60)
    import time
    abi.device_synchronize(); t0 = time.perf_counter(); st = run(st, 200); abi.device_synchronize(); dt = time.perf_counter() - t0
    var.profile_enable(True)
    st = run(st, 40)
    p = var.profile()
    print(name, "dims", var.dims(), f"{1e6*dt/200:.1f} us/iter (graph)")
    for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"   {k:20s} {1e3*ms/40:8.1f} us/iter  ({c/40:.2f} launches/iter)")
    var.close()
