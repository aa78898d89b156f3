"""Whole solves (sdplr: σ/λ schedule, dual bounds, rank updates) of MaxCut / MinBisection at n = 1e5 with the L-BFGS history in
ring form inside the loops and with the stored form (SDPLR_HIP_NO_RING=1): same schedule, objective and dual bound; wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
which = sys.argv[1] if len(sys.argv) > 1 else "maxcut"
A = problems.gnp_graph(100_000, 2e-4, 20240610 if which == "maxcut" else 4)
data = problems.maxcut_data(A) if which == "maxcut" else problems.minimum_bisection_data(A)
kw = dict(r=32, ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=float(data.n), printlevel=0)
sj.sdplr(data=problems.maxcut_data(problems.gnp_graph(2000, 0.01, 1)), **dict(kw, prior_trace_bound=2000.0, r=8))   # contexts, pools
for ring in (True, False, True, False):
    if ring:
        os.environ.pop("SDPLR_HIP_NO_RING", None)
    else:
        os.environ["SDPLR_HIP_NO_RING"] = "1"
    t0 = time.perf_counter()
    res = sj.sdplr(data=data, **kw)
    dt = time.perf_counter() - t0
    print(f"{which} ring={int(ring)}: {dt:.3f} s  iter {res['iter']} majoriter {res['majoriter']} obj {res['obj']:.10e} dual {res['max_dual_value']:.10e} "
          f"primaltime {res.get('primaltime', 0):.3f} dual_time {res.get('dual_time', 0):.3f} r {res['r']}", flush=True)
