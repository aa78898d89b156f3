"""dev tool: randomized sweep of the lockstep driver — groups of random instances (families × sizes; one rank and tolerance
per group) solved side by side (batch.solve_lockstep) and one by one (sdplr): iterations, objective, dual bound, R and λ must
be identical bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch
from helpers import make_data
FAMS = ["maxcut", "maxcut", "lovasz_theta", "lovasz_theta", "lovasz_theta", "minimum_bisection", "cutnorm", "ineq_0.05", "mu_conductance_0.1"]
rng = np.random.Generator(np.random.PCG64(int(sys.argv[1]) if len(sys.argv) > 1 else 0))
G = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = total = 0
for g in range(G):
    r = int(rng.choice([2, 3, 5, 8, 10])); tol = float(rng.choice([1e-2, 1e-3])); B = int(rng.integers(2, 12))
    datas = []
    while len(datas) < B:
        fam = FAMS[int(rng.integers(len(FAMS)))]
        try:
            datas.append(make_data(fam, int(rng.integers(1 << 30)), int(rng.integers(8, 120)), float(rng.uniform(0.05, 0.4)))[0])
        except Exception:
            pass
    kw = dict(printlevel=0, ptol=tol, objtol=tol, seed=g, prior_trace_bound=float(max(d.n for d in datas)), maxmajoriter=25,
              maxiter=5000, maxtime=1e9)
    one = [sj.sdplr(data=d, r=r, **kw) for d in datas]
    many = batch.solve_lockstep(datas, r, **kw)
    for k, (a, b) in enumerate(zip(one, many)):
        total += 1
        same = (not isinstance(b, Exception) and a["iter"] == b["iter"] and a["obj"] == b["obj"]
                and a["max_dual_value"] == b["max_dual_value"] and np.array_equal(a["Rt"], b["Rt"])
                and np.array_equal(a["lambda"], b["lambda"]) and a["schedule"] == b["schedule"])
        if not same:
            bad += 1
            print("DIFF group", g, "instance", k, "n", datas[k].n, "r", r, repr(b)[:120] if isinstance(b, Exception) else (a["iter"], b["iter"], a["obj"], b["obj"]))
print(f"groups {G} instances {total} differing {bad}")
sys.exit(1 if bad else 0)
