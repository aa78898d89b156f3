"""dev tool: randomized parity sweep HIP vs oracle through the native inner loop (families × sizes × ranks × history
lengths, including graphs with isolated vertices and ranks with ragged sub-wave shapes); prints the worst deviations"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import sdplrplus_jl_amd as sj
from oracle import oracle
from helpers import make_data, make_solver, FAMILIES_EQ
ALL_FAMILIES = list(FAMILIES_EQ) + ["ineq_0.01", "ineq_0.05", "ineq_0.1"]

hip, ora = sj.load_hip(), oracle.abi()
rng = np.random.Generator(np.random.PCG64(int(sys.argv[1]) if len(sys.argv) > 1 else 0))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
worst = []
for t in range(N):
    fam = ALL_FAMILIES[int(rng.integers(len(ALL_FAMILIES)))]
    n = int(rng.integers(6, 70)); p = float(rng.uniform(0.03, 0.5)); r = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 32, 33, 40, 64, 70, 130]))
    h = int(rng.choice([0, 1, 2, 3, 4, 4, 4, 5, 8])); iters = int(rng.integers(1, 14))
    try:
        data, C, As, bs = make_data(fam, int(rng.integers(1 << 30)), n, p)
    except Exception as e:   # degenerate random graph for this family
        continue
    g, _ = make_solver(hip, data, r, seed=t, h=h); o, _ = make_solver(ora, data, r, seed=t, h=h)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    arm = data.has_inequalities
    sg, so = g.fg(normC, normb), o.fg(normC, normb)
    try:
        rg = g.inner_loop(normC, normb, True, True, arm, 0.0, -1e300, iters, 0.0, *sg)
        ro = o.inner_loop(normC, normb, True, True, arm, 0.0, -1e300, iters, 0.0, *so)
    except Exception as e:
        print("EXC", fam, n, p, r, h, iters, e); g.close(); o.close(); continue
    scale = np.maximum(np.abs(np.array(sg)), np.abs(np.array(ro[:3]))) + 1e-300
    dev = float(np.max(np.abs(np.array(rg[:3]) - np.array(ro[:3])) / scale))
    dR = float(np.linalg.norm(g.Rt - o.Rt) / max(np.linalg.norm(o.Rt), 1e-300))
    ok = rg[4] == ro[4] and rg[5] == ro[5]
    worst.append((max(dev, dR), fam, n, round(p, 3), r, h, iters, rg[4], ok))
    g.close(); o.close()
worst.sort(reverse=True)
print("cases", len(worst), "mismatched iteration counts / exits:", sum(1 for w in worst if not w[-1]))
for w in worst[:6]:
    print("  %.2e  %s n=%d p=%s r=%d h=%d iters=%d ran=%d ok=%s" % w)
print("equality-constrained families only:")
for w in [w for w in worst if not w[1].startswith("ineq")][:6]:
    print("  %.2e  %s n=%d p=%s r=%d h=%d iters=%d ran=%d ok=%s" % w)
