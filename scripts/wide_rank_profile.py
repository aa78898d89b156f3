"""dev tool: per-kernel device time (event-timed, eager) of the inner iteration on MaxCut G(1e5, 2e-4) at a given rank,
next to each kernel's compulsory bytes (N = 8·n·r): where the wide ranks lose against r = 32.
    python scripts/wide_rank_profile.py 64 [128 …]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
abi = sj.load_hip()
data = problems.maxcut_data(problems.gnp_graph(100_000, 2e-4, 20240610))
normC, normb = data.normC(), float(np.linalg.norm(data.b))
for r in [int(a) for a in sys.argv[1:]] or [32, 64, 128]:
    var = sj.build_solver(abi, data, r, sj.BurerMonteiroConfig(seed=1, printlevel=0))
    d = var.dims()
    N = 8.0 * d["n"] * r
    nnzS = d["nnzS"]
    # compulsory bytes per launch: direction 10N, gather 3N + 12 nnzS + 4(n+1), step (P-less, update fused) 14N
    comp = {"lbfgs_dir": 10 * N, "spmm_W": 3 * N + 12 * nnzS + 4 * (d["n"] + 1), "fast_step": 14 * N}
    st = var.fg(normC, normb)
    run = lambda s, k: var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *s)[:3]
    st = run(st, 40)
    abi.device_synchronize()
    import time
    t0 = time.perf_counter(); st = run(st, 200); abi.device_synchronize(); t1 = time.perf_counter()
    var.profile_enable(True)
    st = run(st, 40)
    p = var.profile()
    print(f"r = {r}: {1e6 * (t1 - t0) / 200:.1f} us/iteration under graph replay; N = {N / 1e6:.1f} MB; iteration bytes (27N + 132 MB·r/32-ish) -> {(38 * N) / 1e6:.0f} MB moved by the three big kernels at best")
    tot = 0.0
    for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1]):
        us = 1e3 * ms / 40
        tot += us
        extra = f"  compulsory {comp[k] / 1e6:7.1f} MB -> {comp[k] / us / 1e6:5.2f} TB/s = {comp[k] / us / 8e6:.2f} of peak" if k in comp else ""
        print(f"   {k:18s} {us:8.1f} us/iter ({c / 40:.2f} launches){extra}")
    print(f"   sum {tot:.1f} us")
    var.close()
