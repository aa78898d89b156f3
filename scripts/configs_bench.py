#!/usr/bin/env python3
"""Measurements for BASELINE.json configs[2..4] on one GPU (not the headline bench; numbers go to DESIGN.md)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems

abi = sj.load_hip()
out = {}


def fixed_iters(var, data, K, W=20):
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    st = var.fg(normC, normb)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        st = var.fg(normC, normb)
    run = lambda s, k: var.inner_loop(normC, normb, True, True, data.has_inequalities, 0.0, -1e300, k, 0.0, *s)[:3]
    st = run(st, W)
    abi.device_synchronize(); t0 = time.perf_counter()
    st = run(st, K)
    abi.device_synchronize(); dt = time.perf_counter() - t0
    return K / dt, st


# config 3: Lovász-θ stand-in
A = problems.chung_lu_graph(50_000, 10.0, 2.5, 3)
data = problems.lovasz_theta_data(A)
var = sj.build_solver(abi, data, 32, sj.BurerMonteiroConfig(seed=1, printlevel=0))
its, _ = fixed_iters(var, data, 200)
out["config3_lovasz_theta"] = {"n": data.n, "m": data.m, "r": 32, "inner_iterations_per_s": its}
var.close()

# config 4: MinBisection + Lanczos
A = problems.gnp_graph(100_000, 2e-4, 4)
data = problems.minimum_bisection_data(A)
var = sj.build_solver(abi, data, 32, sj.BurerMonteiroConfig(seed=2, printlevel=0))
its, _ = fixed_iters(var, data, 200)
v0 = np.random.Generator(np.random.PCG64(5)).standard_normal(data.n)
var.dual_obj(float(data.n), 0, v0)
q = 232
abi.device_synchronize(); t0 = time.perf_counter()
for _ in range(5):
    al, be, k = var.lanczos(q, v0)
abi.device_synchronize(); dt = (time.perf_counter() - t0) / 5
out["config4_minimum_bisection"] = {"n": data.n, "m": data.m, "r": 32, "inner_iterations_per_s": its,
                                    "lanczos_steps": int(k), "lanczos_steps_per_s": k / dt, "lanczos_ms": 1e3 * dt}
var.close()
print(json.dumps(out, indent=1))
