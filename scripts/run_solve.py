#!/usr/bin/env python3
"""Command-line runner mirroring the reference's exps/test.jl (ArgParse flags :9-38, warm-up solve then timed
solve :178-210, randomized rounding callbacks :71-105, JSON dump of the short result :136-161) on the MI355X
library.  Graphs: the Gset fixtures G1..G9 (tests/golden/gset_G1_G9.npz), or gnp:<n>:<p>:<seed>.

    python scripts/run_solve.py --graph G1 --problem MaxCut --rank 10 --ptol 0.01 --objtol 0.01 --seed 0
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdplrplus_jl_amd as sj  # noqa: E402
from sdplrplus_jl_amd import problems  # noqa: E402


def load_graph(name):
    if name.startswith("gnp:"):
        _, n, p, seed = name.split(":")
        return problems.gnp_graph(int(n), float(p), int(seed))
    z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
    return problems.graph_from_edges(int(z[f"{name}_n"]), z[name])


from sdplrplus_jl_amd.rounding import maxcut_rounding, minimumbisection_rounding  # noqa: E402  (exps/test.jl:67-105)


PROBLEMS = {  # name → (builder, rounding callback, trace bound as a function of n)   exps/test.jl:166-176
    "MaxCut": (problems.maxcut_data, maxcut_rounding, lambda n: n),
    "MinimumBisection": (problems.minimum_bisection_data, minimumbisection_rounding, lambda n: n),
    "LovaszTheta": (problems.lovasz_theta_data, None, lambda n: 1),
    "CutNorm": (lambda A: sj.SDPData(*problems.cutnorm(A)), None, lambda n: n),
}


def batch_eval(problem, graph, A, r, seed, out_dir, tag, **kw):
    builder, callback, tb = PROBLEMS[problem]
    data = builder(A)
    res = sj.sdplr(data=data, r=r, prior_trace_bound=float(tb(data.n)), dataset=graph, seed=seed, **kw)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    res["callback_res"] = callback(A, res["Rt"], rng) if callback else 0
    keys = ["grad_norm", "primal_vio", "obj", "max_dual_value", "min_duality_gap", "totaltime", "dual_time",
            "primaltime", "iter", "majoriter", "ptol", "objtol", "fprec", "callback_res", "rankupd_tol", "r"]
    short = {k: (float(res[k]) if isinstance(res[k], (np.floating, float)) else res[k]) for k in keys}
    if out_dir:
        path = os.path.join(out_dir, problem, graph)
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, tag + ".json"), "w") as f:
            json.dump(short, f, indent=4)
    return short


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="G1")
    ap.add_argument("--ptol", type=float, default=1e-2)
    ap.add_argument("--objtol", type=float, default=1e-2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--rank", type=int, default=10)
    ap.add_argument("--problem", default="MaxCut", choices=list(PROBLEMS))
    ap.add_argument("--output", default=None, help="folder for the JSON result (exps/test.jl writes exps/output/)")
    ap.add_argument("--printlevel", type=int, default=0)
    args = ap.parse_args()
    A1 = load_graph("G1")
    batch_eval(args.problem, "G1", A1, args.rank, args.seed, None, "SDPLR-warmup", maxtime=36000.0, objtol=1.0,
               ptol=1.0, printlevel=0)                                           # warm-up, exps/test.jl:180-192
    A = load_graph(args.graph)
    short = batch_eval(args.problem, args.graph, A, args.rank, args.seed, args.output,
                       f"SDPLR-R-{args.rank}-seed-{args.seed}-tol-{args.ptol}", maxtime=36000.0,
                       objtol=args.objtol, ptol=args.ptol, printlevel=args.printlevel)   # :198-210
    print(json.dumps(short))


if __name__ == "__main__":
    main()
