#!/usr/bin/env python3
"""The reference's exps/gen_batch_test.jl for this build: writes the batch file — one solve per line, exactly the
flags of exps/batch_test.txt:1-9 (`--seed --graph --problem --ptol --objtol --rank`) on scripts/run_solve.py — and
the JSON manifest scripts/run_batch.py shards over the GPUs.

    python scripts/gen_batch_test.py                               # G1..G9, MaxCut, rank 10, tol 0.01 (the reference's batch)
    python scripts/gen_batch_test.py --extra-gnp 55                # + 55 seeded G(800, 0.06): BASELINE.json configs[4]
    python scripts/gen_batch_test.py --graphs G1 G2 gnp:2000:0.01:7 --problem MinimumBisection --rank 16 --tol 1e-3
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBLEMS = ["MaxCut", "MinimumBisection", "LovaszTheta", "CutNorm"]       # exps/gen_batch_test.jl:3


def build(graphs, problem, seed, tol, rank):
    """→ (lines of the batch file, manifest entries)"""
    lines, entries = [], []
    for g in graphs:
        lines.append(f"ulimit -d {16 * 1024 * 1024}; cd {ROOT}; {sys.executable} scripts/run_solve.py --seed {seed} "
                     f'--graph "{g}" --problem "{problem}" --ptol {tol} --objtol {tol} --rank {rank}')
        entries.append({"graph": g, "problem": problem, "seed": seed, "ptol": tol, "objtol": tol, "rank": rank})
    return lines, entries


def default_graphs(extra_gnp=0):
    graphs = [f"G{i}" for i in range(1, 10)]                               # exps/gen_batch_test.jl:1
    graphs += [f"gnp:800:0.06:{s}" for s in range(10, 10 + extra_gnp)]
    return graphs


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", nargs="*", default=None)
    ap.add_argument("--extra-gnp", type=int, default=0, help="append this many seeded G(800, 0.06) graphs")
    ap.add_argument("--problem", default="MaxCut", choices=PROBLEMS)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tol", type=float, default=0.01)
    ap.add_argument("--rank", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "scripts", "batch_test.txt"))
    args = ap.parse_args(argv)
    graphs = args.graphs if args.graphs else default_graphs(args.extra_gnp)
    lines, entries = build(graphs, args.problem, args.seed, args.tol, args.rank)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.splitext(args.out)[0] + ".json", "w") as f:
        json.dump(entries, f, indent=1)
    print(f"{len(lines)} solves → {args.out} (+ .json manifest)")


if __name__ == "__main__":
    main()
