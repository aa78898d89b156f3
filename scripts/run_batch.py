#!/usr/bin/env python3
"""BASELINE.json configs[4]: a batch of MaxCut instances — by default Gset G1–G9 (the reference's batch,
exps/batch_test.txt: rank 10, ptol = objtol = 0.01) + 55 seeded G(800, 0.06) graphs, or any manifest written by
scripts/gen_batch_test.py (--batch file.json) — sharded round-robin over the ranks, 8 in flight per GPU,
objectives gathered with one RCCL all_gather.  This is the reference's `parallel --jobs 9 < batch_test.txt`
(exps/README.md:17-21) with GPUs in place of CPU jobs.

    python scripts/run_batch.py [--batch scripts/batch_test.json]   # 1 GPU
    python scripts/run_batch.py --gpus 8                            # launches its own 8 ranks (torch.distributed.run)
"""
import json
import os
import sys
import time

# The threaded driver (SDPLR_BATCH_MODE=threads) wants one hardware queue per instance in flight: ROCm multiplexes a
# process's streams onto 4 hardware queues by default, and a resident loop (one launch that runs for milliseconds) holds its
# queue while it runs.  The lockstep driver (default) carries a whole batch on one stream and is 3–6× SLOWER in a process
# with 16–32 hardware queues (scripts/probes/lockstep_chunks.py): it keeps the default.  Must be set before HIP starts.
if os.environ.get("SDPLR_BATCH_MODE", "lockstep") != "lockstep":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdplrplus_jl_amd import problems  # noqa: E402  (no GPU work at import: the launcher branch runs first)


def load_graph(name):
    if name.startswith("gnp:"):
        _, n, p, seed = name.split(":")
        return problems.gnp_graph(int(n), float(p), int(seed))
    z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
    return problems.graph_from_edges(int(z[f"{name}_n"]), z[name])


def manifest(path):
    if path:
        with open(path) as f:
            return json.load(f)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_batch_test
    return gen_batch_test.build(gen_batch_test.default_graphs(55), "MaxCut", 0, 0.01, 10)[1]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", default=None, help="manifest written by scripts/gen_batch_test.py")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--passes", type=int, default=2, help="passes over the batch; the last one is `wall_s`")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launcher: before this process touches torch or the HIP library (see bench.py)
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus)]
        cmd += ["--passes", str(args.passes)]
        if args.batch:
            cmd += ["--batch", args.batch]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    import sdplrplus_jl_amd as sj
    from sdplrplus_jl_amd import batch
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if world != args.gpus:
        sys.exit(f"run_batch.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dist, device = None, None
    if world > 1:   # torch before the HIP library: one HIP runtime per process (same SONAME in the torch wheel)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=device)
    abi = sj.load_hip()
    assert abi.set_device(local_rank) == 0
    entries = manifest(args.batch)
    builders = {"MaxCut": problems.maxcut_data, "MinimumBisection": problems.minimum_bisection_data,
                "LovaszTheta": problems.lovasz_theta_data, "CutNorm": problems.cutnorm_data}
    e0 = entries[0]
    assert all(e["problem"] == e0["problem"] and e["rank"] == e0["rank"] and e["ptol"] == e0["ptol"] for e in entries)
    graphs = [load_graph(e["graph"]) for e in entries]
    abi.device_synchronize()   # the HIP context (≈ 0.15 s, once per process) is created before the clock starts …
    conc = int(os.environ.get("SDPLR_BATCH_CONCURRENCY", "16"))
    lockstep_mode = os.environ.get("SDPLR_BATCH_MODE", "lockstep") == "lockstep"
    live = len(batch.assign(len(graphs), world)[rank]) if lockstep_mode else conc     # handles alive at once
    assert abi.warmup(max(1, min(live, 256))) == 0                 # … with the library's stream / staging pools …
    import sdplrplus_jl_amd as _sj                                   # … and the device code (first launch of the module)
    _sj.sdplr(data=problems.maxcut_data(problems.gnp_graph(32, 0.3, 1)), r=2, printlevel=0, ptol=1e-1, objtol=1e-1)
    # The reference's clock (`totaltime`, src/sdplr.jl:127-131,416) starts at the sdplr() call: the problem (C, As, b —
    # exps/test.jl:166-176 builds it before) is an input; preprocessing, the solve and the dual bounds are inside.
    tb0 = time.perf_counter()
    mine = set(batch.assign(len(graphs), world)[rank])
    datas = [builders[e0["problem"]](g) if k in mine else None for k, g in enumerate(graphs)]
    build_s = time.perf_counter() - tb0
    tb = 1.0 if e0["problem"] == "LovaszTheta" else float(max(g.shape[0] for g in graphs))     # exps/test.jl:166-176 (CutNorm too: n of the graph, not 2n)
    lockstep = os.environ.get("SDPLR_BATCH_MODE", "lockstep") == "lockstep"     # ("threads": independent driver threads)
    # Two passes over the batch: the first one also fills the library's pools (streams, pinned blocks, device blocks of the
    # sizes this batch uses — a lockstep batch has all its handles alive at once), the second one is the steady state of a
    # process that solves batch after batch.  Both walls are reported.
    walls = []
    for _ in range(max(1, args.passes)):
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        local = batch.solve_local(datas, rank, world, e0["rank"], concurrency=conc, lockstep=lockstep,
                                  ptol=e0["ptol"], objtol=e0["objtol"], seed=e0["seed"], prior_trace_bound=tb)
        res = batch.gather(local, len(graphs), dist, device)
        walls.append(time.perf_counter() - t0)
    dt = walls[-1]
    if rank == 0:
        print(json.dumps({"instances": len(graphs), "n_gpus": world, "wall_s": dt, "first_pass_wall_s": walls[0],
                          "problem_build_s": build_s,
                          "mode": "lockstep" if lockstep else "threads", "in_flight_per_gpu": conc, "instances_per_s": len(graphs) / dt,
                          "objectives": [round(x, 4) for x in res[:, 1]],
                          "dual_bounds": [round(x, 4) for x in res[:, 2]],
                          "iterations": [int(x) for x in res[:, 3]]}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
