#!/usr/bin/env python3
"""BASELINE.json configs[4]: a batch of 64 MaxCut instances — Gset G1–G9 (the reference's batch,
exps/batch_test.txt: rank 10, ptol = objtol = 0.01) + 55 seeded G(800, 0.06) graphs — sharded
round-robin over the ranks, 8 in flight per GPU, objectives gathered with one RCCL all_gather.

    python scripts/run_batch.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_batch.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdplrplus_jl_amd as sj  # noqa: E402
from sdplrplus_jl_amd import batch, problems  # noqa: E402


def instances():
    z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
    gs = [problems.graph_from_edges(int(z[f"G{k}_n"]), z[f"G{k}"]) for k in range(1, 10)]
    gs += [problems.gnp_graph(800, 0.06, seed) for seed in range(10, 65)]
    return gs


def main():
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    dist, device = None, None
    if world > 1:   # torch before the HIP library: one HIP runtime per process (same SONAME in the torch wheel)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=device)
    abi = sj.load_hip()
    assert abi.set_device(local_rank) == 0
    graphs = instances()
    t0 = time.perf_counter()
    conc = int(os.environ.get("SDPLR_BATCH_CONCURRENCY", "8"))
    local = batch.solve_local(graphs, rank, world, 10, concurrency=conc, make_data=problems.maxcut_data,
                              ptol=0.01, objtol=0.01, seed=0, prior_trace_bound=800.0)
    res = batch.gather(local, len(graphs), dist, device)
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"instances": len(graphs), "n_gpus": world, "wall_s": dt,
                          "instances_per_s": len(graphs) / dt,
                          "objectives": [round(x, 4) for x in res[:, 1]],
                          "dual_bounds": [round(x, 4) for x in res[:, 2]],
                          "iterations": [int(x) for x in res[:, 3]]}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
