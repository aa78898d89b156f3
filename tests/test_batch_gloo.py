"""N > 1 path on CPU: world_size-2 gloo processes, instances sharded round-robin, one all_gather of the
per-instance results (the only collective of the workload).  The oracle ABI is injected because there is
no GPU here; on the GPU box bench.py / scripts/run_batch.py run the same code on the HIP library."""
import os
import socket

import numpy as np
import pytest

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import batch, problems


def _instances():
    return [problems.maxcut_data(problems.gnp_graph(24, 0.3, 100 + k)) for k in range(5)]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    inst = _instances()
    local = batch.solve_local(inst, rank, world, 3, abi=oracle.abi(), concurrency=2, seed=5,
                              prior_trace_bound=24.0)
    allres = batch.gather(local, len(inst), dist)
    if rank == 0:
        q.put(allres)
    dist.barrier()
    dist.destroy_process_group()


def test_assign_is_a_partition():
    parts = batch.assign(64, 8)
    assert sorted(sum(parts, [])) == list(range(64)) and all(len(p) == 8 for p in parts)
    assert batch.assign(3, 8)[5] == []


def test_two_rank_gloo_gather_matches_serial(oracle_abi):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    inst = _instances()
    serial = batch.gather(batch.solve_local(inst, 0, 1, 3, abi=oracle_abi, concurrency=1, seed=5,
                                            prior_trace_bound=24.0), len(inst))
    assert got.shape == serial.shape == (5, batch.N_FIELDS)
    assert np.array_equal(got[:, 0], np.arange(5.0))
    assert np.allclose(got[:, 1:4], serial[:, 1:4], rtol=1e-12)   # same seeds ⇒ same solves, any sharding


def _bench_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    dt_max, objs, dts = bench.reduce_over_ranks(dist, 0.25 * (rank + 1), -100.0 - rank, None)
    q.put((rank, dt_max, objs, dts))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_rank_reduction_gloo():
    """bench.py's N > 1 reduction: MAX of the timed region over ranks + all_gather of the objectives and of every
    rank's own seconds (a straggler is visible in the line, not only in the maximum)."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, dt_max, objs, dts in got:
        assert dt_max == 0.5 and objs == [-100.0, -101.0] and dts == [0.25, 0.5]
