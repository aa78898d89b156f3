"""Batches of independent SDP instances — the reference's only parallelism (GNU parallel over graphs,
exps/README.md:17-21, exps/batch_test.txt) mapped to GPUs: instance k → rank k mod world, one process
per GPU, several instances in flight per GPU (one handle = one HIP stream each), and ONE collective at
the end that gathers (objective, dual bound, iterations, seconds) per instance (RCCL over xGMI when the
backend is "nccl"; gloo in the CPU tests).
"""
from __future__ import annotations

import time
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence

import numpy as np

from .sdplr import sdplr

N_FIELDS = 5  # index, obj, max_dual_value, iter, seconds


def assign(n_instances: int, world: int) -> List[List[int]]:
    """Static round-robin partition: instance k belongs to rank k mod world."""
    return [list(range(r, n_instances, world)) for r in range(world)]


def solve_local(instances: Sequence, rank: int, world: int, r: int, *, abi=None, concurrency: int = 8,
                make_data: Optional[Callable] = None, **kwargs) -> np.ndarray:
    """Solve this rank's share.  `instances[k]` is an SDPData, or anything `make_data` turns into one.
    Returns an array [n_local, N_FIELDS]."""
    mine = assign(len(instances), world)[rank]

    def one(k):
        data = make_data(instances[k]) if make_data is not None else instances[k]
        t0 = time.perf_counter()
        res = sdplr(data=data, r=r, abi=abi, printlevel=0, **kwargs)
        return [float(k), res["obj"], res["max_dual_value"], float(res["iter"]), time.perf_counter() - t0]

    if not mine:
        return np.zeros((0, N_FIELDS))
    if concurrency <= 1 or len(mine) == 1:
        rows = [one(k) for k in mine]
    else:  # ctypes releases the GIL inside the library, so the handles' streams really overlap
        # A thread that comes back from the library has to take the GIL again; CPython only asks the thread that holds it
        # to let go every `switchinterval` (5 ms by default) — with 16 drivers making ≈ 100 short calls per solve that
        # wait, not the calls, was most of a small solve's wall time.
        import sys
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, 5e-5))
        try:
            with ThreadPoolExecutor(max_workers=min(concurrency, len(mine))) as ex:
                rows = list(ex.map(one, mine))
        finally:
            sys.setswitchinterval(old_interval)
    return np.asarray(rows, dtype=np.float64)


def gather(local: np.ndarray, n_instances: int, dist=None, device=None) -> np.ndarray:
    """All ranks' result rows, ordered by instance index ([n_instances, N_FIELDS]) — one all_gather."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = local
    else:
        import torch
        world = dist.get_world_size()
        cap = (n_instances + world - 1) // world
        buf = torch.full((cap, N_FIELDS), float("nan"), dtype=torch.float64, device=device)
        if local.shape[0]:
            buf[: local.shape[0]] = torch.as_tensor(local, dtype=torch.float64, device=device)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        out = torch.cat(parts).cpu().numpy()
        out = out[~np.isnan(out[:, 0])]
    order = np.argsort(out[:, 0])
    return out[order]
