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

from . import cabi
from .sdplr import _ASCII_ALIASES, REQ_DUAL, REQ_FG, REQ_MAJOR, build_solver, sdplr, sdplr_steps, serve
from .structs import BurerMonteiroConfig

N_FIELDS = 5  # index, obj, max_dual_value, iter, seconds


def assign(n_instances: int, world: int) -> List[List[int]]:
    """Static round-robin partition: instance k belongs to rank k mod world."""
    return [list(range(r, n_instances, world)) for r in range(world)]


def solve_local(instances: Sequence, rank: int, world: int, r: int, *, abi=None, concurrency: int = 8,
                make_data: Optional[Callable] = None, lockstep: bool = False, **kwargs) -> np.ndarray:
    """Solve this rank's share.  `instances[k]` is an SDPData, or anything `make_data` turns into one.
    ``lockstep``: side by side through ``solve_lockstep`` (one launch per step for all of them) instead of as
    ``concurrency`` independent driver threads.  Returns an array [n_local, N_FIELDS]."""
    mine = assign(len(instances), world)[rank]
    if lockstep and mine:
        datas = [make_data(instances[k]) if make_data is not None else instances[k] for k in mine]
        # (set-up threads: 64 config-5 instances take 78 / 24 / 15 / 14 / 20 ms on 1 / 4 / 8 / 12 / 16 threads —
        # scripts/probes/setup_scaling.py: beyond ≈ 8 the HIP runtime's own locks and the GIL take the gain back)
        res = solve_lockstep(datas, r, abi=abi, setup_workers=min(concurrency, 8), printlevel=0, **kwargs)
        for x in res:
            if isinstance(x, Exception):
                raise x
        return np.asarray([[float(k), x["obj"], x["max_dual_value"], float(x["iter"]), x["totaltime"]]
                           for k, x in zip(mine, res)], dtype=np.float64)

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
        # (measured: no change — the waits are inside the HIP runtime — so the process-wide setting is only touched when
        # asked for: SDPLR_BATCH_SWITCHINTERVAL=<seconds>)
        import os
        import sys
        old_interval = sys.getswitchinterval()
        want = os.environ.get("SDPLR_BATCH_SWITCHINTERVAL")
        if want:
            sys.setswitchinterval(min(old_interval, float(want)))
        try:
            with ThreadPoolExecutor(max_workers=min(concurrency, len(mine))) as ex:
                rows = list(ex.map(one, mine))
        finally:
            if want:
                sys.setswitchinterval(old_interval)
    return np.asarray(rows, dtype=np.float64)


def serve_batch(abi, solvers, reqs) -> list:
    """The pending request of every instance, served side by side: the requests of one kind are ONE library call
    (``sdplr_hip_batch_*``: one kernel launch for all the small instances among them); the rest go one by one.
    → per instance what ``sdplr.serve`` returns, or the exception its call raised."""
    out = [None] * len(solvers)
    for kind, call in ((REQ_MAJOR, cabi.batch_major_iteration), (REQ_DUAL, cabi.batch_dual_obj), (REQ_FG, cabi.batch_fg)):
        ks = [k for k, q in enumerate(reqs) if q[0] == kind]
        if ks:
            for k, res in zip(ks, call(abi, [solvers[k] for k in ks], [reqs[k][1:] for k in ks])):
                out[k] = res
    for k, q in enumerate(reqs):
        if out[k] is None:
            try:
                out[k] = serve(solvers[k], q)
            except Exception as e:           # handed to that instance's stepper, like the batched calls' errors
                out[k] = e
    return out


def solve_lockstep(datas: Sequence, r: int, *, abi=None, setup_workers: int = 8, **kwargs) -> list:
    """``sdplr`` on every SDPData of ``datas`` side by side on ONE device: the solves advance in lockstep — each round
    serves the pending device step of all live instances as one call (``serve_batch``) — instead of as independent
    threads whose launches share the GPU only as far as its hardware queues allow.  Same control flow (``sdplr_steps``),
    same results as ``sdplr(data=…)`` one by one.  → the list of result Dicts (an instance that failed: its exception).

    Times are BATCH walls, not per-instance times: an instance's clock (``totaltime``, ``primaltime``, ``dual_time``, and the
    ``maxtime`` budget it is tested against) starts at its first step and keeps running while the other instances of the
    round are served, so every instance reports roughly the whole batch's wall (plus the mean set-up time).  Iterations,
    objective, dual bound, R and λ are those of the one-by-one solve, bit for bit.  All handles live on the calling thread's
    device (``abi.set_device`` is sticky: the set-up threads and the library's own workers bind to it)."""
    abi = abi if abi is not None else cabi.load_hip()

    def config_of():
        config = BurerMonteiroConfig()
        for key, value in kwargs.items():
            key = _ASCII_ALIASES.get(key, key)
            if not hasattr(config, key):
                raise TypeError(f"unrecognized keyword argument {key}")
            setattr(config, key, value)
        return config

    configs = [config_of() for _ in datas]
    t0 = time.time()
    # set-up (preprocessing, layout, uploads) is host work per instance: a few threads
    if setup_workers > 1 and len(datas) > 1:
        with ThreadPoolExecutor(max_workers=min(setup_workers, len(datas))) as ex:
            solvers = list(ex.map(lambda kc: build_solver(abi, kc[0], int(r), kc[1]), zip(datas, configs)))
    else:
        solvers = [build_solver(abi, d, int(r), c) for d, c in zip(datas, configs)]
    setup_dt = (time.time() - t0) / max(len(datas), 1)
    results: list = [None] * len(datas)
    steppers = [sdplr_steps(d, v, c) for d, v, c in zip(datas, solvers, configs)]
    pending = {}

    def advance(k, response):
        try:
            if response is None:
                pending[k] = next(steppers[k])
            elif isinstance(response, Exception):
                pending[k] = steppers[k].throw(response)
            else:
                pending[k] = steppers[k].send(response)
        except StopIteration as done:
            pending.pop(k, None)
            ans = done.value
            ans["preprocess_time"] = setup_dt
            ans["totaltime"] += setup_dt
            results[k] = ans
            solvers[k].close()
        except Exception as e:
            pending.pop(k, None)
            results[k] = e
            solvers[k].close()

    try:
        for k in range(len(datas)):
            advance(k, None)
        while pending:
            ks = sorted(pending)
            for k, response in zip(ks, serve_batch(abi, [solvers[k] for k in ks], [pending[k] for k in ks])):
                advance(k, response)
    finally:
        for v in solvers:
            v.close()
    return results


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
