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


_KIND_POOL = None


def _kind_pool() -> ThreadPoolExecutor:
    global _KIND_POOL
    if _KIND_POOL is None:
        _KIND_POOL = ThreadPoolExecutor(max_workers=3, thread_name_prefix="sdplr-batch-kind")
    return _KIND_POOL


EXIT_ITERS = 2   # exit_reason of the inner loop: the iteration budget (include/sdplr_hip.h)


class _Cap:
    """Inner iterations per lockstep launch (``SDPLR_LOCKSTEP_CAP``; default 0: none — on BASELINE config 5 the batch's wall
    is its slowest instance's own path, 403 inner iterations on one CU, and shorter rounds only add their fixed costs:
    32.7 ms of batch calls uncapped, 40.3 ms at 64).  A major iteration whose budget
    is larger goes out capped; while it keeps coming back on the budget exit it is RESUMED (``MAJOR_RESUME``: no prologue,
    the loop continues on the state the launch left — bit for bit the iterates of the uncapped call), and only the final
    answer — with the iterations of all its launches — reaches the instance's stepper.  A round then lasts as long as
    ``cap`` iterations, not as long as its slowest member's whole inner loop, and the instances that have left their
    loop move on to their dual bound in the next round."""

    def __init__(self, cap: int):
        self.cap = int(cap)
        self.open = {}        # instance → [original request, iterations so far, wall at first send, (ℒ, ‖grad‖, ‖pv‖)]

    def outgoing(self, k, req):
        if self.cap <= 0 or req[0] != REQ_MAJOR:
            return req
        st = self.open.get(k)
        if st is None:
            if req[10] <= self.cap:
                return req
            self.open[k] = [req, 0, time.time(), None]
            return req[:10] + (self.cap,) + req[11:]
        left = req[10] - st[1]
        tleft = req[11] - (time.time() - st[2])
        return req[:6] + (cabi.MAJOR_RESUME,) + req[7:10] + (min(self.cap, left), max(tleft, 1e-9)) + st[3]

    def incoming(self, k, resp):
        """→ the response to forward to the stepper, or None: the instance's request stays pending (resumed next round)."""
        st = self.open.get(k)
        if st is None:
            return resp
        if isinstance(resp, Exception):
            del self.open[k]
            return resp
        L, gn, pn, alpha, it, why, obj = resp
        sent = min(self.cap, st[0][10] - st[1])
        st[1] += it
        if why == EXIT_ITERS and it == sent and st[1] < st[0][10]:
            st[3] = (L, gn, pn)
            return None
        del self.open[k]
        return (L, gn, pn, alpha, st[1], why, obj)


def serve_batch(abi, solvers, reqs) -> list:
    """The pending request of every instance, served side by side: the requests of one kind are ONE library call
    (``sdplr_hip_batch_*``: one kernel launch for all the small instances among them); the rest go one by one.
    → per instance what ``sdplr.serve`` returns, or the exception its call raised."""
    out = [None] * len(solvers)
    jobs = []
    for kind, call in ((REQ_MAJOR, cabi.batch_major_iteration), (REQ_DUAL, cabi.batch_dual_obj), (REQ_FG, cabi.batch_fg)):
        ks = [k for k, q in enumerate(reqs) if q[0] == kind]
        if ks:
            jobs.append((ks, call))

    def run(job):
        ks, call = job
        try:
            return call(abi, [solvers[k] for k in ks], [reqs[k][1:] for k in ks])
        except Exception as e:               # (a call that failed as a whole: every instance of it gets the error)
            return [e] * len(ks)

    # the kinds of one round are independent library calls on disjoint handles (each grid on its first handle's stream):
    # side by side, so that the dual bounds of the instances that have finished a major iteration run beside the inner
    # loops of those that have not (ctypes releases the GIL inside the library)
    results = list(_kind_pool().map(run, jobs)) if len(jobs) > 1 else [run(j) for j in jobs]
    for (ks, _), res in zip(jobs, results):
        for k, r_ in zip(ks, res):
            out[k] = r_
    for k, q in enumerate(reqs):
        if out[k] is None:
            try:
                out[k] = serve(solvers[k], q)
            except Exception as e:           # handed to that instance's stepper, like the batched calls' errors
                out[k] = e
    return out


def solve_lockstep(datas: Sequence, r: int, *, abi=None, setup_workers: int = 8, iteration_cap: Optional[int] = None,
                   overlap_setup: Optional[bool] = None, **kwargs) -> list:
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

    import os
    from concurrent.futures import FIRST_COMPLETED, wait
    configs = [config_of() for _ in datas]
    n_inst = len(datas)
    results: list = [None] * n_inst
    solvers: list = [None] * n_inst
    steppers: list = [None] * n_inst
    setup_s = [0.0] * n_inst
    pending = {}

    # Set-up (preprocessing, layout, uploads) is host work per instance, on a few threads; the rounds start when every handle
    # is ready.  SDPLR_LOCKSTEP_OVERLAP=1 lets an instance join as soon as ITS handle is ready — measured on BASELINE config 5:
    # 0.116 s against 0.057 s (0.089 s with launches capped at 64 iterations): the instances then sit at different stages, every
    # round is a mix of long inner loops and short dual bounds and lasts as long as its longest member, and the late joiners
    # wait for rounds sized by the early ones.  Lockstep pays because the instances ARE in step.  Each instance's own sequence
    # of steps, and so its result, is the same either way.
    def build(k):
        t0 = time.time()
        v = build_solver(abi, datas[k], int(r), configs[k])
        setup_s[k] = time.time() - t0
        return v

    overlap = overlap_setup if overlap_setup is not None else os.environ.get("SDPLR_LOCKSTEP_OVERLAP", "0") == "1"
    ex = ThreadPoolExecutor(max_workers=min(setup_workers, n_inst)) if setup_workers > 1 and n_inst > 1 else None
    waiting = {}
    if ex is not None:
        waiting = {ex.submit(build, k): k for k in range(n_inst)}

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
            ans["preprocess_time"] = setup_s[k]
            ans["totaltime"] += setup_s[k]
            results[k] = ans
            solvers[k].close()
        except Exception as e:
            pending.pop(k, None)
            results[k] = e
            solvers[k].close()

    def admit(futs):
        for f in futs:
            k = waiting.pop(f)
            try:
                solvers[k] = f.result()
            except Exception as e:
                results[k] = e
                continue
            steppers[k] = sdplr_steps(datas[k], solvers[k], configs[k])
            advance(k, None)

    try:
        if ex is None:
            for k in range(n_inst):
                try:
                    solvers[k] = build(k)
                except Exception as e:
                    results[k] = e
                    continue
                steppers[k] = sdplr_steps(datas[k], solvers[k], configs[k])
                advance(k, None)
        elif not overlap:
            admit(list(wait(list(waiting)).done))
        cap = _Cap(int(os.environ.get("SDPLR_LOCKSTEP_CAP", "0")) if iteration_cap is None else iteration_cap)
        while pending or waiting:
            if waiting:   # whoever is ready joins; with nothing to serve, wait for the next handle
                ready = [f for f in waiting if f.done()]
                if not ready and not pending:
                    ready = list(wait(list(waiting), return_when=FIRST_COMPLETED).done)
                admit(ready)
                if not pending:
                    continue
            ks = sorted(pending)
            reqs = [cap.outgoing(k, pending[k]) for k in ks]
            for k, response in zip(ks, serve_batch(abi, [solvers[k] for k in ks], reqs)):
                response = cap.incoming(k, response)
                if response is not None:
                    advance(k, response)
    finally:
        if ex is not None:
            for f in list(waiting):
                f.cancel()
            ex.shutdown(wait=True)
            for f, k in list(waiting.items()):      # (handles finished after an error aborted the rounds)
                if f.done() and not f.cancelled() and f.exception() is None:
                    f.result().close()
        for v in solvers:
            if v is not None:
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
