#!/usr/bin/env python3
"""bench.py — augmented-Lagrangian inner iterations/s of the SDPLR+ hot path on MI355X.

One "step" = one pass of the inner `while` body of _sdplr (src/sdplr.jl:190-278): L-BFGS direction,
exact line search (two 𝒜 passes), R += αD, g! (S assembly + SpMM), norms, L-BFGS update — on the
north-star instance of BASELINE.json configs[1]: MaxCut on G(n = 1e5, p = 2e-4), r = 32, FP64,
h = 4, σ = σ₀ = 2 fixed, no early exit (gtol = 0, fprec = −∞), all state resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: independent replicas of the instance (graph seed + rank), one process per GPU, no data-path
collective; RCCL only gathers the objectives at the end.  value = N·K / max-over-ranks time.
Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a launcher: before it has
imported torch or loaded the HIP library (it never touches the GPU) it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 … bench.py --gpus N …`
as a child, relays rank 0's JSON line and exits with the child's code.  WORLD_SIZE ≠ N is an error.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (GPU_MAX_HW_QUEUES is left at the runtime's default: config 5 runs in lockstep — one stream carries the whole batch — and
# a process with 16–32 hardware queues makes every launch and wait of that driver 3–6× slower.  The threaded driver that is
# timed next to it would like one queue per instance in flight; SDPLR_BENCH_HW_QUEUES=16 gives it that.)
if os.environ.get("SDPLR_BENCH_HW_QUEUES"):
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["SDPLR_BENCH_HW_QUEUES"]

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming)
N_NODES, P_EDGE, RANK_R, GRAPH_SEED, R_SEED = 100_000, 2e-4, 32, 20240610, 0
PARITY_ITERS = 5


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs block (configs 3–5)")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="launcher/collective self-test WITHOUT a GPU: gloo + the CPU checker on a small instance; "
                         "exercises spawn, rendezvous, barrier, max-over-ranks and the gather — not a measurement")
    return ap.parse_args(argv)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_children(args) -> int:
    """The parent of an N-rank run: one fresh process per GPU through torch.distributed.run.  Nothing in this
    process has imported torch or loaded libsdplr_hip.so, so no GPU state exists here to fork or exec over."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_other_configs:
        cmd.append("--no-other-configs")
    if args.selftest_cpu:
        cmd.append("--selftest-cpu")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    line, rc = None, 1
    for attempt in range(3):
        # free_port() closes its socket before torchrun binds the port: if somebody else took it meanwhile the ranks die
        # at the rendezvous within seconds — a fresh port and another try (a run that got past the rendezvous is not retried)
        cmd[cmd.index("--master-port") + 1] = str(free_port())
        t_start = time.time()
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
        line = None
        for out in proc.stdout:                 # relay rank 0's JSON line (the only thing ranks write to stdout)
            out = out.rstrip("\n")
            if out.startswith("{") and '"metric"' in out:
                line = out
            elif out:
                print(out, file=sys.stderr, flush=True)
        rc = proc.wait()
        if rc == 0 or time.time() - t_start > 30:
            break
        print(f"bench.py: the {args.gpus}-rank launch failed early (code {rc}); retrying on another port", file=sys.stderr)
    if rc != 0:
        print(f"bench.py: the {args.gpus}-rank run exited with code {rc}", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def algorithmic_bytes(d, h):
    """Compulsory bytes per launch of each kernel family and per inner iteration:
    each distinct input read once, each output written once; 8-B values, 4-B indices.  A fused kernel is
    charged the fused dataflow's own compulsory bytes (never more than the sum of the operators it fuses)."""
    n, m, r, nnzT, nnzS, nnzAgg = d["n"], d["m"], d["r"], d["nnzT"], d["nnzS"], d["nnzAgg"]
    N = 8 * n * r
    A_sparse = 4 * (n + 1) + 4 * nnzT + 12 * nnzAgg + 8 * (m + 1)
    per_kernel = {
        "lbfgs_dir": (2 * h + 1) * N + 2 * N,                        # G, s_*, y_* in; dirt, y_next out
        "lbfgs_dir_noynext": (2 * h + 1) * N + N,                    # in-loop form when lbfgs_update! is fused: y_next not parked
        "lbfgs_update": 5 * N,
        "sddmm_linesearch": 2 * N + 4 * (n + 1) + 4 * nnzT,          # R, D rows + pattern (both 𝒜 passes fused)
        "segreduce": 2 * (8 * nnzT + 12 * nnzAgg + 8 * (m + 1)),
        "axpy_R": 3 * N,
        "assemble_triu": 12 * nnzAgg + 8 * (m + 1) + 8 * nnzT,
        "assemble_full": 4 * nnzS + 8 * nnzT + 8 * nnzS,
        "spmm": 2 * N + 4 * (n + 1) + 12 * nnzS,
        # structured fast path: W = A_g·D with the row dots riding along; the fused step
        "spmm_W": 3 * N + 4 * (n + 1) + 12 * nnzS,    # D rows (once), R read; W written; A_g pattern+values (⟨P,D⟩ is formed as ⟨R,W⟩)
        "fast_step": 7 * N,                            # R, D, P, W read; R, P, G written
        # … with lbfgs_update! fused in: R, D, P, W, G_old in; R, P, G, s_j, y_j out (the other history pairs the
        # Gram form re-reads are this design's choice, not compulsory)
        "fast_step_upd": 10 * N,
        # … and without P (the gradient carried forward from G_old; stats: p_less_loops): R, D, W, G_old in; R, G, s_j, y_j out
        "fast_step_upd_pless": 8 * N,
        # … and on the ring form of the history (stats: ring_history_loops): s_j = α_j·D_j and y_j = G_{j+1} − G_j are not stored
        # at all — R, D, W, G_old in; R, G out
        "fast_step_ring": 6 * N,
        "fast_step_ring_pb": 8 * N,                   # … the P-based kernel on the ring form: R, D, P, W, G_old in; R, P, G out
    }
    b_iter = ((2 * h + 1) * N + 2 * N      # lbfgs_dir!
              + 2 * N                      # dot(dirt, Gt)
              + 2 * N + A_sparse           # 𝒜!(A_RD; Rt, Dt)
              + N + A_sparse               # 𝒜!(A_DD; Dt, Dt)
              + 6 * 8 * m                  # line-search scalar stage
              + 3 * N                      # axpy!
              + 8 * 4 * (m + 1) + 12 * nnzAgg + 8 * nnzT + 4 * nnzS + 8 * nnzT + 8 * nnzS  # copy2y + S assembly
              + 2 * N + 4 * (n + 1) + 12 * nnzS   # SpMM
              + N                          # norm(Gt)
              + 5 * N)                     # lbfgs_update!
    return per_kernel, b_iter


def run_fixed(var, normC, normb, state, iters):
    """`iters` inner iterations at fixed σ with every exit test disabled except the iteration budget."""
    out = var.inner_loop(normC, normb, True, True, False, 0.0, -1e300, iters, 0.0, *state)
    if out[4] != iters:
        raise RuntimeError(f"inner loop stopped after {out[4]} of {iters} iterations (exit {out[5]})")
    return out[:3]


def first_iterations(sj, abi, data, r, n_iters):
    """fg! + `n_iters` inner iterations from (R₀ seed R_SEED, λ = 0, σ₀): the state the parity field compares.
    → (solver, [ℒ, ‖grad‖, ‖pv‖, obj])."""
    import numpy as np
    cfg = sj.BurerMonteiroConfig(seed=R_SEED, printlevel=0)
    var = sj.build_solver(abi, data, r, cfg)
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    st = run_fixed(var, normC, normb, var.fg(normC, normb), n_iters)
    return var, [st[0], st[1], st[2], var.obj]


def parity_ok(parity, gpu_first, cpu_first, tol=1e-8) -> bool:
    """ℒ, ‖grad‖, objective and ‖primal_vio‖ within `tol` relative — ‖primal_vio‖ absolutely (tol · max(1, ‖b‖-scale)) when
    it is itself at round-off level, where a relative error means nothing."""
    pv_ok = parity["rel_pv"] < tol or abs(gpu_first[2] - cpu_first[2]) < tol
    return bool(max(parity["rel_L"], parity["rel_grad"], parity["rel_obj"]) < tol and pv_ok)


def effective_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a
    16-CPU share of a much larger host to one GPU; sizing an OpenMP team to the host would oversubscribe it)."""
    n = os.cpu_count() or 1
    if hasattr(os, "sched_getaffinity"):
        n = min(n, len(os.sched_getaffinity(0)))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.999)))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(quota / int(g.read().split()[0]) + 0.999)))
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, int(os.environ.get("SDPLR_BENCH_MAX_THREADS", "64"))))


def cpu_baseline(sj, data, r, gpu_first, repeats=5, sample_s=2.5):
    """The oracle (CPU restatement, kind "port") on the same instance, same host, same run: ONE thread (the
    reference's own protocol, exps/test.jl:46) and all cores (OpenMP in its dense sweeps, SDDMM and SpMM) beside
    it, each the median of `repeats` samples of ≈`sample_s` seconds of inner iterations after fg! + the
    PARITY_ITERS iterations whose (ℒ, ‖grad‖, obj) are compared with the GPU's — the parity evidence of this very
    run.  Bounded: ≈ 2·(repeats·sample_s) seconds of CPU work plus the set-up."""
    import numpy as np
    from oracle import oracle
    one_abi, omp_abi, how, set_threads = oracle.timing_abis()
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    ncores = effective_cpus()

    def timed(abi, threads):
        set_threads(threads)
        t0 = time.perf_counter()
        var, first = first_iterations(sj, abi, data, r, PARITY_ITERS)
        st = first[:3]
        t1 = time.perf_counter()
        st = run_fixed(var, normC, normb, st, 2)             # sizes the samples
        per = (time.perf_counter() - t1) / 2
        iters = int(max(3, min(20, sample_s / max(per, 1e-6))))
        rates = []
        for _ in range(repeats):
            t1 = time.perf_counter()
            st = run_fixed(var, normC, normb, st, iters)
            rates.append(iters / (time.perf_counter() - t1))
        var.close()
        print(f"bench.py: cpu_baseline {threads} thread(s): {statistics.median(rates):.2f} it/s "
              f"({repeats} x {iters} iterations, {time.perf_counter() - t0:.1f} s)", file=sys.stderr, flush=True)
        return statistics.median(rates), first, iters

    v_all, _, it_all = timed(omp_abi, ncores)
    v_one, first, it_one = timed(one_abi, 1)
    relerr = lambda a, b: abs(a - b) / max(abs(b), 1e-300)
    parity = {"iters": PARITY_ITERS, "rel_L": relerr(gpu_first[0], first[0]), "rel_grad": relerr(gpu_first[1], first[1]),
              "rel_pv": relerr(gpu_first[2], first[2]), "rel_obj": relerr(gpu_first[3], first[3]),
              "tolerance": 1e-8, "against": "oracle (CPU restatement), same R0/λ0/σ0, fg! + 5 inner iterations"}
    parity["ok"] = parity_ok(parity, gpu_first, first)
    sample = (f"same MaxCut G(1e5,2e-4) r={r} instance; median of {repeats} samples of {it_one} inner iterations "
              f"after fg! + {PARITY_ITERS} + 2 iterations")
    base = {"value": v_one, "unit": "iterations/s", "cores": 1, "kind": "port", "sample": sample, "build": how,
            "all_cores": {"value": v_all, "unit": "iterations/s", "cores": ncores, "kind": "port",
                          "sample": f"median of {repeats} samples of {it_all} inner iterations",
                          "threads": "OpenMP over the dense n·r sweeps, the SDDMM and the SpMM of the oracle"}}
    return base, parity


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        print("bench.py: --gpus must be ≥ 1", file=sys.stderr)
        return 2
    if env_world is None and (args.gpus > 1 or os.environ.get("SDPLR_BENCH_FORCE_LAUNCH")):
        # (SDPLR_BENCH_FORCE_LAUNCH: take the launcher route for N = 1 too — rehearses spawn + RCCL on a one-GPU box)
        return launch_children(args)            # the parent never imports torch / the HIP library
    if env_world is not None and int(env_world) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch with --nproc-per-node {args.gpus}",
              file=sys.stderr)
        return 2
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    K, W = max(args.steps, 1), max(args.warmup, 1)

    import numpy as np
    import sdplrplus_jl_amd as sj
    from sdplrplus_jl_amd import problems
    selftest = args.selftest_cpu
    dist = None
    if world > 1 or os.environ.get("SDPLR_BENCH_FORCE_DIST") or os.environ.get("SDPLR_BENCH_FORCE_LAUNCH"):
        # torch first: its wheel bundles a HIP runtime with the same SONAME as /opt/rocm's, and whichever is
        # loaded first serves both torch and libsdplr_hip.so — one runtime per process either way.  (The wheel's runtime is
        # the older one — HIP 7.0 against /opt/rocm's 7.2 — and replays the iteration graphs ≈ 2.4 µs per kernel boundary
        # slower: 5865 against 6210 it/s on one GPU with identical kernel times, DESIGN §7.  The other order — the library
        # and /opt/rocm's runtime first — was tried: the rank dies while torch initialises.)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if selftest:
        from oracle import oracle
        abi = oracle.abi()
        n_nodes, p_edge, rank_r = 300, 0.05, 8
    else:
        abi = sj.load_hip()  # no fallback: raises if the HIP library is missing
        if abi.set_device(local_rank) != 0:
            raise RuntimeError((abi.last_error(None) or b"set_device failed").decode())
        n_nodes, p_edge, rank_r = N_NODES, P_EDGE, RANK_R

    def barrier():
        abi.device_synchronize()
        if dist is not None:
            if not selftest:
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    data = problems.maxcut_data(problems.gnp_graph(n_nodes, p_edge, GRAPH_SEED + rank))
    normC, normb = data.normC(), float(np.linalg.norm(data.b))

    # the first PARITY_ITERS iterations from R₀, kept for the comparison with the oracle's (cpu_baseline)
    var, gpu_first = first_iterations(sj, abi, data, rank_r, PARITY_ITERS)
    dims = var.dims()
    h = var.h
    per_kernel_bytes, b_iter = algorithmic_bytes(dims, h)

    # Device pre-warm (setup, not steps): the MI355X takes two ~30-80 ms clock/power-state stalls
    # 20-80 ms after sustained work starts (measured: scripts/iter_trend.py); fg! is idempotent on the
    # solver state, so it is repeated for ~0.4 s to get those transitions out of the way.
    state = var.fg(normC, normb)
    t_pw = time.perf_counter()
    while not selftest and time.perf_counter() - t_pw < float(os.environ.get("SDPLR_BENCH_PREWARM_S", "0.4")):
        state = var.fg(normC, normb)
    state = run_fixed(var, normC, normb, state, W)      # warm-up steps (untimed); captures the hipGraph

    barrier()
    t0 = time.perf_counter()
    state = run_fixed(var, normC, normb, state, K)      # timed region: hipGraph batches, no events
    abi.device_synchronize()                            # this rank's K steps are complete here …
    dt = time.perf_counter() - t0
    barrier()                                           # … the closing barrier brackets the region; the MAX over ranks of
                                                        # dt (below) is the time until the slowest rank was done
    obj = var.obj

    # Per-kernel device time: the same iteration stream replayed eagerly right after the timed
    # region with a hipEvent pair around every launch, on the solver's own stream (events cannot
    # bracket individual nodes of a graph replay).  rocprofv3 --kernel-trace of this command
    # (profiles/) measures the kernels of the timed region itself and must agree.
    P = min(K, 50)
    var.profile_enable(True)
    state = run_fixed(var, normC, normb, state, P)
    prof_all = var.profile()
    var.profile_enable(False)
    if prof_all.get("fast_step", (0, 0.0))[0] and not prof_all.get("lbfgs_update", (0, 0.0))[0]:
        # lbfgs_update! rides the step kernel (k_fast_step2<…,4>): charged the fused dataflow's compulsory bytes,
        # and the in-loop direction kernel no longer parks y_next
        pless = hasattr(var, "stats") and var.stats().get("p_less_loops", 0) > 0     # … and P = A_g·R is neither read nor written
        ring = hasattr(var, "stats") and var.stats().get("ring_history_loops", 0) > 0   # … and the history is kept as (α, D), (G, G')
        per_kernel_bytes["fast_step"] = per_kernel_bytes[("fast_step_ring" if pless else "fast_step_ring_pb") if ring else
                                                         ("fast_step_upd_pless" if pless else "fast_step_upd")]
        per_kernel_bytes["lbfgs_dir"] = per_kernel_bytes["lbfgs_dir_noynext"]
    candidates = [k for k in per_kernel_bytes if k in prof_all]
    dominant = max(candidates, key=lambda k: prof_all[k][1]) if candidates else "fast_step"
    launches, dom_ms = prof_all.get(dominant, (0, 0.0))
    stats = var.stats() if hasattr(var, "stats") else {}

    dt_max, objs, dts = reduce_over_ranks(dist, dt, obj, None if (dist is None or selftest) else "cuda")

    rc = 0
    if rank == 0:
        its = world * K / dt_max
        avg_s = dom_ms / max(launches, 1) / 1e3
        achieved = per_kernel_bytes[dominant] / avg_s / 1e9 if avg_s > 0 else 0.0
        kern = {}
        for name, (cnt, ms) in sorted(prof_all.items(), key=lambda kv: -kv[1][1]):
            e = {"launches_per_step": round(cnt / P, 3), "us_per_step": round(1e3 * ms / P, 2)}
            if name in per_kernel_bytes and cnt:
                e["algorithmic_MB"] = round(per_kernel_bytes[name] / 1e6, 2)
                e["GBps"] = round(per_kernel_bytes[name] / (ms / cnt / 1e3) / 1e9, 1)
            kern[name] = e
        line = {
            "metric": "augmented-Lagrangian iters/sec + HBM GB/s, MaxCut n=1e5 r=32",
            "value": its, "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * dt_max / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "MaxCut SDP, G(n=1e5, p=2e-4) (BASELINE.json configs[1]), r=32, "
                                   "h=4 L-BFGS pairs, sigma=2 fixed, one replica per GPU",
                       "n": dims["n"], "m": dims["m"], "r": dims["r"], "nnzT": dims["nnzT"],
                       "nnzS": dims["nnzS"], "numlbfgsvecs": h, "graph_seed": GRAPH_SEED},
            "hbm_GBps_per_gpu_algorithmic": b_iter * (K / dt_max) / 1e9,
            "bytes_per_iteration_algorithmic": b_iter,
            "bytes_per_iteration_note": "SURVEY §8d: the reference's UNFUSED operator sequence (27N + sparse); "
                                        "the fused kernels below are charged their own compulsory bytes",
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": load_traffic(dominant),
                         "algorithmic_bytes_per_launch": per_kernel_bytes[dominant],
                         "avg_launch_us": 1e6 * avg_s, "launches_timed": launches,
                         # (the PMC bytes the kernel really moves over the same duration: what the memory system is doing)
                         "traffic_GBps": (load_traffic(dominant) or 0) / avg_s / 1e9 if avg_s > 0 else None,
                         "note": ("fast_step on the ring form of the L-BFGS history is charged R, D, W, G_old in and R, G out (6N): "
                                  "s_j, y_j are not stored; the 6N of history its Gram dots re-read are this design's choice"
                                  if dominant == "fast_step" and stats.get("ring_history_loops", 0) > 0 else None)},
            "iteration_roofline": {"achieved": b_iter * (K / dt_max) / 1e9, "peak": HBM_PEAK_GBPS,
                                   "unit": "GB/s", "frac": b_iter * (K / dt_max) / 1e9 / HBM_PEAK_GBPS},
            "kernels_eager_profile": kern,
            "objectives": objs,
            "per_rank": {"seconds": dts, "iterations_per_s": [K / t for t in dts],
                         "slowest_over_fastest": max(dts) / min(dts)},
            "library_stats": stats,
        }
        if selftest:
            line["selftest"] = True
            line["config"]["workload"] = (f"SELF-TEST (no GPU): MaxCut G({n_nodes},{p_edge}) r={rank_r} on the CPU "
                                          "checker over gloo — exercises the launcher and the collectives only")
        elif world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["parity"] = cpu_baseline(sj, data, rank_r, gpu_first)
            if not line["parity"]["ok"]:
                print(f"bench.py: PARITY FAILED against the oracle: {line['parity']}", file=sys.stderr)
                rc = 3
            if not args.no_other_configs:
                var.close()
                var = None
                line["other_configs"] = other_configs(sj, abi)
                bad = [k for k, v in line["other_configs"].items() if isinstance(v, dict) and "parity" in v and not v["parity"]["ok"]]
                if bad:
                    print(f"bench.py: PARITY FAILED against the oracle in {bad}", file=sys.stderr)
                    rc = 3
        elif world > 1 and not selftest and not args.no_cpu_baseline:
            # rank 0 still compares its first iterations with the oracle (one thread, ≈ 10 s): cheap, and a wrong result on
            # a multi-GPU node should not pass for a fast one
            from oracle import oracle
            ovar, cpu_first = first_iterations(sj, oracle.abi(), data, rank_r, PARITY_ITERS)
            ovar.close()
            relerr = lambda a, b: abs(a - b) / max(abs(b), 1e-300)
            par = {"iters": PARITY_ITERS, "rel_L": relerr(gpu_first[0], cpu_first[0]), "rel_grad": relerr(gpu_first[1], cpu_first[1]),
                   "rel_pv": relerr(gpu_first[2], cpu_first[2]), "rel_obj": relerr(gpu_first[3], cpu_first[3]), "tolerance": 1e-8,
                   "against": "oracle (CPU restatement) on rank 0's instance, fg! + 5 inner iterations"}
            par["ok"] = parity_ok(par, gpu_first, cpu_first)
            line["parity"] = par
            if not par["ok"]:
                print(f"bench.py: PARITY FAILED against the oracle: {par}", file=sys.stderr)
                rc = 3
        print(json.dumps(line), flush=True)
    if var is not None:
        var.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def measure_config(sj, abi, data, r, seed, K=200, W=20, P=40, parity_iters=PARITY_ITERS, with_oracle=True):
    """One BASELINE configuration that is not the headline: inner iterations/s at fixed σ (same protocol as the headline:
    fg!, pre-warm, W untimed + K timed iterations on the route the library picks), the dominant kernel's roofline entry
    (hipEvent pairs in an eager replay) and the state after fg! + `parity_iters` iterations against the one-thread oracle."""
    import numpy as np
    normC, normb = data.normC(), float(np.linalg.norm(data.b))
    arm = data.has_inequalities
    cfg = lambda: sj.BurerMonteiroConfig(seed=seed, printlevel=0)
    var = sj.build_solver(abi, data, r, cfg())
    run = lambda v, st, k: v.inner_loop(normC, normb, True, True, arm, 0.0, -1e300, k, 0.0, *st)[:3]
    st = run(var, var.fg(normC, normb), parity_iters)
    first = [st[0], st[1], st[2], var.obj]
    t0 = time.perf_counter()
    st = var.fg(normC, normb)
    while time.perf_counter() - t0 < 0.3:
        st = var.fg(normC, normb)
    st = run(var, st, W)
    abi.device_synchronize()
    t0 = time.perf_counter()
    st = run(var, st, K)
    abi.device_synchronize()
    dt = time.perf_counter() - t0
    dims, h = var.dims(), var.h
    per_kernel, _ = algorithmic_bytes(dims, h)
    N = 8 * dims["n"] * dims["r"]
    # the SpMM that produces G with lbfgs_update! riding on it (generic / edge paths): R rows, G_old, D in; G, s_j, y_j out
    per_kernel["spmm_upd"] = 6 * N + 4 * (dims["n"] + 1) + 12 * dims["nnzS"]
    var.profile_enable(True)
    st = run(var, st, P)
    prof = var.profile()
    var.profile_enable(False)
    fused_step = prof.get("fast_step", (0, 0.0))[0] and not prof.get("lbfgs_update", (0, 0.0))[0]
    fused_spmm = prof.get("spmm", (0, 0.0))[0] and not prof.get("lbfgs_update", (0, 0.0))[0] and not prof.get("fast_step", (0, 0.0))[0]
    if fused_step:
        st_ = var.stats()
        pl_, rg_ = st_.get("p_less_loops", 0) > 0, st_.get("ring_history_loops", 0) > 0
        per_kernel["fast_step"] = per_kernel[("fast_step_ring" if pl_ else "fast_step_ring_pb") if rg_ else
                                             ("fast_step_upd_pless" if pl_ else "fast_step_upd")]
        per_kernel["lbfgs_dir"] = per_kernel["lbfgs_dir_noynext"]
    if fused_spmm:
        per_kernel["spmm"] = per_kernel["spmm_upd"]
        per_kernel["lbfgs_dir"] = per_kernel["lbfgs_dir_noynext"]
    cand = [k for k in per_kernel if k in prof and prof[k][0]]
    out = {"n": dims["n"], "m": dims["m"], "r": dims["r"], "nnzT": dims["nnzT"], "nnzS": dims["nnzS"],
           "inner_iterations_per_s": K / dt, "us_per_iteration": 1e6 * dt / K, "steps": K, "warmup": W,
           "kernels_eager_profile": {k: {"launches_per_step": round(c / P, 3), "us_per_step": round(1e3 * ms / P, 2)}
                                     for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])}}
    if cand:
        dom = max(cand, key=lambda k: prof[k][1])
        cnt, ms = prof[dom]
        avg_s = ms / cnt / 1e3
        ach = per_kernel[dom] / avg_s / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBPS, "traffic": load_traffic(dom + "@" + str(dims["n"])),
                           "algorithmic_bytes_per_launch": per_kernel[dom], "avg_launch_us": 1e6 * avg_s, "launches_timed": cnt}
    if with_oracle:
        from oracle import oracle
        o = sj.build_solver(oracle.abi(), data, r, cfg())
        so = run(o, o.fg(normC, normb), parity_iters)
        cpu_first = [so[0], so[1], so[2], o.obj]
        o.close()
        relerr = lambda a, b: abs(a - b) / max(abs(b), 1e-300)
        par = {"iters": parity_iters, "rel_L": relerr(first[0], cpu_first[0]), "rel_grad": relerr(first[1], cpu_first[1]),
               "rel_pv": relerr(first[2], cpu_first[2]), "rel_obj": relerr(first[3], cpu_first[3]), "tolerance": 1e-8}
        par["ok"] = parity_ok(par, first, cpu_first)
        out["parity"] = par
    return var, out


def other_configs(sj, abi):
    """BASELINE.json configs[2..4] on this GPU, after the headline's timed region: the reference's runner records the
    same timers for every problem (exps/test.jl:109-162).  Lovász-θ (the Chung–Lu stand-in of SURVEY §8d), MinBisection
    n = 1e5 with its Lanczos run (q = 232), and the batch of 64 MaxCut instances of exps/batch_test.txt on one GPU."""
    import numpy as np
    from sdplrplus_jl_amd import batch, problems
    out = {}
    t_all = time.perf_counter()
    # ---- config 3: Lovász-θ stand-in ----
    data = problems.lovasz_theta_data(problems.chung_lu_graph(50_000, 10.0, 2.5, 3))
    var, res = measure_config(sj, abi, data, 32, 1)
    res["workload"] = "Lovász-θ SDP on a Chung–Lu power-law graph n≈5e4, |E|≈2.5e5 (SNAP stand-in), r=32: one constraint per edge + trace"
    out["config3_lovasz_theta"] = res
    var.close()
    # ---- config 4: MinBisection + Lanczos ----
    data = problems.minimum_bisection_data(problems.gnp_graph(100_000, 2e-4, 4))
    var, res = measure_config(sj, abi, data, 32, 2)
    res["workload"] = "Minimum-bisection SDP, G(n=1e5, p=2e-4), r=32: diagonal constraints + the rank-one 1ᵀX1 = 0"
    v0 = np.random.Generator(np.random.PCG64(5)).standard_normal(data.n)
    var.dual_obj(float(data.n), 0, v0)                   # S(y) of the current state; captures the Lanczos graph
    q, reps = 232, 5
    abi.device_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        al, be, k = var.lanczos(q, v0)
    abi.device_synchronize()
    dt = (time.perf_counter() - t0) / reps
    d = var.dims()
    lz_bytes = 12 * d["nnzS"] + 4 * (d["n"] + 1) + 8 * 8 * d["n"]       # S once + ≈ 8 n-vector touches per step
    res["lanczos"] = {"steps": int(k), "steps_per_s": k / dt, "us_per_step": 1e6 * dt / k, "ms_per_run": 1e3 * dt,
                      "roofline": {"bound": "hbm", "kernel": "lanczos step (SpMV + recurrence)", "achieved": lz_bytes * (k / dt) / 1e9,
                                   "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": lz_bytes * (k / dt) / 1e9 / HBM_PEAK_GBPS,
                                   "algorithmic_bytes_per_step": lz_bytes, "traffic": load_traffic("lanczos_step@100000")},
                      "ritz_value": float(var.tridiag_mineig(al, be))}
    out["config4_minimum_bisection"] = res
    var.close()
    # ---- config 5: the batch of 64 MaxCut instances (Gset G1–G9 + 55 G(800, 0.06)), rank 10, ptol = objtol = 1e-2 ----
    z = np.load(os.path.join(ROOT, "tests", "golden", "gset_G1_G9.npz"))
    graphs = [problems.graph_from_edges(int(z[f"G{k}_n"]), z[f"G{k}"]) for k in range(1, 10)]
    graphs += [problems.gnp_graph(800, 0.06, seed) for seed in range(10, 65)]
    conc = 16
    abi.warmup(conc)
    datas = [problems.maxcut_data(g) for g in graphs]     # the problem is the solver's input (exps/test.jl:166-176)
    kw = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=800.0)
    # two ways to drive the batch (sdplrplus.jl_amd/batch.py): in lockstep — every step of all 64 solves is one library
    # call, one launch with a workgroup per instance — or as `conc` independent driver threads.  One untimed pass first
    # (device code, the library's pools: a process that solves batch after batch is in this state), then the clock.
    walls, first, rows = {}, {}, None
    abi.warmup(64)                                        # a stream per handle a lockstep batch keeps alive
    for mode in ("lockstep", "threads"):
        t0 = time.perf_counter()
        batch.solve_local(datas, 0, 1, 10, concurrency=conc, lockstep=mode == "lockstep", **kw)
        first[mode] = time.perf_counter() - t0
        abi.device_synchronize()
        t0 = time.perf_counter()
        r_ = batch.solve_local(datas, 0, 1, 10, concurrency=conc, lockstep=mode == "lockstep", **kw)
        walls[mode] = time.perf_counter() - t0
        assert rows is None or (np.array_equal(rows[:, 1:4], r_[:, 1:4])), "lockstep and threaded drivers disagree"
        rows = r_
    wall = walls["lockstep"]
    gap = (rows[:, 1] - rows[:, 2]) / np.minimum(np.abs(rows[:, 1]), np.abs(rows[:, 2]))
    out["config5_batch64"] = {"workload": "64 MaxCut instances n=800 (Gset G1–G9 + 55 G(800,0.06)), rank 10, ptol=objtol=1e-2, one GPU",
                              "instances": 64, "driver": "lockstep (one launch per step for the whole batch)", "wall_s": wall,
                              "first_pass_wall_s": first["lockstep"],      # (device blocks of these sizes not yet in the library's pool)
                              "instances_per_s": 64 / wall,
                              "wall_s_driver_threads": walls["threads"], "driver_threads_in_flight": conc,
                              "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"),
                              "inner_iterations_total": int(rows[:, 3].sum()), "max_abs_relative_gap": float(np.max(np.abs(gap))),
                              "route": "resident (one launch per major iteration and per dual bound)",
                              "all_converged": bool(np.all(np.abs(gap) <= 1e-2))}
    # the resident kernel itself, one instance on one CU: device time per inner iteration against the bytes an iteration
    # requests from the XCD's L2 (DIR 9 streams in, STEP 11 in + 4 out, the SpMM R in + W out + the ELL lines; the
    # direction itself lives in LDS).  The bound is the CU's L2 REQUEST rate (MI355X_MICROARCH.md: 66–73 GB/s per CU at 8 bytes
    # per lane, twice that at the 16 bytes these phases ask for), not HBM: the state of an instance (≈ 0.8 MB) never leaves L2.
    one = sj.build_solver(abi, datas[0], 10, sj.BurerMonteiroConfig(seed=0, printlevel=0))
    nC, nB = datas[0].normC(), float(np.linalg.norm(datas[0].b))
    st = one.fg(nC, nB)
    st = one.inner_loop(nC, nB, True, True, False, 0.0, -1e300, 50, 0.0, *st)[:3]
    abi.device_synchronize()
    t0 = time.perf_counter()
    st = one.inner_loop(nC, nB, True, True, False, 0.0, -1e300, 400, 0.0, *st)[:3]
    abi.device_synchronize()
    us_it = 1e6 * (time.perf_counter() - t0) / 400
    dd = one.dims()
    Ns = 8.0 * dd["n"] * dd["r"]
    ell_bytes = 4.0 * (dd["nnzS"] - dd["n"]) + 8.0 * 4 * dd["n"]
    # a team of W workgroups per instance (k_resident.h, TEAM; SDPLR_HIP_TEAM, default 3): every member forms the whole
    # direction (9N each), the SpMM, the line search and STEP are shared out
    team = max(1, min(4, int(os.environ.get("SDPLR_HIP_TEAM", "3")))) if dd["n"] >= 128 else 1
    it_bytes = 9 * Ns * team + (2 * Ns + ell_bytes) + 15 * Ns
    one.close()
    out["config5_batch64"]["roofline"] = {
        "bound": f"l2-request-rate of the {team} CU(s) an instance runs on (resident route: a team of {team} workgroup(s) of one XCD per "
                 f"instance; {64 * team} of 256 CUs busy with the batch)",
        "kernel": "k_rs_team / k_rs_loop (one inner iteration = SEAM, DIR, SPMM, LSSUM, SOLVE, COMMIT, STEP inside one launch; "
                  "three team barriers per iteration)",
        "workgroups_per_instance": team,
        "us_per_iteration": us_it, "bytes_per_iteration": it_bytes, "achieved": it_bytes / us_it / 1e3, "peak": 134.0 * team,
        "unit": "GB/s for the team's CUs", "frac": it_bytes / us_it / 1e3 / (134.0 * team), "traffic": None}
    # ---- the reference's batch generator names three more problems (exps/gen_batch_test.jl:3); MinBisection (a rank-one
    # constraint) and CutNorm (2·800 vertices: per-row vectors through global memory) take the resident route too on
    # G1–G9, Lovász-θ (one constraint per edge) the multi-launch edge path ----
    for name, build in (("MinimumBisection", problems.minimum_bisection_data), ("CutNorm", problems.cutnorm_data),
                        ("LovaszTheta", problems.lovasz_theta_data)):
        ds = [build(g) for g in graphs[:9]]
        kw2 = dict(ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=1.0 if name == "LovaszTheta" else 800.0, maxtime=60.0)
        if name != "LovaszTheta":   # (Lovász-θ on these graphs takes ≈ 6e4 inner iterations per instance, ≈ 7 s for the nine: one pass)
            batch.solve_local(ds, 0, 1, 10, concurrency=conc, lockstep=True, **kw2)
        abi.device_synchronize()
        t0 = time.perf_counter()
        r_ = batch.solve_local(ds, 0, 1, 10, concurrency=conc, lockstep=True, **kw2)
        w_ = time.perf_counter() - t0
        gp = (r_[:, 1] - r_[:, 2]) / np.maximum(1e-300, np.minimum(np.abs(r_[:, 1]), np.abs(r_[:, 2])))
        out["batch_G1_G9_" + name] = {"workload": f"{name} on Gset G1–G9 (exps/gen_batch_test.jl:1-3), rank 10, ptol=objtol=1e-2, lockstep driver, one GPU",
                                      "instances": 9, "wall_s": w_, "inner_iterations_total": int(r_[:, 3].sum()),
                                      "route": "multi-launch edge path, one launch per kernel for the group (k_group.h)" if name == "LovaszTheta" else "resident",
                                      "max_abs_relative_gap": float(np.max(np.abs(gp)))}
    out["seconds_spent"] = time.perf_counter() - t_all
    return out


def reduce_over_ranks(dist, dt, obj, device):
    """max-over-ranks of the timed region, the per-rank objectives and per-rank seconds — the only collectives of the
    workload (RCCL over xGMI on the GPU node; gloo in tests/test_batch_gloo.py)."""
    if dist is None or not dist.is_initialized() or (
            dist.get_world_size() == 1 and not os.environ.get("SDPLR_BENCH_FORCE_DIST")
            and not os.environ.get("SDPLR_BENCH_FORCE_LAUNCH")):
        return dt, [obj], [dt]
    import torch
    world = dist.get_world_size()
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    o = torch.tensor([obj, dt], dtype=torch.float64, device=device)      # (objective, this rank's own seconds: stragglers show)
    gathered = [torch.zeros_like(o) for _ in range(world)]
    dist.all_gather(gathered, o)
    return float(t.item()), [float(x[0].item()) for x in gathered], [float(x[1].item()) for x in gathered]


def load_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary (profiles/), or null."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(kernel)
    except (OSError, ValueError):
        return None


if __name__ == "__main__":
    sys.exit(main())
