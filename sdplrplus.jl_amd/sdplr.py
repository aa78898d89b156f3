"""Host-side solver control flow above the C ABI: ``sdplr`` and ``_sdplr``.

This is the caller of the hot path (src/sdplr.jl:91-449), restated in Python so that the device
library can be driven end to end without Julia (SURVEY.md §8b "Who calls it here").  All
factor-sized arithmetic happens behind ``DeviceSolver``; this file only holds the scalar schedule:
tolerances, σ/λ updates, the duality-gap test and rank doubling.
"""
from __future__ import annotations

import math
import sys
import time
from typing import Optional

import numpy as np

from . import cabi
from .cabi import CABI, DeviceSolver, load_hip
from .preprocess import preprocess_sparsecons
from .structs import BurerMonteiroConfig, SDPData, barvinok_pataki

EPS = float(np.finfo(np.float64).eps)

_ASCII_ALIASES = {"sigma_0": "σ_0", "sigmafac": "σfac"}


def _rng(config: BurerMonteiroConfig) -> np.random.Generator:
    if not hasattr(config, "_rng") or config._rng is None:
        config._rng = np.random.Generator(np.random.PCG64(config.seed))
    return config._rng


def initial_point(data: SDPData, r: int, config: BurerMonteiroConfig):
    """Rt0, λ0 of ``SolverVars(data, r, config)`` (src/structs.jl:225-240): uniform(−1,1) factor and
    λ = 0, or the user's ``init_func(data, r, init_args...)`` clipped to λ_ub."""
    λ_ub = np.where(data.constraint_types, 0.0, np.inf)
    if config.init_func is not None:
        Rt0, λ0 = config.init_func(data, r, *config.init_args)
        Rt0 = np.ascontiguousarray(Rt0, dtype=np.float64)
        if Rt0.shape == (r, data.n) and r != data.n:
            Rt0 = np.ascontiguousarray(Rt0.T)  # accept Julia-shaped r×n
        λ0 = np.minimum(np.asarray(λ0, dtype=np.float64), λ_ub)
    else:
        Rt0 = 2.0 * _rng(config).random((data.n, r)) - 1.0
        λ0 = np.zeros(data.m)
    return Rt0, λ0, λ_ub


def build_solver(abi: CABI, data: SDPData, r: int, config: BurerMonteiroConfig,
                 layout=None, native_preprocess: bool = True) -> DeviceSolver:
    """SolverVars + SolverAuxiliary + lbfgs_init behind one handle
    (src/sdplr.jl:114-123,163; src/structs.jl:242-263,296-361).  ``preprocess_sparsecons`` (src/preprocess.jl:24-169)
    runs inside the library (``set_sparse_coo``) unless a ready ``layout`` is handed in or ``native_preprocess`` is
    off, in which case the host-side mirror (``preprocess.py``) builds the arrays and ``set_sparse`` takes them."""
    s = DeviceSolver(abi, data.n, data.m, r, config.numlbfgsvecs)
    if data.sparse.n_matrices > 0:
        if layout is None and native_preprocess:
            s.set_sparse_coo(data.sparse)
            s.layout = None
        else:
            lay = layout if layout is not None else preprocess_sparsecons(data.sparse)
            s.set_sparse(lay)
            s.layout = lay
    for gid, A in data.lowrank:
        s.add_symlowrank(gid, A)
    s.finalize()
    _load_point(s, data, r, config)
    return s


def _load_point(s: DeviceSolver, data: SDPData, r: int, config: BurerMonteiroConfig):
    Rt0, λ0, λ_ub = initial_point(data, r, config)
    s.set_vec(cabi.V_B, data.b)
    s.set_vec(cabi.V_LAMBDA_UB, λ_ub)
    s.set_vec(cabi.V_PV_LB, np.where(np.isinf(λ_ub), -np.inf, 0.0))  # src/structs.jl:247
    s.set_vec(cabi.V_LAMBDA, λ0)
    s.set_factor(cabi.F_RT, Rt0)
    s.set_scalar(cabi.S_SIGMA, config.σ_0)
    s.Rt0, s.λ0 = Rt0, λ0



def SDP_S_eigval(var: DeviceSolver, nevs: int = 1, preprocessed: bool = False, *, which: str = "SA",
                 ncv: Optional[int] = None, tol: float = 0.0, maxiter: int = 1000000, v0=None) -> np.ndarray:
    """``SDP_S_eigval`` (src/coreop.jl:351-374): the ``nevs`` smallest eigenvalues of S.  The reference hands
    x ↦ S·x + x to GenericArpack.symeigs (implicitly restarted Lanczos) and cancels the shift; here the same kind of
    solver — thick-restart Lanczos with full re-orthogonalisation — runs on the device behind
    ``sdplr_hip_S_eigval``: basis vectors and operator (``𝒜t!(y, aux, x, var)``, src/coreop.jl:281-300) stay in HBM.
    ``preprocessed=False`` refreshes S from the current y first (the reference's branch :358-360 calls
    𝒜t_preprocess! with swapped arguments and has no method; every caller passes ``true``)."""
    if not preprocessed:
        var.At_preprocess()
    n = var.n
    ncv = min(ncv if ncv is not None else min(100, n), n)
    vals, _matvecs, nconv = var.S_eigval(nevs, which, ncv, tol, min(int(maxiter), 100000), v0)
    if nconv < nevs:
        print(f"Warning: SDP_S_eigval: {nconv} of {nevs} eigenvalues converged.", file=sys.stderr)
    return np.sort(vals)


def DIMACS_errors(data: SDPData, var: DeviceSolver) -> np.ndarray:
    """The six DIMACS errors (src/coreop.jl:417-453) with X = RRᵀ, Z = C − 𝒜*(λ)."""
    m = data.m
    pv_raw = var.primal_vio_raw
    normb = float(np.linalg.norm(data.b))
    err1 = float(np.linalg.norm(pv_raw[:m])) / (1.0 + normb)
    err2 = err3 = 0.0                                  # X = YYᵀ ⪰ 0, Z = C − 𝒜*(y) by construction (:433)
    λ = var.λ
    var.y = np.concatenate([-λ, [1.0]])                # copy2y_λ! (:238-246)
    var.At_preprocess()                                # (:436)
    ev = SDP_S_eigval(var, 1, True, which="SA", ncv=min(100, data.n), maxiter=1000000)
    normC = data.normC()
    err4 = max(0.0, -float(ev[0])) / (1.0 + normC)
    obj = var.obj
    λb = float(λ @ data.b)
    err5 = (obj - λb) / (1.0 + abs(obj) + abs(λb))
    var.At_left(cabi.F_SCRATCH, cabi.F_RT)             # Rt·S (:449) into a scratch slot: one device dot, no download
    err6 = var.factor_dot(cabi.F_RT, cabi.F_SCRATCH) / (1.0 + abs(obj) + abs(λb))
    return np.array([err1, err2, err3, err4, err5, err6])


def _print_row(config, majoriter, localiter, iter_, L, obj, σ, gtol, ptol, gnorm, pnorm, gap, dobj):
    """printintermediate (src/myprint.jl:17-58), one plain line."""
    if config.printlevel > 0:
        print(f"{config.dataset:>8s} T={majoriter:<4d} it={localiter:<6d} tot={iter_:<8d} "
              f"L={L: .6e} pobj={obj: .6e} σ={σ:.2e} η={gtol:.2e} ω={ptol:.2e} |g|={gnorm:.3e} "
              f"|pinf|={pnorm:.3e} gap={gap:.3e} dobj={dobj: .6e}", file=sys.stdout, flush=True)


# The four device-side steps of _sdplr that a driver may serve for MANY instances in one library call (include/sdplr_hip.h,
# "lockstep"): the stepper below yields them as requests instead of calling them, so that the same restatement of the
# control flow runs one solve (``_sdplr``: every request is served at once on the instance's own handle) or a batch side by
# side (``batch.solve_lockstep``: the same request of all instances is ONE call, one kernel launch for the whole batch).
REQ_FG, REQ_MAJOR, REQ_INNER, REQ_DUAL = "fg", "major_iteration", "inner_loop", "dual_obj"


def serve(var: DeviceSolver, req: tuple) -> tuple:
    """One request of the stepper on one handle, through the single-instance entry points."""
    kind, args = req[0], req[1:]
    if kind == REQ_FG:
        return var.fg(*args) + (var.obj,)
    if kind == REQ_MAJOR:
        return var.major_iteration(*args) + (var.obj,)
    if kind == REQ_INNER:
        return var.inner_loop(*args) + (var.obj,)
    if kind == REQ_DUAL:
        return var.dual_obj(*args) + (None,)      # (third: var.y if the driver already has it — the batched call does)
    raise ValueError(kind)


def _sdplr(data: SDPData, var: DeviceSolver, config: BurerMonteiroConfig,
           native_inner_loop: bool = True) -> dict:
    """``_sdplr`` (src/sdplr.jl:140-449) on one handle."""
    steps = sdplr_steps(data, var, config, native_inner_loop=native_inner_loop)
    try:
        req = next(steps)
        while True:
            req = steps.send(serve(var, req))
    except StopIteration as done:
        return done.value


def sdplr_steps(data: SDPData, var: DeviceSolver, config: BurerMonteiroConfig, native_inner_loop: bool = True):
    """``_sdplr`` (src/sdplr.jl:140-449) as a generator: yields (REQ_*, args…), receives what ``serve`` returns, and
    returns the result Dict.  ``var`` plays the role of (var, aux, lbfgshis, dirt)."""
    n, m = data.n, data.m
    starttime = time.time()
    lastprint = starttime
    dual_time = 0.0
    Rt0, λ0 = var.Rt0, var.λ0

    normb = float(np.linalg.norm(data.b))       # :159
    normC = data.normC()                        # :160
    grel = config.gtol_mode == "relative"
    prel = config.ptol_mode == "relative"

    σ = σ_now = var.σ                           # (σ_now: var.σ[] as the device holds it — only this function writes it)
    cur_gtol = max(1.0 / σ, config.gtol)        # :165-169
    cur_ptol = max(1.0 / σ ** 0.1, config.ptol)
    # The tail of a major iteration — λ update or σ increase (:358-369), lbfgs_clear! (:384), fg! (:389) — is device work
    # with no host decision in between, and so is the while loop it feeds: with the native loop the four travel as ONE
    # call (``major_iteration``; one kernel launch on small instances).  `pending` holds a tail not yet sent.  The fg! of
    # :170 and the first pass of the loop are such a call too: no λ update, σ as it stands, and lbfgs_clear! on the fresh
    # history of lbfgs_init (src/sdplr.jl:163) changes nothing.
    pending = (False, σ) if native_inner_loop else None
    L_val = grad_norm = primal_vio_norm = obj = math.nan
    if pending is None:
        L_val, grad_norm, primal_vio_norm, obj = yield (REQ_FG, normC, normb, grel, prel)  # :170

    iter_ = 0
    majoriter = 0
    localiter = 0
    use_armijo = data.has_inequalities          # :176
    rankupd_tol_cnt = config.rankupd_tol
    duality_gap = 1e20
    min_duality_gap = 1e20
    max_dual_value = -1e20
    best_λ = np.array(λ0, dtype=np.float64, copy=True)     # (= var.λ: the point has just been loaded)
    rng = _rng(config)

    schedule = []   # per major iteration: (majoriter, inner iterations, σ, η, ω, rank) — what printintermediate shows (src/myprint.jl:17-58)
    for _ in range(config.maxmajoriter):        # :185
        majoriter += 1
        localiter = 0
        if native_inner_loop:
            budget = max(config.maxiter + 1 - iter_, 1)
            tleft = config.maxtime - (time.time() - starttime)
            if pending is not None:
                L_val, grad_norm, primal_vio_norm, _α, localiter, _why, obj = yield (
                    REQ_MAJOR, normC, normb, grel, prel, use_armijo, pending[0], pending[1], cur_gtol,
                    config.fprec * EPS, budget, max(tleft, 1e-9))
                σ_now = pending[1]
                pending = None
                iter_ += localiter
            elif grad_norm > cur_gtol:
                L_val, grad_norm, primal_vio_norm, _α, localiter, _why, obj = yield (
                    REQ_INNER, normC, normb, grel, prel, use_armijo, cur_gtol, config.fprec * EPS, budget,
                    max(tleft, 1e-9), L_val, grad_norm, primal_vio_norm)
                iter_ += localiter
        else:
            while grad_norm > cur_gtol:         # :190
                localiter += 1
                iter_ += 1
                descent = var.lbfgs_dir(negate=True)            # :197, :201
                if math.isnan(descent) or descent >= 0:         # :202-205
                    var.descent_fallback()
                lastval = L_val
                if use_armijo:                                  # :210-214
                    α, L_val = var.linesearch_armijo(1.0)
                else:
                    α, L_val = var.linesearch(1.0)
                var.axpy_R(α)                                   # :219
                var.g()                                         # :221
                grad_norm, primal_vio_norm = var.norms(normC, normb, grel, prel)  # :224-234
                rel_delta = (lastval - L_val) / max(1.0, abs(L_val), abs(lastval))
                if rel_delta < config.fprec * EPS:              # :238-241
                    break
                if config.numlbfgsvecs > 0:                     # :244-246
                    var.lbfgs_update(α)
                current_time = time.time()
                if current_time - lastprint >= config.printfreq:
                    lastprint = current_time
                    _print_row(config, majoriter, localiter, iter_, L_val, var.obj, var.σ,
                               cur_gtol, cur_ptol, grad_norm, primal_vio_norm, min_duality_gap,
                               max_dual_value)
                if current_time - starttime > config.maxtime or iter_ > config.maxiter:
                    break                                       # :272-277
            obj = var.obj

        current_time = time.time()
        _print_row(config, majoriter, localiter, iter_, L_val, obj, σ_now, cur_gtol, cur_ptol,
                   grad_norm, primal_vio_norm, min_duality_gap, max_dual_value)
        schedule.append((majoriter, int(localiter), float(σ_now), float(cur_gtol), float(cur_ptol), int(var.r)))
        lastprint = current_time
        if current_time - starttime > config.maxtime:           # :298-301
            print("Warning: Time limit exceeded. Stop optimizing.", file=sys.stderr)
            break
        if iter_ > config.maxiter:                              # :303-306
            print("Warning: Iteration limit exceeded. Stop optimizing.", file=sys.stderr)
            break

        rank_double = False
        σ = σ_now
        if primal_vio_norm <= cur_ptol:                         # :310
            t0 = time.time()
            v0 = rng.standard_normal(n)                         # replaces randn, coreop.jl:473
            y_now = None
            if config.eigval_highprecision:                     # coreop.jl:389-400
                var.y = np.concatenate([-np.minimum(var.λ_ub, var.λ - σ_now * var.primal_vio_raw[:m]), [1.0]])
                var.At_preprocess()
                ev = SDP_S_eigval(var, 1, True, which="SA", ncv=min(100, n), tol=1e-6, maxiter=1000000,
                                  v0=v0)[0]
                dual_value = float(-(var.y[:m] @ data.b) + config.prior_trace_bound * min(ev, 0.0))
            else:
                dual_value, _, y_now = yield (REQ_DUAL, config.prior_trace_bound, iter_, v0)     # :314
            if dual_value > max_dual_value:                     # :324-327
                best_λ = -(y_now if y_now is not None else var.y)
                max_dual_value = dual_value
            if config.objtol_mode == "relative":                # :328-332
                denom = min(abs(obj), abs(max_dual_value))
                duality_gap = (obj - max_dual_value) / denom if denom != 0 else math.inf
            else:
                duality_gap = obj - max_dual_value
            dual_time += time.time() - t0
            if config.printlevel > 0:
                print(f"var.obj = {obj}  max_dual_value = {max_dual_value}  "
                      f"duality_gap = {duality_gap}", flush=True)
            if primal_vio_norm <= config.ptol:                  # :335
                if config.objtol == math.inf:
                    break
                if duality_gap <= config.objtol:
                    min_duality_gap = min(min_duality_gap, duality_gap)
                    break
                if min_duality_gap - duality_gap < config.objtol:   # :347-351
                    rankupd_tol_cnt -= 1
                else:
                    rankupd_tol_cnt = config.rankupd_tol
                min_duality_gap = min(min_duality_gap, duality_gap)
                if rankupd_tol_cnt == 0:
                    rank_double = True
            upd_λ = True                                        # :358-362
            cur_ptol = cur_ptol / σ ** 0.9                      # :363-364
            cur_gtol = cur_gtol / σ
        else:
            upd_λ = False
            σ = σ * config.σfac                                 # :366-369
            cur_ptol = 1 / σ ** 0.1
            cur_gtol = 1 / σ
        fuse_tail = native_inner_loop and not rank_double and majoriter < config.maxmajoriter
        if not fuse_tail:
            if upd_λ:
                var.update_lambda()
            else:
                var.σ = σ
                σ_now = σ

        if rank_double:                                         # :373-382
            newr = min(barvinok_pataki(data.n, data.m), var.r * 2)    # coreop.jl:518-526
            var.reset_rank(newr)
            _load_point(var, data, newr, config)
            σ = σ_now = var.σ
            cur_ptol = 1 / σ ** 0.1
            cur_gtol = 1 / σ
            min_duality_gap = 1e20
            max_dual_value = -1e20
            rankupd_tol_cnt = config.rankupd_tol
            if config.printlevel > 0:
                print(f"rank doubled, newrank is {var.r}.", flush=True)
        elif not fuse_tail:
            var.lbfgs_clear()                                   # :384

        cur_ptol = max(cur_ptol, config.ptol)                   # :387-389
        cur_gtol = max(cur_gtol, config.gtol)
        if fuse_tail:
            pending = (upd_λ, σ)                                # sent with the next pass of the while loop
        else:
            L_val, grad_norm, primal_vio_norm, obj = yield (REQ_FG, normC, normb, grel, prel)
        if majoriter == config.maxmajoriter:
            print("Warning: Major iteration limit exceeded. Stop optimizing.", file=sys.stderr)

    L_val, grad_norm, primal_vio_norm, obj = yield (REQ_FG, normC, normb, grel, prel)   # :396
    _print_row(config, majoriter, -1, iter_, L_val, obj, σ_now, cur_gtol, cur_ptol, grad_norm,
               primal_vio_norm, min_duality_gap, max_dual_value)
    totaltime = time.time() - starttime
    DIMACS_errs = DIMACS_errors(data, var) if config.eval_DIMACS_errs else np.zeros(6)   # :419-425
    Rt = var.Rt
    return {                                                    # :426-448
        "Rt": Rt, "lambda": best_λ, "Rt0": Rt0, "lambda0": λ0, "sigma": σ_now,
        "grad_norm": grad_norm, "primal_vio": primal_vio_norm, "obj": obj,
        "max_dual_value": max_dual_value, "min_duality_gap": min_duality_gap,
        "totaltime": totaltime, "dual_time": dual_time, "primaltime": totaltime - dual_time,
        "iter": iter_, "majoriter": majoriter, "DIMACS_errs": DIMACS_errs, "ptol": config.ptol,
        "objtol": config.objtol, "fprec": config.fprec, "rankupd_tol": config.rankupd_tol,
        "r": Rt.shape[1],
        "schedule": schedule,                                   # not in the reference's Dict: the per-major-iteration log
    }


def sdplr(C=None, As=None, b=None, r: int = 1, *, constraint_types=None,
          config: Optional[BurerMonteiroConfig] = None, data: Optional[SDPData] = None,
          abi: Optional[CABI] = None, native_inner_loop: bool = True, **kwargs) -> dict:
    """``sdplr(C, As, b, r; kwargs...)`` (src/sdplr.jl:91-138).

    Runs on the MI355X library (``load_hip()``) unless an ``abi`` is injected (the test-suite
    injects the CPU oracle's ABI to pin the control flow; the product never does).  ``data`` may
    replace ``(C, As, b)`` with an already-built ``SDPData`` (batched builders)."""
    config = config if config is not None else BurerMonteiroConfig()
    for key, value in kwargs.items():                           # :102-108
        key = _ASCII_ALIASES.get(key, key)
        if hasattr(config, key):
            setattr(config, key, value)
        else:
            print(f"Error: Unrecognized keyword argument {key}", file=sys.stderr)
    t0 = time.time()
    if data is None:
        data = SDPData(C, As, b, constraint_types)
    abi = abi if abi is not None else load_hip()
    var = build_solver(abi, data, int(r), config)
    preprocess_dt = time.time() - t0
    try:
        ans = _sdplr(data, var, config, native_inner_loop=native_inner_loop)
    finally:
        var.close()
    ans["preprocess_time"] = preprocess_dt                      # :130-131
    ans["totaltime"] += preprocess_dt
    return ans
