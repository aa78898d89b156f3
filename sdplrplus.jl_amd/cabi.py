"""ctypes binding of the C ABI declared in include/sdplr_hip.h.

``CABI(lib_path, prefix)`` binds every entry point ``<prefix><name>``.  The product loads
``lib/libsdplr_hip.so`` with prefix ``sdplr_hip_`` (``load_hip()``); it raises if the library has
not been built — there is no CPU fallback in this package.  ``DeviceSolver`` is the host-side
mirror of the reference's ``(var::SolverVars, aux::SolverAuxiliary, lbfgshis)`` triple: one method
per reference operator, same names, same argument meaning, same error behaviour.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

OK = 0
ERR_INVALID_ARG = -1
ERR_HIP = -2
ERR_NOT_DESCENT = -3
ERR_STATE = -4
ERR_NO_DEVICE = -5
ERR_ALLOC = -6
ERR_UNSERVED = -7   # batch calls: the call returned before this item was served
MAJOR_RESUME = 2    # update_lambda of major_iteration: continue a capped loop (include/sdplr_hip.h)

F_RT, F_GT, F_DIRT, F_LBFGS_S, F_LBFGS_Y, F_SCRATCH = 0, 1, 2, 100, 200, 300
(V_LAMBDA, V_LAMBDA_UB, V_B, V_Y, V_PV_RAW, V_PV_LB, V_PV, V_A_RD, V_A_DD, V_LBFGS_RHO, V_LBFGS_A,
 V_UVT, V_TRIU_S_NZVAL, V_S_NZVAL, V_SCRATCH) = range(15)
STAT_NAMES = ("graph_captures", "graph_capture_failures", "graph_capture_skipped", "graph_batches",
              "eager_batches", "lanczos_graph_replays", "lanczos_eager_rounds", "inner_iterations",
              "resident_loops", "resident_lanczos", "resident_fg", "resident_shared_launches", "p_less_loops",
              "group_launch_loops", "ring_history_loops", "ring_materializations")
S_SIGMA, S_OBJ, S_LBFGS_LATEST = 0, 1, 2

_i32, _i64, _f64, _vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p
_pi32, _pi64, _pf64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)



class FgItem(C.Structure):            # sdplr_hip_fg_item
    _fields_ = [("s", _vp), ("normC", _f64), ("normb", _f64), ("gtol_relative", _i32), ("ptol_relative", _i32),
                ("lagrangian", _f64), ("grad_norm", _f64), ("primal_vio_norm", _f64), ("obj", _f64), ("status", _i32)]


class MajorItem(C.Structure):         # sdplr_hip_major_item
    _fields_ = [("s", _vp), ("normC", _f64), ("normb", _f64), ("gtol_relative", _i32), ("ptol_relative", _i32),
                ("use_armijo", _i32), ("update_lambda", _i32), ("sigma", _f64), ("cur_gtol", _f64), ("fprec_eps", _f64),
                ("max_local_iters", _i64), ("time_budget_s", _f64), ("lagrangian", _f64), ("grad_norm", _f64),
                ("primal_vio_norm", _f64), ("last_alpha", _f64), ("obj", _f64), ("iters_done", _i64),
                ("exit_reason", _i32), ("status", _i32)]


class DualItem(C.Structure):          # sdplr_hip_dual_item
    _fields_ = [("s", _vp), ("trace_bound", _f64), ("iter", _i64), ("v0", _pf64), ("y_out", _pf64),
                ("dual_value", _f64), ("mineig", _f64), ("status", _i32)]


# name → argtypes (restype is int32 unless listed in _RESTYPES); this table IS the header, and
# tests/test_cabi_symbols.py checks it against include/sdplr_hip.h.
SIGNATURES = {
    "device_count": [_pi32],
    "set_device": [_i32],
    "last_error": [_vp],
    "version": [],
    "device_synchronize": [],
    "warmup": [_i32],
    "trim_pools": [],
    "create": [_i64, _i64, _i64, _i64, C.POINTER(_vp)],
    "set_sparse": [_vp, _i64, _i64, _pi64, _pi64, _pf64, _pf64, _pi64, _i64, _pi64, _pi64, _i64,
                   _pi64, _pi64, _pi64],
    "set_sparse_coo": [_vp, _i64, _i64, _pi64, _pi64, _pi64, _pf64, _pi64],
    "get_layout": [_vp, _i32, _pi64, _pf64, _i64, _pi64],
    "add_symlowrank": [_vp, _i64, _i64, _i64, _pf64, _pf64],
    "finalize": [_vp],
    "destroy": [_vp],
    "reset_rank": [_vp, _i64],
    "set_factor": [_vp, _i32, _pf64],
    "get_factor": [_vp, _i32, _pf64],
    "set_vec": [_vp, _i32, _pf64, _i64],
    "get_vec": [_vp, _i32, _pf64, _i64],
    "set_scalar": [_vp, _i32, _f64],
    "get_scalar": [_vp, _i32, _pf64],
    "get_dims": [_vp, _pi64, _pi64, _pi64, _pi64, _pi64, _pi64, _pi64],
    "A": [_vp, _i32, _i32, _i32],
    "At_preprocess": [_vp],
    "At_left": [_vp, _i32, _i32],
    "At_right": [_vp, _pf64, _pf64, _i64],
    "At_right_device": [_vp, _vp, _vp, _i64],
    "get_stats": [_vp, _pi64, _i32, _pi32],
    "S_eigval": [_vp, _i64, _i32, _i64, _f64, _i64, _pf64, _pf64, _pi64, _pi64],
    "factor_dot": [_vp, _i32, _i32, _pf64],
    "f": [_vp, _pf64],
    "g": [_vp],
    "fg": [_vp, _f64, _f64, _i32, _i32, _pf64, _pf64, _pf64],
    "lbfgs_clear": [_vp],
    "lbfgs_dir": [_vp, _i32, _pf64],
    "descent_fallback": [_vp],
    "lbfgs_update": [_vp, _f64],
    "linesearch": [_vp, _f64, _pf64, _pf64],
    "linesearch_armijo": [_vp, _f64, _pf64, _pf64],
    "axpy_R": [_vp, _f64],
    "norms": [_vp, _f64, _f64, _i32, _i32, _pf64, _pf64],
    "update_lambda": [_vp],
    "inner_loop": [_vp, _f64, _f64, _i32, _i32, _i32, _f64, _f64, _i64, _f64, _pf64, _pf64, _pf64,
                   _pf64, _pi64, _pi32],
    "major_iteration": [_vp, _f64, _f64, _i32, _i32, _i32, _i32, _f64, _f64, _f64, _i64, _f64, _pf64, _pf64, _pf64,
                        _pf64, _pi64, _pi32],
    "batch_fg": [_i32, C.POINTER(FgItem)],
    "batch_major_iteration": [_i32, C.POINTER(MajorItem)],
    "batch_dual_obj": [_i32, C.POINTER(DualItem)],
    "lanczos": [_vp, _i64, _pf64, _pf64, _pf64, _pi64],
    "tridiag_mineig": [_pf64, _pf64, _i64, _pf64],
    "approx_mineigval_lanczos": [_vp, _i64, _pf64, _pf64],
    "dual_obj": [_vp, _f64, _i64, _pf64, _pf64, _pf64],
    "profile_enable": [_vp, _i32],
    "profile_filter": [_vp, C.c_char_p],
    "profile_count": [_vp, _pi32],
    "profile_get": [_vp, _i32, C.c_char_p, _i32, _pi64, _pf64],
}
_RESTYPES = {"last_error": C.c_char_p, "version": C.c_char_p}


class SdplrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class CABI:
    """All entry points of one shared library exporting the sdplr C ABI under ``prefix``."""

    def __init__(self, lib_path: str, prefix: str):
        if not os.path.exists(lib_path):
            raise FileNotFoundError(
                f"{lib_path} is missing — build it first (python -c 'import __graft_entry__ as g; "
                "g.build()'); this package has no CPU fallback")
        self.path = lib_path
        self.prefix = prefix
        self.lib = C.CDLL(lib_path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(self.lib, prefix + name)  # AttributeError ⇒ symbol missing: fail loudly
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, _i32)
            setattr(self, name, fn)

    def version_string(self) -> str:
        return self.version().decode()


_HIP: Optional[CABI] = None


def hip_library_path() -> str:
    """lib/libsdplr_hip.so of this package; SDPLR_HIP_LIBRARY names another build of the same HIP library
    (kernel experiments) — there is no non-HIP backend to point it at."""
    return os.environ.get("SDPLR_HIP_LIBRARY") or os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "lib", "libsdplr_hip.so")


def load_hip() -> CABI:
    """The product's only backend.  Raises FileNotFoundError if the HIP library is not built."""
    global _HIP
    if _HIP is None:
        _HIP = CABI(hip_library_path(), "sdplr_hip_")
    return _HIP


def _pd(a: np.ndarray):
    return a.ctypes.data_as(_pf64)


def _pl(a: np.ndarray):
    return a.ctypes.data_as(_pi64)


def _f64c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64)


class DeviceSolver:
    """One SDP instance resident behind the C ABI.

    Factor arrays cross the boundary as numpy arrays of shape (n, r), C-contiguous — byte-for-byte
    the reference's r×n column-major ``Rt`` (src/structs.jl:195,236).
    """

    def __init__(self, abi: CABI, n: int, m: int, r: int, numlbfgsvecs: int):
        self.abi = abi
        self.n, self.m, self.r, self.h = int(n), int(m), int(r), int(numlbfgsvecs)
        self._h = _vp()
        rc = abi.create(self.n, self.m, self.r, self.h, C.byref(self._h))
        if rc != OK:
            raise SdplrError(rc, (abi.last_error(None) or b"create failed").decode())
        self._finalized = False

    # -- plumbing ---------------------------------------------------------------------------------
    def _ck(self, rc: int):
        if rc != OK:
            msg = self.abi.last_error(self._h)
            raise SdplrError(rc, msg.decode() if msg else "error")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.abi.destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- construction -----------------------------------------------------------------------------
    def set_sparse(self, lay, index_base: int = 0):
        a = [_i64c(lay.matptr), _i64c(lay.nzind), _f64c(lay.nzval_one), _f64c(lay.nzval_two),
             _i64c(lay.global_inds), _i64c(lay.triu_colptr), _i64c(lay.triu_rowval),
             _i64c(lay.full_colptr), _i64c(lay.full_rowval), _i64c(lay.mappedto_triu)]
        self._ck(self.abi.set_sparse(self._h, index_base, int(lay.n_sparse), _pl(a[0]), _pl(a[1]),
                                     _pd(a[2]), _pd(a[3]), _pl(a[4]), int(a[6].size), _pl(a[5]),
                                     _pl(a[6]), int(a[8].size), _pl(a[7]), _pl(a[8]), _pl(a[9])))

    def set_sparse_coo(self, batch, index_base: int = 0):
        """preprocess_sparsecons (src/preprocess.jl:24-169) inside the library, from the batched COO form of the sparse
        matrices (``structs.SparseBatch``: findnz order, both triangles)."""
        a = [_i64c(batch.ent_ptr), _i64c(batch.I), _i64c(batch.J), _f64c(batch.V), _i64c(batch.global_inds)]
        self._ck(self.abi.set_sparse_coo(self._h, index_base, int(batch.n_matrices), _pl(a[0]), _pl(a[1]), _pl(a[2]),
                                         _pd(a[3]), _pl(a[4])))

    def get_layout(self):
        """The aggregated layout the library holds between set_sparse[_coo] and finalize → AggregatedLayout (0-based)."""
        from .preprocess import AggregatedLayout
        out = []
        for which in range(10):
            L = C.c_int64(0)
            self._ck(self.abi.get_layout(self._h, which, None, None, 0, C.byref(L)))
            if which < 8:
                buf = np.zeros(max(L.value, 1), dtype=np.int64)
                self._ck(self.abi.get_layout(self._h, which, _pl(buf), None, L.value, C.byref(L)))
            else:
                buf = np.zeros(max(L.value, 1), dtype=np.float64)
                self._ck(self.abi.get_layout(self._h, which, None, _pd(buf), L.value, C.byref(L)))
            out.append(buf[: L.value])
        matptr, nzind, gids, tcp, trv, fcp, frv, mapped, one, two = out
        return AggregatedLayout(self.n, int(gids.size), matptr, nzind, one, two, gids, tcp, trv, fcp, frv, mapped)

    def add_symlowrank(self, global_ind: int, A, index_base: int = 0):
        B = np.asfortranarray(A.B, dtype=np.float64)  # n×s column-major
        D = _f64c(A.D)
        self._ck(self.abi.add_symlowrank(self._h, index_base, int(global_ind), int(D.size),
                                         B.ctypes.data_as(_pf64), _pd(D)))

    def finalize(self):
        self._ck(self.abi.finalize(self._h))
        self._finalized = True

    def reset_rank(self, new_r: int):
        self._ck(self.abi.reset_rank(self._h, int(new_r)))
        self.r = int(new_r)

    def dims(self) -> dict:
        v = [C.c_int64() for _ in range(7)]
        self._ck(self.abi.get_dims(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("n", "m", "r", "h", "nnzT", "nnzS", "nnzAgg"), [int(x.value) for x in v]))

    # -- state ------------------------------------------------------------------------------------
    def set_factor(self, slot: int, Rt: np.ndarray):
        Rt = _f64c(Rt)
        if Rt.shape != (self.n, self.r):
            raise ValueError(f"factor must have shape (n, r) = ({self.n}, {self.r}), got {Rt.shape}")
        self._ck(self.abi.set_factor(self._h, slot, _pd(Rt)))

    def get_factor(self, slot: int) -> np.ndarray:
        out = np.empty((self.n, self.r), dtype=np.float64)
        self._ck(self.abi.get_factor(self._h, slot, _pd(out)))
        return out

    def _veclen(self, which: int) -> int:
        d = self.dims()
        m = self.m
        return {V_LAMBDA: m, V_LAMBDA_UB: m, V_B: m, V_Y: m + 1, V_PV_RAW: m + 1, V_PV_LB: m,
                V_PV: m, V_A_RD: m + 1, V_A_DD: m + 1, V_LBFGS_RHO: self.h, V_LBFGS_A: self.h,
                V_UVT: d["nnzT"], V_TRIU_S_NZVAL: d["nnzT"], V_S_NZVAL: d["nnzS"], V_SCRATCH: m + 1}[which]

    def set_vec(self, which: int, v):
        v = _f64c(v)
        self._ck(self.abi.set_vec(self._h, which, _pd(v), int(v.size)))

    def get_vec(self, which: int) -> np.ndarray:
        out = np.empty(self._veclen(which), dtype=np.float64)
        self._ck(self.abi.get_vec(self._h, which, _pd(out), int(out.size)))
        return out

    def set_scalar(self, which: int, value: float):
        self._ck(self.abi.set_scalar(self._h, which, float(value)))

    def get_scalar(self, which: int) -> float:
        v = C.c_double()
        self._ck(self.abi.get_scalar(self._h, which, C.byref(v)))
        return float(v.value)

    # convenience views named as in SolverVars (src/structs.jl:194-223)
    Rt = property(lambda s: s.get_factor(F_RT), lambda s, v: s.set_factor(F_RT, v))
    Gt = property(lambda s: s.get_factor(F_GT), lambda s, v: s.set_factor(F_GT, v))
    dirt = property(lambda s: s.get_factor(F_DIRT), lambda s, v: s.set_factor(F_DIRT, v))
    λ = property(lambda s: s.get_vec(V_LAMBDA), lambda s, v: s.set_vec(V_LAMBDA, v))
    λ_ub = property(lambda s: s.get_vec(V_LAMBDA_UB), lambda s, v: s.set_vec(V_LAMBDA_UB, v))
    b = property(lambda s: s.get_vec(V_B), lambda s, v: s.set_vec(V_B, v))
    y = property(lambda s: s.get_vec(V_Y), lambda s, v: s.set_vec(V_Y, v))
    primal_vio_raw = property(lambda s: s.get_vec(V_PV_RAW), lambda s, v: s.set_vec(V_PV_RAW, v))
    primal_vio_lb = property(lambda s: s.get_vec(V_PV_LB), lambda s, v: s.set_vec(V_PV_LB, v))
    primal_vio = property(lambda s: s.get_vec(V_PV), lambda s, v: s.set_vec(V_PV, v))
    A_RD = property(lambda s: s.get_vec(V_A_RD), lambda s, v: s.set_vec(V_A_RD, v))
    A_DD = property(lambda s: s.get_vec(V_A_DD), lambda s, v: s.set_vec(V_A_DD, v))
    σ = property(lambda s: s.get_scalar(S_SIGMA), lambda s, v: s.set_scalar(S_SIGMA, v))
    obj = property(lambda s: s.get_scalar(S_OBJ), lambda s, v: s.set_scalar(S_OBJ, v))

    # -- operators (names follow the reference; 𝒜 → A, 𝒜t → At) ----------------------------------------
    def A(self, u_slot: int, v_slot: int = -1, out_vec: int = V_PV_RAW):
        """𝒜!(out, aux, Ut[, Vt])  src/coreop.jl:36-70."""
        self._ck(self.abi.A(self._h, u_slot, v_slot, out_vec))

    def At_preprocess(self):
        """𝒜t_preprocess!(var, aux)  src/coreop.jl:248-258."""
        self._ck(self.abi.At_preprocess(self._h))

    def At_left(self, y_slot: int, x_slot: int):
        """𝒜t!(y, x, aux, var)  src/coreop.jl:260-279."""
        self._ck(self.abi.At_left(self._h, y_slot, x_slot))

    def At_right(self, x: np.ndarray) -> np.ndarray:
        """𝒜t!(y, aux, x, var)  src/coreop.jl:281-300; x is (n,) or (n, k)."""
        x = np.asarray(x, dtype=np.float64)
        k = 1 if x.ndim == 1 else x.shape[1]
        xf = np.asfortranarray(x.reshape(self.n, k))
        yf = np.empty((self.n, k), dtype=np.float64, order="F")
        self._ck(self.abi.At_right(self._h, xf.ctypes.data_as(_pf64), yf.ctypes.data_as(_pf64), k))
        return yf.reshape(x.shape)

    def At_right_device(self, x_ptr: int, y_ptr: int, k: int = 1):
        """𝒜t!(y, aux, x, var) on device vectors given as raw device addresses (e.g. ``tensor.data_ptr()``)."""
        self._ck(self.abi.At_right_device(self._h, _vp(x_ptr), _vp(y_ptr), int(k)))

    def A_of(self, Ut: np.ndarray, Vt: Optional[np.ndarray] = None) -> np.ndarray:
        """𝒜!(out, aux, Ut[, Vt]) on the caller's own matrices (not solver state), through the scratch slots."""
        self.set_factor(F_SCRATCH, Ut)
        if Vt is not None:
            self.set_factor(F_SCRATCH + 1, Vt)
        self.A(F_SCRATCH, F_SCRATCH + 1 if Vt is not None else -1, V_SCRATCH)
        return self.get_vec(V_SCRATCH)

    def S_eigval(self, nev: int = 1, which: str = "SA", ncv: int = 0, tol: float = 0.0, maxiter: int = 1000,
                 v0: Optional[np.ndarray] = None):
        """SDP_S_eigval's solver (src/coreop.jl:361-372) on the S left by the last 𝒜t_preprocess!, on the device
        → (eigenvalues[nev], matvecs, converged count)."""
        out = np.zeros(int(nev))
        mv, nc = C.c_int64(0), C.c_int64(0)
        v0c = _f64c(v0) if v0 is not None else None
        self._ck(self.abi.S_eigval(self._h, int(nev), {"SA": 0, "LA": 1}[which], int(ncv), float(tol), int(maxiter),
                                   _pd(v0c) if v0c is not None else None, _pd(out), C.byref(mv), C.byref(nc)))
        return out, int(mv.value), int(nc.value)

    def factor_dot(self, slot_a: int, slot_b: int) -> float:
        """dot of two factor slots on the device (err6 of DIMACS_errors, src/coreop.jl:449)."""
        v = C.c_double()
        self._ck(self.abi.factor_dot(self._h, slot_a, slot_b, C.byref(v)))
        return float(v.value)

    def stats(self) -> dict:
        """Library counters (include/sdplr_hip.h, sdplr_hip_get_stats)."""
        out = (C.c_int64 * len(STAT_NAMES))()
        k = C.c_int32(0)
        self._ck(self.abi.get_stats(self._h, out, len(STAT_NAMES), C.byref(k)))
        return {STAT_NAMES[i]: int(out[i]) for i in range(k.value)}

    def f(self) -> float:
        """f!(data, var, aux)  src/coreop.jl:11-31."""
        v = C.c_double()
        self._ck(self.abi.f(self._h, C.byref(v)))
        return float(v.value)

    def g(self):
        """g!(var, aux)  src/coreop.jl:305-317."""
        self._ck(self.abi.g(self._h))

    def fg(self, normC: float, normb: float, gtol_relative=True, ptol_relative=True):
        """fg!(…)  src/coreop.jl:323-349 → (ℒ, grad_norm, primal_vio_norm)."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._ck(self.abi.fg(self._h, normC, normb, int(gtol_relative), int(ptol_relative),
                             C.byref(a), C.byref(b), C.byref(c)))
        return float(a.value), float(b.value), float(c.value)

    def lbfgs_clear(self):
        """lbfgs_clear!  src/lbfgs.jl:52-59."""
        self._ck(self.abi.lbfgs_clear(self._h))

    def lbfgs_dir(self, negate: bool = True) -> float:
        """lbfgs_dir!  src/lbfgs.jl:77-124; returns dot(dirt, Gt) (src/sdplr.jl:201)."""
        d = C.c_double()
        self._ck(self.abi.lbfgs_dir(self._h, int(negate), C.byref(d)))
        return float(d.value)

    def descent_fallback(self):
        """src/sdplr.jl:203-204."""
        self._ck(self.abi.descent_fallback(self._h))

    def lbfgs_update(self, stepsize: float):
        """lbfgs_update!  src/lbfgs.jl:129-149."""
        self._ck(self.abi.lbfgs_update(self._h, float(stepsize)))

    def linesearch(self, α_max: float = 1.0) -> Tuple[float, float]:
        """linesearch!  src/linesearch.jl:4-127 → (α, ℒ).  Raises like the reference's error(:60-62)."""
        a, L = C.c_double(), C.c_double()
        self._ck(self.abi.linesearch(self._h, float(α_max), C.byref(a), C.byref(L)))
        return float(a.value), float(L.value)

    def linesearch_armijo(self, α_max: float = 1.0) -> Tuple[float, float]:
        """linesearch_armijo!  src/linesearch.jl:139-191 → (α, ℒ_α)."""
        a, L = C.c_double(), C.c_double()
        self._ck(self.abi.linesearch_armijo(self._h, float(α_max), C.byref(a), C.byref(L)))
        return float(a.value), float(L.value)

    def axpy_R(self, α: float):
        """axpy!(α, dirt, var.Rt)  src/sdplr.jl:219."""
        self._ck(self.abi.axpy_R(self._h, float(α)))

    def norms(self, normC: float, normb: float, gtol_relative=True, ptol_relative=True):
        """src/sdplr.jl:224-234 → (grad_norm, primal_vio_norm)."""
        a, b = C.c_double(), C.c_double()
        self._ck(self.abi.norms(self._h, normC, normb, int(gtol_relative), int(ptol_relative),
                                C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def update_lambda(self):
        """src/sdplr.jl:358-362."""
        self._ck(self.abi.update_lambda(self._h))

    def inner_loop(self, normC, normb, gtol_relative, ptol_relative, use_armijo, cur_gtol,
                   fprec_eps, max_local_iters, time_budget_s, lagrangian, grad_norm,
                   primal_vio_norm):
        """The inner ``while`` of _sdplr, src/sdplr.jl:190-278, run natively.
        → (ℒ, grad_norm, primal_vio_norm, last α, iterations, exit_reason)."""
        L, g, p, a = (C.c_double(lagrangian), C.c_double(grad_norm), C.c_double(primal_vio_norm),
                      C.c_double(0.0))
        it, why = C.c_int64(0), C.c_int32(0)
        self._ck(self.abi.inner_loop(self._h, normC, normb, int(gtol_relative), int(ptol_relative),
                                     int(use_armijo), float(cur_gtol), float(fprec_eps),
                                     int(max_local_iters), float(time_budget_s), C.byref(L),
                                     C.byref(g), C.byref(p), C.byref(a), C.byref(it),
                                     C.byref(why)))
        return (float(L.value), float(g.value), float(p.value), float(a.value), int(it.value),
                int(why.value))

    def major_iteration(self, normC, normb, gtol_relative, ptol_relative, use_armijo, update_lambda, sigma,
                        cur_gtol, fprec_eps, max_local_iters, time_budget_s, L_in=0.0, gn_in=0.0, pn_in=0.0):
        """One major iteration's device work as one call: [λ update] → σ → lbfgs_clear! → fg! → inner loop
        (src/sdplr.jl:358-369, :384, :389, :190-278) → (ℒ, grad_norm, primal_vio_norm, last α, iterations, exit_reason).
        ``update_lambda = MAJOR_RESUME``: no prologue — the loop continues on the state a capped call left (its ℒ and norms
        come in as L_in, gn_in, pn_in)."""
        L, g, p, a = C.c_double(float(L_in)), C.c_double(float(gn_in)), C.c_double(float(pn_in)), C.c_double(0.0)
        it, why = C.c_int64(0), C.c_int32(0)
        self._ck(self.abi.major_iteration(self._h, normC, normb, int(gtol_relative), int(ptol_relative), int(use_armijo),
                                          int(update_lambda), float(sigma), float(cur_gtol), float(fprec_eps),
                                          int(max_local_iters), float(time_budget_s), C.byref(L), C.byref(g), C.byref(p),
                                          C.byref(a), C.byref(it), C.byref(why)))
        return (float(L.value), float(g.value), float(p.value), float(a.value), int(it.value), int(why.value))

    def lanczos(self, q: int, v0: np.ndarray):
        """src/coreop.jl:461-500 → (alpha[:steps], beta[:steps], steps)."""
        v0 = _f64c(v0)
        if v0.size != self.n:
            raise ValueError("v0 must have length n")
        q = int(q)
        al, be = np.zeros(max(q, 1)), np.zeros(max(q, 1))
        st = C.c_int64(0)
        self._ck(self.abi.lanczos(self._h, q, _pd(v0), _pd(al), _pd(be), C.byref(st)))
        k = int(st.value)
        return al[:k], be[:k], k

    def tridiag_mineig(self, alpha, beta) -> float:
        """src/coreop.jl:502-513."""
        al, be = _f64c(alpha), _f64c(beta)
        out = C.c_double()
        rc = self.abi.tridiag_mineig(_pd(al), _pd(be), int(al.size), C.byref(out))
        if rc != OK:
            raise SdplrError(rc, "tridiag_mineig failed")
        return float(out.value)

    def approx_mineigval_lanczos(self, q: int, v0) -> float:
        """approx_mineigval_lanczos  src/coreop.jl:461-514."""
        v0 = _f64c(v0)
        out = C.c_double()
        self._ck(self.abi.approx_mineigval_lanczos(self._h, int(q), _pd(v0), C.byref(out)))
        return float(out.value)

    def dual_obj(self, trace_bound: float, iter_: int, v0) -> Tuple[float, float]:
        """dual_obj  src/coreop.jl:376-415 → (dual_value, λ_min)."""
        v0 = _f64c(v0)
        d, e = C.c_double(), C.c_double()
        self._ck(self.abi.dual_obj(self._h, float(trace_bound), int(iter_), _pd(v0), C.byref(d),
                                   C.byref(e)))
        return float(d.value), float(e.value)

    # -- device timing ------------------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        self._ck(self.abi.profile_enable(self._h, int(on)))

    def profile_filter(self, name: str = ""):
        self._ck(self.abi.profile_filter(self._h, name.encode()))

    def profile(self) -> dict:
        """{kernel name: (launches, total_ms)} measured with hipEvents on the solver's own stream."""
        n = C.c_int32(0)
        self._ck(self.abi.profile_count(self._h, C.byref(n)))
        out = {}
        buf = C.create_string_buffer(128)
        for i in range(n.value):
            cnt, ms = C.c_int64(0), C.c_double(0.0)
            self._ck(self.abi.profile_get(self._h, i, buf, 128, C.byref(cnt), C.byref(ms)))
            out[buf.value.decode()] = (int(cnt.value), float(ms.value))
        return out


# -- lockstep batches: the same step of many instances as one library call (include/sdplr_hip.h) ---------------------
def _batch_results(abi: CABI, arr, solvers, unpack, rc: int = OK):
    # the call's own return code: when it failed before any item was touched (argument checks, staging blocks) every item
    # still reads ERR_UNSERVED and the message is the call's, not a handle's
    if rc != OK and all(int(arr[k].status) in (OK, ERR_UNSERVED) for k in range(len(solvers))):
        msg = abi.last_error(None)
        raise SdplrError(int(rc), msg.decode() if msg else "batch call failed")
    out = []
    for k, sv in enumerate(solvers):
        if arr[k].status != OK:
            msg = abi.last_error(sv._h)
            out.append(SdplrError(int(arr[k].status), msg.decode() if msg else "error"))
        else:
            out.append(unpack(arr[k]))
    return out


def batch_fg(abi: CABI, solvers, args):
    """``fg`` of every solver in one call; args[k] = (normC, normb, gtol_relative, ptol_relative).
    → per solver (ℒ, grad_norm, primal_vio_norm, obj), or the SdplrError its own call would have raised."""
    arr = (FgItem * len(solvers))()
    for k, (sv, a) in enumerate(zip(solvers, args)):
        it = arr[k]
        it.s = sv._h.value
        it.normC, it.normb, it.gtol_relative, it.ptol_relative = float(a[0]), float(a[1]), int(a[2]), int(a[3])
    rc = abi.batch_fg(len(solvers), arr)
    return _batch_results(abi, arr, solvers, lambda q: (q.lagrangian, q.grad_norm, q.primal_vio_norm, q.obj), rc)


def batch_major_iteration(abi: CABI, solvers, args):
    """``major_iteration`` of every solver in one call; args[k] = its argument tuple.
    → per solver (ℒ, grad_norm, primal_vio_norm, last α, iterations, exit_reason, obj) or an SdplrError."""
    arr = (MajorItem * len(solvers))()
    for k, (sv, a) in enumerate(zip(solvers, args)):
        it = arr[k]
        it.s = sv._h.value
        (it.normC, it.normb, it.gtol_relative, it.ptol_relative, it.use_armijo, it.update_lambda, it.sigma, it.cur_gtol,
         it.fprec_eps, it.max_local_iters, it.time_budget_s) = (
            float(a[0]), float(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), float(a[6]), float(a[7]), float(a[8]),
            int(a[9]), float(a[10]))
        if len(a) > 11:    # MAJOR_RESUME: ℒ and the norms the capped call returned
            it.lagrangian, it.grad_norm, it.primal_vio_norm = float(a[11]), float(a[12]), float(a[13])
    rc = abi.batch_major_iteration(len(solvers), arr)
    return _batch_results(abi, arr, solvers, lambda q: (q.lagrangian, q.grad_norm, q.primal_vio_norm, q.last_alpha,
                                                        int(q.iters_done), int(q.exit_reason), q.obj), rc)


def batch_dual_obj(abi: CABI, solvers, args):
    """``dual_obj`` of every solver in one call; args[k] = (trace_bound, iter, v0) → per solver (dual_value, λ_min, y):
    ``var.y`` as dual_obj leaves it comes back with the results (one transfer for the whole batch)."""
    arr = (DualItem * len(solvers))()
    keep, ys = [], []
    for k, (sv, a) in enumerate(zip(solvers, args)):
        v0 = _f64c(a[2])
        if v0.size != sv.n:
            raise ValueError("v0 must have length n")
        keep.append(v0)
        ys.append(np.empty(sv.m + 1))
        it = arr[k]
        it.s = sv._h.value
        it.trace_bound, it.iter, it.v0, it.y_out = float(a[0]), int(a[1]), _pd(v0), _pd(ys[k])
    rc = abi.batch_dual_obj(len(solvers), arr)
    out = _batch_results(abi, arr, solvers, lambda q: (q.dual_value, q.mineig), rc)
    return [o if isinstance(o, Exception) else o + (ys[k],) for k, o in enumerate(out)]
