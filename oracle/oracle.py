"""TEST INFRASTRUCTURE — ctypes access to the CPU restatement (oracle/libsdplr_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  The
oracle exports the same C ABI as libsdplr_hip.so under the prefix ``sdplr_oracle_``, so it is bound
with the package's generic ``CABI`` class; the extras (preprocess restatement, SymLowRank norm,
quartic scalar stage) are bound here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsdplr_oracle.so")


LIB_OMP = os.path.join(HERE, "libsdplr_oracle_omp.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "sdplr_oracle.c")
    stale = lambda lib: not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src)
    if force or stale(LIB) or stale(LIB_OMP):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB


def timing_abis(scratch: str = None):
    """The two builds bench.py's cpu_baseline times: (one-thread ABI, all-cores OpenMP ABI, how they were
    built, set_threads(k) for the OpenMP build).  When a C compiler is present on the box that runs the bench, both are rebuilt there with
    -march=native into a scratch directory; otherwise the prebuilt generic-x86-64 libraries that
    travelled with the snapshot are used and the returned note says so."""
    import shutil
    import tempfile
    import sdplrplus_jl_amd as sj
    note = "prebuilt, -O3 generic x86-64 (no compiler on this box)"
    one, omp = LIB, LIB_OMP
    if shutil.which(os.environ.get("CC", "gcc")) and shutil.which("make"):
        out = scratch or tempfile.mkdtemp(prefix="sdplr_oracle_native_")
        try:
            subprocess.check_call(["make", "-C", HERE, "-s", "-B", "native", f"OUT={out}"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            one, omp = os.path.join(out, "libsdplr_oracle_1t.so"), os.path.join(out, "libsdplr_oracle_omp.so")
            note = "built on this box, gcc -O3 -march=native"
        except (subprocess.CalledProcessError, OSError):
            pass
    if not os.path.exists(omp):
        build()
    one_abi, omp_abi = sj.CABI(one, "sdplr_oracle_"), sj.CABI(omp, "sdplr_oracle_")
    gomp = C.CDLL("libgomp.so.1")          # already mapped as a dependency of the OpenMP build: same handle
    gomp.omp_set_num_threads.argtypes = [C.c_int]

    def set_threads(k: int):
        gomp.omp_set_num_threads(int(k))
    return one_abi, omp_abi, note, set_threads


class _Layout(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n", "nA", "nnzT", "nnzS", "nnzAgg")] + [
        ("triu_colptr", C.POINTER(C.c_int64)), ("triu_rowval", C.POINTER(C.c_int64)),
        ("full_colptr", C.POINTER(C.c_int64)), ("full_rowval", C.POINTER(C.c_int64)),
        ("matptr", C.POINTER(C.c_int64)), ("nzind", C.POINTER(C.c_int64)),
        ("nzval_one", C.POINTER(C.c_double)), ("nzval_two", C.POINTER(C.c_double)),
        ("mappedto_triu", C.POINTER(C.c_int64))]


_abi = None


def abi():
    """The oracle as a ``CABI`` (same entry points as the HIP library, prefix sdplr_oracle_)."""
    global _abi
    if _abi is None:
        import sdplrplus_jl_amd as sj
        build()
        _abi = sj.CABI(LIB, "sdplr_oracle_")
        lib = _abi.lib
        lib.sdplr_oracle_preprocess.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64),
                                                C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                                C.POINTER(C.c_double), C.POINTER(_Layout)]
        lib.sdplr_oracle_preprocess.restype = C.c_int32
        lib.sdplr_oracle_layout_free.argtypes = [C.POINTER(_Layout)]
        lib.sdplr_oracle_layout_free.restype = None
        lib.sdplr_oracle_symlowrank_norm.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_double),
                                                     C.POINTER(C.c_double), C.c_int32,
                                                     C.POINTER(C.c_double)]
        lib.sdplr_oracle_symlowrank_norm.restype = C.c_int32
        lib.sdplr_oracle_quartic_argmin.argtypes = [C.POINTER(C.c_double), C.c_double,
                                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.sdplr_oracle_quartic_argmin.restype = C.c_int32
    return _abi


def preprocess(batch):
    """preprocess_sparsecons restated in C (src/preprocess.jl:24-169) → AggregatedLayout."""
    import sdplrplus_jl_amd as sj
    lib = abi().lib
    ptr = np.ascontiguousarray(batch.ent_ptr, dtype=np.int64)
    I = np.ascontiguousarray(batch.I, dtype=np.int64)
    J = np.ascontiguousarray(batch.J, dtype=np.int64)
    V = np.ascontiguousarray(batch.V, dtype=np.float64)
    out = _Layout()
    p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    rc = lib.sdplr_oracle_preprocess(batch.n, batch.n_matrices, 0, p64(ptr), p64(I), p64(J),
                                     V.ctypes.data_as(C.POINTER(C.c_double)), C.byref(out))
    if rc != 0:
        raise RuntimeError(f"oracle preprocess failed: {rc}")
    cp = lambda p, k, dt: np.ctypeslib.as_array(p, shape=(max(k, 1),))[:k].astype(dt, copy=True)
    n = batch.n
    lay = sj.AggregatedLayout(
        n, batch.n_matrices, cp(out.matptr, out.nA + 1, np.int64), cp(out.nzind, out.nnzAgg, np.int64),
        cp(out.nzval_one, out.nnzAgg, np.float64), cp(out.nzval_two, out.nnzAgg, np.float64),
        np.asarray(batch.global_inds, dtype=np.int64).copy(), cp(out.triu_colptr, n + 1, np.int64),
        cp(out.triu_rowval, out.nnzT, np.int64), cp(out.full_colptr, n + 1, np.int64),
        cp(out.full_rowval, out.nnzS, np.int64), cp(out.mappedto_triu, out.nnzS, np.int64))
    lib.sdplr_oracle_layout_free(C.byref(out))
    return lay


def symlowrank_norm(B, D, p=2) -> float:
    """norm(A::SymLowRankMatrix, p), src/structs.jl:61-82."""
    lib = abi().lib
    Bf = np.asfortranarray(B, dtype=np.float64)
    Dc = np.ascontiguousarray(D, dtype=np.float64)
    out = C.c_double()
    rc = lib.sdplr_oracle_symlowrank_norm(Bf.shape[0], Bf.shape[1], Bf.ctypes.data_as(C.POINTER(C.c_double)),
                                          Dc.ctypes.data_as(C.POINTER(C.c_double)),
                                          1 if p == np.inf else 0, C.byref(out))
    if rc != 0:
        raise RuntimeError("symlowrank_norm failed")
    return float(out.value)


def quartic_argmin(biquadratic, alpha_max=1.0):
    """scalar stage of linesearch!, src/linesearch.jl:58-112 → (rc, α*, f(α*))."""
    lib = abi().lib
    bq = np.ascontiguousarray(biquadratic, dtype=np.float64)
    a, f = C.c_double(), C.c_double()
    rc = lib.sdplr_oracle_quartic_argmin(bq.ctypes.data_as(C.POINTER(C.c_double)), alpha_max,
                                         C.byref(a), C.byref(f))
    return rc, float(a.value), float(f.value)
