"""sdplrplus.jl_amd — MI355X-native device backend for the SDPLR+ hot path.

The directory name contains a dot, so import it through the repo-root shim::

    import sdplrplus_jl_amd as sj

Layout: ``csrc/`` hand-written HIP kernels + the C ABI of include/sdplr_hip.h (built into
``lib/libsdplr_hip.so``), ``cabi.py`` ctypes binding / ``DeviceSolver``, ``preprocess.py`` one-time
host layout, ``problems.py`` SDP builders, ``sdplr.py`` host control flow, ``build.py`` hipcc recipe.
"""
from . import batch, cabi, io, preprocess, problems, rounding, structs  # noqa: F401
from .cabi import CABI, DeviceSolver, SdplrError, load_hip, hip_library_path  # noqa: F401
from .preprocess import AggregatedLayout, preprocess_sparsecons  # noqa: F401
from .sdplr import DIMACS_errors, SDP_S_eigval, _sdplr, build_solver, initial_point, sdplr  # noqa: F401
from .structs import (BurerMonteiroConfig, Diagonal, SDPData, SparseBatch, SparseMatrixCOO,  # noqa: F401
                      SymLowRankMatrix, barvinok_pataki)

__all__ = ["sdplr", "SymLowRankMatrix", "SDPData", "BurerMonteiroConfig", "DeviceSolver", "CABI",
           "load_hip", "preprocess_sparsecons", "problems"]
