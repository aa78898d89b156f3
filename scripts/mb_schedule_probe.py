import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import ctypes as C
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems
from oracle import oracle
oracle.build()
omp = sj.CABI(oracle.LIB_OMP, "sdplr_oracle_")
C.CDLL("libgomp.so.1").omp_set_num_threads(16)
n = 100_000
data = problems.minimum_bisection_data(problems.gnp_graph(n, 2e-4, 4))
kw = dict(r=32, ptol=1e-2, objtol=1e-2, seed=0, prior_trace_bound=float(n), printlevel=0)
for K in (3, 6):
    a = sj.sdplr(data=data, abi=sj.load_hip(), maxmajoriter=K, **kw)
    b = sj.sdplr(data=data, abi=omp, maxmajoriter=K, **kw)
    print(K, "HIP", a["schedule"], a["obj"], a["primal_vio"], a["iter"])
    print(K, "OMP", b["schedule"], b["obj"], b["primal_vio"], b["iter"])
