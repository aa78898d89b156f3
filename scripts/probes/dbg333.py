import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sdplrplus_jl_amd as sj
from helpers import make_data
which = sys.argv[1]
if which == "oracle":
    from oracle import oracle
    abi = oracle.abi()
else:
    abi = sj.load_hip()
d = make_data("maxcut", 4, 333, 0.08)[0]
r = sj.sdplr(data=d, r=10, abi=abi, ptol=1e-2, objtol=1e-2, maxtime=10.0, printlevel=1, printfreq=1e9)
print("RESULT", r["iter"], r["majoriter"], r["r"], r["obj"], r["max_dual_value"], r["dual_time"])
