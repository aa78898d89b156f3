"""dev tool: time consecutive chunks of inner iterations on the north-star instance"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, sdplrplus_jl_amd as sj
abi = sj.load_hip()
data, var = bench.build_instance(abi, bench.GRAPH_SEED)
normC, normb = data.normC(), float(np.linalg.norm(data.b))
st = var.fg(normC, normb)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ts = []
T0 = time.perf_counter()
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    abi.device_synchronize(); t0 = time.perf_counter()
    st = bench.run_fixed(var, normC, normb, st, chunk)
    abi.device_synchronize(); dt = time.perf_counter() - t0
    ts.append((t0 - T0, dt))
med = np.median([d for _, d in ts])
print(f"median {1e6*med/chunk:.1f} us/iter over {len(ts)} chunks of {chunk}")
for k, (t, d) in enumerate(ts):
    if d > 1.5 * med:
        print(f"  slow chunk {k} at t={1e3*t:.1f} ms: {1e3*d:.2f} ms (median {1e3*med:.2f} ms)")
