for i in 1 2 3; do
  for L in libsdplr_hip.so libsdplr_hip_NT_G.so libsdplr_hip_NT_R.so; do
    SDPLR_HIP_LIBRARY=$PWD/sdplrplus.jl_amd/lib/$L timeout -k 10 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_eager_profile'];print('$L', round(d['value'],1), {n:k[n]['us_per_step'] for n in ('fast_step','spmm_W','lbfgs_dir') if n in k})"
  done
done
