#!/bin/bash
# dev tool: kernel trace of the config-5 batch (64 MaxCut instances, n = 800, rank 10) driven in lockstep — every kernel
# launched by two passes over the batch (scripts/run_batch.py), per-kernel counts / medians.  Writes
# gpurun_out/ktrace_lockstep/summary.txt (copied to profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace_lockstep
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/run_batch.py > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import sys, glob, csv, collections, statistics, json
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
line = [l for l in open(sys.argv[1] + "/log.txt") if l.startswith("{")][-1]
res = json.loads(line)
iters = sum(res["iterations"])
d = collections.defaultdict(list)
grid = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    grid.setdefault(name, set()).add(int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
print("# rocprofv3 --kernel-trace of scripts/run_batch.py (lockstep driver): 2 passes over 64 MaxCut instances (n = 800, rank 10,")
print(f"# ptol = objtol = 1e-2) + one tiny warm-up solve; one pass: {iters} inner iterations in all; wall of the 2nd pass {res['wall_s']:.4f} s under the profiler")
print(f"# kernels launched in all: {len(rows)}; per pass and instance: {len(rows) / 2 / 64:.2f}; per inner iteration: {len(rows) / 2 / max(1, iters):.4f}")
print("kernel,launches,workgroups_per_launch,median_us,max_us,total_ms")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    g = sorted(grid[k])
    print(f"{k},{len(v)},{g[0] if len(g) == 1 else str(g[0]) + '-' + str(g[-1])},{statistics.median(v):.2f},{max(v):.2f},{sum(v) / 1e3:.3f}")
PY
