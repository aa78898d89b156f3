#!/bin/bash
# dev tool: kernel trace of small solves (Gset G1, n = 800, rank 10: BASELINE config 5) on the resident route —
# every kernel launched, per-kernel counts / medians, and launches per inner iteration.  Writes
# gpurun_out/ktrace_small/summary.txt (copied to profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace_small
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/small_instance_profile.py > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import sys, glob, csv, collections, statistics, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
log = open(sys.argv[1] + "/log.txt").read()
m = re.search(r"iterations (\d+) majors (\d+)", log)
iters, majors = (int(m.group(1)), int(m.group(2))) if m else (0, 0)
d = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
n_solves = 3   # small_instance_profile.py: warm-up, timed, profiled
print(f"# rocprofv3 --kernel-trace of scripts/small_instance_profile.py: {n_solves} solves of Gset G1 (n = 800, rank 10, ptol = objtol = 1e-2)")
print(f"# one solve: {iters} inner iterations, {majors} major iterations; kernels launched in all: {len(rows)}")
print(f"# launches per inner iteration: {len(rows) / max(1, n_solves * iters):.3f}")
print("kernel,launches,launches_per_solve,median_us,total_ms")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k},{len(v)},{len(v) / n_solves:.1f},{statistics.median(v):.2f},{sum(v) / 1e3:.3f}")
PY
