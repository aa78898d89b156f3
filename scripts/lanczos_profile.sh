#!/bin/bash
# Lanczos (approx_mineigval_lanczos, src/coreop.jl:461-500) under rocprofv3: kernel trace + HBM-side traffic
# (FETCH_SIZE, WRITE_SIZE in separate --pmc passes) + SQ counters of the SpMV kernel.
#   gpurun -- 'bash scripts/lanczos_profile.sh r02'   → gpurun_out/lz_prof/{<tag>_lanczos_kernel_medians.csv,<tag>_lanczos_pmc.csv}
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/lz_prof
rm -rf $OUT; mkdir -p $OUT
PROG="python3 $GRAFT_REPO_ROOT/scripts/lanczos_only.py"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $PROG > $OUT/trace.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -o p --output-format csv -- $PROG > $OUT/$C.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $OUT/sq -o p --output-format csv -- $PROG > $OUT/sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/tcc -o p --output-format csv -- $PROG > $OUT/tcc.log 2>&1 || true
python3 - "$OUT" "$TAG" <<'PY'
import sys, glob, csv, collections, statistics
out, tag = sys.argv[1], sys.argv[2]
short = lambda name: name.split("(")[0].replace("void ", "").split("<")[0]
f = glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    d[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
with open(f"{out}/{tag}_lanczos_kernel_medians.csv", "w") as g:
    g.write("# rocprofv3 --kernel-trace --stats -- python3 scripts/lanczos_only.py  (MaxCut G(1e5,2e-4), 3 x 232 Lanczos steps); durations in us\n")
    g.write("kernel,launches,median_us,mean_us,min_us,total_ms\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        g.write(f"{k},{len(v)},{statistics.median(v):.2f},{sum(v)/len(v):.2f},{min(v):.2f},{sum(v)/1e3:.3f}\n")
rows = collections.defaultdict(dict)
for sub in ("FETCH_SIZE", "WRITE_SIZE", "sq", "tcc"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(short(row["Kernel_Name"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            big = [x for x in v if x > 0.05 * max(v)] or [0.0]   # real launches only (fall-through launches move nothing)
            rows[k][c] = statistics.median(big)
with open(f"{out}/{tag}_lanczos_pmc.csv", "w") as g:
    g.write("# rocprofv3 --pmc <counters> --kernel-trace -- python3 scripts/lanczos_only.py ; separate passes for FETCH_SIZE / WRITE_SIZE / SQ / TCC\n")
    g.write("# FETCH_SIZE, WRITE_SIZE in KiB per launch (median over real launches); fetch_MB_x2 applies the gfx950 correction (MI355X_MICROARCH.md, HBM)\n")
    names = sorted({c for r in rows.values() for c in r})
    g.write("kernel,fetch_MB_x2,write_MB," + ",".join(n for n in names if n not in ("FETCH_SIZE", "WRITE_SIZE")) + "\n")
    for k, r in sorted(rows.items()):
        if not k.startswith("k_lz") and not k.startswith("k_spmv"): continue
        g.write(f"{k},{2*r.get('FETCH_SIZE',0)*1024/1e6:.3f},{r.get('WRITE_SIZE',0)*1024/1e6:.3f}," +
                ",".join(f"{r.get(n, float('nan')):.1f}" for n in names if n not in ("FETCH_SIZE", "WRITE_SIZE")) + "\n")
print(open(f"{out}/{tag}_lanczos_kernel_medians.csv").read())
print(open(f"{out}/{tag}_lanczos_pmc.csv").read())
print(open(f"{out}/trace.log").read()[-400:])
PY
