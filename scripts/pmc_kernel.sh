#!/bin/bash
# dev tool: SQ counters of one kernel (default k_spmm_tile) over a short bench run
#   gpurun -- 'bash scripts/pmc_kernel.sh k_spmm_tile'
K=${1:-k_spmm_tile}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$K
rm -rf $OUT; mkdir -p $OUT
export SDPLR_BENCH_PREWARM_S=0
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d $OUT/a -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM -d $OUT/b -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OUT/c.log 2>&1
python3 - "$OUT" "$K" <<'PY'
import sys, glob, csv, collections
out, kern = sys.argv[1], sys.argv[2]
for sub in "abc":
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            big = [x for x in v if x > 0.05 * max(v)] or [0.0]     # real launches only
            print(f"{k:28s} n={len(big):4d} median={sorted(big)[len(big)//2]:16.1f}")
PY
