#!/bin/bash
# PMC counters for the GEMM / attention microbench (own runs, no trace domains)
export TMPDIR=/tmp
WHAT=${1:-gemm}
OUT=gpurun_out/pmc_$WHAT
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/a -o a -- python tools/microbench.py $WHAT > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/b -o b -- python tools/microbench.py $WHAT > $OUT/b.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/c -o c -- python tools/microbench.py $WHAT > $OUT/c.log 2>&1
python - $OUT <<'PY'
import sqlite3, sys, os, glob
out = sys.argv[1]
for sub in "abc":
    for db in glob.glob(f"{out}/{sub}/*.db"):
        c = sqlite3.connect(db)
        q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
        rows = [r for r in c.execute(q) if any(k in r[0] for k in ("gemm_", "attn_", "Cijk"))]
        for n, ctr, cnt, avg in rows:
            print(f"{n.replace('void ','').split('(')[0][:28]:28s} {ctr:34s} n={cnt:4d} avg={avg:16.1f}")
PY
tail -2 $OUT/a.log
