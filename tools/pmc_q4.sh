export TMPDIR=/tmp
OUT=gpurun_out/pmc_q4
mkdir -p $OUT
S2V_IMPLS=7,8 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES -d $OUT/b -o b -- python tools/microbench.py gemm > $OUT/b.log 2>&1
python - $OUT <<'PY'
import sqlite3, sys, glob
out = sys.argv[1]
for db in glob.glob(f"{out}/b/*.db"):
    c = sqlite3.connect(db)
    q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
    for n, ctr, cnt, avg in c.execute(q):
        if any(k in n for k in ("gemm_", "Cijk")):
            print(f"{n.replace('void ','').split('(')[0][:28]:28s} {ctr:34s} n={cnt:4d} avg={avg:16.1f}")
PY
tail -14 $OUT/b.log
