#!/bin/bash
# SQ / GRBM / TCC counters of the attention kernels at the C3 shape (rocprofv3 --pmc, separate passes, no trace domains):
#   bash tools/pmc_attn.sh   ->  gpurun_out/pmc_attn/summary.txt   (VARIANTS=0,4: attn_pp_k per q-block / persistent; 2 = round-1 lock-step kernel)
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/pmc_attn}
mkdir -p $OUT
CMD="tools/attn_harness ${VARIANTS:-0,4} 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c -o c -- $CMD > $OUT/c.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- $CMD > $OUT/t.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for sub in "abc":
    for fn in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if int(r["Grid_Size"]) not in (7200 * 512, 256 * 512, 7200 * 256, 256 * 256): continue  # the C3 launches only (per-q-block / persistent grid, 8- or 4-wave kernel)
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
dur = collections.defaultdict(lambda: [0.0, 0])
for fn in glob.glob(f"{out}/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if int(r["Grid_Size_X"]) not in (7200 * 512, 256 * 512, 7200 * 256, 256 * 256): continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; dur[k][1] += 1
with open(f"{out}/summary.txt", "w") as f:
    for (k, c), (s, n) in sorted(agg.items()):
        f.write(f"{k:24s} {c:28s} n={n:3d} avg={s / n:18.1f}\n")
    for k, (s, n) in sorted(dur.items()):
        f.write(f"{k:24s} {'duration_ms':28s} n={n:3d} avg={s / n:18.4f}\n")
print(open(f"{out}/summary.txt").read())
PY
