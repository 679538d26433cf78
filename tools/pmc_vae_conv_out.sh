#!/bin/bash
# SQ counters of conv_out_direct_k inside an untiled 49 x 480 x 720 decode (two separate --pmc passes, no trace domains)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pa /tmp/pb
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d /tmp/pa -o a -- python /root/repo/tools/vae_trace_once.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pb -o b -- python /root/repo/tools/vae_trace_once.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for fn in glob.glob("/tmp/pa/**/*counter_collection.csv", recursive=True) + glob.glob("/tmp/pb/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "conv_out_direct" not in r["Kernel_Name"]:
            continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(agg.items()):
    print(f"conv_out_direct_k {k:28s} launches {n:3d}  avg {s / n:16.1f}")
PY
