#!/bin/bash
# kernel-trace durations of the GEMM microbench (to compare with the event-timed launch period)
export TMPDIR=/tmp
OUT=gpurun_out/trace_gemm
rm -rf $OUT; mkdir -p $OUT
S2V_IMPLS=${S2V_IMPLS:-7,8} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python tools/microbench.py gemm > $OUT/log.txt 2>&1
grep -v amdgpu.ids $OUT/log.txt | grep "gemm\|hipBLASLt"
python - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in ("gemm_", "Cijk")):
        d[(n.split("(")[0][:40], r.get("Grid_Size_X", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in d.items():
    dur = [e - s for s, e in v]
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    gaps = [g for g in gaps if g < 1e6]
    print(f"{k[0]:40s} grid {k[1]:>8s} n={len(v):3d}  duration avg {sum(dur)/len(dur)/1e3:8.1f} us  min {min(dur)/1e3:8.1f}   gap-to-next avg {sum(gaps)/max(len(gaps),1)/1e3:6.1f} us")
PY
