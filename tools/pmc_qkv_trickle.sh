#!/bin/bash
# SQ / GRBM counters of the fused QKV projection at the C3 shape (rocprofv3 --pmc, two separate counter passes, no trace domains): gemm_g4<4> with the
# C++ q/k-norm epilogue, gemm_g4t<4> with the trickled one, the plain-bias projection on both kernels (tools/qkv_trickle_bench.py launches all four).
#   bash tools/pmc_qkv_trickle.sh  ->  gpurun_out/pmc_qkv/summary.md
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/pmc_qkv}
mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o a -- python tools/qkv_trickle_bench.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/b -o b -- python tools/qkv_trickle_bench.py > $OUT/b.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("gemm_g4"): continue
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
lines = ["## The fused QKV projection at the C3 shape (M = 38144, N = 9216, K = 3072): `bash tools/pmc_qkv_trickle.sh`", "",
         "| kernel | launches | MFMA busy = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024) | GUI_ACTIVE / 8 (cycles per launch) | WAIT_ANY / WAVE_CYCLES | VALU instructions | LDS instructions |", "|---|---|---|---|---|---|---|"]
for k, c in sorted(agg.items()):
    g = lambda n: (c[n][0] / c[n][1]) if n in c and c[n][1] else float("nan")
    lines.append(f"| `{k}` | {c['GRBM_GUI_ACTIVE'][1]} | **{100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (g('GRBM_GUI_ACTIVE') / 8 * 1024):.1f}%** | {g('GRBM_GUI_ACTIVE') / 8 / 1e6:.3f} M | "
                 f"{100 * g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.1f}% | {g('SQ_INSTS_VALU') / 1e6:.1f} M | {g('SQ_INSTS_LDS') / 1e6:.1f} M |")
open(f"{out}/summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
