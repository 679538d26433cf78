#!/bin/bash
# SQ / GRBM counters of the PRODUCT attention kernels inside the engine's step (rocprofv3 --pmc, two separate counter passes per configuration, no
# trace domains): attn_q4 (bf16 P), attn_q4h (fp16 P) at C3; attn_q4f / attn_q4fh (fp8 QK^T) at C3 and at the configs[4] geometry (N = 50 626).
#   bash tools/pmc_attn_r05.sh  ->  gpurun_out/pmc_attn_r05/summary.md  (copy to profiles/r05_pmc_sq.md)
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/pmc_attn_r05}
mkdir -p $OUT
export S2V_BENCH_SKIP_PFMT=1 S2V_BENCH_SKIP_PARITY_PASS=1
B="python bench.py --steps 1 --warmup 0 --graph 0 --single-mode --no-cpu-baseline --no-vae --no-roofline"
run() {  # tag, attn_p, workload
  S2V_ATTN_P=$2 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/$1/a -o a -- $B --workload $3 > $OUT/$1.a.log 2>&1
  S2V_ATTN_P=$2 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/$1/b -o b -- $B --workload $3 > $OUT/$1.b.log 2>&1
}
run c3_bf16p bf16 cogvideox-5b-49x480x720
run c3_f16p f16 cogvideox-5b-49x480x720
run c3_fp8qk_bf16p bf16 cogvideox-5b-fp8qk-49x480x720
run c3_fp8qk_f16p f16 cogvideox-5b-fp8qk-49x480x720
run c5_f16p f16 cogvideox-5b-49x720x1280
run c5_fp8qk_bf16p bf16 cogvideox-5b-fp8qk-49x720x1280
run c5_fp8qk_f16p f16 cogvideox-5b-fp8qk-49x720x1280
python - $OUT <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
lines = ["# r05 SQ / GRBM counters of the product attention kernels inside the engine's denoise step (rocprofv3 --pmc, separate passes, no trace domains)", "",
         "`bash tools/pmc_attn_r05.sh`: `bench.py --steps 1 --warmup 0 --graph 0` per configuration; averages over the launches of the attention kernel (84 per run: 42 layers x "
         "[census step of bench.py's set-up is off; the eager step + the pass's own]).  GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs; "
         "SQ_WAVE_CYCLES / SQ_WAIT_* are quad-cycles summed over waves.", "",
         "| configuration | kernel | launches | MFMA busy = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024) | SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES | WAIT_ANY / WAVE_CYCLES | WAIT_INST_ANY / WAVE_CYCLES | VALU instructions | LDS instructions |",
         "|---|---|---|---|---|---|---|---|---|"]
raw = []
for cfg in sorted(d for d in os.listdir(out) if os.path.isdir(os.path.join(out, d))):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for fn in glob.glob(f"{out}/{cfg}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not k.startswith("attn_"): continue
            a = agg[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, c in sorted(agg.items()):
        g = lambda n: (c[n][0] / c[n][1]) if n in c and c[n][1] else float("nan")
        n = max(v[1] for v in c.values())
        busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024)
        lines.append(f"| {cfg} | `{k}` | {n} | **{busy:.1%}** | ACTIVE_INST_VALU / BUSY_CYCLES = {g('SQ_ACTIVE_INST_VALU') / g('SQ_BUSY_CYCLES'):.3f} | {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.1%} | "
                     f"{g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.1%} | {g('SQ_INSTS_VALU') / 1e6:.1f} M | {g('SQ_INSTS_LDS') / 1e6:.1f} M |")
        for ctr, (s, m) in sorted(c.items()):
            raw.append(f"{cfg:16s} {k:40s} {ctr:32s} n={m:4d} avg={s / m:18.1f}")
open(f"{out}/summary.md", "w").write("\n".join(lines) + "\n\nRaw averages per launch:\n\n```\n" + "\n".join(raw) + "\n```\n")
print("\n".join(lines))
PY
