#!/bin/bash
# rocprofv3 evidence for the VAE decode (untiled + tiled, 49 x 480 x 720, real widths): kernel-trace stats, then separate FETCH_SIZE /
# WRITE_SIZE passes (never combined with other trace domains).  On the GPU box:  bash tools/profile_vae.sh r03
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_vae_$TAG
mkdir -p "$OUT/summary"
export TMPDIR=/tmp
CMD="python tools/microbench.py vae"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- $CMD > "$OUT/pmc_write.log" 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
st = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(f"{out}/summary/{tag}_vae_kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows[:30])
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for fn in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") != ctr: continue
            k = r["Kernel_Name"].split("(")[0][:80]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    with open(f"{out}/summary/{tag}_vae_{name[4:]}.csv", "w") as f:
        f.write(f"kernel,launches,mean_{ctr}_KB_raw,total_{ctr}_KB_raw\n")
        for k, (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
            f.write(f'"{k}",{n},{s/n:.1f},{s:.1f}\n')
PY
tail -3 "$OUT/trace.log"
