#!/bin/bash
# rocprofv3 evidence for the bench: kernel-trace stats in one run, HBM PMC counters in separate runs
# (never combined with other trace domains).  Usage on the GPU box:  bash tools/profile.sh r01 [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT/summary"
export TMPDIR=/tmp
BENCH="python bench.py --steps 2 --warmup 1 --graph 0 --single-mode --no-cpu-baseline --no-vae $*"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- $BENCH > "$OUT/pmc_write.log" 2>&1

python - "$OUT" "$TAG" <<'EOF'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = {}
st = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    fields = list(rows[0].keys())
    keep = [r for r in rows][:40]
    # the 128 x 128 kernel's rows in a step trace are the ROW TAILS of split GEMMs (108 rows at M = 38 252, api.hip linear()): launched on the side
    # stream beside the 256-row kernel and scheduled behind it, so their "duration" is mostly queueing, not work -- say so in the file
    for r in keep:
        name = r.get("Name", "")
        r["note"] = ("side-stream row tail of a split GEMM (the last partial 256-row tile): duration = queueing behind the main launch, not work; "
                     "exclude from shares") if "gemm_bf16_128<" in name else ""
    with open(f"{out}/summary/{tag}_kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=fields + ["note"]); w.writeheader(); w.writerows(keep)
    res["kernel_stats"] = f"{out}/summary/{tag}_kernel_stats.csv"
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for fn in files:
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") != ctr: continue
            k = r["Kernel_Name"].split("(")[0][:80]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    with open(f"{out}/summary/{tag}_{name}.csv", "w") as f:
        f.write(f"kernel,launches,mean_{ctr}_KB_raw,total_{ctr}_KB_raw\n")
        for k, (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
            f.write(f'"{k}",{n},{s/n:.1f},{s:.1f}\n')
    res[name] = f"{out}/summary/{tag}_{name}.csv"
print(json.dumps(res))
EOF
tail -3 "$OUT/trace.log"
