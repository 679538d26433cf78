#!/bin/bash
# configs[4] attention locality, second experiment: a head's q-blocks on ONE XCD (S2V_ATTN_ORDER=0, the default) against dealt over the eight XCDs (1).
# the knob is read by the DIAGNOSTICS build only since round 6: S2V_LIB selects it
export S2V_LIB="$(cd "$(dirname "$0")/.." && pwd)/disentangled-subject-to-vid_amd/libs2v_hip_diag.so"
export TMPDIR=/tmp S2V_BENCH_SKIP_PFMT=1 S2V_BENCH_SKIP_PARITY_PASS=1
for W in cogvideox-5b-fp8lin-49x720x1280 cogvideox-5b-fp8-49x720x1280 cogvideox-5b-49x480x720; do
 for o in 0 1; do
  S2V_ATTN_ORDER=$o python bench.py --steps 2 --warmup 1 --workload $W --no-cpu-baseline --no-vae 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['roofline']['per_kernel']['attention']; print('$W order $o: step', d['ms_per_step'], 'ms; attention', a['avg_ms'], 'ms', a['tflops'], 'TFLOP/s, clock', a.get('shader_clock_mhz'), 'finite', d['config']['outputs_finite'])"
  rm -rf /tmp/pf
  S2V_ATTN_ORDER=$o rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python bench.py --steps 1 --warmup 0 --graph 0 --single-mode --no-cpu-baseline --no-vae --no-roofline --workload $W > /dev/null 2>&1
  python - /tmp/pf "$W order $o" <<'PY'
import csv, glob, sys
tot = n = 0
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Kernel_Name"].startswith("void attn_") and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print(f"{sys.argv[2]}: FETCH_SIZE {tot / max(n, 1) / 1e6:.2f} GB raw per attention launch, {n} rows")
PY
 done
done
