#!/bin/bash
# configs[4] attention locality experiment: start stagger of the persistent attention launch (S2V_ATTN_STAGGER = units of 64
# the knob is read by the DIAGNOSTICS build only since round 6: S2V_LIB selects it
export S2V_LIB="$(cd "$(dirname "$0")/.." && pwd)/disentangled-subject-to-vid_amd/libs2v_hip_diag.so"
# cycles per XCD slot) at N = 50 626 -- ms per launch + shader clock from the bench's profile pass, FETCH_SIZE from a --pmc pass.
export TMPDIR=/tmp S2V_BENCH_SKIP_PFMT=1 S2V_BENCH_SKIP_PARITY_PASS=1
W=${W:-cogvideox-5b-fp8-49x720x1280}
for s in ${STAGGERS:-0 26 52 104 208}; do
  S2V_ATTN_STAGGER=$s python bench.py --steps 2 --warmup 1 --workload $W --no-cpu-baseline --no-vae 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['roofline']['per_kernel']['attention']; print('stagger $s: step', d['ms_per_step'], 'ms; attention', a['avg_ms'], 'ms', a['tflops'], 'TFLOP/s, clock', a.get('shader_clock_mhz'))"
done
for s in ${PMC_STAGGERS:-0 52}; do
  rm -rf /tmp/pf$s
  S2V_ATTN_STAGGER=$s rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf$s -o f -- python bench.py --steps 1 --warmup 0 --graph 0 --single-mode --no-cpu-baseline --no-vae --no-roofline --workload $W > /dev/null 2>&1
  python - /tmp/pf$s $s <<'PY'
import csv, glob, sys
tot = n = 0
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Kernel_Name"].startswith("void attn_") and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print(f"stagger {sys.argv[2]}: FETCH_SIZE {tot / max(n, 1) / 1e6:.2f} GB raw per attention launch (x2 per the guide = {2 * tot / max(n, 1) / 1e6:.2f} GB), {n} launches")
PY
done
