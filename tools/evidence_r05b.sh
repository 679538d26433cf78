#!/bin/bash
# round-5 evidence, part B (one GPU box): smoke, the default bench line + its rocprofv3 kernel trace and FETCH / WRITE passes, the other workloads'
# lines (bf16), configs[0] in its own dtype (fp32) and in fp16, C3 in fp32 (the matrix-pipe reference mode).  Outputs under gpurun_out/ev5b/.
set -x
export TMPDIR=/tmp
E=gpurun_out/ev5b; mkdir -p $E
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 > $E/smoke.txt
python bench.py --steps 20 --warmup 5 2> $E/bench_n1.err | tail -1 > $E/r05_bench_n1.json
bash tools/profile.sh r05 > $E/profile.log 2>&1
cp gpurun_out/prof_r05/summary/* $E/; rm -rf gpurun_out/prof_r05
for w in cogvideox-2b-9x256x256 cogvideox-2b-49x480x720 cogvideox-5b-49x720x1280 cogvideox-5b-fp8-49x480x720 cogvideox-5b-fp8-49x720x1280 cogvideox-5b-fp8lin-49x720x1280 cogvideox-5b-fp8qk-49x480x720; do
  python bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $E/r05_bench_$w.json
done
python bench.py --steps 30 --warmup 5 --workload cogvideox-2b-9x256x256 --dtype f32 2>/dev/null | tail -1 > $E/r05_bench_cogvideox-2b-9x256x256_f32.json
python bench.py --steps 30 --warmup 5 --workload cogvideox-2b-9x256x256 --dtype f16 2>/dev/null | tail -1 > $E/r05_bench_cogvideox-2b-9x256x256_f16.json
python bench.py --steps 2 --warmup 1 --single-mode --no-vae --dtype f32 2>/dev/null | tail -1 > $E/r05_bench_n1_f32.json
python bench.py --steps 3 --warmup 1 --single-mode --workload cogvideox-2b-49x480x720 --dtype f16 2>/dev/null | tail -1 > $E/r05_bench_cogvideox-2b-49x480x720_f16.json
for f in $E/r05_bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], d['dtype'][:24], d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('frac_of_peak_at_measured_clock'), r.get('calibrated_peak'), (d.get('wall_clock_per_video') or {}).get('s_per_video_measured'))"; done
cat $E/smoke.txt
