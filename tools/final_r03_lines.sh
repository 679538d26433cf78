# the other workloads' bench lines on the round's last tree (configs[4] fp8 / bf16 at 49 x 720 x 1280, fp8 and 2B at 49 x 480 x 720)
mkdir -p gpurun_out/lines
for w in cogvideox-5b-fp8-49x720x1280 cogvideox-5b-49x720x1280 cogvideox-5b-fp8-49x480x720 cogvideox-2b-49x480x720; do
  python bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/lines/r03_bench_$w.json
  python -c "
import json; d=json.loads(open('gpurun_out/lines/r03_bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d.get('wall_clock_per_video',{}).get('vae_decode_tiled_s'))"
done
