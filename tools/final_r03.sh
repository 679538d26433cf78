mkdir -p gpurun_out/fin
python bench.py 2> gpurun_out/fin/bench.err | tail -1 > gpurun_out/fin/r03_bench_n1_final.json
bash tools/profile.sh r03 > gpurun_out/fin/profile.log 2>&1
cp -r gpurun_out/prof_r03/summary gpurun_out/fin/summary; rm -rf gpurun_out/prof_r03
python -c "
import json; d=json.loads(open('gpurun_out/fin/r03_bench_n1_final.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['wall_clock_per_video']['vae_decode_tiled_s'])"
head -8 gpurun_out/fin/summary/r03_kernel_stats.csv | cut -c1-120
