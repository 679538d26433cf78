#!/bin/bash
# round-3 evidence refresh on the GPU box: smoke, the default bench line, the C1 line + its kernel trace, the VAE profile, SQ counters of the GEMMs
set -x
mkdir -p gpurun_out/ev
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py 2> gpurun_out/ev/bench_n1.err | tail -1 > gpurun_out/ev/r03_bench_n1.json
python bench.py --workload cogvideox-2b-9x256x256 --steps 50 --warmup 10 2> gpurun_out/ev/bench_c1.err | tail -1 > gpurun_out/ev/r03_bench_2b_9x256x256.json
bash tools/profile_vae.sh r03 > gpurun_out/ev/profile_vae.log 2>&1
bash tools/profile.sh r03c1 --workload cogvideox-2b-9x256x256 > gpurun_out/ev/profile_c1.log 2>&1
S2V_IMPLS=9,7 bash tools/pmc_gemm.sh gemm > gpurun_out/ev/r03_pmc_sq_gemm_raw.txt 2>&1
# the raw traces exceed what gpurun merges back: keep the summaries
cp -r gpurun_out/prof_vae_r03/summary gpurun_out/ev/vae_summary; cp -r gpurun_out/prof_r03c1/summary gpurun_out/ev/c1_summary
rm -rf gpurun_out/prof_vae_r03 gpurun_out/prof_r03c1 gpurun_out/pmc_gemm
for f in gpurun_out/ev/r03_bench_*.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('wall_clock_per_video',{}).get('vae_decode_tiled_s'))"; done
