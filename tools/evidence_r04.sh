#!/bin/bash
# round-4 evidence on ONE GPU box: smoke, the default bench line, its rocprofv3 kernel trace + HBM PMC passes (bf16 and fp8), the C1 line,
# the other workloads' lines, SQ counters of the GEMM microbench, the g4 / g4t / A3 probes, the VAE profile, the fp8 depth probe.
# Summaries land in gpurun_out/ev4/ (copy what is to be judged into profiles/).
set -x
E=gpurun_out/ev4; mkdir -p $E
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 2> $E/bench_n1.err | tail -1 > $E/r04_bench_n1.json
bash tools/profile.sh r04 > $E/profile.log 2>&1
cp gpurun_out/prof_r04/summary/* $E/; rm -rf gpurun_out/prof_r04
bash tools/profile.sh r04fp8 --workload cogvideox-5b-fp8-49x480x720 > $E/profile_fp8.log 2>&1
cp gpurun_out/prof_r04fp8/summary/* $E/; rm -rf gpurun_out/prof_r04fp8
python bench.py --workload cogvideox-2b-9x256x256 --steps 50 --warmup 10 2> $E/bench_c1.err | tail -1 > $E/r04_bench_2b_9x256x256.json
bash tools/profile.sh r04_c1 --workload cogvideox-2b-9x256x256 > $E/profile_c1.log 2>&1
cp gpurun_out/prof_r04_c1/summary/* $E/; rm -rf gpurun_out/prof_r04_c1
for w in cogvideox-5b-fp8-49x720x1280 cogvideox-5b-fp8qk-49x720x1280 cogvideox-5b-49x720x1280 cogvideox-5b-fp8-49x480x720 cogvideox-5b-fp8qk-49x480x720 cogvideox-2b-49x480x720; do
  python bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $E/r04_bench_$w.json
done
S2V_IMPLS=9,7 S2V_NO_G4T=1 bash tools/pmc_gemm.sh gemm > $E/r04_pmc_sq_gemm_raw.txt 2>&1; rm -rf gpurun_out/pmc_gemm
S2V_IMPLS=9 bash tools/pmc_gemm.sh gemm > $E/r04_pmc_sq_gemm_g4t_raw.txt 2>&1; rm -rf gpurun_out/pmc_gemm
S2V_ATTN_P=f16 python bench.py --steps 3 --warmup 1 --workload cogvideox-5b-fp8qk-49x720x1280 --no-cpu-baseline 2>/dev/null | tail -1 > $E/r04_bench_cogvideox-5b-fp8qk-f16p-49x720x1280.json
S2V_ATTN_P=f16 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vae 2>/dev/null | tail -1 > $E/r04_bench_n1_f16p.json
python tools/g4t_probe.py > $E/r04_gemm_g4t.txt 2>&1
python tools/stall_g4.py > $E/r04_stall_g4.txt 2>&1
python tools/probe_blaslt.py > $E/r04_probe_blaslt.txt 2>&1
bash tools/profile_vae.sh r04 > $E/profile_vae.log 2>&1
cp gpurun_out/prof_vae_r04/summary/* $E/; rm -rf gpurun_out/prof_vae_r04
python tools/fp8_depth_probe.py > $E/r04_fp8_depth.txt 2>&1
for f in $E/r04_bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], (d.get('wall_clock_per_video') or {}).get('vae_decode_tiled_s'))"; done
