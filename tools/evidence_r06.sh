#!/bin/bash
# round-6 evidence (one GPU box): smoke, the default bench line + its rocprofv3 kernel trace and FETCH / WRITE passes, EVERY other workload's line WITH its
# cpu_baseline and calibrated roofline (no --no-cpu-baseline any more), configs[0] in its own dtype (fp32: the full oracle step beside it)
# and in fp16, C3 in fp32, the CFG-parallel half step (--batch 1) and the two-ranks-on-one-GPU functional run of --gpus 2 --cfg-parallel.
# Outputs under gpurun_out/ev6/.  PART=a (default line + profiles), b (the other workloads), c (cfg-parallel) or all.
set -x
export TMPDIR=/tmp
PART=${1:-all}
E=gpurun_out/ev6; mkdir -p $E
if [ $PART = a ] || [ $PART = all ]; then
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 > $E/smoke.txt
python bench.py --steps 20 --warmup 5 2> $E/bench_n1.err | tail -1 > $E/r06_bench_n1.json
bash tools/profile.sh r06 > $E/profile.log 2>&1
cp gpurun_out/prof_r06/summary/* $E/; rm -rf gpurun_out/prof_r06
fi
if [ $PART = b ] || [ $PART = all ]; then
for w in cogvideox-2b-9x256x256 cogvideox-2b-49x480x720 cogvideox-5b-49x720x1280 cogvideox-5b-fp8-49x480x720 cogvideox-5b-fp8-49x720x1280 cogvideox-5b-fp8auto-49x720x1280 cogvideox-5b-fp8qk-49x480x720; do
  python bench.py --steps 5 --warmup 2 --workload $w 2> $E/bench_$w.err | tail -1 > $E/r06_bench_$w.json
done
python bench.py --steps 30 --warmup 5 --workload cogvideox-2b-9x256x256 --dtype f32 2> $E/bench_c1_f32.err | tail -1 > $E/r06_bench_cogvideox-2b-9x256x256_f32.json
python bench.py --steps 30 --warmup 5 --workload cogvideox-2b-9x256x256 --dtype f16 2>/dev/null | tail -1 > $E/r06_bench_cogvideox-2b-9x256x256_f16.json
python bench.py --steps 2 --warmup 1 --single-mode --no-vae --dtype f32 2>/dev/null | tail -1 > $E/r06_bench_n1_f32.json
python bench.py --steps 3 --warmup 1 --single-mode --workload cogvideox-2b-49x480x720 --dtype f16 2>/dev/null | tail -1 > $E/r06_bench_cogvideox-2b-49x480x720_f16.json
fi
if [ $PART = c ] || [ $PART = all ]; then
python bench.py --steps 10 --warmup 3 --batch 1 2>/dev/null | tail -1 > $E/r06_bench_n1_batch1.json
# two ranks of ONE CFG-parallel pair on the one GPU of this box (gloo carries the all-gather; RCCL refuses two ranks on one device): a functional run of
# the --gpus 2 --cfg-parallel wiring -- both ranks share the GPU, so the step takes about twice the --batch 1 time; NOT a two-GPU measurement
S2V_BENCH_ONE_DEVICE=1 S2V_BENCH_BACKEND=gloo python bench.py --gpus 2 --cfg-parallel --steps 5 --warmup 2 --no-roofline 2> $E/bench_cfgp_onedev.err | tail -1 > $E/r06_bench_cfg_parallel_one_device.json
S2V_BENCH_ONE_DEVICE=1 S2V_BENCH_BACKEND=gloo python bench.py --gpus 2 --cfg-parallel --steps 20 --warmup 5 --no-roofline --workload cogvideox-2b-9x256x256 2>/dev/null | tail -1 > $E/r06_bench_cfg_parallel_one_device_c1.json
fi
for f in $E/r06_bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; c=d.get('cpu_baseline') or {}; print('$f'.split('/')[-1], d['dtype'][:24], d['value'], d['ms_per_step'], r.get('kernel'), r.get('frac'), r.get('calibrated_peak'), 'cpu', c.get('value'), c.get('max_abs_vs_hip'), c.get('rel_l2_vs_hip'), (d.get('wall_clock_per_video') or {}).get('s_per_video_measured'))"; done
cat $E/smoke.txt 2>/dev/null
if [ $PART = d ] || [ $PART = all ]; then
# the gemm_g4t (persistent, trickled bias epilogue) probe on the out-projection / FF2 shapes, as a file
python tools/g4t_shapes_probe.py 2>/dev/null > $E/r06_g4t_out_ff2_probe.txt
python tools/vae_decode_time.py 2>/dev/null > $E/r06_vae_decode_times.txt
python tools/f32m_bench.py c1 2>/dev/null > $E/r06_f32m_c1_tiles.txt
fi
if [ $PART = e ]; then
# FETCH_SIZE / WRITE_SIZE passes of the other BASELINE configurations (roofline.traffic of their lines reads them from profiles/): configs[1], configs[0], configs[4]
for t in "r06_c2 cogvideox-2b-49x480x720" "r06_c1 cogvideox-2b-9x256x256" "r06_c5fp8 cogvideox-5b-fp8-49x720x1280"; do
  set -- $t; bash tools/profile.sh $1 --workload $2 > $E/profile_$1.log 2>&1; cp gpurun_out/prof_$1/summary/* $E/; rm -rf gpurun_out/prof_$1
done
fi
