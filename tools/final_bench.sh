set -x
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/r02_bench_n1.json
python bench.py --steps 6 --warmup 2 --workload cogvideox-5b-fp8-49x480x720 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_bench_fp8_49x480x720.json
python bench.py --steps 3 --warmup 1 --workload cogvideox-5b-fp8-49x720x1280 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_bench_fp8_49x720x1280.json
python bench.py --steps 3 --warmup 1 --workload cogvideox-5b-49x720x1280 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_bench_bf16_49x720x1280.json
python bench.py --steps 10 --warmup 3 --workload cogvideox-2b-49x480x720 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_bench_2b_49x480x720.json
for f in gpurun_out/r02_bench_*.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
