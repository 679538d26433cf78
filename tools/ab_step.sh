#!/bin/bash
# Same-box A/B of the denoise step between builds of the product library that differ in the generated GEMM loops (boxes differ by ~5 %
# in clock: numbers from different gpurun calls do not compare).
#   bash tools/ab_step.sh build "<name>=<ENV ...>" ...   (ENV reaches gen_gemm_g4.py / gen_gemm_g4t.py; "<name>=" = the tree as it is)
#   bash tools/ab_step.sh run [bench.py args]            on the GPU box: one bench line per build, twice, interleaved
set -u
PKG=disentangled-subject-to-vid_amd
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-unused-value -Wno-inline-asm -fno-slp-vectorize"
if [ "$1" = build ]; then
  shift
  rm -rf tools/ababl; mkdir -p tools/ababl
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}; [ "$envs" = "$spec" ] && envs=""
    d=tools/ababl/$name; mkdir -p $d
    env $envs python $PKG/csrc/gen_gemm_g4.py; env $envs python $PKG/csrc/gen_gemm_g4t.py
    /opt/rocm/bin/hipcc $FL -c $PKG/csrc/gemm_g4.hip -o $d/gemm_g4.o || exit 1
    /opt/rocm/bin/hipcc $FL -c $PKG/csrc/gemm_g4t.hip -o $d/gemm_g4t.o || exit 1
    objs=$(ls $PKG/build/*.o | grep -v "gemm_g4.o\|gemm_g4t.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libs2v_hip.so $objs $d/gemm_g4.o $d/gemm_g4t.o || exit 1
    rm $d/*.o
  done
  python $PKG/csrc/gen_gemm_g4.py; python $PKG/csrc/gen_gemm_g4t.py
else
  shift
  for rep in 1 2; do
    for d in tools/ababl/*/; do
      S2V_LIB=$d/libs2v_hip.so python bench.py --steps 8 --warmup 2 --no-vae --no-cpu-baseline --single-mode "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']
print('%-14s %7.2f ms/step  ' % ('$(basename $d)', d['ms_per_step']) + '  '.join('%s %.3f' % (k.replace('gemm_',''), v['avg_ms']) for k, v in pk.items() if 'avg_ms' in v and k in ('gemm_qkv','attention','gemm_out','gemm_ff1_gelu','gemm_ff2')))"
    done
  done
fi
