#!/bin/bash
# same-box A/B: the product library against a build without the shader-clock stamps / stagger check in the four-wave kernels
#   bash tools/ab_noclk.sh build   (here)      bash tools/ab_noclk.sh run   (GPU box)
PKG=disentangled-subject-to-vid_amd
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-unused-value -Wno-inline-asm -fno-slp-vectorize"
if [ "$1" = build ]; then
  rm -rf tools/ababl; mkdir -p tools/ababl/noclk tools/ababl/product
  for f in gemm_g4 gemm_g4t gemm_g4f attention_q4; do /opt/rocm/bin/hipcc $FL -DS2V_NO_CLK_STAMP -c $PKG/csrc/$f.hip -o tools/ababl/noclk/$f.o || exit 1; done
  objs=$(ls $PKG/build/*.o | grep -v "gemm_g4.o\|gemm_g4t.o\|gemm_g4f.o\|attention_q4.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ababl/noclk/libs2v_hip.so $objs tools/ababl/noclk/*.o || exit 1
  rm tools/ababl/noclk/*.o; cp $PKG/libs2v_hip.so tools/ababl/product/
else
  export S2V_BENCH_SKIP_PFMT=1 S2V_BENCH_SKIP_PARITY_PASS=1
  for rep in 1 2 3; do for d in tools/ababl/*/; do
    S2V_LIB=$d/libs2v_hip.so python bench.py --steps 8 --warmup 2 --no-vae --no-cpu-baseline --single-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']
print('%-10s %7.2f ms/step  ' % ('$(basename $d)', d['ms_per_step']) + '  '.join('%s %.3f' % (k.replace('gemm_',''), v['avg_ms']) for k, v in pk.items() if k in ('gemm_qkv','attention','gemm_out','gemm_ff1_gelu','gemm_ff2')), ' cal', d['roofline'].get('calibrated_peak'))"
  done; done
fi
