#!/bin/bash
# Timing experiments on the generated gemm_g4 K loop (G4_ABLATE: nodma / noread / nobar -- results are wrong by construction).
# One diagnostics library per experiment in tools/g4abl/<name>/:  bash tools/g4_ablate.sh build "<name>=<ENV ...>" ...   then on the GPU
# box:  bash tools/g4_ablate.sh run
set -u
PKG=disentangled-subject-to-vid_amd
if [ "$1" = build ]; then
  shift
  rm -rf tools/g4abl; mkdir -p tools/g4abl
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}; [ "$envs" = "$spec" ] && envs=""
    d=tools/g4abl/$name; mkdir -p $d
    env $envs python $PKG/csrc/gen_gemm_g4.py
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-unused-value -Wno-inline-asm -DS2V_DIAG -fno-slp-vectorize -c $PKG/csrc/gemm_g4.hip -o $d/gemm_g4.o || exit 1
    objs=$(ls $PKG/build_diag/*.o | grep -v gemm_g4.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libs2v_hip_diag.so $objs $d/gemm_g4.o || exit 1
    rm $d/gemm_g4.o
  done
  python $PKG/csrc/gen_gemm_g4.py   # restore the real loop
else
  for d in tools/g4abl/*/; do
    echo "== $(basename $d)"
    S2V_DIAG_LIB=$d/libs2v_hip_diag.so python tools/stall_g4.py 2>&1 | grep "K-tiles" | cut -c1-175
  done
fi
