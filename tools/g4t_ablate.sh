#!/bin/bash
# Timing experiments on the generated gemm_g4t body (G4T_TPS = trickle slots per MFMA; G4T_ABLATE: nogelu / nostore / nodrain / notrickle --
# results are wrong by construction under an ablation).  One diagnostics library per experiment in tools/g4tabl/<name>/:
#   bash tools/g4t_ablate.sh build "<name>=<ENV ...>" ...     then on the GPU box:   bash tools/g4t_ablate.sh run
set -u
PKG=disentangled-subject-to-vid_amd
if [ "$1" = build ]; then
  shift
  rm -rf tools/g4tabl; mkdir -p tools/g4tabl
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}; [ "$envs" = "$spec" ] && envs=""
    d=tools/g4tabl/$name; mkdir -p $d
    env $envs python $PKG/csrc/gen_gemm_g4t.py
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-unused-value -Wno-inline-asm -DS2V_DIAG -fno-slp-vectorize -c $PKG/csrc/gemm_g4t.hip -o $d/gemm_g4t.o || exit 1
    objs=$(ls $PKG/build_diag/*.o | grep -v gemm_g4t.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libs2v_hip_diag.so $objs $d/gemm_g4t.o || exit 1
    rm $d/gemm_g4t.o
  done
  python $PKG/csrc/gen_gemm_g4t.py   # restore the real body
else
  for d in tools/g4tabl/*/; do
    echo "== $(basename $d)"
    S2V_DIAG_LIB=$d/libs2v_hip_diag.so python tools/g4t_probe.py ${2:-} 2>&1 | grep -v amdgpu.ids | grep "^ff1\|^qkv\|g4t:\|g4 :\|bitwise"
  done
fi
