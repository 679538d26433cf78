#!/bin/bash
# Timing experiments on the generated attn_q4 / attn_q8 bodies: ablations (Q4_ABLATE: results are wrong by construction) and placement
# variants (Q4_ORDER, Q4_READPOS: results stay right).  One diagnostics library per experiment in tools/q4abl/<name>/ (git-ignored, travels
# to the GPU box):   bash tools/q4_ablate.sh build "<name>=<ENV ...>" ...      then on the GPU box:   bash tools/q4_ablate.sh run
set -u
PKG=disentangled-subject-to-vid_amd
if [ "$1" = build ]; then
  shift
  rm -rf tools/q4abl; mkdir -p tools/q4abl
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}; [ "$envs" = "$spec" ] && envs=""
    d=tools/q4abl/$name; mkdir -p $d
    env $envs python $PKG/csrc/gen_attn_q4.py
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-unused-value -Wno-inline-asm -DS2V_DIAG -fno-slp-vectorize -c $PKG/csrc/attention_q4.hip -o $d/attention_q4.o || exit 1
    objs=$(ls $PKG/build_diag/*.o | grep -v attention_q4.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libs2v_hip_diag.so $objs $d/attention_q4.o || exit 1
    rm $d/attention_q4.o
  done
  python $PKG/csrc/gen_attn_q4.py   # restore the real bodies
else
  for d in tools/q4abl/*/; do
    a=$(basename $d)
    echo "== $a: $(HARNESS_TIME_ONLY=1 LD_LIBRARY_PATH=$d tools/attn_harness ${VARIANTS:-7,9} ${ROUNDS:-3} 2>&1 | grep '^time' | sed 's/(incl. V^T transpose ~0.1 ms)//' | tr '\n' ' ')"
  done
fi
