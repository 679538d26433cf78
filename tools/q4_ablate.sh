#!/bin/bash
# Timing ablations of the generated attn_q4 loop (results are wrong by construction; timing only).  Builds one diagnostics library
# per ablation into tools/q4abl/<name>/ on the BUILD host:   bash tools/q4_ablate.sh build
# and times them on the GPU box:                                    bash tools/q4_ablate.sh run
set -u
PKG=disentangled-subject-to-vid_amd
LIST="base nodma noread nosoft movexp nobar nowait nodma,noread nodma,noread,nosoft nosoft,nodma nomfma"
if [ "$1" = build ]; then
  for a in $LIST; do
    d=tools/q4abl/$a; mkdir -p $d
    Q4_ABLATE=$([ $a = base ] && echo "" || echo $a) python $PKG/csrc/gen_attn_q4.py
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-unused-value -Wno-inline-asm -DS2V_DIAG -fno-slp-vectorize -c $PKG/csrc/attention_q4.hip -o $d/attention_q4.o || exit 1
    objs=$(ls $PKG/build_diag/*.o | grep -v attention_q4.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libs2v_hip_diag.so $objs $d/attention_q4.o || exit 1
    rm $d/attention_q4.o
  done
  python $PKG/csrc/gen_attn_q4.py   # restore the real loop
else
  for a in $LIST; do
    echo "== $a: $(HARNESS_TIME_ONLY=1 LD_LIBRARY_PATH=tools/q4abl/$a tools/attn_harness ${VARIANTS:-7} 3 2>&1 | grep '^time')"
  done
fi
