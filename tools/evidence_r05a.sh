#!/bin/bash
# round-5 evidence, part A (one GPU box): SQ counters of the product attention kernels, per-convolution VAE rates with the fixed matcher, the
# 50-step soak, whole-run parity at the configs[4] geometry and for the 2B model.  Outputs under gpurun_out/ev5/ (copy into profiles/).
set -x
export TMPDIR=/tmp
E=gpurun_out/ev5; mkdir -p $E
OUT=$E/pmc_attn bash tools/pmc_attn_r05.sh > $E/pmc_attn.log 2>&1
cp $E/pmc_attn/summary.md $E/r05_pmc_sq.md; rm -rf $E/pmc_attn/*/a $E/pmc_attn/*/b
rm -f /tmp/convs.txt; rm -rf /tmp/vt
S2V_LIB=disentangled-subject-to-vid_amd/libs2v_hip_diag.so S2V_VAE_CONV_LOG=/tmp/convs.txt rocprofv3 --kernel-trace --output-format csv -d /tmp/vt -o t -- python tools/vae_profile_probe.py once > $E/vae_trace.log 2>&1
python tools/vae_conv_rates.py /tmp/convs.txt /tmp/vt > $E/r05_vae_conv_rates.txt 2>&1
python tools/soak_pipeline.py > $E/r05_soak_pipeline.txt 2>&1
python tools/whole_run_parity.py --steps 10 --preset cogvideox_2b --formats bf16,bf16-p16 > $E/r05_whole_run_2b_c3_10steps.txt 2>&1
python tools/whole_run_parity.py --steps 10 --geometry c5 --formats bf16,fp8,fp8-qk,fp8-qk-p16 --no-arith-ref > $E/r05_whole_run_c5_10steps.txt 2>&1
tail -5 $E/r05_vae_conv_rates.txt $E/r05_soak_pipeline.txt; grep SUMMARY $E/r05_whole_run_*.txt; head -20 $E/r05_pmc_sq.md
