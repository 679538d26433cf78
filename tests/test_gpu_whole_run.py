"""Whole-run parity at the headline configuration (BASELINE configs[2] geometry: CogVideoX-5B, 42 layers, 49 x 480 x 720 -> 19 126 tokens,
CFG pair): the first 10 steps of the 50-step DDIM loop (custom_cogvideox_pipe.py:237-311) in bf16 (both attention P formats), fp8
and fp8-qk against the SAME loop in the fp32 model dtype on the fp32 matrix pipe -- the oracle-pinned mode (tests/test_gpu_f32m.py,
tests/test_gpu_fullsize_oracle.py [5b-f32-mfma], the fp32 goldens of tests/test_gpu_parity.py).  ~100 s on an MI355X.

Bars = 2 x the values measured in round 5 (profiles/r05_whole_run_c3_50steps.txt; the full 50 steps there: bf16 final rel-L2 1.8e-2,
fp8 2.4e-2, of which 1.4e-2 is the bf16 STORAGE of the latents between steps that the reference's bf16 pipeline has as well):
    step 10: bf16 / bf16-p16 rel-L2 8.4e-3, max-abs 0.136 (max|latent| 4.7);  fp8 / fp8-qk rel-L2 8.7e-3, max-abs 0.136
and the deviation must grow no faster than measured: step 1 <= 4e-3 (measured 1.7e-3 / 1.8e-3: half a bf16 ulp of the largest latent).
"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("whole_run_parity", os.path.join(ROOT, "tools", "whole_run_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_whole_run_c3_10_steps_vs_fp32_matrix_pipe(s2v):
    lines = []
    res, secs = _tool().whole_run(s2v, "cogvideox_5b", "c3", steps=10, schedule=50, formats=("bf16", "bf16-p16", "fp8", "fp8-qk"),
                                  log=lines.append, arith_ref=False)
    print("\n".join(lines))
    for name, rows in res.items():
        assert len(rows) == 10, name
        assert rows[0][2] <= 4e-3, f"{name}: step 1 rel-L2 {rows[0][2]}"
        for (i, ma, rl, rm, _, _) in rows:
            assert rl <= 2e-2 and ma <= 0.3, f"{name} step {i + 1}: rel-L2 {rl}, max-abs {ma} (max|ref| {rm})"
    # fp16 probabilities / fp8 QK^T add nothing measurable on top of their base format over a whole run
    assert abs(res["bf16-p16"][-1][2] - res["bf16"][-1][2]) <= 1e-3
    assert abs(res["fp8-qk"][-1][2] - res["fp8"][-1][2]) <= 1e-3


def test_whole_run_2b_c3_10_steps_bf16_and_fp16_vs_fp32_matrix_pipe(s2v):
    """BASELINE configs[1] (CogVideoX-2B, 30 layers, 49 x 480 x 720 -> 19 126 tokens): the same comparison in bf16 and in the fp16 model dtype
    (what the reference loads a 2B checkpoint in, inference.py:191,209).  Measured in round 5 (profiles/r05_whole_run_2b_c3_50steps.txt):
    step 10 bf16 rel-L2 9.9e-3 / max-abs 0.125, fp16 1.19e-3 / 1.94e-2 (max|latent| 4.7); all 50 steps: bf16 2.0e-2, fp16 2.4e-3.
    Bars = 2 x measured.  ~45 s."""
    lines = []
    res, secs = _tool().whole_run(s2v, "cogvideox_2b", "c3", steps=10, schedule=50, formats=("bf16", "f16"), log=lines.append, arith_ref=False)
    print("\n".join(lines))
    bars = {"bf16": (2e-2, 0.25), "f16": (2.4e-3, 4e-2)}
    for name, rows in res.items():
        assert len(rows) == 10, name
        for (i, ma, rl, rm, _, _) in rows:
            assert rl <= bars[name][0] and ma <= bars[name][1], f"{name} step {i + 1}: rel-L2 {rl}, max-abs {ma} (max|ref| {rm})"
    assert res["f16"][-1][2] < 0.25 * res["bf16"][-1][2]  # three more mantissa bits in every stored tensor
