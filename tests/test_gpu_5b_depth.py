"""CogVideoX-5B at REAL depth and width (42 layers, D = 3072, 48 heads, rotary embedding) against the CPU oracle on a geometry the
oracle finishes in under a minute (9 frames 64 x 96 -> 322 tokens, 3 DDIM steps, CFG 6, S2VPipeline with hipGraph replay):
fp32 final latents within 1e-3 (measured 2.9e-5); bf16 drift no larger than the oracle's own bf16-vs-fp32 drift allows.
The body lives in tools/parity_5b_depth.py (it prints the per-step numbers kept in profiles/r02_parity_5b_depth.txt)."""
import os
import runpy

import pytest

pytestmark = pytest.mark.gpu


def test_5b_real_depth_three_steps_vs_oracle(s2v):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runpy.run_path(os.path.join(root, "tools", "parity_5b_depth.py"), run_name="__main__")
