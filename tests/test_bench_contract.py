"""bench.py's --gpus contract: the flag is a claim the run has to earn -- the script spawns its own ranks when no
torchrun environment is present, refuses when the GPUs do not exist, and computes `value` / `n_gpus` from what the ranks report.
No reference line: the reference has no benchmark and no multi-GPU path."""
import importlib
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "S2V_BENCH_ONE_DEVICE", "S2V_BENCH_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    sys.path.insert(0, ROOT)
    try:
        yield importlib.import_module("bench")
    finally:
        sys.path.remove(ROOT)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the no-GPU refusal")
def test_gpus_2_without_gpus_or_env_refuses(bench):
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert e.value.code not in (0, None)
    assert "no GPU visible" in str(e.value.code)


def test_flag_and_world_size_must_agree(bench, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8"])
    assert "WORLD_SIZE=1" in str(e.value.code)
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "1"])
    assert "WORLD_SIZE=4" in str(e.value.code)
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "0"])


def test_value_is_never_the_flag_times_anything():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "args.gpus *" not in src and "* args.gpus" not in src
    assert '"n_gpus": args.gpus' not in src


def test_pmc_pass_follows_the_workload(bench):
    """the fp8 profile used to be picked for the bf16 line ('_' < 'f')"""
    b, files = bench.pmc_traffic_bytes("attention", "cogvideox-5b-49x480x720")
    assert files and all("fp8" not in f and "_c1_" not in f for f in files), files
    assert b is not None and 1.0e9 < b < 4.0e9
    b8, files8 = bench.pmc_traffic_bytes("attention", "cogvideox-5b-fp8-49x480x720")
    assert files8 and all("fp8" in f for f in files8), files8
    assert bench.pmc_traffic_bytes("attention", "cogvideox-5b-49x720x1280") == (None, [])


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_refuses():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "S2V_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout
    assert "only 1 GPU(s) visible" in r.stderr


@pytest.mark.gpu
def test_self_spawned_ranks_report_the_devices_they_ran_on():
    """two self-spawned ranks sharing cuda:0 (gloo: RCCL rejects two ranks on one device): the line counts ONE GPU and two ranks,
    and value = the two ranks' steps over the slower rank's time"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(S2V_BENCH_ONE_DEVICE="1", S2V_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload",
                        "cogvideox-2b-9x256x256", "--no-vae", "--no-roofline", "--no-cpu-baseline", "--single-mode"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 2 and line["config"]["backend"] == "gloo"
    assert line["config"]["launcher"] == "self-spawned" and len(line["config"]["ranks"]) == 2
    assert abs(line["value"] - 2 * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) <= 0.02 * line["value"]
    assert line["config"]["weight_broadcast_gb"] > 0 and line["config"]["outputs_finite"]


def test_cfg_parallel_needs_an_even_number_of_ranks(bench):
    """--cfg-parallel pairs the ranks up (dist.cfg_pair_layout): an odd or single-rank request is refused before anything is launched"""
    for n in ("1", "3"):
        with pytest.raises(SystemExit) as e:
            bench.main(["--gpus", n, "--cfg-parallel"])
        assert "even" in str(e.value.code)


@pytest.mark.gpu
def test_cfg_parallel_pair_on_one_device_reports_video_steps():
    """two self-spawned ranks of ONE CFG-parallel pair sharing cuda:0 (gloo carries the all-gather): the line counts video steps -- each step
    is done by BOTH ranks, so value = steps / time, not 2 x --, names the exchange, and the latents stayed finite on both ranks"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(S2V_BENCH_ONE_DEVICE="1", S2V_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cfg-parallel", "--steps", "3", "--warmup", "1", "--workload",
                        "cogvideox-2b-9x256x256", "--no-roofline", "--single-mode"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["rccl_ranks"] == 2 and cfg["samples_per_gpu_per_step"] == 1
    assert cfg["parallelism"] == "cfg-parallel pairs x1" and cfg["cfg_parallel"]["pairs"] == 1 and "all_gather" in cfg["cfg_parallel"]["exchange"]
    assert abs(line["value"] - line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) <= 0.02 * line["value"]
    assert cfg["outputs_finite"] and line["scaling"] == "strong"


@pytest.mark.gpu
def test_batch_1_reports_the_half_step_as_the_projection():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "S2V_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "1", "--steps", "3", "--warmup", "1", "--workload",
                        "cogvideox-2b-9x256x256", "--no-roofline", "--single-mode"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["samples_per_gpu_per_step"] == 1 and line["config"]["projected_cfg_parallel_ms"] == line["ms_per_step"]
    assert abs(line["value"] - 0.5e3 / line["ms_per_step"]) <= 0.02 * line["value"]   # half a step per GPU-step: two GPUs make one step
    assert "cpu_baseline" not in line and line["wall_clock_per_video"] is None
