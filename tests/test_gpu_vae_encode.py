"""GPU parity of the reference-image VAE encode (pytest -m gpu): C ABI (s2v_vae_enc_create /
s2v_vae_encode / s2v_vae_gaussian_sample) vs the fixtures generated from the reference and vs the CPU oracle.
Tolerances: fp32 max-abs <= 1e-3 (measured ~1e-5); bf16 relative L2 <= 3e-2 against the oracle's own bf16 run."""
import numpy as np
import pytest
import torch

from conftest import load_golden, weights_of
from oracle import vae_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = dict(block_out_channels=(16, 16, 32, 32), layers_per_block=1, norm_num_groups=4, latent_channels=16,
            sample_height=96, sample_width=160, scaling_factor=0.7, temporal_compression_ratio=4)


def t(x, dt=torch.float32):
    return torch.from_numpy(np.asarray(x)).to(dt)


def make_vae(s2v, cfgd, dt, sd, force_simple=False):
    cfg = s2v.VAEConfig(**cfgd)
    vae = s2v.HipAutoencoderKLCogVideoX(cfg, dt, DEV, force_simple)
    vae.load_state_dict(sd)  # encoder.* only: the decoder half stays unloaded
    return vae


class _Replay(torch.Generator):
    pass


@pytest.mark.parametrize("tiling", [False, True])
def test_encode_tiny_fp32_vs_reference_golden(s2v, tiling):
    g = load_golden("vae_enc_tiny.npz")
    name = "tiled" if tiling else "untiled"
    vae = make_vae(s2v, TINY, torch.float32, weights_of(g))
    if tiling:
        vae.enable_tiling()
    post = vae.encode(t(g["image"]).to(DEV)).latent_dist
    torch.cuda.synchronize()
    assert tuple(post.parameters.shape) == (1, 32, 1, 12, 20)
    assert (post.parameters.cpu() - t(g[f"moments_{name}"])).abs().max().item() <= 1e-3
    # the reference drew its noise from Generator().manual_seed(7): the same CPU generator must reproduce the sample
    z = post.sample(generator=torch.Generator().manual_seed(7)) * vae.config.scaling_factor
    torch.cuda.synchronize()
    lat = z.permute(0, 2, 1, 3, 4).cpu()
    assert (lat - t(g[f"latent_{name}"])).abs().max().item() <= 1e-3


def test_encode_small_window_and_errors(s2v):
    g = load_golden("vae_enc_tiny.npz")
    vae = make_vae(s2v, TINY, torch.float32, weights_of(g))
    x = t(g["image"])[..., :40, :56].contiguous()
    mom = vae.encode(x.to(DEV)).latent_dist.parameters
    torch.cuda.synchronize()
    assert (mom.cpu() - t(g["moments_small"])).abs().max().item() <= 1e-3
    with pytest.raises(NotImplementedError):
        vae.encode(torch.zeros(1, 3, 2, 16, 16, device=DEV))  # video encode is outside the path
    with pytest.raises(s2v.S2VError):
        vae.encode(torch.zeros(1, 3, 1, 20, 16, device=DEV))  # sides must be multiples of 8
    dec_only = s2v.HipAutoencoderKLCogVideoX(s2v.VAEConfig(**TINY), torch.float32, DEV)
    with pytest.raises(s2v.S2VError):
        dec_only.encode(x.to(DEV))


def test_gaussian_sample_bit_exact_bf16(s2v):
    """mean + exp(0.5 * clamp(logvar)) * noise with every op rounded to bf16, against torch's own bf16 tensor ops"""
    gen = torch.Generator().manual_seed(3)
    mom = (torch.randn(1, 32, 1, 9, 11, generator=gen) * 3.0)
    mom[0, 16:, 0, 0, :4] = torch.tensor([-50.0, 40.0, 19.9, -29.9])  # clamp edges
    mom = mom.bfloat16()
    vae = s2v.HipAutoencoderKLCogVideoX(s2v.VAEConfig(**TINY), torch.bfloat16, DEV)
    dist = importlib_vae(s2v).HipDiagonalGaussianDistribution(mom.to(DEV))
    z = dist.sample(generator=torch.Generator().manual_seed(5)).cpu()
    noise = torch.randn(1, 16, 1, 9, 11, generator=torch.Generator().manual_seed(5), dtype=torch.bfloat16)
    mean, logvar = torch.chunk(mom, 2, dim=1)
    exp = mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise
    assert torch.equal(z, exp)
    del vae


def importlib_vae(s2v):
    import importlib
    return importlib.import_module(s2v.__name__ + ".vae")


@pytest.mark.parametrize("tiling", [False, True])
def test_encode_mfma_width_bf16_vs_oracle(s2v, tiling):
    """64/128-channel encoder in bf16: every conv except conv_in runs on the MFMA implicit GEMM, the downsamplers on its
    stride-2 form; 144 x 288 image, tiles of (96|64) x (160|160|32) when tiled."""
    cfgd = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=1, norm_num_groups=8, latent_channels=16,
                sample_height=192, sample_width=320, scaling_factor=0.7, temporal_compression_ratio=4)
    cfg = s2v.VAEConfig(**cfgd)
    sd = s2v.weights.synthetic_vae_encoder_state_dict(cfg, seed=41)
    sdb = {k: v.bfloat16() for k, v in sd.items()}
    img = (torch.rand(1, 3, 1, 144, 288, generator=torch.Generator().manual_seed(42)) * 2 - 1).bfloat16()
    outs = {}
    for simple in (False, True):
        vae = make_vae(s2v, cfgd, torch.bfloat16, sdb, force_simple=simple)
        if tiling:
            vae.enable_tiling()
        outs[simple] = vae.encode(img.to(DEV)).latent_dist.parameters.float().cpu()
        torch.cuda.synchronize()
    with torch.no_grad():
        exp = vae_ref.encode_moments(sdb, cfgd, img, tiling).float()
    assert outs[False].shape == exp.shape
    for simple, y in outs.items():
        rel = ((y - exp).norm() / exp.norm()).item()
        assert rel <= 3e-2, (simple, rel)


@pytest.mark.parametrize("dt_name", ["f32", "bf16"])
def test_encode_real_width_vs_oracle(s2v, dt_name):
    """the encoder the checkpoints ship -- (128, 256, 256, 512) channels, 3 resnets per block, 32 groups
    (autoencoder_kl_cogvideox.py:755-814) -- on a 64 x 96 image, against the fp32 oracle"""
    dt = torch.float32 if dt_name == "f32" else torch.bfloat16
    cfgd = dict(block_out_channels=(128, 256, 256, 512), layers_per_block=3, norm_num_groups=32, latent_channels=16,
                sample_height=480, sample_width=720, scaling_factor=0.7, temporal_compression_ratio=4)
    cfg = s2v.VAEConfig(**cfgd)
    sd = {k: v.to(dt).float() for k, v in s2v.weights.synthetic_vae_encoder_state_dict(cfg, seed=71).items()}
    img = (torch.rand(1, 3, 1, 64, 96, generator=torch.Generator().manual_seed(72)) * 2 - 1).to(dt).float()
    with torch.no_grad():
        exp = vae_ref.encode_moments(sd, cfgd, img, False)
    vae = make_vae(s2v, cfgd, dt, sd)
    y = vae.encode(img.to(DEV, dt)).latent_dist.parameters.float().cpu()
    torch.cuda.synchronize()
    assert y.shape == exp.shape == (1, 32, 1, 8, 12)
    err = (y - exp).abs().max().item()
    if dt_name == "f32":
        assert err <= 1e-3, err
    else:
        rel = ((y - exp).norm() / exp.norm()).item()
        assert rel <= 2e-2 and err <= 6e-2 * exp.abs().max().item(), (rel, err)


def test_encode_full_size_determinism(s2v):
    """480 x 720 reference image through the full-width encoder, tiled (9 tiles of <= 240 x 360): finite, the right shape,
    and bit-identical across two runs (deterministic GroupNorm reductions)."""
    cfg = s2v.VAEConfig(scaling_factor=0.7)
    sd = s2v.weights.synthetic_vae_encoder_state_dict(cfg, seed=51, dtype=torch.bfloat16)
    vae = s2v.HipAutoencoderKLCogVideoX(cfg, torch.bfloat16, DEV)
    vae.load_state_dict(sd)
    vae.enable_tiling()
    img = (torch.rand(1, 3, 1, 480, 720, generator=torch.Generator().manual_seed(52)) * 2 - 1).bfloat16().to(DEV)
    a = vae.encode(img).latent_dist.parameters.clone()
    b = vae.encode(img).latent_dist.parameters
    torch.cuda.synchronize()
    assert tuple(a.shape) == (1, 32, 1, 60, 90)
    assert torch.isfinite(a.float()).all()
    assert torch.equal(a, b)
