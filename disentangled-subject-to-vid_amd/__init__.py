"""MI355X-native CogVideoX denoise + VAE-decode hot path of carpedkm/disentangled-subject-to-vid.
Host side (Python) mirrors the reference's plug-in objects; all arithmetic lives in libs2v_hip.so (csrc/)."""
from . import _lib, config, tables, weights  # noqa: F401
from ._lib import S2VError, lib  # noqa: F401
from .config import TransformerConfig, VAEConfig, cogvideox_2b, cogvideox_5b, tiny  # noqa: F401
from .engine import S2VEngine  # noqa: F401
from .schedulers import CogVideoXDDIMScheduler, CogVideoXDPMScheduler  # noqa: F401
from .transformer import HipCogVideoXAttnProcessor2_0, HipCogVideoXBlock, HipCogVideoXTransformer3DModel  # noqa: F401
from .pipeline import S2VPipeline  # noqa: F401
from .vae import HipAutoencoderKLCogVideoX  # noqa: F401
from . import dist  # noqa: F401
from . import checkpoint  # noqa: F401
from .t5 import HipT5EncoderModel, T5Config  # noqa: F401
from . import video_generate  # noqa: F401
