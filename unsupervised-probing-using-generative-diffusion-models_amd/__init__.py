"""MI355X-native rolling-window reverse-diffusion MPV sampler.

Drop-in for the NsDiff hot path of curious-child/Unsupervised-probing-Using-Generative-
Diffusion-Models: same `models.models.diffusion_models` factory, checkpoint dict, YAML
surface and `uncertainty_ews` harness, with the arithmetic in hand-written gfx950 HIP
kernels behind the C ABI of include/mpv_hip.h (no CPU fallback).
"""
__version__ = "0.1.0"

from . import _native  # noqa: F401  (ctypes binding; does not dlopen until first use)
