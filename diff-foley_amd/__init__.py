"""MI355X-native Diff-Foley Stage-2 sampling path (UNet denoise loop, DDIM/DPM/PLMS samplers,
VAE spectrogram decode) behind the reference's ``LatentDiffusion`` API."""
from . import synth  # noqa: F401
