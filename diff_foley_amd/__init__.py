"""MI355X-native Diff-Foley Stage-2 sampling path (UNet denoise loop, DDIM/DPM/PLMS samplers,
VAE spectrogram decode) behind the reference's ``LatentDiffusion`` API.

Compute lives in ``libdfengine.so`` (hand-written HIP for gfx950, C ABI in ``include/df_engine.h``);
this package is the Python host side mirroring the reference's interface.
"""
from . import synth  # noqa: F401
from . import schedule  # noqa: F401
from . import engine  # noqa: F401
from .ldm import LatentDiffusion, AlignmentClassifier, CAVPInference, instantiate_from_config  # noqa: F401
from .samplers import DDIMSampler, PLMSSampler, DPMSolverSampler  # noqa: F401
from .video import ExtractCAVPFeatures, frames_to_tensor  # noqa: F401
from .vocoder import inverse_op, mel_to_wave  # noqa: F401


def stage2_config(unet=None, vae=None, cond=None):
    """``params`` of inference/config/Stage2_LDM.yaml as a plain dict (optionally with shrunken sub-configs)."""
    unet = dict(synth.UNET_FULL if unet is None else unet)
    vae = dict(synth.VAE_FULL if vae is None else vae)
    cond = dict(synth.COND_FULL if cond is None else cond)
    return dict(
        linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000,
        first_stage_key="mix_spec", cond_stage_key="mix_video_feat", image_size=64, channels=4,
        cond_stage_trainable=True, conditioning_key="crossattn", scale_factor=0.18215, use_ema=False,
        unet_config=dict(target="diff_foley.modules.diffusionmodules.openai_unetmodel.UNetModel",
                         params=dict(image_size=32, use_spatial_transformer=True, transformer_depth=1,
                                     use_checkpoint=True, legacy=False, **unet)),
        first_stage_config=dict(target="diff_foley.models.autoencoder.AutoencoderKL",
                                params=dict(embed_dim=vae["embed_dim"], monitor="val/rec_loss",
                                            ddconfig=dict(double_z=True, z_channels=vae["z_channels"], resolution=256,
                                                          in_channels=3, out_ch=vae["out_ch"], ch=vae["ch"],
                                                          ch_mult=list(vae["ch_mult"]),
                                                          num_res_blocks=vae["num_res_blocks"], attn_resolutions=[],
                                                          dropout=0.0),
                                            lossconfig=dict(target="torch.nn.Identity"))),
        cond_stage_config=dict(target="diff_foley.modules.cond_stage.video_feat_encoder.Video_Feat_Encoder_Posembed",
                               params=cond))
