"""Oracle: first-stage decode + cond-stage encoder.  Test infrastructure only.

  * LatentDiffusion.decode_first_stage   diff_foley/models/diffusion/ddpm.py:739-797 (z / scale_factor)
  * AutoencoderKL.decode                 diff_foley/models/autoencoder.py:330-333
  * Decoder.forward                      diff_foley/modules/stage1_autoencoder/model.py:630-663
  * ResnetBlock.forward (temb=None)      model.py:216-236
  * AttnBlock.forward                    model.py:273-297
  * Upsample.forward                     model.py:148-152
  * Video_Feat_Encoder_Posembed.forward  diff_foley/modules/cond_stage/video_feat_encoder.py:12-18
"""
import torch
import torch.nn.functional as F

VAE_FULL = dict(z_channels=4, embed_dim=4, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, out_ch=3)
VAE_TINY = dict(z_channels=4, embed_dim=4, ch=64, ch_mult=[1, 2, 2], num_res_blocks=1, out_ch=3)


def _gn(x, sd, p):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)


def _conv(x, sd, p, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def _swish(x):
    return x * torch.sigmoid(x)


def resnet_block(sd, p, x):
    h = _conv(_swish(_gn(x, sd, p + ".norm1")), sd, p + ".conv1")
    h = _conv(_swish(_gn(h, sd, p + ".norm2")), sd, p + ".conv2")
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(x, sd, p + ".nin_shortcut", padding=0)
    return x + h


def attn_block(sd, p, x):
    h = _gn(x, sd, p + ".norm")
    q = _conv(h, sd, p + ".q", 0)
    k = _conv(h, sd, p + ".k", 0)
    v = _conv(h, sd, p + ".v", 0)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(h, sd, p + ".proj_out", 0)


@torch.no_grad()
def vae_decode(sd, cfg, z):
    """AutoencoderKL.decode.  ``sd`` keys relative to ``first_stage_model.``."""
    nres = len(cfg["ch_mult"])
    h = _conv(z, sd, "post_quant_conv", 0)
    h = _conv(h, sd, "decoder.conv_in")
    h = resnet_block(sd, "decoder.mid.block_1", h)
    h = attn_block(sd, "decoder.mid.attn_1", h)
    h = resnet_block(sd, "decoder.mid.block_2", h)
    for lvl in reversed(range(nres)):
        for ib in range(cfg["num_res_blocks"] + 1):
            h = resnet_block(sd, f"decoder.up.{lvl}.block.{ib}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, sd, f"decoder.up.{lvl}.upsample.conv")
    h = _swish(_gn(h, sd, "decoder.norm_out"))
    return _conv(h, sd, "decoder.conv_out")


@torch.no_grad()
def decode_first_stage(sd, cfg, z, scale_factor=0.18215):
    return vae_decode(sd, cfg, (1.0 / scale_factor) * z)


@torch.no_grad()
def cond_stage(sd, x):
    """Video_Feat_Encoder_Posembed.forward.  ``sd`` keys relative to ``cond_stage_model.``."""
    bs, seq_len, _ = x.shape
    y = F.linear(x, sd["embedder.0.weight"], sd["embedder.0.bias"])
    return y + sd["pos_emb.weight"][:seq_len][None]
