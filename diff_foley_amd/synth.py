"""Synthetic weights and inputs for the Stage-2 sampling path.

The trained checkpoints (``ldm_epoch240.ckpt`` ...) are not in the reference tree and
cannot be fetched, so benchmarks and parity tests run on *procedurally generated*
weights: every tensor of the reference ``state_dict`` layout is filled from a
counter-based generator keyed by (seed, crc32(name)), so the same 944 M parameters
can be regenerated bit-identically anywhere without being stored.

``state_dict_spec`` enumerates the key layout the reference's
``LatentDiffusion.state_dict()`` has for the sub-modules on the hot path
(SURVEY.md section 5 "Checkpoint"): ``model.diffusion_model.*``,
``first_stage_model.{post_quant_conv,decoder}.*``, ``cond_stage_model.*`` and the
classifier's ``model.*``.  tests/golden/make_golden.py asserts that names and shapes
agree with the real reference modules.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

UNET_FULL = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                 num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, context_dim=768)
UNET_TINY = dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1],
                 num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=2, context_dim=128)
VAE_FULL = dict(z_channels=4, embed_dim=4, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, out_ch=3)
VAE_TINY = dict(z_channels=4, embed_dim=4, ch=64, ch_mult=[1, 2, 2], num_res_blocks=1, out_ch=3)
COND_FULL = dict(origin_dim=512, embed_dim=768, seq_len=40)
COND_TINY = dict(origin_dim=64, embed_dim=128, seq_len=40)
CLS_FULL = dict(in_channels=4, out_channels=1, model_channels=128, attention_resolutions=[2, 4],
                num_res_blocks=1, channel_mult=[1, 2, 2], num_heads=8, context_dim=512)
CLS_TINY = dict(in_channels=4, out_channels=1, model_channels=64, attention_resolutions=[2, 4],
                num_res_blocks=1, channel_mult=[1, 2, 2], num_heads=4, context_dim=64)


# ----------------------------------------------------------------------------- key layout
def _conv(spec, p, cin, cout, k):
    spec[p + ".weight"] = (cout, cin, k, k)
    spec[p + ".bias"] = (cout,)


def _lin(spec, p, cin, cout, bias=True):
    spec[p + ".weight"] = (cout, cin)
    if bias:
        spec[p + ".bias"] = (cout,)


def _norm(spec, p, c):
    spec[p + ".weight"] = (c,)
    spec[p + ".bias"] = (c,)


def _res(spec, p, cin, cout, temb):
    _norm(spec, p + ".in_layers.0", cin)
    _conv(spec, p + ".in_layers.2", cin, cout, 3)
    _lin(spec, p + ".emb_layers.1", temb, cout)
    _norm(spec, p + ".out_layers.0", cout)
    _conv(spec, p + ".out_layers.3", cout, cout, 3)
    if cin != cout:
        _conv(spec, p + ".skip_connection", cin, cout, 1)


def _st(spec, p, c, ctx):
    _norm(spec, p + ".norm", c)
    _conv(spec, p + ".proj_in", c, c, 1)
    b = p + ".transformer_blocks.0"
    for a, cd in ((".attn1", c), (".attn2", ctx)):
        _lin(spec, b + a + ".to_q", c, c, bias=False)
        _lin(spec, b + a + ".to_k", cd, c, bias=False)
        _lin(spec, b + a + ".to_v", cd, c, bias=False)
        _lin(spec, b + a + ".to_out.0", c, c)
    _lin(spec, b + ".ff.net.0.proj", c, 8 * c)
    _lin(spec, b + ".ff.net.2", 4 * c, c)
    for n in (".norm1", ".norm2", ".norm3"):
        _norm(spec, b + n, c)
    _conv(spec, p + ".proj_out", c, c, 1)


def unet_spec(cfg, prefix="", encoder_only=False):
    """Key layout of UNetModel / Classifier_Backbone (openai_unetmodel.py:506-692)."""
    spec = OrderedDict()
    mc, mult, nrb = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    attn, ctx = set(cfg["attention_resolutions"]), cfg["context_dim"]
    temb = 4 * mc
    _lin(spec, prefix + "time_embed.0", mc, temb)
    _lin(spec, prefix + "time_embed.2", temb, temb)
    _conv(spec, prefix + "input_blocks.0.0", cfg["in_channels"], mc, 3)
    chans, ch, ds, idx = [mc], mc, 1, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            _res(spec, prefix + f"input_blocks.{idx}.0", ch, m * mc, temb)
            ch = m * mc
            if ds in attn:
                _st(spec, prefix + f"input_blocks.{idx}.1", ch, ctx)
            chans.append(ch)
            idx += 1
        if level != len(mult) - 1:
            _conv(spec, prefix + f"input_blocks.{idx}.0.op", ch, ch, 3)
            chans.append(ch)
            idx += 1
            ds *= 2
    _res(spec, prefix + "middle_block.0", ch, ch, temb)
    _st(spec, prefix + "middle_block.1", ch, ctx)
    _res(spec, prefix + "middle_block.2", ch, ch, temb)
    if encoder_only:                       # Classifier_Backbone head (alignment_backbone.py:630-638)
        last = mc * mult[-1]
        _norm(spec, prefix + "out.0", ch)
        _conv(spec, prefix + "out.2", last, last // 2, 3)
        _lin(spec, prefix + "classifier", last // 2, cfg["out_channels"])
        return spec
    idx = 0
    for level in reversed(range(len(mult))):
        for i in range(nrb + 1):
            ich = chans.pop()
            _res(spec, prefix + f"output_blocks.{idx}.0", ch + ich, mc * mult[level], temb)
            ch = mc * mult[level]
            j = 1
            if ds in attn:
                _st(spec, prefix + f"output_blocks.{idx}.{j}", ch, ctx)
                j += 1
            if level and i == nrb:
                _conv(spec, prefix + f"output_blocks.{idx}.{j}.conv", ch, ch, 3)
                ds //= 2
            idx += 1
    _norm(spec, prefix + "out.0", ch)
    _conv(spec, prefix + "out.2", mc, cfg["out_channels"], 3)
    return spec


def _vae_res(spec, p, cin, cout):
    _norm(spec, p + ".norm1", cin)
    _conv(spec, p + ".conv1", cin, cout, 3)
    _norm(spec, p + ".norm2", cout)
    _conv(spec, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(spec, p + ".nin_shortcut", cin, cout, 1)


def vae_decoder_spec(cfg, prefix=""):
    """post_quant_conv + Decoder key layout (stage1_autoencoder/model.py:557-628)."""
    spec = OrderedDict()
    ch, mult, nrb = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"]
    _conv(spec, prefix + "post_quant_conv", cfg["embed_dim"], cfg["z_channels"], 1)
    d = prefix + "decoder."
    bi = ch * mult[-1]
    _conv(spec, d + "conv_in", cfg["z_channels"], bi, 3)
    _vae_res(spec, d + "mid.block_1", bi, bi)
    _norm(spec, d + "mid.attn_1.norm", bi)
    for n in ("q", "k", "v", "proj_out"):
        _conv(spec, d + "mid.attn_1." + n, bi, bi, 1)
    _vae_res(spec, d + "mid.block_2", bi, bi)
    for lvl in reversed(range(len(mult))):
        bo = ch * mult[lvl]
        for ib in range(nrb + 1):
            _vae_res(spec, d + f"up.{lvl}.block.{ib}", bi, bo)
            bi = bo
        if lvl != 0:
            _conv(spec, d + f"up.{lvl}.upsample.conv", bi, bi, 3)
    _norm(spec, d + "norm_out", bi)
    _conv(spec, d + "conv_out", bi, cfg["out_ch"], 3)
    return spec


def cond_spec(cfg, prefix=""):
    spec = OrderedDict()
    _lin(spec, prefix + "embedder.0", cfg["origin_dim"], cfg["embed_dim"])
    spec[prefix + "pos_emb.weight"] = (cfg["seq_len"], cfg["embed_dim"])
    return spec


def state_dict_spec(unet=UNET_FULL, vae=VAE_FULL, cond=COND_FULL):
    """(name -> shape) for the LatentDiffusion sub-modules on the hot path."""
    spec = OrderedDict()
    spec.update(unet_spec(unet, "model.diffusion_model."))
    spec.update(vae_decoder_spec(vae, "first_stage_model."))
    spec.update(cond_spec(cond, "cond_stage_model."))
    return spec


CAVP_FULL = dict(stage_blocks=[3, 4, 6, 3], base_channels=64, embed_dim=512)
CAVP_TINY = dict(stage_blocks=[1, 1, 1, 1], base_channels=64, embed_dim=64)


def _cm3d(spec, p, cin, cout, k):
    spec[p + ".conv.weight"] = (cout, cin) + tuple(k)
    for n in ("weight", "bias", "running_mean", "running_var"):
        spec[p + ".bn." + n] = (cout,)


def cavp_spec(cfg=CAVP_FULL):
    """CAVP_Inference.state_dict() keys of the video branch: SlowOnly-R50 backbone (inference/model/cavp_modules.py:
    1233-1268, 757-779, 484-518, 167-330) + video_project_head (inference/model/cavp_model.py:27-29)."""
    spec = OrderedDict()
    base = cfg["base_channels"]
    _cm3d(spec, "video_encoder.conv1", 3, base, (1, 7, 7))
    inplanes = base
    for li, nb in enumerate(cfg["stage_blocks"]):
        planes = base * 2 ** li
        inflate = li >= 2                                   # SlowOnly: inflate=(0,0,1,1)
        for bi in range(nb):
            p = f"video_encoder.layer{li + 1}.{bi}"
            _cm3d(spec, p + ".conv1", inplanes, planes, (3, 1, 1) if inflate else (1, 1, 1))
            _cm3d(spec, p + ".conv2", planes, planes, (1, 3, 3))
            _cm3d(spec, p + ".conv3", planes, planes * 4, (1, 1, 1))
            if bi == 0:                                     # stride != 1 or inplanes != planes*4 holds for every stage
                _cm3d(spec, p + ".downsample", inplanes, planes * 4, (1, 1, 1))
            inplanes = planes * 4
    spec["video_project_head.weight"] = (cfg["embed_dim"], inplanes)
    spec["video_project_head.bias"] = (cfg["embed_dim"],)
    return spec


def synthetic_video(batch, frames=32, size=224, seed=77):
    """Frames as Extract_CAVP_Features feeds them (inference/demo_util.py:100-103,150-161): RGB in [0,1], (B,T,3,H,W).
    Smooth random fields (a few low-frequency sinusoids per channel) so that the 7x7 stem sees image-like input."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, size), torch.linspace(0, 1, size), indexing="ij")
    out = torch.zeros(batch, frames, 3, size, size)
    for b in range(batch):
        for c in range(3):
            k = torch.rand(4, 5, generator=g)
            for j in range(4):
                fx, fy, ph, sp, am = (k[j, 0] * 6, k[j, 1] * 6, k[j, 2] * 6.28, k[j, 3] * 2, 0.1 + 0.15 * k[j, 4])
                t = torch.arange(frames).view(-1, 1, 1) / 4.0
                out[b, :, c] += am * torch.sin(6.28 * (fx * xx + fy * yy) + ph + sp * t)
    out = out + 0.03 * torch.randn(out.shape, generator=g)
    return (out + 0.5).clamp_(0, 1)


def classifier_spec(cfg=CLS_FULL):
    """Alignment_Classifier_Double_Guidance.state_dict() keys for the backbone (``model.*``)."""
    return unet_spec(cfg, "model.", encoder_only=True)


# ----------------------------------------------------------------------------- values
def make_tensor(name, shape, seed=0):
    """Deterministic fp32 tensor for (name, shape, seed).

    ndim>=2: U(-1,1)*sqrt(3/fan_in)  (unit-gain, also for the reference's zero-initialised
    modules, which would otherwise make the whole network output exactly 0);
    1-D '.weight' (norm scales): 1 + 0.1*U;  1-D '.bias' / running_mean: 0.05*U;  running_var: 1 + 0.2*U;
    pos_emb: 0.05*U;  CAVP backbone convs: He-uniform U*sqrt(6/fan_in)."""
    key = [int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]
    g = np.random.Generator(np.random.Philox(key=key))
    n = int(np.prod(shape))
    u = g.random(n, dtype=np.float32)
    u *= 2.0
    u -= 1.0
    if len(shape) >= 2:
        if name.endswith("pos_emb.weight"):
            u *= 0.05
        elif name.startswith("video_encoder."):      # ReLU network: He-uniform; the residual branch's last conv at 1/2
            u *= np.float32(np.sqrt(6.0 / float(np.prod(shape[1:]))) * (0.5 if ".conv3." in name else 1.0))
        else:
            u *= np.float32(np.sqrt(3.0 / float(np.prod(shape[1:]))))
    elif name.endswith("running_var"):
        u *= 0.2
        u += 1.0
    elif name.endswith(".weight"):
        u *= 0.1
        u += 1.0
    else:
        u *= 0.05
    return torch.from_numpy(u.reshape(shape))


def make_state_dict(spec, seed=0):
    return OrderedDict((k, make_tensor(k, s, seed)) for k, s in spec.items())


def synthetic_cavp(batch, frames=32, dim=512, seed=1234):
    """CAVP-like features: N(0,1) rows, L2-normalised (encode_video(normalize=True),
    inference/model/cavp_model.py:62-63)."""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(batch, frames, dim, generator=g)
    return v / v.norm(dim=-1, keepdim=True)


def synthetic_xT(batch, seed=21, shape=(4, 16, 64), first_index=0):
    """x_T per *global sample index* (seed + index) so results do not depend on how the
    batch is sharded over ranks."""
    out = []
    for i in range(batch):
        g = torch.Generator().manual_seed(seed + first_index + i)
        out.append(torch.randn(shape, generator=g))
    return torch.stack(out)
