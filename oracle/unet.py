"""Oracle: UNetModel forward, functional torch-fp32 restatement.  Test infrastructure only.

Operates directly on a reference-layout ``state_dict`` (keys as in
``model.diffusion_model.*`` with the prefix stripped) so no module tree is
built.  Follows

  * timestep_embedding                 diff_foley/modules/diffusionmodules/util.py:151-171
  * UNetModel.__init__ topology        diff_foley/modules/diffusionmodules/openai_unetmodel.py:506-692
  * UNetModel.forward                  openai_unetmodel.py:710-742
  * ResBlock._forward                  openai_unetmodel.py:255-275   (GroupNorm32 eps 1e-5, util.py:214-216)
  * Downsample / Upsample              openai_unetmodel.py:100-119, 143-160
  * SpatialTransformer.forward         diff_foley/modules/diffusionmodules/attention_openai.py:250-261 (GN eps 1e-6)
  * BasicTransformerBlock._forward     attention_openai.py:211-215
  * CrossAttention.forward             attention_openai.py:170-193
  * GEGLU / FeedForward                attention_openai.py:37-64
"""
import math
import torch
import torch.nn.functional as F

UNET_FULL = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                 num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, context_dim=768)
UNET_TINY = dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1],
                 num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=2, context_dim=128)
# Classifier_Backbone hyper-parameters (inference/config/Double_Guidance_Classifier.yaml:35-50)
CLS_FULL = dict(in_channels=4, out_channels=1, model_channels=128, attention_resolutions=[2, 4],
                num_res_blocks=1, channel_mult=[1, 2, 2], num_heads=8, context_dim=512)
CLS_TINY = dict(in_channels=4, out_channels=1, model_channels=64, attention_resolutions=[2, 4],
                num_res_blocks=1, channel_mult=[1, 2, 2], num_heads=4, context_dim=64)


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(x, sd, p, eps):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(x, sd, p, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def resblock(sd, p, x, emb):
    h = _conv(F.silu(_gn(x, sd, p + ".in_layers.0", 1e-5)), sd, p + ".in_layers.2")
    e = _lin(F.silu(emb), sd, p + ".emb_layers.1")
    h = h + e[:, :, None, None]
    h = _conv(F.silu(_gn(h, sd, p + ".out_layers.0", 1e-5)), sd, p + ".out_layers.3")
    if (p + ".skip_connection.weight") in sd:
        x = _conv(x, sd, p + ".skip_connection", padding=0)
    return x + h


def attention(sd, p, x, ctx, heads):
    q = _lin(x, sd, p + ".to_q")
    ctx = x if ctx is None else ctx
    k = _lin(ctx, sd, p + ".to_k")
    v = _lin(ctx, sd, p + ".to_v")
    b, n, c = q.shape
    d = c // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)
    return _lin(out, sd, p + ".to_out.0")


def transformer_block(sd, p, x, ctx, heads):
    ln = lambda t, q: F.layer_norm(t, (t.shape[-1],), sd[q + ".weight"], sd[q + ".bias"], 1e-5)
    x = attention(sd, p + ".attn1", ln(x, p + ".norm1"), None, heads) + x
    x = attention(sd, p + ".attn2", ln(x, p + ".norm2"), ctx, heads) + x
    y = _lin(ln(x, p + ".norm3"), sd, p + ".ff.net.0.proj")
    a, g = y.chunk(2, dim=-1)
    x = _lin(a * F.gelu(g), sd, p + ".ff.net.2") + x
    return x


def spatial_transformer(sd, p, x, ctx, heads):
    b, c, h, w = x.shape
    y = _conv(_gn(x, sd, p + ".norm", 1e-6), sd, p + ".proj_in", padding=0)
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    y = transformer_block(sd, p + ".transformer_blocks.0", y, ctx, heads)
    y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
    return _conv(y, sd, p + ".proj_out", padding=0) + x


def unet_layout(cfg):
    """Block list derived from the constructor loops (openai_unetmodel.py:516-680).

    Returns (input_blocks, middle, output_blocks); each block is a list of
    (kind, prefix) with kind in {conv, res, st, down, up}."""
    mc, mult, nrb = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    attn = set(cfg["attention_resolutions"])
    inp = [[("conv", "input_blocks.0.0")]]
    ds, idx = 1, 1
    for level, _ in enumerate(mult):
        for _ in range(nrb):
            blk = [("res", f"input_blocks.{idx}.0")]
            if ds in attn:
                blk.append(("st", f"input_blocks.{idx}.1"))
            inp.append(blk)
            idx += 1
        if level != len(mult) - 1:
            inp.append([("down", f"input_blocks.{idx}.0")])
            idx += 1
            ds *= 2
    mid = [("res", "middle_block.0"), ("st", "middle_block.1"), ("res", "middle_block.2")]
    out, idx = [], 0
    for level in reversed(range(len(mult))):
        for i in range(nrb + 1):
            blk = [("res", f"output_blocks.{idx}.0")]
            j = 1
            if ds in attn:
                blk.append(("st", f"output_blocks.{idx}.{j}"))
                j += 1
            if level and i == nrb:
                blk.append(("up", f"output_blocks.{idx}.{j}"))
                ds //= 2
            out.append(blk)
            idx += 1
    return inp, mid, out


def _run_block(sd, blk, h, emb, ctx, heads):
    for kind, p in blk:
        if kind == "conv":
            h = _conv(h, sd, p)
        elif kind == "res":
            h = resblock(sd, p, h, emb)
        elif kind == "st":
            h = spatial_transformer(sd, p, h, ctx, heads)
        elif kind == "down":
            h = _conv(h, sd, p + ".op", stride=2)
        elif kind == "up":
            h = _conv(F.interpolate(h, scale_factor=2, mode="nearest"), sd, p + ".conv")
    return h


def time_embed(sd, cfg, t):
    e = timestep_embedding(t, cfg["model_channels"])
    return _lin(F.silu(_lin(e, sd, "time_embed.0")), sd, "time_embed.2")


@torch.no_grad()
def unet_forward(sd, cfg, x, t, context):
    """UNetModel.forward (openai_unetmodel.py:710-742).  ``sd`` keys are relative to
    ``model.diffusion_model.``; ``t`` may be long or float (DPM path)."""
    heads = cfg["num_heads"]
    inp, mid, out = unet_layout(cfg)
    emb = time_embed(sd, cfg, t)
    hs, h = [], x.float()
    for blk in inp:
        h = _run_block(sd, blk, h, emb, context, heads)
        hs.append(h)
    h = _run_block(sd, mid, h, emb, context, heads)
    for blk in out:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, blk, h, emb, context, heads)
    return _conv(F.silu(_gn(h, sd, "out.0", 1e-5)), sd, "out.2")


def classifier_forward(sd, cfg, x, t, context):
    """Classifier_Backbone.forward (diff_foley/modules/double_guidance/alignment_backbone.py:656-686):
    encoder half + middle of the UNet, then GN->SiLU->conv3x3->avgpool->Linear->sigmoid
    (:630-638).  Differentiable (no no_grad) because guidance needs d/dx."""
    heads = cfg["num_heads"]
    inp, mid, _ = unet_layout(cfg)
    emb = time_embed(sd, cfg, t)
    h = x.float()
    for blk in inp:
        h = _run_block(sd, blk, h, emb, context, heads)
    h = _run_block(sd, mid, h, emb, context, heads)
    h = _conv(F.silu(_gn(h, sd, "out.0", 1e-5)), sd, "out.2")
    h = h.mean(dim=(2, 3))
    return torch.sigmoid(_lin(h, sd, "classifier"))


def sub_state_dict(sd, prefix):
    """Strip ``prefix`` from matching keys (e.g. 'model.diffusion_model.')."""
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}
