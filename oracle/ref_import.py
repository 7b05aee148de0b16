"""Import the *reference* Diff-Foley modules on CPU (build container only).

Used only by ``tests/golden/make_golden.py`` to generate the golden vectors that pin
the oracle.  ``/root/reference`` does not exist on the GPU box, so nothing that runs
there may import this module.  Recipe from SURVEY.md section 8(c): three stub packages
(omegaconf, pytorch_lightning, torchvision) injected into ``sys.modules`` and one
monkeypatch of the samplers' ``register_buffer`` (they hard-code ``.to("cuda")``,
ddim.py:21-25, plms.py:18-22, sampler.py:18-22).  No reference source is copied.
"""
import os
import sys
import types

import torch
import torch.nn as nn
import yaml

REF = os.environ.get("DIFF_FOLEY_REFERENCE", "/root/reference")


class AttrDict(dict):
    __getattr__ = dict.__getitem__

    @staticmethod
    def wrap(o):
        if isinstance(o, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in o.items()})
        if isinstance(o, list):
            return [AttrDict.wrap(v) for v in o]
        return o


def install_stubs():
    if "pytorch_lightning" in sys.modules and getattr(sys.modules["pytorch_lightning"], "_df_stub", False):
        return
    oc = types.ModuleType("omegaconf")
    lc = types.ModuleType("omegaconf.listconfig")

    class ListConfig(list):
        pass
    lc.ListConfig = ListConfig
    oc.listconfig = lc
    oc.ListConfig = ListConfig
    sys.modules["omegaconf"] = oc
    sys.modules["omegaconf.listconfig"] = lc

    pl = types.ModuleType("pytorch_lightning")
    pl._df_stub = True

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass
    pl.LightningModule = LightningModule
    ut = types.ModuleType("pytorch_lightning.utilities")
    dist = types.ModuleType("pytorch_lightning.utilities.distributed")
    dist.rank_zero_only = lambda f: f
    ut.distributed = dist
    pl.utilities = ut
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.utilities"] = ut
    sys.modules["pytorch_lightning.utilities.distributed"] = dist

    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tv.utils = tvu
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.utils"] = tvu

    if REF not in sys.path:
        sys.path.insert(0, REF)


def import_reference():
    """Returns a namespace with the reference classes needed for golden generation."""
    install_stubs()
    from diff_foley.models.diffusion import ddpm as ref_ddpm
    from diff_foley.models.diffusion.ddim import DDIMSampler
    from diff_foley.models.diffusion.plms import PLMSSampler
    from diff_foley.models.diffusion.dpm_solver import DPMSolverSampler
    for cls in (DDIMSampler, PLMSSampler, DPMSolverSampler):
        cls.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    ns = types.SimpleNamespace(ddpm=ref_ddpm, LatentDiffusion=ref_ddpm.LatentDiffusion,
                               DDIMSampler=DDIMSampler, PLMSSampler=PLMSSampler,
                               DPMSolverSampler=DPMSolverSampler)
    from diff_foley.modules.diffusionmodules import util as ref_util
    from diff_foley.models.diffusion.dpm_solver import dpm_solver as ref_dpm
    ns.util = ref_util
    ns.dpm = ref_dpm
    return ns


def load_ldm_config(unet=None, vae=None, cond=None):
    """inference/config/Stage2_LDM.yaml as an attr-dict, optionally with the three
    sub-configs shrunk to a tiny variant (same code path, fewer channels)."""
    with open(os.path.join(REF, "inference/config/Stage2_LDM.yaml")) as f:
        cfg = yaml.safe_load(f)["model"]["params"]
    if unet:
        cfg["unet_config"]["params"].update(unet)
    if vae:
        dd = cfg["first_stage_config"]["params"]["ddconfig"]
        dd.update({k: v for k, v in vae.items() if k in dd or k in ("ch", "ch_mult", "num_res_blocks")})
        cfg["first_stage_config"]["params"]["embed_dim"] = vae.get("embed_dim", 4)
    if cond:
        cfg["cond_stage_config"]["params"].update(cond)
    return AttrDict.wrap(cfg)


def build_reference_ldm(cfg, state_dict):
    ns = import_reference()
    torch.manual_seed(0)
    model = ns.LatentDiffusion(**cfg)
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    assert not unexpected, unexpected
    bad = [k for k in missing if not (k.startswith("first_stage_model.encoder") or
                                      k.startswith("first_stage_model.quant_conv") or
                                      k.startswith("first_stage_model.loss") or
                                      k in ("logvar",) or k.startswith("lvlb") or k.endswith("alphas_cumprod")
                                      or "betas" in k or "posterior" in k or "alphas" in k)]
    assert not bad, bad[:10]
    return model.eval(), ns


def build_reference_classifier(cls_cfg, state_dict):
    """Alignment classifier backbone (alignment_backbone.py:417)."""
    install_stubs()
    from diff_foley.modules.double_guidance.alignment_backbone import Classifier_Backbone
    torch.manual_seed(0)
    m = Classifier_Backbone(image_size=32, use_spatial_transformer=True, transformer_depth=1,
                            use_checkpoint=True, legacy=False, **cls_cfg)
    n = len("model.")
    missing, unexpected = m.load_state_dict({k[n:]: v for k, v in state_dict.items()}, strict=True)
    return m.eval()


def import_reference_cavp():
    """Import the reference's ``CAVP_Inference`` (inference/model/cavp_model.py) on CPU.

    ``mmcv`` 1.7.1 is a third-party dependency that is neither under /root/reference nor installed, so the few names
    cavp_modules.py:8-18 imports from it are provided by a DECLARED STAND-IN: ``ConvModule`` = Conv3d(bias as given) ->
    BatchNorm3d (registered as ``.bn``) -> ReLU when ``act_cfg`` is not None, which is mmcv's documented behaviour for
    ``conv_cfg=dict(type='Conv3d'), norm_cfg=dict(type='BN3d')`` with the default order ('conv','norm','act').  The
    reference's own code then decides every kernel size, stride, inflation and downsample (the topology the oracle
    restates); the numerical behaviour of ConvModule itself is NOT pinned by the reference (parity unpinned at the
    mmcv boundary).  Used only by tests/golden/make_golden.py --cavp."""
    cnn = types.ModuleType("mmcv.cnn")

    class ConvModule(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                     bias="auto", conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), **kw):
            super().__init__()
            assert conv_cfg is not None and conv_cfg["type"] == "Conv3d"
            with_norm = norm_cfg is not None
            if bias == "auto":
                bias = not with_norm
            self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                  dilation=dilation, groups=groups, bias=bias)
            if with_norm:
                assert norm_cfg["type"] == "BN3d"
                self.bn = nn.BatchNorm3d(out_channels)
            self.with_norm = with_norm
            self.act = nn.ReLU() if act_cfg is not None else None

        @property
        def norm(self):
            return self.bn

        def forward(self, x):
            x = self.conv(x)
            if self.with_norm:
                x = self.bn(x)
            return self.act(x) if self.act is not None else x

    cnn.ConvModule = ConvModule
    cnn.NonLocal3d = None
    cnn.build_activation_layer = lambda cfg: nn.ReLU()
    cnn.constant_init = lambda *a, **k: None
    cnn.kaiming_init = lambda *a, **k: None
    runner = types.ModuleType("mmcv.runner")
    runner._load_checkpoint = runner.load_checkpoint = lambda *a, **k: None
    utils = types.ModuleType("mmcv.utils")
    utils.print_log = lambda *a, **k: None
    utils._BatchNorm = nn.modules.batchnorm._BatchNorm
    mm = types.ModuleType("mmcv")
    mm.cnn, mm.runner, mm.utils = cnn, runner, utils
    sys.modules.update({"mmcv": mm, "mmcv.cnn": cnn, "mmcv.runner": runner, "mmcv.utils": utils})
    inf = os.path.join(REF, "inference")
    if inf not in sys.path:
        sys.path.insert(0, inf)
    from model.cavp_model import CAVP_Inference
    return CAVP_Inference
