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
