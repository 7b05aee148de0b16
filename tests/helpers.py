"""Shared helpers for the parity tests (seeded inputs, golden loading, oracle closures)."""
import os

import numpy as np
import torch

import diff_foley_amd  # noqa: F401
from diff_foley_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rnd(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def gold(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        import pytest
        pytest.skip(f"golden fixture {name} not generated")
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(path).items()}


_cache = {}


def tiny_state_dict(seed=0):
    if ("tiny", seed) not in _cache:
        spec = synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
        _cache[("tiny", seed)] = synth.make_state_dict(spec, seed)
    return _cache[("tiny", seed)]


def full_state_dict(seed=0):
    if ("full", seed) not in _cache:
        _cache[("full", seed)] = synth.make_state_dict(synth.state_dict_spec(), seed)
    return _cache[("full", seed)]


def tiny_classifier_sd(seed=0):
    if ("cls_tiny", seed) not in _cache:
        _cache[("cls_tiny", seed)] = synth.make_state_dict(synth.classifier_spec(synth.CLS_TINY), seed)
    return _cache[("cls_tiny", seed)]


def full_classifier_sd(seed=0):
    if ("cls_full", seed) not in _cache:
        _cache[("cls_full", seed)] = synth.make_state_dict(synth.classifier_spec(synth.CLS_FULL), seed)
    return _cache[("cls_full", seed)]


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def fuzz_seeds(n):
    """Seeds of a randomised differential test: 0 .. n - 1 in the suite; ``DF_FUZZ_SEED0`` / ``DF_FUZZ_CASES`` move and widen the
    range for an exploratory sweep on a GPU box (every case is a function of its seed alone, so a failing seed reproduces)."""
    s0 = int(os.environ.get("DF_FUZZ_SEED0", "0"))
    return range(s0, s0 + int(os.environ.get("DF_FUZZ_CASES", str(n))))
