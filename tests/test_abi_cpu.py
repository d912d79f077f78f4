"""CPU-side checks of the C-ABI boundary: the library loads, exports every
symbol include/srlhip.h declares, and fails loudly (no fallback) without a GPU."""
import ctypes
import os
import re

import pytest

from srlhip import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "srlhip.h")).read()
    return sorted(set(re.findall(r"\b(srlhip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(declared) == sorted(_lib.EXPORTS)
    assert lib.srlhip_abi_version() == _lib.ABI_VERSION == 5


def test_config_struct_matches_header():
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    assert cfg.struct_size == ctypes.sizeof(_lib.Config)
    assert (cfg.is_discrete, cfg.force_down, cfg.action_repeat, cfg.img_h, cfg.img_w) == (1, 1, 1, 224, 224)
    assert cfg.max_distance == 0.8
    assert _lib.default_config(_lib.ENV_MOBILE).max_distance == 1.6


def test_create_rejects_bad_configs_without_touching_a_gpu():
    lib = _lib.load()
    h = ctypes.c_void_p()
    cfg = _lib.default_config(_lib.ENV_MOBILE)
    cfg.struct_size = 4
    assert lib.srlhip_create(ctypes.byref(cfg), ctypes.byref(h)) == -22
    cfg = _lib.default_config(_lib.ENV_MOBILE_1D)
    cfg.is_discrete = 0
    assert lib.srlhip_create(ctypes.byref(cfg), ctypes.byref(h)) == -95
    assert b"Only discrete actions" in lib.srlhip_last_error(None)
    cfg = _lib.default_config(_lib.ENV_MOBILE)
    cfg.rng_mode = _lib.RNG_HOST          # auto_reset defaults to 1
    assert lib.srlhip_create(ctypes.byref(cfg), ctypes.byref(h)) == -22


def test_no_cpu_fallback_when_no_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.SrlHipError):
        _lib.Handle(_lib.default_config(_lib.ENV_MOBILE))
