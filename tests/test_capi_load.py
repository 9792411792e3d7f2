"""The C-ABI library loads and exports every symbol include/nuzero_amd.h
declares.  No compute; CPU only."""
import os
import re

from nuzero_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported_and_bound():
    text = open(os.path.join(REPO, "include", "nuzero_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(nz_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(_lib.lib, name), f"{name} not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.lib.nz_version().startswith(b"nuzero_amd")


def test_create_without_gpu_fails_loudly():
    import ctypes
    import torch
    if torch.cuda.is_available():
        return
    from nuzero_amd.search_config import legacy_ttt_search_config, to_struct
    h = ctypes.c_void_p(0)
    cfg = to_struct(legacy_ttt_search_config(), True)
    game = _lib.GameDesc(0, 2)
    st = _lib.lib.nz_engine_create(ctypes.byref(h), ctypes.byref(cfg), ctypes.byref(game), 4, 0)
    assert st == _lib.NZ_ERR_HIP and not h.value
    assert b"no HIP device" in _lib.lib.nz_last_error(None)
