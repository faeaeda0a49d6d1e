"""The C-ABI library loads without a GPU and exports every symbol the public headers declare; without a
device the compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from bibim_renderer_amd import _capi


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    return sorted(set(re.findall(r"\b(bb[rs]_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    L = C.CDLL(_capi.LIB_PATH)
    names = declared("bibim_hip.h") + declared("bibim_scene.h")
    assert len(names) >= 45
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ but not exported"
    bound = set(_capi.SIGNATURES) | set(_capi.SCENE_SIGNATURES)
    assert set(names) == bound, f"binding table and headers disagree: {set(names) ^ bound}"


def test_library_path_is_in_tree():
    assert os.path.dirname(_capi.LIB_PATH) == os.path.join(ROOT, "bibim_renderer_amd")


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _capi.lib()
    assert L.bbr_device_count() == 0
    ctx = C.c_void_p()
    rc = L.bbr_create(64, 64, 0, C.byref(ctx))
    assert rc == -2 and not ctx.value  # BBR_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.bbr_last_error(None)
    from bibim_renderer_amd import BibimError, Renderer
    with pytest.raises(BibimError):
        Renderer(64, 64)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bibim_renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert not re.search(r"#\s*include[^\n]*oracle", txt), f
                assert "bb_oracle" not in txt and "libbb_ref" not in txt and "bbo_" not in txt, f
