import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- before libbibim_hip.so: torch bundles its own HIP runtime and must load it first

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_once():
    """The oracle is rebuilt from source if stale; the HIP library is NOT built here (it travels prebuilt
    to the GPU box) unless it is missing and hipcc exists."""
    from oracle import bbo
    bbo.build()
    lib = os.path.join(ROOT, "bibim_renderer_amd", "libbibim_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def ball_vertices():
    from oracle import scenes
    return scenes.load_shaderball_vertices()


@pytest.fixture(scope="session")
def maps64():
    from bibim_renderer_amd import textures
    return textures.make_material(64)


@pytest.fixture(scope="session")
def maps256():
    from bibim_renderer_amd import textures
    return textures.make_material(256)


def assert_frame_close(img, ref, tol=1e-4):
    """BASELINE tolerance: |d| <= 1e-4 * max(1, |ref|) per channel; NaNs must sit at the same places."""
    img = np.asarray(img); ref = np.asarray(ref)
    assert img.shape == ref.shape
    nan_i, nan_r = np.isnan(img), np.isnan(ref)
    assert np.array_equal(nan_i, nan_r), f"NaN placement differs at {int((nan_i != nan_r).sum())} values"
    ok = ~nan_r
    d = np.abs(img[ok] - ref[ok])
    lim = tol * np.maximum(1.0, np.abs(ref[ok]))
    bad = int((d > lim).sum())
    assert bad == 0, f"{bad} channel values outside {tol}*max(1,|ref|); max |diff| = {float(d.max())}"


_ROUTES = {"short-frame route": {}, "long-frame route": {"no_tail_items": 0},
           "long-frame route, heavy tiles first": {"no_tail_items": 0, "heavy_tiles": 4}}


def _route(request, monkeypatch):
    """every Renderer the test creates gets the route's options right after construction (a wrapper around the constructor, put
    there by the fixture: the product code has no hook for it)"""
    from bibim_renderer_amd import renderer as R
    options = _ROUTES[request.param]
    plain_init = R.Renderer.__init__

    def init(self, *args, **kwargs):
        plain_init(self, *args, **kwargs)
        for name, value in options.items():
            self.set_option(name, value)
    monkeypatch.setattr(R.Renderer, "__init__", init)
    return request.param


@pytest.fixture(params=list(_ROUTES)[:2])
def item_route(request, monkeypatch):
    """Both ways a frame's shading work list comes about.  A frame with at most `no_tail_items` item slots --
    nearly every frame of this suite -- has its raster tiles append their items themselves and is shaded by one full-coverage
    launch; BASELINE's 4K / 8K frames take the other route: k_shade_items scans the tiles in screen order, the main launch
    is sized from the slot's previous frame and a tail launch covers the rest.  With no_tail_items = 0 a small frame takes the
    long route too."""
    return _route(request, monkeypatch)


@pytest.fixture(params=list(_ROUTES))
def item_route_heavy(request, monkeypatch):
    """item_route plus a third form (round 4): the long route with k_raster starting its heavy tiles first (option heavy_tiles;
    the automatic choice while one frame is in flight), with a threshold low enough (4 references in a bin) for these small
    frames to have heavy tiles.  For the tests that reach k_raster's route decision -- the deferred pass, partitions, the
    overflow / replay paths (round 5: the others ran it to no purpose, a third of the suite's time)."""
    return _route(request, monkeypatch)
