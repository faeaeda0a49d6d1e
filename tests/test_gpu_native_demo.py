"""The C++ demo (examples/shaderball_demo.cpp) -- bb::ShaderBallScene + bb::drawFrame + bbr_present driven natively, no
Python in the loop -- must write the image the oracle computes for the reference's default ShaderBall scene."""
import os
import subprocess

import numpy as np
import pytest

from bibim_renderer_amd import configs, scene as S
from oracle import bbo, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("deferred", [False, True])
def test_native_demo_writes_the_oracles_image(tmp_path, deferred):
    exe = tmp_path / "demo"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "shaderball_demo.cpp"), "-L" + os.path.join(ROOT, "bibim_renderer_amd"),
                           "-lbibim_hip", "-Wl,-rpath," + os.path.join(ROOT, "bibim_renderer_amd"), "-o", str(exe)])
    ball = S.load_shaderball_vertices()
    (tmp_path / "ball.bin").write_bytes(np.ascontiguousarray(ball).tobytes())
    W, H = 480, 270
    cmd = [str(exe), "--vertices-bin", str(tmp_path / "ball.bin"), "--size", str(W), str(H), "--frames", "5", "--tone-map", "1.5",
           "--out", str(tmp_path / "f.ppm")] + (["--deferred"] if deferred else [])
    out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    assert "Mpixels/s" in out and ("deferred" if deferred else "forward") in out
    raw = (tmp_path / "f.ppm").read_bytes()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert raw.startswith(header)
    got = np.frombuffer(raw[len(header):], np.uint8).reshape(H, W, 3)
    # the same scene for the oracle: reference defaults (1 ball, its 3 lights, camera at the origin), default material,
    # normal map on (the demo turns it on), tone map on with exposure 1.5
    osc = scenes.shaderball_scene(configs.C2.scaled(W, H, 64), bbo.MaterialData())
    osc.frame = scenes.frame_uniforms(scenes.reference_default_lights(), 1, 1.5)
    osc.view["enable_normal_map"] = 1
    hdr = bbo.render_deferred(osc)[0] if deferred else bbo.render(osc)[0]
    want = bbo.present(hdr, 1, 1.5)[..., :3]
    assert np.array_equal(got, want)
