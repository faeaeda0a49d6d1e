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
    # a gizmo.obj written from the committed fixture (one triangle per face, one material per colour)
    g = np.load(os.path.join(ROOT, "tests", "golden", "gizmo.npz"))
    gvx, gix = g["vertices"], g["indices"]
    colours = sorted({tuple(c) for c in gvx[:, 3:6].tolist()})
    with open(tmp_path / "g.mtl", "w") as f:
        for k, c in enumerate(colours):
            f.write(f"newmtl m{k}\nKd {c[0]!r} {c[1]!r} {c[2]!r}\n")
    with open(tmp_path / "g.obj", "w") as f:
        f.write("mtllib g.mtl\n")
        for v in gvx:
            f.write(f"v {float(v[0])!r} {float(v[1])!r} {float(v[2])!r}\nvn {float(v[6])!r} {float(v[7])!r} {float(v[8])!r}\n")
        for t in gix.reshape(-1, 3):
            f.write(f"usemtl m{colours.index(tuple(gvx[t[0], 3:6].tolist()))}\n")
            f.write("f " + " ".join(f"{i + 1}//{i + 1}" for i in t) + "\n")
    cmd = [str(exe), "--vertices-bin", str(tmp_path / "ball.bin"), "--size", str(W), str(H), "--frames", "5", "--tone-map", "1.5",
           "--gizmo", str(tmp_path / "g.obj"), "--out", str(tmp_path / "f.ppm")] + (["--deferred"] if deferred else [])
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
    if deferred:
        hdr, _, _, depth, _ = bbo.render_deferred(osc)
    else:
        hdr, _, depth, _ = bbo.render(osc)
    gv = np.zeros(len(gvx), bbo.GIZMO_VERTEX_DTYPE)
    gv["pos"], gv["color"], gv["normal"] = gvx[:, 0:3], gvx[:, 3:6], gvx[:, 6:9]
    # the OBJ round trip expands to one vertex per corner, which draws the same triangles
    want, _ = bbo.overlay(osc.frame, osc.view, depth, bbo.present(hdr, 1, 1.5), gv, gix, 100)
    assert np.array_equal(got, want[..., :3])


@pytest.mark.parametrize("ranks,form", [(3, "packed"), (8, "rgba8"), (2, "rgba32f"), (4, "rgba16f")])
def test_multi_gpu_demo_completes_the_frame_with_peer_pushes(tmp_path, ranks, form):
    """examples/multi_gpu_demo.cpp: one C++ process, `ranks` contexts (all on this box's one GPU), every rank renders its
    bands and pushes its block into every rank's gather buffer (bbr_push_shard), bbr_unpack_whole: the last rank's whole
    frame, presented, must be the oracle's presented image of the unpartitioned frame.  (rgba16f -- the reference's HDR
    attachment format on the wire -- gives the SAME image: presentation rounds to binary16 first, and that is idempotent.)"""
    exe = tmp_path / "mgdemo"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "multi_gpu_demo.cpp"), "-L" + os.path.join(ROOT, "bibim_renderer_amd"),
                           "-lbibim_hip", "-Wl,-rpath," + os.path.join(ROOT, "bibim_renderer_amd"), "-o", str(exe)])
    ball = S.load_shaderball_vertices()
    (tmp_path / "ball.bin").write_bytes(np.ascontiguousarray(ball).tobytes())
    W, H = 640, 360
    cmd = [str(exe), "--vertices-bin", str(tmp_path / "ball.bin"), "--size", str(W), str(H), "--grid", "4", "--frames", "4",
           "--ranks", str(ranks), "--form", form, "--tone-map", "1.2", "--out", str(tmp_path / "f.ppm")]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    assert f"over {ranks} ranks" in out and "Mpixels/s" in out
    raw = (tmp_path / "f.ppm").read_bytes()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert raw.startswith(header)
    got = np.frombuffer(raw[len(header):], np.uint8).reshape(H, W, 3)
    # the same scene for the oracle: the C3 layout (16 balls, camera of the demo) with the scene class's default lights and
    # the default material, normal map on, tone map on with exposure 1.2
    osc = scenes.shaderball_scene(configs.C3.scaled(W, H, 64), bbo.MaterialData())
    osc.frame = scenes.frame_uniforms(scenes.reference_default_lights(), 1, 1.2)
    osc.view["enable_normal_map"] = 1
    hdr, _, _, _ = bbo.render(osc)
    want = bbo.present(hdr, 1, 1.2)
    assert np.array_equal(got, want[..., :3])
