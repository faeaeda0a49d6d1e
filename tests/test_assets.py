"""Asset ingestion (include/bibim_assets.h): the C++ FBX / OBJ / PNG readers against
  * files written by the encoders in this test (every PNG colour type, bit depth, filter, Adam7, tRNS; a binary FBX
    with zlib arrays, Direct and IndexToDirect layers; an OBJ with negative indices and polygons),
  * the in-repo Python readers that minted the committed fixtures (tools/fbx_geometry.py, tools/obj_loader.py),
  * and, where the reference is mounted (authoring container only), its real assets against the committed fixtures:
    ShaderBall.fbx -> shaderball_vertices.npz, gizmo.obj -> gizmo.npz, every PNG -> the hash of what the reference's own
    stb_image 2.25 decodes (tests/golden/reference_png_sha256.json).
Host-only functions: no GPU needed."""
import hashlib
import json
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from bibim_renderer_amd import assets

REF = os.environ.get("BB_REFERENCE", "/root/reference")
have_ref = os.path.isdir(os.path.join(REF, "resources"))


# ------------------------------------------------------------------------------------------------ PNG
def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _filter_rows(rows, bpp, start):
    """rows: list of bytes; returns the filtered stream, cycling through the five filter types"""
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = (start + y) % 5
        out.append(ft)
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i] if len(prev) == len(row) else 0
            c = prev[i - bpp] if i >= bpp and len(prev) == len(row) else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ft]
            out.append((v - pred) & 255)
        prev = row
    return bytes(out)


def _pack_rows(samples, depth):
    """samples [h, w*channels] ints at file precision -> list of packed row bytes"""
    rows = []
    for r in samples:
        if depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in r))
        elif depth == 8:
            rows.append(bytes(int(v) for v in r))
        else:
            bits = "".join(format(int(v), f"0{depth}b") for v in r)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


_ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def encode_png(samples, color, depth, palette=None, trns=None, interlace=False, split_idat=1):
    """samples: int array [h, w, channels] at file precision"""
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    stream = b""
    passes = _ADAM7 if interlace else [(0, 0, 1, 1)]
    for k, (x0, y0, dx, dy) in enumerate(passes):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        stream += _filter_rows(_pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, start=k)
    z = zlib.compress(stream, 6)
    parts = [z[i * len(z) // split_idat:(i + 1) * len(z) // split_idat] for i in range(split_idat)]
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
    out += _chunk(b"gAMA", struct.pack(">I", 45455))  # an ancillary chunk to skip
    if palette is not None:
        out += _chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    if trns is not None:
        out += _chunk(b"tRNS", trns)
    for p in parts:
        out += _chunk(b"IDAT", p)
    return out + _chunk(b"IEND", b"")


def expected_rgba(samples, color, depth, palette=None, trns=None):
    """the conversions of stbi_load(..., STBI_rgb_alpha)"""
    h, w, ch = samples.shape
    s = samples.astype(np.int64)
    scale = {1: 255, 2: 85, 4: 17, 8: 1, 16: 1}[depth] if color == 0 else 1
    to8 = (lambda v: v >> 8) if depth == 16 else (lambda v: v * scale)
    out = np.zeros((h, w, 4), np.uint8)
    if color == 3:
        pal = np.zeros((256, 4), np.uint8)
        pal[:, 3] = 255
        pal[:len(palette), :3] = palette
        if trns is not None:
            pal[:len(trns), 3] = list(trns)
        out[:] = pal[s[..., 0]]
    elif color in (0, 4):
        out[..., 0] = out[..., 1] = out[..., 2] = to8(s[..., 0])
        out[..., 3] = to8(s[..., 1]) if color == 4 else 255
        if trns is not None:
            key = struct.unpack(">H", trns)[0]
            out[..., 3] = np.where(s[..., 0] == key if depth == 16 else out[..., 0] == ((key & 255) * scale) & 255, 0, out[..., 3])
    else:
        for c in range(3):
            out[..., c] = to8(s[..., c])
        out[..., 3] = to8(s[..., 3]) if color == 6 else 255
        if trns is not None:
            key = struct.unpack(">HHH", trns)
            m = np.ones((h, w), bool)
            for c in range(3):
                m &= (s[..., c] == key[c]) if depth == 16 else (out[..., c] == (key[c] & 255))
            out[..., 3] = np.where(m, 0, out[..., 3])
    return out


CASES = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("color,depth", CASES)
@pytest.mark.parametrize("interlace", [False, True])
def test_png_every_colour_type_depth_filter_and_interlace(color, depth, interlace):
    rng = np.random.default_rng(color * 100 + depth + 7 * interlace)
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    for (w, h) in ((1, 1), (7, 5), (33, 19)):
        hi = 1 << depth
        palette = rng.integers(0, 256, (min(hi, 256), 3)) if color == 3 else None
        samples = rng.integers(0, hi, (h, w, ch))
        data = encode_png(samples, color, depth, palette, interlace=interlace, split_idat=1 + (w % 3))
        got = assets.decode_png(data)
        assert np.array_equal(got, expected_rgba(samples, color, depth, palette)), (color, depth, interlace, w, h)


def test_png_transparency_chunks():
    rng = np.random.default_rng(3)
    # colour key on grey (sub-byte and 16-bit) and RGB; palette alpha
    for color, depth, key in ((0, 4, struct.pack(">H", 9)), (0, 16, struct.pack(">H", 0x1234)), (2, 8, struct.pack(">HHH", 10, 20, 30)),
                              (2, 16, struct.pack(">HHH", 0x0102, 0x0304, 0x0506))):
        ch = 1 if color == 0 else 3
        samples = rng.integers(0, 1 << depth, (9, 11, ch))
        keyv = struct.unpack(">" + "H" * ch, key)
        samples[2, 3] = keyv
        samples[8, 10] = keyv
        got = assets.decode_png(encode_png(samples, color, depth, trns=key))
        want = expected_rgba(samples, color, depth, trns=key)
        assert np.array_equal(got, want) and got[2, 3, 3] == 0 and got[8, 10, 3] == 0
    palette = rng.integers(0, 256, (16, 3))
    samples = rng.integers(0, 16, (6, 6, 1))
    trns = bytes([0, 128, 255, 7])
    assert np.array_equal(assets.decode_png(encode_png(samples, 3, 4, palette, trns)), expected_rgba(samples, 3, 4, palette, trns))


def test_png_rejects_garbage():
    with pytest.raises(assets.AssetError):
        assets.decode_png(b"not a png at all")
    good = encode_png(np.zeros((4, 4, 3), int), 2, 8)
    with pytest.raises(assets.AssetError):
        assets.decode_png(good[:40])
    with pytest.raises(assets.AssetError):
        assets.load_png("/nonexistent/file.png")
    bad_filter = bytearray(zlib.decompress(b"".join([])) if False else b"")
    corrupt = good.replace(good[good.index(b"IDAT") + 4:good.index(b"IEND") - 8], b"\x00" * 10)
    with pytest.raises(assets.AssetError):
        assets.decode_png(corrupt)


@pytest.mark.skipif(not have_ref, reason="reference assets are only mounted in the authoring container")
def test_png_every_reference_texture_decodes_like_the_references_stb_image():
    pins = json.load(open(os.path.join(GOLDEN, "reference_png_sha256.json")))
    assert len(pins) >= 40
    for rel, pin in pins.items():
        img = assets.load_png(os.path.join(REF, "resources", rel))
        assert img.shape == (pin["h"], pin["w"], 4), rel
        assert hashlib.sha256(img.tobytes()).hexdigest() == pin["sha256_rgba8"], rel
    d = json.load(open(os.path.join(GOLDEN, "default_texels.json")))["maps"]
    for name, m in d.items():
        if m:
            img = assets.load_png(os.path.join(REF, "resources", "pbr", "default", name + ".png"))
            assert (img == np.array(m["texel"], np.uint8)).all()


# ------------------------------------------------------------------------------------------------ FBX
def _fbx_prop(v):
    if isinstance(v, bytes):
        return b"S" + struct.pack("<I", len(v)) + v
    if isinstance(v, int):
        return b"I" + struct.pack("<i", v)
    if isinstance(v, tuple):  # (type char, numpy array, compress?)
        t, a, comp = v
        raw = np.ascontiguousarray(a).tobytes()
        payload = zlib.compress(raw) if comp else raw
        return t + struct.pack("<III", len(a), 1 if comp else 0, len(payload)) + payload
    raise TypeError(v)


_FBX_WIDE = [False]   # FBX >= 7500: 64-bit EndOffset / NumProperties / PropertyListLen, 25-byte NULL record


def _fbx_node(name, props=(), children=(), base=0):
    """returns bytes of the record placed at absolute offset `base`"""
    pb = b"".join(_fbx_prop(p) for p in props)
    hdr = 25 if _FBX_WIDE[0] else 13
    head_len = hdr + len(name)
    body = b""
    for c in children:
        body += c(base + head_len + len(pb) + len(body))
    if children:
        body += b"\x00" * hdr
    end = base + head_len + len(pb) + len(body)
    return struct.pack("<QQQB" if _FBX_WIDE[0] else "<IIIB", end, len(props), len(pb), len(name)) + name + pb + body


def N(name, props=(), children=()):
    return lambda base: _fbx_node(name, props, children, base)


def write_fbx(path, ctrl, pvi, normals, tangents, uv, uv_index, compress=True, uv_indexed=True, version=7400):
    _FBX_WIDE[0] = version >= 7500
    def layer(name, data_name, data, index_name=None, index=None):
        kids = [N(b"MappingInformationType", [b"ByPolygonVertex"]),
                N(b"ReferenceInformationType", [b"IndexToDirect" if index is not None else b"Direct"]),
                N(data_name, [(b"d", np.asarray(data, "<f8").reshape(-1), compress)])]
        if index is not None:
            kids.append(N(index_name, [(b"i", np.asarray(index, "<i4"), compress)]))
        return N(name, [0], kids)
    geom = N(b"Geometry", [b"Geometry::ball", b"Mesh"], [
        N(b"Vertices", [(b"d", np.asarray(ctrl, "<f8").reshape(-1), compress)]),
        N(b"PolygonVertexIndex", [(b"i", np.asarray(pvi, "<i4"), compress)]),
        layer(b"LayerElementNormal", b"Normals", normals),
        layer(b"LayerElementTangent", b"Tangents", tangents),
        layer(b"LayerElementUV", b"UV", uv, b"UVIndex", uv_index if uv_indexed else None)])
    top = [N(b"FBXHeaderExtension", [], [N(b"Creator", [b"tests/test_assets.py"])]), N(b"Objects", [], [geom])]
    out = b"Kaydara FBX Binary  \x00\x1a\x00" + struct.pack("<I", version)
    for t in top:
        out += t(len(out))
    out += b"\x00" * (25 if _FBX_WIDE[0] else 13)
    _FBX_WIDE[0] = False
    open(path, "wb").write(out)


@pytest.mark.parametrize("compress,uv_indexed,version", [(True, True, 7400), (False, True, 7400), (True, False, 7400),
                                                         (True, True, 7500), (False, False, 7700)])
def test_fbx_reader_on_a_written_file(tmp_path, compress, uv_indexed, version):
    from tools.fbx_geometry import load_vertices
    rng = np.random.default_rng(12)
    n_ctrl, n_tri = 40, 25
    ctrl = rng.standard_normal((n_ctrl, 3)) * 50
    idx = rng.integers(0, n_ctrl, n_tri * 3)
    pvi = idx.copy()
    pvi[2::3] = ~pvi[2::3]                                   # the last corner of a polygon is stored as -(i+1)
    normals, tangents = rng.standard_normal((n_tri * 3, 3)), rng.standard_normal((n_tri * 3, 3))
    uv_table = rng.random((17, 2))
    uv_index = rng.integers(0, 17, n_tri * 3)
    uv = uv_table if uv_indexed else uv_table[uv_index]
    path = str(tmp_path / "ball.fbx")
    write_fbx(path, ctrl, pvi, normals, tangents, uv, uv_index, compress, uv_indexed, version)
    got = assets.load_fbx_vertices(path)
    want = np.concatenate([ctrl[idx], uv_table[uv_index], normals, tangents], axis=1).astype(np.float32)
    assert got.shape == (n_tri * 3, 11) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if version < 7500:                                       # (the Python reader only knows the 32-bit records)
        py, _ = load_vertices(path)                          # the reader that minted the committed fixture agrees
        assert np.array_equal(py.view(np.uint32), got.view(np.uint32))


def test_fbx_polygons_are_fanned_from_their_first_corner(tmp_path):
    """aiProcess_Triangulate (src/scene.cpp:61) on quads and larger convex polygons: triangles (0, i, i+1); per-corner
    layers (ByPolygonVertex) follow the corners"""
    rng = np.random.default_rng(5)
    ctrl = rng.standard_normal((9, 3))
    polys = [[0, 1, 2], [2, 3, 4, 5], [5, 6, 7, 8, 0], [1, 3, 5]]
    pvi = np.concatenate([np.array(q[:-1] + [~q[-1]]) for q in polys])
    n_pv = len(pvi)
    normals, tangents = rng.standard_normal((n_pv, 3)), rng.standard_normal((n_pv, 3))
    uv_table, uv_index = rng.random((6, 2)), rng.integers(0, 6, n_pv)
    path = str(tmp_path / "poly.fbx")
    write_fbx(path, ctrl, pvi, normals, tangents, uv_table, uv_index)
    got = assets.load_fbx_vertices(path)
    corners, first = [], 0
    for q in polys:
        for i in range(1, len(q) - 1):
            corners += [first, first + i, first + i + 1]
        first += len(q)
    flat = np.concatenate([np.array(q) for q in polys])
    want = np.concatenate([ctrl[flat[corners]], uv_table[uv_index[corners]], normals[corners], tangents[corners]], axis=1).astype(np.float32)
    assert got.shape == (3 * (1 + 2 + 3 + 1), 11) and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_fbx_reader_errors(tmp_path):
    p = tmp_path / "x.fbx"
    p.write_bytes(b"Kaydara FBX Binary  \x00\x1a\x00" + struct.pack("<I", 7500) + b"\x00" * 64)
    with pytest.raises(assets.AssetError):                   # 64-bit records are read; this one has no Geometry
        assets.load_fbx_vertices(str(p))
    p.write_bytes(b"solid ascii stl")
    with pytest.raises(assets.AssetError):
        assets.load_fbx_vertices(str(p))
    rng = np.random.default_rng(1)
    open_poly = np.array([0, 1, 2, 3])                       # the last polygon never ends
    write_fbx(str(p), rng.random((4, 3)), open_poly, rng.random((4, 3)), rng.random((4, 3)), rng.random((4, 2)), np.arange(4))
    with pytest.raises(assets.AssetError, match="closed"):
        assets.load_fbx_vertices(str(p))
    two = np.array([0, ~1, 2, 3, ~0])                        # a two-vertex "polygon"
    write_fbx(str(p), rng.random((4, 3)), two, rng.random((5, 3)), rng.random((5, 3)), rng.random((5, 2)), np.arange(5))
    with pytest.raises(assets.AssetError, match="fewer than three"):
        assets.load_fbx_vertices(str(p))
    good = tmp_path / "g.fbx"
    write_fbx(str(good), rng.random((3, 3)), np.array([0, 1, ~2]), rng.random((3, 3)), rng.random((3, 3)), rng.random((3, 2)), np.arange(3))
    data = good.read_bytes()
    (tmp_path / "t.fbx").write_bytes(data[:len(data) // 2])
    with pytest.raises(assets.AssetError):
        assets.load_fbx_vertices(str(tmp_path / "t.fbx"))


def test_packaged_shaderball_mesh_is_the_pinned_conversion():
    """the benchmark mesh ships as package data (bibim_renderer_amd/data), not as a file of the test tree; what pins it is
    the hash minted with it (tests/golden/shaderball_vertices.json, tools/make_fixtures.py)"""
    from bibim_renderer_amd import scene as S
    v = S.load_shaderball_vertices()
    info = json.load(open(os.path.join(GOLDEN, "shaderball_vertices.json")))
    assert v.shape == (29328, 11) and v.dtype == np.float32
    assert hashlib.sha256(v.tobytes()).hexdigest() == info["sha256_f32le"]


@pytest.mark.skipif(not have_ref, reason="reference assets are only mounted in the authoring container")
def test_fbx_reader_on_shaderball_matches_the_committed_fixture():
    v = assets.load_fbx_vertices(os.path.join(REF, "resources", "ShaderBall.fbx"))
    want = np.load(os.path.join(ROOT, "bibim_renderer_amd", "data", "shaderball_vertices.npz"))["vertices"]   # package data, hash pinned below
    info = json.load(open(os.path.join(GOLDEN, "shaderball_vertices.json")))
    assert v.shape == (29328, 11) and np.array_equal(v.view(np.uint32), want.view(np.uint32))
    assert hashlib.sha256(v.tobytes()).hexdigest() == info["sha256_f32le"]


# ------------------------------------------------------------------------------------------------ OBJ
OBJ = """# test
mtllib m.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 0.5 1
vn 0 0 1
vn 0 1 0
usemtl red
f 1//1 2//1 3//1 4//1
usemtl green
f -1//2 1//2 2//2
f 5/7/1 3/9/2 4/1/1 1/1/1 2/2/2
"""
MTL = "newmtl red\nKd 1 0 0\nnewmtl green\nKd 0 0.5 0.25\n"


def test_obj_reader(tmp_path):
    from tools.obj_loader import load_gizmo
    (tmp_path / "g.obj").write_text(OBJ)
    (tmp_path / "m.mtl").write_text(MTL)
    v, idx = assets.load_obj_gizmo(str(tmp_path / "g.obj"))
    assert v.shape == (12, 9) and idx.tolist() == [0, 1, 2, 0, 2, 3, 4, 5, 6, 7, 8, 9, 7, 9, 10, 7, 10, 11]
    assert v[0].tolist() == [0, 0, 0, 1, 0, 0, 0, 0, 1] and v[4].tolist() == [0.5, 0.5, 1, 0, 0.5, 0.25, 0, 1, 0]
    pv, pidx, _ = load_gizmo(str(tmp_path / "g.obj"))
    assert np.array_equal(np.asarray(pv, np.float32), v) and np.array_equal(np.asarray(pidx, np.uint32), idx)
    with pytest.raises(assets.AssetError):
        assets.load_obj_gizmo(str(tmp_path / "missing.obj"))


@pytest.mark.skipif(not have_ref, reason="reference assets are only mounted in the authoring container")
def test_obj_reader_on_the_gizmo_matches_the_committed_fixture():
    v, idx = assets.load_obj_gizmo(os.path.join(REF, "resources", "gizmo.obj"))
    g = np.load(os.path.join(GOLDEN, "gizmo.npz"))
    assert np.array_equal(v.view(np.uint32), np.asarray(g["vertices"], np.float32).view(np.uint32))
    assert np.array_equal(idx, g["indices"]) and idx.size == 594 * 3


def test_shaderball_scene_from_an_fbx_file(tmp_path):
    """bb::ShaderBallScene imports its mesh itself (src/scene.cpp:57-86): same instance data and lights as when the
    vertices are handed over (host-side only: no context, nothing is uploaded)"""
    from bibim_renderer_amd import scene as S
    rng = np.random.default_rng(2)
    tri = 8
    write_fbx(str(tmp_path / "b.fbx"), rng.random((9, 3)), np.where(np.arange(tri * 3) % 3 == 2, ~rng.integers(0, 9, tri * 3), rng.integers(0, 9, tri * 3)),
              rng.random((tri * 3, 3)), rng.random((tri * 3, 3)), rng.random((5, 2)), rng.integers(0, 5, tri * 3))
    a = S.ShaderBallScene(None, fbx_path=tmp_path / "b.fbx", grid=2)
    b = S.ShaderBallScene(None, assets.load_fbx_vertices(tmp_path / "b.fbx"), grid=2)
    assert np.array_equal(a.instances(0), b.instances(0)) and len(a.instances(0)) == 4
    assert a.lights().tobytes() == b.lights().tobytes()
    a.close(); b.close()
    with pytest.raises(Exception):
        S.ShaderBallScene(None, fbx_path=tmp_path / "missing.fbx")
