"""Minimal binary-FBX (7.x, 32-bit record offsets) geometry reader.

Reads the first `Geometry` object of a binary FBX file and expands it to the non-indexed
`Vertex[]` triangle list that ShaderBallScene builds from assimp's output
(reference: src/scene.cpp:57-86 -- `aiProcess_Triangulate | aiProcess_CalcTangentSpace`,
mMeshes[0], three vertices per face, Pos/UV0/Normal/Tangent).

assimp itself is a binary-only dependency of the reference (Windows import libs, DLLs absent), so
its output is "parity unpinned"; the rule stated in SURVEY.md section 8(c) is implemented here:
  vertex k = (Vertices[PolygonVertexIndex[k] ^ (-1 if it closes a polygon)], UV0[UVIndex0[k]],
              Normals[k], Tangents[k]), doubles cast to float32, file tangents kept, no UV flip,
              no unit scaling.
Pure data conversion: used by tools/make_fixtures.py to mint tests/golden/shaderball_vertices.npz.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_ARRAY = {b"f": ("<f4", 4), b"d": ("<f8", 8), b"l": ("<i8", 8), b"i": ("<i4", 4), b"b": ("u1", 1)}
_SCALAR = {b"Y": "<h", b"C": "<?", b"I": "<i", b"F": "<f", b"D": "<d", b"L": "<q"}


class Node:
    __slots__ = ("name", "props", "children")

    def __init__(self, name, props, children):
        self.name, self.props, self.children = name, props, children

    def find(self, name):
        for c in self.children:
            if c.name == name:
                return c
        return None

    def find_all(self, name):
        return [c for c in self.children if c.name == name]


def _read_props(buf, pos, count):
    props = []
    for _ in range(count):
        t = buf[pos:pos + 1]
        pos += 1
        if t in _SCALAR:
            fmt = _SCALAR[t]
            (v,) = struct.unpack_from(fmt, buf, pos)
            pos += struct.calcsize(fmt)
            props.append(v)
        elif t in _ARRAY:
            n, enc, clen = struct.unpack_from("<III", buf, pos)
            pos += 12
            raw = buf[pos:pos + clen]
            pos += clen
            if enc == 1:
                raw = zlib.decompress(raw)
            dt, sz = _ARRAY[t]
            props.append(np.frombuffer(raw, dtype=dt, count=n))
        elif t in (b"S", b"R"):
            (n,) = struct.unpack_from("<I", buf, pos)
            pos += 4
            props.append(bytes(buf[pos:pos + n]))
            pos += n
        else:
            raise ValueError(f"unknown FBX property type {t!r} at {pos - 1}")
    return props, pos


def _read_node(buf, pos):
    end, nprops, plen, nlen = struct.unpack_from("<IIIB", buf, pos)
    if end == 0:
        return None, pos + 13
    pos += 13
    name = bytes(buf[pos:pos + nlen]).decode("ascii", "replace")
    pos += nlen
    props, p2 = _read_props(buf, pos, nprops)
    assert p2 == pos + plen, "FBX property list length mismatch"
    pos = p2
    children = []
    while pos < end:
        child, pos = _read_node(buf, pos)
        if child is None:
            break
        children.append(child)
    return Node(name, props, children), end


def parse(path):
    buf = memoryview(open(path, "rb").read())
    if bytes(buf[:20]) != b"Kaydara FBX Binary  ":
        raise ValueError("not a binary FBX file")
    (version,) = struct.unpack_from("<I", buf, 23)
    if version >= 7500:
        raise ValueError("64-bit FBX records (>= 7500) not supported")
    pos = 27
    top = []
    while pos < len(buf):
        node, pos = _read_node(buf, pos)
        if node is None:
            break
        top.append(node)
    return Node("", [version], top)


def _layer(geom, name, data_name, index_name=None):
    el = geom.find(name)
    if el is None:
        return None
    mapping = el.find("MappingInformationType").props[0]
    ref = el.find("ReferenceInformationType").props[0]
    data = np.asarray(el.find(data_name).props[0], dtype=np.float64)
    idx = None
    if ref == b"IndexToDirect" and index_name is not None:
        idx = np.asarray(el.find(index_name).props[0], dtype=np.int64)
    return mapping, ref, data, idx


def load_vertices(path):
    """Returns (vertices[N, 11] float32 in bb::Vertex order, info dict)."""
    root = parse(path)
    geom = root.find("Objects").find("Geometry")
    ctrl = np.asarray(geom.find("Vertices").props[0], dtype=np.float64).reshape(-1, 3)
    pvi = np.asarray(geom.find("PolygonVertexIndex").props[0], dtype=np.int64)
    ends = pvi < 0
    idx = np.where(ends, ~pvi, pvi)
    # polygon sizes: every polygon must already be a triangle (true for ShaderBall.fbx)
    sizes = np.diff(np.concatenate([[-1], np.nonzero(ends)[0]]))
    if not np.all(sizes == 3):
        raise ValueError("non-triangle polygons present; triangulation not implemented")
    n = len(pvi)

    def per_polygon_vertex(layer, width):
        mapping, ref, data, lidx = layer
        data = data.reshape(-1, width)
        if mapping != b"ByPolygonVertex":
            raise ValueError(f"unsupported mapping {mapping!r}")
        return data[lidx] if lidx is not None else data

    normals = per_polygon_vertex(_layer(geom, "LayerElementNormal", "Normals", "NormalsIndex"), 3)
    tangents = per_polygon_vertex(_layer(geom, "LayerElementTangent", "Tangents", "TangentsIndex"), 3)
    uv = per_polygon_vertex(_layer(geom, "LayerElementUV", "UV", "UVIndex"), 2)
    assert len(normals) == n and len(tangents) == n and len(uv) == n
    out = np.empty((n, 11), dtype=np.float32)
    out[:, 0:3] = ctrl[idx].astype(np.float32)
    out[:, 3:5] = uv.astype(np.float32)
    out[:, 5:8] = normals.astype(np.float32)
    out[:, 8:11] = tangents.astype(np.float32)
    info = {
        "fbx_version": int(root.props[0]),
        "control_points": int(len(ctrl)),
        "polygon_vertices": int(n),
        "triangles": int(n // 3),
        "bbox_min": ctrl.min(0).tolist(),
        "bbox_max": ctrl.max(0).tolist(),
    }
    return out, info


if __name__ == "__main__":
    import sys

    v, info = load_vertices(sys.argv[1])
    print(info, v.shape, v[:2])
