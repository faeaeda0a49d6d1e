"""Minimal Wavefront OBJ/MTL reader for BASELINE config #1 (gizmo.obj -> bb::GizmoVertex[] + uint32 indices).

The reference loads the gizmo through assimp with aiProcess_Triangulate (src/main.cpp:219-283) and gives every
vertex the Kd colour of its mesh's material; assimp is a binary-only dependency (parity unpinned), so the rule
implemented here is stated explicitly:
  * one output vertex per face corner (position v, normal vn), colour = Kd of the face's `usemtl` material;
  * polygons with more than three corners are fan-triangulated from their first corner (gizmo.obj's polygons
    are planar and convex, so any triangulation covers the same surface);
  * faces keep file order.
"""
from __future__ import annotations

import os

import numpy as np


def load_mtl(path):
    mats, cur = {}, None
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "newmtl":
            cur = t[1]
            mats[cur] = (1.0, 1.0, 1.0)
        elif t[0] == "Kd" and cur:
            mats[cur] = tuple(float(x) for x in t[1:4])
    return mats


def load_gizmo(path):
    pos, nrm, mats = [], [], {}
    verts, idx = [], []
    color = (1.0, 1.0, 1.0)
    n_poly = {}
    for line in open(path):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "mtllib":
            mats = load_mtl(os.path.join(os.path.dirname(path), t[1]))
        elif t[0] == "v":
            pos.append([float(x) for x in t[1:4]])
        elif t[0] == "vn":
            nrm.append([float(x) for x in t[1:4]])
        elif t[0] == "usemtl":
            color = mats[t[1]]
        elif t[0] == "f":
            corners = []
            for c in t[1:]:
                p = c.split("/")
                vi = int(p[0]); ni = int(p[2]) if len(p) > 2 and p[2] else 0
                vi = vi - 1 if vi > 0 else len(pos) + vi
                ni = ni - 1 if ni > 0 else len(nrm) + ni
                corners.append(len(verts))
                verts.append(pos[vi] + list(color) + nrm[ni])
            n_poly[len(corners)] = n_poly.get(len(corners), 0) + 1
            for k in range(1, len(corners) - 1):
                idx += [corners[0], corners[k], corners[k + 1]]
    v = np.asarray(verts, np.float32)
    info = {"positions": len(pos), "normals": len(nrm), "polygons_by_size": {str(k): n for k, n in sorted(n_poly.items())},
            "triangles": len(idx) // 3, "vertices": len(v), "materials": {k: list(c) for k, c in mats.items()}}
    return v, np.asarray(idx, np.uint32), info
