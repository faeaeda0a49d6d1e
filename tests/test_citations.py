"""Every `file:line` citation of the reference in the oracle, the public headers, the kernels and the host shim points at lines
that exist (authoring container only: /root/reference does not travel).  The judge checks parity through these citations; a
citation that has drifted past the end of its file, or names a file the reference does not have, is caught here.  For the two
shader files the whole path rests on, a few anchor identifiers are also checked at the cited lines (identifiers only: no source
text is kept in the repository)."""
import os
import re

import pytest

from conftest import REFERENCE, ROOT

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src")), reason="/root/reference is not present")

CITING = ["oracle/bb_oracle.c", "oracle/bb_oracle.h", "oracle/scenes.py", "include/bibim_hip.h", "include/bibim_scene.h",
          "include/bibim_assets.h", "bibim_renderer_amd/csrc/bb_kernels.hip.h", "bibim_renderer_amd/csrc/bb_types.h",
          "bibim_renderer_amd/csrc/bb_scene.cpp", "bibim_renderer_amd/csrc/bb_assets.cpp", "bibim_renderer_amd/csrc/bibim_hip.hip",
          "INTEGRATION.md", "DESIGN.md"]
CITE = re.compile(r"((?:src/)?(?:[\w]+/)*[\w]+\.(?:frag|vert|geom|glsl|cpp|h|inl|bff|toml)):(\d+)(?:-(\d+))?(?:,\s*(\d+)(?:-(\d+))?)*")


def _reference_files():
    out = {}
    for base, _dirs, files in os.walk(REFERENCE):
        if os.sep + "external" in base:
            continue
        for f in files:
            out.setdefault(f, []).append(os.path.join(base, f))
    return out


def test_cited_lines_exist():
    files = _reference_files()
    own = {os.path.basename(p) for p in CITING} | {"bb_oracle.c", "bench.py"}
    checked = 0
    for rel in CITING:
        text = open(os.path.join(ROOT, rel)).read()
        for m in CITE.finditer(text):
            name = m.group(1)
            base = os.path.basename(name)
            if base in own or base not in files:     # the repository's own files (bench.py:343) or third-party headers not cited by line
                assert base in own or not name.startswith("src/"), f"{rel} cites {name}, which the reference does not have"
                continue
            cands = [p for p in files[base] if p.endswith(name)] or files[base]
            n_lines = max(sum(1 for _ in open(p, errors="replace")) for p in cands)
            last = max(int(g) for g in re.findall(r"\d+", m.group(0).split(":", 1)[1]))
            assert last <= n_lines, f"{rel} cites {m.group(0)} but {base} has {n_lines} lines"
            checked += 1
    assert checked > 150, checked      # (the oracle and the headers cite the reference line by line)


def _lines(path, a, b):
    return "".join(open(path, errors="replace").read().splitlines(True)[a - 1:b])


def test_anchor_identifiers_stand_where_the_oracle_says():
    sh = os.path.join(REFERENCE, "src", "shaders")
    brdf, frag, vert = (os.path.join(sh, f) for f in ("brdf.glsl", "forward_brdf.frag", "forward_brdf.vert"))
    for path, a, b, tokens in ((brdf, 2, 36, ("distributionGGX", "geometrySchlickGGX", "geometrySmith", "fresnelSchlick", "pow(")),
                               (brdf, 4, 15, ("NdotH2", "a2 - 1", "PI * denom * denom")),
                               (frag, 15, 76, ("uNumLights", "innerCutOff", "0.001", "vec3(0.03)", "outColor")),
                               (frag, 29, 70, ("light.type == 0", "light.type == 2", "normalize(L + V)", "kD *= (1 - metallic)")),
                               (vert, 24, 37, ("aModel", "uProjMat", "uViewMat", "aInvModel", "cross("))):
        text = _lines(path, a, b)
        for t in tokens:
            assert t in text, f"{os.path.basename(path)}:{a}-{b} no longer holds `{t}`"
