"""Identity of the kernels a measurement was taken on: sha256 over the HIP/C++ sources and the build recipe under csrc/.
The counter summaries committed under profiles/ carry it (tools/profile_summary.py) and bench.py compares it with the
tree it runs from: a summary of another build is not quoted (roofline.traffic / roofline.valu become null)."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_source_sha256() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()
