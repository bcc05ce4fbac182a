#!/usr/bin/env python3
"""sha256 over the engine's sources (goldrush_amd/csrc/**).  A PMC summary
(tools/pmc_summary.py) stores it; bench.py applies the summary's HBM bytes per probe
only to a build of exactly these sources (roofline.traffic is null otherwise)."""
import glob
import hashlib
import os


def csrc_tree_hash():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "goldrush_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "**", "*"), recursive=True)):
        if os.path.isfile(f) and f.endswith((".hip", ".inc", ".h", ".hpp", ".cpp")):
            h.update(os.path.relpath(f, root).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_tree_hash())
