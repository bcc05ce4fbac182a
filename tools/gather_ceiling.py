#!/usr/bin/env python3
"""Measure the random-64-byte-line ceiling of this GPU's memory system with
tools/gather_bench.hip and archive it as JSON (profiles/<tag>_gather_ceiling.json is what
bench.py reads for `roofline.line_rate_ceiling_Gps`).

  python3 tools/gather_ceiling.py <out.json> [table_GiB ...]

mode 10 = what k_query does: the 4 lanes of a quad read one 64-byte bucket with one
coalesced access (one line per item); mode 7 = one 16-byte load per lane and bucket.
The table sizes default to the bucket arrays of C1 (14 GiB) and C2 (64 GiB): the ceiling
depends on the footprint (TLB reach), so the one matching the workload is reported as
`quad_64B_lines_Gps` (the largest table measured).
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "gather_bench")


def build():
    if os.path.exists(BIN) and os.path.getmtime(BIN) >= os.path.getmtime(BIN + ".hip"):
        return
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", BIN, BIN + ".hip"], check=True)


def run(table_mib: int, mode: int, unr: int, items: int = 64):
    r = subprocess.run([BIN, str(table_mib), "64", str(items), str(mode), str(unr)], capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("mode")]
    if r.returncode != 0 or not line:
        raise RuntimeError("gather_bench failed: %s %s" % (r.stdout[-500:], r.stderr[-500:]))
    g = float(re.search(r"([0-9.]+) G items/s", line[-1]).group(1))
    return g, line[-1]


def main():
    out = sys.argv[1]
    sizes = [int(x) for x in sys.argv[2:]] or [14, 64]
    build()
    res = {"tool": "tools/gather_bench.hip", "what": "random gathers over a table of the given size, 8192 workgroups x 256 lanes x 64 items; G items/s of the 3rd repetition",
           "runs": []}
    best_quad = {}
    for gib in sizes:
        for mode, name in ((10, "quad_64B_line"), (7, "lane_16B_load")):
            for unr in (2, 4):
                g, raw = run(gib * 1024, mode, unr)
                res["runs"].append({"table_GiB": gib, "mode": mode, "shape": name, "unroll": unr, "G_items_per_s": g, "raw": raw})
                if mode == 10:
                    best_quad[gib] = max(best_quad.get(gib, 0.0), g)
    res["quad_64B_lines_Gps_by_table_GiB"] = best_quad
    res["quad_64B_lines_Gps"] = best_quad[max(best_quad)]
    res["quad_64B_lines_table_GiB"] = max(best_quad)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
