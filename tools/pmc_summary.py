#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py per kernel.

  tools/pmc_summary.py <out.json> <bench.json> <counter_collection.csv> [<counter_collection.csv> ...]

Every pass is a separate run of the same bench command (MI355X_MICROARCH.md: FETCH_SIZE
takes 3 of the 4 TCC slots).  Counter values are summed over all dispatches of a kernel
and divided by the probes the bench line of that run reports, so that passes with a
different number of launches stay comparable.  HBM bytes: TCC_EA0_RDREQ counts 64-byte
fabric read requests (32-byte ones are counted separately); FETCH_SIZE is reported in
units of 1024 B and equals RDREQ x 64 B for this kernel's 64-byte random bucket reads
(the gfx950 halving only concerns wide 128-byte streaming requests)."""
import csv
import json
import os
import sys


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_tree_hash  # noqa: E402

out_path, bench_paths, csv_paths = sys.argv[1], [], []
for a in sys.argv[2:]:
    (csv_paths if a.endswith(".csv") else bench_paths).append(a)
assert len(bench_paths) == len(csv_paths), "one bench JSON line per counter pass"
summary = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "csrc_tree_sha256": csrc_tree_hash(), "passes": []}
per_kernel = {}
for bp, cp in zip(bench_paths, csv_paths):
    line = [l for l in open(bp).read().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    probes = b["roofline"]["probes_per_launch"] * b["roofline"]["launches"]  # timed region only
    sums, launches = {}, {}
    with open(cp) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            c = row["Counter_Name"]
            sums.setdefault(k, {}).setdefault(c, 0.0)
            sums[k][c] += float(row["Counter_Value"])
            launches.setdefault(k, set()).add(row["Dispatch_Id"])
    summary["passes"].append({"bench": bp, "counters": cp, "timed_probes": probes, "fill_probes": (b.get("aux") or {}).get("fill_probes")})
    for k, cs in sums.items():
        if not k.startswith(("k_query", "k_fill", "void k_query", "void k_fill")):
            continue
        e = per_kernel.setdefault(k, {})
        for c, v in cs.items():
            e[c] = v
        e["dispatches_in_pass"] = len(launches[k])
summary["kernels"] = per_kernel
# the whole run's query probes (warm-up + timed) are not in the bench line; relate the
# counters to the timed probes through the dispatch counts instead
for k, e in per_kernel.items():
    if "TCC_EA0_RDREQ_sum" in e:
        e["hbm_read_bytes_from_RDREQ"] = e["TCC_EA0_RDREQ_sum"] * 64.0 - e.get("TCC_EA0_RDREQ_32B_sum", 0.0) * 32.0
    if "FETCH_SIZE" in e:
        e["hbm_read_bytes_from_FETCH_SIZE"] = e["FETCH_SIZE"] * 1024.0
# all query variants together, per probe of the bench line (valid when the run has no
# untimed warm-up steps: --warmup 0)
tot = sum(e.get("hbm_read_bytes_from_RDREQ", 0.0) for k, e in per_kernel.items() if "k_query" in k)
tot_f = sum(e.get("hbm_read_bytes_from_FETCH_SIZE", 0.0) for k, e in per_kernel.items() if "k_query" in k)
summary["k_query_all_variants"] = {
    "hbm_read_bytes_from_RDREQ": tot, "hbm_read_bytes_from_FETCH_SIZE": tot_f,
    "hbm_bytes_per_probe": tot / summary["passes"][0]["timed_probes"] if tot else None,
    "hbm_bytes_per_probe_FETCH_SIZE": tot_f / summary["passes"][min(1, len(summary["passes"]) - 1)]["timed_probes"] if tot_f else None,
}
# the fill (round 6, bench.py roofline_fill): read requests + write requests of k_fill per probe of the pass's fill
fill = {}
for k, e in per_kernel.items():
    if "k_fill" in k:
        for c, v in e.items():
            if isinstance(v, float):
                fill[c] = fill.get(c, 0.0) + v
fp = next((p["fill_probes"] for p in summary["passes"] if p.get("fill_probes")), None)
if fill and fp:
    rd = fill.get("TCC_EA0_RDREQ_sum", 0.0) * 64.0 - fill.get("TCC_EA0_RDREQ_32B_sum", 0.0) * 32.0
    wr = None
    if "TCC_EA0_WRREQ_sum" in fill:
        w64 = fill.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        wr = w64 * 64.0 + (fill["TCC_EA0_WRREQ_sum"] - w64) * 32.0
    summary["k_fill"] = {"fill_probes": fp, "hbm_read_bytes_per_probe": rd / fp if rd else None, "hbm_write_bytes_per_probe": wr / fp if wr is not None else None,
                         "hbm_bytes_per_probe": ((rd + (wr or 0.0)) / fp) if rd else None, "atomics_at_memory_per_probe": fill.get("TCC_EA0_ATOMIC_sum", 0.0) / fp if "TCC_EA0_ATOMIC_sum" in fill else None,
                         "note": "TCC_EA0_RDREQ x 64 B (+ TCC_EA0_WRREQ: 64-B and 32-B write requests) summed over the k_fill dispatches of the pass, per probe of that pass's fill (aux.fill_probes)"}
json.dump(summary, open(out_path, "w"), indent=1)
print(json.dumps(summary["kernels"], indent=1)[:3000])
