"""Developer tool: raw kernel throughput (fill / query) on synthetic reads.
Not the judged benchmark (see bench.py); used to iterate on kernels."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldrush_amd import native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200000)
    ap.add_argument("--genome", type=float, default=100e6)
    ap.add_argument("--h", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--populate", type=int, default=4000, help="reads inserted as IDs before querying")
    ap.add_argument("--qbatches", type=int, default=8)
    a = ap.parse_args()
    G = int(a.genome)
    k, w, tile = 22, 16, 1000
    preset = "1011011110110111101101"
    seeds = [preset[:11] + "0" * i + preset[11:] for i in range(a.h)]
    U = int(np.float32(min(4 ** w, 2 * G)) * np.float32(0.5) * np.float32(a.h))
    n = int(-float(U) / np.log(1.0 - 0.1))
    m = n + (64 - n % 64)
    t = time.time()
    dr = native.synth_reads(a.reads, G)
    print(f"synth {a.reads} reads, {int(dr.lens.sum())/1e9:.2f} Gbases: {time.time()-t:.2f}s", flush=True)
    eng = native.Engine(k, a.h, tile, m, seeds)
    rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    t = time.time()
    eng.bv_insert(rb)
    eng.sync()
    t_fill = time.time() - t
    t = time.time()
    pop = eng.finalize()
    t_fin = time.time() - t
    print(f"m={m} pop={pop} occ={pop/m:.3f} fill {t_fill:.3f}s finalize {t_fin:.3f}s", flush=True)
    t = time.time()
    idn = 0
    for r in range(min(a.populate, a.reads)):
        nt = int(dr.lens[r]) // tile
        idn += 1
        for bs in range(0, nt, 10):
            eng.insert_tiles(rb, r, bs, min(bs + 10, nt), idn + bs // 10)
        idn += nt // 10
    eng.sync()
    print(f"populate {a.populate} reads: {time.time()-t:.3f}s", flush=True)
    eng.reset_kernel_stats()
    t = time.time()
    nq = 0
    hits = 0
    for b in range(a.qbatches):
        first = (a.populate + b * a.batch) % max(a.reads - a.batch, 1)
        tiles, lists, st = eng.query_tiles(rb, first, a.batch)
        nq += a.batch
        hits += st["hits"]
    wall = time.time() - t
    ks = eng.kernel_stats()
    q = ks["query"]
    gbps = q["units"] * 128 / (q["ms"] * 1e-3) / 1e9
    out = {"reads_per_s_wall": nq / wall, "reads_per_s_kernel": nq / (q["ms"] * 1e-3), "query_ms_per_launch": q["ms"] / q["launches"],
           "probes_per_launch": q["units"] / q["launches"], "GBps_128B_per_probe": gbps, "frac_of_8TBps": gbps / 8000,
           "hit_frac": hits / max(q["units"], 1), "fill_s": t_fill, "fill_Gprobes_per_s": ks["fill"]["units"] / max(t_fill, 1e-9) / 1e9 if ks["fill"]["units"] else None}
    ks_all = eng.kernel_stats()
    print(json.dumps(out))
    print(json.dumps(ks_all))


if __name__ == "__main__":
    main()
