#!/usr/bin/env python3
"""Developer: where the seconds of the slow small GPU tests go (engine creation, fill, finalize, classifier run)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import orc as oracle
from goldrush_amd import host, native, synth
from helpers import default_seeds

def lap(msg, t0):
    t1 = time.time(); print("  %-28s %.2f s" % (msg, t1 - t0), flush=True); return t1

for mode in sys.argv[1:] or ["force", "auto"]:
    os.environ["GRP_BATCH"] = mode
    print("GRP_BATCH", mode)
    tile, k, h, block = 500, 22, 3, 1
    seeds = default_seeds(h)
    g = synth.random_genome(150_000, 21)
    reads = [r[1] for r in synth.make_reads(g, 85, mean_len=5000, min_len=3500, seed=22, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    t0 = time.time()
    eng = native.Engine(k, h, tile, m, seeds); t0 = lap("engine", t0)
    b = eng.upload(reads); t0 = lap("upload", t0)
    eng.bv_insert(b); eng.sync(); t0 = lap("fill", t0)
    eng.finalize(); eng.sync(); t0 = lap("finalize", t0)
    eng.set_timing(True)
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h)
    cls.run(b._h, b.lens); eng.sync(); t0 = lap("classifier run", t0)
    st = cls.state()
    print("  ", {k_: st[k_] for k_ in ("windows", "batches", "batches_undone", "inserts", "seconds_windows", "seconds_commit", "stream_inserts")})
    print("  ", {k_: (v["launches"], round(v["ms"], 1)) for k_, v in eng.kernel_stats().items() if v["launches"]})
    eng.close(); t0 = lap("close", t0)
