#!/usr/bin/env python3
"""Developer tool: hunt for rare hand-over races of the streaming windows — classify a long
synthetic stream with streaming windows several times and compare every commit and the
final ID / count arrays with one synchronous-window run."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldrush_amd import host, native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=300000)
ap.add_argument("--genome", type=float, default=30e6)
ap.add_argument("--repeats", type=int, default=3)
ap.add_argument("--direct", action="store_true", help="hunt in the zero-copy latency path instead: small synchronous windows with and without it")
ap.add_argument("--max-window", type=int, default=0)
a = ap.parse_args()
k, h, tile, block, G = 22, 3, 1000, 10, int(a.genome)
hl = host.load()
seeds = host.make_seed_pattern("1011011110110111101101", k, 16, h)
m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, G, h), 1, 0.1)
dr = native.synth_reads(a.reads, G)
lens = np.ascontiguousarray(dr.lens, dtype=np.uint32)


def run(env):
    for key in ("GRP_STREAM", "GRP_PIPELINE", "GRP_NO_DIRECT"):
        os.environ.pop(key, None)
    os.environ.update(env)
    eng = native.Engine(k, h, tile, m, seeds)
    rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    eng.bv_insert(rb)
    eng.finalize()
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, target_bases=int(0.9 * G), max_paths=1, silver_path=False, max_window=a.max_window)
    for first in range(0, a.reads, 8192):
        cls.run_range(rb._h, lens, first, min(8192, a.reads - first))
    eng.sync()
    st = cls.state()
    ids, counts = eng.export_ids()
    out = ([c[:8] for c in cls.commits], ids.copy(), counts.copy(), st)
    cls.close()
    eng.close()
    return out


if a.direct:
    ref = run({"GRP_STREAM": "off", "GRP_PIPELINE": "off", "GRP_NO_DIRECT": "1"})
    print("general path: inserts %d windows %d" % (ref[3]["inserts"], ref[3]["windows"]))
    for r in range(a.repeats):
        got = run({"GRP_STREAM": "off", "GRP_PIPELINE": "off"})
        same = got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
        print("zero-copy run %d: inserts %d windows %d -> %s" % (r, got[3]["inserts"], got[3]["windows"], "identical" if same else "DIFFERENT"))
        if not same:
            sys.exit(1)
    sys.exit(0)
ref = run({"GRP_STREAM": "off", "GRP_PIPELINE": "off"})
print("synchronous: inserts %d windows %d" % (ref[3]["inserts"], ref[3]["windows"]))
for r in range(a.repeats):
    got = run({"GRP_STREAM": "force"})
    same = got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    print("streaming run %d: inserts %d windows %d queried %d -> %s" % (r, got[3]["inserts"], got[3]["windows"], got[3]["reads_queried"], "identical" if same else "DIFFERENT"))
    if not same:
        bad = next(i for i, (x, y) in enumerate(zip(got[0], ref[0])) if x != y)
        print("first difference at commit", bad, got[0][bad], ref[0][bad])
        sys.exit(1)
