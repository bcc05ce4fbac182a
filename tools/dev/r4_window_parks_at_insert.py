#!/usr/bin/env python3
"""Developer: repeat the streaming part of tests/test_gpu_parity.py::test_tiles_where_every_probe_returns_another_id[5]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import orc as oracle
from goldrush_amd import native
from helpers import default_seeds, random_reads

h = 5
seeds = default_seeds(h)
m = oracle.load().orc_calc_optimal_size(300_000, 1, 0.1)
eng = native.Engine(22, h, 1000, m, seeds)
reads = random_reads(5, 2500, 5200, seed=43)
b = eng.upload(reads)
eng.bv_insert(b)
pop = eng.finalize()
rng = np.random.default_rng(8)
ids = rng.integers(1, 1 << 30, size=pop, dtype=np.uint32)
eng.import_ids(0, ids=ids, counts=np.zeros(pop, dtype=np.uint32))
dec = eng.classify_reads(b)
print("sync decisions", [(int(d["kind"]), int(d["num_tiles"])) for d in dec], eng.verify_stats())
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    v = eng.stream_begin(b, 0, len(reads), 0)
    t0 = time.time()
    while not eng.stream_poll(0):
        assert time.time() - t0 < 60
    pads, kinds = v["pad"].copy(), v["kind"].copy()
    rc = eng.stream_end(0)
    if not (np.all(pads == 1) and np.all(kinds != 0)):
        bad += 1
        print("iteration", it, "pads", pads, "kinds", kinds, "stream_end", rc, eng.verify_stats(), flush=True)
print("bad", bad)
