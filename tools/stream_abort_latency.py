#!/usr/bin/env python3
"""Developer tool: how fast does a streaming window stop after grp_classify_stream_abort?
Fills a C1-like filter with 100k reads, starts a 4096-read window, aborts it after
--delay-us and reports when the launch ended and how many reads had been decided."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldrush_amd import host, native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=100000)
ap.add_argument("--window", type=int, default=4096)
ap.add_argument("--delay-us", type=float, nargs="*", default=[0, 200, 1000, 3000, 1e9])
ap.add_argument("--poll", action="store_true", help="call grp_classify_stream_poll (hipEventQuery) while waiting")
ap.add_argument("--trials", type=int, default=0, help="random delays, both slots in flight (cur + next), report the slowest aborts")
a = ap.parse_args()
G, k, w, tile, h = 100_000_000, 22, 16, 1000, 3
hl = host.load()
seeds = host.make_seed_pattern("1011011110110111101101", k, w, h)
m = hl.gr_calc_optimal_size(hl.gr_hash_universe(w, G, h), 1, 0.1)
dr = native.synth_reads(a.reads, G)
eng = native.Engine(k, h, tile, m, seeds)
rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
eng.bv_insert(rb)
eng.finalize()
for i in range(0, 2000, 7):
    eng.insert_read(rb, i, 0, int(dr.lens[i]) // tile, 10, 1 + i, 0)
eng.sync()
for d_us in a.delay_us:
    for slot in (0, 1):
        t0 = time.perf_counter()
        v = eng.stream_begin(rb, 20000, a.window, slot)
        t1 = time.perf_counter()
        while (time.perf_counter() - t1) * 1e6 < d_us and not eng.stream_poll(slot):
            pass
        before = int((v["pad"] == 1).sum())
        ta = time.perf_counter()
        eng.stream_abort(slot)
        while not eng.stream_poll(slot):
            pass
        te = time.perf_counter()
        n = eng.stream_end(slot)
        print(f"delay {d_us:>10.0f} us slot {slot}: begin {1e6*(t1-t0):7.1f} us, decided at abort {before:5d}, "
              f"launch ended {1e6*(te-ta):8.1f} us after the abort, decided in the end {n:5d}")

if a.trials:
    rng = np.random.default_rng(1)
    worst = []
    for t in range(a.trials):
        d_us = float(rng.uniform(0, 8000))
        s0 = t & 1
        t1 = time.perf_counter()
        v0 = eng.stream_begin(rb, 20000, a.window, s0)
        v1 = eng.stream_begin(rb, 20000 + a.window, a.window, s0 ^ 1)
        first_seen = -1.0
        while (time.perf_counter() - t1) * 1e6 < d_us:
            if first_seen < 0 and v0["pad"][0] == 1:
                first_seen = (time.perf_counter() - t1) * 1e6
            if a.poll:
                eng.stream_poll(s0)
        ta = time.perf_counter()
        eng.stream_abort(s0)
        eng.stream_abort(s0 ^ 1)
        b0, b1 = int((v0["pad"] == 1).sum()), int((v1["pad"] == 1).sum())
        while not eng.stream_poll(s0) or not eng.stream_poll(s0 ^ 1):
            pass
        te = time.perf_counter()
        n0, n1 = eng.stream_end(s0), eng.stream_end(s0 ^ 1)
        worst.append((1e6 * (te - ta), d_us, b0, n0, b1, n1, first_seen))
    worst.sort(reverse=True)
    for w_ in worst[:8]:
        print("abort->end %8.1f us  delay %7.0f us  cur decided %5d -> %5d   next decided %5d -> %5d  first record seen at %7.0f us" % w_)
    print("median abort->end %.1f us" % sorted(x[0] for x in worst)[len(worst) // 2])
    late = [x for x in worst if x[1] > 500 and x[6] < 0]
    print("trials with delay > 500 us in which the first record was never seen before the abort: %d of %d" % (len(late), sum(1 for x in worst if x[1] > 500)))

# steady-state rate of one long streaming launch vs the synchronous window
for nwin in (8192,):
    for rep in range(3):
        t0 = time.perf_counter()
        v = eng.stream_begin(rb, 30000, nwin, 0)
        while not eng.stream_poll(0):
            pass
        t1 = time.perf_counter()
        eng.stream_end(0)
        t2 = time.perf_counter()
        eng.classify_reads(rb, 30000, nwin)
        t3 = time.perf_counter()
        probes = float(sum(int(x) // tile * tile for x in dr.lens[30000:30000 + nwin])) * h
        print(f"{nwin} reads: streaming launch {1e3*(t1-t0):.2f} ms ({probes/(t1-t0)/1e9:.1f} G probes/s), synchronous window {1e3*(t3-t2):.2f} ms ({probes/(t3-t2)/1e9:.1f} G probes/s)")
