"""Test helper (GPU): ONE resumable streaming window over a stream in which a few reads insert among many that do not,
driven the way tests/test_gpu_stream_insert.py drives it; returns the records the host saw, the engine's final ID / count
arrays' digest and what the in-launch inserts kept (grp_debug_stream_stats).  Run as a script it prints that as JSON —
tests/test_gpu_stream_keep.py starts it with GRP_STREAM_KEEP = 0 / 1 / 2 (the engine reads the switch once per process)."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

K, H, TILE, BLOCK = 22, 3, 500, 4


def make_stream():
    """reads of a covered genome A; clusters of overlapping reads of islands B1..B4 nobody has covered come in between: the
    first read of a cluster inserts, the ones behind it — queried by the launch BEFORE that insert — must be decided against it"""
    from goldrush_amd import synth

    ga = synth.random_genome(160_000, 101)
    cover = [r[1] for r in synth.make_reads(ga, 70, mean_len=5000, min_len=3500, seed=102, max_len=9000)]   # the head: these fill the path
    steady = [r[1] for r in synth.make_reads(ga, 260, mean_len=5000, min_len=3500, seed=103, max_len=9000)]  # covered: (mostly) no insert
    reads = cover + steady
    rng = np.random.default_rng(104)
    for i, at in enumerate((120, 170, 230, 300)):
        gb = synth.random_genome(9_000, 200 + i)
        cluster = [gb[o:o + 6000].tobytes() for o in (0, 1500, 3000, 700)]  # four reads of one island, overlapping
        for j, s in enumerate(cluster):
            reads.insert(at + j + int(rng.integers(0, 2)), s)
    return reads


def run(native, seeds, m, reads, limit=120.0):
    eng = native.Engine(K, H, TILE, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    pop = eng.finalize()
    n = len(reads)
    v = eng.stream_begin(b, 0, n, 0, resumable=True)
    gen, ids_inserted, n_ins = 1, 0, 0
    got = []
    for j in range(n):
        t0 = time.time()
        while int(v["pad"][j]) != gen:
            assert time.time() - t0 < limit, "record %d of generation %d never came" % (j, gen)
            assert not eng.stream_poll(0) or int(v["pad"][j]) == gen, "the launch ended without record %d" % j
        d = v[j].copy()
        kind = int(d["kind"])
        assert kind != 0
        first_id = 0
        if kind in (2, 4):
            ids_inserted += 1
            first_id = ids_inserted
            if kind == 2:
                ts, te, off = 0, int(d["num_tiles"]), 0
                ids_inserted += len(reads[j]) // (TILE * BLOCK)
            else:
                ts, te, off = int(d["trim_start"]), int(d["trim_end"]) + 1, 1
                ids_inserted += (int(d["trim_end"]) - int(d["trim_start"])) // BLOCK
            gen = eng.stream_insert(0, j, ts, te, BLOCK, first_id, off)
            n_ins += 1
        got.append((j, kind, int(d["num_tiles"]), int(d["num_assigned"]), int(d["trim_start"]) if kind == 4 else 0, int(d["trim_end"]) if kind == 4 else 0, first_id, 1,
                    int(d["hits"]), int(d["misses"])))
    t0 = time.time()
    while not eng.stream_poll(0):
        assert time.time() - t0 < 60
    eng.stream_end(0)
    ids, counts = eng.export_ids()
    st = eng.stream_stats()
    eng.close()
    return {"pop": int(pop), "records": got, "inserts": n_ins, "ids": ids, "counts": counts, "stats": st}


def main():
    from goldrush_amd import native
    from helpers import default_seeds
    import orc

    orc.build()
    orc.load()
    native.load()
    seeds = default_seeds(H)
    m = orc.load().orc_calc_optimal_size(2_500_000, 1, 0.1)
    r = run(native, seeds, m, make_stream())
    dig = hashlib.sha256(r["ids"].tobytes() + r["counts"].tobytes()).hexdigest()
    print(json.dumps({"records": r["records"], "inserts": r["inserts"], "arrays": dig, "stats": r["stats"], "pop": r["pop"]}))


if __name__ == "__main__":
    main()
