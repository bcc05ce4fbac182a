"""GPU: a streaming window that applies the inserts the host commits inside its own launch
(grp_classify_stream_begin_resumable / grp_classify_stream_insert, round 3) — the engine
protocol driven from Python against the oracle's serial loop (process_read,
goldrush_path.cpp:892-1094), and the product's classifier on top of it."""
import time

import numpy as np
import pytest

from helpers import default_seeds

pytestmark = pytest.mark.gpu
FIELDS = ["kind", "num_tiles", "num_assigned", "trim_start", "trim_end", "hits", "misses"]


def _wait(view, j, gen, eng, slot, limit=60.0):
    t0 = time.time()
    while int(view["pad"][j]) != gen:
        assert time.time() - t0 < limit, f"record {j} of generation {gen} never came (pad {int(view['pad'][j])})"
        assert not eng.stream_poll(slot) or int(view["pad"][j]) == gen, "the launch ended without the record"


def _serial(orc, m, seeds, tile, k, reads, block):
    from oracle_engine import serial_reference

    return serial_reference(orc, m, seeds, tile, k, reads, block=block)


@pytest.mark.parametrize("h,tile,block", [(3, 500, 4), (5, 300, 3), (1, 700, 10)])
def test_window_applies_inserts_itself(oracle, native, h, tile, block):
    """ONE launch over the whole range: every insert record is answered with stream_insert (the IDs
    the serial loop allocates); records, hits / misses, and the final ID / count arrays equal the
    oracle's serial loop.  The head of a path: most reads insert, i.e. dozens of in-launch inserts."""
    from goldrush_amd import host, synth

    k = 22 if h != 5 else 20
    seeds = default_seeds(h) if h != 5 else host.make_seed_pattern("", k, 14, h)
    g = synth.random_genome(150_000, 31)
    reads = [r[1] for r in synth.make_reads(g, 90, mean_len=5000, min_len=3500, seed=32, max_len=9000)]
    reads.insert(7, reads[3][:tile - 1])  # a read without a single tile: decided by the host, in the new generation too
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    exp, mf_ref = _serial(oracle, m, seeds, tile, k, reads, block)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    assert eng.finalize() == mf_ref.pop
    n = len(reads)
    v = eng.stream_begin(b, 0, n, 0, resumable=True)
    gen, ids_inserted, n_ins = 1, 0, 0
    got = []
    for j in range(n):
        _wait(v, j, gen, eng, 0)
        d = v[j].copy()
        kind = int(d["kind"])
        assert kind != 0
        first_id = 0
        if kind in (2, 4):
            ids_inserted += 1
            first_id = ids_inserted
            if kind == 2:
                ts, te, off = 0, int(d["num_tiles"]), 0
                ids_inserted += len(reads[j]) // (tile * block)
            else:
                ts, te, off = int(d["trim_start"]), int(d["trim_end"]) + 1, 1
                ids_inserted += (int(d["trim_end"]) - int(d["trim_start"])) // block
            gen = eng.stream_insert(0, j, ts, te, block, first_id, off)
            n_ins += 1
            assert gen == 1 + n_ins
        got.append((j, kind, int(d["num_tiles"]), int(d["num_assigned"]), int(d["trim_start"]) if kind == 4 else 0, int(d["trim_end"]) if kind == 4 else 0, first_id, 1))
    t0 = time.time()
    while not eng.stream_poll(0):  # every read decided, nothing parked: the launch ends by itself
        assert time.time() - t0 < 60
    eng.stream_end(0)
    assert got == exp
    assert n_ins >= 20
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, mf_ref.ids()) and np.array_equal(counts, mf_ref.counts())
    # the engine is as usable as ever: classic calls behind the launch
    dec = eng.classify_reads(b)
    assert len(dec) == n
    eng.close()


def test_abort_answers_a_parked_resumable_window(oracle, native):
    """A resumable window waits where it parks; _abort ends it (nothing inserted), _insert on an
    ordinary window is refused, and an abort that overtakes a posted insert is reported by _end."""
    from goldrush_amd import synth

    k, h, tile, block = 22, 3, 500, 4
    seeds = default_seeds(h)
    g = synth.random_genome(100_000, 5)
    reads = [r[1] for r in synth.make_reads(g, 40, mean_len=5000, min_len=3500, seed=6, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    eng.finalize()
    ref = eng.classify_reads(b)
    assert int(ref["kind"][0]) == 2  # empty ID array: the first read inserts
    v = eng.stream_begin(b, 0, len(reads), 0, resumable=True)
    _wait(v, 0, 1, eng, 0)
    time.sleep(0.2)
    assert not eng.stream_poll(0), "a parked resumable window must wait for the host"
    eng.stream_abort(0)
    t0 = time.time()
    while not eng.stream_poll(0):
        assert time.time() - t0 < 30
    eng.stream_end(0)
    ids, counts = eng.export_ids()
    assert not ids.any() and not counts.any()
    # an ordinary window ends where it parks and takes no insert
    v = eng.stream_begin(b, 0, len(reads), 1)
    _wait(v, 0, 1, eng, 1)
    with pytest.raises(native.GrpError):
        eng.stream_insert(1, 0, 0, int(v["num_tiles"][0]), block, 1, 0)
    t0 = time.time()
    while not eng.stream_poll(1):
        assert time.time() - t0 < 30
    eng.stream_end(1)
    assert np.array_equal(eng.classify_reads(b), ref)
    eng.close()


@pytest.mark.parametrize("max_window", [4096, 40])
def test_classifier_golden_mode_streams_through_inserts(oracle, native, max_window, monkeypatch):
    """The product's classifier in golden-path mode, streaming windows forced: one launch per
    window, inserts applied inside it; commits, hits / misses and the miBF equal the serial loop."""
    from goldrush_amd import host, synth

    monkeypatch.setenv("GRP_STREAM", "force")
    monkeypatch.setenv("GRP_BATCH", "off")
    k, h, tile, block = 22, 3, 500, 4
    seeds = default_seeds(h)
    g = synth.random_genome(150_000, 21)
    reads = [r[1] for r in synth.make_reads(g, 150, mean_len=5000, min_len=3500, seed=22, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    exp, mf_ref = _serial(oracle, m, seeds, tile, k, reads, block)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    assert eng.finalize() == mf_ref.pop
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, max_window=max_window)
    cls.run(b._h, b.lens)
    eng.sync()
    assert [c[:8] for c in cls.commits] == exp
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, mf_ref.ids()) and np.array_equal(counts, mf_ref.counts())
    st = cls.state()
    n_ins = sum(1 for e in exp if e[1] in (2, 4))
    assert st["inserts"] == n_ins and n_ins >= 30
    if max_window == 4096:
        assert st["windows"] == 1  # ONE launch for 150 reads and all their inserts
    # the same through the classic form (windows end at inserts): identical
    monkeypatch.setenv("GRP_STREAM_RESUME", "off")
    eng2 = native.Engine(k, h, tile, m, seeds)
    b2 = eng2.upload(reads)
    eng2.bv_insert(b2)
    eng2.finalize()
    cls2 = host.Classifier(eng2._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, max_window=max_window)
    cls2.run(b2._h, b2.lens)
    eng2.sync()
    assert cls2.commits == cls.commits
    assert cls2.state()["windows"] > n_ins // 2
    eng.close()
    eng2.close()


def test_classifier_silver_mode_streams_through_inserts(oracle, native, monkeypatch):
    """Round 4: silver mode with streaming windows forced — the parked launch applies the inserts, the insert behind
    which a silver path is complete ends the launches (the ID array is reset, goldrush_path.cpp:156-187).  Commits
    (paths included) and the final arrays equal the oracle's serial loop."""
    from goldrush_amd import host, synth
    from oracle_engine import serial_reference

    monkeypatch.setenv("GRP_STREAM", "force")
    monkeypatch.setenv("GRP_BATCH", "off")
    k, h, tile, block = 22, 3, 500, 4
    seeds = default_seeds(h)
    g = synth.random_genome(150_000, 23)
    reads = [r[1] for r in synth.make_reads(g, 160, mean_len=5000, min_len=3500, seed=24, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    target = 120_000
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=target, max_paths=4)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    assert eng.finalize() == mf_ref.pop
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, silver_path=True, target_bases=target, max_paths=4)
    cls.run(b._h, b.lens)
    eng.sync()
    assert [c[:8] for c in cls.commits] == exp
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, mf_ref.ids()) and np.array_equal(counts, mf_ref.counts())
    st = cls.state()
    assert len({e[7] for e in exp}) >= 3
    assert st["stream_rollovers"] >= 2 and st["stream_inserts"] >= 20
    eng.close()
