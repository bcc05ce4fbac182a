"""GPU: order-exact classification on the HIP engine (speculative windows) vs the
oracle's serial loop; the committed golden fixture through the CLI."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import default_seeds

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("max_window,mode", [(1, "auto"), (5, "auto"), (4096, "auto"), (4096, "stream"), (48, "stream"), (4096, "pipeline"), (4096, "sync"),
                                             (1, "windows"), (4096, "windows"), (4096, "batch"), (6, "batch"), (4096, "batch_refused")])
def test_hip_classifier_matches_serial_loop(oracle, native, max_window, mode, monkeypatch):
    from goldrush_amd import host, synth
    from oracle_engine import cached_serial_reference

    # auto: the product's own choice (batches while inserts are frequent, streaming windows after); the others: one form forced
    env = {"auto": {}, "windows": {"GRP_BATCH": "off"}, "stream": {"GRP_BATCH": "off", "GRP_STREAM": "force"},
           "pipeline": {"GRP_BATCH": "off", "GRP_STREAM": "off", "GRP_PIPELINE": "force"},
           "sync": {"GRP_BATCH": "off", "GRP_STREAM": "off", "GRP_PIPELINE": "off"},
           "batch": {"GRP_BATCH": "force"},
           # a chain store of 8 entries: every batch whose reads share more than 8 ranks is refused ON THE DEVICE (the owners'
           # touches are taken back from the records), the classifier falls back to the classic commit and halves the batch
           "batch_refused": {"GRP_BATCH": "force", "GRP_BATCH_OVF_CAP": "8"}}[mode]
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    g = synth.random_genome(150_000, 21)
    reads = [r[1] for r in synth.make_reads(g, 120, mean_len=5000, min_len=3500, seed=22, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    exp, ref_ids, ref_counts, ref_pop = cached_serial_reference("classifier_modes", oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=120_000, max_paths=3)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    assert eng.finalize() == ref_pop
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, target_bases=120_000, max_paths=3, silver_path=True, max_window=max_window)
    cls.run(b._h, b.lens)
    eng.sync()
    assert [c[:8] for c in cls.commits] == exp
    # hits / misses per read agree with the oracle's per-tile counters (re-derived on the final state is not
    # possible, so compare the miBF end state instead)
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, ref_ids) and np.array_equal(counts, ref_counts)
    st = cls.state()
    assert st["reads_committed"] == len(exp) and st["inserts"] == sum(1 for e in exp if e[1] in (2, 4))
    if mode == "batch":
        assert st["batches"] >= 2 and st["batch_reads"] > 0
    if mode == "batch_refused":
        assert st["batches_refused"] >= 2 and st["batches"] >= 1


@pytest.mark.parametrize("mode", ["auto", "batch", "stream"])
def test_hip_classifier_on_a_repeat_rich_genome(oracle, native, mode, monkeypatch):
    """The product's classifier (its own choice of windows and batches, batches forced, streaming windows forced) on a
    genome a third of which is repeat copies: commits, IDs and counts of the oracle's serial loop."""
    from goldrush_amd import host, synth
    from oracle_engine import cached_serial_reference

    for key, val in {"auto": {}, "batch": {"GRP_BATCH": "force"}, "stream": {"GRP_BATCH": "off", "GRP_STREAM": "force"}}[mode].items():
        monkeypatch.setenv(key, val)
    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    g = synth.repeat_genome(150_000, 31)
    reads = [r[1] for r in synth.make_reads(g, 160, mean_len=5000, min_len=3500, seed=33, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    exp, ref_ids, ref_counts, ref_pop = cached_serial_reference("classifier_repeats", oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=120_000, max_paths=3)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    assert eng.finalize() == ref_pop
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, target_bases=120_000, max_paths=3, silver_path=True, max_window=4096)
    cls.run(b._h, b.lens)
    eng.sync()
    assert [c[:8] for c in cls.commits] == exp
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, ref_ids) and np.array_equal(counts, ref_counts)
    eng.close()


@pytest.mark.parametrize("mode", ["auto", "batch", "stream", "batch_flagged", "stream_flagged"])
def test_hip_classifier_with_tiles_of_12000_bases(oracle, native, mode, monkeypatch):
    """-t 12000 at h = 3: 36 000 possible IDs per tile do not fit the LDS (rounds 1-4: a loud error at grp_create; the
    reference has no such limit).  The product's windows, batches and streaming launches on that geometry against the
    oracle's serial loop (-x 1500: the reference's threshold is a count per tile, and a tile of 12 000 frames collects
    a few hundred chance hits per ID at this occupancy)."""
    from goldrush_amd import host, synth
    from oracle_engine import cached_serial_reference

    # *_flagged: a first-step table of 840 slots (71 distinct IDs): most tiles are flagged and redone with the worst-case table,
    # which at this geometry lives in global memory — the plain form, the form through a batch's log (the redo launch that
    # reads the list's length on the device included) and the hand-back of a streaming window
    env = {"auto": {}, "batch": {"GRP_BATCH": "force"}, "stream": {"GRP_BATCH": "off", "GRP_STREAM": "force"},
           "batch_flagged": {"GRP_BATCH": "force", "GRP_SMALL_HIST": "840"},
           "stream_flagged": {"GRP_BATCH": "off", "GRP_STREAM": "force", "GRP_SMALL_HIST": "840"}}[mode]
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    tile, k, h, block = 12000, 22, 3, 2
    seeds = default_seeds(h)
    g = synth.random_genome(3_000_000, 23)
    reads = [r[1] for r in synth.make_reads(g, 48, mean_len=200000, min_len=120000, seed=24, max_len=330000)]
    m = oracle.load().orc_calc_optimal_size(12_000_000, 1, 0.1)
    exp, ref_ids, ref_counts, ref_pop = cached_serial_reference("classifier_tile12000", oracle, m, seeds, tile, k, reads, block=block, threshold=1500, u=3, silver=True, target_bases=2_500_000, max_paths=3)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    assert eng.finalize() == ref_pop
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, threshold=1500, unassigned_min=3, k=k, h=h, target_bases=2_500_000, max_paths=3, silver_path=True, max_window=4096)
    cls.run(b._h, b.lens)
    eng.sync()
    assert [c[:8] for c in cls.commits] == exp
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, ref_ids) and np.array_equal(counts, ref_counts)
    st = cls.state()
    assert st["inserts"] == sum(1 for e in exp if e[1] in (2, 4)) and st["inserts"] >= 3
    if mode.endswith("_flagged"):
        assert eng.verify_stats()["window_flagged"] > 0


def test_golden_fixture_through_cli(native, tmp_path):
    from goldrush_amd import host

    fx = json.load(open(os.path.join(GOLD, "fixtures.json")))
    r = subprocess.run([host.CLI_PATH] + fx["tiny_args"] + ["-i", os.path.join(GOLD, "tiny.fq"), "-p", str(tmp_path / "out")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"m_filterSize: {fx['tiny_filter_size']}" in r.stderr
    for name, sha in fx["tiny_outputs"].items():
        assert hashlib.sha256(open(tmp_path / name, "rb").read()).hexdigest() == sha, name


def test_full_size_properties(native):
    """BASELINE-scale geometry (G=100e6 filter, 25 kb reads): size-independent
    properties — fill is idempotent and order-free, pop equals the popcount of the
    exported bits, rank is monotone and consistent with the bits, a query of an
    inserted read returns its own IDs, reset empties everything."""
    from goldrush_amd import host

    k, h, tile = 22, 3, 1000
    seeds = default_seeds(h)
    hl = host.load()
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, 100_000_000, h), 1, 0.1)
    dr = native.synth_reads(3000, 100_000_000)
    eng = native.Engine(k, h, tile, m, seeds)
    rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    eng.bv_insert(rb, 0, 1500)
    eng.bv_insert(rb, 1500, 1500)
    eng.bv_insert(rb, 700, 900)  # again: idempotent
    pop = eng.finalize()
    eng2 = native.Engine(k, h, tile, m, seeds)
    rb2 = eng2.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    eng2.bv_insert(rb2, 2000, 1000)  # different order / slicing
    eng2.bv_insert(rb2, 0, 2000)
    assert eng2.finalize() == pop
    bits = eng.export_bits()
    assert np.array_equal(bits, eng2.export_bits())
    eng2.close()
    assert pop == int(np.unpackbits(bits.view(np.uint8)).sum())
    rng = np.random.default_rng(1)
    pos = np.sort(rng.integers(0, m, size=200_000, dtype=np.uint64))
    bit, rank = eng.rank(pos)
    assert np.all(np.diff(rank.astype(np.int64)) >= 0) and rank[-1] <= pop
    assert np.array_equal(bit, ((bits[pos >> np.uint64(6)] >> (pos & np.uint64(63))) & np.uint64(1)).astype(np.uint8))
    # query after insert: every tile of the inserted read votes for the block's ID
    nt = int(dr.lens[5]) // tile
    for bs in range(0, nt, 10):
        eng.insert_tiles(rb, 5, bs, min(bs + 10, nt), 100 + bs // 10)
    tiles, lists, st = eng.query_tiles(rb, 5, 1)
    assert [int(t["top_id"]) for t in tiles] == [100 + i // 10 for i in range(nt)]
    assert all(int(t["top_count"]) == (min(tile + k - 1, int(dr.lens[5]) - i * tile) - k + 1) for i, t in enumerate(tiles))
    assert st["queries"] == sum(int(t["top_count"]) for t in tiles) and st["hits"] == 3 * st["queries"] and st["misses"] == 0
    eng.reset_ids()
    tiles, lists, st = eng.query_tiles(rb, 5, 1)
    assert not tiles["top_id"].any() and st["hits"] == 0 and st["misses"] == 3 * st["queries"]
    dr.free()


def test_device_decisions_match_host_and_oracle(oracle, native):
    """grp_classify_reads (decision kernel) == host decide on grp_query_tiles output
    == oracle smoothing/decision, on the same miBF state; includes reads longer than
    the kernel LDS scratch (global-scratch path) and very short reads."""
    from goldrush_amd import host, synth

    tile, k, h = 250, 22, 3
    seeds = default_seeds(h)
    g = synth.random_genome(200_000, 31)
    reads = [r[1] for r in synth.make_reads(g, 70, mean_len=6000, min_len=2000, seed=32, max_len=30000)]
    reads += [g[1000:1000 + 20000].tobytes(), g[50000:50000 + 700].tobytes(), g[60000:60000 + 249].tobytes()]  # 80 tiles, 2 tiles, 0 tiles
    m = oracle.load().orc_calc_optimal_size(3_000_000, 1, 0.1)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    eng.finalize()
    nid = 0
    for ri in range(0, 40, 3):  # populate IDs from a subset of the reads
        nt = len(reads[ri]) // tile
        nid += 1
        for bs in range(0, nt, 4):
            eng.insert_tiles(b, ri, bs, min(bs + 4, nt), nid + bs // 4)
        nid += nt // 4
    dec = eng.classify_reads(b, threshold=10, unassigned_min=5, assigned_max=1)
    tiles, lists, _ = eng.query_tiles(b)
    lists_arr = np.ascontiguousarray(lists) if len(lists) else np.zeros(1, dtype=native.id_count_dtype)
    kinds = set()
    for ri in range(len(reads)):
        a, e = int(b.tile0[ri]), int(b.tile0[ri + 1])
        n = e - a
        t = np.ascontiguousarray(tiles[a:e]) if n else np.zeros(1, dtype=native.tile_summary_dtype)
        d = host.decide_read(t, lists_arr, n, threshold=10, unassigned_min=5, assigned_max=1)
        got = dec[ri]
        assert (int(got["kind"]), int(got["num_tiles"]), int(got["num_assigned"]), int(got["trim_start"]), int(got["trim_end"]), int(got["hits"]), int(got["misses"])) == \
               (d.kind, d.num_tiles, d.num_assigned, d.trim_start, d.trim_end, d.hits, d.misses), ri
        # oracle on the same summaries
        ol = [np.array([(int(x["id"]), int(x["count"])) for x in lists[int(tt["list_off"]): int(tt["list_off"]) + int(tt["list_n"])]], dtype=oracle.id_count_dtype) for tt in tiles[a:e]]
        o_ids, o_b, na = oracle.smooth_tiles([int(tt["top_id"]) for tt in tiles[a:e]], ol, 10)
        assert na == int(got["num_assigned"]), ri
        kinds.add(int(got["kind"]))
    assert len(kinds) >= 3, kinds
    assert int(dec[-3]["num_tiles"]) == 80 and int(dec[-1]["num_tiles"]) == 0 and int(dec[-1]["kind"]) == 3
    # sub-window
    sub = eng.classify_reads(b, 10, 7)
    assert np.array_equal(sub, dec[10:17])
    # two windows in flight (grp_classify_reads_begin / _end): same decisions; an insert
    # issued after a _begin is ordered behind that window; an abandoned slot is reusable
    n = len(reads)
    eng.classify_begin(b, 0, 30, 0)
    eng.classify_begin(b, 30, n - 30, 1)
    with pytest.raises(native.GrpError):
        eng.classify_begin(b, 0, 1, 1)  # slot busy
    ri = 41
    eng.insert_tiles(b, ri, 0, len(reads[ri]) // tile, 7777)  # changes read 41's own decision afterwards
    w0 = eng.classify_end(0)
    w1 = eng.classify_end(1)
    assert np.array_equal(w0, dec[:30]) and np.array_equal(w1, dec[30:])
    with pytest.raises(native.GrpError):
        eng.classify_end(1)  # nothing in flight
    eng.classify_begin(b, 0, n, 0)
    eng.classify_end(0, abandon=True)
    eng.classify_begin(b, 5, 3, 1)
    eng.classify_begin(b, 0, n, 0)
    after = eng.classify_end(0)
    assert np.array_equal(eng.classify_end(1), after[5:8])
    assert np.array_equal(after, eng.classify_reads(b))
    assert int(after[ri]["num_assigned"]) == int(after[ri]["num_tiles"]) > int(dec[ri]["num_assigned"])
    # empty window
    eng.classify_begin(b, 3, 0, 0)
    assert len(eng.classify_end(0)) == 0
    # streaming windows (grp_classify_stream_*): every record appears, equal to the window's
    # decisions; two launches queued; abort leaves the slot reusable
    import time

    def first_stop(kinds):  # the window parks itself behind the first insert / hand-back record
        for i, kd in enumerate(kinds):
            if int(kd) in (0, 2, 4):
                return i
        return len(kinds) - 1

    def wait_all(view, slot, expect):
        """Wait for the launch; every record up to the first insert / hand-back one must be there."""
        t0 = time.time()
        while not eng.stream_poll(slot):
            assert time.time() - t0 < 60, "launch does not end"
        must = first_stop(expect["kind"]) + 1
        assert np.all(view["pad"][:must] == 1), (must, view["pad"][:must])
        return view["pad"] == 1

    ref = eng.classify_reads(b)
    v0 = eng.stream_begin(b, 0, 30, 0)
    v1 = eng.stream_begin(b, 30, n - 30, 1)
    with pytest.raises(native.GrpError):
        eng.stream_begin(b, 0, 1, 1)
    d0 = wait_all(v0, 0, ref[:30])
    d1 = wait_all(v1, 1, ref[30:])
    fields = ["kind", "num_tiles", "num_assigned", "trim_start", "trim_end", "hits", "misses"]
    for f in fields:  # whatever completed is right
        assert np.array_equal(v0[f][d0], ref[f][:30][d0]) and np.array_equal(v1[f][d1], ref[f][30:][d1]), f
    assert eng.stream_end(0) == int(d0.sum()) and eng.stream_end(1) == int(d1.sum())
    with pytest.raises(native.GrpError):
        eng.stream_end(0)
    # walking the whole batch the way the classifier does: restart behind every parking record
    pos, seen = 0, np.zeros(n, dtype=bool)
    while pos < n:
        v = eng.stream_begin(b, pos, n - pos, 0)
        done = wait_all(v, 0, ref[pos:])
        stop = first_stop(ref["kind"][pos:])
        for f in fields:
            assert np.array_equal(v[f][: stop + 1], ref[f][pos: pos + stop + 1]), (f, pos)
        eng.stream_end(0)
        seen[pos: pos + stop + 1] = True
        pos += stop + 1
    assert seen.all()
    v = eng.stream_begin(b, 0, n, 1)
    eng.stream_abort(1)
    decided = eng.stream_end(1)
    assert 0 <= decided <= n
    done = v["pad"] == 1
    for f in fields:  # whatever completed is right
        assert np.array_equal(v[f][done], ref[f][done]), f
    assert len(eng.stream_begin(b, 4, 0, 0)) == 0 and eng.stream_end(0) == 0
    assert np.array_equal(eng.classify_reads(b), ref)
    # striped window (several ranks share it): only the owner's stripes are worked on, up to
    # the owner's first insert / hand-back record
    zero_tiles = np.array([len(r) // tile == 0 for r in reads])
    for owner in range(3):
        v = eng.stream_begin(b, 0, n, owner & 1, stripe=7, n_owners=3, owner=owner)
        mine = (np.arange(n) // 7) % 3 == owner
        t0 = time.time()
        while not eng.stream_poll(owner & 1):
            assert time.time() - t0 < 60
        idx = np.flatnonzero(mine)
        stop = idx[first_stop(ref["kind"][idx])]
        must = mine & (np.arange(n) <= stop)
        assert np.all(v["pad"][must] == 1) and not np.any(v["pad"][~mine & ~zero_tiles])
        done = (v["pad"] == 1) & mine
        for f in fields:
            assert np.array_equal(v[f][done], ref[f][done]), f
        assert eng.stream_end(owner & 1) == int(v["pad"].sum())


def test_hip_classifier_h5_designed_seed(oracle, native):
    """C4-like geometry through the whole product path: h = 5 designed (not preset)
    seeds, 5 silver paths, tile 300, ID blocks of 3 tiles."""
    from goldrush_amd import host, synth
    from oracle_engine import serial_reference

    tile, k, h, block = 300, 20, 5, 3
    seeds = host.make_seed_pattern("", k, 14, h)
    assert seeds == oracle.make_seed_pattern("", k, 14, h)
    g = synth.random_genome(120_000, 41)
    reads = [r[1] for r in synth.make_reads(g, 90, mean_len=4000, min_len=2500, seed=42, max_len=8000)]
    m = oracle.load().orc_calc_optimal_size(2_500_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block, threshold=8, silver=True, target_bases=60_000, max_paths=5)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    assert eng.finalize() == mf_ref.pop
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, threshold=8, k=k, h=h, target_bases=60_000, max_paths=5, silver_path=True)
    cls.run(b._h, b.lens)
    eng.sync()
    assert [c[:8] for c in cls.commits] == exp
    assert len(cls.rollovers) >= 2
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, mf_ref.ids()) and np.array_equal(counts, mf_ref.counts())


@pytest.mark.parametrize("mode", ["stream", "pipeline"])
def test_long_reads_small_tiles_all_decision_paths(oracle, native, mode, monkeypatch):
    """tile = 100: reads of 40 .. 320 tiles take every decision path of the device —
    register state (<= 64 tiles), LDS arrays (<= 256), and, in a streaming window, the
    hand-back to the synchronous path (kind 0) for longer reads."""
    from goldrush_amd import host, synth
    from oracle_engine import cached_serial_reference

    for key, val in ({"GRP_BATCH": "off", "GRP_STREAM": "force"} if mode == "stream" else {"GRP_BATCH": "off", "GRP_STREAM": "off", "GRP_PIPELINE": "force"}).items():
        monkeypatch.setenv(key, val)
    tile, k, h, block = 100, 22, 3, 10
    seeds = default_seeds(h)
    g = synth.random_genome(120_000, 41)
    reads = [r[1] for r in synth.make_reads(g, 70, mean_len=9000, min_len=4000, seed=42, max_len=32000)]
    reads += [g[5000:5000 + 30500].tobytes(), g[40000:40000 + 26000].tobytes(), g[70000:70000 + 6400].tobytes()]
    assert max(len(r) for r in reads) // tile > 256 and any(64 < len(r) // tile <= 256 for r in reads)
    m = oracle.load().orc_calc_optimal_size(3_000_000, 1, 0.1)
    exp, ref_ids, ref_counts, ref_pop = cached_serial_reference("long_reads_small_tiles", oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=100_000, max_paths=4)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    assert eng.finalize() == ref_pop
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, target_bases=100_000, max_paths=4, silver_path=True, max_window=64)
    cls.run(b._h, b.lens)
    eng.sync()
    assert [c[:8] for c in cls.commits] == exp
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, ref_ids) and np.array_equal(counts, ref_counts)
    kinds = [e[1] for e in exp]
    assert kinds.count(2) >= 3 and (kinds.count(3) + kinds.count(5)) >= 10, kinds


@pytest.mark.parametrize("h,n", [(3, 40_000), (5, 16_000)])
def test_full_size_streaming_equals_synchronous_windows(native, monkeypatch, h, n):
    """BASELINE-scale determinism: 40 000 reads of the C1 stream (G = 8e6 here, so that the
    golden path keeps inserting) classified three times on fresh engines — streaming
    windows, pipelined windows, synchronous windows — must commit the same decisions and
    leave the same IDs / counts behind.  The streaming launch hands summaries from one
    workgroup to another inside a launch (coherent stores, no fences) and is aborted
    hundreds of times here: any stale hand-over would show."""
    from goldrush_amd import host

    k, tile, block, G = 22, 1000, 10, 8_000_000
    seeds = default_seeds(h)
    hl = host.load()
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, G, h), 1, 0.1)
    dr = native.synth_reads(n, G)
    lens = np.ascontiguousarray(dr.lens, dtype=np.uint32)
    results = []
    # the first run also commits its insert-heavy start as batches (the product's default), the others are classic windows only
    for mode in ({"GRP_STREAM": "force"}, {"GRP_BATCH": "off", "GRP_STREAM": "off", "GRP_PIPELINE": "force"}, {"GRP_BATCH": "off", "GRP_STREAM": "off", "GRP_PIPELINE": "off"}):
        for key in ("GRP_STREAM", "GRP_PIPELINE", "GRP_BATCH"):
            monkeypatch.delenv(key, raising=False)
        for key, val in mode.items():
            monkeypatch.setenv(key, val)
        eng = native.Engine(k, h, tile, m, seeds)
        rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
        eng.bv_insert(rb)
        eng.finalize()
        cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, target_bases=int(0.9 * G), max_paths=1, silver_path=False)
        for first in range(0, n, 8192):
            cls.run_range(rb._h, lens, first, min(8192, n - first))
        eng.sync()
        st = cls.state()
        ids, counts = eng.export_ids()
        # the whole commit record: decision, first ID, path AND the hits / misses of the read's query
        results.append(([c[:10] for c in cls.commits], ids.copy(), counts.copy(), st["inserts"], st["windows"]))
        cls.close()
        eng.close()
    a = results[0]
    assert a[3] > 200 and len(a[0]) == n  # many aborts in the streaming run
    for b in results[1:]:
        assert a[0] == b[0]
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert results[0][4] < results[2][4]  # the streaming run needs far fewer launches
    dr.free()


def test_c1_stream_streaming_equals_synchronous_windows(native, monkeypatch):
    """The race hunt of tools/stress_stream.py inside the suite: 200 000 reads of the C1 stream
    (G = 100e6, 25 kb reads: the filter, bucket layout and insert rate of the benchmark) through
    the streaming windows — one workgroup hands summaries to another inside a launch, the
    launches are aborted and parked at every insert — and through the synchronous windows:
    every commit and a 20 M-rank sample of the ID / count arrays must agree."""
    from goldrush_amd import host

    k, h, tile, block, G, n = 22, 3, 1000, 10, 100_000_000, 200_000
    seeds = default_seeds(h)
    hl = host.load()
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, G, h), 1, 0.1)
    dr = native.synth_reads(n, G)
    lens = np.ascontiguousarray(dr.lens, dtype=np.uint32)
    results = []
    # batches + streaming windows (the product's default) against classic synchronous windows
    for mode in ({"GRP_STREAM": "force"}, {"GRP_BATCH": "off", "GRP_STREAM": "off", "GRP_PIPELINE": "off"}):
        for key in ("GRP_STREAM", "GRP_PIPELINE", "GRP_BATCH"):
            monkeypatch.delenv(key, raising=False)
        for key, val in mode.items():
            monkeypatch.setenv(key, val)
        eng = native.Engine(k, h, tile, m, seeds)
        rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
        eng.bv_insert(rb)
        pop = eng.finalize()
        cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, target_bases=int(0.9 * G), max_paths=1, silver_path=False)
        for first in range(0, n, 8192):
            cls.run_range(rb._h, lens, first, min(8192, n - first))
        eng.sync()
        st = cls.state()
        ids, counts = eng.export_ids(pop // 3, 20_000_000)
        results.append(([c[:10] for c in cls.commits], ids, counts, {key: st[key] for key in ("inserts", "hits", "misses", "queries", "ids_inserted", "inserted_bases")}))
        cls.close()
        eng.close()
    dr.free()
    a, b = results
    assert a[3] == b[3] and a[3]["inserts"] > 5000
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and int((a[1] != 0).sum()) > 100_000


def test_window_overlap_names_the_reads_that_overlap_one_in_front_of_them(oracle, native):
    """grp_window_overlap (round 4): a hash-only pass over a range of reads — one frame in 32 under seed 0, canonical,
    so either strand — returns per read the closest read in front of it that owns >= threshold of its samples.
    Error-free reads with known positions: 5 kb shared is ~150 samples, 1 kb ~30, unrelated reads ~0."""
    from goldrush_amd import synth

    NONE = 0xFFFFFFFF
    k, h, tile = 22, 3, 1000
    g = synth.random_genome(400_000, 91)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = [g[0:10_000].tobytes(), g[50_000:60_000].tobytes(), g[200_000:210_000].tobytes(), g[52_000:62_000].tobytes().translate(comp)[::-1], g[100_000:110_000].tobytes(),
             g[5_000:15_000].tobytes(), g[109_000:112_500].tobytes(), g[300_000:300_900].tobytes(), g[8_000:14_000].tobytes()]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    eng = native.Engine(k, h, tile, m, default_seeds(h))
    b = eng.upload(reads)  # (no fill, no finalize: the call reads nothing of the filter)
    # read 3 = the other strand of read 1's second half; 5 overlaps 0; 6 overlaps 4 by 1 kb; 7 has no tile; 8 overlaps 0 AND 5:
    # the samples of 8's first 2 kb are owned by read 0 (the first holder), the 4 kb behind them by read 5 -> the closest is 5
    assert list(eng.window_overlap(b, 0, 9)) == [NONE, NONE, NONE, 1, NONE, 0, 4, NONE, 5]
    assert list(eng.window_overlap(b, 0, 3)) == [NONE] * 3
    assert list(eng.window_overlap(b, 2, 7)) == [NONE, NONE, NONE, NONE, 2, NONE, 3]   # [2, 9): 3 and 5 have lost their partners, 8 overlaps 5 (index 3)
    assert list(eng.window_overlap(b, 2, 7, threshold=70)) == [NONE, NONE, NONE, NONE, NONE, NONE, 3]  # 6 shares only ~30 samples with 4, 8 ~125 with 5
    assert list(eng.window_overlap(b, 0, 1)) == [NONE] and len(eng.window_overlap(b, 0, 0)) == 0
    # the engine is as usable as ever
    eng.bv_insert(b)
    eng.finalize()
    assert len(eng.classify_reads(b)) == 9
    eng.close()
