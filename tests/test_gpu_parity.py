"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on the
same seeded inputs.  Integer / bit / index work: the bar is bit-exact."""
import numpy as np
import pytest

from helpers import SEED22, canon_list, default_seeds, random_reads

pytestmark = pytest.mark.gpu

K, TILE = 22, 1000


def _mk(oracle, native, h=3, m=None, tile=TILE, k=K, preset=SEED22):
    seeds = default_seeds(h, preset)
    m = m or oracle.load().orc_calc_optimal_size(3_000_000, 1, 0.1)
    eng = native.Engine(k, h, tile, m, seeds)
    oseeds = oracle.Seeds(seeds)
    omf = oracle.MiBF(m, oseeds, tile, k)
    return eng, oseeds, omf, m


@pytest.mark.parametrize("h", [1, 3, 5])
def test_tile_hashes_match_oracle(oracle, native, h):
    eng, oseeds, omf, m = _mk(oracle, native, h=h)
    # lengths chosen to hit: exact multiple of tile, last tile clipped to
    # fewer than `tile` frames (len % tile < k-1), stale frames of longer seeds
    reads = random_reads(3, 2000, 6000, seed=11)
    reads += [random_reads(1, 3000, 3000, 12)[0], random_reads(1, 3005, 3005, 13)[0], random_reads(1, 3021, 3021, 14)[0],
              random_reads(1, 3020, 3020, 15)[0], random_reads(1, 1000, 1000, 16)[0]]
    b = eng.upload(reads)
    for ri, seq in enumerate(reads):
        for t in range(len(seq) // TILE):
            got = eng.tile_hashes(b, ri, t)
            exp = oseeds.tile_hashes(seq, TILE, K, t)
            assert got.shape == exp.shape, (ri, t)
            assert np.array_equal(got, exp), (ri, t)


def test_nthash_known_answers_btllib(native):
    """The HIP hash kernel against the known answers of btllib's own test-suite
    (tests/nthash.cpp, ACATGCATGCA, k = 5; see tests/test_oracle.py)."""
    eng = native.Engine(5, 1, 7, 1 << 16, ["11111"])
    b = eng.upload([b"ACATGCATGCA"])
    got = eng.tile_hashes(b, 0, 0)  # tile 0 = substr(0, 7 + 5 - 1): all 7 frames
    assert [int(x) for x in got[:3]] == [0xF59ECB45F0E22B9C, 0x38CC00F940AEBDAE, 0x603A48C5A11C794A]
    assert len(got) == 7
    eng.close()


def _device_streams(native):
    """per seed of a family, every position's hash of one read as the DEVICE computes it: the family's engine (seeds share
    a frame's halves, grp_kernels.inc seed_halves) with one tile covering the whole read; seed j is stale in the tile's
    last j frames (multiLensfrHashIterator.hpp:54-60), which SeedNtHash itself never emits — cut off"""
    def hashes_of(seeds, seq):
        k, h = len(seeds[0]), len(seeds)
        tile = len(seq) - k + 1
        eng = native.Engine(k, h, tile, 1 << 20, seeds)
        b = eng.upload([seq])
        got = eng.tile_hashes(b, 0, 0).reshape(tile, h)
        eng.close()
        return [got[: tile - j, j] for j in range(h)]
    return hashes_of


def test_seed_hashes_match_a_real_btllib(native):
    """The device's side of the pin of rows a2 / a3 (tools/make_btllib_kat.py, tests/test_oracle.py): the HIP hashing,
    through the C ABI, against values a real btllib produced.  Skips while tests/golden/btllib_seed_kat.json is absent
    (btllib is neither in this image nor on the GPU box)."""
    from helpers import check_against_btllib_kat, load_btllib_kat

    kat = load_btllib_kat()
    if kat is None:
        pytest.skip("no tests/golden/btllib_seed_kat.json: run tools/make_btllib_kat.py where btllib is installed")
    assert check_against_btllib_kat(kat, _device_streams(native)) > 0


def test_device_side_of_the_btllib_pin_hook(oracle, native):
    """The same checker over a file of the SAME shape whose values are the oracle's (a stand-in: pins nothing): the three
    families of the tool — the pipeline's seeds at h = 3 and h = 5, spans 60..64 — over the reads of tiny.fq, every
    position, device against oracle.  Shows the device-side hook is live."""
    import importlib.util
    import os

    from helpers import check_against_btllib_kat

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_btllib_kat", os.path.join(root, "tools", "make_btllib_kat.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kat = {"stride": mod.STRIDE, "families": {}}
    for name, seeds in mod.FAMILIES.items():
        kat["families"][name] = {"seeds": seeds, "reads": {rid: [mod.summarise([int(v) for v in oracle.Seeds([sd]).multi_hash(seq.encode())]) for sd in seeds]
                                                           for rid, seq in mod.tiny_reads(6)}}
    assert check_against_btllib_kat(kat, _device_streams(native)) == (3 + 5 + 5) * 6


def test_fill_bits_pop_rank(oracle, native):
    eng, oseeds, omf, m = _mk(oracle, native)
    reads = random_reads(12, 1500, 9000, seed=21) + [b"ACGT" * 6, b"A" * 25, b"ACGTTGCA" * 40]
    b = eng.upload(reads)
    eng.bv_insert(b, 0, 5)
    eng.bv_insert(b, 5)        # second call; fill is order-free and idempotent
    eng.bv_insert(b, 2, 3)     # repeat some reads
    for s in reads:
        if len(s) >= K + 3 - 1:  # shorter reads are reference-undefined
            omf.bv_insert_read(s)
    pop = eng.finalize()
    assert pop == omf.finalize()
    assert np.array_equal(eng.export_bits(), omf.bits())
    rng = np.random.default_rng(5)
    pos = rng.integers(0, m, size=20000, dtype=np.uint64)
    pos[:4] = [0, 1, m - 1, m - 2]
    bit, rank = eng.rank(pos)
    for i in range(0, pos.size, 7):
        assert bit[i] == omf.bit(int(pos[i]))
        assert rank[i] == omf.rank(int(pos[i]))


def _compare_queries(eng, omf, batch, reads):
    tiles, lists, stats = eng.query_tiles(batch)
    ti = 0
    q = h = ms = 0
    for seq in reads:
        for res in omf.query_read(seq):
            top_id, top_count, lst, ctr = res
            t = tiles[ti]
            assert (int(t["top_id"]), int(t["top_count"])) == (top_id, top_count), ti
            got = [(int(a), int(c)) for a, c in lists[t["list_off"]: t["list_off"] + t["list_n"]]]
            assert got == canon_list(lst), ti
            q += ctr[0]; h += ctr[1]; ms += ctr[2]
            ti += 1
    assert ti == len(tiles)
    assert (stats["queries"], stats["hits"], stats["misses"]) == (q, h, ms)


def test_insert_and_query_match_oracle(oracle, native):
    eng, oseeds, omf, m = _mk(oracle, native)
    genome = random_reads(1, 60000, 60000, seed=31)[0]
    reads = random_reads(10, 3000, 12000, seed=32, genome=genome) + random_reads(2, 4000, 5000, seed=33)
    b = eng.upload(reads)
    eng.bv_insert(b)
    for s in reads:
        omf.bv_insert_read(s)
    assert eng.finalize() == omf.finalize()
    _compare_queries(eng, omf, b, reads)  # empty ID array: all misses
    # inserts: whole reads in blocks of 10 tiles (goldrush_path.cpp:982-994),
    # overlapping reads so that counts > 1 and the reservoir rule is exercised
    next_id = 0
    for ri in (0, 3, 5, 1, 7):
        seq = reads[ri]
        nt = len(seq) // TILE
        next_id += 1
        for bs in range(0, nt, 10):
            be = min(bs + 10, nt)
            eng.insert_tiles(b, ri, bs, be, next_id + bs // 10)
            omf.insert_read_tiles(seq, bs, be, next_id + bs // 10)
        next_id += len(seq) // (TILE * 10)
        ids, counts = eng.export_ids()
        assert np.array_equal(ids, omf.ids())
        assert np.array_equal(counts, omf.counts())
    _compare_queries(eng, omf, b, reads)
    # partial (trimmed) insert with a dedup scope that is not tile-0 based
    eng.insert_tiles(b, 8, 1, 3, 77)
    omf.insert_read_tiles(reads[8], 1, 3, 77)
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts())
    _compare_queries(eng, omf, b, reads)
    # reset (silver path rollover, goldrush_path.cpp:180-181)
    eng.reset_ids(); omf.reset_ids()
    ids, counts = eng.export_ids()
    assert not ids.any() and not counts.any()
    _compare_queries(eng, omf, b, reads)


def test_query_dense_random_ids_and_saturation_bit(oracle, native):
    """Many distinct IDs per tile, ties, saturated (bit 31) values."""
    eng, oseeds, omf, m = _mk(oracle, native, m=oracle.load().orc_calc_optimal_size(200_000, 1, 0.1))
    reads = random_reads(6, 2500, 5200, seed=41)
    b = eng.upload(reads)
    eng.bv_insert(b)
    for s in reads:
        omf.bv_insert_read(s)
    pop = eng.finalize()
    assert pop == omf.finalize()
    rng = np.random.default_rng(7)
    for n_ids in (3, 40, 100000):
        ids = rng.integers(0, n_ids, size=pop, dtype=np.uint32)  # 0 = empty
        sat = rng.random(pop) < 0.05
        ids[sat] |= np.uint32(0x80000000)  # includes bare 0x80000000 ("saturated empty")
        eng.import_ids(0, ids=ids, counts=np.zeros(pop, dtype=np.uint32))
        omf.ids()[:] = ids
        _compare_queries(eng, omf, b, reads)


@pytest.mark.parametrize("h", [5, 8])
def test_tiles_where_every_probe_returns_another_id(oracle, native, h):
    """(Almost) every probe of a tile another ID: tile * h distinct IDs.
    h = 5, tile 1000 — up to 5000 IDs: since round 4 the count table (6 bytes per slot, 7168 slots) holds the worst
    case of the default geometries, nothing is flagged and a streaming window decides every read itself (at
    human-genome scale half of a read's error k-mers are some other locus's k-mer: ~2500 distinct IDs per tile are
    the NORMAL case of the C4 geometry, which overflowed the 4096-slot table of rounds 1-3 on 40 % of the tiles).
    h = 8 — up to 8000 IDs do not fit the LDS share of the first launch: the tile is flagged and recomputed with the
    worst-case table, in grp_query_tiles, in grp_classify_reads and (kind 0 hand-back) in a streaming window.
    Same results as the oracle either way."""
    eng, oseeds, omf, m = _mk(oracle, native, h=h, m=oracle.load().orc_calc_optimal_size(300_000, 1, 0.1))
    reads = random_reads(5, 2500, 5200, seed=43)
    b = eng.upload(reads)
    eng.bv_insert(b)
    for s in reads:
        omf.bv_insert_read(s)
    pop = eng.finalize()
    assert pop == omf.finalize()
    rng = np.random.default_rng(8)
    ids = rng.integers(1, 1 << 30, size=pop, dtype=np.uint32)  # (almost) every probe another ID
    eng.import_ids(0, ids=ids, counts=np.zeros(pop, dtype=np.uint32))
    omf.ids()[:] = ids
    _compare_queries(eng, omf, b, reads)
    dec = eng.classify_reads(b)
    assert all(int(d["num_tiles"]) == len(r) // TILE for d, r in zip(dec, reads))
    assert (eng.verify_stats()["window_flagged"] > 0) == (h == 8)
    v = eng.stream_begin(b, 0, len(reads), 0)
    import time
    t0 = time.time()
    while not eng.stream_poll(0):
        assert time.time() - t0 < 60
    done = v["pad"] == 1
    if h == 8:
        # flagged tiles: the read is handed back to the synchronous path (kind 0) and the window
        # parks itself behind that record
        first = int(np.flatnonzero(done & (v["kind"] == 0))[0])
        assert np.all(done[: first + 1])
    else:
        # nothing is handed back; the window parks itself at its first insert record (kinds 2 / 4), whatever of the
        # reads behind it had not been handed out by then stays undecided (the host would begin again behind it)
        ins = np.flatnonzero(done & ((v["kind"] == 2) | (v["kind"] == 4)))
        upto = int(ins[0]) + 1 if ins.size else len(reads)
        assert np.all(done[:upto]) and np.all(v["kind"][done] != 0)
    ok = done & (v["kind"] != 0)
    for f in ("kind", "num_tiles", "num_assigned", "trim_start", "trim_end", "hits", "misses"):
        assert np.array_equal(v[f][ok], dec[f][ok]), f
    eng.stream_end(0)


@pytest.mark.parametrize("tile,h", [(12000, 3), (20000, 2), (9000, 5)])
def test_tiles_whose_worst_case_table_does_not_fit_the_lds(oracle, native, tile, h):
    """tile x h distinct IDs beyond what 160 KB of LDS hold (the reference has no such limit; rounds 1-4 refused these
    geometries): the first step's table is LDS-sized, the tiles it cannot hold are redone with the table in global
    memory (k_query<.., GT>) - in grp_query_tiles and in grp_classify_reads.  Every probe another ID = the worst case;
    a handful of IDs = the first step alone.  Same results as the oracle either way."""
    eng, oseeds, omf, m = _mk(oracle, native, h=h, tile=tile, m=oracle.load().orc_calc_optimal_size(600_000, 1, 0.1))
    reads = random_reads(4, 2 * tile + 300, 3 * tile + 900, seed=71 + h)
    b = eng.upload(reads)
    eng.bv_insert(b)
    for s in reads:
        omf.bv_insert_read(s)
    pop = eng.finalize()
    assert pop == omf.finalize()
    rng = np.random.default_rng(9)
    for n_ids, flagged in ((1 << 30, True), (25, False)):
        ids = rng.integers(1, n_ids, size=pop, dtype=np.uint32)
        eng.import_ids(0, ids=ids, counts=np.zeros(pop, dtype=np.uint32))
        omf.ids()[:] = ids
        before = eng.verify_stats()["window_flagged"]
        _compare_queries(eng, omf, b, reads)
        dec = eng.classify_reads(b)
        assert all(int(d["num_tiles"]) == len(r) // tile for d, r in zip(dec, reads))
        assert (eng.verify_stats()["window_flagged"] > before) == flagged


@pytest.mark.parametrize("h,tile", [(1, 1000), (5, 1000), (3, 500), (2, 64), (4, 1000), (6, 250), (7, 1000), (4, 190)])
def test_other_geometries(oracle, native, h, tile):
    k = 22
    seeds = default_seeds(h)
    m = oracle.load().orc_calc_optimal_size(400_000, 1, 0.1)
    eng = native.Engine(k, h, tile, m, seeds)
    oseeds = oracle.Seeds(seeds)
    omf = oracle.MiBF(m, oseeds, tile, k)
    reads = random_reads(5, 2 * tile, 6 * tile + 40, seed=51 + h)
    b = eng.upload(reads)
    eng.bv_insert(b)
    for s in reads:
        omf.bv_insert_read(s)
    assert eng.finalize() == omf.finalize()
    assert np.array_equal(eng.export_bits(), omf.bits())
    for ri in (0, 2):
        nt = len(reads[ri]) // tile
        eng.insert_tiles(b, ri, 0, nt, ri + 1)
        omf.insert_read_tiles(reads[ri], 0, nt, ri + 1)
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts())
    tiles, lists, stats = eng.query_tiles(b)
    ti = 0
    for seq in reads:
        for res in omf.query_read(seq):
            t = tiles[ti]
            assert (int(t["top_id"]), int(t["top_count"])) == (res[0], res[1])
            got = [(int(a), int(c)) for a, c in lists[t["list_off"]: t["list_off"] + t["list_n"]]]
            assert got == canon_list(res[2])
            ti += 1


def _symmetric_seed(k, weight, seed):
    """A palindromic care pattern of span k (ends are care positions), like the reference's designed seeds."""
    rng = np.random.default_rng(seed)
    half = k // 2
    left = np.zeros(half, dtype=bool)
    left[0] = True
    left[rng.choice(np.arange(1, half), size=max(weight // 2 - 1, 0), replace=False)] = True
    s = "".join("1" if b else "0" for b in left)
    return s + ("1" if k % 2 else "") + s[::-1]


@pytest.mark.parametrize("k,h,tile", [(31, 3, 400), (32, 1, 300), (33, 2, 400), (40, 3, 500), (48, 5, 700), (62, 3, 400), (64, 1, 1000)])
def test_spans_beyond_one_window(oracle, native, k, h, tile):
    """k + h - 1 up to 64 bases: the frames of seeds wider than 32 bases are hashed from two
    64-bit windows of 2-bit bases.  Hashes, fill, inserts and queries against the oracle, bit for bit."""
    base = _symmetric_seed(k, min(k // 2 * 2 - 2, 24), 7 + k)
    seeds = default_seeds(h, base)
    assert all(len(sd) == k + i for i, sd in enumerate(seeds)) and len(seeds[-1]) <= 64
    m = oracle.load().orc_calc_optimal_size(300_000, 1, 0.1)
    eng = native.Engine(k, h, tile, m, seeds)
    oseeds = oracle.Seeds(seeds)
    omf = oracle.MiBF(m, oseeds, tile, k)
    reads = random_reads(5, 2 * tile, 6 * tile + 70, seed=151 + k)
    reads += [reads[0][: 3 * tile // 2 + k - 2], reads[1][: 2 * tile + k + h - 2], b"ACGT" * (tile // 2), reads[2][: tile + 3]]
    b = eng.upload(reads)
    for ri, seq in enumerate(reads):
        for t in range(len(seq) // tile):
            assert np.array_equal(eng.tile_hashes(b, ri, t), oseeds.tile_hashes(seq, tile, k, t)), (ri, t)
    eng.bv_insert(b)
    for sq in reads:
        omf.bv_insert_read(sq)
    assert eng.finalize() == omf.finalize()
    assert np.array_equal(eng.export_bits(), omf.bits())
    for ri in (0, 2, 5):
        nt = len(reads[ri]) // tile
        eng.insert_tiles(b, ri, 0, nt, ri + 1)
        omf.insert_read_tiles(reads[ri], 0, nt, ri + 1)
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts())
    _compare_queries(eng, omf, b, reads)
    # the classification window (hash + query + decisions in one call) against the host decision on the queried tiles
    from goldrush_amd import host
    dp = dict(threshold=2, unassigned_min=2, assigned_max=1 << 30)
    dec = eng.classify_reads(b, 0, len(reads), **dp)
    tiles, lists, _ = eng.query_tiles(b)
    lists_arr = np.ascontiguousarray(lists) if len(lists) else np.zeros(1, dtype=native.id_count_dtype)
    for ri in range(len(reads)):
        a0, e0 = int(b.tile0[ri]), int(b.tile0[ri + 1])
        t = np.ascontiguousarray(tiles[a0:e0]) if e0 > a0 else np.zeros(1, dtype=native.tile_summary_dtype)
        d = host.decide_read(t, lists_arr, e0 - a0, **dp)
        got = dec[ri]
        assert (int(got["kind"]), int(got["num_tiles"]), int(got["num_assigned"]), int(got["hits"]), int(got["misses"])) == (d.kind, d.num_tiles, d.num_assigned, d.hits, d.misses), ri
    eng.close()


def test_errors_are_loud(native):
    with pytest.raises(native.GrpError):
        native.Engine(22, 3, 10, 1 << 20, default_seeds(3))  # tile shorter than the seeds
    with pytest.raises(native.GrpError):
        native.Engine(63, 3, 1000, 1 << 20, ["1" * 63, "1" * 64, "1" * 65])  # span k + h - 1 > 64
    with pytest.raises(native.GrpError):
        native.Engine(22, 3, 70000, 1 << 20, default_seeds(3))  # more than 65 535 frames per tile (16-bit counts)
    eng = native.Engine(22, 3, 1000, 1 << 20, default_seeds(3))
    b = eng.upload([b"ACGT" * 600])
    with pytest.raises(native.GrpError):
        eng.query_tiles(b)  # before finalize
    eng.bv_insert(b)
    eng.finalize()
    with pytest.raises(native.GrpError):
        eng.bv_insert(b)  # bit vector immutable after finalize


def test_bucket_overflow_side_table(oracle, native):
    """Occupancy ~0.2 -> buckets of W~30 bits, some with more than 13 set bits: their
    14th.. IDs live in the side table; inserts, queries, export and reset must still
    match the oracle bit for bit."""
    lib = oracle.load()
    m = 300_032
    eng, oseeds, omf, _ = _mk(oracle, native, m=m)
    reads = random_reads(6, 3000, 5000, seed=61)
    b = eng.upload(reads)
    eng.bv_insert(b)
    for s in reads:
        omf.bv_insert_read(s)
    pop = eng.finalize()
    assert pop == omf.finalize()
    assert 0.12 < pop / m < 0.35
    bits = omf.bits()
    # at least one 64-bit aligned window of the chosen width holds > 13 set bits
    assert np.array_equal(eng.export_bits(), bits)
    rng = np.random.default_rng(3)
    ids = rng.integers(1, 50, size=pop, dtype=np.uint32)
    cnt = rng.integers(0, 4, size=pop, dtype=np.uint32)
    eng.import_ids(0, ids=ids, counts=cnt)
    omf.ids()[:] = ids
    omf.counts()[:] = cnt
    gi, gc = eng.export_ids()
    assert np.array_equal(gi, ids) and np.array_equal(gc, cnt)
    _compare_queries(eng, omf, b, reads)
    for ri in (0, 2, 4):
        nt = len(reads[ri]) // TILE
        eng.insert_tiles(b, ri, 0, nt, 1000 + ri)
        omf.insert_read_tiles(reads[ri], 0, nt, 1000 + ri)
    gi, gc = eng.export_ids()
    assert np.array_equal(gi, omf.ids()) and np.array_equal(gc, omf.counts())
    _compare_queries(eng, omf, b, reads)
    eng.reset_ids()
    omf.reset_ids()
    gi, gc = eng.export_ids()
    assert not gi.any() and not gc.any()
    # partial export / import windows
    eng.import_ids(100, ids=np.arange(1, 51, dtype=np.uint32))
    gi, _ = eng.export_ids(90, 70)
    assert gi[:10].sum() == 0 and np.array_equal(gi[10:60], np.arange(1, 51)) and gi[60:].sum() == 0


def test_ragged_and_degenerate_batches(oracle, native):
    """Empty batch, reads shorter than a tile (0 tiles), exactly one tile, one and two
    tiles (no smoothing), a read whose last tile is clipped to a single frame."""
    eng, oseeds, omf, m = _mk(oracle, native)
    empty = eng.upload([])
    eng.bv_insert(empty)
    reads = [random_reads(1, n, n, seed=70 + i)[0] for i, n in enumerate((999, 1000, 1021, 1999, 2000, 2021, 2022, 3000 + 21, 500, 24))]
    b = eng.upload(reads)
    eng.bv_insert(b)
    for s in reads:
        omf.bv_insert_read(s)
    assert eng.finalize() == omf.finalize()
    assert np.array_equal(eng.export_bits(), omf.bits())
    tiles, lists, st = eng.query_tiles(empty)
    assert len(tiles) == 0 and st["queries"] == 0
    assert [int(x) for x in np.diff(b.tile0)] == [len(r) // TILE for r in reads]
    for ri in (1, 3, 5, 7):
        nt = len(reads[ri]) // TILE
        eng.insert_tiles(b, ri, 0, nt, 10 + ri)
        omf.insert_read_tiles(reads[ri], 0, nt, 10 + ri)
    eng.insert_tiles(b, 0, 0, 0, 99)  # empty tile range: no-op
    _compare_queries(eng, omf, b, reads)
    # sub-ranges of the batch, including ranges that hold no tile at all
    t, l, s = eng.query_tiles(b, 8, 2)
    assert len(t) == 0
    t, l, s = eng.query_tiles(b, 3, 3)
    exp = [r for i in (3, 4, 5) for r in omf.query_read(reads[i])]
    assert [(int(x["top_id"]), int(x["top_count"])) for x in t] == [(e[0], e[1]) for e in exp]
    with pytest.raises(native.GrpError):
        eng.query_tiles(b, 5, 50)
    with pytest.raises(native.GrpError):
        eng.insert_tiles(b, 1, 0, 5, 1)  # read 1 has one tile


def test_max_geometry_h8_span32(oracle, native):
    """Implementation limits: h = 8 seeds, longest span k + h - 1 = 32."""
    k, h, tile = 25, 8, 300
    preset = "1101101110111011101110111"  # 25 wide, palindromic halves not required by the engine
    half = len(preset) // 2
    seeds = [preset[:half] + "0" * i + preset[half:half * 2] for i in range(h)]
    # spans must be k+i with k = 2*half
    k = 2 * half
    seeds = [preset[:half] + "0" * i + preset[half:2 * half] for i in range(h)]
    m = oracle.load().orc_calc_optimal_size(300_000, 1, 0.1)
    eng = native.Engine(k, h, tile, m, seeds)
    oseeds = oracle.Seeds(seeds)
    omf = oracle.MiBF(m, oseeds, tile, k)
    reads = random_reads(4, 700, 1500, seed=81)
    b = eng.upload(reads)
    for ri, seq in enumerate(reads):
        for t in range(len(seq) // tile):
            assert np.array_equal(eng.tile_hashes(b, ri, t), oseeds.tile_hashes(seq, tile, k, t))
    eng.bv_insert(b)
    for s in reads:
        omf.bv_insert_read(s)
    assert eng.finalize() == omf.finalize()
    assert np.array_equal(eng.export_bits(), omf.bits())
    eng.insert_tiles(b, 1, 0, len(reads[1]) // tile, 7)
    omf.insert_read_tiles(reads[1], 0, len(reads[1]) // tile, 7)
    tiles, lists, st = eng.query_tiles(b)
    ti = 0
    for seq in reads:
        for res in omf.query_read(seq):
            assert (int(tiles[ti]["top_id"]), int(tiles[ti]["top_count"])) == (res[0], res[1])
            ti += 1


def test_insert_read_equals_block_loop(oracle, native):
    """grp_insert_read (two launches for all ID blocks of a read) == the reference's
    sequence of insertMIBF calls, including ranks shared by several blocks (a read
    built from a repeated segment) and the trimmed-read ID rule."""
    eng, oseeds, omf, m = _mk(oracle, native, m=oracle.load().orc_calc_optimal_size(400_000, 1, 0.1))
    unit = random_reads(1, 1700, 1700, seed=91)[0]
    rep = (unit * 8)[:12500]                       # blocks of 3 tiles share k-mers across blocks
    reads = random_reads(4, 4000, 9000, seed=92) + [rep, random_reads(1, 70000, 70000, seed=93)[0]]
    b = eng.upload(reads)
    eng.bv_insert(b)
    for s in reads:
        omf.bv_insert_read(s)
    assert eng.finalize() == omf.finalize()
    next_id = 0
    # whole-read rule (goldrush_path.cpp:982-994), block of 3 tiles
    for ri in (4, 0, 4, 2):
        seq = reads[ri]
        nt = len(seq) // TILE
        next_id += 1
        eng.insert_read(b, ri, 0, nt, 3, next_id, 0)
        for bs in range(0, nt, 3):
            omf.insert_read_tiles(seq, bs, min(bs + 3, nt), next_id + bs // 3)
        next_id += len(seq) // (TILE * 3)
        ids, counts = eng.export_ids()
        assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts()), ri
    assert omf.counts().max() >= 4  # shared ranks really were replayed block by block
    # trimmed rule (:1040-1053): tiles [ts, te], IDs first + (bs - ts + 1) // block
    for ri, ts, te, blk in ((1, 1, len(reads[1]) // TILE - 1, 2), (3, 0, len(reads[3]) // TILE - 2, 1), (4, 2, 11, 4)):
        seq = reads[ri]
        next_id += 1
        eng.insert_read(b, ri, ts, te + 1, blk, next_id, 1)
        bs = ts
        while bs <= te:
            be = min(bs + blk - 1, te)
            omf.insert_read_tiles(seq, bs, be + 1, next_id + (bs - ts + 1) // blk)
            bs += blk
        next_id += (te - ts) // blk
        ids, counts = eng.export_ids()
        assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts()), (ri, ts, te)
    # more than 64 blocks: falls back to one launch per block
    seq = reads[5]
    nt = len(seq) // TILE
    eng.insert_read(b, 5, 0, nt, 1, 5000, 0)
    for bs in range(nt):
        omf.insert_read_tiles(seq, bs, bs + 1, 5000 + bs)
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts())
    _compare_queries(eng, omf, b, reads)


def test_sharded_fill_or_merge(oracle, native):
    """Multi-GPU fill path on one device: two contexts fill disjoint read shards, the
    bit vectors are exchanged through device memory and OR-merged; the result equals
    the single-context fill (and the oracle)."""
    eng, oseeds, omf, m = _mk(oracle, native)
    reads = random_reads(10, 1500, 6000, seed=101)
    seeds = default_seeds(3)
    a = native.Engine(K, 3, TILE, m, seeds)
    b2 = native.Engine(K, 3, TILE, m, seeds)
    ba, bb = a.upload(reads), b2.upload(reads)
    a.bv_insert(ba, 0, 5)
    b2.bv_insert(bb, 5, 5)
    lib = native.load()
    n = a.bv_words()
    assert n == (m + 31) // 32
    buf = lib.grp_synth_alloc(n * 4 + 64)
    b2.bv_export_device(buf)
    a.bv_merge_device(buf)
    lib.grp_synth_free(buf)
    for s in reads:
        omf.bv_insert_read(s)
    assert a.finalize() == omf.finalize()
    assert np.array_equal(a.export_bits(), omf.bits())
    with pytest.raises(native.GrpError):
        a.bv_merge_device(1 << 20)  # after finalize


def test_ntcard_tables_match_oracle(oracle, native):
    """grp_ntcard_* (ntcard.hpp:81-112 on the device) vs the oracle: zero buckets of
    every sample table, for plain reads (iterator rule), reads shorter than the longest
    span, reads given as ACGT runs with explicit stale repeats, two calls + two batches;
    and the deferred filter size (grp_params.m = 0 -> grp_set_filter_size)."""
    from goldrush_amd import host

    k, h, tile = 22, 3, 500
    seeds = default_seeds(h)
    osd = oracle.Seeds(seeds)
    reads = random_reads(40, 1500, 9000, seed=91) + [b"ACGT" * 5 + b"AC", b"ACGT" * 5 + b"ACG", b"ACGT" * 6, b"ACGT" * 5 + b"A", b"ACG"]
    dirty = []
    rng = np.random.default_rng(92)
    for i, r in enumerate(random_reads(6, 800, 5000, seed=93)):
        r = bytearray(r)
        for p in rng.integers(0, len(r), size=2 + i):
            r[p] = ord("N")
        dirty.append(bytes(r))
    eng = native.Engine(k, h, tile, 0, seeds)  # size not known yet
    b = eng.upload(reads)
    with pytest.raises(native.GrpError):
        eng.bv_insert(b)  # no filter yet
    with pytest.raises(native.GrpError):
        eng.ntcard_add(b)  # not begun
    eng.ntcard_begin(7)
    eng.ntcard_add(b, 0, 17)
    eng.ntcard_add(b, 17)
    runs_all, extra_all = [], []
    for seq in dirty:
        runs, extra = host.ntcard_split(seq, k, h)
        runs_all += [seq[o:o + n] for o, n in runs]
        extra_all.append(extra)
    b2 = eng.upload(runs_all)
    eng.ntcard_add(b2, stale_extra=np.concatenate(extra_all).ravel())
    z = eng.ntcard_finish()
    nc = oracle.NtCard(osd, 1000)
    for seq in reads + dirty:
        nc.add_read(seq)
    assert np.array_equal(z, nc.zero_buckets())
    assert int((z < (1 << 27)).sum()) == 2 * h  # every table was hit
    for s in range(h):
        assert host.load().gr_ntcard_f0(int(z[s][0]), int(z[s][1]), 7) == nc.f0(s)
    st = eng.kernel_stats()["ntcard"]
    assert st["launches"] == 3 and st["units"] > 0
    # the filter gets its size now; fill works as in a context created with m
    m = oracle.load().orc_calc_optimal_size(sum(nc.f0(s) for s in range(h)), 1, 0.1)
    nc.close()
    eng.set_filter_size(m)
    with pytest.raises(native.GrpError):
        eng.set_filter_size(m)
    eng.bv_insert(b)
    omf = oracle.MiBF(m, osd, tile, k)
    for seq in reads:
        if len(seq) >= k + h - 1:
            omf.bv_insert_read(seq)
    assert eng.finalize() == omf.finalize()
    assert np.array_equal(eng.export_bits(), omf.bits())
    eng.close()


def test_bv_insert_is_reentrant(native):
    """grp_bv_insert from eight host threads at once (the reference calls insertBV from an
    `omp parallel` region, goldrush_path.cpp:257-305) sets the bits one thread sets."""
    import threading

    seeds = default_seeds(3)
    m = 200_000_000
    dr = native.synth_reads(4000, 5_000_000, mean_len=9000, min_len=7000)
    a = native.Engine(K, 3, TILE, m, seeds)
    ba = a.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    a.bv_insert(ba)
    a.sync()
    ref = a.export_bits()
    a.close()
    b = native.Engine(K, 3, TILE, m, seeds)
    bb = b.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    errs = []

    def work(t):
        try:
            for lo in range(t * 25, 4000, 8 * 25):
                b.bv_insert(bb, lo, 25)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    b.sync()
    assert np.array_equal(b.export_bits(), ref)
    b.close()
    dr.free()


def test_bit_vector_merge_primitives(native):
    """grp_words_or_device / grp_bv_export_device / grp_bv_import_device (the pieces of the
    multi-GPU fill merge, bench.py): two engines fill halves of the reads, the vectors are
    OR-ed in caller memory and imported — the result is the vector of one engine filling all."""
    seeds = default_seeds(3)
    m = 150_000_000
    dr = native.synth_reads(1500, 3_000_000, mean_len=9000, min_len=7000)
    full = native.Engine(K, 3, TILE, m, seeds)
    bf = full.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    full.bv_insert(bf)
    full.sync()
    ref = full.export_bits()
    full.close()
    a = native.Engine(K, 3, TILE, m, seeds)
    b = native.Engine(K, 3, TILE, m, seeds)
    ba, bb = a.wrap_device(dr.d_ptr, dr.word_off, dr.lens), b.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    a.bv_insert(ba, 0, 700)
    b.bv_insert(bb, 700, 800)
    a.sync()
    b.sync()
    n = a.bv_words()
    pad = (n + 3) // 4 * 4
    bufa = native.load().grp_synth_alloc(pad * 4)
    bufb = native.load().grp_synth_alloc(pad * 4)
    a.bv_export_device(bufa)
    b.bv_export_device(bufb)
    a.words_or_device(bufa, bufb, n)
    b.bv_import_device(bufa)
    assert np.array_equal(b.export_bits(), ref)
    assert a.finalize() < b.finalize()  # a alone holds fewer bits than the merged vector
    native.load().grp_synth_free(bufa)
    native.load().grp_synth_free(bufb)
    a.close()
    b.close()
    dr.free()


def test_rccl_merge_entry_points_on_a_communicator_of_one(native):
    """grp_comm_unique_id / grp_comm_init / grp_bv_merge_ranks — the engine's own RCCL calls (dlopen of
    librccl.so, ncclCommInitRank with the by-value id, ncclAllToAll, the OR pass, ncclAllGather on the
    engine's stream) — as far as ONE GPU can run them: a communicator of one rank, whose merge must leave
    the bit vector as it is (a vector whose word count is not a multiple of the slice padding included).
    Two ranks on one device are refused by RCCL and take the staged merge (tests/test_gpu_cli.py)."""
    seeds = default_seeds(3)
    dr = native.synth_reads(300, 2_000_000, mean_len=8000, min_len=6000)
    for m in (150_000_000, 150_000_067):
        eng = native.Engine(K, 3, TILE, m, seeds)
        b = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
        eng.bv_insert(b)
        eng.sync()
        before = eng.export_bits()
        uid = eng.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        eng.comm_init(uid, 1, 0)
        with pytest.raises(native.GrpError):
            eng.comm_init(uid, 1, 0)  # one communicator per context
        eng.bv_merge_ranks()
        eng.bv_merge_ranks()
        assert np.array_equal(eng.export_bits(), before)
        pop = eng.finalize()
        assert pop == int(np.unpackbits(before.view(np.uint8)).sum())
        with pytest.raises(native.GrpError):
            eng.bv_merge_ranks()  # only between the fill and grp_finalize
        eng.close()
    dr.free()
