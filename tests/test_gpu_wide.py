"""GPU parity beyond 32 bits: filters with m >= 2^32 against the CPU oracle (the 64-bit legs
of x % m, the bucket index, the superbucket rank table; buckets of W ~ 60 bits as C2 has
them), and launches larger than one HIP grid (2^32 work-items)."""
import ctypes as C

import numpy as np
import pytest

from helpers import SEED22, canon_list, default_seeds

pytestmark = pytest.mark.gpu

K, TILE, BLOCK = 22, 1000, 10


def _oracle_bits_view(omf):
    """the oracle's plain bit vector without a copy"""
    n = omf.lib.orcpy_mibf_n_words(omf._h)
    return np.ctypeslib.as_array(C.cast(omf.lib.orcpy_mibf_bv(omf._h), C.POINTER(C.c_uint64)), shape=(n,))


def _check_rank_samples(eng, omf, m, seed, n=6000):
    rng = np.random.default_rng(seed)
    pos = rng.integers(0, m, size=n, dtype=np.uint64)
    pos[:6] = [0, 1, (1 << 32) - 1, 1 << 32, m - 1, m - 2]
    bit, rank = eng.rank(pos)
    for i in range(n):
        p = int(pos[i])
        assert bit[i] == omf.bit(p), p
        assert rank[i] == omf.rank(p), p


def _compare_queries(eng, omf, batch, reads, first, count):
    tiles, lists, stats = eng.query_tiles(batch, first, count)
    ti = 0
    q = h = ms = 0
    for seq in reads[first:first + count]:
        for top_id, top_count, lst, ctr in omf.query_read(seq):
            t = tiles[ti]
            assert (int(t["top_id"]), int(t["top_count"])) == (top_id, top_count), ti
            got = [(int(a), int(c)) for a, c in lists[t["list_off"]: t["list_off"] + t["list_n"]]]
            assert got == canon_list(lst), ti
            q += ctr[0]; h += ctr[1]; ms += ctr[2]
            ti += 1
    assert ti == len(tiles)
    assert (stats["queries"], stats["hits"], stats["misses"]) == (q, h, ms)


def _compare_pshard(eng, omf, batch, reads, first, count, n_owners):
    """the position-sharded form of the query (grp_pshard_query: partition into owners' bins, gather in owner order, vote
    on what comes back; csrc/grp_pshard.inc) against the oracle, tile by tile"""
    tiles, lists, times = eng.pshard_query(batch, first, count, n_owners)
    ti = 0
    for seq in reads[first:first + count]:
        for top_id, top_count, lst, ctr in omf.query_read(seq):
            t = tiles[ti]
            assert (int(t["top_id"]), int(t["top_count"])) == (top_id, top_count), (ti, n_owners)
            got = sorted(((int(a), int(c)) for a, c in lists[t["list_off"]: t["list_off"] + t["list_n"]]), key=lambda e: (-e[1], e[0]))
            assert got == canon_list(lst), (ti, n_owners)
            assert (int(t["hits"]), int(t["misses"])) == (ctr[1], ctr[2]), (ti, n_owners)
            ti += 1
    assert ti == len(tiles) and all(v >= 0 for v in times.values())


def _insert_whole(eng, omf, batch, reads, ri, next_id):
    seq = reads[ri]
    nt = len(seq) // TILE
    next_id += 1
    eng.insert_read(batch, ri, 0, nt, BLOCK, next_id, 0)
    for bs in range(0, nt, BLOCK):
        omf.insert_read_tiles(seq, bs, min(bs + BLOCK, nt), next_id + bs // BLOCK)
    return next_id + len(seq) // (TILE * BLOCK)


def test_filter_of_2_pow_33_bits_matches_oracle(oracle, native):
    """m = 2^33 + 64, 17.5 k reads of 25 kb: occupancy ~0.1, W = 50..64 filter bits per bucket
    (the C2 regime), pop ~ 9e8.  Fill, rank build, 50 whole-read inserts, queries: bits, pop,
    sampled ranks, every ID / count and the tile summaries equal the oracle's."""
    m = (1 << 33) + 64
    h = 3
    seeds = default_seeds(h, SEED22)
    G = 100_000_000
    n_reads = 17_500
    dr = native.synth_reads(n_reads, G, mean_len=25000, min_len=20000, seed=7)
    eng = native.Engine(K, h, TILE, m, seeds)
    batch = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    eng.bv_insert(batch)
    oseeds = oracle.Seeds(seeds)
    omf = oracle.MiBF(m, oseeds, TILE, K)
    reads = dr.download(0, n_reads)
    omf.bv_insert_reads(reads)  # ~15 CPU-minutes of oracle hashing, spread over the host cores
    assert np.array_equal(eng.export_bits(), _oracle_bits_view(omf))  # phase-1 layout
    pop = eng.finalize()
    assert pop == omf.finalize()
    assert 0.09 < pop / m < 0.12  # W = 6 / occupancy: 50 .. 64 bits per bucket
    bits = eng.export_bits()  # rebuilt from the 64-byte buckets
    assert np.array_equal(bits, _oracle_bits_view(omf))
    del bits
    _check_rank_samples(eng, omf, m, 11)
    _compare_queries(eng, omf, batch, reads, 0, 4)  # empty ID array: all misses
    next_id = 0
    rng = np.random.default_rng(3)
    ins = [int(x) for x in rng.choice(n_reads, size=50, replace=False)]
    for ri in ins:
        next_id = _insert_whole(eng, omf, batch, reads, ri, next_id)
    # a second pass over some of them: counts > 1, the reservoir rule decides
    for ri in ins[:10]:
        next_id = _insert_whole(eng, omf, batch, reads, ri, next_id)
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts())
    assert int((ids != 0).sum()) > 2_500_000
    del ids, counts
    for ri in ins[:6] + [0, 1]:
        _compare_queries(eng, omf, batch, reads, ri, 1)
    # the order-exact classifier over a stretch of the stream against the oracle's decisions
    dec = eng.classify_reads(batch, ins[0], 1)
    res = omf.query_read(reads[ins[0]])
    o_ids, o_b, na = oracle.smooth_tiles([r[0] for r in res], [r[2] for r in res], 10)
    assert int(dec[0]["num_assigned"]) == na and int(dec[0]["num_tiles"]) == len(res)
    eng.close()
    dr.free()


def test_filter_of_c2_size_matches_oracle(oracle, native):
    """BASELINE C2's filter size (m = 61 146 729 472, 7.6 GB of bits, ~1e9 buckets in 228
    superbuckets) with a few reads: positions, ranks, IDs and tile summaries equal the
    oracle's — the 64-bit legs of `hash % m`, the bucket division and the rank table."""
    from goldrush_amd import host

    hl = host.load()
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, 3_000_000_000, 3), 1, 0.1)
    assert m == 61146729472
    h = 3
    seeds = default_seeds(h, SEED22)
    n_reads = 160
    dr = native.synth_reads(n_reads, 3_000_000, mean_len=25000, min_len=20000, seed=9)
    eng = native.Engine(K, h, TILE, m, seeds)
    batch = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    eng.bv_insert(batch)
    oseeds = oracle.Seeds(seeds)
    omf = oracle.MiBF(m, oseeds, TILE, K)
    reads = dr.download(0, n_reads)
    omf.bv_insert_reads(reads)
    pop = eng.finalize()
    assert pop == omf.finalize()
    bits = eng.export_bits()
    obits = _oracle_bits_view(omf)
    assert bits.size == obits.size
    nz = np.flatnonzero(bits)
    assert np.array_equal(nz, np.flatnonzero(obits)) and np.array_equal(bits[nz], obits[nz])
    assert int(nz[-1]) > (1 << 29)  # words beyond bit 2^35 are in use
    del bits, nz
    _check_rank_samples(eng, omf, m, 12, n=3000)
    # ranks at set bits (random positions almost never hit one at this occupancy)
    rng = np.random.default_rng(4)
    ob = np.flatnonzero(obits)
    words = rng.choice(ob, size=2000)
    pos = np.array([int(w) * 64 + int(obits[w]).bit_length() - 1 for w in words], dtype=np.uint64)
    bit, rank = eng.rank(pos)
    for i in range(pos.size):
        assert bit[i] == 1 and rank[i] == omf.rank(int(pos[i]))
    next_id = 0
    for ri in (5, 17, 18, 90, 5):
        next_id = _insert_whole(eng, omf, batch, reads, ri, next_id)
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts())
    _compare_queries(eng, omf, batch, reads, 0, 24)
    _compare_queries(eng, omf, batch, reads, 88, 6)
    # developer builds (make DEV=1, include/grpath_dev.h) also carry the priced-and-rejected position-sharded form of the
    # query: the same reads through it (8 virtual owners as on a node, and 3: a number that does not divide anything)
    if native.load().grp_dev_hooks():
        _compare_pshard(eng, omf, batch, reads, 0, 24, 8)
        _compare_pshard(eng, omf, batch, reads, 88, 6, 3)
    eng.close()
    dr.free()


def test_filter_of_c4_size_matches_oracle(oracle, native):
    """BASELINE C4's filter at its own size: m = 101 911 215 744 (> 2^36, 12.7 GB of
    bits, 1.6e9 buckets in 380 superbuckets), h = 5 designed-family seeds: bits, pop, ranks at set
    bits beyond 2^36, five inserts, every ID and count, tile summaries — against the oracle — and
    one silver-path rollover (grp_reset_ids) on that table."""
    from goldrush_amd import host

    hl = host.load()
    h = 5
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, 3_000_000_000, h), 1, 0.1)
    assert m == 101911215744 and m > (1 << 36)
    seeds = default_seeds(h, SEED22)
    n_reads = 120
    dr = native.synth_reads(n_reads, 3_000_000, mean_len=25000, min_len=20000, seed=11)
    eng = native.Engine(K, h, TILE, m, seeds)
    batch = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    eng.bv_insert(batch)
    oseeds = oracle.Seeds(seeds)
    omf = oracle.MiBF(m, oseeds, TILE, K)
    reads = dr.download(0, n_reads)
    omf.bv_insert_reads(reads)
    pop = eng.finalize()
    assert pop == omf.finalize()
    bits = eng.export_bits()
    obits = _oracle_bits_view(omf)
    assert bits.size == obits.size
    nz = np.flatnonzero(bits)
    assert np.array_equal(nz, np.flatnonzero(obits)) and np.array_equal(bits[nz], obits[nz])
    assert int(nz[-1]) > (1 << 30)  # words beyond bit 2^36 are in use
    del bits, nz
    _check_rank_samples(eng, omf, m, 13, n=3000)
    rng = np.random.default_rng(5)
    ob = np.flatnonzero(obits)
    high = ob[ob > (1 << 30)]
    assert high.size > 1000
    words = np.concatenate([rng.choice(ob, size=1000), rng.choice(high, size=1000)])
    pos = np.array([int(w) * 64 + int(obits[w]).bit_length() - 1 for w in words], dtype=np.uint64)
    bit, rank = eng.rank(pos)
    for i in range(pos.size):
        assert bit[i] == 1 and rank[i] == omf.rank(int(pos[i]))
    assert int(pos.max()) > (1 << 36)
    next_id = 0
    for ri in (3, 40, 41, 77, 3):
        next_id = _insert_whole(eng, omf, batch, reads, ri, next_id)
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts())
    del ids, counts
    _compare_queries(eng, omf, batch, reads, 0, 8)
    _compare_queries(eng, omf, batch, reads, 76, 4)
    # silver-path rollover: both arrays zeroed (goldrush_path.cpp:180-181), bits and ranks untouched
    eng.reset_ids()
    omf.reset_ids()
    ids, counts = eng.export_ids()
    assert not ids.any() and not counts.any()
    del ids, counts
    next_id = _insert_whole(eng, omf, batch, reads, 40, 0)
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts())
    _compare_queries(eng, omf, batch, reads, 39, 3)
    eng.close()
    dr.free()


def test_fill_larger_than_one_grid_equals_fill_in_slices(native):
    """1.4 M reads = 18 M fill workgroups = 4.7e9 work-items, more than one HIP grid holds
    (2^32): one grp_bv_insert call must set the same bits as many small calls."""
    from goldrush_amd import host

    hl = host.load()
    h = 3
    seeds = default_seeds(h, SEED22)
    G = 100_000_000
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, G, h), 1, 0.1)
    n_reads = 1_400_000
    dr = native.synth_reads(n_reads, G)
    a = native.Engine(K, h, TILE, m, seeds)
    ba = a.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    assert int(ba.tile0[-1]) > 0
    a.bv_insert(ba)
    a.sync()
    bits_a = a.export_bits()
    pop_a = a.finalize()
    a.close()
    b = native.Engine(K, h, TILE, m, seeds)
    bb = b.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    step = 50_000
    for lo in range(0, n_reads, step):
        b.bv_insert(bb, lo, min(step, n_reads - lo))
    b.sync()
    bits_b = b.export_bits()
    pop_b = b.finalize()
    b.close()
    dr.free()
    assert pop_a == pop_b
    assert np.array_equal(bits_a, bits_b)
