"""CPU: the product's C++ host logic (libgrpath_host.so) against the oracle
(CPU restatement of the reference) on the same seeded inputs.  Bit-exact."""
import numpy as np
import pytest

from helpers import SEED22


@pytest.fixture(scope="module")
def host(native):
    from goldrush_amd import host as h

    h.load()
    return h


def test_seed_patterns(oracle, host):
    for preset, k, w, h in [(SEED22, 22, 16, 3), (SEED22, 22, 16, 5), ("", 22, 16, 3), ("", 32, 20, 4), ("", 16, 10, 1), ("110011", 6, 4, 2), ("1110111", 7, 6, 3)]:
        assert host.make_seed_pattern(preset, k, w, h) == oracle.make_seed_pattern(preset, k, w, h)
    # palindromic design: seed i = left + i zeros + mirror(left)
    s = host.make_seed_pattern("", 22, 16, 3)
    assert s[0] == s[0][::-1] and s[0].count("1") == 16 and [len(x) for x in s] == [22, 23, 24]


def test_sizing(oracle, host):
    ol, hl = oracle.load(), host.load()
    for w, g, h in [(16, 1_000_000, 3), (16, 100_000_000, 3), (16, 3_000_000_000, 3), (16, 3_000_000_000, 5), (12, 3_100_000_000, 3), (14, 123_456_789, 7)]:
        u = hl.gr_hash_universe(w, g, h)
        assert u == ol.orc_hash_universe(w, g, h)
        for occ in (0.1, 0.05, 0.37):
            assert hl.gr_calc_optimal_size(u, 1, occ) == ol.orc_calc_optimal_size(u, 1, occ)
    # SURVEY.md §8 table (derived with the reference's formulas)
    assert hl.gr_hash_universe(16, 1_000_000, 3) == 3_000_000
    assert hl.gr_calc_optimal_size(3_000_000, 1, 0.1) == 28_473_728
    assert hl.gr_calc_optimal_size(hl.gr_hash_universe(16, 100_000_000, 3), 1, 0.1) == 2_847_366_528
    assert hl.gr_calc_optimal_size(hl.gr_hash_universe(16, 3_000_000_000, 3), 1, 0.1) == 61_146_729_472
    assert hl.gr_calc_optimal_size(hl.gr_hash_universe(16, 3_000_000_000, 5), 1, 0.1) == 101_911_215_744


def test_phred(oracle, host):
    rng = np.random.default_rng(3)
    quals = [b"5" * 100, b"5" * 101, b"I", b"!" * 7, b"5" * 50 + b"#" * 50]
    for _ in range(200):
        n = int(rng.integers(2, 400))
        quals.append((rng.integers(0, 42, size=n) + 33).astype(np.uint8).tobytes())
    ol, hl = oracle.load(), host.load()
    for q in quals:
        assert host.calc_phred_average(q) == oracle.calc_phred_average(q)
        assert hl.gr_sum_phred(q, len(q)) == ol.orc_sum_phred(q, len(q))


def test_pack_2bit(native, host):
    rng = np.random.default_rng(4)
    acgt = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    for n in (0, 1, 15, 16, 17, 31, 32, 33, 1000, 1021):
        s = acgt[rng.integers(0, 8, size=n)].tobytes()
        rc, words = host.pack_2bit(s)
        assert rc == 0
        exp, _, _ = native.pack_reads([s])
        assert np.array_equal(words, exp)
    assert host.pack_2bit(b"ACGTN")[0] == -1
    assert host.pack_2bit(b"ACGT" * 8 + b"-")[0] == -1


def _random_tiles(rng, n):
    """Random per-tile query results shaped like real ones: a few IDs close to
    each other (so the +-1 rules fire), lists of (id,count>2)."""
    base = int(rng.integers(1, 50))
    pool = [base + int(d) for d in rng.integers(0, 4, size=4)] + [int(rng.integers(1, 1000))]
    if rng.random() < 0.1:
        pool.append(0)
    ids, lists = [], []
    for _ in range(n):
        k = int(rng.integers(0, 4))
        chosen = list(dict.fromkeys(int(pool[i]) for i in rng.integers(0, len(pool), size=k)))
        chosen = [c for c in chosen if c != 0]
        lst = sorted(((c, int(rng.choice([3, 4, 9, 10, 11, 12, 40, 400]))) for c in chosen), key=lambda t: (-t[1], t[0]))
        if lst and rng.random() < 0.9:
            top = min((c for c in lst if c[1] == lst[0][1]), key=lambda t: t[0])[0]
        else:
            top = int(pool[int(rng.integers(0, len(pool)))]) if not lst else lst[0][0]
        ids.append(top)
        lists.append(lst)
    return ids, lists


def test_smoothing_decision_differential(oracle, host):
    rng = np.random.default_rng(11)
    n_cases = 6000
    kinds = {}
    for case in range(n_cases):
        n = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 12, 14, 15, 16, 20, 25, 31, 60]))
        ids, lists = _random_tiles(rng, n)
        x = int(rng.choice([10, 10, 10, 3, 11]))
        # oracle
        ol = [np.array(l, dtype=oracle.id_count_dtype) if l else np.zeros(0, dtype=oracle.id_count_dtype) for l in lists]
        o_ids, o_b, o_na = oracle.smooth_tiles(ids, ol, x)
        # product
        tiles, flat = host.tiles_from(ids, lists)
        p_ids, p_b, p_na = host.smooth_tiles(tiles, flat, n, x)
        assert p_na == o_na, (case, ids, lists)
        assert np.array_equal(p_ids, o_ids[:n]), (case, ids, lists)
        assert np.array_equal(p_b, o_b), (case, ids, lists)
        if n:
            assert host.find_longest_stretch(p_b) == oracle.find_longest_stretch(o_b)
            ls, le = oracle.find_longest_stretch(o_b)
            assert host.eval_flanks(ls, le, p_ids) == oracle.eval_flanks(ls, le, o_ids[:n]), (case, ids, lists)
        d = host.decide_read(tiles, flat, n, threshold=x)
        # decision as process_read takes it (goldrush_path.cpp:967-1040)
        nu = n - o_na
        if nu >= 5 and o_na <= 1:
            exp = (2, 0, 0)
        elif o_na == n:
            exp = (3, 0, 0)
        else:
            ls, le = oracle.find_longest_stretch(o_b)
            good, ts, te = oracle.eval_flanks(ls, le, o_ids[:n])
            exp = (4, ts, te) if good else (5, 0, 0)
        assert (d.kind, d.trim_start, d.trim_end) == exp, (case, ids, lists)
        assert (d.num_tiles, d.num_assigned) == (n, o_na)
        kinds[d.kind] = kinds.get(d.kind, 0) + 1
    assert all(kinds.get(k, 0) > 20 for k in (2, 3, 4, 5)), kinds  # every branch exercised


def test_flank_edge_cases(oracle, host):
    rng = np.random.default_rng(12)
    for _ in range(4000):
        n = int(rng.integers(1, 40))
        ids = rng.integers(1, 6, size=n).astype(np.uint32)
        a = int(rng.integers(0, n))
        b = int(rng.integers(a, n))
        assert host.eval_flanks(a, b, ids) == oracle.eval_flanks(a, b, ids), (n, a, b, ids.tolist())
        bools = (rng.random(n) < rng.random()).astype(np.uint8)
        assert host.find_longest_stretch(bools) == oracle.find_longest_stretch(bools), bools.tolist()


def test_ntcard_host_arithmetic(oracle, host):
    """--ntcard host side (gr_ntcard_*) vs the oracle's restatement of ntcard.hpp: sBits,
    F0 from zero buckets, and the decomposition of a record with non-ACGT characters
    into ACGT runs + stale repeats (what grp_ntcard_add is fed) reproduces the oracle's
    sample tables exactly."""
    from helpers import default_seeds, random_reads

    lib = host.load()
    assert lib.gr_ntcard_sbits(0) == 7 and lib.gr_ntcard_sbits(49_999_999_999) == 7 and lib.gr_ntcard_sbits(50_000_000_000) == 11
    k, h = 22, 3
    seeds = default_seeds(h)
    sd = oracle.Seeds(seeds)
    rng = np.random.default_rng(5)
    reads = random_reads(8, 500, 4000, seed=6)
    out = []
    for i, r in enumerate(reads):
        r = bytearray(r)
        for p in rng.integers(0, len(r), size=i):  # read 0 stays clean
            r[p] = ord("N")
        if i == 3:
            r[10:60] = r[10:60].lower()
            r[-23] = ord("n")  # last run holds only the two shorter seeds
        if i == 4:
            r[-1] = ord("X")
        out.append(bytes(r))
    out += [b"ACGT" * 5 + b"AC", b"ACGT" * 5 + b"ACG", b"N" * 30, b"ACGTACGTAC" * 3 + b"N" + b"ACGTACGTAC" * 3]
    sbits, rbits = 7, 27
    nc = oracle.NtCard(sd, 123)
    exp = {}
    for seq in out:
        nc.add_read(seq.upper())
        runs, extra = host.ntcard_split(seq, k, h)
        for (off, ln), ex in zip(runs, extra):
            run = seq[off:off + ln].upper()
            assert set(run) <= set(b"ACGT") and ln >= k
            assert (off == 0 or seq[off - 1:off].upper() not in (b"A", b"C", b"G", b"T")) and seq[off + ln:off + ln + 1].upper() not in (b"A", b"C", b"G", b"T")
            for s, pat in enumerate(seeds):
                K = len(pat)
                one = oracle.Seeds([pat])
                for p in range(ln - K + 1):
                    hv = int(one.multi_hash(run[p:p + K])[0])
                    times = 1 + (int(ex[s]) if p + K == ln else 0)
                    ind = 2
                    if hv >> (63 - sbits) == 1:
                        ind = 0
                    if hv >> (64 - sbits) == (1 << (sbits - 1)) - 1:
                        ind = 1
                    if ind < 2:
                        key = (s, ind, hv & ((1 << rbits) - 1))
                        exp[key] = exp.get(key, 0) + times
    cnt = nc.counters()
    got = {(int(a), int(b), int(c)): int(cnt[a, b, c]) for a, b, c in np.argwhere(cnt)}
    assert got == exp and len(exp) > 100
    z = nc.zero_buckets()
    for s in range(h):
        assert lib.gr_ntcard_f0(int(z[s][0]), int(z[s][1]), sbits) == nc.f0(s)
    assert lib.gr_ntcard_f0(1 << 27, 1 << 27, 7) == 0
    nc.close()


@pytest.mark.parametrize("gz", [False, True])
def test_input_reader_delivers_the_file_whatever_the_request_size(host, tmp_path, gz):
    """The CLI's chunk reader: plain files by pread (requests >= 32 MiB by several threads),
    gzip data through zlib — the same bytes in the same order for every request size, a short
    last slice and a file that ends inside a request included."""
    import ctypes as C
    import gzip

    hl = host.load()
    rng = np.random.default_rng(5)
    n = (70 << 20) + 12345 if not gz else (3 << 20) + 77
    data = rng.integers(0, 256, size=n, dtype=np.uint8)
    p = tmp_path / ("in.gz" if gz else "in.txt")
    if gz:
        with gzip.open(p, "wb", compresslevel=1) as f:
            f.write(data.tobytes())
    else:
        p.write_bytes(data.tobytes())
    for req in (1 << 30, (64 << 20) + 1, 33 << 20, 1 << 20, 4097):
        if gz and req < (1 << 20):
            continue
        dst = np.zeros(n + 4096, dtype=np.uint8)
        got = hl.gr_input_read(str(p).encode(), req, dst.ctypes.data_as(C.c_void_p), dst.size)
        assert got == n, (req, got)
        assert np.array_equal(dst[:n], data), req
    assert hl.gr_input_read(str(tmp_path / "missing").encode(), 1 << 20, None, 0) == 2**64 - 1
    (tmp_path / "empty").write_bytes(b"")
    assert hl.gr_input_read(str(tmp_path / "empty").encode(), 1 << 20, dst.ctypes.data_as(C.c_void_p), dst.size) == 0
