"""The oracle AND the product host (and, under -m gpu, the device's decision kernel) against the
reference's OWN lines for the libstdc++-only part of the hot path's host logic:
find_longest_stretch, eval_flanks, sort_by_sec, the tail of calc_num_assigned_tiles (threshold
test + smoothing passes P1..P10 + final count), the vote statements inside its per-tile loop
(tabulation of a frame's unique IDs, selection of the tile's ID and list),
MIBloomFilter::calcOptimalSize and the
hash-universe arithmetic of main — cut out of /root/reference at build time and compiled
unchanged (oracle/Makefile `ref`, oracle/extract_ref_funcs.py, oracle/ref_funcs_shim.cpp).

tests/golden/reference_funcs.json holds the reference's outputs on seeded tile states
(tests/golden/make_reference_fixtures.py); where /root/reference exists the library is rebuilt
and thousands of fresh random states are compared live."""
import json
import os
import subprocess

import numpy as np
import pytest

import ref_funcs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "reference_funcs.json")
HAVE_REFERENCE = os.path.isdir("/root/reference/goldrush_path")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(GOLD))


@pytest.fixture(scope="module")
def host(native):
    from goldrush_amd import host as h

    h.load()
    return h


def _lists(c):
    return [[tuple(e) for e in l] for l in c["lists"]]


def _decision(n, na, stretch_flanks, u=5, a=1):
    """process_read's rule (goldrush_path.cpp:960-1040) on the reference's own outputs"""
    if n - na >= u and na <= a:
        return (2, 0, 0)
    if na == n:
        return (3, 0, 0)
    good, ts, te = stretch_flanks()
    return (4, ts, te) if good else (5, 0, 0)


def _check_case(oracle, host, ids, lists, x, out_ids, out_bools, na, stretch, flanks):
    n = len(ids)
    ol = [np.array(l, dtype=oracle.id_count_dtype) if l else np.zeros(0, dtype=oracle.id_count_dtype) for l in lists]
    o_ids, o_b, o_na = oracle.smooth_tiles(ids, ol, x)
    assert o_na == na and list(o_ids[:n]) == list(out_ids) and list(o_b[:n]) == list(out_bools), ("oracle", ids, lists)
    tiles, flat = host.tiles_from(ids, lists)
    p_ids, p_b, p_na = host.smooth_tiles(tiles, flat, n, x)
    assert p_na == na and list(p_ids) == list(out_ids) and list(p_b) == list(out_bools), ("host", ids, lists)
    b = np.array(out_bools, dtype=np.uint8)
    assert oracle.find_longest_stretch(b) == tuple(stretch) and host.find_longest_stretch(b) == tuple(stretch), (out_bools,)
    want = (bool(flanks[0]), flanks[1], flanks[2])
    got_o = oracle.eval_flanks(stretch[0], stretch[1], np.array(out_ids, dtype=np.uint32))
    got_h = host.eval_flanks(stretch[0], stretch[1], np.array(out_ids, dtype=np.uint32))
    assert (got_o[0], got_h[0]) == (want[0], want[0]), (out_ids, stretch)
    if want[0]:  # the trim range only means something for a good flank
        assert got_o == want and got_h == want, (out_ids, stretch)
    d = host.decide_read(tiles, flat, n, threshold=x)
    exp = _decision(n, na, lambda: want)
    assert (d.kind, d.trim_start, d.trim_end) == exp and (d.num_tiles, d.num_assigned) == (n, na), (ids, lists)


def test_smoothing_stretch_flanks_match_the_reference(gold, oracle, host):
    assert len(gold["tiles"]) >= 600 and len(gold["tiles_long"]) >= 44
    kinds = set()
    for c in gold["tiles"] + gold["tiles_long"]:
        _check_case(oracle, host, c["ids"], _lists(c), c["x"], c["out_ids"], c["out_bools"], c["assigned"], c["stretch"], c["flanks"])
        kinds.add(_decision(len(c["ids"]), c["assigned"], lambda: (bool(c["flanks"][0]), c["flanks"][1], c["flanks"][2]))[0])
    assert kinds == {2, 3, 4, 5}


def test_filter_sizing_matches_the_reference(gold, oracle, host):
    ol, hl = oracle.load(), host.load()
    seen = 0
    for c in gold["sizes"]:
        if "w" in c:
            assert hl.gr_hash_universe(c["w"], c["g"], c["h"]) == c["universe"] and ol.orc_hash_universe(c["w"], c["g"], c["h"]) == c["universe"], c
            assert hl.gr_calc_optimal_size(c["universe"], 1, c["occupancy"]) == c["m"] and ol.orc_calc_optimal_size(c["universe"], 1, c["occupancy"]) == c["m"], c
        else:
            assert hl.gr_calc_optimal_size(c["entries"], c["hash_num"], c["occupancy"]) == c["m"] and ol.orc_calc_optimal_size(c["entries"], c["hash_num"], c["occupancy"]) == c["m"], c
        seen += 1
    # the SURVEY 8 size table, now from the reference's own arithmetic
    by = {(c["w"], c["g"], c["h"], c["occupancy"]): c["m"] for c in gold["sizes"] if "w" in c}
    assert by[(16, 1_000_000, 3, 0.1)] == 28_473_728 and by[(16, 100_000_000, 3, 0.1)] == 2_847_366_528
    assert by[(16, 3_000_000_000, 3, 0.1)] == 61_146_729_472 and by[(16, 3_000_000_000, 5, 0.1)] == 101_911_215_744
    assert seen >= 60


def test_tile_vote_matches_the_reference(gold, oracle):
    """The per-tile vote (goldrush_path.cpp:597-622: a frame's unique IDs tabulated into the tile's
    std::map, the highest count with ties to the smallest ID, the list of IDs seen more than twice) —
    the oracle's restatement, which the device's k_query is held against in the -m gpu parity tests,
    on the frames' IDs of the fixture, against what the reference's own statements return."""
    assert len(gold["votes"]) >= 150
    ties = big = 0
    for c in gold["votes"]:
        got = ref_funcs.canon_vote(oracle.vote_tile(c["frames"]))
        assert got == (c["id"], c["count"], [tuple(e) for e in c["list"]]), c["frames"]
        counts = {}
        for f in c["frames"]:
            for i in set(f):
                counts[i] = counts.get(i, 0) + 1
        ties += sum(1 for v in counts.values() if v == c["count"]) > 1
        big += len(c["list"]) >= 2
    assert ties >= 10 and big >= 30  # the fixture does exercise ties for the maximum and multi-entry lists


@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference is only present in the build container")
def test_fixture_is_what_the_reference_computes_now(gold, oracle, host):
    """Rebuild the library from the reference's lines, re-derive the committed fixture, and compare
    4000 fresh random tile states + the run patterns live (oracle and product host)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, stdout=subprocess.DEVNULL)
    rf = ref_funcs.RefFuncs()
    for c in gold["tiles"][::7]:
        o_ids, o_b, na = rf.smooth_tiles(c["ids"], _lists(c), c["x"])
        assert list(o_ids) == c["out_ids"] and list(o_b) == c["out_bools"] and na == c["assigned"]
        assert list(rf.find_longest_stretch(o_b)) == c["stretch"]
    for c in gold["sizes"]:
        if "w" in c:
            assert rf.hash_universe(c["w"], c["g"], c["h"]) == c["universe"]
            assert rf.calc_optimal_size(c["universe"], 1, c["occupancy"]) == c["m"]
    for c in gold["votes"][::5]:
        assert ref_funcs.canon_vote(rf.vote_tile(c["frames"])) == (c["id"], c["count"], [tuple(e) for e in c["list"]])
    rng = np.random.default_rng(98)
    for i in range(3000):  # the vote, live: oracle against the reference's statements
        frames = ref_funcs.random_frames(rng, max_frames=(600 if i % 50 == 0 else 80))
        raw = rf.vote_tile(frames)
        assert all(raw[2][j][1] >= raw[2][j + 1][1] for j in range(len(raw[2]) - 1))
        assert ref_funcs.canon_vote(oracle.vote_tile(frames)) == ref_funcs.canon_vote(raw), frames
    rng = np.random.default_rng(99)

    def live(ids, lists, x):
        o_ids, o_b, na = rf.smooth_tiles(ids, lists, x)
        st = rf.find_longest_stretch(o_b)
        fl = rf.eval_flanks(st[0], st[1], o_ids)
        _check_case(oracle, host, ids, lists, x, [int(v) for v in o_ids], [int(v) for v in o_b], na, list(st), [int(fl[0]), fl[1], fl[2]])

    for i in range(4000):
        n = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 14, 15, 16, 20, 25, 31, 60, 90, 128, 129, 260] + ([700] if i % 40 == 0 else [])))
        ids, lists = ref_funcs.random_tiles(rng, n, wrap=(i % 11 == 0))
        live(ids, lists, int(rng.choice([10, 10, 10, 3, 11, 0, 400])))
    for n in (3, 4, 5, 7, 14, 15, 16, 17, 30):
        for pat in ref_funcs.run_patterns(n):
            for step in (0, 1, 2):
                ids, lists = ref_funcs.tiles_from_pattern(pat, step=step)
                live(ids, lists, 10)


@pytest.mark.gpu
def test_device_decision_kernel_matches_the_reference(gold, native):
    """k_decide (LaneState: tile i in lane i, the passes as wave-uniform scalar code; the LDS / global
    form for reads of more than 64 tiles) on the fixture's tile states: per-tile IDs and flags after
    the passes, assigned count, decision and trim range — against the reference's own outputs."""
    from goldrush_amd import host
    from helpers import default_seeds

    eng = native.Engine(22, 3, 1000, 64 * 1024, default_seeds(3))
    all_cases = gold["tiles"] + gold["tiles_long"]  # the long ones: 65 .. 256 tiles — two / four tiles per lane; .. 4096 — LDS state; 4100 — global arrays
    for x in sorted({c["x"] for c in all_cases}):
        cases = [c for c in all_cases if c["x"] == x]
        # a 70-tile and a 300-tile read made of fixture states glued together would change the result; the long
        # forms are exercised by tests/test_gpu_classifier.py — here every fixture state is one read
        tile0 = np.zeros(len(cases) + 1, dtype=np.uint64)
        all_tiles, all_lists = [], []
        for i, c in enumerate(cases):
            tiles, flat = host.tiles_from(c["ids"], _lists(c))
            tiles = tiles[: len(c["ids"])].copy()
            tiles["list_off"] += len(all_lists)
            all_tiles.extend(tiles.tolist())
            all_lists.extend(flat[: sum(len(l) for l in c["lists"])].tolist())
            tile0[i + 1] = tile0[i] + len(c["ids"])
        tiles = np.array(all_tiles, dtype=native.tile_summary_dtype)
        lists = np.array(all_lists if all_lists else [(0, 0)], dtype=native.id_count_dtype)
        dec, ids, asg = eng.debug_decide(tile0, tiles, lists, threshold=x)
        for i, c in enumerate(cases):
            a, b = int(tile0[i]), int(tile0[i + 1])
            assert list(ids[a:b]) == c["out_ids"] and list(asg[a:b]) == c["out_bools"], (i, c["ids"], c["lists"])
            want = _decision(b - a, c["assigned"], lambda: (bool(c["flanks"][0]), c["flanks"][1], c["flanks"][2]))
            got = (int(dec[i]["kind"]), int(dec[i]["trim_start"]) if dec[i]["kind"] == 4 else 0, int(dec[i]["trim_end"]) if dec[i]["kind"] == 4 else 0)
            assert got == want and int(dec[i]["num_assigned"]) == c["assigned"] and int(dec[i]["num_tiles"]) == b - a, (i, c["ids"], c["lists"])
    eng.close()


@pytest.mark.gpu
def test_device_vote_matches_the_reference_statements(native, oracle):
    """k_query's per-tile vote against the reference's OWN vote statements (goldrush_path.cpp:597-622,
    oracle/_ref/libref_funcs.so — the compiled library travels to the GPU box; without it the oracle's
    restatement, which the CPU suite pins to the same statements, stands in): the filter is filled from
    random reads, every set bit gets an ID from a small pool (ties, counts at the `> 2` border, saturated
    values, empty slots), the frames' IDs are read back rank by rank (device hashes and ranks) and voted
    on by the reference; the kernel's tile summary — ID, count, the count > 2 list — must be that vote."""
    from helpers import canon_list, default_seeds, random_reads

    rf = ref_funcs.RefFuncs() if os.path.exists(ref_funcs.LIB) else None
    k, tile = 22, 1000
    for h, pool_size in ((3, 3), (3, 40), (5, 12), (1, 5)):
        m = oracle.load().orc_calc_optimal_size(250_000, 1, 0.1)
        eng = native.Engine(k, h, tile, m, default_seeds(h))
        reads = random_reads(5, 2300, 4800, seed=60 + h + pool_size)
        b = eng.upload(reads)
        eng.bv_insert(b)
        pop = eng.finalize()
        rng = np.random.default_rng(h * 100 + pool_size)
        ids = rng.integers(0, pool_size + 1, size=pop, dtype=np.uint32)  # 0 = empty
        sat = rng.random(pop) < 0.08
        ids[sat] |= np.uint32(0x80000000)  # includes the bare saturation bit
        eng.import_ids(0, ids=ids, counts=np.zeros(pop, dtype=np.uint32))
        tiles, lists, _ = eng.query_tiles(b)
        ti = 0
        for ri, seq in enumerate(reads):
            for t in range(len(seq) // tile):
                hv = eng.tile_hashes(b, ri, t).reshape(-1, h)  # [frame][seed]
                bit, rank = eng.rank((hv % np.uint64(m)).ravel())
                bit, rank = bit.reshape(-1, h), rank.reshape(-1, h)
                frames = []
                for fr in range(hv.shape[0]):
                    if not bit[fr].all():  # atRank: one clear bit and the frame counts nothing
                        frames.append([])
                        continue
                    d = ids[rank[fr]]
                    # goldrush_path.cpp:573-594: only values ABOVE the mask are stripped — the bare saturation bit
                    # (0x80000000, "saturated, no ID") is neither above it nor zero and is counted as an ID
                    d = np.where(d > 0x80000000, d & 0x7FFFFFFF, d)
                    frames.append([int(x) for x in d if x != 0])
                want = ref_funcs.canon_vote(rf.vote_tile(frames) if rf else oracle.vote_tile(frames))
                got = tiles[ti]
                glist = [(int(a), int(c)) for a, c in lists[got["list_off"]: got["list_off"] + got["list_n"]]]
                assert (int(got["top_id"]), int(got["top_count"]), glist) == (want[0], want[1], canon_list(want[2])), (h, pool_size, ri, t)
                ti += 1
        assert ti == len(tiles)
        eng.close()


def _strand_states(rng, sizes):
    """Tile states of reads that overlap earlier reads: long runs of assigned tiles whose IDs rise (same strand) or
    FALL (other strand) along the read, in steps of 0 / 1 (ID blocks of `block` tiles), broken by unassigned
    tiles, foreign IDs and repeats of an earlier ID (P7's work) — the worst case of P7's sort is the falling run."""
    out = []
    for n in sizes:
        for direction in (1, -1):
            ids, lists = [], []
            cur = int(rng.integers(5000, 9000))
            for i in range(n):
                r = rng.random()
                if r < 0.08:       # an unassigned tile with a weak vote
                    tid = int(rng.integers(1, 20000))
                    ids.append(tid)
                    lists.append([] if rng.random() < 0.5 else [(tid, int(rng.integers(3, 9)))])
                    continue
                if r < 0.12:       # a foreign ID with a strong vote
                    tid = int(rng.integers(1, 20000))
                    ids.append(tid)
                    lists.append([(tid, int(rng.integers(11, 60)))])
                    continue
                if r < 0.16 and i > 10:  # an earlier ID again (two non-adjacent occurrences)
                    tid = ids[int(rng.integers(0, i - 5))]
                    ids.append(tid)
                    lists.append([(tid, int(rng.integers(11, 60)))])
                    continue
                if rng.random() < 0.45:
                    cur += direction
                ids.append(cur)
                l = [(cur, int(rng.integers(9, 200)))]
                if rng.random() < 0.3:
                    l.append((cur - direction, int(rng.integers(3, l[0][1] + 1))))
                l.sort(key=lambda t: (-t[1], t[0]))
                lists.append(l)
            out.append((ids, lists, 10))
    return out


def test_strand_states_host_core_against_the_reference_live(oracle):
    """Rising / falling ID runs (reads on either strand over earlier reads), 20 .. 700 tiles: the shared decision core
    (host build; the device builds the same template around its own states) against the reference's lines."""
    from goldrush_amd import host

    if not os.path.exists(ref_funcs.LIB):
        pytest.skip("oracle/_ref not built here")
    rf = ref_funcs.RefFuncs()
    rng = np.random.default_rng(414)
    for ids, lists, x in _strand_states(rng, [20, 40, 64, 65, 100, 128, 200, 256, 300, 700]):
        o_ids, o_b, na = rf.smooth_tiles(ids, lists, x)
        st = rf.find_longest_stretch(o_b)
        fl = rf.eval_flanks(st[0], st[1], o_ids)
        _check_case(oracle, host, ids, lists, x, [int(v) for v in o_ids], [int(v) for v in o_b], na, list(st), [int(fl[0]), fl[1], fl[2]])


@pytest.mark.gpu
def test_device_decisions_of_strand_states(native):
    """k_decide in all its forms — tile per lane (<= 64 tiles), two / four tiles per lane (.. 256), LDS state walked
    by the wave (.. 4096), global arrays — on rising / falling ID runs: per-tile IDs and flags, assigned count and
    decision against the host build of the same core, which the CPU suite pins to the reference's own lines
    (test_strand_states_host_core_against_the_reference_live, the fixtures).  P7's wave-parallel sort is the point."""
    from goldrush_amd import host
    from helpers import default_seeds

    eng = native.Engine(22, 3, 1000, 64 * 1024, default_seeds(3))
    rng = np.random.default_rng(414)
    cases = _strand_states(rng, [20, 40, 64, 65, 100, 128, 200, 256, 300, 700])
    cases += _strand_states(np.random.default_rng(415), [3, 5, 17, 63, 129, 255, 257, 1100, 4096, 4200])
    tile0 = np.zeros(len(cases) + 1, dtype=np.uint64)
    all_tiles, all_lists, per = [], [], []
    for i, (ids, lists, x) in enumerate(cases):
        tiles, flat = host.tiles_from(ids, lists)
        per.append((tiles[: len(ids)].copy(), flat))
        tiles = tiles[: len(ids)].copy()
        tiles["list_off"] += len(all_lists)
        all_tiles.extend(tiles.tolist())
        all_lists.extend(flat[: sum(len(l) for l in lists)].tolist())
        tile0[i + 1] = tile0[i] + len(ids)
    tiles = np.array(all_tiles, dtype=native.tile_summary_dtype)
    lists = np.array(all_lists, dtype=native.id_count_dtype)
    dec, ids_out, asg_out = eng.debug_decide(tile0, tiles, lists, threshold=10)
    kinds = set()
    for i, (ids, lsts, x) in enumerate(cases):
        a, b = int(tile0[i]), int(tile0[i + 1])
        t, fl = per[i]
        h_ids, h_b, na = host.smooth_tiles(t, fl, b - a, 10)
        assert np.array_equal(ids_out[a:b], h_ids) and np.array_equal(asg_out[a:b], h_b), (i, b - a)
        d = host.decide_read(t, fl, b - a, threshold=10)
        got = dec[i]
        assert (int(got["kind"]), int(got["num_tiles"]), int(got["num_assigned"]), int(got["trim_start"]), int(got["trim_end"])) == (d.kind, d.num_tiles, d.num_assigned, d.trim_start, d.trim_end), (i, b - a)
        kinds.add(int(got["kind"]))
    assert len(kinds) >= 2, kinds
    eng.close()
