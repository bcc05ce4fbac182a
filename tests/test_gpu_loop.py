"""GPU: the ordered commit loop on the device (grp_commit_loop_*) against the oracle's serial
process_read loop — records, ID allocation, the miBF end state — at the engine level
(through the C ABI), including a silver-path rollover, a read handed back (more tiles than
the device decision holds) and reads without a tile."""
import numpy as np
import pytest

from helpers import default_seeds

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_dev_hooks(native):
    """The commit loop is frozen and only compiled into developer builds (make -C goldrush_amd/csrc DEV=1)."""
    if not native.load().grp_dev_hooks():
        pytest.skip("libgrpath_hip.so was built without GRP_DEV_HOOKS (make DEV=1): the commit loop is not in the product build")


def _setup(oracle, native, reads, tile, k, h, m):
    seeds = default_seeds(h)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    eng.finalize()
    return eng, b, seeds


def _records(rec):
    return [(int(r["kind"]), int(r["num_tiles"]), int(r["num_assigned"]), int(r["trim_start"]) if r["kind"] == 4 else 0, int(r["trim_end"]) if r["kind"] == 4 else 0,
             int(r["pad"])) for r in rec]


@pytest.mark.parametrize("depth,lt,whole", [(0, 0, False), (1, 0, False), (3, 512, False), (64, 0, False), (0, 0, True), (2, 512, True), (64, 0, True)])
def test_commit_loop_equals_serial_reference(oracle, native, depth, lt, whole, monkeypatch):
    from goldrush_amd import synth
    from oracle_engine import cached_serial_reference

    if lt:
        monkeypatch.setenv("GRP_LOOP_LT", str(lt))
    tile, k, h, block = 500, 22, 3, 4
    g = synth.random_genome(150_000, 21)
    reads = [r[1] for r in synth.make_reads(g, 140, mean_len=5000, min_len=3500, seed=22, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    eng, b, seeds = _setup(oracle, native, reads, tile, k, h, m)
    exp, ref_ids, ref_counts, _ = cached_serial_reference("loop_basic", oracle, m, seeds, tile, k, reads, block=block)
    rec, res = eng.commit_loop(b, 0, len(reads), block=block, max_depth=depth, whole_tiles=whole)
    assert res["status"] == native.GRP_LOOP_DONE and res["reads_committed"] == len(reads)
    assert _records(rec) == [(e[1], e[2], e[3], e[4], e[5], e[6]) for e in exp]
    assert res["inserts"] == sum(1 for e in exp if e[1] in (2, 4)) and res["inserts"] > 20
    assert {e[1] for e in exp} >= {2, 3, 5}  # the stream exercises the decision kinds
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, ref_ids) and np.array_equal(counts, ref_counts)
    eng.close()


def test_commit_loop_in_pieces_with_rollover_handback_and_empty_reads(oracle, native):
    """The loop stops by itself at a silver-path rollover and in front of a read it cannot
    decide; the caller continues behind it — the sequence of launches equals the serial loop."""
    from goldrush_amd import synth
    from oracle_engine import serial_reference

    tile, k, h, block = 250, 22, 3, 10
    g = synth.random_genome(200_000, 31)
    reads = [r[1] for r in synth.make_reads(g, 90, mean_len=6000, min_len=2000, seed=32, max_len=30000)]
    # 0 tiles, 300 tiles (more than the device decision holds), 1 tile
    reads[3] = g[60000:60000 + 249].tobytes()
    reads[40] = g[1000:1000 + 75_100].tobytes()
    reads[41] = g[90000:90000 + 260].tobytes()
    reads[0] = g[70000:70000 + 100].tobytes()  # a launch that starts with a read decided by the host
    m = oracle.load().orc_calc_optimal_size(3_000_000, 1, 0.1)
    target = 150_000
    eng, b, seeds = _setup(oracle, native, reads, tile, k, h, m)
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=target, max_paths=100)
    got = []
    pos, ids_inserted, inserted_bases, path = 0, 0, 0, 1
    n_roll = n_hand = 0
    while pos < len(reads):
        rec, res = eng.commit_loop(b, pos, len(reads) - pos, block=block, silver=True, target_bases=target, ids_inserted=ids_inserted, inserted_bases=inserted_bases)
        for j, r in enumerate(_records(rec)):
            got.append((pos + j,) + r + (path,))
        pos += res["reads_committed"]
        ids_inserted, inserted_bases = res["ids_inserted"], res["inserted_bases"]
        if res["status"] == native.GRP_LOOP_ROLLOVER:
            assert inserted_bases > target
            n_roll += 1
            path += 1
            eng.reset_ids()
            ids_inserted = inserted_bases = 0
        elif res["status"] == native.GRP_LOOP_HANDBACK:
            # the caller's synchronous path for this one read
            n_hand += 1
            d = eng.classify_reads(b, pos, 1)[0]
            kind, nt = int(d["kind"]), int(d["num_tiles"])
            first = 0
            if kind in (2, 4):
                ids_inserted += 1
                first = ids_inserted
                if kind == 2:
                    eng.insert_read(b, pos, 0, nt, block, first, 0)
                    ids_inserted += len(reads[pos]) // (tile * block)
                    inserted_bases += len(reads[pos])
                else:
                    ts, te = int(d["trim_start"]), int(d["trim_end"])
                    eng.insert_read(b, pos, ts, te + 1, block, first, 1)
                    ids_inserted += (te - ts) // block
                    off = ts * tile
                    inserted_bases += len(reads[pos]) - off if te == nt - 1 else min(len(reads[pos]) - off, (te - ts + 1) * tile)
            got.append((pos, kind, nt, int(d["num_assigned"]), int(d["trim_start"]) if kind == 4 else 0, int(d["trim_end"]) if kind == 4 else 0, first, path))
            pos += 1
            if kind in (2, 4) and inserted_bases > target:
                path += 1
                eng.reset_ids()
                ids_inserted = inserted_bases = 0
        else:
            assert res["status"] == native.GRP_LOOP_DONE
    assert got == exp
    assert n_roll >= 1 and n_hand == 1
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, mf_ref.ids()) and np.array_equal(counts, mf_ref.counts())
    eng.close()


def test_commit_loop_full_size_equals_windows(native, monkeypatch):
    """BASELINE geometry (25 kb reads, G = 100e6 filter, W = 13 buckets with overflow IDs):
    the first 6000 reads of the C1 stream through the commit loop and through the
    synchronous windows: identical records and identical ID / count arrays."""
    from goldrush_amd import host

    k, h, tile, block = 22, 3, 1000, 10
    seeds = default_seeds(h)
    hl = host.load()
    G = 100_000_000
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, G, h), 1, 0.1)
    n = 6000
    dr = native.synth_reads(n, G)
    out = []
    for mode in ("loop", "windows"):
        monkeypatch.setenv("GRP_LOOP", "force" if mode == "loop" else "off")
        eng = native.Engine(k, h, tile, m, seeds)
        rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
        eng.bv_insert(rb)
        pop = eng.finalize()
        cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, target_bases=int(0.9 * G), max_paths=1)
        cls.run(rb._h, rb.lens)
        eng.sync()
        st = cls.state()
        # a sample of the arrays (10 M ranks from the middle) and all records
        ids, counts = eng.export_ids(pop // 2, 10_000_000)
        out.append((list(cls.commits), ids, counts, {k_: st[k_] for k_ in ("hits", "misses", "queries", "ids_inserted", "inserted_bases", "inserts")}))
        cls.close()
        eng.close()
    dr.free()
    assert out[0][0] == out[1][0]
    assert out[0][3] == out[1][3] and out[0][3]["inserts"] > 3000
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
