"""GPU: a window of reads committed as ONE batch (grp_batch_insert_reads / _classify / _undo /
_end, include/grpath.h) against the oracle's serial process_read loop — records, ID
allocation and the miBF end state (IDs and counts of every rank), at the engine level.
The driver below is the protocol of the header comment: decide the window against the state
in front of it, apply its inserts at once, decide again through the log, compare in order,
take back the inserts from the first read that differs on."""
import numpy as np
import pytest

from helpers import default_seeds

from goldrush_amd import native as native_mod

pytestmark = pytest.mark.gpu


def _plan(d, reads, first, tile, block, ids_inserted, shared=False):
    """the inserts the decisions `d` ask for, with the IDs the serial loop would allocate;
    shared: the last insert in front was a trimmed read whose last ID block carries the next
    insert's first ID (bit 31 of the floors, include/grpath.h)"""
    ins, floors, firsts = [], [], []
    for j, r in enumerate(d):
        floors.append((ids_inserted + 1) | (0x80000000 if shared else 0))
        kind = int(r["kind"])
        if kind == 2:
            ins.append((first + j, 0, int(r["num_tiles"]), ids_inserted + 1, 0))
            firsts.append(ids_inserted + 1)
            ids_inserted += 1 + len(reads[first + j]) // (tile * block)
            shared = False
        elif kind == 4:
            ts, te = int(r["trim_start"]), int(r["trim_end"])
            ins.append((first + j, ts, te + 1, ids_inserted + 1, 1))
            firsts.append(ids_inserted + 1)
            ids_inserted += 1 + (te - ts) // block
            shared = (te - ts + 1) % block == 0
        else:
            firsts.append(0)
    return ins, floors, firsts, ids_inserted, shared


def _same(a, b):
    return int(a["kind"]) == int(b["kind"]) and int(a["num_tiles"]) == int(b["num_tiles"]) and (
        int(a["kind"]) != 4 or (int(a["trim_start"]), int(a["trim_end"])) == (int(b["trim_start"]), int(b["trim_end"])))


def batch_commit(eng, b, reads, tile, block, window, stats, verify=None):
    """-> [(read, kind, num_tiles, num_assigned, trim_start, trim_end, first_id)]
    verify: None = the second query (grp_batch_classify); "check" = grp_batch_verify, and the second query behind
    it: identical bytes; "chain" = grp_batch_verify alone, the first decisions of the next window taken from the
    reads it decided behind the batch (what the classifier does)"""
    out = []
    pos, ids_inserted, shared = 0, 0, False
    n = len(reads)
    carried = None  # (position, decisions) of the reads decided behind the previous batch

    def record(j, r, first_id):
        k4 = int(r["kind"]) == 4
        out.append((j, int(r["kind"]), int(r["num_tiles"]), int(r["num_assigned"]), int(r["trim_start"]) if k4 else 0, int(r["trim_end"]) if k4 else 0, first_id))

    shrink = 0
    while pos < n:
        cnt = max(1, min(window >> shrink, n - pos))
        shrink = 0
        if carried is not None and carried[0] == pos and len(carried[1]) > 0:
            cnt = min(cnt, len(carried[1]))
            d0 = carried[1][:cnt]
            stats["carried"] = stats.get("carried", 0) + 1
        else:
            d0 = eng.classify_reads(b, pos, cnt)
        carried = None
        ins, floors, firsts, ids_end, shared_end = _plan(d0, reads, pos, tile, block, ids_inserted, shared)
        if not ins:
            for j in range(cnt):
                record(pos + j, d0[j], 0)
            pos += cnt
            continue
        try:
            eng.batch_insert_reads(b, ins, block, pos)
        except native_mod.GrpError as e:
            # reads of the window overlap each other: too many ranks shared inside the batch;
            # nothing was inserted — a smaller window
            assert e.code == native_mod.GRP_ERR_NOMEM and cnt > 1
            stats["too_crowded"] = stats.get("too_crowded", 0) + 1
            shrink = 1
            while (window >> shrink) >= cnt:
                shrink += 1
            continue
        extra = 0
        if verify is None:
            d1 = eng.batch_classify(b, pos, cnt, floors)
        else:
            extra = min(max(1, window // 2), n - pos - cnt)
            dv = eng.batch_verify(b, pos, cnt, extra, floors + [0x7FFFFFFF] * extra)
            d1 = dv[:cnt]
            if verify == "check":
                dq = eng.batch_classify(b, pos, cnt + extra, floors + [0x7FFFFFFF] * extra)
                assert dv.tobytes() == dq.tobytes(), [(j, dv[j], dq[j]) for j in range(cnt + extra) if dv[j].tobytes() != dq[j].tobytes()][:3]
        bad = next((j for j in range(cnt) if not _same(d0[j], d1[j])), None)
        stats["batches"] += 1
        if bad is None:
            eng.batch_end()
            if verify == "chain" and extra:
                carried = (pos + cnt, dv[cnt:])
            for j in range(cnt):
                record(pos + j, d1[j], firsts[j])
            ids_inserted, shared = ids_end, shared_end
            pos += cnt
            continue
        # the batch was not the serial loop from read `bad` on: its insert and the ones behind it
        # are taken back; `bad` is committed by its second decision (taken against the state in
        # front of its own insert, the reads in front of it being confirmed)
        stats["undone"] += 1
        eng.batch_undo(pos + bad, floors[bad] & 0x7FFFFFFF)
        ins1, _, firsts1, ids_inserted, shared = _plan(d1[: bad + 1], reads, pos, tile, block, ids_inserted, shared)
        if ins1 and ins1[-1][0] == pos + bad:
            ri, ts, te, fid, off = ins1[-1]
            eng.insert_read(b, ri, ts, te, block, fid, off)
        for j in range(bad + 1):
            record(pos + j, d1[j], firsts1[j])
        pos += bad + 1
    return out


def _run(oracle, native, reads, tile, k, h, m, block, window, key, verify=None):
    from oracle_engine import cached_serial_reference

    seeds = default_seeds(h)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    eng.finalize()
    exp, ref_ids, ref_counts, _ = cached_serial_reference(key, oracle, m, seeds, tile, k, reads, block=block)
    stats = {"batches": 0, "undone": 0}
    got = batch_commit(eng, b, reads, tile, block, window, stats, verify)
    assert got == [e[:7] for e in exp]
    ids, counts = eng.export_ids()
    assert np.array_equal(counts, ref_counts)
    assert np.array_equal(ids, ref_ids)
    if verify is not None:
        stats["verify"] = eng.verify_stats()
        assert stats["verify"]["fallbacks"] == 0 and stats["verify"]["patched"] > 0, stats
        # the patch's self-check never fired (an older ID gained a frame / a count below zero would be a logic error of
        # k_batch_delta: the run would stay exact — the tile is queried again — but it must not go unnoticed)
        assert stats["verify"]["impossible_deltas"] == 0, stats
    eng.close()
    return stats, exp


@pytest.mark.parametrize("window,verify", [(2, None), (7, None), (32, None), (200, None), (7, "check"), (32, "check"), (200, "check"), (7, "chain"), (32, "chain"), (200, "chain")])
def test_batches_equal_the_serial_loop(oracle, native, window, verify):
    """A genome covered ~5x: the first reads insert, later ones are assigned or trimmed, so
    windows mix confirmed batches and batches taken back (a read that overlaps an earlier
    read of its own window decides differently once that read is in the filter)."""
    from goldrush_amd import synth

    tile, k, h, block = 500, 22, 3, 4
    g = synth.random_genome(150_000, 21)
    reads = [r[1] for r in synth.make_reads(g, 140, mean_len=5000, min_len=3500, seed=22, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    stats, exp = _run(oracle, native, reads, tile, k, h, m, block, window, "loop_basic", verify)
    assert stats["batches"] > 0
    if verify == "chain" and window == 7:
        assert stats.get("carried", 0) > 0  # first decisions taken from the reads decided behind a batch (larger windows on this genome are nearly always taken back)
    if window >= 7:
        assert stats["undone"] > 0  # the take-back path ran
    assert {e[1] for e in exp} >= {2, 3, 5}


def test_claim_epochs_wrap(oracle, native, monkeypatch):
    """The claim of the collect pass lives in the rank's count word under the batch's epoch (round 4); after 1023
    batches the epochs wrap and the claims are swept out of the words.  Here after every third batch: the same
    commits, IDs and counts as the serial loop, with chained ranks, undone batches and the verify pass looking
    records up through claims of the current epoch only."""
    from goldrush_amd import synth

    monkeypatch.setenv("GRP_BATCH_EPOCHS", "3")
    tile, k, h, block = 500, 22, 3, 4
    g = synth.random_genome(150_000, 21)
    reads = [r[1] for r in synth.make_reads(g, 140, mean_len=5000, min_len=3500, seed=22, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    from oracle_engine import cached_serial_reference

    seeds = default_seeds(h)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    eng.finalize()
    exp, ref_ids, ref_counts, _ = cached_serial_reference("loop_basic", oracle, m, seeds, tile, k, reads, block=block)
    stats = {"batches": 0, "undone": 0}
    got = batch_commit(eng, b, reads, tile, block, 7, stats, "check")
    assert got == [e[:7] for e in exp]
    ids, counts = eng.export_ids()
    assert np.array_equal(counts, ref_counts) and np.array_equal(ids, ref_ids)
    vs = eng.verify_stats()
    assert stats["batches"] >= 9 and vs["claim_sweeps"] >= stats["batches"] // 3 - 1 and vs["claim_sweeps"] >= 2, (stats, vs)
    eng.close()


@pytest.mark.parametrize("window,verify", [(7, None), (32, "check"), (200, "chain")])
def test_batches_on_a_repeat_rich_genome(oracle, native, window, verify):
    """a third of this genome is repeat copies (units of 2-5 kb, 5-12 copies, 1-4 % divergence, either
    strand), so reads of one window share ranks WITHOUT overlapping — chains, confirmations that fail far from any
    overlap, IDs of other loci in the votes.  Everything the batches do must still be the serial loop: records, IDs,
    every count."""
    from goldrush_amd import synth

    tile, k, h, block = 500, 22, 3, 4
    g = synth.repeat_genome(150_000, 31)
    reads = [r[1] for r in synth.make_reads(g, 140, mean_len=5000, min_len=3500, seed=32, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    stats, exp = _run(oracle, native, reads, tile, k, h, m, block, window, "loop_repeats", verify)
    assert stats["batches"] > 0 and {e[1] for e in exp} >= {2, 3}


@pytest.mark.parametrize("verify", [None, "check", "chain"])
def test_batches_on_a_small_crowded_filter(oracle, native, verify):
    """A filter far too small for its reads (occupancy ~0.5, most ranks shared by many tiles):
    ranks written by several reads and blocks of one batch, overwrites of non-zero IDs, the
    overflow table — the log has to restore all of it."""
    from goldrush_amd import synth

    tile, k, h, block = 250, 22, 3, 2
    g = synth.random_genome(400_000, 5)
    reads = [r[1] for r in synth.make_reads(g, 60, mean_len=4000, min_len=1500, seed=6, max_len=8000)]
    m = 1 << 19
    stats, exp = _run(oracle, native, reads, tile, k, h, m, block, 16, "batch_crowded", verify)
    assert stats["batches"] > 0
    if verify is not None:  # on this filter nearly every frame is dirty: tiles that cannot be certified are part of the test
        assert stats["verify"]["flagged"] > 0


@pytest.mark.parametrize("h", [3, 5])
def test_batch_of_long_reads_equals_inserts_one_by_one(native, h):
    """BASELINE geometry (25 kb reads, tile 1000, block 10, a small genome so that the reads of a
    batch share many ranks): 16 whole-read inserts as ONE batch — several hundred workgroups,
    more than one per compute unit — leave the IDs and counts of every rank exactly as the
    same inserts one by one (grp_insert_read, itself pinned to the oracle in test_gpu_parity);
    the second decisions equal decisions taken between the serial inserts; taking back the
    second half equals inserting only the first half."""
    from goldrush_amd import host

    n, nb = 64, 16
    k, tile, block, G = 22, 1000, 10, 8_000_000
    seeds = default_seeds(h)
    hl = host.load()
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, G, h), 1, 0.1)
    dr = native.synth_reads(n, G)
    lens = np.ascontiguousarray(dr.lens, dtype=np.uint32)

    def engine():
        eng = native.Engine(k, h, tile, m, seeds)
        rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
        eng.bv_insert(rb)
        eng.finalize()
        return eng, rb

    ins, floors, ids = [], [], 0
    for j in range(nb):
        floors.append(ids + 1)
        ins.append((j, 0, int(lens[j]) // tile, ids + 1, 0))
        ids += 1 + int(lens[j]) // (tile * block)

    a, ra = engine()
    serial_dec, half_state = [], None
    for j, (ri, ts, te, fid, off) in enumerate(ins):
        if j == nb // 2:
            half_state = a.export_ids()
        serial_dec.append(a.classify_reads(ra, j, 1)[0])
        a.insert_read(ra, ri, ts, te, block, fid, off)
    ids_a, counts_a = a.export_ids()
    a.close()

    b, rb_ = engine()
    b.classify_reads(rb_, 0, nb)  # the first decisions: their summaries stay with the engine for grp_batch_verify
    b.batch_insert_reads(rb_, ins, block, 0)
    dv = b.batch_verify(rb_, 0, nb, 4, floors + [0x7FFFFFFF] * 4)
    vs = b.verify_stats()
    assert vs["fallbacks"] == 0 and vs["patched"] == sum(int(lens[j]) // tile for j in range(nb)) and vs["queried"] == 0, vs
    d1 = b.batch_classify(rb_, 0, nb, floors)
    assert dv[:nb].tobytes() == d1.tobytes()
    assert dv[nb:].tobytes() == b.batch_classify(rb_, nb, 4, [0x7FFFFFFF] * 4).tobytes()
    # a stripe of the window (what one rank of a multi-GPU run asks for), and an empty one
    stripe = b.batch_classify(rb_, 5, 6, floors[5:11])
    assert stripe.tobytes() == d1[5:11].tobytes()
    assert len(b.batch_classify(rb_, nb, 0, [0])) == 0
    b.batch_end()
    ids_b, counts_b = b.export_ids()
    b.close()
    assert np.array_equal(ids_a, ids_b) and np.array_equal(counts_a, counts_b)
    fields = ("kind", "num_tiles", "num_assigned", "trim_start", "trim_end", "hits", "misses")
    assert [[int(x[f]) for f in fields] for x in serial_dec] == [[int(x[f]) for f in fields] for x in d1]

    c, rc_ = engine()
    c.batch_insert_reads(rc_, ins, block, 0)
    c.batch_classify(rc_, 0, nb, floors)
    c.batch_undo(nb // 2, floors[nb // 2])
    ids_c, counts_c = c.export_ids()
    c.close()
    assert np.array_equal(ids_c, half_state[0]) and np.array_equal(counts_c, half_state[1])
    dr.free()


@pytest.mark.parametrize("h", [3, 5])
def test_very_long_read_inserted_whole(oracle, native, h):
    """A 130 kb read (130 tiles of 1000: more than 256 workgroups of the collect kernels are in
    flight at once) inserted whole by grp_insert_read and as a batch of one: IDs / counts of every
    rank against the oracle's block-by-block inserts."""
    from goldrush_amd import synth

    tile, k, block = 1000, 22, 10
    seeds = default_seeds(h)
    g = synth.random_genome(400_000, 77)
    reads = [g[1000:1000 + 130_500].tobytes(), g[200_000:200_000 + 30_000].tobytes()]
    orc = oracle
    m = orc.load().orc_calc_optimal_size(4_000_000, 1, 0.1)
    mf = orc.MiBF(m, orc.Seeds(seeds), tile, k)
    for s in reads:
        mf.bv_insert_read(s)
    mf.finalize()
    nt = len(reads[0]) // tile
    for bs in range(0, nt, block):
        mf.insert_read_tiles(reads[0], bs, min(bs + block, nt), 1 + bs // block)
    for variant in ("insert_read", "batch"):
        eng = native.Engine(k, h, tile, m, seeds)
        b = eng.upload(reads)
        eng.bv_insert(b)
        eng.finalize()
        if variant == "insert_read":
            eng.insert_read(b, 0, 0, nt, block, 1, 0)
        else:
            eng.batch_insert_reads(b, [(0, 0, nt, 1, 0)], block, 0)
            eng.batch_end()
        ids, counts = eng.export_ids()
        assert np.array_equal(counts, mf.counts()), variant
        assert np.array_equal(ids, mf.ids()), variant
        eng.close()
    mf.close()


def test_batch_api_refuses_misuse(native):
    """Error behaviour of the grp_batch_* entry points (include/grpath.h): arguments are checked
    before anything is touched, one batch at a time, calls without a batch say so."""
    from goldrush_amd import synth

    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    g = synth.random_genome(60_000, 3)
    reads = [r[1] for r in synth.make_reads(g, 6, mean_len=4000, min_len=3000, seed=4, max_len=6000)]
    eng = native.Engine(k, h, tile, 1 << 22, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    nt = [len(r) // tile for r in reads]

    def code(fn, *a):
        with pytest.raises(native_mod.GrpError) as e:
            fn(*a)
        return e.value.code

    assert code(eng.batch_insert_reads, b, [(0, 0, nt[0], 1, 0)], block, 0) == native_mod.GRP_ERR_STATE  # before grp_finalize
    eng.finalize()
    ids0, counts0 = eng.export_ids()
    assert code(eng.batch_classify, b, 0, 2, [1, 1]) == native_mod.GRP_ERR_STATE        # no batch
    assert code(eng.batch_undo, 0, 1) == native_mod.GRP_ERR_STATE                        # no batch
    eng.batch_end()                                                                      # nothing to end: fine
    assert code(eng.batch_insert_reads, b, [(1, 0, nt[1], 1, 0), (0, 0, nt[0], 9, 0)], block, 0) == native_mod.GRP_ERR_INVALID  # not ascending
    assert code(eng.batch_insert_reads, b, [(0, 0, nt[0] + 1, 1, 0)], block, 0) == native_mod.GRP_ERR_INVALID                   # tiles outside the read
    assert code(eng.batch_insert_reads, b, [(0, 0, nt[0], 1, 0)], block, 1) == native_mod.GRP_ERR_INVALID                       # a read in front of the window
    assert code(eng.batch_insert_reads, b, [(0, 2, 2, 1, 0)], block, 0) == native_mod.GRP_ERR_INVALID                           # empty range
    assert code(eng.batch_insert_reads, b, [(0, 0, nt[0], 1, 2)], block, 0) == native_mod.GRP_ERR_INVALID                       # id_offset is 0 or 1
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, ids0) and np.array_equal(counts, counts0)                 # nothing was touched
    eng.batch_insert_reads(b, [(0, 0, nt[0], 1, 0), (2, 0, nt[2], 9, 0)], block, 0)
    assert code(eng.batch_insert_reads, b, [(3, 0, nt[3], 20, 0)], block, 3) == native_mod.GRP_ERR_STATE  # one batch at a time
    eng.batch_undo(0, 1)                                                                  # everything taken back
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, ids0) and np.array_equal(counts, counts0)
    eng.close()


def test_batches_with_one_tile_id_blocks(oracle, native, monkeypatch):
    """-b 1: the only geometry where a trimmed read's last ID block carries the next
    insert's first ID — the ambiguous-floor path of k_query<.., VER> (bit 31 of id_floor, the
    writer looked up through the batch's records) against the oracle's serial loop."""
    from goldrush_amd import host, synth
    from oracle_engine import serial_reference

    monkeypatch.setenv("GRP_BATCH", "force")
    tile, k, h, block = 500, 22, 3, 1
    seeds = default_seeds(h)
    g = synth.random_genome(150_000, 21)
    reads = [r[1] for r in synth.make_reads(g, 120, mean_len=5000, min_len=3500, seed=22, max_len=9000)]
    m = oracle.load().orc_calc_optimal_size(2_000_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block)
    assert sum(1 for e in exp if e[1] == 4) >= 4
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    assert eng.finalize() == mf_ref.pop
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h)
    cls.run(b._h, b.lens)
    eng.sync()
    assert [c[:8] for c in cls.commits] == exp
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, mf_ref.ids()) and np.array_equal(counts, mf_ref.counts())
    assert cls.state()["batches"] >= 2
    eng.close()


@pytest.mark.parametrize("entry", ["batch", "read", "stream"])
def test_claim_loops_keep_every_touch(oracle, native, entry):
    """The claim-loop rule of the insert kernels: every kernel that claims ranks
    per seed — k_batch_collect [batch], k_insert_collect [read], the collect unit inside the streaming launch
    [stream] — walks ONE seed per claim loop; a form that kept three seeds live across the loops lost ~10 % of the
    third seed's touches once more than 256 workgroups were in flight.  Small enough for every run, large enough
    for that: ~1000 tiles x 3 seeds x 4 workgroups of overlapping reads, every ID and count against the oracle."""
    from goldrush_amd import synth

    k, h, tile, block = 22, 3, 1000, 10
    seeds = default_seeds(h)
    g = synth.random_genome(400_000, 77)
    reads = [r[1] for r in synth.make_reads(g, 44, mean_len=24000, min_len=18000, seed=78, max_len=40000)]
    m = oracle.load().orc_calc_optimal_size(4_000_000, 1, 0.1)
    eng = native.Engine(k, h, tile, m, seeds)
    b = eng.upload(reads)
    eng.bv_insert(b)
    omf = oracle.MiBF(m, oracle.Seeds(seeds), tile, k)
    omf.bv_insert_reads(reads)
    assert eng.finalize() == omf.finalize()
    ins, next_id = [], 0
    for ri, seq in enumerate(reads):
        nt = len(seq) // tile
        ins.append((ri, 0, nt, next_id + 1, 0))
        for bs in range(0, nt, block):
            omf.insert_read_tiles(seq, bs, min(bs + block, nt), next_id + 1 + bs // block)
        next_id += 1 + len(seq) // (tile * block)
    assert sum(i[2] for i in ins) * h * 4 > 10_000  # workgroups of the collect grids, all reads together
    if entry == "batch":
        eng.batch_insert_reads(b, ins, block, 0)
        eng.batch_end()
    elif entry == "read":
        for (ri, a, e, fid, off) in ins:
            eng.insert_read(b, ri, a, e, block, fid, off)
    else:
        # a resumable window whose every read is a whole-read insert record (no tile ever counts as assigned): the
        # launch parks behind each, stream_insert is applied by its persistent workgroups (768 of them)
        import time
        dp = dict(threshold=1 << 30, unassigned_min=1, assigned_max=1 << 30)
        first, gen, restarts, applied = 0, 1, 0, 0
        v = eng.stream_begin(b, 0, len(reads), 0, resumable=True, **dp)
        j = 0
        while j < len(ins):
            ri, a, e, fid, off = ins[j]
            t0 = time.time()
            while int(v["pad"][j - first]) != gen:
                assert time.time() - t0 < 60 and not eng.stream_poll(0)
            if int(v["kind"][j - first]) == 0:
                # handed back (the window's list arena is used up by the tiles queried again behind every insert):
                # the product takes such a read through the synchronous path; here a new window starts at it
                eng.stream_abort(0)
                eng.stream_end(0)
                restarts += 1
                assert restarts < 20
                first, gen = j, 1
                v = eng.stream_begin(b, j, len(reads) - j, 0, resumable=True, **dp)
                continue
            assert int(v["kind"][j - first]) == 2 and int(v["num_tiles"][j - first]) == e
            gen = eng.stream_insert(0, ri, a, e, block, fid, off)
            applied += 1
            j += 1
        assert applied == len(ins)
        t0 = time.time()
        while not eng.stream_poll(0):
            assert time.time() - t0 < 60
        eng.stream_end(0)
    ids, counts = eng.export_ids()
    oi, oc = omf.ids(), omf.counts()
    bad = np.flatnonzero((ids != oi) | (counts != oc))
    assert bad.size == 0, (entry, bad.size, bad[:10], ids[bad[:10]], oi[bad[:10]], counts[bad[:10]], oc[bad[:10]])
    assert int((counts > 1).sum()) > 1_000
    eng.close()


def test_large_batch_at_c2_filter_size_against_the_oracle(oracle, native):
    """ONE batch of 320 whole-read inserts on C2's own filter (m = 61 146 729 472:
    65 GB of buckets) — 24 M (frame, seed) records, a collect grid of ~96 000 workgroups, thousands
    of them in flight — against the ORACLE's serial inserts: every ID and count, the chained ranks
    (reads of a 3 Mbp genome overlap each other heavily) and a partial take-back included."""
    from goldrush_amd import host
    from test_gpu_wide import _oracle_bits_view  # noqa: F401  (same helpers, same sizes)

    hl = host.load()
    k, h, tile, block = 22, 3, 1000, 10
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, 3_000_000_000, h), 1, 0.1)
    seeds = default_seeds(h)
    n_reads = 320  # 24 M (frame, seed) records, ~96 000 collect workgroups (round 4 ran 200 for a while: the time was the oracle's OpenMP team on the GPU box, tests/conftest.py)
    dr = native_mod.synth_reads(n_reads, 3_000_000, mean_len=25000, min_len=20000, seed=19)
    eng = native_mod.Engine(k, h, tile, m, seeds)
    batch = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    eng.bv_insert(batch)
    reads = dr.download(0, n_reads)
    omf = oracle.MiBF(m, oracle.Seeds(seeds), tile, k)
    omf.bv_insert_reads(reads)
    assert eng.finalize() == omf.finalize()
    ins, next_id = [], 0
    for ri, seq in enumerate(reads):
        nt = len(seq) // tile
        ins.append((ri, 0, nt, next_id + 1, 0))
        for bs in range(0, nt, block):
            omf.insert_read_tiles(seq, bs, min(bs + block, nt), next_id + 1 + bs // block)
        next_id += 1 + len(seq) // (tile * block)
    eng.batch_insert_reads(batch, ins, block, 0)
    eng.batch_end()
    ids, counts = eng.export_ids()
    oi, oc = omf.ids(), omf.counts()
    bad = np.flatnonzero((ids != oi) | (counts != oc))
    assert bad.size == 0, (bad.size, bad[:10], ids[bad[:10]], oi[bad[:10]], counts[bad[:10]], oc[bad[:10]])
    assert int((counts > 1).sum()) > 10_000  # ranks touched by several ID blocks: chains were replayed
    # the same batch again on top, then the second half taken back: the state of the first half on top
    ins2 = [(r, a, b2, fid + next_id, off) for (r, a, b2, fid, off) in ins]
    half = n_reads // 2
    for (r, a, b2, fid, off) in ins2[:half]:
        for bs in range(0, b2, block):
            omf.insert_read_tiles(reads[r], bs, min(bs + block, b2), fid + bs // block)
    eng.batch_insert_reads(batch, ins2, block, 0)
    eng.batch_undo(half, ins2[half][3])
    ids, counts = eng.export_ids()
    assert np.array_equal(ids, omf.ids()) and np.array_equal(counts, omf.counts())
    eng.close()
    dr.free()
