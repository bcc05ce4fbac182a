"""CPU: the product's order-exact classifier (speculative windows + in-order
commit, gr_classifier.cpp) driven by an oracle-backed engine must reproduce the
reference's serial loop exactly, for every window size and for 2 ranks."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import default_seeds


def _workload(seed=7, n=60, glen=120_000, tile=500):
    from goldrush_amd import synth

    g = synth.random_genome(glen, seed)
    reads = [r[1] for r in synth.make_reads(g, n, mean_len=5000, min_len=3500, seed=seed + 1, max_len=9000)]
    return reads


def _strip(commits):
    return [c[:8] for c in commits]


@pytest.mark.parametrize("max_window", [1, 3, 16, 4096])
def test_classifier_matches_serial_loop(oracle, native, max_window):
    from goldrush_amd import host
    from oracle_engine import OracleEngine, serial_reference

    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    reads = _workload()
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=90_000, max_paths=2)
    eng = OracleEngine(oracle, m, seeds, tile, k, reads)
    cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, target_bases=90_000, max_paths=2, silver_path=True, max_window=max_window)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    finished = cls.run(None, lens, skipped_before=np.zeros(len(reads), dtype=np.uint32))
    assert _strip(cls.commits) == exp
    kinds = [e[1] for e in exp]
    assert kinds.count(2) >= 5 and kinds.count(4) >= 1 and (kinds.count(3) + kinds.count(5)) >= 5, kinds
    assert cls.rollovers == [2] and finished  # path 2 opened, then -M 2 reached: exit(0) point
    # the miBF ends in the same state as the serial loop's
    assert np.array_equal(eng.mf.ids(), mf_ref.ids()) and np.array_equal(eng.mf.counts(), mf_ref.counts())
    st = cls.state()
    assert st["reads_committed"] == len(exp)
    if max_window == 1:
        assert st["reads_queried"] == len(exp)  # no speculation, no waste


@pytest.mark.parametrize("max_window", [16, 40, 4096])
def test_pipelined_windows_match_serial_loop(oracle, native, max_window, monkeypatch):
    """Two windows in flight (classify_begin / classify_end): a window begun before an
    insert must never be committed after it."""
    from goldrush_amd import host
    from oracle_engine import OracleEngine, serial_reference

    monkeypatch.setenv("GRP_PIPELINE", "force")
    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    reads = _workload()
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=90_000, max_paths=2)
    eng = OracleEngine(oracle, m, seeds, tile, k, reads, pipelined=True)
    cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, target_bases=90_000, max_paths=2, silver_path=True, max_window=max_window)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    finished = cls.run(None, lens, skipped_before=np.zeros(len(reads), dtype=np.uint32))
    assert _strip(cls.commits) == exp
    assert cls.rollovers == [2] and finished
    assert np.array_equal(eng.mf.ids(), mf_ref.ids()) and np.array_equal(eng.mf.counts(), mf_ref.counts())
    assert eng.n_begun >= 2 and eng.n_abandoned >= 1  # the path was exercised, stale windows were dropped
    assert cls.state()["reads_committed"] == len(exp)


def test_kept_commits_are_the_commits_of_their_ranges(oracle, native):
    """gr_classifier_keep_commits: the commits of two ranges of reads stay inside the classifier (the witness bench.py
    hands to the oracle) — with and without a commit callback, in commit order, the rest is not kept."""
    from goldrush_amd import host
    from oracle_engine import OracleEngine

    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    reads = _workload()
    n = len(reads)
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    got = []
    for record in (True, False):
        eng = OracleEngine(oracle, m, seeds, tile, k, reads, pipelined=True)
        cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, max_window=16, record=record)
        cls.keep_commits(0, 5, n - 7, 7)
        cls.run(None, lens, skipped_before=np.zeros(n, dtype=np.uint32))
        kept = cls.kept_commits()
        assert [c[0] for c in kept] == list(range(5)) + list(range(n - 7, n))
        if record:
            assert kept == [c for c in cls.commits if c[0] < 5 or c[0] >= n - 7]
        got.append(kept)
    assert got[0] == got[1]


@pytest.mark.parametrize("max_window,redo_every", [(32, 0), (4096, 0), (40, 7)])
def test_streaming_windows_match_serial_loop(oracle, native, max_window, redo_every, monkeypatch):
    """Streaming windows (stream_begin / _abort / _end): records behind an insert are
    never committed, a record of kind 0 goes through the synchronous path."""
    from goldrush_amd import host
    from oracle_engine import OracleEngine, serial_reference

    monkeypatch.setenv("GRP_STREAM", "force")
    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    reads = _workload()
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=90_000, max_paths=2)
    eng = OracleEngine(oracle, m, seeds, tile, k, reads, streaming=True, redo_every=redo_every)
    cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, target_bases=90_000, max_paths=2, silver_path=True, max_window=max_window)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    skipped = np.zeros(len(reads), dtype=np.uint32)
    skipped[5] = 2
    finished = cls.run(None, lens, skipped_before=skipped)
    assert _strip(cls.commits) == exp
    assert cls.rollovers == [2] and finished
    assert np.array_equal(eng.mf.ids(), mf_ref.ids()) and np.array_equal(eng.mf.counts(), mf_ref.counts())
    assert eng.n_streams >= 3 and eng.n_stream_aborts >= 2
    assert (eng.n_redo > 0) == (redo_every > 0)
    assert cls.state()["reads_committed"] == len(exp)


@pytest.mark.parametrize("max_window,redo_every,refuse_every,lost_every", [(4096, 0, 0, 0), (32, 0, 0, 0), (40, 7, 0, 0), (4096, 0, 3, 0), (4096, 0, 0, 4), (32, 5, 2, 5)])
def test_streaming_windows_apply_inserts_themselves(oracle, native, max_window, redo_every, refuse_every, lost_every, monkeypatch):
    """Round 3: a streaming window parked at an insert record takes the insert from the host
    (stream_insert) and carries on behind the read — records of an older generation are never
    committed; an engine that refuses, or a launch that ends without applying the insert, falls
    back to the classic abort + insert_read."""
    from goldrush_amd import host
    from oracle_engine import OracleEngine, serial_reference

    monkeypatch.setenv("GRP_STREAM", "force")
    monkeypatch.setenv("GRP_BATCH", "off")
    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    reads = _workload()
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block)  # golden-path mode: one path, no rollover
    eng = OracleEngine(oracle, m, seeds, tile, k, reads, streaming=True, redo_every=redo_every, resume=True, resume_refuse_every=refuse_every, resume_lost_every=lost_every)
    cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, max_window=max_window)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    skipped = np.zeros(len(reads), dtype=np.uint32)
    skipped[5] = 2
    cls.run(None, lens, skipped_before=skipped)
    assert _strip(cls.commits) == exp
    assert [c[8:10] for c in cls.commits] == [c[8:10] for c in _hits_misses(oracle, m, seeds, tile, k, reads, block, silver=False)]
    assert np.array_equal(eng.mf.ids(), mf_ref.ids()) and np.array_equal(eng.mf.counts(), mf_ref.counts())
    kinds = [e[1] for e in exp]
    assert kinds.count(2) + kinds.count(4) >= 8
    assert eng.n_stream_inserts >= (3 if not lost_every else 2)
    if not (redo_every or refuse_every or lost_every) and max_window == 4096:
        assert eng.n_streams == 1 and eng.n_stream_aborts == 0  # ONE launch for the whole range, inserts and all
    assert (eng.n_stream_refused > 0) == (refuse_every > 0) and (eng.n_stream_lost > 0) == (lost_every > 0)
    if lost_every:
        # after a launch that could not apply its insert the windows end at inserts again
        assert eng.n_stream_lost == 1
    if refuse_every and max_window < 60:
        assert getattr(eng, "n_stream_busy", 0) >= 1  # a next window the engine could not begin yet: begun when the current one had ended
    assert cls.state()["reads_committed"] == len(exp)


@pytest.mark.parametrize("overlap_every,verify,silver", [(5, True, False), (3, False, False), (2, True, False), (4, True, True)])
def test_windows_of_batches_end_in_front_of_overlaps(oracle, native, overlap_every, verify, silver, monkeypatch):
    """Round 4: where most reads insert the classifier asks the engine (grp_window_overlap) for the first read of the
    window it is about to query that overlaps a read in front of it, and ends the window there.  Wherever the engine
    says so — here every n-th read of the stream, right or wrong — the commits and the filter stay the serial loop's;
    only the batch boundaries move."""
    from goldrush_amd import host
    from oracle_engine import OracleEngine, serial_reference

    monkeypatch.setenv("GRP_BATCH", "force")
    monkeypatch.setenv("GRP_BATCH_OVERLAP_P", "0")  # ask in every regime
    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    reads = _workload()
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    kw = dict(silver=True, target_bases=sum(len(r) for r in reads) // 9, max_paths=3) if silver else {}
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block, **kw)
    eng = OracleEngine(oracle, m, seeds, tile, k, reads, batching=True, batch_verify=verify, overlap_every=overlap_every)
    ckw = dict(silver_path=True, target_bases=kw["target_bases"], max_paths=3) if silver else {}
    cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, max_window=64, **ckw)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    cls.run(None, lens)
    assert _strip(cls.commits) == exp
    assert np.array_equal(eng.mf.ids(), mf_ref.ids()) and np.array_equal(eng.mf.counts(), mf_ref.counts())
    st = cls.state()
    assert st["batch_overlap_cuts"] >= 2 and 1 <= eng.n_overlap_calls == st["overlap_calls"] and st["reads_committed"] == len(exp)
    # switched off: the same commits, the engine is not asked
    monkeypatch.setenv("GRP_BATCH_OVERLAP", "off")
    eng2 = OracleEngine(oracle, m, seeds, tile, k, reads, batching=True, batch_verify=verify, overlap_every=overlap_every)
    cls2 = host.Classifier(None, eng2.vt, tile=tile, block=block, k=k, h=h, max_window=64, **ckw)
    cls2.run(None, lens)
    assert _strip(cls2.commits) == exp and cls2.state()["batch_overlap_cuts"] == 0 and eng2.n_overlap_calls == 0


@pytest.mark.parametrize("max_window", [4096, 32])
def test_streaming_windows_apply_inserts_in_silver_mode(oracle, native, max_window, monkeypatch):
    """Round 4: in silver mode too the parked window applies the host's inserts itself; the insert
    behind which the path rolls over (silver_path_check, goldrush_path.cpp:156-187: the ID array is reset) is known
    to the host in front of it — it ends the launches and goes the classic way."""
    from goldrush_amd import host
    from oracle_engine import OracleEngine, serial_reference

    monkeypatch.setenv("GRP_STREAM", "force")
    monkeypatch.setenv("GRP_BATCH", "off")
    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    reads = _workload()
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    target = sum(len(r) for r in reads) // 9
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=target, max_paths=3)
    eng = OracleEngine(oracle, m, seeds, tile, k, reads, streaming=True, resume=True)
    cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, max_window=max_window, silver_path=True, target_bases=target, max_paths=3)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    cls.run(None, lens)
    assert _strip(cls.commits) == exp
    assert np.array_equal(eng.mf.ids(), mf_ref.ids()) and np.array_equal(eng.mf.counts(), mf_ref.counts())
    st = cls.state()
    assert len({e[7] for e in exp}) >= 2                       # the run crossed a rollover
    assert st["stream_rollovers"] >= 1 and eng.n_stream_inserts >= 3
    assert st["stream_inserts"] + st["stream_rollovers"] <= st["inserts"]


@pytest.mark.parametrize("max_window,crowded,verify", [(2, 0, True), (5, 0, True), (64, 0, True), (4096, 0, True), (64, 3, True), (5, 0, False), (4096, 0, False)])
def test_batched_windows_match_serial_loop(oracle, native, max_window, crowded, verify, monkeypatch):
    """Windows committed as batches (batch_insert / _classify / _undo / _end): the window's
    inserts are applied before its reads are decided a second time; a read that decides
    differently behind an earlier read of its own window takes the batch back; a silver-path
    rollover ends a batch; a refused batch (too crowded) falls back to the classic commit."""
    from goldrush_amd import host
    from oracle_engine import OracleEngine, serial_reference

    monkeypatch.setenv("GRP_BATCH", "force")
    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    reads = _workload()
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=90_000, max_paths=2)
    eng = OracleEngine(oracle, m, seeds, tile, k, reads, batching=True, batch_crowded_above=crowded, batch_verify=verify)
    cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, target_bases=90_000, max_paths=2, silver_path=True, max_window=max_window)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    skipped = np.zeros(len(reads), dtype=np.uint32)
    skipped[4] = 3
    finished = cls.run(None, lens, skipped_before=skipped)
    assert _strip(cls.commits) == exp
    assert [c[8:10] for c in cls.commits] == [c[8:10] for c in _hits_misses(oracle, m, seeds, tile, k, reads, block)]
    assert cls.rollovers == [2] and finished
    assert np.array_equal(eng.mf.ids(), mf_ref.ids()) and np.array_equal(eng.mf.counts(), mf_ref.counts())
    st = cls.state()
    assert st["reads_committed"] == len(exp)
    assert eng.n_batches >= 2 and st["batches"] >= 1 and st["batch_reads"] > 0
    assert (eng.n_verifies > 0) == verify  # an engine with grp_batch_verify is asked through it, one without through grp_batch_classify
    if max_window >= 64 and not crowded:
        assert eng.n_batch_undone >= 1 and st["batches_undone"] == eng.n_batch_undone
    if crowded:
        assert eng.n_batch_refused >= 1 and st["batches_refused"] == eng.n_batch_refused


@pytest.mark.parametrize("max_window", [5, 4096])
def test_batches_with_one_tile_id_blocks(oracle, native, max_window, monkeypatch):
    """-b 1: the only geometry where the last ID block of a trimmed read carries the next insert's
    first ID (goldrush_path.cpp:1048-1049 / :1074) — the ambiguous floor of the batches' second
    query (bit 31 of id_floor) is exercised against the serial loop."""
    from goldrush_amd import host
    from oracle_engine import OracleEngine, serial_reference

    monkeypatch.setenv("GRP_BATCH", "force")
    tile, k, h, block = 500, 22, 3, 1
    seeds = default_seeds(h)
    reads = _workload()
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block)
    assert sum(1 for e in exp if e[1] == 4) >= 3  # trimmed inserts in front of other inserts
    eng = OracleEngine(oracle, m, seeds, tile, k, reads, batching=True)
    cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, max_window=max_window)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    cls.run(None, lens, skipped_before=np.zeros(len(reads), dtype=np.uint32))
    assert _strip(cls.commits) == exp
    assert np.array_equal(eng.mf.ids(), mf_ref.ids()) and np.array_equal(eng.mf.counts(), mf_ref.counts())
    assert cls.state()["batches"] >= 2


@pytest.mark.parametrize("mode", ["sync", "pipeline", "batch"])
def test_windows_are_capped_in_tiles(oracle, native, mode, monkeypatch):
    """the engine takes 2^22 tiles per window; the classifier caps its windows in reads —
    a small -t or very long reads must shrink the window, not abort the run (cap forced to 25 tiles here)."""
    from goldrush_amd import host
    from oracle_engine import OracleEngine, serial_reference

    for key, val in {"sync": {"GRP_PIPELINE": "off", "GRP_BATCH": "off"}, "pipeline": {"GRP_PIPELINE": "force", "GRP_BATCH": "off"}, "batch": {"GRP_BATCH": "force"}}[mode].items():
        monkeypatch.setenv(key, val)
    monkeypatch.setenv("GRP_MAX_WINDOW_TILES", "25")
    tile, k, h, block = 500, 22, 3, 4
    seeds = default_seeds(h)
    reads = _workload()
    m = oracle.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, tile, k, reads, block=block, silver=True, target_bases=90_000, max_paths=2)
    eng = OracleEngine(oracle, m, seeds, tile, k, reads, pipelined=(mode == "pipeline"), batching=(mode == "batch"))
    cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, target_bases=90_000, max_paths=2, silver_path=True, max_window=4096)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    cls.run(None, lens, skipped_before=np.zeros(len(reads), dtype=np.uint32))
    assert _strip(cls.commits) == exp
    assert 0 < eng.max_window_tiles <= max(25, max(len(r) // tile for r in reads))


def _hits_misses(oracle, m, seeds, tile, k, reads, block, silver=True):
    """hits / misses per committed read as a window-of-one classifier reports them"""
    from goldrush_amd import host
    from oracle_engine import OracleEngine

    eng = OracleEngine(oracle, m, seeds, tile, k, reads)
    if silver:
        cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=3, target_bases=90_000, max_paths=2, silver_path=True, max_window=1)
    else:
        cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=3, max_window=1)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    cls.run(None, lens, skipped_before=np.zeros(len(reads), dtype=np.uint32))
    return cls.commits


def test_skipped_reads_advance_counter(oracle, native):
    from goldrush_amd import host
    from oracle_engine import OracleEngine

    tile, k, h = 500, 22, 3
    seeds = default_seeds(h)
    reads = _workload(n=8)
    m = oracle.load().orc_calc_optimal_size(500_000, 1, 0.1)
    eng = OracleEngine(oracle, m, seeds, tile, k, reads)
    cls = host.Classifier(None, eng.vt, tile=tile, block=4, k=k, h=h, target_bases=10**9)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    cls.run(None, lens, skipped_before=np.array([2, 0, 0, 1, 0, 0, 0, 3], dtype=np.uint32), skipped_after=4)
    st = cls.state()
    assert st["id"] == 1 + 8 + 6 + 4            # uint32 id = 1 at start (goldrush_path.cpp:1225)
    assert st["valid_reads"] == 8


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import ctypes as C
import numpy as np
import torch, torch.distributed as dist
import orc
from goldrush_amd import host
from oracle_engine import OracleEngine, serial_reference
from helpers import default_seeds
from test_classifier_cpu import _workload, _strip

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
tile, k, h, block = 500, 22, 3, 4
seeds = default_seeds(h)
reads = _workload()
m = orc.load().orc_calc_optimal_size(1_500_000, 1, 0.1)
PIPE = os.environ.get("GRP_PIPELINE") == "force"
STREAM = os.environ.get("GRP_STREAM") == "force"
BATCH = os.environ.get("GRP_BATCH") == "force"   # windows committed as batches: every rank applies and checks the whole batch on its replica
# round 5: the ranks' striped windows apply inserts themselves.  RESUME = "plain": every launch takes every insert; "refuse":
# every third window of rank 1 is one that ends where it parks (a refused cooperative launch: the ranks must agree per
# window); "lost": every second insert reaches rank 1's launch too late (it had left: the ranks end the round together)
RESUME = os.environ.get("TEST_RESUME", "")
eng = OracleEngine(orc, m, seeds, tile, k, reads, pipelined=PIPE, streaming=STREAM, redo_every=11 if STREAM else 0, batching=BATCH, resume=bool(RESUME),
                   resume_refuse_every=3 if RESUME == "refuse" and rank == 1 else 0, resume_lost_every=2 if RESUME == "lost" and rank == 1 else 0)   # every rank holds a full replica


def allgather(user, send, nbytes, recv):
    src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
    dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
    t_in = torch.from_numpy(src.copy())
    t_out = torch.empty(nbytes * world, dtype=torch.uint8)
    dist.all_gather_into_tensor(t_out, t_in)
    dst[:] = t_out.numpy()
    return 0


cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=h, target_bases=90_000, max_paths=2, silver_path=True, max_window=40 if STREAM else (8 if PIPE else 16),
                      world=world, rank=rank, allgather=allgather)
lens = np.array([len(r) for r in reads], dtype=np.uint32)
cls.run(None, lens)
exp, mf_ref = serial_reference(orc, m, seeds, tile, k, reads, block=block, silver=True, target_bases=90_000, max_paths=2)
assert _strip(cls.commits) == exp, "rank %d: commits differ" % rank
assert np.array_equal(eng.mf.ids(), mf_ref.ids()) and np.array_equal(eng.mf.counts(), mf_ref.counts())
# the window really was sharded: each rank queried only part of the reads
tot = torch.tensor([eng.n_queries], dtype=torch.int64)
mine = int(tot.item())
dist.all_reduce(tot)
assert mine < int(tot.item())
assert not PIPE or (eng.n_begun >= 2 and eng.n_abandoned >= 1)
assert not STREAM or (eng.n_streams >= 3 and eng.n_stream_aborts >= 2 and eng.n_redo >= 1)
if RESUME:
    st = cls.state()
    ins = torch.tensor([eng.n_stream_inserts, st["stream_inserts"], st["stream_insert_fallbacks"]], dtype=torch.int64)
    dist.all_reduce(ins)
    assert int(ins[0]) >= 2 and int(ins[1]) >= 2, "no insert was applied inside the ranks' launches: %s" % ins.tolist()
    assert RESUME != "lost" or int(ins[2]) >= 1, "no launch was lost"
assert not BATCH or (eng.n_batches >= 2 and cls.state()["batch_reads"] > 0)  # both queries of a batch are striped over the ranks (GRP_BATCH_STRIPE_MIN=2)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", mine, int(tot.item()), eng.n_begun, eng.n_abandoned)
"""


@pytest.mark.parametrize("pipeline", ["off", "force", "stream", "batch", "stream_resume", "stream_resume_refuse", "stream_resume_lost"])
def test_two_ranks_gloo(oracle, native, tmp_path, pipeline):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=root))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2",
                   **({"GRP_STREAM": "force", "GRP_STRIPE": "3", "GRP_BATCH": "off", "TEST_RESUME": pipeline[14:] or "plain"} if pipeline.startswith("stream_resume") else
                      {"GRP_STREAM": "force", "GRP_STRIPE": "3"} if pipeline == "stream" else {"GRP_BATCH": "force", "GRP_STREAM": "off", "GRP_BATCH_STRIPE_MIN": "2"} if pipeline == "batch"
                      else {"GRP_PIPELINE": pipeline, "GRP_STREAM": "off"}))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
