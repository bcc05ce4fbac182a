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


@pytest.mark.parametrize("max_window,crowded", [(2, 0), (5, 0), (64, 0), (4096, 0), (64, 3)])
def test_batched_windows_match_serial_loop(oracle, native, max_window, crowded, monkeypatch):
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
    eng = OracleEngine(oracle, m, seeds, tile, k, reads, batching=True, batch_crowded_above=crowded)
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
    if max_window >= 64 and not crowded:
        assert eng.n_batch_undone >= 1 and st["batches_undone"] == eng.n_batch_undone
    if crowded:
        assert eng.n_batch_refused >= 1 and st["batches_refused"] == eng.n_batch_refused


def _hits_misses(oracle, m, seeds, tile, k, reads, block):
    """hits / misses per committed read as a window-of-one classifier reports them"""
    from goldrush_amd import host
    from oracle_engine import OracleEngine

    eng = OracleEngine(oracle, m, seeds, tile, k, reads)
    cls = host.Classifier(None, eng.vt, tile=tile, block=block, k=k, h=3, target_bases=90_000, max_paths=2, silver_path=True, max_window=1)
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
eng = OracleEngine(orc, m, seeds, tile, k, reads, pipelined=PIPE, streaming=STREAM, redo_every=11 if STREAM else 0, batching=BATCH)   # every rank holds a full replica


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
assert not BATCH or (eng.n_batches >= 2 and cls.state()["batch_reads"] > 0)  # both queries of a batch are striped over the ranks (GRP_BATCH_STRIPE_MIN=2)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", mine, int(tot.item()), eng.n_begun, eng.n_abandoned)
"""


@pytest.mark.parametrize("pipeline", ["off", "force", "stream", "batch"])
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
                   **({"GRP_STREAM": "force", "GRP_STRIPE": "3"} if pipeline == "stream" else {"GRP_BATCH": "force", "GRP_STREAM": "off", "GRP_BATCH_STRIPE_MIN": "2"} if pipeline == "batch"
                      else {"GRP_PIPELINE": pipeline, "GRP_STREAM": "off"}))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
