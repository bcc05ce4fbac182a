"""GPU: what a streaming window keeps across an in-launch insert (round 6).  Every workgroup holds the last two
tiles it has finished against the slots the insert changed: clean ones stand, dirty ones are queried again.  The stream of
tests/stream_keep_scenario.py puts clusters of overlapping reads of an uncovered island into a covered genome — the launch has
queried the reads behind a cluster's first read BEFORE that read's insert, and they decide differently behind it — and is
checked against the oracle's serial loop (process_read, goldrush_path.cpp:892-1094) and across the three forms of
GRP_STREAM_KEEP (0: everything behind the read is queried again, rounds 3 - 5's form)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import default_seeds

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_kept_tiles_stand_dirty_ones_are_queried_again(oracle, native):
    import stream_keep_scenario as sc
    from oracle_engine import serial_reference

    seeds = default_seeds(sc.H)
    reads = sc.make_stream()
    m = oracle.load().orc_calc_optimal_size(2_500_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, sc.TILE, sc.K, reads, block=sc.BLOCK)
    r = sc.run(native, seeds, m, reads)
    assert r["pop"] == mf_ref.pop
    assert [t[:8] for t in r["records"]] == exp
    assert np.array_equal(r["ids"], mf_ref.ids()) and np.array_equal(r["counts"], mf_ref.counts())
    st = r["stats"]
    kinds = [t[1] for t in r["records"]]
    assert r["inserts"] >= 30 and sum(1 for q in kinds[70:] if q not in (2, 4)) >= 150, "the stream is not what the test means: a head of inserts, then mostly reads that do not insert"
    assert st["inserts_kept"] + st["inserts_kept_nothing"] == r["inserts"]
    assert st["inserts_kept"] > 0 and st["tiles_kept"] > 0, st               # the launch runs hundreds of tiles ahead of the read that inserts: it holds them
    assert st["tiles_redone_dirty"] > 0, st                                   # the reads behind a cluster's first read overlap it: their probes read slots it changed
    assert st["coop_refused"] == 0


def _child(keep):
    env = dict(os.environ, GRP_STREAM_KEEP=str(keep))
    r = subprocess.run([sys.executable, os.path.join(HERE, "stream_keep_scenario.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_the_three_forms_of_keeping_decide_alike():
    """GRP_STREAM_KEEP = 0 (nothing kept), 1 (the last finished tile), 2 (the last two, the default): the same records, the
    same ID and count arrays; only what was queried twice differs."""
    got = {k: _child(k) for k in (0, 1, 2)}
    for k in (0, 1):
        assert got[k]["records"] == got[2]["records"], "GRP_STREAM_KEEP=%d decides differently" % k
        assert got[k]["arrays"] == got[2]["arrays"] and got[k]["inserts"] == got[2]["inserts"]
    assert got[0]["stats"]["tiles_kept"] == 0 and got[0]["stats"]["inserts_kept"] == 0 and got[0]["stats"]["inserts_kept_nothing"] == got[0]["inserts"]
    assert got[2]["stats"]["tiles_kept"] > 0 and got[1]["stats"]["tiles_kept"] > 0
    assert got[2]["stats"]["tiles_kept"] >= got[1]["stats"]["tiles_kept"]


def test_a_device_shared_with_other_kernels_times_no_insert_out(oracle, native):
    """The waits of an in-launch insert are for the workgroups that have BEGUN (stream_register), not for the grid: while
    another stream keeps the CUs busy (large GEMMs through torch), the window's workgroups become resident a few at a time —
    some of them in the middle of an insert.  Rounds 3 - 5 waited for the grid: such an insert timed out (2 s) and went back
    to the host.  Here a timed-out wait ends the launch, and the scenario fails on the record that never comes."""
    import threading

    torch = pytest.importorskip("torch")
    import stream_keep_scenario as sc
    from oracle_engine import serial_reference

    seeds = default_seeds(sc.H)
    reads = sc.make_stream()
    m = oracle.load().orc_calc_optimal_size(2_500_000, 1, 0.1)
    exp, mf_ref = serial_reference(oracle, m, seeds, sc.TILE, sc.K, reads, block=sc.BLOCK)
    stop = threading.Event()
    launched = [0]

    def hammer():
        torch.cuda.set_device(0)
        side = torch.cuda.Stream()
        a = torch.randn(6144, 6144, device="cuda", dtype=torch.float16)
        with torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(3):
                    a @ a
                    launched[0] += 1
                side.synchronize()

    t = threading.Thread(target=hammer, daemon=True)
    t.start()
    try:
        r = sc.run(native, seeds, m, reads, limit=60.0)
    finally:
        stop.set()
        t.join(timeout=120)
    assert not t.is_alive()
    assert launched[0] > 0
    assert [q[:8] for q in r["records"]] == exp
    assert np.array_equal(r["ids"], mf_ref.ids()) and np.array_equal(r["counts"], mf_ref.counts())
    assert r["stats"]["inserts_kept"] + r["stats"]["inserts_kept_nothing"] == r["inserts"]
