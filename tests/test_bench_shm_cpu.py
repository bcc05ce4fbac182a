"""CPU: the node-local shared-memory all-gather of libgrpath_host (gr_shm_allgather_*: what
the ranks of a multi-GPU classification exchange their decision records through) — three
processes, many rounds of varying size, every rank must see every block of every round."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, key, rounds, q):
    sys.path.insert(0, ROOT)
    from goldrush_amd import host

    hl = host.load()
    h = hl.gr_shm_allgather_open(world, rank, key.encode(), 60.0)
    ok = bool(h)
    rng = np.random.default_rng(5)  # the same sizes on every rank
    for r in range(rounds if ok else 0):
        n = int(rng.integers(1, 5000)) * 32
        src = np.full(n, (rank * 37 + r) & 0xFF, dtype=np.uint8)
        src[: 8] = np.frombuffer(np.uint64(r * world + rank).tobytes(), dtype=np.uint8)
        dst = np.zeros(n * world, dtype=np.uint8)
        ok = ok and hl.gr_shm_allgather(h, src.ctypes.data_as(C.c_void_p), n, dst.ctypes.data_as(C.c_void_p)) == 0
        for p in range(world):
            blk = dst[p * n:(p + 1) * n]
            ok = ok and int(blk[:8].view(np.uint64)[0]) == r * world + p and bool(np.all(blk[8:] == ((p * 37 + r) & 0xFF)))
    if h:
        hl.gr_shm_allgather_close(h)
    q.put((rank, ok))


def test_shared_memory_allgather_three_ranks():
    world, rounds = 3, 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = "test_%d" % os.getpid()
    procs = [ctx.Process(target=_worker, args=(r, world, key, rounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert res == [(r, True) for r in range(world)]
    assert not os.path.exists("/dev/shm/grp_" + key)
