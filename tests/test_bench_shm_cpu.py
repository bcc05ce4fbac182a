"""CPU: the node-local shared-memory all-gather bench.py uses for the decision records
(three processes, many rounds of varying size, every rank must see every block of every
round exactly)."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, key, barrier, rounds, q):
    sys.path.insert(0, ROOT)
    import bench

    ag = bench.ShmAllgather(world, rank, key, barrier.wait)
    rng = np.random.default_rng(5)  # the same sizes on every rank
    ok = True
    for r in range(rounds):
        n = int(rng.integers(1, 5000)) * 32
        src = np.full(n, (rank * 37 + r) & 0xFF, dtype=np.uint8)
        src[: 8] = np.frombuffer(np.uint64(r * world + rank).tobytes(), dtype=np.uint8)
        dst = np.zeros(n * world, dtype=np.uint8)
        ag(src, dst)
        for p in range(world):
            blk = dst[p * n:(p + 1) * n]
            ok = ok and int(blk[:8].view(np.uint64)[0]) == r * world + p and bool(np.all(blk[8:] == ((p * 37 + r) & 0xFF)))
    ag.close(barrier.wait)
    q.put((rank, ok))


def test_shared_memory_allgather_three_ranks():
    world, rounds = 3, 400
    ctx = mp.get_context("fork")
    barrier, q = ctx.Barrier(world), ctx.Queue()
    key = "test_%d" % os.getpid()
    procs = [ctx.Process(target=_worker, args=(r, world, key, barrier, rounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert res == [(r, True) for r in range(world)]
    assert not os.path.exists("/dev/shm/grp_bench_%s" % key)
