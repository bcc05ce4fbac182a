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


def _late_worker(rank, world, key, delay, q):
    import time

    sys.path.insert(0, ROOT)
    from goldrush_amd import host

    time.sleep(delay)
    hl = host.load()
    h = hl.gr_shm_allgather_open(world, rank, key.encode(), 30.0)
    ok = bool(h)
    got = None
    if ok:
        src = np.full(64, 0x10 + rank, dtype=np.uint8)
        dst = np.zeros(64 * world, dtype=np.uint8)
        ok = hl.gr_shm_allgather(h, src.ctypes.data_as(C.c_void_p), 64, dst.ctypes.data_as(C.c_void_p)) == 0
        got = [int(dst[p * 64]) for p in range(world)]
        hl.gr_shm_allgather_close(h)
    q.put((rank, ok, got))


def test_stale_file_of_a_crashed_run_is_never_joined():
    """a run killed under `timeout` leaves /dev/shm/grp_<key> behind with its header
    complete (every rank attached, round counters high, 0xEE payload).  The next run uses the same
    key; rank 1 comes first and finds the stale file — it must end up in rank 0's NEW file."""
    world = 2
    key = "stale_%d" % os.getpid()
    path = "/dev/shm/grp_" + key
    size = 64 * 1024 + 2 * world * (1 << 20)
    buf = np.full(size, 0xEE, dtype=np.uint8)
    hdr = buf[: 64 * 1024].view(np.uint64)
    hdr[:] = 0
    hdr[0] = 0x47525053484d3032  # the header of a finished hand-shake
    for r in range(world):
        hdr[8 + r * 8 + 0] = 1000  # round
        hdr[8 + r * 8 + 1] = 0x1234 + r  # token
        hdr[8 + r * 8 + 3] = 0x1234 + r  # echo
    buf.tofile(path)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_late_worker, args=(1, world, key, 0.0, q)), ctx.Process(target=_late_worker, args=(0, world, key, 1.5, q))]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=30)
        assert res == [(0, True, [0x10, 0x11]), (1, True, [0x10, 0x11])], res
    finally:
        if os.path.exists(path):
            os.unlink(path)


def _leaver(rank, world, key, q):
    sys.path.insert(0, ROOT)
    from goldrush_amd import host

    hl = host.load()
    h = hl.gr_shm_allgather_open(world, rank, key.encode(), 30.0)
    rc = None
    if h:
        src = np.zeros(32, dtype=np.uint8)
        dst = np.zeros(32 * world, dtype=np.uint8)
        rc = hl.gr_shm_allgather(h, src.ctypes.data_as(C.c_void_p), 32, dst.ctypes.data_as(C.c_void_p))
        if rank == 0:  # the second round: rank 1 has left (an error on its side)
            rc = hl.gr_shm_allgather(h, src.ctypes.data_as(C.c_void_p), 32, dst.ctypes.data_as(C.c_void_p))
        hl.gr_shm_allgather_close(h)
    q.put((rank, rc))


def test_a_peer_that_left_is_reported_at_once():
    """a rank waiting for a peer that has closed its handle gets an error (-3), not a
    spin of hours."""
    import time

    world = 2
    key = "leave_%d" % os.getpid()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_leaver, args=(r, world, key, q)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert res == {0: -3, 1: 0}, res
    assert time.time() - t0 < 60


def _bench(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env)


def test_bench_refuses_a_world_that_is_not_its_gpus_flag():
    """`--gpus N` is the number of ranks: a launcher that started another number is an error, not a silent N = WORLD_SIZE
    (or N = 1) line (VERDICT r05 missing #2).  Decided before torch is imported: runs without a GPU."""
    r = _bench(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr and not r.stdout.strip()
    r = _bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    r = _bench(["--gpus", "0"])
    assert r.returncode != 0


def test_bench_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE around it starts two rank processes itself (torch.distributed.run on
    127.0.0.1) and hands their exit code on.  Here there is no GPU: both ranks must fail loudly ("no HIP device", no CPU
    fallback) — and the parent with them."""
    r = _bench(["--gpus", "2", "--config", "C1", "--reads", "2000", "--steps", "2", "--warmup", "0"])
    assert "starting 2 ranks" in r.stderr, r.stderr[-1500:]
    import torch

    if not torch.cuda.is_available():
        assert r.returncode != 0 and "no HIP device visible" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
