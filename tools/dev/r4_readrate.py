#!/usr/bin/env python3
"""Developer: how fast can N threads copy a file out of the page cache on this host (pread into one buffer)?"""
import os, sys, threading, time
import numpy as np
path, size = "/tmp/readrate.bin", 6 << 30
with open(path, "wb") as f:
    blk = np.random.default_rng(1).integers(0, 255, 64 << 20, dtype=np.uint8).tobytes()
    for _ in range(size // len(blk)):
        f.write(blk)
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
fd = os.open(path, os.O_RDONLY)
buf = np.empty(512 << 20, dtype=np.uint8)
mv = memoryview(buf)
def worker(lo, hi, off):
    pos = lo
    while pos < hi:
        n = os.preadv(fd, [mv[pos:min(hi, pos + (16 << 20))]], off + pos)
        if n <= 0:
            break
        pos += n
for threads in (1, 2, 4, 8, 16, 32):
    t0 = time.time(); done = 0
    for chunk in range(0, size, len(buf)):
        per = len(buf) // threads
        ts = [threading.Thread(target=worker, args=(i * per, (i + 1) * per, chunk)) for i in range(threads)]
        [t.start() for t in ts]; [t.join() for t in ts]
        done += len(buf)
    dt = time.time() - t0
    print("%2d threads: %.1f GB/s" % (threads, done / dt / 1e9), flush=True)
os.remove(path)
