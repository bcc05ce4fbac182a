"""GPU box: batches + streaming windows against classic synchronous windows on the workload of
tests/test_gpu_classifier.py::test_full_size_streaming_equals_synchronous_windows — where do they differ first?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from goldrush_amd import host, native  # noqa: E402
from helpers import default_seeds  # noqa: E402

h, n = int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 40000
k, tile, block, G = 22, 1000, 10, 8_000_000
seeds = default_seeds(h)
hl = host.load()
m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, G, h), 1, 0.1)
dr = native.synth_reads(n, G)
lens = np.ascontiguousarray(dr.lens, dtype=np.uint32)
res = []
extra = dict(kv.split("=") for kv in sys.argv[3:])
print("extra", extra)
for mode in (dict({"GRP_STREAM": "force"}, **extra), {"GRP_BATCH": "off", "GRP_STREAM": "off", "GRP_PIPELINE": "off"}):
    for key in ("GRP_STREAM", "GRP_PIPELINE", "GRP_BATCH", "GRP_BATCH_READS"):
        os.environ.pop(key, None)
    os.environ.update(mode)
    eng = native.Engine(k, h, tile, m, seeds)
    rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    eng.bv_insert(rb)
    eng.finalize()
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, k=k, h=h, target_bases=int(0.9 * G), max_paths=1, silver_path=False)
    for first in range(0, n, 8192):
        cls.run_range(rb._h, lens, first, min(8192, n - first))
    eng.sync()
    st = cls.state()
    ids, counts = eng.export_ids()
    res.append((list(cls.commits), ids.copy(), counts.copy(), st))
    cls.close()
    eng.close()
a, b = res
print("states", {k_: a[3][k_] for k_ in ("windows", "inserts", "batches", "batches_undone", "batch_reads")}, {k_: b[3][k_] for k_ in ("windows", "inserts")})
first = next((i for i in range(min(len(a[0]), len(b[0]))) if a[0][i][:8] != b[0][i][:8]), None)
print("first differing commit", first)
if first is not None:
    for i in range(max(0, first - 3), min(len(a[0]), first + 3)):
        print(i, a[0][i], b[0][i])
hm = next((i for i in range(min(len(a[0]), len(b[0]))) if a[0][i] != b[0][i]), None)
print("first commit differing in hits/misses too", hm, a[0][hm] if hm is not None else None, b[0][hm] if hm is not None else None)
d = np.nonzero(a[1] != b[1])[0]
c = np.nonzero(a[2] != b[2])[0]
print("ids differ at", d.size, "ranks; counts differ at", c.size, "ranks")
for r in d[:10]:
    print(" rank", r, "ids", a[1][r], b[1][r], "counts", a[2][r], b[2][r])
for r in c[:10]:
    print(" rank", r, "counts", a[2][r], b[2][r], "ids", a[1][r], b[1][r])
