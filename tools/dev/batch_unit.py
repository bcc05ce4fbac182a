"""GPU box: at the geometry of the full-size tests (G = 8e6, 25 kb reads, tile 1000, block 10) —
(1) a batch of whole-read inserts against the same inserts one by one: IDs / counts of every rank;
(2) the second decisions of the batch (through the view) against decisions taken between serial inserts."""
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from goldrush_amd import host, native  # noqa: E402
from helpers import default_seeds  # noqa: E402

h, n, nb = 3, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
serial_dec = []
for j, (ri, ts, te, fid, off) in enumerate(ins):
    serial_dec.append(a.classify_reads(ra, j, 1)[0])
    a.insert_read(ra, ri, ts, te, block, fid, off)
ids_a, counts_a = a.export_ids()

b, rbb = engine()
b.batch_insert_reads(rbb, ins, block, 0)
d1 = b.batch_classify(rbb, 0, nb, floors)
b.batch_end()
ids_b, counts_b = b.export_ids()
print("state: ids differ at", int((ids_a != ids_b).sum()), "counts differ at", int((counts_a != counts_b).sum()), "of", ids_a.size)
dd = np.nonzero((ids_a != ids_b) | (counts_a != counts_b))[0]
print("floors", floors, "tiles", [int(l) // tile for l in lens[:nb]])
print("serial IDs at differing ranks:", dict(zip(*[x.tolist() for x in np.unique(ids_a[dd], return_counts=True)])))
print("batch IDs at differing ranks:", dict(zip(*[x.tolist() for x in np.unique(ids_b[dd], return_counts=True)])))
print("serial counts at differing ranks:", dict(zip(*[x.tolist() for x in np.unique(counts_a[dd], return_counts=True)])))
print("batch counts at differing ranks:", dict(zip(*[x.tolist() for x in np.unique(counts_b[dd], return_counts=True)])))
print("IDs overall serial:", dict(zip(*[x.tolist() for x in np.unique(ids_a, return_counts=True)])))
print("IDs overall batch:", dict(zip(*[x.tolist() for x in np.unique(ids_b, return_counts=True)])))
for r in dd[:10]:
    print("  rank", r, "serial id/count", ids_a[r], counts_a[r], "batch", ids_b[r], counts_b[r])
# the oracle's state after the same inserts
sys.path.insert(0, os.path.join(root, "oracle"))
import orc  # noqa: E402
seqs = dr.download(0, n)
oseeds = orc.Seeds(seeds)
mf = orc.MiBF(m, oseeds, tile, k)
mf.bv_insert_reads(seqs)
mf.finalize()
for (ri, ts, te, fid, off) in ins:
    bs = ts
    while bs < te:
        mf.insert_read_tiles(seqs[ri], bs, min(bs + block, te), fid + (bs - ts + off) // block)
        bs += block
ids_o, counts_o = mf.ids().copy(), mf.counts().copy()
print("oracle vs serial: ids differ", int((ids_o != ids_a).sum()), "counts differ", int((counts_o != counts_a).sum()))
print("oracle vs batch : ids differ", int((ids_o != ids_b).sum()), "counts differ", int((counts_o != counts_b).sum()))

# per tile of every inserted read: the count of the tile's top ID in both end states
for j in range(nb):
    ta, _, _ = a.query_tiles(ra, j, 1)
    tb_, _, _ = b.query_tiles(rbb, j, 1)
    names = ta.dtype.names
    ca = [int(t[names[1]]) for t in ta]
    cb = [int(t[names[1]]) for t in tb_]
    print("read", j, "hits serial/batch", int(ta["hits"].sum()), int(tb_["hits"].sum()), "top", ca[:6], cb[:6])
    print("read", j, "per-tile hits lost", [int(x) - int(y) for x, y in zip(ta["hits"], tb_["hits"])])
    if ca != cb:
        print("read", j, "top counts serial", ca)
        print("read", j, "top counts batch ", cb)
print("tile fields", ta.dtype.names)
for j in range(nb):
    s, d = serial_dec[j], d1[j]
    same = all(int(s[f]) == int(d[f]) for f in ("kind", "num_tiles", "num_assigned", "trim_start", "trim_end", "hits", "misses"))
    print("read", j, "same" if same else "DIFF", [int(s[f]) for f in ("kind", "num_assigned", "hits", "misses")], [int(d[f]) for f in ("kind", "num_assigned", "hits", "misses")])

# (3) take the batch back from read nb // 2 on: must equal the serial state after the first nb // 2 inserts
c, rc_ = engine()
half = nb // 2
c.batch_insert_reads(rc_, ins, block, 0)
c.batch_classify(rc_, 0, nb, floors)
c.batch_undo(half, floors[half])
ids_c, counts_c = c.export_ids()
e, re_ = engine()
for (ri, ts, te, fid, off) in ins[:half]:
    e.insert_read(re_, ri, ts, te, block, fid, off)
ids_e, counts_e = e.export_ids()
print("partial undo: ids differ at", int((ids_c != ids_e).sum()), "counts differ at", int((counts_c != counts_e).sum()))
