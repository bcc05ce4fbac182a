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
if os.environ.get("GRP_BATCH_DUMP"):
    raw = np.fromfile(os.environ["GRP_BATCH_DUMP"], dtype=np.uint64)
    nrec = raw.size // 2
    key, old = raw[:nrec], raw[nrec:]
    rank = (key >> np.uint64(26)).astype(np.int64) - 1
    has = (key >> np.uint64(26)) != 0
    print("non-record codes:", dict(zip(*[x.tolist() for x in np.unique(key[~has], return_counts=True)])))
    un = np.arange(nrec) // 768
    if os.environ.get("GRP_BATCH_DEBUG") == "5":
        mk = old
        zk = key == 0
        print("marker present overall:", int(((mk >> np.uint64(56)) == 0xAB).sum()), "of", nrec, "; on claimed:", int(((mk >> np.uint64(56)) == 0xAB)[~zk].sum()), "sample raw zero-key markers", [hex(int(x)) for x in mk[zk][:4]])
        print("marker present on zero-key records:", int(((mk >> np.uint64(56)) == 0xAB)[zk].sum()), "of", int(zk.sum()))
        zs2 = zk & ((np.arange(nrec) // 256) % 3 == 2) & (un >= 256)
        m2 = mk[zs2]
        print("seed-2 units>=256 zero-key: marker present", int(((m2 >> np.uint64(56)) == 0xAB).sum()), "valid", int(((m2 >> np.uint64(40)) & np.uint64(1)).sum()), "live", int(((m2 >> np.uint64(41)) & np.uint64(1)).sum()), "claimed", int(((m2 >> np.uint64(42)) & np.uint64(1)).sum()))
    z = np.nonzero(key == 0)[0]
    print("zero records per unit 250..290:", [int((un[z] == u).sum()) for u in range(250, 290)])
    zz = z[un[z] >= 256]
    print("zero records (units >= 256) by seed:", np.bincount((zz % 768) // 256, minlength=3), "by wave:", np.bincount((zz % 256) // 64, minlength=4), "by lane%64 head:", np.bincount(zz % 64, minlength=64)[:16])
    print("zero records with tid >= 232 in last part / total:", int((((z % 256) >= 232) & ((un[z] % 4) == 3)).sum()), z.size)
    print("code 1 (no set bit) per unit, units 250..275:", [int(((key == 1) & (un == u)).sum()) for u in range(250, 276)])
    print("code 2 per unit 250..275:", [int(((key == 2) & (un == u)).sum()) for u in range(250, 276)])
    print("code 3 per unit 250..275:", [int(((key == 3) & (un == u)).sum()) for u in range(250, 276)])
    lost = np.nonzero((counts_o != counts_b))[0]
    rec_of = {}
    idxs = np.nonzero(has)[0]
    order = np.argsort(rank[idxs])
    sr = rank[idxs][order]
    pos = np.searchsorted(sr, lost)
    found = (pos < sr.size) & (sr[np.minimum(pos, sr.size - 1)] == lost)
    print("records", nrec, "claimed", int(has.sum()), "lost ranks", lost.size, "of which have a record:", int(found.sum()))
    fi = idxs[order][pos[found]]
    print("  their record index // 768 (unit) histogram head:", np.unique(fi // 768, return_counts=True)[0][:20], "lane%256 sample", (fi % 256)[:20], "seed", ((fi // 256) % 3)[:20])
    print("  rec_old of found: wrote", int((old[fi] >> np.uint64(63)).sum()), "count0 nonzero", int((((old[fi] >> np.uint64(32)) & np.uint64(0x7FFFFFFF)) != 0).sum()))
    # duplicates: the same rank claimed twice
    u, cnt = np.unique(rank[idxs], return_counts=True)
    print("  ranks with more than one record:", int((cnt > 1).sum()))
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
