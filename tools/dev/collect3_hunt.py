#!/usr/bin/env python3
"""GPU box, developer: the lost-touch fault of round 2's three-seeds-per-lane collect kernel
 — one large batch at the benchmark geometry against the oracle, for the product's
one-seed form and the resurrected three-seed form in its variants (GRP_BATCH_COLLECT3 = 1 + 2 x
variant: bit 0 agent-scope owner touch, bit 1 run-time care loop); which records / seeds / counts
differ, several runs each."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import orc as oracle
from goldrush_amd import host, native
from helpers import default_seeds
oracle.build(); oracle.load()
hl = host.load()
k, h, tile, block = 22, 3, 1000, 10
G = float(os.environ.get("HUNT_G", "3e9"))
m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, int(G), h), 1, 0.1)
seeds = default_seeds(h)
n_reads = int(os.environ.get("HUNT_READS", "320"))
dr = native.synth_reads(n_reads, 3_000_000, mean_len=25000, min_len=20000, seed=19)
eng = native.Engine(k, h, tile, m, seeds)
batch = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
eng.bv_insert(batch)
reads = dr.download(0, n_reads)
omf = oracle.MiBF(m, oracle.Seeds(seeds), tile, k)
omf.bv_insert_reads(reads)
assert eng.finalize() == omf.finalize()
ins, next_id = [], 0
for ri, seq in enumerate(reads):
    nt = len(seq) // tile
    ins.append((ri, 0, nt, next_id + 1, 0))
    for bs in range(0, nt, block):
        omf.insert_read_tiles(seq, bs, min(bs + block, nt), next_id + 1 + bs // block)
    next_id += 1 + len(seq) // (tile * block)
eng.batch_insert_reads(batch, ins, block, 0)
eng.batch_end()
ids, counts = eng.export_ids()
oi, oc = omf.ids(), omf.counts()
bad_c = np.flatnonzero(counts != oc)
bad_i = np.flatnonzero(ids != oi)
print("mode", os.environ.get("GRP_BATCH_COLLECT3", "product"), "ranks touched", int((oc > 0).sum()), "count differs", bad_c.size, "id differs", bad_i.size,
      "sum(count) gpu", int(counts.sum()), "oracle", int(oc.sum()), "lower", int((counts < oc).sum()), "higher", int((counts > oc).sum()))
if bad_c.size:
    d = (oc[bad_c].astype(np.int64) - counts[bad_c].astype(np.int64))
    print("  count deficit histogram", np.bincount(np.clip(d, -1, 5) + 1)[:8].tolist(), "first ranks", bad_c[:8].tolist())
    if os.environ.get("HUNT_WHO"):
        # which (tile part, lane, seed) probes land on the ranks that lost a touch: all probes of the first reads
        oseeds = oracle.Seeds(seeds)
        badset = np.zeros(oc.size, dtype=bool)
        badset[bad_c[counts[bad_c] < oc[bad_c]]] = True
        hist = np.zeros((4, 256, h), dtype=np.int64)   # [part][lane][seed]
        tot = np.zeros((4, 256, h), dtype=np.int64)
        for ri in range(24):
            seq = reads[ri]
            for t in range(len(seq) // tile):
                hv = oseeds.tile_hashes(seq, tile, k, t).reshape(-1, h)   # [frame][seed]
                pos = (hv %% np.uint64(m)).astype(np.uint64)
                bit, rank = eng.rank(pos.ravel())
                rank = rank.reshape(-1, h)
                fr = np.arange(rank.shape[0])
                for s2 in range(h):
                    hit = badset[rank[:, s2]]
                    np.add.at(hist, (fr // 256, fr %% 256, s2), hit)
                    np.add.at(tot, (fr // 256, fr %% 256, s2), 1)
        frac = hist / np.maximum(tot, 1)
        print("  share of probes on ranks that lost a touch, by seed:", [round(float(hist[:, :, s2].sum() / tot[:, :, s2].sum()), 4) for s2 in range(h)])
        print("  by tile part:", [round(float(hist[p2].sum() / max(tot[p2].sum(), 1)), 4) for p2 in range(4)])
        lane = hist.sum(axis=(0, 2)) / np.maximum(tot.sum(axis=(0, 2)), 1)
        print("  by lane (top 12):", sorted(((round(float(v), 4), i) for i, v in enumerate(lane)), reverse=True)[:12])
        print("  by wave:", [round(float(lane[w * 64:(w + 1) * 64].mean()), 4) for w in range(4)])
'''


def main():
    modes = sys.argv[1:] or ["product", "1", "1", "3", "5", "7"]
    for mode in modes:
        env = dict(os.environ)
        env.pop("GRP_BATCH_COLLECT3", None)
        if mode != "product":
            env["GRP_BATCH_COLLECT3"] = mode
        r = subprocess.run([sys.executable, "-c", CODE % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=1200)
        print(r.stdout.strip() or r.stderr[-800:])
        sys.stdout.flush()


if __name__ == "__main__":
    main()
