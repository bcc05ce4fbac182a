import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from goldrush_amd import native, host

def run(use_dev, n_reads=3000, G=2_000_000):
    k, h, tile = 22, 3, 1000
    seeds = host.make_seed_pattern("1011011110110111101101", k, 16, h)
    hl = host.load()
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, G, h), 1, 0.1)
    dr = native.synth_reads(n_reads, G)
    eng = native.Engine(k, h, tile, m, seeds)
    rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    eng.bv_insert(rb); eng.finalize()
    vt = host.hip_engine_vt()
    if not use_dev:
        vt.classify_reads = C.cast(None, dict(host.VT_TYPES)["classify_reads"])
    cls = host.Classifier(eng._h, vt, tile=tile, block=10, k=k, h=h, target_bases=10**15, max_paths=1)
    cls.run(rb._h, dr.lens)
    return cls.commits, eng, rb, dr

a, *_ = run(False)
b, eng, rb, dr = run(True)
print(len(a), len(b), sum(1 for x in a if x[1] in (2,4)), sum(1 for x in b if x[1] in (2,4)))
for i,(x,y) in enumerate(zip(a,b)):
    if x != y:
        print("first divergence at commit", i, x, y)
        break
else:
    print("identical")

# replay: host-decide classifier for reads [0,22), then compare both paths on read 22
k, h, tile = 22, 3, 1000
seeds = host.make_seed_pattern("1011011110110111101101", k, 16, h)
hl = host.load()
G = 2_000_000
m = hl.gr_calc_optimal_size(hl.gr_hash_universe(16, G, h), 1, 0.1)
dr = native.synth_reads(3000, G)
eng = native.Engine(k, h, tile, m, seeds)
rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
eng.bv_insert(rb); eng.finalize()
vt = host.hip_engine_vt()
vt.classify_reads = C.cast(None, dict(host.VT_TYPES)["classify_reads"])
cls = host.Classifier(eng._h, vt, tile=tile, block=10, k=k, h=h, target_bases=10**15, max_paths=1)
lens = np.ascontiguousarray(dr.lens, dtype=np.uint32)
cls.run_range(rb._h, lens, 0, 22)
eng.sync()
for first, cnt in ((22, 1), (20, 5), (0, 40)):
    dec = eng.classify_reads(rb, first, cnt)
    tiles, lists, _ = eng.query_tiles(rb, first, cnt)
    la = np.ascontiguousarray(lists) if len(lists) else np.zeros(1, dtype=native.id_count_dtype)
    t0 = int(rb.tile0[first])
    for j in range(cnt):
        a, e = int(rb.tile0[first + j]) - t0, int(rb.tile0[first + j + 1]) - t0
        d = host.decide_read(np.ascontiguousarray(tiles[a:e]), la, e - a)
        g = dec[j]
        if (int(g["kind"]), int(g["num_assigned"])) != (d.kind, d.num_assigned):
            print("MISMATCH window", first, cnt, "read", first + j, "dev", g, "host", (d.kind, d.num_assigned, d.trim_start, d.trim_end))
            print(" tiles list_n", [int(x) for x in tiles[a:e]["list_n"]], "top", [int(x) for x in tiles[a:e]["top_count"]])
            break
    else:
        print("window", first, cnt, "ok")
