import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from goldrush_amd import native, host
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
tiles, lists, _ = eng.query_tiles(rb, 22, 1)
np.save("gpurun_out/read22_tiles.npy", tiles); np.save("gpurun_out/read22_lists.npy", lists)
print(len(tiles), len(lists))
dec = eng.classify_reads(rb, 22, 1)
print("dev decision", dec[0])
ids, asg = eng.tile_states(26)
print("dev ids", list(map(int, ids))); print("dev asg", list(map(int, asg)))
