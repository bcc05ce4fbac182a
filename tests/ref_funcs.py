"""Test infrastructure: ctypes view of oracle/_ref/libref_funcs.so — the reference's own
find_longest_stretch / eval_flanks / smoothing passes / calcOptimalSize / hash-universe lines,
compiled from /root/reference at build time (oracle/Makefile `ref`, oracle/extract_ref_funcs.py)
— and the random tile states the fixtures and the live comparison are drawn from."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_funcs.so")


class RefFuncs:
    def __init__(self):
        lib = C.CDLL(LIB)
        vp = C.c_void_p
        lib.ref_find_longest_stretch.argtypes = [vp, C.c_size_t, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        lib.ref_eval_flanks.restype = C.c_int
        lib.ref_eval_flanks.argtypes = [C.c_long, C.c_long, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        lib.ref_smooth_tiles.restype = C.c_size_t
        lib.ref_smooth_tiles.argtypes = [C.c_size_t, vp, vp, vp, vp, vp, C.c_size_t]
        lib.ref_calc_optimal_size.restype = C.c_uint64
        lib.ref_calc_optimal_size.argtypes = [C.c_uint64, C.c_uint, C.c_double]
        lib.ref_hash_universe.restype = C.c_uint64
        lib.ref_hash_universe.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        lib.ref_vote_tile.restype = C.c_uint32
        lib.ref_vote_tile.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_uint32), vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        self.lib = lib

    def vote_tile(self, frames):
        """The reference's own vote statements on the frames' IDs -> (id, count, [(id, count) ...] in ITS order)."""
        off = np.zeros(len(frames) + 1, dtype=np.uint64)
        for i, f in enumerate(frames):
            off[i + 1] = off[i] + len(f)
        ids = np.array([x for f in frames for x in f] + [0], dtype=np.uint32)
        cap = max(int(off[-1]), 1)
        li, lc = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32)
        tc, n = C.c_uint32(), C.c_size_t()
        tid = self.lib.ref_vote_tile(ids.ctypes.data, off.ctypes.data, len(frames), C.byref(tc), li.ctypes.data, lc.ctypes.data, cap, C.byref(n))
        return int(tid), int(tc.value), [(int(li[i]), int(lc[i])) for i in range(n.value)]

    def smooth_tiles(self, ids, lists, threshold):
        """ids: top ID per tile; lists: per tile [(id, count), ...] count descending.  -> (ids, bools, assigned)"""
        n = len(ids)
        a = np.array(ids, dtype=np.uint32)
        b = np.zeros(max(n, 1), dtype=np.uint8)
        off = np.zeros(n + 1, dtype=np.uint64)
        for i, l in enumerate(lists):
            off[i + 1] = off[i] + len(l)
        li = np.array([e[0] for l in lists for e in l] + [0], dtype=np.uint32)
        lc = np.array([e[1] for l in lists for e in l] + [0], dtype=np.uint32)
        na = self.lib.ref_smooth_tiles(n, a.ctypes.data, b.ctypes.data, off.ctypes.data, li.ctypes.data, lc.ctypes.data, threshold)
        return a, b[:n], int(na)

    def find_longest_stretch(self, bools):
        b = np.ascontiguousarray(bools, dtype=np.uint8)
        s, e = C.c_long(), C.c_long()
        self.lib.ref_find_longest_stretch(b.ctypes.data, b.size, C.byref(s), C.byref(e))
        return s.value, e.value

    def eval_flanks(self, ls, le, ids):
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        ts, te = C.c_size_t(), C.c_size_t()
        g = self.lib.ref_eval_flanks(ls, le, a.ctypes.data, a.size, C.byref(ts), C.byref(te))
        return bool(g), ts.value, te.value

    def calc_optimal_size(self, entries, hash_num, occupancy):
        return int(self.lib.ref_calc_optimal_size(entries, hash_num, occupancy))

    def hash_universe(self, weight, genome_size, hash_num):
        return int(self.lib.ref_hash_universe(weight, genome_size, hash_num))


def random_tiles(rng, n, wrap=False):
    """Random per-tile query results shaped like real ones: a few IDs close to each other (so the
    +-1 rules fire), lists of (id, count > 2); wrap: IDs around 0 / 2^32 - 1 (uint32 wrap-around
    of the id +- 1 tests)."""
    base = int(rng.integers(1, 50))
    pool = [base + int(d) for d in rng.integers(0, 4, size=4)] + [int(rng.integers(1, 1000))]
    if wrap:
        pool = [0xFFFFFFFF, 0xFFFFFFFE, 1, 2, int(rng.integers(1, 1000))]
    if rng.random() < 0.1:
        pool.append(0)
    ids, lists = [], []
    for _ in range(n):
        k = int(rng.integers(0, 4))
        chosen = list(dict.fromkeys(int(pool[i]) for i in rng.integers(0, len(pool), size=k)))
        chosen = [c for c in chosen if c != 0]
        lst = sorted(((c, int(rng.choice([3, 4, 9, 10, 11, 12, 40, 400]))) for c in chosen), key=lambda t: (-t[1], t[0]))
        if lst and rng.random() < 0.9:
            top = min((c for c in lst if c[1] == lst[0][1]), key=lambda t: t[0])[0]
        else:
            top = int(pool[int(rng.integers(0, len(pool)))]) if not lst else lst[0][0]
        ids.append(top)
        lists.append(lst)
    return ids, lists


def run_patterns(n):
    """Tile states built from an assigned / unassigned pattern: the edge cases of P5 / P10 and
    find_longest_stretch (runs touching tile 0, n-2, n-1)."""
    pats = []
    for a in range(0, min(n, 9)):
        for b in range(a, min(n, a + 9)):
            pats.append([0 if a <= i <= b else 1 for i in range(n)])   # one unassigned run [a, b]
            pats.append([1 if a <= i <= b else 0 for i in range(n)])   # one assigned run
    for a in range(max(0, n - 9), n):
        pats.append([0 if i >= a else 1 for i in range(n)])            # runs reaching the end
        pats.append([1 if i >= a else 0 for i in range(n)])
        if a >= 1:
            pats.append([0 if a - 1 <= i < n - 1 else 1 for i in range(n)])  # ... reaching n-2
            pats.append([1 if a - 1 <= i < n - 1 else 0 for i in range(n)])
    return pats


def tiles_from_pattern(pat, id0=7, step=0, count_hi=40, count_lo=4):
    """assigned tile: list [(id, count_hi)]; unassigned: [(id, count_lo)] (below the threshold 10)"""
    ids, lists = [], []
    for i, p in enumerate(pat):
        tid = (id0 + step * (i // 10)) & 0xFFFFFFFF
        ids.append(tid)
        lists.append([(tid, count_hi if p else count_lo)])
    return ids, lists


def random_frames(rng, max_frames=120):
    """The IDs each frame of a tile returns (after the saturation bit is stripped, zeros dropped): a small
    pool, so that counts tie, sit at the `> 2` border of the list and repeat inside a frame (the
    reference's std::set counts a frame once per ID)."""
    pool = [int(v) for v in rng.integers(1, 40, size=int(rng.integers(1, 7)))]
    if rng.random() < 0.2:
        pool += [0xFFFFFFFF, 0x7FFFFFFF]
    n = int(rng.integers(0, max_frames + 1))
    frames = []
    for _ in range(n):
        k = int(rng.choice([0, 0, 1, 1, 2, 3, 5]))
        frames.append([int(pool[i]) for i in rng.integers(0, len(pool), size=k)])
    return frames


def canon_vote(res):
    """(id, count, list) with the list in the canonical order count descending, ID ascending (the reference's
    std::sort leaves equal counts in an unspecified order; only list[0].second and membership are consumed)"""
    tid, tc, lst = res
    return tid, tc, sorted(((int(a), int(c)) for a, c in lst), key=lambda t: (-t[1], t[0]))
