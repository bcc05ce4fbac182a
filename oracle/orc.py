"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_build/liboracle.so, the CPU restatement of the
reference algorithm (see orc_path.h).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product path
(goldrush_amd/, the goldrush-path CLI) never does.

PARITY UNPINNED: the reference holds no golden vectors for this path and
cannot be built in this image; its hash arithmetic lives in the un-vendored
btllib (requirements.txt:8, "btllib >=1.6.2").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
CLI_PATH = os.path.join(_HERE, "_build", "goldrush-path-oracle")

_lib = None

id_count_dtype = np.dtype([("id", "<u4"), ("count", "<u4")])


class orc_opts(C.Structure):
    _fields_ = [
        ("assigned_max", C.c_size_t), ("unassigned_min", C.c_size_t), ("tile_length", C.c_size_t),
        ("hash_universe", C.c_uint64), ("genome_size", C.c_uint64), ("kmer_size", C.c_size_t),
        ("weight", C.c_size_t), ("min_length", C.c_size_t), ("hash_num", C.c_size_t),
        ("occupancy", C.c_double), ("ratio", C.c_double), ("jobs", C.c_size_t),
        ("block_size", C.c_size_t), ("max_paths", C.c_size_t), ("threshold", C.c_size_t),
        ("phred_min", C.c_uint32), ("phred_delta", C.c_uint32),
        ("prefix_file", C.c_char * 4096), ("input", C.c_char * 4096), ("seed_preset", C.c_char * 512),
        ("filter_file", C.c_char * 4096),
        ("help", C.c_int), ("ntcard", C.c_int), ("silver_path", C.c_int), ("verbose", C.c_int), ("debug", C.c_int),
    ]


class orc_decision(C.Structure):
    _fields_ = [
        ("decision", C.c_int), ("num_tiles", C.c_size_t), ("num_assigned", C.c_size_t),
        ("trim_start", C.c_size_t), ("trim_end", C.c_size_t), ("first_id", C.c_uint32),
        ("path_at_write", C.c_uint64), ("finished", C.c_int),
    ]


class orc_record(C.Structure):
    _fields_ = [("id", C.c_char_p), ("seq", C.c_char_p), ("qual", C.c_char_p), ("len", C.c_size_t), ("qlen", C.c_size_t)]


class orc_reads(C.Structure):
    _fields_ = [("rec", C.POINTER(orc_record)), ("n", C.c_size_t), ("is_fastq", C.c_int)]


class orc_log_info(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "valid_reads", "total_tiles_per_path", "total_assigned_tiles_per_path", "total_unassigned_tiles_per_path",
        "total_queries_per_path", "total_hits_per_path", "total_misses_per_path", "num_reads_in_path")] + [("phred_sum_in_path", C.c_double)]


DEC_NAMES = ("skip_short", "skip_filtered", "insert_whole", "assigned_all", "insert_trimmed", "assigned")


def build() -> str:
    res = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building the oracle failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, u64, u32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t
    sig = {
        "orc_srol1": (u64, [u64]), "orc_sror1": (u64, [u64]), "orc_srol": (u64, [u64, C.c_uint]),
        "orc_base_seed": (u64, [C.c_ubyte]),
        "orcpy_seeds_new": (vp, [C.POINTER(C.c_char_p), C.c_uint]), "orcpy_seeds_free": (None, [vp]),
        "orc_seed_hash_at": (u64, [vp, C.c_char_p, sz]),
        "orc_multi_hash": (sz, [vp, C.c_uint, C.c_char_p, sz, vp, sz]),
        "orcpy_tile_hashes": (sz, [vp, C.c_uint, C.c_char_p, sz, sz, sz, sz, vp, sz]),
        "orc_calc_optimal_size": (u64, [u64, C.c_uint, C.c_double]),
        "orc_ntcard_new": (vp, [C.c_uint, u64]), "orc_ntcard_free": (None, [vp]),
        "orc_ntcard_add_read": (None, [vp, vp, C.c_char_p, sz]),
        "orc_ntcard_zero_buckets": (u64, [vp, C.c_uint, C.c_uint]), "orc_ntcard_f0": (u64, [vp, C.c_uint]),
        "orcpy_ntcard_counters": (vp, [vp]),
        "orc_mibf_create": (vp, [u64, C.c_uint]), "orc_mibf_destroy": (None, [vp]),
        "orc_mibf_insert_bv": (None, [vp, vp, sz]), "orc_mibf_finalize": (None, [vp]),
        "orc_mibf_bit": (C.c_int, [vp, u64]), "orc_mibf_rank": (u64, [vp, u64]),
        "orc_mibf_reset_ids": (None, [vp]),
        "orcpy_bv_insert_read": (None, [vp, vp, C.c_uint, C.c_char_p, sz]),
        "orcpy_bv_insert_reads": (None, [vp, vp, C.c_uint, vp, vp, sz]),
        "orcpy_insert_read_tiles": (None, [vp, vp, C.c_uint, C.c_char_p, sz, sz, sz, sz, sz, u32]),
        "orcpy_mibf_bv": (vp, [vp]), "orcpy_mibf_n_words": (u64, [vp]), "orcpy_mibf_pop": (u64, [vp]),
        "orcpy_mibf_data": (vp, [vp]), "orcpy_mibf_counts": (vp, [vp]), "orcpy_mibf_m": (u64, [vp]),
        "orcpy_sizeof_opts": (sz, []), "orcpy_sizeof_decision": (sz, []),
        "orc_query_tile": (sz, [vp, vp, sz, C.c_uint, C.POINTER(u32), C.POINTER(u32), vp, sz, vp]),
        "orc_vote_tile": (sz, [vp, vp, sz, C.POINTER(u32), C.POINTER(u32), vp, sz]),
        "orc_smooth_tiles": (sz, [sz, vp, vp, vp, vp, sz, vp]),
        "orc_find_longest_stretch": (None, [vp, sz, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
        "orc_eval_flanks": (C.c_int, [C.c_long, C.c_long, vp, sz, C.POINTER(sz), C.POINTER(sz)]),
        "orc_make_seed_pattern": (C.c_int, [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, vp, sz]),
        "orc_hash_universe": (u64, [u64, u64, u64]),
        "orc_calc_phred_average": (None, [C.c_char_p, sz, C.POINTER(u32), C.POINTER(u32)]),
        "orc_sum_phred": (C.c_double, [C.c_char_p, sz]),
        "orc_opts_default": (None, [C.POINTER(orc_opts)]),
        "orc_process_options": (C.c_int, [C.POINTER(orc_opts), C.c_int, C.POINTER(C.c_char_p)]),
        "orc_reads_load": (C.c_int, [C.POINTER(orc_reads), C.c_char_p]), "orc_reads_free": (None, [C.POINTER(orc_reads)]),
        "orc_path_open": (vp, [C.POINTER(orc_opts), C.POINTER(orc_reads), vp, C.POINTER(C.c_int)]),
        "orc_path_process_read": (None, [vp, sz, C.POINTER(orc_decision)]),
        "orc_path_close": (None, [vp]),
        "orc_path_use_external_bits": (None, [vp, u64]),
        "orc_path_set_state": (None, [vp, u32, u64, u32]),
        "orc_path_mibf": (vp, [vp]), "orc_path_log_info": (C.POINTER(orc_log_info), [vp]),
        "orc_path_phred_min": (u32, [vp]), "orc_path_seed": (C.c_char_p, [vp, C.c_uint]),
        "orc_path_filter_size": (u64, [vp]), "orc_path_is_filtered": (C.c_int, [vp, sz]),
        "orc_path_timers": (None, [vp, C.POINTER(C.c_double)]),
        "orc_main": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    assert lib.orcpy_sizeof_opts() == C.sizeof(orc_opts), "orc_opts layout mismatch"
    assert lib.orcpy_sizeof_decision() == C.sizeof(orc_decision), "orc_decision layout mismatch"
    _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_seed_pattern(preset: str, k: int, w: int, h: int):
    lib = load()
    stride = 512
    buf = C.create_string_buffer(stride * h)
    rc = lib.orc_make_seed_pattern(preset.encode(), k, w, h, buf, stride)
    if rc != 0:
        raise ValueError("orc_make_seed_pattern failed")
    return [buf.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode() for i in range(h)]


class Seeds:
    def __init__(self, patterns):
        self.lib = load()
        self.patterns = [p if isinstance(p, str) else p.decode() for p in patterns]
        self.h = len(patterns)
        arr = (C.c_char_p * self.h)(*[p.encode() for p in self.patterns])
        self._h = self.lib.orcpy_seeds_new(arr, self.h)
        if not self._h:
            raise ValueError("bad seed pattern")

    def multi_hash(self, seq: bytes) -> np.ndarray:
        frames = self.lib.orc_multi_hash(self._h, self.h, seq, len(seq), None, 0)
        out = np.zeros(frames * self.h, dtype=np.uint64)
        self.lib.orc_multi_hash(self._h, self.h, seq, len(seq), _p(out), out.size)
        return out

    def tile_hashes(self, seq: bytes, tile: int, k: int, tile_idx: int) -> np.ndarray:
        out = np.zeros((tile + 1) * self.h, dtype=np.uint64)
        n = self.lib.orcpy_tile_hashes(self._h, self.h, seq, len(seq), tile, k, tile_idx, _p(out), out.size)
        return out[:n]

    def __del__(self):
        try:
            self.lib.orcpy_seeds_free(self._h)
        except Exception:
            pass


class NtCard:
    """orc_ntcard wrapper (--ntcard restatement, goldrush_path/ntcard.hpp)."""

    def __init__(self, seeds: Seeds, input_bytes: int = 0):
        self.lib = load()
        self.seeds = seeds
        self._h = self.lib.orc_ntcard_new(seeds.h, input_bytes)

    def add_read(self, seq: bytes):
        self.lib.orc_ntcard_add_read(self._h, self.seeds._h, seq, len(seq))

    def zero_buckets(self) -> np.ndarray:
        return np.array([[self.lib.orc_ntcard_zero_buckets(self._h, s, t) for t in range(2)] for s in range(self.seeds.h)], dtype=np.uint64)

    def f0(self, seed: int) -> int:
        return int(self.lib.orc_ntcard_f0(self._h, seed))

    def counters(self) -> np.ndarray:
        """[h, 2, 2^27] uint16 view of the sample tables."""
        n = self.seeds.h * 2 * (1 << 27)
        a = np.ctypeslib.as_array(C.cast(self.lib.orcpy_ntcard_counters(self._h), C.POINTER(C.c_uint16)), shape=(n,))
        return a.reshape(self.seeds.h, 2, 1 << 27)

    def close(self):
        if self._h:
            self.lib.orc_ntcard_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MiBF:
    """orc_mibf wrapper (MIBloomFilter + MIBFConstructSupport restatement)."""

    def __init__(self, m: int, seeds: Seeds, tile: int, k: int, handle=None):
        self.lib = load()
        self.seeds = seeds
        self.h = seeds.h
        self.tile, self.k = tile, k
        self._own = handle is None
        self._h = self.lib.orc_mibf_create(m, self.h) if handle is None else handle
        self.m = m

    def bv_insert_read(self, seq: bytes):
        self.lib.orcpy_bv_insert_read(self._h, self.seeds._h, self.h, seq, len(seq))

    def bv_insert_reads(self, seqs):
        """insertBV of many reads, OpenMP over the reads (the reference's fill loop is parallel too)"""
        n = len(seqs)
        arr = (C.c_char_p * n)(*seqs)
        lens = np.array([len(s) for s in seqs], dtype=np.uint64)
        self.lib.orcpy_bv_insert_reads(self._h, self.seeds._h, self.h, C.cast(arr, C.c_void_p), _p(lens), n)

    def finalize(self) -> int:
        self.lib.orc_mibf_finalize(self._h)
        return self.pop

    @property
    def pop(self) -> int:
        return self.lib.orcpy_mibf_pop(self._h)

    def bits(self) -> np.ndarray:
        n = self.lib.orcpy_mibf_n_words(self._h)
        p = C.cast(self.lib.orcpy_mibf_bv(self._h), C.POINTER(C.c_uint64))
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def ids(self) -> np.ndarray:
        p = C.cast(self.lib.orcpy_mibf_data(self._h), C.POINTER(C.c_uint32))
        return np.ctypeslib.as_array(p, shape=(max(self.pop, 1),))[: self.pop]

    def counts(self) -> np.ndarray:
        p = C.cast(self.lib.orcpy_mibf_counts(self._h), C.POINTER(C.c_uint32))
        return np.ctypeslib.as_array(p, shape=(max(self.pop, 1),))[: self.pop]

    def rank(self, pos: int) -> int:
        return self.lib.orc_mibf_rank(self._h, pos)

    def bit(self, pos: int) -> int:
        return self.lib.orc_mibf_bit(self._h, pos)

    def query_tile(self, hashes: np.ndarray):
        """Returns (top_id, top_count, list[(id,count)], (queries, hits, misses))."""
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        cap = max(hashes.size, 1)
        lst = np.zeros(cap, dtype=id_count_dtype)
        ctr = np.zeros(3, dtype=np.uint64)
        tid, tc = C.c_uint32(), C.c_uint32()
        n = self.lib.orc_query_tile(self._h, _p(hashes), hashes.size, self.h, C.byref(tid), C.byref(tc), _p(lst), cap, _p(ctr))
        return tid.value, tc.value, lst[:n].copy(), tuple(int(x) for x in ctr)

    def query_read(self, seq: bytes):
        """All tiles of a read: list of query_tile results."""
        nt = len(seq) // self.tile
        return [self.query_tile(self.seeds.tile_hashes(seq, self.tile, self.k, t)) for t in range(nt)]

    def insert_read_tiles(self, seq: bytes, start: int, end: int, id_: int):
        self.lib.orcpy_insert_read_tiles(self._h, self.seeds._h, self.h, seq, len(seq), self.tile, self.k, start, end, id_)

    def reset_ids(self):
        self.lib.orc_mibf_reset_ids(self._h)

    def close(self):
        if self._own and self._h:
            self.lib.orc_mibf_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def vote_tile(frames):
    """The per-tile vote given every frame's IDs (goldrush_path.cpp:597-622).  frames: list of ID lists.
    -> (top_id, top_count, [(id, count), ...] with count > 2, count descending then ID ascending)"""
    lib = load()
    off = np.zeros(len(frames) + 1, dtype=np.uint64)
    for i, f in enumerate(frames):
        off[i + 1] = off[i] + len(f)
    ids = np.array([x for f in frames for x in f] + [0], dtype=np.uint32)
    cap = max(int(off[-1]), 1)
    lst = np.zeros(cap, dtype=id_count_dtype)
    tid, tc = C.c_uint32(), C.c_uint32()
    n = lib.orc_vote_tile(_p(ids), _p(off), len(frames), C.byref(tid), C.byref(tc), _p(lst), cap)
    return tid.value, tc.value, [(int(a), int(c)) for a, c in lst[:n]]


def smooth_tiles(ids, lists, threshold: int):
    """orc_smooth_tiles: ids = per-tile top ids; lists = per-tile arrays of id_count_dtype.
    Returns (ids_out, bools_out, n_assigned)."""
    lib = load()
    n = len(ids)
    ids_a = np.array(ids, dtype=np.uint32)
    bools = np.zeros(max(n, 1), dtype=np.uint8)
    keep = [np.ascontiguousarray(l, dtype=id_count_dtype) for l in lists]
    ptrs = (C.c_void_p * max(n, 1))(*[k.ctypes.data if k.size else None for k in keep])
    ln = np.array([k.size for k in keep], dtype=np.uint64)
    na = lib.orc_smooth_tiles(n, _p(ids_a), _p(bools), C.cast(ptrs, C.c_void_p), _p(ln), threshold, None)
    return ids_a, bools[:n], na


def find_longest_stretch(bools):
    lib = load()
    b = np.ascontiguousarray(bools, dtype=np.uint8)
    s, e = C.c_long(), C.c_long()
    lib.orc_find_longest_stretch(_p(b), b.size, C.byref(s), C.byref(e))
    return s.value, e.value


def eval_flanks(ls: int, le: int, ids):
    lib = load()
    a = np.ascontiguousarray(ids, dtype=np.uint32)
    ts, te = C.c_size_t(), C.c_size_t()
    good = lib.orc_eval_flanks(ls, le, _p(a), a.size, C.byref(ts), C.byref(te))
    return bool(good), ts.value, te.value


def calc_phred_average(qual: bytes):
    lib = load()
    a, d = C.c_uint32(), C.c_uint32()
    lib.orc_calc_phred_average(qual, len(qual), C.byref(a), C.byref(d))
    return a.value, d.value


def parse_opts(argv):
    """argv without the program name. Returns (orc_opts, exit_code or -1)."""
    lib = load()
    o = orc_opts()
    lib.orc_opts_default(C.byref(o))
    args = [b"goldrush_path"] + [a.encode() if isinstance(a, str) else a for a in argv]
    arr = (C.c_char_p * (len(args) + 1))(*args, None)
    rc = lib.orc_process_options(C.byref(o), len(args), arr)
    return o, rc


class Path:
    """orc_path: main() + process_read() of the reference, read by read."""

    def __init__(self, argv, log_path: str | None = None, external_bits=None):
        self.lib = load()
        if external_bits is not None:  # uint64 words of the whole data set's bit vector, OR-ed in before the rank build
            self.lib.orc_path_use_external_bits(_p(external_bits), external_bits.size)
        self.opts, rc = parse_opts(argv)
        if rc >= 0:
            raise SystemExit(rc)
        self.reads = orc_reads()
        if self.lib.orc_reads_load(C.byref(self.reads), self.opts.input) != 0:
            raise FileNotFoundError(self.opts.input)
        self._logf = None
        if log_path:
            libc = C.CDLL(None)
            libc.fopen.restype = C.c_void_p
            libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
            self._libc = libc
            self._logf = libc.fopen(log_path.encode(), b"w")
        ec = C.c_int(-1)
        self._h = self.lib.orc_path_open(C.byref(self.opts), C.byref(self.reads), self._logf, C.byref(ec))
        self.exit_code = ec.value
        self.n_reads = self.reads.n

    @property
    def ok(self):
        return bool(self._h)

    def process(self, idx: int) -> orc_decision:
        d = orc_decision()
        self.lib.orc_path_process_read(self._h, idx, C.byref(d))
        return d

    def run_all(self):
        out = []
        for i in range(self.n_reads):
            d = self.process(i)
            if d.finished and d.decision == 0 and d.num_tiles == 0:
                break
            out.append((d.decision, d.num_tiles, d.num_assigned, d.trim_start, d.trim_end, d.first_id, d.path_at_write))
            if d.finished:
                break
        return out

    def start_producers(self, n: int, first: int, count: int):
        """n hashing threads ahead of process() for reads [first, first + count) (read_hashing.cpp:77-117: the reference runs 6)"""
        self.lib.orc_path_start_producers.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t]
        if self.lib.orc_path_start_producers(self._h, n, first, count) != 0:
            raise ValueError("bad producer range")

    def stop_producers(self):
        self.lib.orc_path_stop_producers.argtypes = [C.c_void_p]
        self.lib.orc_path_stop_producers(self._h)

    def set_state(self, ids_inserted: int, inserted_bases: int, id_: int):
        self.lib.orc_path_set_state(self._h, ids_inserted, inserted_bases, id_)

    def mibf_handle(self):
        return self.lib.orc_path_mibf(self._h)

    def mibf(self, tile: int, k: int, seeds: Seeds) -> MiBF:
        h = self.lib.orc_path_mibf(self._h)
        return MiBF(self.lib.orcpy_mibf_m(h), seeds, tile, k, handle=h)

    def seed(self, i: int) -> str:
        return self.lib.orc_path_seed(self._h, i).decode()

    def filter_size(self) -> int:
        return self.lib.orc_path_filter_size(self._h)

    def phred_min(self) -> int:
        return self.lib.orc_path_phred_min(self._h)

    def is_filtered(self, idx: int) -> bool:
        return bool(self.lib.orc_path_is_filtered(self._h, idx))

    def log_info(self) -> dict:
        li = self.lib.orc_path_log_info(self._h).contents
        return {n: getattr(li, n) for n, _ in orc_log_info._fields_}

    def timers(self):
        t = (C.c_double * 2)()
        self.lib.orc_path_timers(self._h, t)
        return t[0], t[1]

    def record(self, idx: int):
        r = self.reads.rec[idx]
        return r.id, r.seq, r.qual

    def close(self):
        if self._h:
            self.lib.orc_path_close(self._h)
            self._h = None
        if self._logf:
            self._libc.fclose.argtypes = [C.c_void_p]
            self._libc.fclose(self._logf)
            self._logf = None
        self.lib.orc_reads_free(C.byref(self.reads))


def run_cli(argv, cwd=None, timeout=None):
    """Run the oracle CLI as a subprocess; returns CompletedProcess."""
    if not os.path.exists(CLI_PATH):
        build()
    return subprocess.run([CLI_PATH] + list(argv), cwd=cwd, capture_output=True, text=True, timeout=timeout)
