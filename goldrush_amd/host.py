"""ctypes binding of include/grpath_host.h (libgrpath_host.so): the C++ host
logic of goldrush-path (seed design, sizing, Phred, tile decisions, the
order-exact classifier).  Used by the tests and bench.py; the CLI links the same
sources."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import native

LIB_PATH = os.path.join(native.LIB_DIR, "libgrpath_host.so")
CLI_PATH = os.path.join(os.path.dirname(native.LIB_DIR), "bin", "goldrush-path")

_vp = C.c_void_p
decision_dtype = np.dtype([("kind", "<u4"), ("num_tiles", "<u4"), ("num_assigned", "<u4"), ("trim_start", "<u4"), ("trim_end", "<u4"),
                           ("hits", "<u4"), ("misses", "<u4"), ("pad", "<u4")])
KIND_NAMES = {2: "insert_whole", 3: "assigned_all", 4: "insert_trimmed", 5: "assigned"}


class gr_read_decision(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("kind", "num_tiles", "num_assigned", "trim_start", "trim_end", "hits", "misses", "pad")]


class gr_commit(C.Structure):
    _fields_ = [("read", C.c_uint32), ("dec", gr_read_decision), ("first_id", C.c_uint32), ("path", C.c_uint64)]


class gr_classifier_params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("tile_length", C.c_uint32), ("block_size", C.c_uint32), ("threshold", C.c_uint32),
                ("unassigned_min", C.c_uint32), ("assigned_max", C.c_uint32), ("kmer_size", C.c_uint32), ("hash_num", C.c_uint32),
                ("target_bases", C.c_uint64), ("max_paths", C.c_uint64), ("silver_path", C.c_int32), ("verbose", C.c_int32),
                ("max_window", C.c_uint32), ("world", C.c_uint32), ("rank", C.c_uint32), ("debug", C.c_int32)]


class gr_classifier_state(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("valid_reads", "total_tiles", "assigned_tiles", "unassigned_tiles", "queries", "hits", "misses", "num_reads_in_path")] + \
               [("phred_sum_in_path", C.c_double), ("inserted_bases", C.c_uint64), ("curr_path", C.c_uint64), ("id", C.c_uint32), ("ids_inserted", C.c_uint32)] + \
               [(n, C.c_uint64) for n in ("windows", "reads_queried", "reads_committed", "inserts")] + \
               [("seconds_windows", C.c_double), ("seconds_commit", C.c_double)] + \
               [(n, C.c_uint64) for n in ("batches", "batches_undone", "batch_reads", "batches_refused", "batches_fused", "stream_inserts", "stream_insert_fallbacks", "stream_relaunches", "stream_handbacks", "stream_rollovers", "batch_overlap_cuts", "overlap_calls")]


# engine function table: members typed exactly like include/grpath.h
VT_TYPES = [
    ("create", C.CFUNCTYPE(C.c_int, C.POINTER(native.grp_params), C.POINTER(_vp))),
    ("destroy", C.CFUNCTYPE(None, _vp)),
    ("last_error", C.CFUNCTYPE(_vp, _vp)),  # const char*: a Python engine returns the address of a buffer it keeps (a c_char_p result of a callback cannot be kept alive by ctypes)
    ("reads_upload", C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(_vp))),
    ("reads_free", C.CFUNCTYPE(None, _vp)),
    ("bv_insert", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32)),
    ("finalize", C.CFUNCTYPE(C.c_int, _vp, C.POINTER(C.c_uint64))),
    ("query_tiles", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64), _vp)),
    ("insert_tiles", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)),
    ("reset_ids", C.CFUNCTYPE(C.c_int, _vp)),
    ("sync", C.CFUNCTYPE(C.c_int, _vp)),
    ("classify_reads", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp)),
    ("insert_read", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)),
    ("classify_begin", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32)),
    ("classify_end", C.CFUNCTYPE(C.c_int, _vp, C.c_uint32, _vp)),
    ("stream_begin", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp)),
    ("stream_abort", C.CFUNCTYPE(C.c_int, _vp, C.c_uint32)),
    ("stream_poll", C.CFUNCTYPE(C.c_int, _vp, C.c_uint32)),
    ("stream_end", C.CFUNCTYPE(C.c_int, _vp, C.c_uint32, _vp)),
    ("batch_insert", C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32)),
    ("batch_classify", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp)),
    ("batch_undo", C.CFUNCTYPE(C.c_int, _vp, C.c_uint32, C.c_uint32)),
    ("batch_end", C.CFUNCTYPE(C.c_int, _vp)),
    ("ntcard_begin", C.CFUNCTYPE(C.c_int, _vp, C.c_uint32)),
    ("ntcard_add", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, _vp)),
    ("ntcard_finish", C.CFUNCTYPE(C.c_int, _vp, _vp)),
    ("set_filter_size", C.CFUNCTYPE(C.c_int, _vp, C.c_uint64)),
    ("fastq_parse", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint64, C.c_int, C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int))),
    ("fastq_records", C.CFUNCTYPE(C.c_int, _vp, _vp)),
    ("fastq_pack", C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, C.c_uint32, C.POINTER(_vp))),
    ("fastq_free", C.CFUNCTYPE(None, _vp)),
    ("stream_begin_resumable", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, _vp)),
    ("stream_insert", C.CFUNCTYPE(C.c_int, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32))),
    ("comm_unique_id", C.CFUNCTYPE(C.c_int, _vp, C.c_size_t)),
    ("comm_init", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32)),
    ("bv_merge_ranks", C.CFUNCTYPE(C.c_int, _vp)),
    ("bv_words", C.CFUNCTYPE(C.c_int, _vp, C.POINTER(C.c_uint64))),
    ("bv_export_words", C.CFUNCTYPE(C.c_int, _vp, C.c_uint64, C.c_uint64, _vp)),
    ("bv_or_words", C.CFUNCTYPE(C.c_int, _vp, C.c_uint64, C.c_uint64, _vp)),
    ("fastq_pin", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint64)),
    ("fastq_unpin", C.CFUNCTYPE(C.c_int, _vp)),
    ("batch_verify", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp)),
    ("window_overlap", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp)),
    ("stream_begin_striped_resumable", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp)),
    ("stream_resumable", C.CFUNCTYPE(C.c_int, _vp, C.c_uint32)),
    ("stream_insert_done", C.CFUNCTYPE(C.c_int, _vp, C.c_uint32)),
    ("fastq_prefetch", C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint64)),
    ("occupancy_hint", C.CFUNCTYPE(C.c_int, _vp, C.c_double)),
]


class grp_engine_vt(C.Structure):
    _fields_ = VT_TYPES


COMMIT_FN = C.CFUNCTYPE(C.c_double, _vp, C.POINTER(gr_commit))
ROLLOVER_FN = C.CFUNCTYPE(None, _vp, C.c_uint64)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint64, _vp)

SIGNATURES = {
    "gr_make_seed_pattern": (C.c_int, [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, _vp, C.c_size_t, C.c_int]),
    "gr_hash_universe": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64]),
    "gr_calc_optimal_size": (C.c_uint64, [C.c_uint64, C.c_uint, C.c_double]),
    "gr_calc_phred_average": (None, [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gr_sum_phred": (C.c_double, [C.c_char_p, C.c_size_t]),
    "gr_pack_2bit": (C.c_int, [C.c_char_p, C.c_size_t, _vp]),
    "gr_effective_cpus": (C.c_uint, []),
    "gr_process_options_dump": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_size_t]),
    "gr_ntcard_sbits": (C.c_uint, [C.c_uint64]),
    "gr_ntcard_f0": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint]),
    "gr_ntcard_split": (C.c_size_t, [C.c_char_p, C.c_size_t, C.c_uint, C.c_uint, _vp, _vp, _vp, C.c_size_t]),
    "gr_decide_read": (None, [C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, C.POINTER(gr_read_decision)]),
    "gr_smooth_tiles": (C.c_size_t, [C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp]),
    "gr_find_longest_stretch": (None, [_vp, C.c_size_t, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "gr_eval_flanks": (C.c_int, [C.c_long, C.c_long, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "gr_classifier_create": (C.c_int, [C.POINTER(gr_classifier_params), C.POINTER(grp_engine_vt), _vp, C.POINTER(_vp)]),
    "gr_classifier_destroy": (None, [_vp]),
    "gr_classifier_set_allgather": (None, [_vp, _vp, _vp]),
    "gr_classifier_set_debug": (None, [_vp, _vp]),
    "gr_classifier_keep_commits": (None, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "gr_classifier_kept_commits": (C.c_size_t, [_vp, _vp, C.c_size_t]),
    "gr_shm_allgather_open": (_vp, [C.c_uint32, C.c_uint32, C.c_char_p, C.c_double]),
    "gr_shm_allgather": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "gr_fill_merge_plan": (C.c_int, [C.POINTER(grp_engine_vt), _vp, _vp, C.c_uint32, C.c_uint32, C.c_int]),
    "gr_fill_merge_run": (C.c_int, [C.POINTER(grp_engine_vt), _vp, _vp, C.c_uint32, C.c_uint32, C.c_int]),
    "gr_ranks_same_u64": (C.c_int, [_vp, C.c_uint32, C.c_uint64]),
    "gr_ranks_share_device": (C.c_int, [_vp, C.c_uint32, C.c_int]),
    "gr_shm_allgather_close": (None, [_vp]),
    "gr_classifier_set_callbacks": (None, [_vp, COMMIT_FN, ROLLOVER_FN, ALLGATHER_FN, _vp]),
    "gr_classifier_run": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_int)]),
    "gr_classifier_run_range": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_int)]),
    "gr_classifier_error": (C.c_char_p, [_vp]),
    "gr_classifier_get_state": (None, [_vp, C.POINTER(gr_classifier_state)]),
    "gr_path_main": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(grp_engine_vt)]),
    "gr_input_read": (C.c_uint64, [C.c_char_p, C.c_uint64, _vp, C.c_uint64]),
}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        native.build()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def make_seed_pattern(preset: str, k: int, w: int, h: int):
    lib = load()
    stride = 512
    buf = C.create_string_buffer(stride * h)
    if lib.gr_make_seed_pattern(preset.encode(), k, w, h, buf, stride, 0) != 0:
        raise ValueError("gr_make_seed_pattern failed")
    return [buf.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode() for i in range(h)]


def calc_phred_average(qual: bytes):
    a, d = C.c_uint32(), C.c_uint32()
    load().gr_calc_phred_average(qual, len(qual), C.byref(a), C.byref(d))
    return a.value, d.value


def pack_2bit(seq: bytes):
    out = np.zeros((len(seq) + 15) // 16 or 1, dtype=np.uint32)
    rc = load().gr_pack_2bit(seq, len(seq), _p(out))
    return rc, out[: (len(seq) + 15) // 16]


def process_options_dump(argv):
    """(exit code or -1, {name: value}) of the host's process_options on argv (no program name)."""
    args = [b"goldrush_path"] + [a.encode() if isinstance(a, str) else a for a in argv]
    arr = (C.c_char_p * (len(args) + 1))(*args, None)
    buf = C.create_string_buffer(1 << 16)
    rc = load().gr_process_options_dump(len(args), arr, buf, len(buf))
    return rc, dict(l.split("=", 1) for l in buf.value.decode().splitlines() if "=" in l)


def ntcard_split(seq: bytes, k: int, h: int):
    """ACGT runs (offset, length) of a record and the stale repeats [runs, h]."""
    cap = len(seq) // max(k, 1) + 2
    off = np.zeros(cap, dtype=np.uint64)
    ln = np.zeros(cap, dtype=np.uint64)
    ex = np.zeros(cap * h, dtype=np.uint32)
    n = load().gr_ntcard_split(seq, len(seq), k, h, _p(off), _p(ln), _p(ex), cap)
    return [(int(off[i]), int(ln[i])) for i in range(n)], ex[: n * h].reshape(n, h)


def tiles_from(ids, lists):
    """Build (tiles, flat lists) arrays from per-tile top ids and [(id,count)] lists
    (lists must be sorted count desc / id asc, as the engine returns them)."""
    n = len(ids)
    tiles = np.zeros(max(n, 1), dtype=native.tile_summary_dtype)
    flat = []
    for i in range(n):
        tiles[i]["top_id"] = ids[i]
        tiles[i]["list_off"] = len(flat)
        tiles[i]["list_n"] = len(lists[i])
        tiles[i]["top_count"] = lists[i][0][1] if len(lists[i]) else 0
        flat.extend(lists[i])
    fl = np.zeros(max(len(flat), 1), dtype=native.id_count_dtype)
    for j, (a, b) in enumerate(flat):
        fl[j] = (a, b)
    return tiles, fl


def smooth_tiles(tiles, lists, n, threshold):
    ids = np.zeros(max(n, 1), dtype=np.uint32)
    bools = np.zeros(max(n, 1), dtype=np.uint8)
    na = load().gr_smooth_tiles(n, _p(tiles), _p(lists), threshold, _p(ids), _p(bools))
    return ids[:n], bools[:n], na


def decide_read(tiles, lists, n, threshold=10, unassigned_min=5, assigned_max=1):
    d = gr_read_decision()
    load().gr_decide_read(threshold, unassigned_min, assigned_max, n, _p(tiles), _p(lists), C.byref(d))
    return d


def find_longest_stretch(bools):
    b = np.ascontiguousarray(bools, dtype=np.uint8)
    s, e = C.c_long(), C.c_long()
    load().gr_find_longest_stretch(_p(b), b.size, C.byref(s), C.byref(e))
    return s.value, e.value


def eval_flanks(ls, le, ids):
    a = np.ascontiguousarray(ids, dtype=np.uint32)
    ts, te = C.c_size_t(), C.c_size_t()
    g = load().gr_eval_flanks(ls, le, _p(a), a.size, C.byref(ts), C.byref(te))
    return bool(g), ts.value, te.value


def hip_engine_vt() -> grp_engine_vt:
    """Function table filled with the grp_* symbols of libgrpath_hip.so."""
    lib = native.load()
    vt = grp_engine_vt()
    alias = {"classify_begin": "classify_reads_begin", "classify_end": "classify_reads_end", "stream_begin": "classify_stream_begin_striped",
             "stream_abort": "classify_stream_abort", "stream_insert": "classify_stream_insert", "stream_begin_resumable": "classify_stream_begin_resumable", "stream_poll": "classify_stream_poll", "stream_end": "classify_stream_end",
             "batch_insert": "batch_insert_reads", "stream_begin_striped_resumable": "classify_stream_begin_striped_resumable", "stream_resumable": "classify_stream_resumable", "stream_insert_done": "classify_stream_insert_done", "occupancy_hint": "set_occupancy_hint"}
    for name, ftype in VT_TYPES:
        sym = getattr(lib, "grp_" + alias.get(name, name))
        setattr(vt, name, C.cast(sym, ftype))
    return vt


class Classifier:
    """gr_classifier over an engine (HIP engine by default)."""

    def __init__(self, engine_handle, vt: grp_engine_vt, tile=1000, block=10, threshold=10, unassigned_min=5, assigned_max=1, k=22, h=3,
                 target_bases=0, max_paths=1, silver_path=False, verbose=False, max_window=0, world=1, rank=0, allgather=None, record=True):
        self.lib = load()
        self.vt = vt
        p = gr_classifier_params(C.sizeof(gr_classifier_params), tile, block, threshold, unassigned_min, assigned_max, k, h, target_bases, max_paths,
                                 1 if silver_path else 0, 1 if verbose else 0, max_window, world, rank, 0)
        out = _vp()
        rc = self.lib.gr_classifier_create(C.byref(p), C.byref(vt), engine_handle, C.byref(out))
        if rc != 0:
            raise RuntimeError(f"gr_classifier_create failed: {rc}")
        self._h = out
        self.commits = []
        self.rollovers = []

        def _commit(user, c):
            c = c.contents
            d = c.dec
            self.commits.append((c.read, d.kind, d.num_tiles, d.num_assigned, d.trim_start, d.trim_end, c.first_id, c.path, d.hits, d.misses))
            return 0.0

        def _roll(user, path):
            self.rollovers.append(path)

        self._cb = (COMMIT_FN(_commit) if record else C.cast(None, COMMIT_FN), ROLLOVER_FN(_roll),
                    ALLGATHER_FN(allgather) if allgather else C.cast(None, ALLGATHER_FN))
        self.lib.gr_classifier_set_callbacks(self._h, self._cb[0], self._cb[1], self._cb[2], None)

    def run(self, reads_handle, lens, skipped_before=None, skipped_after=0):
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        sb = None if skipped_before is None else np.ascontiguousarray(skipped_before, dtype=np.uint32)
        fin = C.c_int()
        rc = self.lib.gr_classifier_run(self._h, reads_handle, _p(lens), lens.size, _p(sb), skipped_after, C.byref(fin))
        if rc != 0:
            raise RuntimeError(f"gr_classifier_run: {rc}: {self.lib.gr_classifier_error(self._h).decode()}")
        return bool(fin.value)

    def run_range(self, reads_handle, lens, first, count):
        """reads [first, first+count) of the batch; lens = the batch's full length array"""
        fin = C.c_int()
        rc = self.lib.gr_classifier_run_range(self._h, reads_handle, _p(lens), first, count, None, 0, C.byref(fin))
        if rc != 0:
            raise RuntimeError(f"gr_classifier_run_range: {rc}: {self.lib.gr_classifier_error(self._h).decode()}")
        return bool(fin.value)

    def keep_commits(self, first0: int, count0: int, first1: int = 0, count1: int = 0):
        """keep the commits of these two ranges of reads inside the classifier (no callback per read)"""
        self.lib.gr_classifier_keep_commits(self._h, first0, count0, first1, count1)

    def kept_commits(self):
        """-> [(read, kind, num_tiles, num_assigned, trim_start, trim_end, first_id, path, hits, misses)] of the kept ranges, in commit order"""
        n = self.lib.gr_classifier_kept_commits(self._h, None, 0)
        arr = (gr_commit * max(n, 1))()
        self.lib.gr_classifier_kept_commits(self._h, arr, n)
        return [(c.read, c.dec.kind, c.dec.num_tiles, c.dec.num_assigned, c.dec.trim_start, c.dec.trim_end, c.first_id, c.path, c.dec.hits, c.dec.misses) for c in arr[:n]]

    def state(self) -> dict:
        s = gr_classifier_state()
        self.lib.gr_classifier_get_state(self._h, C.byref(s))
        return {n: getattr(s, n) for n, _ in gr_classifier_state._fields_}

    def close(self):
        if self._h:
            self.lib.gr_classifier_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
