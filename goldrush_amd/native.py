"""ctypes binding of the C ABI in include/grpath.h (libgrpath_hip.so).

The binding is plumbing only: every compute call goes to the hand-written HIP
kernels.  There is no Python / CPU fallback — if the library is missing or no
HIP device is usable the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
CSRC_DIR = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(LIB_DIR, "libgrpath_hip.so")

GRP_OK = 0
GRP_ERR_INVALID, GRP_ERR_NO_DEVICE, GRP_ERR_HIP, GRP_ERR_STATE, GRP_ERR_NOMEM, GRP_ERR_BUSY = -1, -2, -3, -4, -5, -6
GRP_K_FILL, GRP_K_RANK, GRP_K_QUERY, GRP_K_INSERT, GRP_K_DECIDE, GRP_K_NTCARD, GRP_K_QUERY_LAT, GRP_K_VERIFY, GRP_K_BATCH, GRP_K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
KERNEL_NAMES = ("fill", "rank", "query", "insert", "decide", "ntcard", "query_latency", "verify", "batch_insert")


class GrpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"grpath error {code}: {msg}")
        self.code = code


class grp_params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("k", C.c_uint32),
        ("h", C.c_uint32),
        ("tile", C.c_uint32),
        ("m", C.c_uint64),
        ("seeds", C.POINTER(C.c_char_p)),
        ("device", C.c_int32),
        ("flags", C.c_uint32),
    ]


tile_summary_dtype = np.dtype([("top_id", "<u4"), ("top_count", "<u4"), ("list_off", "<u4"), ("list_n", "<u4"), ("hits", "<u4"), ("misses", "<u4")])
id_count_dtype = np.dtype([("id", "<u4"), ("count", "<u4")])


class grp_decide_params(C.Structure):
    _fields_ = [("threshold", C.c_uint32), ("unassigned_min", C.c_uint32), ("assigned_max", C.c_uint32), ("reserved", C.c_uint32)]


decision_dtype = np.dtype([("kind", "<u4"), ("num_tiles", "<u4"), ("num_assigned", "<u4"), ("trim_start", "<u4"), ("trim_end", "<u4"),
                           ("hits", "<u4"), ("misses", "<u4"), ("pad", "<u4")])


class grp_query_stats(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("hits", C.c_uint64), ("misses", C.c_uint64)]


class grp_kernel_stat(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("units", C.c_uint64), ("ms", C.c_double)]


# name -> (restype, argtypes); every symbol include/grpath.h declares
_vp = C.c_void_p
SIGNATURES = {
    "grp_create": (C.c_int, [C.POINTER(grp_params), C.POINTER(_vp)]),
    "grp_destroy": (None, [_vp]),
    "grp_last_error": (C.c_char_p, [_vp]),
    "grp_reads_upload": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(_vp)]),
    "grp_reads_wrap_device": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(_vp)]),
    "grp_reads_free": (None, [_vp]),
    "grp_reads_tile0": (C.POINTER(C.c_uint64), [_vp]),
    "grp_ntcard_begin": (C.c_int, [_vp, C.c_uint32]),
    "grp_ntcard_add": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp]),
    "grp_ntcard_finish": (C.c_int, [_vp, _vp]),
    "grp_set_filter_size": (C.c_int, [_vp, C.c_uint64]),
    "grp_bv_insert": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32]),
    "grp_bv_words": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "grp_bv_export_device": (C.c_int, [_vp, _vp]),
    "grp_bv_merge_device": (C.c_int, [_vp, _vp]),
    "grp_words_or_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "grp_bv_import_device": (C.c_int, [_vp, _vp]),
    "grp_finalize": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "grp_set_occupancy_hint": (C.c_int, [_vp, C.c_double]),
    "grp_debug_finalize_times": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "grp_query_tiles": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(grp_query_stats)]),
    "grp_classify_reads": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(grp_decide_params), _vp]),
    "grp_classify_reads_begin": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(grp_decide_params), C.c_uint32]),
    "grp_classify_reads_end": (C.c_int, [_vp, C.c_uint32, _vp]),
    "grp_classify_stream_begin": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(grp_decide_params), C.c_uint32, C.POINTER(C.c_void_p)]),
    "grp_classify_stream_begin_striped": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(grp_decide_params), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "grp_classify_stream_abort": (C.c_int, [_vp, C.c_uint32]),
    "grp_classify_stream_poll": (C.c_int, [_vp, C.c_uint32]),
    "grp_classify_stream_end": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "grp_comm_unique_id": (C.c_int, [_vp, C.c_size_t]),
    "grp_comm_init": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32]),
    "grp_bv_merge_ranks": (C.c_int, [_vp]),
    "grp_bv_export_words": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp]),
    "grp_bv_or_words": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp]),
    "grp_debug_decide": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp, C.c_uint64, C.POINTER(grp_decide_params), _vp, _vp, _vp]),
    "grp_classify_stream_begin_resumable": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(grp_decide_params), C.c_uint32, C.POINTER(C.c_void_p)]),
    "grp_classify_stream_insert": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "grp_comm_info": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "grp_classify_stream_begin_striped_resumable": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(grp_decide_params), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "grp_classify_stream_resumable": (C.c_int, [_vp, C.c_uint32]),
    "grp_classify_stream_insert_done": (C.c_int, [_vp, C.c_uint32]),
    "grp_batch_insert_reads": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "grp_batch_classify": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(grp_decide_params), _vp, _vp]),
    "grp_batch_verify": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(grp_decide_params), _vp, _vp]),
    "grp_window_overlap": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "grp_debug_verify_stats": (C.c_int, [_vp, _vp]),
    "grp_debug_stream_stats": (C.c_int, [_vp, _vp]),
    "grp_batch_undo": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "grp_batch_end": (C.c_int, [_vp]),
    "grp_insert_tiles": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "grp_insert_read": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "grp_reset_ids": (C.c_int, [_vp]),
    "grp_sync": (C.c_int, [_vp]),
    "grp_filter_bits": (C.c_uint64, [_vp]),
    "grp_pop": (C.c_uint64, [_vp]),
    "grp_export_bits": (C.c_int, [_vp, _vp, C.c_uint64]),
    "grp_rank": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "grp_export_ids": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp, _vp]),
    "grp_import_ids": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp, _vp]),
    "grp_debug_tile_hashes": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "grp_debug_tile_states": (C.c_int, [_vp, C.c_uint64, _vp, _vp]),
    "grp_debug_locate": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp]),
    "grp_dev_hooks": (C.c_int, []),
    "grp_set_timing": (C.c_int, [_vp, C.c_int]),
    "grp_get_kernel_stats": (C.c_int, [_vp, C.POINTER(grp_kernel_stat)]),
    "grp_reset_kernel_stats": (C.c_int, [_vp]),
    "grp_stream": (_vp, [_vp]),
}


class grp_synth_params(C.Structure):
    _fields_ = [("genome_len", C.c_uint64), ("genome_seed", C.c_uint64), ("error_seed", C.c_uint64),
                ("p_sub", C.c_float), ("p_ins", C.c_float), ("p_del", C.c_float), ("repeat_frac", C.c_float)]


# include/grpath_synth.h (measurement support)
SIGNATURES.update({
    "grp_synth_reads": (C.c_int, [C.POINTER(grp_synth_params), _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp]),
    "grp_synth_alloc": (_vp, [C.c_uint64]),
    "grp_synth_free": (None, [_vp]),
    "grp_synth_download": (C.c_int, [_vp, C.c_uint64, _vp]),
    "grp_synth_last_error": (C.c_char_p, []),
})

fastq_record_dtype = np.dtype([("id_off", "<u8"), ("seq_off", "<u8"), ("qual_off", "<u8"), ("id_len", "<u4"), ("seq_len", "<u4"),
                              ("qual_len", "<u4"), ("flags", "<u4"), ("phred_sum", "<f8"), ("phred_first", "<f8")])

# include/grpath_ingest.h
SIGNATURES.update({
    "grp_fastq_prefetch": (C.c_int, [_vp, _vp, C.c_uint64]),
    "grp_fastq_parse": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "grp_fastq_records": (C.c_int, [_vp, _vp]),
    "grp_fastq_pack": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(_vp)]),
    "grp_fastq_free": (None, [_vp]),
    "grp_fastq_pin": (C.c_int, [_vp, _vp, C.c_uint64]),
    "grp_fastq_unpin": (C.c_int, [_vp]),
})

_lib = None


def build(verbose: bool = False) -> str:
    """Compile the gfx950 library in-tree (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", CSRC_DIR], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libgrpath_hip.so failed:\n" + res.stdout + res.stderr)
    if verbose:
        sys.stderr.write(res.stdout)
    return LIB_PATH


# developer builds only (make -C goldrush_amd/csrc DEV=1, include/grpath_dev.h): measurement prototypes, not exported by the product library
DEV_SIGNATURES = {
    "grp_debug_touch_filter": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "grp_pshard_query": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]),
}


def load():
    """dlopen libgrpath_hip.so and type every exported symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GrpError(GRP_ERR_NO_DEVICE, f"{LIB_PATH} is missing: run goldrush_amd.native.build() (no CPU fallback exists)")
    # torch bundles its own libamdhip64.so.7; load it first when torch is around
    # so that one HIP runtime serves both (torch is used for torch.distributed only)
    if "torch" in sys.modules or os.environ.get("GRP_PRELOAD_TORCH", "1") == "1":
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch absent: plain ROCm runtime
            pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.grp_dev_hooks():
        for name, (res, args) in DEV_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------
# 2-bit packing (host side; the CLI has its own C++ packer)
# ---------------------------------------------------------------------------
_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _ch in enumerate(b"ACGT"):
    _CODE[_ch] = _i
    _CODE[_ch + 32] = _i  # lower case


def pack_reads(seqs):
    """seqs: list of bytes (pure ACGT). Returns (packed u32, word_off u64[n+1], len u32[n])."""
    n = len(seqs)
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    words = (lens.astype(np.uint64) + 15) // 16
    word_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(words, out=word_off[1:])
    packed = np.zeros(int(word_off[-1]) + 1, dtype=np.uint32)
    for i, s in enumerate(seqs):
        codes = _CODE[np.frombuffer(s, dtype=np.uint8)]
        if codes.size and codes.max() > 3:
            raise ValueError(f"read {i} contains a non-ACGT base")
        nw = int(words[i])
        buf = np.zeros(nw * 16, dtype=np.uint32)
        buf[: codes.size] = codes
        buf = buf.reshape(nw, 16)
        shifts = (np.arange(16, dtype=np.uint32) * 2)[None, :]
        packed[int(word_off[i]): int(word_off[i]) + nw] = np.bitwise_or.reduce(buf << shifts, axis=1)
    return packed[:-1].copy() if packed.size > 1 else packed[:0].copy(), word_off, lens


class ReadBatch:
    """A batch of packed reads resident in HBM (grp_reads)."""

    def __init__(self, engine: "Engine", handle, n_reads: int, lens: np.ndarray, keep=None):
        self.engine = engine
        self._h = handle
        self.n_reads = n_reads
        self.lens = lens
        self._keep = keep
        p = engine.lib.grp_reads_tile0(handle)
        self.tile0 = np.ctypeslib.as_array(p, shape=(n_reads + 1,)).copy()
        engine._batches.add(self)

    def free(self):
        # a batch must not outlive its engine (grp_reads_free touches the ctx)
        if self._h and self.engine._h:
            self.engine.lib.grp_reads_free(self._h)
        self._h = None
        self.engine._batches.discard(self)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One miBF on one MI355X (grp_ctx)."""

    def __init__(self, k: int, h: int, tile: int, m: int, seeds, device: int = -1):
        self.lib = load()
        self.k, self.h, self.tile, self.m = k, h, tile, m
        arr = (C.c_char_p * h)(*[s.encode() if isinstance(s, str) else s for s in seeds])
        p = grp_params(C.sizeof(grp_params), k, h, tile, m, arr, device, 0)
        out = C.c_void_p()
        rc = self.lib.grp_create(C.byref(p), C.byref(out))
        if rc != GRP_OK:
            raise GrpError(rc, (self.lib.grp_last_error(None) or b"").decode())
        self._h = out
        self.pop = 0
        self._batches = weakref.WeakSet()

    def _check(self, rc):
        if rc != GRP_OK:
            raise GrpError(rc, (self.lib.grp_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            for b in list(self._batches):
                b.free()
            self.lib.grp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reads
    def upload(self, seqs) -> ReadBatch:
        packed, word_off, lens = pack_reads(seqs)
        return self.upload_packed(packed, word_off, lens)

    def upload_packed(self, packed, word_off, lens) -> ReadBatch:
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        word_off = np.ascontiguousarray(word_off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        out = C.c_void_p()
        self._check(self.lib.grp_reads_upload(self._h, _ptr(packed), _ptr(word_off), _ptr(lens), len(lens), C.byref(out)))
        return ReadBatch(self, out, len(lens), lens)

    def wrap_device(self, d_ptr: int, word_off, lens, keep=None) -> ReadBatch:
        word_off = np.ascontiguousarray(word_off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        out = C.c_void_p()
        self._check(self.lib.grp_reads_wrap_device(self._h, C.c_void_p(d_ptr), _ptr(word_off), _ptr(lens), len(lens), C.byref(out)))
        return ReadBatch(self, out, len(lens), lens, keep=keep)

    # -- FASTQ ingest on the device (include/grpath_ingest.h)
    def fastq_parse(self, text: bytes, final_chunk: bool = True):
        """Returns (handle, records[fastq_record_dtype], bytes_consumed, stopped)."""
        buf = np.frombuffer(text, dtype=np.uint8) if len(text) else np.zeros(1, dtype=np.uint8)
        out = C.c_void_p()
        n_rec, used, stopped = C.c_uint64(), C.c_uint64(), C.c_int()
        self._check(self.lib.grp_fastq_parse(self._h, _ptr(buf), len(text), 1 if final_chunk else 0, C.byref(out), C.byref(n_rec), C.byref(used), C.byref(stopped)))
        rec = np.zeros(n_rec.value, dtype=fastq_record_dtype)
        if n_rec.value:
            self._check(self.lib.grp_fastq_records(out, _ptr(rec)))
        return out, rec, used.value, bool(stopped.value)

    def fastq_pack(self, fq_handle, sel, lens) -> "ReadBatch":
        sel = np.ascontiguousarray(sel, dtype=np.uint32)
        out = C.c_void_p()
        self._check(self.lib.grp_fastq_pack(self._h, fq_handle, _ptr(sel), sel.size, C.byref(out)))
        return ReadBatch(self, out, sel.size, np.ascontiguousarray(lens, dtype=np.uint32))

    def fastq_free(self, fq_handle):
        self.lib.grp_fastq_free(fq_handle)

    def fastq_pin(self, buf: np.ndarray):
        """Page-lock `buf` (a uint8 array the caller keeps alive) for the uploads of fastq_parse; unpin before freeing it."""
        self._check(self.lib.grp_fastq_pin(self._h, _ptr(buf), buf.size))

    def fastq_unpin(self):
        self._check(self.lib.grp_fastq_unpin(self._h))

    def fastq_prefetch(self, body: "np.ndarray | None", n_bytes: int = 0) -> bool:
        """grp_fastq_prefetch: the upload of a coming chunk's body (a uint8 view the caller keeps alive and unchanged until its
        parse); False: no device buffer free right now (GRP_ERR_BUSY).  (None, 0): every pending prefetch is forgotten."""
        rc = self.lib.grp_fastq_prefetch(self._h, _ptr(body) if body is not None else None, n_bytes)
        if rc == -6:
            return False
        self._check(rc)
        return True

    def fastq_parse_at(self, buf: np.ndarray, n_bytes: int, final_chunk: bool = True):
        """fastq_parse on the first n_bytes of a caller-owned uint8 array (no copy: what a pinned chunk buffer needs)."""
        out = C.c_void_p()
        n_rec, used, stopped = C.c_uint64(), C.c_uint64(), C.c_int()
        self._check(self.lib.grp_fastq_parse(self._h, _ptr(buf), n_bytes, 1 if final_chunk else 0, C.byref(out), C.byref(n_rec), C.byref(used), C.byref(stopped)))
        rec = np.zeros(n_rec.value, dtype=fastq_record_dtype)
        if n_rec.value:
            self._check(self.lib.grp_fastq_records(out, _ptr(rec)))
        return out, rec, used.value, bool(stopped.value)

    # -- phase 0 (--ntcard)
    def ntcard_begin(self, sbits: int = 7):
        self._check(self.lib.grp_ntcard_begin(self._h, sbits))

    def ntcard_add(self, batch: ReadBatch, first: int = 0, count: int | None = None, stale_extra=None):
        count = batch.n_reads - first if count is None else count
        ex = None
        if stale_extra is not None:
            ex = np.ascontiguousarray(stale_extra, dtype=np.uint32)
            assert ex.size == count * self.h
        self._check(self.lib.grp_ntcard_add(self._h, batch._h, first, count, _ptr(ex) if ex is not None else None))

    def ntcard_finish(self) -> np.ndarray:
        """Zero buckets [h, 2] (seed, sample table)."""
        z = np.zeros(self.h * 2, dtype=np.uint64)
        self._check(self.lib.grp_ntcard_finish(self._h, _ptr(z)))
        return z.reshape(self.h, 2)

    def set_filter_size(self, m: int):
        self._check(self.lib.grp_set_filter_size(self._h, m))
        self.m = m

    # -- phase 1
    def bv_insert(self, batch: ReadBatch, first: int = 0, count: int | None = None):
        count = batch.n_reads - first if count is None else count
        self._check(self.lib.grp_bv_insert(self._h, batch._h, first, count))

    def bv_words(self) -> int:
        n = C.c_uint64()
        self._check(self.lib.grp_bv_words(self._h, C.byref(n)))
        return n.value

    def bv_export_device(self, d_ptr: int):
        self._check(self.lib.grp_bv_export_device(self._h, C.c_void_p(d_ptr)))

    def bv_merge_device(self, d_ptr: int):
        self._check(self.lib.grp_bv_merge_device(self._h, C.c_void_p(d_ptr)))

    def words_or_device(self, d_dst: int, d_src: int, n_words32: int):
        self._check(self.lib.grp_words_or_device(self._h, C.c_void_p(d_dst), C.c_void_p(d_src), n_words32))

    def bv_import_device(self, d_ptr: int):
        self._check(self.lib.grp_bv_import_device(self._h, C.c_void_p(d_ptr)))

    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        rc = self.lib.grp_comm_unique_id(buf, 128)
        if rc != 0:
            raise GrpError(rc, "grp_comm_unique_id failed (librccl.so missing?)")
        return buf.raw

    def comm_init(self, unique_id: bytes, world: int, rank: int):
        self._check(self.lib.grp_comm_init(self._h, unique_id, world, rank))

    def bv_merge_ranks(self):
        self._check(self.lib.grp_bv_merge_ranks(self._h))

    def set_occupancy_hint(self, occupancy: float):
        """the -o the filter was sized for: the phase-2 tables are allocated beside the fill (grp_set_occupancy_hint)"""
        self._check(self.lib.grp_set_occupancy_hint(self._h, float(occupancy)))

    def finalize(self) -> int:
        pop = C.c_uint64()
        self._check(self.lib.grp_finalize(self._h, C.byref(pop)))
        self.pop = pop.value
        return self.pop

    def finalize_times(self) -> dict:
        t = (C.c_double * 6)()
        self._check(self.lib.grp_debug_finalize_times(self._h, t))
        return {"popcount_and_wait_for_fill_s": t[0], "table_allocation_s": t[1], "rank_build_s": t[2], "far_and_overflow_tables_s": t[3],
                "prepared_tables_used": bool(t[4]), "release_plain_vector_s": t[5]}

    # -- phase 2
    def query_tiles(self, batch: ReadBatch, first: int = 0, count: int | None = None, list_cap: int | None = None):
        """Returns (tiles[structured], lists[structured], stats dict)."""
        count = batch.n_reads - first if count is None else count
        if first < 0 or count < 0 or first + count > batch.n_reads:
            raise GrpError(GRP_ERR_INVALID, f"reads [{first}, {first + count}) outside the batch of {batch.n_reads}")
        nt = int(batch.tile0[first + count] - batch.tile0[first])
        tiles = np.zeros(nt, dtype=tile_summary_dtype)
        cap = list_cap if list_cap is not None else max(4 * nt, 1024)
        while True:
            lists = np.zeros(cap, dtype=id_count_dtype)
            used = C.c_uint64()
            st = grp_query_stats()
            rc = self.lib.grp_query_tiles(self._h, batch._h, first, count, _ptr(tiles), _ptr(lists), cap, C.byref(used), C.byref(st))
            if rc == GRP_ERR_NOMEM and used.value > cap:
                cap = int(used.value)
                continue
            self._check(rc)
            return tiles, lists[: used.value], {"queries": st.queries, "hits": st.hits, "misses": st.misses}

    def pshard_query(self, batch: ReadBatch, first: int, count: int, n_owners: int = 8):
        """Measurement: query_tiles' result through the position-sharded form with `n_owners` virtual owners on this one
        device (grp_pshard_query, developer builds only).  Returns (tiles, lists, {"partition_ms", "gather_ms", "vote_ms"})."""
        if not self.lib.grp_dev_hooks():
            raise GrpError(GRP_ERR_STATE, "pshard_query: libgrpath_hip.so is the product build; the prototype needs make -C goldrush_amd/csrc DEV=1")
        nt = int(batch.tile0[first + count] - batch.tile0[first])
        tiles = np.zeros(nt, dtype=tile_summary_dtype)
        cap = max(8 * nt, 1024)
        lists = np.zeros(cap, dtype=id_count_dtype)
        used = C.c_uint64()
        times = (C.c_float * 3)()
        self._check(self.lib.grp_pshard_query(self._h, batch._h, first, count, n_owners, _ptr(tiles), _ptr(lists), cap, C.byref(used), times))
        if used.value > cap:
            raise GrpError(GRP_ERR_NOMEM, "pshard_query: %d list entries, room for %d" % (used.value, cap))
        return tiles, lists[: used.value], {"partition_ms": times[0], "gather_ms": times[1], "vote_ms": times[2]}

    def classify_reads(self, batch: ReadBatch, first: int = 0, count: int | None = None, threshold=10, unassigned_min=5, assigned_max=1):
        """Query + decision on the device; returns an array of decision_dtype."""
        count = batch.n_reads - first if count is None else count
        if first < 0 or count < 0 or first + count > batch.n_reads:
            raise GrpError(GRP_ERR_INVALID, f"reads [{first}, {first + count}) outside the batch of {batch.n_reads}")
        out = np.zeros(count, dtype=decision_dtype)
        dp = grp_decide_params(threshold, unassigned_min, assigned_max, 0)
        self._check(self.lib.grp_classify_reads(self._h, batch._h, first, count, C.byref(dp), _ptr(out)))
        return out

    def debug_decide(self, tile0, tiles, lists, threshold=10, unassigned_min=5, assigned_max=1):
        """k_decide on given tile summaries (grp_debug_decide): -> (decisions, per-tile ids, per-tile assigned flags)."""
        tile0 = np.ascontiguousarray(tile0, dtype=np.uint64)
        n = tile0.size - 1
        tiles = np.ascontiguousarray(tiles, dtype=tile_summary_dtype)
        lists = np.ascontiguousarray(lists, dtype=id_count_dtype)
        nt = int(tile0[-1])
        out = np.zeros(n, dtype=decision_dtype)
        ids = np.zeros(max(nt, 1), dtype=np.uint32)
        asg = np.zeros(max(nt, 1), dtype=np.uint8)
        dp = grp_decide_params(threshold, unassigned_min, assigned_max, 0)
        self._check(self.lib.grp_debug_decide(self._h, n, _ptr(tile0), _ptr(tiles), _ptr(lists), lists.size, C.byref(dp), _ptr(out), _ptr(ids), _ptr(asg)))
        return out, ids[:nt], asg[:nt]

    def classify_begin(self, batch: ReadBatch, first: int, count: int, slot: int, threshold=10, unassigned_min=5, assigned_max=1):
        """Enqueue query + decision of a window in slot 0 / 1 (asynchronous)."""
        if first < 0 or count < 0 or first + count > batch.n_reads:
            raise GrpError(GRP_ERR_INVALID, f"reads [{first}, {first + count}) outside the batch of {batch.n_reads}")
        dp = grp_decide_params(threshold, unassigned_min, assigned_max, 0)
        self._check(self.lib.grp_classify_reads_begin(self._h, batch._h, first, count, C.byref(dp), slot))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[slot] = count

    def classify_end(self, slot: int, abandon: bool = False):
        """Wait for the slot's window and return its decisions (None when abandoned)."""
        count = getattr(self, "_inflight", {}).pop(slot, 0)
        if abandon:
            self._check(self.lib.grp_classify_reads_end(self._h, slot, None))
            return None
        out = np.zeros(count, dtype=decision_dtype)
        self._check(self.lib.grp_classify_reads_end(self._h, slot, _ptr(out)))
        return out

    def stream_begin(self, batch: ReadBatch, first: int, count: int, slot: int, threshold=10, unassigned_min=5, assigned_max=1,
                     stripe: int = 0, n_owners: int = 1, owner: int = 0, resumable: bool = False) -> np.ndarray:
        """Start a streaming window; returns a live view of the decision records
        (record j is complete once its "pad" field reads 1 — the record's generation, +1 per
        stream_insert of a resumable window).  n_owners > 1: only the
        stripes of `owner` are worked on (grp_classify_stream_begin_striped)."""
        dp = grp_decide_params(threshold, unassigned_min, assigned_max, 0)
        ptr = C.c_void_p()
        if resumable and n_owners > 1:  # round 5: this rank's stripes of a window that applies inserts itself
            self._check(self.lib.grp_classify_stream_begin_striped_resumable(self._h, batch._h, first, count, C.byref(dp), slot, stripe, n_owners, owner, C.byref(ptr)))
        elif resumable:
            self._check(self.lib.grp_classify_stream_begin_resumable(self._h, batch._h, first, count, C.byref(dp), slot, C.byref(ptr)))
        elif n_owners > 1:
            self._check(self.lib.grp_classify_stream_begin_striped(self._h, batch._h, first, count, C.byref(dp), slot, stripe, n_owners, owner, C.byref(ptr)))
        else:
            self._check(self.lib.grp_classify_stream_begin(self._h, batch._h, first, count, C.byref(dp), slot, C.byref(ptr)))
        if count == 0:
            return np.zeros(0, dtype=decision_dtype)
        buf = (C.c_uint8 * (count * decision_dtype.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=decision_dtype)

    def stream_abort(self, slot: int):
        self._check(self.lib.grp_classify_stream_abort(self._h, slot))

    def stream_poll(self, slot: int) -> bool:
        rc = self.lib.grp_classify_stream_poll(self._h, slot)
        if rc < 0:
            self._check(rc)
        return rc == 1

    def stream_end(self, slot: int) -> int:
        n = C.c_uint32()
        rc = self.lib.grp_classify_stream_end(self._h, slot, C.byref(n))
        if rc in (1, 2):
            raise RuntimeError("grp_classify_stream_end: the insert posted last was not applied by the launch (%d)" % rc)
        self._check(rc)
        return n.value

    def stream_insert(self, slot: int, read_idx: int, tile_start: int, tile_end: int, block: int, first_id: int, id_offset: int) -> int:
        """The window in `slot` applies this insert itself (it is parked at the read's record) and
        carries on behind the read; returns the generation (.pad) of the records that follow."""
        gen = C.c_uint32()
        self._check(self.lib.grp_classify_stream_insert(self._h, slot, read_idx, tile_start, tile_end, block, first_id, id_offset, C.byref(gen)))
        return gen.value

    def stream_resumable(self, slot: int) -> bool:
        return self.lib.grp_classify_stream_resumable(self._h, slot) == 1

    def stream_insert_done(self, slot: int) -> int:
        """1: the insert posted last has been applied, 0: not yet, 2: the launch ended without it"""
        rc = self.lib.grp_classify_stream_insert_done(self._h, slot)
        if rc < 0:
            self._check(rc)
        return rc

    def batch_insert_reads(self, batch: ReadBatch, inserts, block: int, first_read: int):
        """inserts: list of (read, tile_start, tile_end, first_id, id_offset), ascending reads"""
        arr = np.ascontiguousarray(np.array(inserts, dtype=np.uint32).reshape(-1, 5))
        self._check(self.lib.grp_batch_insert_reads(self._h, batch._h, _ptr(arr), arr.shape[0], block, first_read))

    def batch_classify(self, batch: ReadBatch, first: int, count: int, id_floor, threshold=10, unassigned_min=5, assigned_max=1):
        out = np.zeros(count, dtype=decision_dtype)
        fl = np.ascontiguousarray(id_floor, dtype=np.uint32)
        dp = grp_decide_params(threshold, unassigned_min, assigned_max, 0)
        self._check(self.lib.grp_batch_classify(self._h, batch._h, first, count, C.byref(dp), _ptr(fl), _ptr(out)))
        return out

    def batch_verify(self, batch: ReadBatch, first: int, count: int, extra: int, id_floor, threshold=10, unassigned_min=5, assigned_max=1):
        """the second decisions of the batch's reads [first, first + count) from the batch's records (and `extra`
        reads behind them through the plain query); id_floor: count + extra entries"""
        out = np.zeros(count + extra, dtype=decision_dtype)
        fl = np.ascontiguousarray(id_floor, dtype=np.uint32)
        assert fl.shape[0] == count + extra
        dp = grp_decide_params(threshold, unassigned_min, assigned_max, 0)
        self._check(self.lib.grp_batch_verify(self._h, batch._h, first, count, extra, C.byref(dp), _ptr(fl), _ptr(out)))
        return out

    def window_overlap(self, batch: ReadBatch, first: int, count: int, threshold: int = 4) -> np.ndarray:
        """grp_window_overlap: per read of [first, first + count) the closest read in front of it (index relative to
        first) that owns >= threshold of its sampled k-mers; 0xFFFFFFFF: none."""
        out = np.full(max(count, 1), 0xFFFFFFFF, dtype=np.uint32)
        self._check(self.lib.grp_window_overlap(self._h, batch._h, first, count, threshold, _ptr(out)))
        return out[:count]

    def verify_stats(self) -> dict:
        out = np.zeros(12, dtype=np.uint64)
        self._check(self.lib.grp_debug_verify_stats(self._h, _ptr(out)))
        return dict(zip(("patched", "queried", "flagged", "fallbacks", "uncertified", "unpatched", "window_flagged", "flagged_distinct", "flagged_list", "claim_sweeps", "impossible_deltas", "far_count_words"), (int(x) for x in out)))

    def stream_stats(self) -> dict:
        """what the in-launch inserts of the streaming windows did with the tiles queried behind the inserting read"""
        out = np.zeros(8, dtype=np.uint64)
        self._check(self.lib.grp_debug_stream_stats(self._h, _ptr(out)))
        names = ("coop_refused", "tiles_kept", "tiles_redone_dirty", "tiles_redone_lost", None, None, "inserts_kept_nothing", "inserts_kept")
        return {k: int(x) for k, x in zip(names, out) if k}

    def batch_undo(self, from_read: int, id_floor: int):
        """takes back the inserts of reads >= from_read (batch index); id_floor = the first ID read from_read could allocate"""
        self._check(self.lib.grp_batch_undo(self._h, from_read, id_floor))

    def batch_end(self):
        self._check(self.lib.grp_batch_end(self._h))

    def tile_states(self, n_tiles: int):
        """(ids, assigned) per tile after the smoothing passes of the last classify_reads window."""
        ids = np.zeros(max(n_tiles, 1), dtype=np.uint32)
        asg = np.zeros(max(n_tiles, 1), dtype=np.uint8)
        self._check(self.lib.grp_debug_tile_states(self._h, n_tiles, _ptr(ids), _ptr(asg)))
        return ids[:n_tiles], asg[:n_tiles]

    def insert_tiles(self, batch: ReadBatch, read_idx: int, tile_start: int, tile_end: int, id_: int):
        self._check(self.lib.grp_insert_tiles(self._h, batch._h, read_idx, tile_start, tile_end, id_))

    def insert_read(self, batch: ReadBatch, read_idx: int, tile_start: int, tile_end: int, block_tiles: int, first_id: int, id_offset: int = 0):
        self._check(self.lib.grp_insert_read(self._h, batch._h, read_idx, tile_start, tile_end, block_tiles, first_id, id_offset))

    def reset_ids(self):
        self._check(self.lib.grp_reset_ids(self._h))

    def sync(self):
        self._check(self.lib.grp_sync(self._h))

    # -- inspection
    def export_bits(self) -> np.ndarray:
        n = (self.m + 63) // 64
        out = np.zeros(n, dtype=np.uint64)
        self._check(self.lib.grp_export_bits(self._h, _ptr(out), n))
        return out

    def rank(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        bit = np.zeros(pos.size, dtype=np.uint8)
        rank = np.zeros(pos.size, dtype=np.uint64)
        self._check(self.lib.grp_rank(self._h, _ptr(pos), pos.size, _ptr(bit), _ptr(rank)))
        return bit, rank

    def export_ids(self, first: int = 0, n: int | None = None):
        n = self.pop - first if n is None else n
        ids = np.zeros(n, dtype=np.uint32)
        counts = np.zeros(n, dtype=np.uint32)
        self._check(self.lib.grp_export_ids(self._h, first, n, _ptr(ids), _ptr(counts)))
        return ids, counts

    def import_ids(self, first: int, ids=None, counts=None):
        n = len(ids) if ids is not None else len(counts)
        ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
        counts = None if counts is None else np.ascontiguousarray(counts, dtype=np.uint32)
        self._check(self.lib.grp_import_ids(self._h, first, n, _ptr(ids), _ptr(counts)))

    def tile_hashes(self, batch: ReadBatch, read_idx: int, tile_idx: int) -> np.ndarray:
        cap = (self.tile + 1) * self.h
        out = np.zeros(cap, dtype=np.uint64)
        nv = C.c_uint64()
        self._check(self.lib.grp_debug_tile_hashes(self._h, batch._h, read_idx, tile_idx, _ptr(out), cap, C.byref(nv)))
        return out[: nv.value]

    def debug_locate(self, x, m: int, W: int, on_device: bool = True):
        """(x % m, (x % m) // W) through the kernels' reciprocal / magic-number shortcuts."""
        return debug_locate(x, m, W, on_device, self)

    # -- measurement
    def kernel_stats(self) -> dict:
        arr = (grp_kernel_stat * GRP_K_COUNT)()
        self._check(self.lib.grp_get_kernel_stats(self._h, arr))
        return {KERNEL_NAMES[i]: {"launches": arr[i].launches, "units": arr[i].units, "ms": arr[i].ms} for i in range(GRP_K_COUNT)}

    def reset_kernel_stats(self):
        self._check(self.lib.grp_reset_kernel_stats(self._h))

    def set_timing(self, on: bool):
        self._check(self.lib.grp_set_timing(self._h, 1 if on else 0))


def debug_locate(x, m: int, W: int, on_device: bool = False, engine: "Engine | None" = None):
    """grp_debug_locate: the host instantiation needs no engine (and no GPU)."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.uint64)
    mod = np.zeros(x.size, dtype=np.uint64)
    div = np.zeros(x.size, dtype=np.uint64)
    rc = lib.grp_debug_locate(engine._h if engine is not None else None, _ptr(x), x.size, m, W, 1 if on_device else 0, _ptr(mod), _ptr(div))
    if rc != GRP_OK:
        raise GrpError(rc, (lib.grp_last_error(engine._h if engine is not None else None) or b"").decode())
    return mod, div


# ---------------------------------------------------------------------------
# synthetic reads generated on the GPU (include/grpath_synth.h)
# ---------------------------------------------------------------------------
class DeviceReads:
    """Packed synthetic reads living in a hipMalloc'ed buffer."""

    def __init__(self, d_ptr: int, word_off: np.ndarray, lens: np.ndarray):
        self.d_ptr, self.word_off, self.lens = d_ptr, word_off, lens

    def download(self, first: int, count: int):
        """ASCII sequences of reads [first, first+count) (for the CPU baseline / checks)."""
        lib = load()
        w0, w1 = int(self.word_off[first]), int(self.word_off[first + count])
        buf = np.zeros(max(w1 - w0, 1), dtype=np.uint32)
        rc = lib.grp_synth_download(C.c_void_p(self.d_ptr + 4 * w0), 4 * (w1 - w0), _ptr(buf))
        if rc != 0:
            raise GrpError(rc, lib.grp_synth_last_error().decode())
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        out = []
        for i in range(first, first + count):
            a = int(self.word_off[i]) - w0
            nw = (int(self.lens[i]) + 15) // 16
            words = buf[a:a + nw]
            codes = ((words[:, None] >> (np.arange(16, dtype=np.uint32) * 2)[None, :]) & 3).reshape(-1)[: int(self.lens[i])]
            out.append(acgt[codes].tobytes())
        return out

    def free(self):
        if self.d_ptr:
            load().grp_synth_free(C.c_void_p(self.d_ptr))
            self.d_ptr = 0


def synth_read_plan(n_reads: int, genome_len: int, mean_len: int = 25000, min_len: int = 20000, sigma: float = 0.25, seed: int = 2,
                    max_len: int | None = None):
    """Host-side plan (start, len, strand, word_off) of a synthetic read set."""
    rng = np.random.default_rng(seed)
    mu = np.log(mean_len) - 0.5 * sigma * sigma
    lens = np.maximum(rng.lognormal(mu, sigma, size=n_reads).astype(np.int64), min_len)
    if max_len is not None:
        lens = np.minimum(lens, max_len)
    lens = lens.astype(np.uint32)
    start = rng.integers(0, genome_len, size=n_reads, dtype=np.uint64)
    strand = (rng.random(n_reads) < 0.5).astype(np.uint8)
    word_off = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum((lens.astype(np.uint64) + 15) // 16, out=word_off[1:])
    return start, lens, strand, word_off


def synth_reads_range(plan, lo: int, hi: int, genome_len: int, genome_seed: int = 1, error_seed: int = 3, p_sub=0.03, p_ins=0.01, p_del=0.01, stream: int = 0, repeat_frac: float = 0.0) -> DeviceReads:
    """Reads [lo, hi) of a plan (synth_read_plan) generated into their own device buffer: a read
    set larger than HBM is streamed through it batch by batch.  The error pattern of a read
    depends on its index INSIDE the call, so the passes of one run must use the same batches."""
    lib = load()
    start, lens, strand, word_off = plan
    n = hi - lo
    wo = np.ascontiguousarray(word_off[lo:hi + 1] - word_off[lo], dtype=np.uint64)
    st = np.ascontiguousarray(start[lo:hi])
    ln = np.ascontiguousarray(lens[lo:hi])
    sd = np.ascontiguousarray(strand[lo:hi])
    d = lib.grp_synth_alloc(int(wo[-1]) * 4 + 64)
    if not d:
        raise GrpError(GRP_ERR_NOMEM, lib.grp_synth_last_error().decode())
    p = grp_synth_params(genome_len, genome_seed, error_seed, p_sub, p_ins, p_del, repeat_frac)
    rc = lib.grp_synth_reads(C.byref(p), _ptr(st), _ptr(ln), _ptr(sd), _ptr(wo), n, C.c_void_p(d), C.c_void_p(stream))
    if rc != 0:
        raise GrpError(rc, lib.grp_synth_last_error().decode())
    return DeviceReads(d, wo, ln)


def synth_reads(n_reads: int, genome_len: int, genome_seed: int = 1, error_seed: int = 3, p_sub=0.03, p_ins=0.01, p_del=0.01,
                stream: int = 0, repeat_frac: float = 0.0, **plan_kw) -> DeviceReads:
    lib = load()
    start, lens, strand, word_off = synth_read_plan(n_reads, genome_len, **plan_kw)
    d = lib.grp_synth_alloc(int(word_off[-1]) * 4 + 64)
    if not d:
        raise GrpError(GRP_ERR_NOMEM, lib.grp_synth_last_error().decode())
    p = grp_synth_params(genome_len, genome_seed, error_seed, p_sub, p_ins, p_del, repeat_frac)
    rc = lib.grp_synth_reads(C.byref(p), _ptr(start), _ptr(lens), _ptr(strand), _ptr(word_off), n_reads, C.c_void_p(d), C.c_void_p(stream))
    if rc != 0:
        raise GrpError(rc, lib.grp_synth_last_error().decode())
    return DeviceReads(d, word_off, lens)
