"""Test infrastructure: an engine function table (include/grpath_host.h,
grp_engine_vt) backed by the CPU oracle, so the product's order-exact
classifier can be exercised without a GPU (CPU tests, gloo world_size-2)."""
import ctypes as C

import zlib

import numpy as np

from goldrush_amd import host, native


class OracleEngine:
    def __init__(self, orc, m, seeds, tile, k, reads, pipelined=False, streaming=False, redo_every=0, batching=False, batch_crowded_above=0,
                 resume=False, resume_refuse_every=0, resume_lost_every=0, batch_verify=True, overlap_every=0):
        self.overlap_every = overlap_every  # grp_window_overlap: every n-th read of the stream "overlaps" (0: the engine has no such call)
        self.n_overlap_calls = 0
        self.batch_verify = batch_verify  # with batching: the engine has grp_batch_verify (the second decisions + the reads behind the batch)
        self.n_verifies = 0
        self.pipelined = pipelined
        self.streaming = streaming
        self.resume = resume                          # stream_insert: the window applies the insert itself and carries on
        self.resume_refuse_every = resume_refuse_every  # every n-th stream_insert is refused (GRP_ERR_STATE)
        self.resume_lost_every = resume_lost_every      # every n-th one is accepted but never applied (the launch "timed out")
        self.n_stream_inserts = self.n_stream_refused = self.n_stream_lost = 0
        self.batching = batching      # classify_reads / insert_read / batch_* (windows committed as batches)
        self.batch_crowded_above = batch_crowded_above  # batches of more inserts are refused (GRP_ERR_NOMEM)
        self.n_batches = self.n_batch_undone = self.n_batch_refused = 0
        self.redo_every = redo_every  # every n-th streamed record asks for the synchronous path (kind 0)
        self.n_streams = 0
        self.n_stream_aborts = 0
        self.n_redo = 0
        self.orc = orc
        self.oseeds = orc.Seeds(seeds)
        self.mf = orc.MiBF(m, self.oseeds, tile, k)
        self.tile, self.k = tile, k
        self.reads = reads
        self.n_queries = 0
        self.n_begun = 0
        self.n_abandoned = 0
        for s in reads:
            self.mf.bv_insert_read(s)
        self.mf.finalize()
        self._keep = []
        self.vt = self._make_vt()

    def _make_vt(self):
        vt = host.grp_engine_vt()
        types = dict(host.VT_TYPES)

        def _window(first, count):  # the largest window (in tiles) any call was handed
            nt = sum(len(self.reads[r]) // self.tile for r in range(first, first + count))
            self.max_window_tiles = max(getattr(self, "max_window_tiles", 0), nt)

        def query_tiles(ctx, reads, first, count, tiles_p, lists_p, cap, used_p, stats_p):
            _window(first, count)
            self.n_queries += count
            res = []
            for r in range(first, first + count):
                res.extend(self.mf.query_read(self.reads[r]))
            nt = len(res)
            need = sum(len(x[2]) for x in res)
            used_p[0] = need
            if need > cap:
                return native.GRP_ERR_NOMEM
            tiles = np.ctypeslib.as_array(C.cast(tiles_p, C.POINTER(C.c_uint32)), shape=(max(nt, 1) * 6,)).view(native.tile_summary_dtype)
            lists = np.ctypeslib.as_array(C.cast(lists_p, C.POINTER(C.c_uint32)), shape=(max(cap, 1) * 2,)).view(native.id_count_dtype)
            off = 0
            for i, (tid, tc, lst, ctr) in enumerate(res):
                lst = sorted(((int(a), int(b)) for a, b in lst), key=lambda t: (-t[1], t[0]))
                tiles[i] = (tid, tc, off, len(lst), ctr[1], ctr[2])
                for j, (a, b) in enumerate(lst):
                    lists[off + j] = (a, b)
                off += len(lst)
            return 0

        # pipelined windows: the decisions are taken at _begin (the GPU engine runs the
        # window in stream order, i.e. before any insert issued later) and handed out
        # at _end; a NULL `out` abandons the window
        slots = {}

        def classify_begin(ctx, reads, first, count, dp_p, slot):
            assert slot in (0, 1) and slot not in slots, "slot busy"
            dp = C.cast(dp_p, C.POINTER(native.grp_decide_params))[0]
            _window(first, count)
            self.n_queries += count
            self.n_begun += 1
            out = []
            for r in range(first, first + count):
                res = self.mf.query_read(self.reads[r])
                lists = [sorted(((int(a), int(b)) for a, b in x[2]), key=lambda t: (-t[1], t[0])) for x in res]
                tiles, flat = host.tiles_from([x[0] for x in res], lists)
                d = host.decide_read(tiles, flat, len(res), dp.threshold, dp.unassigned_min, dp.assigned_max)
                d.hits = sum(int(x[3][1]) for x in res)
                d.misses = sum(int(x[3][2]) for x in res)
                out.append(d)
            slots[slot] = out
            return 0

        def classify_end(ctx, slot, out_p):
            assert slot in slots, "no window in flight"
            decs = slots.pop(slot)
            if not out_p:
                self.n_abandoned += 1
                return 0
            arr = C.cast(out_p, C.POINTER(host.gr_read_decision))
            for i, d in enumerate(decs):
                arr[i] = d
            return 0

        # streaming windows: all records are complete at _begin (decided against the state
        # at that moment, like a launch that ran to the end before the host looked)
        streams = {}
        stream_info = {}

        def _decide(r, dp):
            res = self.mf.query_read(self.reads[r])
            lists = [sorted(((int(a), int(b)) for a, b in x[2]), key=lambda t: (-t[1], t[0])) for x in res]
            tiles, flat = host.tiles_from([x[0] for x in res], lists)
            d = host.decide_read(tiles, flat, len(res), dp.threshold, dp.unassigned_min, dp.assigned_max)
            d.hits = sum(int(x[3][1]) for x in res)
            d.misses = sum(int(x[3][2]) for x in res)
            return d

        def stream_begin_resumable(ctx, reads, first, count, dp_p, slot, out_pp):
            if streams and self.resume_refuse_every:  # a window behind one in flight: "not now" every other time (GRP_ERR_BUSY)
                self.n_busy_calls = getattr(self, "n_busy_calls", 0) + 1
                if self.n_busy_calls % 2 == 1:
                    self.n_stream_busy = getattr(self, "n_stream_busy", 0) + 1
                    return native.GRP_ERR_BUSY
            rc = stream_begin(ctx, reads, first, count, dp_p, slot, 0, 1, 0, out_pp)
            stream_info[slot]["resumable"] = True
            return rc

        # round 5: this rank's stripes of a window that applies inserts itself (grp_classify_stream_begin_striped_resumable)
        def stream_begin_striped_resumable(ctx, reads, first, count, dp_p, slot, stripe, n_owners, owner, out_pp):
            rc = stream_begin(ctx, reads, first, count, dp_p, slot, stripe, n_owners, owner, out_pp)
            self.n_striped_resumable = getattr(self, "n_striped_resumable", 0) + 1
            # every n-th window is one the runtime "refused to launch cooperatively": it ends where it parks (the ranks must agree)
            stream_info[slot]["resumable"] = not (self.resume_refuse_every and self.n_striped_resumable % self.resume_refuse_every == 0)
            return rc

        def stream_resumable(ctx, slot):
            return 1 if slot in streams and stream_info[slot]["resumable"] else 0

        def stream_insert_done(ctx, slot):
            assert slot in streams
            return 2 if stream_info[slot]["lost"] else 1

        def stream_begin(ctx, reads, first, count, dp_p, slot, stripe, n_owners, owner, out_pp):
            assert slot in (0, 1) and slot not in streams and slot not in slots, "slot busy"
            dp = C.cast(dp_p, C.POINTER(native.grp_decide_params))[0]
            arr = (host.gr_read_decision * max(count, 1))()
            parked = False
            for j in range(count):
                if n_owners > 1 and (j // stripe) % n_owners != owner:
                    continue  # another rank's stripe: the record stays at pad = 0
                if parked:
                    continue  # the window parked itself behind an insert / hand-back record: never decided
                self.n_stream_records = getattr(self, "n_stream_records", 0) + 1
                if self.redo_every and self.n_stream_records % self.redo_every == 0:
                    arr[j] = host.gr_read_decision()  # kind 0
                    self.n_redo += 1
                else:
                    arr[j] = _decide(first + j, dp)
                arr[j].pad = 1
                parked = arr[j].kind in (0, 2, 4)
            self.n_queries += sum(1 for j in range(count) if arr[j].pad)
            self.n_streams += 1
            streams[slot] = (arr, count)
            stream_info[slot] = dict(first=first, dp=(dp.threshold, dp.unassigned_min, dp.assigned_max), striped=n_owners > 1, gen=1, lost=False, resumable=False, stripe=stripe, n_owners=n_owners, owner=owner)
            C.cast(out_pp, C.POINTER(C.c_void_p))[0] = C.addressof(arr)
            return 0

        # grp_classify_stream_insert: the parked window applies the insert and decides the reads behind
        # it again (new generation in .pad); older records stay where they are, stale
        def stream_insert(ctx, slot, read_idx, ts, te, block, first_id, off, gen_p):
            assert slot in streams
            info = stream_info[slot]
            arr, count = streams[slot]
            x = read_idx - info["first"]
            mine = lambda j: not info["striped"] or (j // info["stripe"]) % info["n_owners"] == info["owner"]  # noqa: E731
            assert 0 <= x < count
            if mine(x):
                assert arr[x].pad == info["gen"] and arr[x].kind in (2, 4), "the window is not parked at this read"
            self.n_stream_insert_calls = getattr(self, "n_stream_insert_calls", 0) + 1
            if not info["resumable"] or (not info["striped"] and self.resume_refuse_every and self.n_stream_insert_calls % self.resume_refuse_every == 0):
                self.n_stream_refused += 1
                return native.GRP_ERR_STATE
            info["gen"] += 1
            C.cast(gen_p, C.POINTER(C.c_uint32))[0] = info["gen"]
            if self.resume_lost_every and self.n_stream_insert_calls % self.resume_lost_every == 0:
                self.n_stream_lost += 1
                info["lost"] = True  # accepted, never applied: no record of the new generation will come
                return 0
            insert_read(ctx, None, read_idx, ts, te, block, first_id, off)
            self.n_stream_inserts += 1
            dp = native.grp_decide_params(*info["dp"], 0)
            parked = False
            for j in range(x + 1, count):
                if parked:
                    break
                if not mine(j):
                    continue  # another rank's stripe
                self.n_stream_records = getattr(self, "n_stream_records", 0) + 1
                if self.redo_every and self.n_stream_records % self.redo_every == 0:
                    arr[j] = host.gr_read_decision()
                    self.n_redo += 1
                else:
                    arr[j] = _decide(info["first"] + j, dp)
                arr[j].pad = info["gen"]
                self.n_queries += 1
                parked = arr[j].kind in (0, 2, 4)
            # a window queued behind this one has not started (it runs when this launch ends): it
            # will see the insert — its records, computed eagerly at _begin here, are decided again
            for o, (oarr, ocount) in streams.items():
                if o == slot:
                    continue
                oinfo = stream_info[o]
                assert oinfo["gen"] == 1
                odp = native.grp_decide_params(*oinfo["dp"], 0)
                parked = False
                for j in range(ocount):
                    oarr[j] = host.gr_read_decision()
                    if parked or (oinfo["striped"] and (j // oinfo["stripe"]) % oinfo["n_owners"] != oinfo["owner"]):
                        continue
                    oarr[j] = _decide(oinfo["first"] + j, odp)
                    oarr[j].pad = 1
                    parked = oarr[j].kind in (0, 2, 4)
            return 0

        def stream_abort(ctx, slot):
            assert slot in streams
            self.n_stream_aborts += 1
            return 0

        def stream_poll(ctx, slot):
            assert slot in streams
            return 1

        def stream_end(ctx, slot, n_p):
            arr, count = streams.pop(slot)
            info = stream_info.pop(slot)
            self._keep_last = getattr(self, "_keep_last", {})
            self._keep_last[slot] = arr  # valid until the slot's next _begin
            if n_p:
                C.cast(n_p, C.POINTER(C.c_uint32))[0] = count
            return 1 if info["lost"] else 0

        def insert_tiles(ctx, reads, ri, ts, te, id_):
            self.mf.insert_read_tiles(self.reads[ri], ts, te, id_)
            return 0

        # grp_insert_read: blocks of `block` tiles from tile_start, tile t gets first_id + (t - tile_start + id_offset) / block
        def insert_read(ctx, reads, ri, ts, te, block, first_id, off):
            bs = ts
            while bs < te:
                self.mf.insert_read_tiles(self.reads[ri], bs, min(bs + block, te), first_id + (bs - ts + off) // block)
                bs += block
            return 0

        def classify_reads(ctx, reads, first, count, dp_p, out_p):
            dp = C.cast(dp_p, C.POINTER(native.grp_decide_params))[0]
            arr = C.cast(out_p, C.POINTER(host.gr_read_decision))
            _window(first, count)
            self.n_queries += count
            for j in range(count):
                arr[j] = _decide(first + j, dp)
            return 0

        # windows committed as batches (include/grpath.h, grp_batch_*): the "state in front of
        # read j's own insert" is produced literally — back to the state in front of the
        # window, then read by read
        bt = {}

        def batch_insert(ctx, reads, ins_p, n_ins, block, first_read):
            assert not bt, "one batch at a time"
            if self.batch_crowded_above and n_ins > self.batch_crowded_above:
                self.n_batch_refused += 1
                return native.GRP_ERR_NOMEM
            ins = np.ctypeslib.as_array(C.cast(ins_p, C.POINTER(C.c_uint32)), shape=(n_ins * 5,)).reshape(n_ins, 5).copy()
            assert all(ins[i, 0] < ins[i + 1, 0] for i in range(n_ins - 1)) and ins[0, 0] >= first_read
            bt.update(ins=ins, first=first_read, block=block, ids0=self.mf.ids().copy(), counts0=self.mf.counts().copy())
            for e in ins:
                insert_read(ctx, reads, int(e[0]), int(e[1]), int(e[2]), block, int(e[3]), int(e[4]))
            self.n_batches += 1
            return 0

        def _batch_restore():
            self.mf.ids()[:] = bt["ids0"]
            self.mf.counts()[:] = bt["counts0"]

        def batch_classify(ctx, reads, first, count, dp_p, floor_p, out_p):
            # any range of the window's reads (a rank's stripe): read by read from the window's first
            assert bt and first >= bt["first"]
            dp = C.cast(dp_p, C.POINTER(native.grp_decide_params))[0]
            floors = np.ctypeslib.as_array(C.cast(floor_p, C.POINTER(C.c_uint32)), shape=(max(count, 1),))
            arr = C.cast(out_p, C.POINTER(host.gr_read_decision))
            by_read = {int(e[0]): e for e in bt["ins"]}
            _batch_restore()
            self.n_queries += count
            last = max([first + count - 1] + list(by_read))
            for r in range(bt["first"], last + 1):
                if first <= r < first + count:
                    arr[r - first] = _decide(r, dp)
                e = by_read.get(r)
                if e is not None:
                    if first <= r < first + count:
                        assert int(e[3]) == int(floors[r - first]) & 0x7FFFFFFF  # a read's first ID is the first one it could allocate (bit 31: see grpath.h)
                    insert_read(ctx, reads, int(e[0]), int(e[1]), int(e[2]), bt["block"], int(e[3]), int(e[4]))
            return 0

        def batch_verify(ctx, reads, first, count, extra, dp_p, floor_p, out_p):
            # the same answers as batch_classify over count + extra reads (the reads behind the last insert see the
            # filter as the batch leaves it); what the GPU engine does differently is HOW it gets them
            assert bt and first == bt["first"] and count > 0
            self.n_verifies += 1
            return batch_classify(ctx, reads, first, count + extra, dp_p, floor_p, out_p)

        def window_overlap(ctx, reads, first, count, threshold, out_p):
            # per read of the range the closest read in front of it that it "overlaps": every n-th read of the stream
            # overlaps the read two places in front of it (right or wrong, the commits must not depend on it)
            self.n_overlap_calls += 1
            out = C.cast(out_p, C.POINTER(C.c_uint32))
            for j in range(count):
                out[j] = 0xFFFFFFFF
                if j >= 2 and (first + j) % self.overlap_every == 0:
                    out[j] = j - 2
            return 0

        def batch_undo(ctx, from_read, floor_id):
            assert bt and from_read >= bt["first"]
            later = [e for e in bt["ins"] if int(e[0]) >= from_read]
            assert not later or floor_id <= int(later[0][3])
            _batch_restore()
            for e in bt["ins"]:  # the inserts in front of from_read stay
                if int(e[0]) < from_read:
                    insert_read(ctx, None, int(e[0]), int(e[1]), int(e[2]), bt["block"], int(e[3]), int(e[4]))
            bt.clear()
            self.n_batch_undone += 1
            return 0

        def batch_end(ctx):
            bt.clear()
            return 0

        def reset_ids(ctx):
            self.mf.reset_ids()
            return 0

        def sync(ctx):
            return 0

        self._err_text = C.create_string_buffer(b"oracle engine")  # kept by the engine: the table hands out its address

        def last_error(ctx):
            return C.addressof(self._err_text)

        impl = {}
        if self.pipelined:
            impl.update({"classify_begin": classify_begin, "classify_end": classify_end})
        if self.streaming:
            impl.update({"stream_begin": stream_begin, "stream_abort": stream_abort, "stream_poll": stream_poll, "stream_end": stream_end})
            if self.resume:
                impl.update({"stream_begin_resumable": stream_begin_resumable, "stream_insert": stream_insert, "insert_read": insert_read,
                             "stream_begin_striped_resumable": stream_begin_striped_resumable, "stream_resumable": stream_resumable, "stream_insert_done": stream_insert_done})
        if self.batching:
            impl.update({"classify_reads": classify_reads, "insert_read": insert_read, "batch_insert": batch_insert, "batch_classify": batch_classify,
                         "batch_undo": batch_undo, "batch_end": batch_end})
            if self.batch_verify:
                impl["batch_verify"] = batch_verify
            if self.overlap_every:
                impl["window_overlap"] = window_overlap
        impl.update({"query_tiles": query_tiles, "insert_tiles": insert_tiles, "reset_ids": reset_ids, "sync": sync, "last_error": last_error})
        for name, ftype in host.VT_TYPES:
            if name in impl:
                cb = ftype(impl[name])
                self._keep.append(cb)
                setattr(vt, name, cb)
        return vt


def serial_reference(orc, m, seeds, tile, k, reads, block=10, threshold=10, u=5, a=1, silver=False, target_bases=0, max_paths=1):
    """The reference's serial loop (process_read, goldrush_path.cpp:892-1094)
    stated with oracle primitives; returns the commit tuples the classifier
    should produce: (read, kind, num_tiles, num_assigned, trim_start, trim_end, first_id, path)."""
    oseeds = orc.Seeds(seeds)
    mf = orc.MiBF(m, oseeds, tile, k)
    for s in reads:
        mf.bv_insert_read(s)
    mf.finalize()
    out = []
    ids_inserted = 0
    inserted_bases = 0
    path = 1
    finished = False

    def check():
        nonlocal path, inserted_bases, ids_inserted, finished
        if target_bases < inserted_bases:
            path += 1
            if max_paths < path:
                finished = True
                return
            inserted_bases = 0
            mf.reset_ids()
            ids_inserted = 0

    for ri, seq in enumerate(reads):
        res = mf.query_read(seq)
        n = len(res)
        o_ids, o_b, na = orc.smooth_tiles([r[0] for r in res], [r[2] for r in res], threshold)
        wpath = path
        if n - na >= u and na <= a:
            ids_inserted += 1
            first = ids_inserted
            for bs in range(0, n, block):
                mf.insert_read_tiles(seq, bs, min(bs + block, n), ids_inserted + bs // block)
            ids_inserted += len(seq) // (tile * block)
            inserted_bases += len(seq)
            out.append((ri, 2, n, na, 0, 0, first, wpath))
            if silver:
                check()
        elif na == n:
            out.append((ri, 3, n, na, 0, 0, 0, wpath))
        else:
            ls, le = orc.find_longest_stretch(o_b)
            good, ts, te = orc.eval_flanks(ls, le, o_ids[:n])
            if good:
                ids_inserted += 1
                first = ids_inserted
                bs = ts
                while bs <= te:
                    be = min(bs + block - 1, te)
                    mf.insert_read_tiles(seq, bs, be + 1, ids_inserted + (bs - ts + 1) // block)
                    bs += block
                ids_inserted += (te - ts) // block
                off = ts * tile
                n_out = len(seq) - off if te == n - 1 else min(len(seq) - off, (te - ts + 1) * tile)
                inserted_bases += n_out
                out.append((ri, 4, n, na, ts, te, first, wpath))
                if silver:
                    check()
            else:
                out.append((ri, 5, n, na, 0, 0, 0, wpath))
        if finished:
            break
    return out, mf


_REF_CACHE = {}


def cached_serial_reference(key, orc, m, seeds, tile, k, reads, **kw):
    """serial_reference once per test session and key (the parametrised GPU tests compare many
    engine modes with the same expectation): returns (commits, ids, counts, pop)."""
    if key not in _REF_CACHE:
        exp, mf = serial_reference(orc, m, seeds, tile, k, reads, **kw)
        _REF_CACHE[key] = (exp, mf.ids().copy(), mf.counts().copy(), mf.pop)
        mf.close()
    return _REF_CACHE[key]


class OracleCliEngine:
    """A COMPLETE engine function table backed by the CPU oracle (create, read upload, fill,
    finalize, query, insert, reset), so that the product's whole host program
    (gr_path_main: options, FASTQ reading, filters, passes, classifier, output files) can
    run without a GPU.  No ingest / classify / stream entry points: the host takes its
    plain paths; the --ntcard entry points restate the kernel's rule with oracle hashes."""

    def __init__(self, orc, ingest=False):
        self.orc = orc
        self.ingest = ingest  # also offer the FASTQ ingest entry points (a Python restatement of grpath_ingest.h):
                              # the host then takes its chunked source (GpuSource), reader thread and all
        self.pins = []        # what the host pinned / unpinned, in order
        self.n_parse = 0
        self.ctx = None       # one context at a time is enough for the CLI
        self.batches = {}     # handle -> list of reads (bytes)
        self._next = 1
        self._keep = []
        self.vt = self._make_vt()

    def _make_vt(self):
        vt = host.grp_engine_vt()
        orc = self.orc

        def create(params_p, out_pp):
            p = params_p[0]
            seeds = [p.seeds[i].decode() for i in range(p.h)]
            self.ctx = {"seeds": orc.Seeds(seeds), "k": p.k, "h": p.h, "tile": p.tile, "m": p.m, "mf": None}
            if p.m:  # m = 0: sized after the --ntcard pass (set_filter_size)
                self.ctx["mf"] = orc.MiBF(p.m, self.ctx["seeds"], p.tile, p.k)
            out_pp[0] = 1
            return 0

        def destroy(ctx):
            self.ctx = None

        self._err_text = C.create_string_buffer(b"oracle cli engine")

        def last_error(ctx):
            return C.addressof(self._err_text)

        def reads_upload(ctx, packed_p, word_off_p, len_p, n, out_pp):
            word_off = np.ctypeslib.as_array(C.cast(word_off_p, C.POINTER(C.c_uint64)), shape=(n + 1,))
            lens = np.ctypeslib.as_array(C.cast(len_p, C.POINTER(C.c_uint32)), shape=(n,))
            nw = int(word_off[n])
            packed = np.ctypeslib.as_array(C.cast(packed_p, C.POINTER(C.c_uint32)), shape=(max(nw, 1),))
            lut = np.frombuffer(b"ACGT", dtype=np.uint8)
            reads = []
            for i in range(n):
                w = packed[int(word_off[i]):int(word_off[i + 1])]
                codes = ((w[:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3).astype(np.uint8).ravel()[: int(lens[i])]
                reads.append(lut[codes].tobytes())
            hnd = self._next
            self._next += 1
            self.batches[hnd] = reads
            out_pp[0] = hnd
            return 0

        def reads_free(hnd):
            self.batches.pop(hnd, None)

        def bv_insert(ctx, hnd, first, count):
            k, h = self.ctx["k"], self.ctx["h"]
            for s in self.batches[hnd][first:first + count]:
                if len(s) >= k + h - 1:
                    self.ctx["mf"].bv_insert_read(s)
            return 0

        def finalize(ctx, pop_p):
            pop_p[0] = self.ctx["mf"].finalize()
            return 0

        def query_tiles(ctx, hnd, first, count, tiles_p, lists_p, cap, used_p, stats_p):
            res = []
            for s in self.batches[hnd][first:first + count]:
                res.extend(self.ctx["mf"].query_read(s))
            need = sum(len(x[2]) for x in res)
            used_p[0] = need
            if need > cap:
                return native.GRP_ERR_NOMEM
            tiles = np.ctypeslib.as_array(C.cast(tiles_p, C.POINTER(C.c_uint32)), shape=(max(len(res), 1) * 6,)).view(native.tile_summary_dtype)
            lists = np.ctypeslib.as_array(C.cast(lists_p, C.POINTER(C.c_uint32)), shape=(max(cap, 1) * 2,)).view(native.id_count_dtype)
            off = 0
            for i, (tid, tc, lst, ctr) in enumerate(res):
                lst = sorted(((int(a), int(b)) for a, b in lst), key=lambda t: (-t[1], t[0]))
                tiles[i] = (tid, tc, off, len(lst), ctr[1], ctr[2])
                for j, (a, b) in enumerate(lst):
                    lists[off + j] = (a, b)
                off += len(lst)
            return 0

        def insert_tiles(ctx, hnd, ri, ts, te, id_):
            self.ctx["mf"].insert_read_tiles(self.batches[hnd][ri], ts, te, id_)
            return 0

        def reset_ids(ctx):
            self.ctx["mf"].reset_ids()
            return 0

        def sync(ctx):
            return 0

        # --ntcard entry points: the kernel's rule (grpath.h) stated with oracle hashes — every
        # entry is an ACGT run, each seed counts all its windows once and its last window
        # `extra` more times (default: span_s - k); zero buckets count modulo 2^16
        nt = {}

        def ntcard_begin(ctx, sbits):
            nt.clear()
            nt["sbits"] = sbits
            nt["tables"] = [dict() for _ in range(2 * self.ctx["h"])]
            nt["single"] = [orc.Seeds([p_]) for p_ in self.ctx["seeds"].patterns]
            return 0

        def ntcard_add(ctx, hnd, first, count, extra_p):
            h, k, sbits = self.ctx["h"], self.ctx["k"], nt["sbits"]
            extra = np.ctypeslib.as_array(C.cast(extra_p, C.POINTER(C.c_uint32)), shape=(count * h,)) if extra_p else None
            for j, run in enumerate(self.batches[hnd][first:first + count]):
                for s_ in range(h):
                    K = k + s_
                    if len(run) < K:
                        continue
                    hv = nt["single"][s_].multi_hash(run).astype(np.uint64)
                    times = np.ones(hv.size, dtype=np.uint64)
                    times[-1] += int(extra[j * h + s_]) if extra is not None else s_
                    ind = np.full(hv.size, 2, dtype=np.uint64)
                    ind[(hv >> np.uint64(63 - sbits)) == 1] = 0
                    ind[(hv >> np.uint64(64 - sbits)) == np.uint64((1 << (sbits - 1)) - 1)] = 1
                    sel = ind < 2
                    for t_, b_, w_ in zip(ind[sel], hv[sel] & np.uint64((1 << 27) - 1), times[sel]):
                        tab = nt["tables"][2 * s_ + int(t_)]
                        tab[int(b_)] = tab.get(int(b_), 0) + int(w_)
            return 0

        def ntcard_finish(ctx, zeros_p):
            z = C.cast(zeros_p, C.POINTER(C.c_uint64))
            for i, tab in enumerate(nt["tables"]):
                z[i] = (1 << 27) - sum(1 for v in tab.values() if v % 65536 != 0)
            return 0

        def set_filter_size(ctx, m):
            self.ctx["m"] = m
            self.ctx["mf"] = orc.MiBF(m, self.ctx["seeds"], self.ctx["tile"], self.ctx["k"])
            return 0

        # the host-staged merge of a sharded fill: the oracle's plain bit vector as 32-bit words
        def _bv32():
            mf = self.ctx["mf"]
            n = mf.lib.orcpy_mibf_n_words(mf._h)
            return np.ctypeslib.as_array(C.cast(mf.lib.orcpy_mibf_bv(mf._h), C.POINTER(C.c_uint32)), shape=(2 * n,))

        def bv_words(ctx, n_p):
            n_p[0] = _bv32().size
            return 0

        def bv_export_words(ctx, first, n, words_p):
            self.n_bv_exports = getattr(self, "n_bv_exports", 0) + 1
            np.ctypeslib.as_array(C.cast(words_p, C.POINTER(C.c_uint32)), shape=(n,))[:] = _bv32()[first:first + n]
            return 0

        def bv_or_words(ctx, first, n, words_p):
            _bv32()[first:first + n] |= np.ctypeslib.as_array(C.cast(words_p, C.POINTER(C.c_uint32)), shape=(n,))
            return 0

        # ---- FASTQ ingest (include/grpath_ingest.h), restated: lines by '\n' (a last line without one
        # counts in the final chunk), four per record, extents without trailing CR / blank / tab, id up to
        # the first whitespace, a header that is empty or does not start with '@' ends the input
        rec_dtype = np.dtype([("id_off", "<u8"), ("seq_off", "<u8"), ("qual_off", "<u8"), ("id_len", "<u4"), ("seq_len", "<u4"),
                              ("qual_len", "<u4"), ("flags", "<u4"), ("phred_sum", "<f8"), ("phred_first", "<f8")])
        parsed = {}

        # grp_fastq_prefetch restated as its CONTRACT (grpath_ingest.h): up to two bodies pending, parsed in order, each parse's
        # text ends with the body at the body's address with at most 1 MiB in front, and the body's bytes are the ones that
        # were there when it was handed over (the host must not touch a buffer whose upload may be running)
        pending = []
        self.prefetch_stats = {"issued": 0, "matched": 0, "dropped": 0, "busy": 0}

        def fastq_prefetch(ctx, body_p, n):
            if not body_p and n == 0:  # the caller's buffers are going away
                self.prefetch_stats["dropped"] += len(pending)
                pending.clear()
                return 0
            if len(pending) >= 2:
                self.prefetch_stats["busy"] += 1
                return -6  # GRP_ERR_BUSY
            pending.append((int(body_p), int(n), zlib.crc32(C.string_at(body_p, n))))
            self.prefetch_stats["issued"] += 1
            return 0

        def fastq_parse(ctx, text_p, n, final, out_pp, nrec_p, used_p, stopped_p):
            hl = host.load()
            self.n_parse += 1
            text = C.string_at(text_p, n) if n else b""
            if pending:
                body, nb, crc = pending[0]
                front = n - nb
                if 0 <= front <= (1 << 20) and int(text_p) + front == body:
                    assert zlib.crc32(text[front:]) == crc, "a prefetched body changed before its parse"
                    pending.pop(0)
                    self.prefetch_stats["matched"] += 1
                else:
                    self.prefetch_stats["dropped"] += len(pending)
                    pending.clear()
            lines, pos = [], 0
            while pos < len(text):
                nl = text.find(b"\n", pos)
                if nl < 0:
                    if final:
                        lines.append((pos, len(text)))
                    break
                lines.append((pos, nl))
                pos = nl + 1
            recs, seqs, stopped = [], [], 0
            used = 0
            for r in range(len(lines) // 4):
                ext = []
                for a, e in lines[4 * r: 4 * r + 4]:
                    while e > a and text[e - 1:e] in (b"\r", b" ", b"\t"):
                        e -= 1
                    ext.append((a, e))
                (hs, he), (ss, se), _, (qs, qe) = ext
                if he == hs or text[hs:hs + 1] != b"@":
                    stopped = 1
                    break
                ie = hs + 1
                while ie < he and not text[ie:ie + 1].isspace():
                    ie += 1
                seq, qual = text[ss:se], text[qs:qe]
                half = qual[: len(qual) // 2]
                recs.append((hs + 1, ss, qs, ie - hs - 1, se - ss, qe - qs, 1 if seq.translate(None, b"ACGTacgt") else 0,
                             hl.gr_sum_phred(qual, len(qual)), hl.gr_sum_phred(half, len(half)) if len(qual) >= 2 else 0.0))
                seqs.append(seq.upper())
                used = min(lines[4 * r + 3][1] + 1, len(text))
            if final and not stopped:
                used = len(text)
            hnd = self._next
            self._next += 1
            parsed[hnd] = (np.array(recs, dtype=rec_dtype), seqs)
            out_pp[0] = hnd
            nrec_p[0] = len(recs)
            used_p[0] = used
            stopped_p[0] = stopped
            return 0

        def fastq_records(fq, out_p):
            rec = parsed[fq][0]
            C.memmove(out_p, rec.ctypes.data, rec.nbytes)
            return 0

        def fastq_pack(ctx, fq, sel_p, n_sel, out_pp):
            sel = np.ctypeslib.as_array(C.cast(sel_p, C.POINTER(C.c_uint32)), shape=(max(n_sel, 1),))[:n_sel]
            hnd = self._next
            self._next += 1
            self.batches[hnd] = [parsed[fq][1][int(i)] for i in sel]
            out_pp[0] = hnd
            return 0

        def fastq_free(fq):
            parsed.pop(fq, None)

        def fastq_pin(ctx, buf, n):
            self.pins.append(("pin", buf, n))
            return 0

        def fastq_unpin(ctx):
            self.pins.append(("unpin",))
            return 0

        impl = {"bv_words": bv_words, "bv_export_words": bv_export_words, "bv_or_words": bv_or_words,
                "create": create, "destroy": destroy, "last_error": last_error, "reads_upload": reads_upload, "reads_free": reads_free, "bv_insert": bv_insert,
                "finalize": finalize, "query_tiles": query_tiles, "insert_tiles": insert_tiles, "reset_ids": reset_ids, "sync": sync,
                "ntcard_begin": ntcard_begin, "ntcard_add": ntcard_add, "ntcard_finish": ntcard_finish, "set_filter_size": set_filter_size}
        if self.ingest:
            impl.update({"fastq_parse": fastq_parse, "fastq_records": fastq_records, "fastq_pack": fastq_pack, "fastq_free": fastq_free,
                         "fastq_pin": fastq_pin, "fastq_unpin": fastq_unpin, "fastq_prefetch": fastq_prefetch})
        for name, ftype in host.VT_TYPES:
            if name in impl:
                cb = ftype(impl[name])
                self._keep.append(cb)
                setattr(vt, name, cb)
        return vt
