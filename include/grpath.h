/*
 * grpath.h — C ABI of the MI355X (gfx950) GoldRush-Path engine, libgrpath_hip.so
 *
 * This is the drop-in boundary for the hot path of bcgsc/goldrush's
 * goldrush-path (spaced-seed ntHash -> multi-index Bloom filter insert/query
 * -> per-tile ID hit counts).  The reference has no plugin / FFI layer; each
 * entry point below replaces a set of C++ call sites in the reference host
 * (file:line relative to /root/reference/goldrush_path/), so that a host
 * program keeps its control flow and swaps the data-structure calls for
 * these.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - every function returns GRP_OK (0) or a negative grp_status;
 *     grp_last_error() gives the message.  The reference never throws on this
 *     path, it prints to std::cerr and exit(1)s (goldrush_path.cpp:247-250,
 *     327-334); a host built on this ABI prints grp_last_error() and exits
 *     with the same code.
 *   - there is NO CPU fallback: if no HIP device is usable, grp_create fails
 *     with GRP_ERR_NO_DEVICE.
 *   - threading: one host thread per grp_ctx (the reference's consumer side is
 *     single-threaded too, goldrush_path.cpp:1229-1256) — except grp_bv_insert,
 *     which is re-entrant like MIBFConstructSupport::insertBV under `omp parallel`
 *     (goldrush_path.cpp:257-305): any number of host threads, any order of reads,
 *     idempotent (the enqueue is serialised inside the library).
 *   - ownership: the library owns all device memory; caller-provided host
 *     buffers are only read/written during the call.
 *
 * Sequence encoding handed to the library: 2 bits per base, A=0 C=1 G=2 T=3
 * (case-folded), 16 bases per little-endian uint32 word, base i of a read in
 * bits [2*(i%16), 2*(i%16)+2) of word i/16; every read starts on a word
 * boundary.  Only reads that are pure ACGT reach the hot path in the
 * reference (goldrush_path.cpp:293-301), so the encoding is lossless there.
 */
#ifndef GRPATH_H
#define GRPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRP_ABI_VERSION 3

typedef enum
{
  GRP_OK = 0,
  GRP_ERR_INVALID = -1,   /* bad argument / unsupported parameter */
  GRP_ERR_NO_DEVICE = -2, /* no usable HIP device (no CPU fallback exists) */
  GRP_ERR_HIP = -3,       /* a HIP runtime call failed */
  GRP_ERR_STATE = -4,     /* call not valid in the current phase */
  GRP_ERR_NOMEM = -5,
  GRP_ERR_BUSY = -6       /* not now: the call would have to wait for the device while a resumable streaming
                             window may be waiting for the caller (grp_classify_stream_begin*); repeat it later */
} grp_status;

typedef struct grp_ctx grp_ctx;     /* one miBF + its device, stream, scratch */
typedef struct grp_reads grp_reads; /* a batch of packed reads resident in HBM */

/* limits of this implementation (checked by grp_create) */
#define GRP_MAX_SEEDS 8  /* -h */
#define GRP_MAX_SPAN 64  /* k + h - 1 <= 64 bases (round 4; up to 32: one 64-bit window of 2-bit bases, beyond: two) */
#define GRP_MAX_TILE 65535 /* -t: an ID's count per tile is 16 bits (round 5: tile x h IDs need not fit the LDS any more) */

typedef struct
{
  uint32_t struct_size; /* = sizeof(grp_params) */
  uint32_t k;           /* span of the base seed (-k) */
  uint32_t h;           /* number of seed patterns (-h) */
  uint32_t tile;        /* tile length (-t) */
  uint64_t m;           /* filter size in bits, computed by the host with
                           MIBloomFilter::calcOptimalSize
                           (MIBloomFilter.hpp:94-101, goldrush_path.cpp:1183);
                           0 = not known yet, grp_set_filter_size follows (--ntcard) */
  const char* const* seeds; /* h strings of '0'/'1'; seed i has span k+i
                               (make_seed_pattern, spaced_seeds.cpp:63-66) */
  int32_t device;       /* HIP device ordinal; -1 = current device */
  uint32_t flags;       /* reserved, 0 */
} grp_params;

/*
 * Replaces: MIBFConstructSupport ctor (MIBFConstructSupport.hpp:66-84,
 * called at goldrush_path.cpp:1185-1191).  Allocates the zeroed bit vector
 * (in its rank-interleaved HBM layout) on the device.
 */
int grp_create(const grp_params* params, grp_ctx** out);
void grp_destroy(grp_ctx* ctx);
/* message of the last failure on ctx (ctx may be NULL: last grp_create failure) */
const char* grp_last_error(const grp_ctx* ctx);

/* ---- read batches --------------------------------------------------------- */
/*
 * Upload n_reads packed reads.  word_off has n_reads+1 entries (offset of each
 * read in `packed`, in 32-bit words; word_off[n_reads] = total words);
 * len[i] = bases in read i.  Replaces the per-read std::string the reference
 * hands to multiLensfrHashIterator (goldrush_path.cpp:304, read_hashing.cpp:44-46).
 */
int grp_reads_upload(grp_ctx* ctx,
                     const uint32_t* packed,
                     const uint64_t* word_off,
                     const uint32_t* len,
                     uint32_t n_reads,
                     grp_reads** out);
/*
 * Same, but `d_packed` already lives in device memory of ctx's device (e.g.
 * synthetic reads generated on the GPU); it is borrowed, not copied, and must
 * outlive the grp_reads.  word_off / len are host arrays.
 */
int grp_reads_wrap_device(grp_ctx* ctx,
                          const void* d_packed,
                          const uint64_t* word_off,
                          const uint32_t* len,
                          uint32_t n_reads,
                          grp_reads** out);
void grp_reads_free(grp_reads* reads);
/* number of tiles of read i = len[i] / tile (read_hashing.cpp:29-30);
 * tile0[i] = index of its first tile in the batch-wide tile numbering;
 * tile0 has n_reads+1 entries. Pointer valid until grp_reads_free. */
const uint64_t* grp_reads_tile0(const grp_reads* reads);

/* ---- phase 0 (only with --ntcard): expected entries from the reads ---------- */
/*
 * Replaces calc_ntcard_genome_size -> getHist -> stRead -> ntComp
 * (ntcard.hpp:248-275, :156-246, :96-112, :81-94; called at goldrush_path.cpp:1109-1112):
 * the multiLensfrHashIterator stream of EVERY record (no read filter) is sampled into
 * two tables of 2^27 counters per seed; the caller turns the number of zero buckets
 * into F0 with compEst's arithmetic (ntcard.hpp:124-136) — the only value the
 * reference consumes.  The filter size depends on the result, so the context is
 * created with grp_params.m = 0 and sized afterwards with grp_set_filter_size.
 *
 *   grp_ntcard_begin   sbits = nts::sBits (7 below 50e9 input bytes, else 11; :177-178)
 *   grp_ntcard_add     reads [first, first+count) of a batch; every entry is a run of
 *                      ACGT bases.  Each seed counts all its windows once and its last
 *                      window stale_extra[(i-first)*h + s] more times (the iterator
 *                      repeats a seed that can no longer roll); stale_extra == NULL
 *                      means the plain-read rule: span_s - k repeats.  A record with
 *                      non-ACGT characters is passed as its maximal ACGT runs, the
 *                      record's repeats attached to the run holding the seed's last
 *                      window.  Synchronous.
 *   grp_ntcard_finish  zero_buckets[s*2 + t] = buckets of sample table t of seed s
 *                      whose count is 0 mod 2^16 (the reference counts in uint16_t);
 *                      frees the tables.
 */
int grp_ntcard_begin(grp_ctx* ctx, uint32_t sbits);
int grp_ntcard_add(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, const uint32_t* stale_extra);
int grp_ntcard_finish(grp_ctx* ctx, uint64_t* zero_buckets);
/* size the filter of a context created with m = 0 (once, before grp_bv_insert) */
int grp_set_filter_size(grp_ctx* ctx, uint64_t m);

/* ---- phase 1: bit-vector fill -------------------------------------------- */
/*
 * Replaces: multiLensfrHashIterator itr(record.seq, seeds); miBFCS.insertBV(itr)
 * (goldrush_path.cpp:304-305 -> MIBFConstructSupport.hpp:134-147) for reads
 * [first, first+count) of the batch: every position of the WHOLE read, every
 * seed, bit (hash % m) is set.  Asynchronous on the context's stream.
 */
int grp_bv_insert(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count);

/*
 * Multi-GPU fill: the fill is order-free and its merge is a
 * bitwise OR, so every rank fills the reads of its shard and the bit vectors are
 * OR-merged before grp_finalize.  RCCL has no OR reduction; the exchange is an
 * all-gather of the plain bit vectors (done by the caller, e.g. torch.distributed
 * on device buffers) followed by local ORs:
 *   grp_bv_words          number of 32-bit words of the plain bit vector
 *   grp_bv_export_device  copy it into caller-provided DEVICE memory
 *   grp_bv_merge_device   bv |= another rank's copy (device memory, 16-B aligned)
 * All three are only valid before grp_finalize.
 */
int grp_bv_words(const grp_ctx* ctx, uint64_t* n_words32);
int grp_bv_export_device(grp_ctx* ctx, void* d_dst);
int grp_bv_merge_device(grp_ctx* ctx, const void* d_src);
/*
 * The bandwidth-optimal form of the same merge for N ranks (an OR "all-reduce" done as
 * reduce-scatter + all-gather: every rank receives slice r of every other rank's vector
 * (all-to-all), ORs them, and the merged slices are all-gathered — 2 x (N-1)/N of one
 * vector per rank instead of N-1 whole vectors):
 *   grp_words_or_device   dst |= src for n 32-bit words of DEVICE memory (16-byte aligned;
 *                         any buffers of ctx's device, not the bit vector itself)
 *   grp_bv_import_device  overwrite the bit vector with caller-provided DEVICE memory
 *                         (grp_bv_words words; the merged vector)
 */
int grp_words_or_device(grp_ctx* ctx, void* d_dst, const void* d_src, uint64_t n_words32);
int grp_bv_import_device(grp_ctx* ctx, const void* d_src);
/*
 * The same merge INSIDE the engine (round 3), for a host that is not a framework with its own
 * collectives (the goldrush-path binary, one process per GPU): RCCL straight from the library
 * (librccl.so, loaded on first use) over xGMI, on the context's stream —
 *   grp_comm_unique_id  rank 0: the 128 bytes of an ncclUniqueId, handed to the other ranks by the host
 *   grp_comm_init       every rank: ncclCommInitRank on the context's device (world >= 1: a communicator
 *                       of one rank is valid, its merge leaves the vector as it is)
 *   grp_bv_merge_ranks  every rank, after its share of grp_bv_insert, before grp_finalize:
 *                       bv = OR over all ranks (ncclAllToAll of the slices, OR, ncclAllGather)
 * and, for ranks that cannot form a communicator (several ranks on ONE device: the plumbing tests of
 * a one-GPU box), the host-staged form: words [first, first + n) of the plain bit vector out to /
 * ORed in from host memory (first a multiple of 4), the exchange between the ranks being the host's.
 */
int grp_comm_unique_id(void* out, size_t cap);
int grp_comm_init(grp_ctx* ctx, const void* unique_id, uint32_t world, uint32_t rank);
int grp_bv_merge_ranks(grp_ctx* ctx);
/* what the context's communicator is (round 5: a bench line says which path merged its fill): *world = 0 without one;
 * *rccl_version = ncclGetVersion's code (e.g. 22606), 0 if the library does not say; *merges = grp_bv_merge_ranks calls
 * that returned GRP_OK on this context */
int grp_comm_info(const grp_ctx* ctx, uint32_t* world, uint32_t* rank, int* rccl_version, uint32_t* merges);
int grp_bv_export_words(grp_ctx* ctx, uint64_t first, uint64_t n_words32, uint32_t* words);
int grp_bv_or_words(grp_ctx* ctx, uint64_t first, uint64_t n_words32, const uint32_t* words);

/*
 * Replaces: miBFCS.setup(); miBFCS.getEmptyMIBF()
 * (goldrush_path.cpp:1203-1205 -> MIBFConstructSupport.hpp:165-181,
 * MIBloomFilter.hpp:165-184, getPop :538-546).  Builds the rank structure,
 * returns pop (number of set bits) and allocates the zeroed ID and count
 * arrays (pop entries each).  After this call the bit vector is immutable.
 */
int grp_finalize(grp_ctx* ctx, uint64_t* pop);
/*
 * Optional, before the fill (round 6): the occupancy the filter was sized for — the `-o` of
 * calcOptimalSize (MIBloomFilter.hpp:94-101, goldrush_path.cpp:1183-1184).  The phase-2 tables
 * (128 bytes per ~6 / occupancy filter bits: 130 GB at C2) are then allocated by a helper thread
 * while the fill kernels run — hipMalloc of that size takes seconds of host time, which
 * grp_finalize used to spend with the device idle (the reference's own setup / allocation timers:
 * goldrush_path.cpp:1180-1208).  A hint only: grp_finalize measures the occupancy as before and
 * allocates again where the tables prepared for the hint are too small; the filter is the same
 * bit for bit either way.  Nothing is prepared where the tables do not fit beside two more bit
 * vectors (a merge of several ranks' fills) and 16 GB of reads in flight (C4's 246 GB on a 288 GB
 * device): grp_finalize allocates then.  0 < occupancy < 1.
 */
int grp_set_occupancy_hint(grp_ctx* ctx, double occupancy);
/* seconds grp_finalize spent by part (diagnostics of the line above): [0] popcount + wait for the fill, [1] waiting for /
 * doing the table allocations, [2] rank build kernels, [3] far / overflow tables, [4] 1.0 if the prepared tables were used, [5] (part of [2]) releasing the plain bit vector */
int grp_debug_finalize_times(const grp_ctx* ctx, double out[6]);

/* ---- phase 2: tile query -------------------------------------------------- */
typedef struct
{
  uint32_t top_id;    /* arg-max ID of the tile's count table; ties -> smallest
                         ID; 0 if the table is empty (goldrush_path.cpp:607-615) */
  uint32_t top_count; /* its count (frames whose ID set contains it) */
  uint32_t list_off;  /* first entry of this tile in the list array */
  uint32_t list_n;    /* entries with count > 2 (goldrush_path.cpp:616-619),
                         sorted by count descending, then ID ascending */
  uint32_t hits;      /* probes of this tile that returned a non-zero ID
                         (total_hits_per_path, goldrush_path.cpp:577-591) */
  uint32_t misses;    /* probes that returned ID 0 (total_misses_per_path) */
} grp_tile_summary;

typedef struct
{
  uint32_t id;
  uint32_t count;
} grp_id_count;

typedef struct
{
  uint64_t queries; /* frames probed          (goldrush_path.cpp:567-568) */
  uint64_t hits;    /* probes with non-zero ID (:577-591) */
  uint64_t misses;  /* probes with ID 0        (:577-591) */
} grp_query_stats;

/*
 * Replaces, for reads [first, first+count) of the batch: the hashing producer
 * (start_read_hashing / read_hashing, read_hashing.cpp:29-54) and loop 1 of
 * calc_num_assigned_tiles (goldrush_path.cpp:544-626: atRank
 * MIBloomFilter.hpp:465-476, getData :614-621, per-frame ID dedup, per-tile
 * count table, top ID, count>2 list).  The hashes are never materialised.
 *
 * tiles_out receives one summary per tile, reads in order, tiles in order
 * (tile0[first+count]-tile0[first] entries; caller sizes it from
 * grp_reads_tile0).  lists_out receives the count>2 lists, capacity
 * list_cap entries; *list_used returns the number needed — if it exceeds
 * list_cap the call returns GRP_ERR_NOMEM, nothing else is lost, and the call
 * can be repeated with a larger array.  stats may be NULL.
 * Synchronous: returns when the results are in the caller's arrays.
 * Read-only on the miBF.
 */
int grp_query_tiles(grp_ctx* ctx,
                    const grp_reads* reads,
                    uint32_t first,
                    uint32_t count,
                    grp_tile_summary* tiles_out,
                    grp_id_count* lists_out,
                    uint64_t list_cap,
                    uint64_t* list_used,
                    grp_query_stats* stats);

/* ---- phase 2: tile query + read decision on the device --------------------- */
typedef struct
{
  uint32_t threshold;      /* -x */
  uint32_t unassigned_min; /* -u */
  uint32_t assigned_max;   /* -a */
  uint32_t reserved;
} grp_decide_params;

/* one read's decision: 32 bytes, trivially copyable (what ranks exchange) */
typedef struct
{
  uint32_t kind;         /* 2 insert whole read ("_untrimmed"), 3 every tile assigned,
                            4 insert trimmed ("_trimmed"), 5 assigned (wood path) */
  uint32_t num_tiles;
  uint32_t num_assigned; /* after the smoothing passes */
  uint32_t trim_start;   /* valid for kind 4 */
  uint32_t trim_end;
  uint32_t hits;         /* summed over the read's tiles */
  uint32_t misses;
  uint32_t pad;
} grp_read_decision;

/*
 * Replaces, for reads [first, first+count) of the batch: the hashing producer,
 * the whole of calc_num_assigned_tiles (goldrush_path.cpp:529-890: per-tile
 * query, threshold, smoothing passes P1..P10) and the decision of process_read
 * (:960-1040: unassigned rule, find_longest_stretch :195-233, eval_flanks
 * :341-527).  Same query kernel as grp_query_tiles followed by a decision kernel
 * (one lane per read); only the 32-byte decisions come back.  Read-only on the
 * miBF; the caller still has to commit the decisions in file order and issue
 * the inserts (grp_insert_tiles).  Synchronous.
 */
int grp_classify_reads(grp_ctx* ctx,
                       const grp_reads* reads,
                       uint32_t first,
                       uint32_t count,
                       const grp_decide_params* params,
                       grp_read_decision* decisions_out);

/*
 * The same in two halves, so that a second window can run on the GPU while the host
 * commits the first: _begin enqueues query + decision + copy-back of the window into
 * one of two slots (0 / 1) and returns at once; _end waits for that slot and delivers
 * the decisions.  Windows execute in the order of their _begin calls, stream-ordered
 * with grp_insert_tiles / grp_insert_read like every other call.
 */
int grp_classify_reads_begin(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, const grp_decide_params* params, uint32_t slot);
int grp_classify_reads_end(grp_ctx* ctx, uint32_t slot, grp_read_decision* decisions_out);

/*
 * Streaming window: ONE launch over the reads [first, first+count); the workgroup that
 * finishes the last tile of a read takes that read's decision on the device and
 * publishes the 32-byte record in host-visible memory at once, so the host commits
 * read j while the same launch works on the reads behind it.
 *   _begin  enqueues the window in slot 0 / 1 and returns the array the records appear
 *           in: (*decisions)[j].pad becomes non-zero (release) when record j is complete — the
 *           record's generation, 1 until the window has applied an insert (_insert below);
 *           kind == 0 then means "take this read through grp_classify_reads" (a tile
 *           needed the worst-case table or the list arena was too small).  Records
 *           complete roughly in read order.  The window PARKS ITSELF behind the first
 *           record that is an insert (kind 2 / 4) or a hand-back (kind 0): every read
 *           after it is stale whatever the earlier reads decide, so their records may
 *           never complete (all records before it still do).  The array stays valid
 *           until the slot's next _begin.
 *   _abort  (after an insert made the rest of the window stale) workgroups that have
 *           not started yet exit immediately; records already being worked on may
 *           still complete.
 *   _poll   1 when the launch has finished, 0 while it runs (a host spinning on .pad
 *           calls it now and then to notice a failed launch).
 *   _end    waits for the launch, frees the slot; *reads_decided = records completed.
 *   _begin_resumable / _insert (round 3; one rank): a window begun this way does NOT end where
 *           it parks — its workgroups wait for the host's word, _abort or _insert (one of the two
 *           MUST follow a parking record; a window nobody answers gives up after ~30 s).
 *           _insert: the host has committed the insert record the window parked at — read `read_idx`, tiles [tile_start, tile_end),
 *           the block IDs of grp_insert_read — and the LAUNCH applies it: every workgroup collects
 *           the read's ranks, they wait for each other, replay them (exactly grp_insert_read's
 *           result), and the window carries on with the read behind it, without a launch
 *           boundary (goldrush_path.cpp:988-990 / :1048-1049 followed by the next process_read).
 *           Records of the reads behind `read_idx` that were complete before are stale: the
 *           records that count carry .pad == *generation (1 for a fresh window, +1 per insert).
 *           While a resumable window is in flight no call of this context may wait for the device
 *           (the window may be waiting for the caller): a _begin in the other slot that would have
 *           to grow its buffers returns GRP_ERR_BUSY instead — begin it when the window has ended.
 *           GRP_ERR_STATE: this window cannot (not resumable; more than 64 ID blocks; GRP_STREAM_RESUME=off)
 *           — the caller aborts it and issues grp_insert_read as before.
 *   _end    returns 1 or 2 instead of GRP_OK when the launch ended WITHOUT applying the insert posted
 *           last: nothing was inserted, the caller issues grp_insert_read for it.  1: the launch had left
 *           before the command reached it (an abort overtook it; a striped window whose own stripes were
 *           all decided) — nothing is wrong.  2: its workgroups were not all resident (a shared device) and
 *           the first grid-wide wait ran into its time limit — parked windows do not work here.
 * Stream-ordered like every other call: an insert issued after _abort runs behind the
 * (draining) window.
 * Memory ordering inside the launch (gfx942 / gfx950 only, checked at _begin): summaries, lists
 * and counters move between workgroups as relaxed agent-scope accesses + s_waitcnt; the ID
 * words an in-launch insert writes reach the queries behind it through a release fence, a
 * grid-wide wait and an acquire fence (buffer_wbl2 / buffer_inv).
 */
int grp_classify_stream_begin(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, const grp_decide_params* params, uint32_t slot, const grp_read_decision** decisions);
/* The same for a window shared by several ranks (one process per GPU, replicated miBF):
 * stripe t of the window, reads [t*stripe_reads, (t+1)*stripe_reads), belongs to rank
 * t % n_owners; this launch works on the stripes of `owner` only (records of other
 * stripes stay at pad = 0), in window order, so that the ranks can exchange finished
 * stripes while their launches run.  n_owners = 1 is grp_classify_stream_begin. */
int grp_classify_stream_begin_striped(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, const grp_decide_params* params, uint32_t slot, uint32_t stripe_reads, uint32_t n_owners, uint32_t owner, const grp_read_decision** decisions);
int grp_classify_stream_begin_resumable(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, const grp_decide_params* params, uint32_t slot, const grp_read_decision** decisions);
/* Round 5 — the two together: the stripes of `owner` in a window that applies inserts itself, so that the ranks of a
 * multi-GPU run keep their launches across an insert like one rank does (goldrush_path.cpp:988-990 / :1048-1049 inside
 * the serial loop :1229-1256, every rank on its replica).  The read that inserts lies in ONE rank's stripe: that rank's
 * launch parks itself at the record; the others learn of it from the ranks' record exchange and their launches are
 * parked BY THE COMMAND — grp_classify_stream_insert may name any read of the window behind the last insert, whoever
 * owns it; the launch drops what it holds of the reads behind it, applies the insert, and carries on with its own
 * stripes' reads behind `read_idx` under the new generation.  Such a window does not leave by itself when its own
 * stripes are decided (the next insert may be another rank's): the host ends it with _abort once it has committed the
 * window's last read.  grp_classify_stream_resumable: 1 if the window in
 * `slot` takes _insert (the runtime may have refused the cooperative launch: a shared device), 0 if it ends where it
 * parks — the ranks agree on it before they rely on it. */
int grp_classify_stream_begin_striped_resumable(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, const grp_decide_params* params, uint32_t slot, uint32_t stripe_reads, uint32_t n_owners, uint32_t owner, const grp_read_decision** decisions);
int grp_classify_stream_resumable(grp_ctx* ctx, uint32_t slot);
/* ... and whether the insert posted last has been applied: 1 yes, 0 not yet, 2 the launch has ended (or given up) without
 * it — a striped window's launch leaves when its own stripes are decided, and the read that inserts may be another
 * rank's: the caller then ends the window (grp_classify_stream_end says 1 or 2) and issues grp_insert_read.  A rank of
 * several waits for this before it goes on, so that no window queued behind this one starts on a replica without
 * the insert unnoticed. */
int grp_classify_stream_insert_done(grp_ctx* ctx, uint32_t slot);
int grp_classify_stream_insert(grp_ctx* ctx, uint32_t slot, uint32_t read_idx, uint32_t tile_start, uint32_t tile_end, uint32_t block_tiles, uint32_t first_id, uint32_t id_offset, uint32_t* generation);
int grp_classify_stream_abort(grp_ctx* ctx, uint32_t slot);
int grp_classify_stream_poll(grp_ctx* ctx, uint32_t slot);
int grp_classify_stream_end(grp_ctx* ctx, uint32_t slot, uint32_t* reads_decided);

/* ---- phase 2: ID insert ---------------------------------------------------- */
/*
 * Replaces: miBFCS.insertMIBF(*miBF, hashed_values, tile_start, tile_end, id)
 * (goldrush_path.cpp:988-989, 1048-1049 -> MIBFConstructSupport.hpp:247-283,
 * getRankPos MIBloomFilter.hpp:488-491, setData :593-602) for one read of the
 * batch: the set of distinct ranks over all hashes of tiles
 * [tile_start, tile_end) ; for each distinct rank  c = ++counts[rank];
 * if (uint32(rank ^ id) % c == c-1) ids[rank] = id (bit 31 preserved).
 * One call = one dedup scope.  Asynchronous, ordered on the context's stream
 * before any later grp_query_tiles / grp_insert_tiles.
 */
int grp_insert_tiles(grp_ctx* ctx,
                     const grp_reads* reads,
                     uint32_t read_idx,
                     uint32_t tile_start,
                     uint32_t tile_end,
                     uint32_t id);

/*
 * Replaces the whole insert loop of process_read for one read
 * (goldrush_path.cpp:982-990 whole read, :1040-1051 trimmed read): consecutive
 * blocks of block_tiles tiles starting at tile_start, one insertMIBF call (one
 * dedup scope, one ID) per block, in order.  Block j covers tiles
 * [tile_start + j*block_tiles, min(tile_start + (j+1)*block_tiles, tile_end)) and
 * gets the ID  first_id + (j*block_tiles + id_offset) / block_tiles
 * (id_offset 0 = the whole-read rule uint32(block_start / block), :985-986;
 *  id_offset 1 = the trimmed rule uint32((block_start - trim_start + 1) / block),
 *  :1045-1047).  Two launches for the whole read instead of one per block; the
 * result is identical to the sequence of grp_insert_tiles calls.  Asynchronous,
 * stream-ordered like grp_insert_tiles.
 */
int grp_insert_read(grp_ctx* ctx,
                    const grp_reads* reads,
                    uint32_t read_idx,
                    uint32_t tile_start,
                    uint32_t tile_end,
                    uint32_t block_tiles,
                    uint32_t first_id,
                    uint32_t id_offset);

/* ---- phase 2: a batch of reads committed at once (exact speculation across inserts) -------- */
/*
 * Replaces the reference's serial loop (goldrush_path.cpp:1229-1256) for a window of consecutive
 * reads where most reads insert: two reads share about one rank, so decisions taken for the whole
 * window against ONE state almost always hold, and checking them costs a second query instead of a
 * round trip per read.  The caller
 *   1. decides the window against the state in front of it (grp_classify_reads);
 *   2. grp_batch_insert_reads: applies the inserts those decisions ask for (one entry per inserting
 *      read, ascending, with the block IDs the serial loop would allocate: exactly the sequence of
 *      grp_insert_read calls) and keeps a log of what later reads of the window must not see;
 *   3. grp_batch_classify: decides the window again, every read against the state in front of ITS
 *      OWN insert — id_floor[j] = the first ID read j could allocate.  IDs grow with the read
 *      order: a probe returning an ID > id_floor[j] was written by read j or a later one, an ID
 *      equal to it by one of those or by the last ID block of the trimmed read in front
 *      (goldrush_path.cpp:1048-1049 / :1074: with block_tiles == 1 the blocks of a trimmed read are
 *      numbered first + 1 ... while the counter advances to first + (trim_end - trim_start), so the
 *      last block carries the next read's first ID; with larger blocks nothing is shared): the caller sets bit 31 of
 *      id_floor[j] where the last insert in front of read j — inside the batch or before it — was
 *      such a read, and the engine then looks up who wrote the rank — hits / misses included;
 *      Any range of the window's reads may be asked for ([first, first + count) with first >= the
 *      window's first read, id_floor[0] belonging to read `first`): the ranks of a multi-GPU run,
 *      each holding a replica with the same batch applied, take one stripe each;
 *   4. compares in order: while kind / trim range agree, the batch WAS the serial loop and the
 *      second set of decisions are the records.  At the first read that differs: grp_batch_undo
 *      (that read's batch index and id_floor) takes back its insert and those of the reads behind
 *      it and ends the batch — the filter is in the state in front of that read, whose second
 *      decision was taken against exactly this state: commit it through grp_insert_read and
 *      continue behind it;
 *   5. grp_batch_end drops the log (no-op after grp_batch_undo).
 * One batch at a time; nothing else may query or insert between _insert_reads and _end / _undo.
 * GRP_ERR_NOMEM (from _insert_reads, or — found on the device — from _classify / _end): the batch
 * is too large or too many ranks are shared by several ID blocks of it; nothing was inserted, the
 * batch is over: use a smaller window.
 */
typedef struct
{
  uint32_t read;       /* index in the batch of reads */
  uint32_t tile_start; /* tiles [tile_start, tile_end) are inserted */
  uint32_t tile_end;
  uint32_t first_id;   /* as grp_insert_read */
  uint32_t id_offset;  /* 0 whole read, 1 trimmed */
} grp_batch_insert;

int grp_batch_insert_reads(grp_ctx* ctx, const grp_reads* reads, const grp_batch_insert* inserts, uint32_t n_inserts, uint32_t block_tiles, uint32_t first_read);
int grp_batch_classify(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, const grp_decide_params* params, const uint32_t* id_floor, grp_read_decision* decisions_out);
/*
 * Step 3 without the second query of the tiles the batch inserted (round 4).  The collect pass of
 * grp_batch_insert_reads has visited every probe of every inserted tile and knows which of their ranks
 * a second (read, ID block) of the batch touches; every other probe reads, in front of its own read's
 * insert, what the first query read in front of the batch.  grp_batch_verify therefore takes the FIRST
 * decisions' tile summaries — kept by the engine: of the grp_classify_reads call that produced them, or
 * of the previous grp_batch_verify's `extra` reads — evaluates only the frames with a probe on such a
 * rank (through the same log as grp_batch_classify, in front of the batch and in front of the read's own
 * insert) and patches the summaries with what those frames lost and gained; a tile whose new top ID
 * cannot be certified from what the summary holds is queried again.  Tiles without records (reads that
 * do not insert, the rest of a trimmed read) are queried again as by grp_batch_classify.
 *   first         the batch's first read (= first_read of grp_batch_insert_reads)
 *   count         reads of the batch to decide again
 *   extra         reads right behind them, decided against the filter as it is now (the plain query):
 *                 the next window's first decisions if this batch is confirmed in full — their
 *                 summaries stay with the engine for the next grp_batch_verify
 *   id_floor, decisions_out   count + extra entries (floor of an extra read: 0x7FFFFFFF)
 * Results are those of grp_batch_classify over the same count + extra reads, bit for bit.  Falls back
 * to that call by itself when the first decisions' summaries are gone (something else used the engine
 * in between) or GRP_BATCH_VERIFY=off is set.
 */
int grp_batch_verify(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, uint32_t extra, const grp_decide_params* params, const uint32_t* id_floor, grp_read_decision* decisions_out);
/*
 * Where windows should END (round 4).  A batch is taken back from the first read on that decides differently
 * behind the inserts of the reads in front of it — in practice a read that OVERLAPS an inserting read of its own
 * window (14 - 19 % of the batches).  This call hashes the reads [first, first + count) under seed 0 only, keeps one
 * frame in 32 (the canonical hash: either strand; one in 16 until the end of round 5), attributes every sample to the first read of the range that holds
 * it, and returns for every read j (index relative to first) in prev_out[j] the CLOSEST read in front of it that owns
 * at least `threshold` of j's samples, 0xFFFFFFFF if there is none (also for ranges of more than 2^19 tiles, which are
 * not examined).  One call covers many windows: a window [s, e) of the range ends best in front of the first j in
 * (s, e) with prev_out[j] >= s.  Reads nothing of the filter, changes nothing, synchronous; a hint — exactness never
 * depends on it.  An error-free overlap of 1 kb is ~30 samples; unrelated 25 kb reads of a 3 Gbp genome share ~0.005.
 */
int grp_window_overlap(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, uint32_t threshold, uint32_t* prev_out);
int grp_batch_undo(grp_ctx* ctx, uint32_t from_read, uint32_t id_floor);
int grp_batch_end(grp_ctx* ctx);

/*
 * Replaces: miBFCS.reset_counts(); mibf->reset_ID_vector()
 * (goldrush_path.cpp:180-181 -> MIBFConstructSupport.hpp:183-186,
 * MIBloomFilter.hpp:679-682).  Bit vector and rank structure untouched.
 */
int grp_reset_ids(grp_ctx* ctx);

/* wait for all queued work of ctx */
int grp_sync(grp_ctx* ctx);

/* ---- inspection (used by the parity tests; not on the hot path) ----------- */
/* m, number of set bits (0 before finalize) */
uint64_t grp_filter_bits(const grp_ctx* ctx);
uint64_t grp_pop(const grp_ctx* ctx);
/* plain bit vector, bit i = word i>>6 bit i&63 (sdsl::bit_vector layout,
 * MIBFConstructSupport.hpp:140-142); n_words = ceil(m/64) */
int grp_export_bits(grp_ctx* ctx, uint64_t* words, uint64_t n_words);
/* bit[i] / rank[i] (= ones in [0,pos[i])) for n positions; needs finalize */
int grp_rank(grp_ctx* ctx, const uint64_t* pos, uint64_t n, uint8_t* bit, uint64_t* rank);
/* ids / counts of ranks [first, first+n) */
int grp_export_ids(grp_ctx* ctx, uint64_t first, uint64_t n, uint32_t* ids, uint32_t* counts);
/* overwrite ids / counts of ranks [first, first+n) (either may be NULL) */
int grp_import_ids(grp_ctx* ctx, uint64_t first, uint64_t n, const uint32_t* ids, const uint32_t* counts);
/* the hash values of one tile exactly as read_hashing.cpp:47-53 lays them out
 * (frame-major [f*h+s], stale values included); returns values written via
 * *n_values, capacity cap */
int grp_debug_tile_hashes(grp_ctx* ctx,
                          const grp_reads* reads,
                          uint32_t read_idx,
                          uint32_t tile_idx,
                          uint64_t* out,
                          uint64_t cap,
                          uint64_t* n_values);

/* The two exact-arithmetic shortcuts of a probe, for adversarial tests:
 *   mod_out[i] = x[i] % m   through the reciprocal multiply the kernels use in place of
 *                           MIBloomFilter.hpp:468 `hashes[i] % m_bv.size()`
 *   div_out[i] = mod_out[i] / W   through the bucket-index magic multiply (W in [13,64])
 * on_device = 0: the host instantiation of the same inline functions (ctx may be NULL);
 * on_device = 1: a kernel on ctx's device (the __umul64hi instantiation). */
int grp_debug_locate(grp_ctx* ctx, const uint64_t* x, uint64_t n, uint64_t m, uint32_t W, int on_device, uint64_t* mod_out, uint64_t* div_out);

/* per-tile IDs / assigned flags after the smoothing passes of the LAST
 * grp_classify_reads window (tiles in window order); n_tiles = tiles of that window */
int grp_debug_tile_states(grp_ctx* ctx, uint64_t n_tiles, uint32_t* ids, uint8_t* assigned);

/* the decision kernel alone (the device instantiation of the host's threshold / smoothing passes /
 * longest stretch / flank test, goldrush_path.cpp:628-889, :195-233, :341-527, :960-1040) on
 * caller-provided tile summaries: read j owns tiles [tile0[j], tile0[j+1]) (tile0 has n_reads + 1
 * entries), list_off indexes `lists`.  ids_out / asg_out (may be NULL): per-tile state after the passes. */
int grp_debug_decide(grp_ctx* ctx, uint32_t n_reads, const uint64_t* tile0, const grp_tile_summary* tiles, const grp_id_count* lists, uint64_t n_lists, const grp_decide_params* params, grp_read_decision* decisions_out, uint32_t* ids_out, uint8_t* asg_out);

/* grp_batch_verify's counters since grp_create: [0] tiles patched from the batch's records, [1] tiles of those
 * calls queried again (no records), [2] tiles of those calls redone with the worst-case table (flagged by the
 * query kernels or given up by the patch), [3] calls that fell back to grp_batch_classify (the first decisions'
 * summaries were gone), [4] patches given up because the old top ID lost frames and no count > 2 was left,
 * [5] patches given up for another reason (delta table full, shared first ID, flagged first summary),
 * [6] flagged tiles redone with the worst-case table by all synchronous forms (grp_classify_reads / grp_batch_* /
 * grp_query_tiles), of which [7] held more distinct IDs than the small count table takes and [8] had a count > 2
 * list longer than its LDS area; [9] times the batch epochs wrapped and the claims were swept out of the count
 * words (every 1023 batches; GRP_BATCH_EPOCHS=<n> for tests); round 5: [10] tiles whose patch failed its self-check
 * (an impossible delta: queried again through the log, the run goes on), [11] count words living in the far table
 * (ranks beyond their bucket's 8th set bit, csrc/grp_device.h) */
int grp_debug_verify_stats(const grp_ctx* ctx, uint64_t out[12]);

/* What the in-launch inserts of the streaming windows ended so far (since grp_create) did with the tiles the launches had
 * queried behind the inserting read (round 6): [1] finished tiles kept (no probe of theirs read a slot the insert changed),
 * [2] finished tiles queried again because one did, [3] finished tiles queried again because their fingerprints were
 * gone (it keeps the last two tiles of every workgroup), [4], [5] unused (0: tiles are not suspended between passes),
 * [6] inserts that kept nothing (no LDS room for fingerprints in the window's geometry, or the list arena three quarters
 * full), [7] inserts that kept; [0] resumable windows whose cooperative launch was refused */
int grp_debug_stream_stats(const grp_ctx* ctx, uint64_t out[8]);

/* 1: the library was built with GRP_DEV_HOOKS (make DEV=1): the measurement-only prototypes of include/grpath_dev.h are
 * compiled in; 0: the product build (they are not exported) */
int grp_dev_hooks(void);

/* ---- measurement ------------------------------------------------------------ */
enum
{
  GRP_K_FILL = 0,     /* bit-vector fill kernel */
  GRP_K_RANK = 1,     /* rank build kernels (finalize) */
  GRP_K_QUERY = 2,    /* fused hash + probe + tile histogram kernel */
  GRP_K_INSERT = 3,   /* ID insert kernel */
  GRP_K_DECIDE = 4,   /* read decision kernel */
  GRP_K_NTCARD = 5,   /* --ntcard sampling kernel (units = hashes) */
  GRP_K_QUERY_LAT = 6, /* the query kernel in its latency form: windows of a few reads written straight to
                          host memory (insert-heavy stretches); GRP_K_QUERY holds the throughput forms */
  GRP_K_VERIFY = 7,   /* grp_batch_verify: inserted tiles patched from the batch's records (units = their probes) */
  GRP_K_BATCH = 8,    /* grp_batch_insert_reads: collect + apply of a whole batch (units = (frame, seed) records); large launches,
                         timed by default — GRP_K_INSERT holds the latency-critical single-read inserts */
  GRP_K_COUNT = 9
};

typedef struct
{
  uint64_t launches;
  uint64_t units;     /* probes (fill, query, insert) or blocks (rank) */
  double ms;          /* sum of per-launch durations from HIP events recorded
                         on the context's stream around each launch */
} grp_kernel_stat;

/* 0: no per-launch HIP-event timing; 1 (default): fill, rank, query and decide
 * launches are timed; 2: the latency-critical insert launches too */
int grp_set_timing(grp_ctx* ctx, int enabled);
/* drains outstanding events (synchronises) and returns cumulative stats */
int grp_get_kernel_stats(grp_ctx* ctx, grp_kernel_stat out[GRP_K_COUNT]);
int grp_reset_kernel_stats(grp_ctx* ctx);

/* HIP stream the context launches on (hipStream_t as void*) */
void* grp_stream(grp_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* GRPATH_H */
