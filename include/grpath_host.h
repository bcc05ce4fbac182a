/*
 * grpath_host.h — C ABI of libgrpath_host.so: the host side of GoldRush-Path
 * that sits above the engine ABI (grpath.h).  It mirrors the reference host
 * (the .cpp files under goldrush_path/) function by function so the parity tests can call the
 * product's host logic directly; the `goldrush-path` CLI is built from the
 * same sources.  Pure C++17 (no HIP): the engine is reached through the
 * function table grp_engine_vt, whose members have exactly the signatures of
 * grpath.h (opaque handles as void*).
 */
#ifndef GRPATH_HOST_H
#define GRPATH_HOST_H

#include "grpath.h"
#include "grpath_ingest.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- engine function table (filled with the grp_* symbols of grpath.h) ---- */
typedef struct
{
  int (*create)(const grp_params*, void** ctx_out);
  void (*destroy)(void* ctx);
  const char* (*last_error)(const void* ctx);
  int (*reads_upload)(void* ctx, const uint32_t* packed, const uint64_t* word_off, const uint32_t* len, uint32_t n_reads, void** out);
  void (*reads_free)(void* reads);
  int (*bv_insert)(void* ctx, const void* reads, uint32_t first, uint32_t count);
  int (*finalize)(void* ctx, uint64_t* pop);
  int (*query_tiles)(void* ctx, const void* reads, uint32_t first, uint32_t count, grp_tile_summary* tiles, grp_id_count* lists, uint64_t list_cap, uint64_t* list_used, grp_query_stats* stats);
  int (*insert_tiles)(void* ctx, const void* reads, uint32_t read_idx, uint32_t tile_start, uint32_t tile_end, uint32_t id);
  int (*reset_ids)(void* ctx);
  int (*sync)(void* ctx);
  /* optional (may be NULL): query + decision on the device, grp_classify_reads */
  int (*classify_reads)(void* ctx, const void* reads, uint32_t first, uint32_t count, const grp_decide_params* params, grp_read_decision* out);
  /* optional (may be NULL): all ID blocks of one read at once, grp_insert_read */
  int (*insert_read)(void* ctx, const void* reads, uint32_t read_idx, uint32_t tile_start, uint32_t tile_end, uint32_t block_tiles, uint32_t first_id, uint32_t id_offset);
  /* optional (both or none): pipelined windows, grp_classify_reads_begin / _end */
  int (*classify_begin)(void* ctx, const void* reads, uint32_t first, uint32_t count, const grp_decide_params* params, uint32_t slot);
  int (*classify_end)(void* ctx, uint32_t slot, grp_read_decision* out);
  /* optional (all four or none): streaming windows, grp_classify_stream_* */
  int (*stream_begin)(void* ctx, const void* reads, uint32_t first, uint32_t count, const grp_decide_params* params, uint32_t slot, uint32_t stripe_reads, uint32_t n_owners, uint32_t owner, const grp_read_decision** decisions); /* grp_classify_stream_begin_striped */
  int (*stream_abort)(void* ctx, uint32_t slot);
  int (*stream_poll)(void* ctx, uint32_t slot);
  int (*stream_end)(void* ctx, uint32_t slot, uint32_t* reads_decided);
  /* optional (all four or none): a window of reads committed as one batch, grp_batch_* */
  int (*batch_insert)(void* ctx, const void* reads, const grp_batch_insert* inserts, uint32_t n_inserts, uint32_t block_tiles, uint32_t first_read);
  int (*batch_classify)(void* ctx, const void* reads, uint32_t first, uint32_t count, const grp_decide_params* params, const uint32_t* id_floor, grp_read_decision* out);
  int (*batch_undo)(void* ctx, uint32_t from_read, uint32_t id_floor);
  int (*batch_end)(void* ctx);
  /* optional (all four or none): --ntcard on the device, grp_ntcard_* / grp_set_filter_size */
  int (*ntcard_begin)(void* ctx, uint32_t sbits);
  int (*ntcard_add)(void* ctx, const void* reads, uint32_t first, uint32_t count, const uint32_t* stale_extra);
  int (*ntcard_finish)(void* ctx, uint64_t* zero_buckets);
  int (*set_filter_size)(void* ctx, uint64_t m);
  /* optional (all four or none): FASTQ ingest on the device, grpath_ingest.h */
  int (*fastq_parse)(void* ctx, const char* text, uint64_t n_bytes, int final_chunk, void** fq_out, uint64_t* n_records, uint64_t* bytes_consumed, int* stopped);
  int (*fastq_records)(void* fq, grp_fastq_record* out);
  int (*fastq_pack)(void* ctx, void* fq, const uint32_t* sel, uint32_t n_sel, void** reads_out);
  void (*fastq_free)(void* fq);
  /* optional (both, with the stream_* members; round 3): a window that waits where it parks and
   * applies the insert the host commits itself, grp_classify_stream_begin_resumable / _insert;
   * stream_end may then return 1 (the insert posted last was not applied) */
  int (*stream_begin_resumable)(void* ctx, const void* reads, uint32_t first, uint32_t count, const grp_decide_params* params, uint32_t slot, const grp_read_decision** decisions);
  int (*stream_insert)(void* ctx, uint32_t slot, uint32_t read_idx, uint32_t tile_start, uint32_t tile_end, uint32_t block_tiles, uint32_t first_id, uint32_t id_offset, uint32_t* generation);
  /* optional (round 3): the fill sharded over the ranks of one node, the bit vectors OR-merged —
   * through RCCL inside the engine (comm_unique_id / comm_init / bv_merge_ranks, all three or none)
   * or, where the ranks cannot form a communicator, staged through host memory (bv_words /
   * bv_export_words / bv_or_words, all three or none; the host exchanges the words) */
  int (*comm_unique_id)(void* out, size_t cap);
  int (*comm_init)(void* ctx, const void* unique_id, uint32_t world, uint32_t rank);
  int (*bv_merge_ranks)(void* ctx);
  int (*bv_words)(const void* ctx, uint64_t* n_words32);
  int (*bv_export_words)(void* ctx, uint64_t first, uint64_t n_words32, uint32_t* words);
  int (*bv_or_words)(void* ctx, uint64_t first, uint64_t n_words32, const uint32_t* words);
  /* optional: page-lock / release the chunk buffer handed to fastq_parse (grp_fastq_pin / _unpin) */
  int (*fastq_pin)(void* ctx, const char* buffer, uint64_t n_bytes);
  int (*fastq_unpin)(void* ctx);
  /* optional (with the batch_* members; round 4): the second decisions of a batch from the batch's own
   * records instead of a second query, grp_batch_verify */
  int (*batch_verify)(void* ctx, const void* reads, uint32_t first, uint32_t count, uint32_t extra, const grp_decide_params* params, const uint32_t* id_floor, grp_read_decision* out);
  /* optional (round 4): per read of a range the closest read in front of it that it overlaps, grp_window_overlap —
   * where most of the reads insert the classifier ends its batches in front of such reads */
  int (*window_overlap)(void* ctx, const void* reads, uint32_t first, uint32_t count, uint32_t threshold, uint32_t* prev_out);
  /* optional (all three, with stream_insert; round 5): striped windows that apply inserts themselves — the ranks of a
   * multi-GPU run keep their launches across an insert, grp_classify_stream_begin_striped_resumable /
   * grp_classify_stream_resumable; stream_end may then also return 2 (a wait inside the launch timed out) */
  int (*stream_begin_striped_resumable)(void* ctx, const void* reads, uint32_t first, uint32_t count, const grp_decide_params* params, uint32_t slot, uint32_t stripe_reads, uint32_t n_owners, uint32_t owner, const grp_read_decision** decisions);
  int (*stream_resumable)(void* ctx, uint32_t slot);
  int (*stream_insert_done)(void* ctx, uint32_t slot); /* grp_classify_stream_insert_done */
  /* optional (with the fastq_* members; round 5): the next chunk's upload started ahead of its parse, grp_fastq_prefetch */
  int (*fastq_prefetch)(void* ctx, const char* text, uint64_t n_bytes);
  /* optional (round 6): the occupancy the filter was sized for (-o): the phase-2 tables are allocated beside the fill, grp_set_occupancy_hint */
  int (*occupancy_hint)(void* ctx, double occupancy);
} grp_engine_vt;

/* ---- pure functions --------------------------------------------------------- */
/* make_seed_pattern (spaced_seeds.cpp:7-69): h NUL-terminated strings, `stride`
 * bytes apart, into out.  log_to_stderr != 0 prints the reference's messages. */
int gr_make_seed_pattern(const char* preset, unsigned k, unsigned weight, unsigned h, char* out, size_t stride, int log_to_stderr);
/* hash universe (goldrush_path.cpp:1113-1121), single-precision product */
uint64_t gr_hash_universe(uint64_t weight, uint64_t genome_size, uint64_t hash_num);
/* MIBloomFilter::calcOptimalSize (MIBloomFilter.hpp:94-101) */
uint64_t gr_calc_optimal_size(uint64_t entries, unsigned hash_num, double occupancy);
/* calc_phred_average / sum_phred (calc_phred_average.cpp:8-58) */
void gr_calc_phred_average(const char* qual, size_t n, uint32_t* avg, uint32_t* delta);
double gr_sum_phred(const char* qual, size_t n);
/* 2-bit packing of one read into ceil(n/16) words; returns 0, or -1 if the read
 * holds a non-ACGT character */
int gr_pack_2bit(const char* seq, size_t n, uint32_t* out_words);
/* --ntcard host arithmetic (goldrush_path/ntcard.hpp): nts::sBits for the input size
 * (:177-178); compEst's F0 from the zero buckets of the two sample tables (:124-136,
 * :232); and a record with non-ACGT characters cut into the ACGT runs (>= k bases)
 * grp_ntcard_add takes, with the iterator's stale repeats per run and seed
 * (extra[run*h + s]).  gr_ntcard_split returns the number of runs (writes at most cap). */
unsigned gr_ntcard_sbits(uint64_t input_bytes);
uint64_t gr_ntcard_f0(uint64_t zero0, uint64_t zero1, unsigned sbits);
size_t gr_ntcard_split(const char* seq, size_t n, unsigned k, unsigned h, uint64_t* run_off, uint64_t* run_len, uint32_t* extra, size_t cap);
/* process_options (goldrush_path/opt.cpp:89-217) on argv; writes one "name=value" line per
 * option into out (for the parity tests).  Returns -1 when the run would continue,
 * otherwise the exit code the reference exits with (messages go to stdout / stderr). */
int gr_process_options_dump(int argc, char** argv, char* out, size_t cap);
/* CPUs this process may use: min(affinity mask, cgroup cpu.max quota) */
unsigned gr_effective_cpus(void);

/* ---- tile decision (goldrush_path.cpp:628-889, 195-233, 341-527, 960-1040) - */
typedef grp_read_decision gr_read_decision; /* see grpath.h */

/* decision of one read from its tile summaries (list_off indexes `lists`) */
void gr_decide_read(size_t threshold, size_t unassigned_min, size_t assigned_max, size_t num_tiles, const grp_tile_summary* tiles, const grp_id_count* lists, gr_read_decision* out);
/* the intermediate results, for the parity tests: ids/bools have num_tiles entries */
size_t gr_smooth_tiles(size_t num_tiles, const grp_tile_summary* tiles, const grp_id_count* lists, size_t threshold, uint32_t* ids_out, uint8_t* bools_out);
void gr_find_longest_stretch(const uint8_t* bools, size_t num_tiles, long* start, long* end);
int gr_eval_flanks(long longest_start, long longest_end, const uint32_t* ids, size_t num_tiles, size_t* trim_start, size_t* trim_end);

/* ---- order-exact classifier (process_read + silver_path_check) ------------- */
typedef struct
{
  uint32_t struct_size;
  uint32_t tile_length;   /* -t */
  uint32_t block_size;    /* -b */
  uint32_t threshold;     /* -x */
  uint32_t unassigned_min; /* -u */
  uint32_t assigned_max;  /* -a */
  uint32_t kmer_size;     /* -k */
  uint32_t hash_num;      /* -h */
  uint64_t target_bases;  /* uint64(r * G), goldrush_path.cpp:1223 */
  uint64_t max_paths;     /* -M */
  int32_t silver_path;    /* --silver_path */
  int32_t verbose;
  uint32_t max_window;    /* speculation window cap (reads); 0 = default */
  uint32_t world, rank;   /* ranks sharing each window (1, 0 = single GPU) */
  int32_t debug;          /* --debug: reads are decided one by one on the host and the reference's per-read
                             lines and per-pass tile-state dumps go to stderr (goldrush_path.cpp:109-124, 938-1086) */
} gr_classifier_params;

/* one committed read, in file order */
typedef struct
{
  uint32_t read;        /* index in the batch */
  gr_read_decision dec;
  uint32_t first_id;    /* ids_inserted after its ++ (ID of the first block) */
  uint64_t path;        /* curr_path the read was written to */
} gr_commit;

/* called for every classified read in order; returns sum_phred of what it wrote
 * (only used for the verbose per-path log) */
typedef double (*gr_commit_fn)(void* user, const gr_commit* c);
/* called when a silver path rolls over (new output file) */
typedef void (*gr_rollover_fn)(void* user, uint64_t new_path);
/* all-gather of `bytes` per rank (rank r's block at recv + r*bytes); NULL for world==1 */
typedef int (*gr_allgather_fn)(void* user, const void* send, uint64_t bytes, void* recv);

/* ---- node-local all-gather through /dev/shm (what the ranks of a multi-GPU classification
 * exchange: 32-byte decision records, a few KB per call) ---------------------------------
 * _open   rank 0 creates /dev/shm/grp_<key>, the others attach; returns NULL on failure /
 *         after timeout_s.  All `world` ranks of one node call it with the same key.
 * gr_shm_allgather has the signature of gr_allgather_fn with user = the handle.
 * _close  detaches; rank 0 removes the file. */
void* gr_shm_allgather_open(uint32_t world, uint32_t rank, const char* key, double timeout_s);
int gr_shm_allgather(void* handle, const void* send, uint64_t bytes, void* recv);
void gr_shm_allgather_close(void* handle);

/* ---- the fill of several ranks merged into every rank's bit vector (csrc/host/gr_ranks.cpp) -------
 * One implementation for the goldrush-path binary and bench.py's N > 1 runs.  All three are COLLECTIVE over the ranks of
 * `shm` (gr_shm_allgather_open).  _plan: how the ranks will merge — GR_MERGE_RCCL inside the engine (one GPU per rank:
 * grp_comm_unique_id / grp_comm_init, the communicator is up on return), GR_MERGE_STAGED through host memory and /dev/shm
 * (ranks sharing a device, engines without RCCL), GR_MERGE_NONE (every rank fills every read); every decision is taken by
 * all ranks alike.  _run: the merge, between the ranks' fills and finalize — 0 done, 1 an engine call failed, 2 the
 * exchange failed.  gr_ranks_same_u64: 1 if every rank holds `value` (the population behind the merge), 0 / -1. */
#define GR_MERGE_NONE 0
#define GR_MERGE_STAGED 1
#define GR_MERGE_RCCL 2
int gr_fill_merge_plan(const grp_engine_vt* vt, void* ctx, void* shm, uint32_t world, uint32_t rank, int device);
int gr_fill_merge_run(const grp_engine_vt* vt, void* ctx, void* shm, uint32_t world, uint32_t rank, int plan);
int gr_ranks_same_u64(void* shm, uint32_t world, uint64_t value);
/* 1: at least two ranks run on the same device (then their persistent launches must share it: one workgroup per CU each,
 * GRP_STREAM_WGS_PER_CU), 0: none do, -1: the exchange failed.  Collective. */
int gr_ranks_share_device(void* shm, uint32_t world, int device);

typedef struct gr_classifier gr_classifier;

int gr_classifier_create(const gr_classifier_params* p, const grp_engine_vt* vt, void* engine_ctx, gr_classifier** out);
void gr_classifier_destroy(gr_classifier* c);
void gr_classifier_set_callbacks(gr_classifier* c, gr_commit_fn commit, gr_rollover_fn rollover, gr_allgather_fn allgather, void* user);
/* --debug: called (with the commit callbacks' user pointer) in front of every read that is classified:
 * the host prints what the reference prints there — the skipped records before it, "name:", "num tiles:" */
typedef void (*gr_debug_fn)(void* user, uint32_t read);
void gr_classifier_set_debug(gr_classifier* c, gr_debug_fn fn);
/* A witness of the decisions that costs nothing per read: the commits of the reads [first0, first0 + count0) and
 * [first1, first1 + count1) — numbers in the batch of reads handed to gr_classifier_run[_range] — are kept inside
 * the classifier (whether or not a commit callback is set) and handed out by gr_classifier_kept_commits (returns
 * how many there are, writes at most cap).  bench.py compares them with the oracle's serial loop on the same reads. */
void gr_classifier_keep_commits(gr_classifier* c, uint32_t first0, uint32_t count0, uint32_t first1, uint32_t count1);
size_t gr_classifier_kept_commits(const gr_classifier* c, gr_commit* out, size_t cap);
/* the all-gather with its own user pointer (e.g. gr_shm_allgather + its handle) */
void gr_classifier_set_allgather(gr_classifier* c, gr_allgather_fn allgather, void* allgather_user);
/*
 * Classify reads [0, n_reads) of an uploaded batch, in order, exactly as the
 * reference's serial process_read loop would (goldrush_path.cpp:1229-1256):
 * speculative windows are queried on the GPU, committed in order, and
 * everything after a read that inserts is queried again.  skipped_before[i]
 * (may be NULL) = number of non-eligible records (too short / filtered) that
 * precede read i in the file since the previous eligible read — they only
 * advance the read counter (:907-932); skipped_after likewise for the tail.
 * Returns GRP_OK, or a negative grp_status.  *finished is set when the
 * reference would have exit(0)'d (path M complete, :173-176).
 */
int gr_classifier_run(gr_classifier* c, void* reads, const uint32_t* lens, uint32_t n_reads, const uint32_t* skipped_before, uint32_t skipped_after, int* finished);
/* same for reads [first, first+count) of the batch; lens / skipped_before are
 * indexed by the read's number in the batch */
int gr_classifier_run_range(gr_classifier* c, void* reads, const uint32_t* lens, uint32_t first, uint32_t count, const uint32_t* skipped_before, uint32_t skipped_after, int* finished);
const char* gr_classifier_error(const gr_classifier* c);

typedef struct
{
  uint64_t valid_reads, total_tiles, assigned_tiles, unassigned_tiles, queries, hits, misses, num_reads_in_path;
  double phred_sum_in_path;
  uint64_t inserted_bases, curr_path;
  uint32_t id, ids_inserted;
  /* speculation statistics */
  uint64_t windows, reads_queried, reads_committed, inserts;
  /* wall-clock split of gr_classifier_run: engine calls for the windows vs ordered commit */
  double seconds_windows, seconds_commit;
  /* windows committed as batches (grp_batch_*): batches checked, batches ended early, reads committed through them */
  uint64_t batches, batches_undone, batch_reads;
  uint64_t batches_refused; /* batches the engine refused (GRP_ERR_NOMEM): committed the classic way */
  /* round 3: batches whose first decisions came out of the previous batch's second-query launch; inserts a
   * streaming window applied inside its own launch (grp_classify_stream_insert) */
  uint64_t batches_fused, stream_inserts;
  /* round 4: inserts a parked window did not apply (applied the classic way; parked windows are tried again after 64
   * clean windows), windows begun again because their launch had left without deciding a read (idle limit) */
  uint64_t stream_insert_fallbacks, stream_relaunches;
  uint64_t stream_handbacks; /* records a streaming window handed back to the synchronous path (kind 0: a read of more tiles than the
                                in-launch decision holds, a tile that needed the worst-case table) */
  uint64_t stream_rollovers; /* silver mode (round 4): inserts a parked window did not apply itself because the silver path rolls over behind them (the ID array is reset: the launches end there) */
  uint64_t batch_overlap_cuts; /* windows of batches ended in front of a read grp_window_overlap named (round 4) */
  uint64_t overlap_calls;      /* grp_window_overlap calls (one per block of 4096 reads where it is asked) */
} gr_classifier_state;
void gr_classifier_get_state(const gr_classifier* c, gr_classifier_state* out);

/* ---- the input reader (what btllib::SeqReader's file handling is to goldrush_path.cpp:235-339) --
 * Reads `path` (plain text by pread — requests of >= 32 MiB by several threads — or gzip / a pipe
 * through zlib) in requests of `request_bytes` into dst[0..cap).  Returns the bytes delivered
 * (the whole input if it fits cap), UINT64_MAX if the file cannot be opened.  Test hook of the
 * CLI's chunk reader. */
uint64_t gr_input_read(const char* path, uint64_t request_bytes, char* dst, uint64_t cap);

/* ---- the CLI as a function (main of goldrush_path.cpp:1096-1275) ----------- */
int gr_path_main(int argc, char** argv, const grp_engine_vt* vt);

#ifdef __cplusplus
}
#endif
#endif
