/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * compile, link, import or execute anything under oracle/.
 *
 * CPU restatement of GoldRush-Path's first-party logic, function by function:
 *   goldrush_path/spaced_seeds.cpp        make_seed_pattern
 *   goldrush_path/calc_phred_average.cpp  calc_phred_average, sum_phred
 *   goldrush_path/read_hashing.cpp        tile cutting + hashing
 *   goldrush_path/goldrush_path.cpp       everything else (cited per function)
 *   goldrush_path/opt.cpp                 flags
 *
 * PARITY UNPINNED for the path as a whole: the reference has no unit tests / golden
 * vectors for it and cannot be compiled in this image (btllib, sdsl-lite, sparsehash
 * absent).  PINNED parts: make_seed_pattern, calc_phred_average /
 * sum_phred and process_options are checked against the reference's own translation
 * units (spaced_seeds.cpp, calc_phred_average.cpp, opt.cpp build from their sources
 * alone: `make ref` -> oracle/_ref/, tests/test_reference_parts.py,
 * tests/golden/reference_parts.json).
 */
#ifndef ORC_PATH_H
#define ORC_PATH_H

#include "orc_mibf.h"
#include "orc_nthash.h"

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- options: namespace opt (opt.cpp:5-34) ------------------------------ */
typedef struct
{
  size_t assigned_max;   /* -a [1]     */
  size_t unassigned_min; /* -u [5]     */
  size_t tile_length;    /* -t [1000]  */
  uint64_t hash_universe; /* -H [0]    */
  uint64_t genome_size;  /* -g         */
  size_t kmer_size;      /* -k         */
  size_t weight;         /* -w         */
  size_t min_length;     /* -m [20000] */
  size_t hash_num;       /* -h [3]     */
  double occupancy;      /* -o [0.1]   */
  double ratio;          /* -r [0.9]   */
  size_t jobs;           /* -j [48]    */
  size_t block_size;     /* -b [10]    */
  size_t max_paths;      /* -M [1]     */
  size_t threshold;      /* -x [10]    */
  uint32_t phred_min;    /* -P [0]     */
  uint32_t phred_delta;  /* -d [5]     */
  char prefix_file[4096]; /* -p ["goldrush_out"] */
  char input[4096];       /* -i */
  char seed_preset[512];  /* -s */
  char filter_file[4096]; /* -f */
  int help, ntcard, silver_path, verbose, debug;
} orc_opts;

void orc_opts_default(orc_opts* o);
/* process_options (opt.cpp:89-217).  Returns -1 to continue, else the exit
 * code the reference would have exited with. */
int orc_process_options(orc_opts* o, int argc, char** argv);

/* ---- small pure functions ------------------------------------------------ */
/* make_seed_pattern (spaced_seeds.cpp:7-69): writes h NUL-terminated seeds of
 * stride `stride` bytes into out. Returns 0 on success. */
int orc_make_seed_pattern(const char* preset, unsigned k, unsigned weight, unsigned h, char* out, size_t stride);
/* hash universe (goldrush_path.cpp:1113-1121): single-precision float product */
uint64_t orc_hash_universe(uint64_t weight, uint64_t genome_size, uint64_t hash_num);
/* calc_phred_average (calc_phred_average.cpp:8-43) */
void orc_calc_phred_average(const char* qual, size_t n, uint32_t* avg, uint32_t* delta);
/* sum_phred (calc_phred_average.cpp:45-58) */
double orc_sum_phred(const char* qual, size_t n);

/* ---- FASTQ records (btllib::SeqReader semantics) ---------- */
typedef struct
{
  char* id; /* header up to first whitespace, without '@' */
  char* seq; /* upper-cased */
  char* qual;
  size_t len; /* strlen(seq) */
  size_t qlen;
} orc_record;

typedef struct
{
  orc_record* rec;
  size_t n;
  int is_fastq;
} orc_reads;

int orc_reads_load(orc_reads* r, const char* path);
void orc_reads_free(orc_reads* r);

/* ---- tile hashing (read_hashing.cpp:29-54) ------------------------------- */
typedef struct
{
  size_t num_tiles;
  uint64_t** tile_hashes; /* [num_tiles] flat arrays, frame-major f*h+s */
  size_t* tile_sizes;     /* number of uint64 values per tile */
} orc_tile_hashes;

/* computes hashes only when `hashed` != 0 (len >= min_seq_len and not
 * filtered); otherwise every tile vector is empty, as in the reference */
void orc_hash_read_tiles(orc_tile_hashes* th,
                         const char* seq,
                         size_t len,
                         size_t tile_size,
                         size_t k,
                         const orc_seed* seeds,
                         unsigned h,
                         int hashed);
void orc_tile_hashes_free(orc_tile_hashes* th);

/* ---- per-tile query (calc_num_assigned_tiles loop 1, :544-626) ----------- */
typedef struct
{
  uint32_t id;
  uint32_t count;
} orc_id_count;

/*
 * One tile.  top_id/top_count: the arg-max of the count table, ties -> the
 * smallest id (std::map ascending + strict '>', :607-615).  list: every
 * (id,count) with count > 2, sorted by count descending then id ascending
 * (the reference's std::sort is unstable and only list[0].second and list
 * membership are consumed, :628-682).  Returns the list length (may exceed
 * list_cap; only list_cap entries are written).
 * counters[0..2] += queries, hits, misses (:567-591).
 */
size_t orc_query_tile(const orc_mibf* f,
                      const uint64_t* hashes,
                      size_t n_values,
                      unsigned h,
                      uint32_t* top_id,
                      uint32_t* top_count,
                      orc_id_count* list,
                      size_t list_cap,
                      uint64_t counters[3]);

/* The vote of orc_query_tile alone, given every frame's IDs: frame f holds ids[frame_off[f] ..
 * frame_off[f+1]) — saturation bit stripped, no zeros; duplicates inside a frame count once
 * (std::set, :569).  Pinned against the reference's own statements (:597-622,
 * tests/test_reference_funcs.py). */
size_t orc_vote_tile(const uint32_t* ids, const uint64_t* frame_off, size_t n_frames, uint32_t* top_id, uint32_t* top_count, orc_id_count* list, size_t list_cap);

/* ---- threshold + smoothing passes (:628-889) ----------------------------- */
/*
 * In: per-tile top id (ids[]), per-tile count>2 lists (lists[i], list_n[i]).
 * Out: ids[] (rewritten), bools[] ; returns number of assigned tiles.
 * dbg, if non-NULL, receives the --debug tile-state dumps (:109-124).
 */
size_t orc_smooth_tiles(size_t num_tiles,
                        uint32_t* ids,
                        uint8_t* bools,
                        const orc_id_count* const* lists,
                        const size_t* list_n,
                        size_t threshold,
                        FILE* dbg);

/* find_longest_stretch (:195-233) */
void orc_find_longest_stretch(const uint8_t* bools, size_t num_tiles, long* start, long* end);
/* eval_flanks (:341-527); returns good_flank */
int orc_eval_flanks(long longest_start, long longest_end, const uint32_t* ids, size_t num_tiles, size_t* trim_start, size_t* trim_end);

/* ---- whole path state (main + process_read + silver_path_check) ---------- */
typedef struct
{
  uint64_t valid_reads;
  uint64_t total_tiles_per_path;
  uint64_t total_assigned_tiles_per_path;
  uint64_t total_unassigned_tiles_per_path;
  uint64_t total_queries_per_path;
  uint64_t total_hits_per_path;
  uint64_t total_misses_per_path;
  uint64_t num_reads_in_path;
  double phred_sum_in_path;
} orc_log_info; /* log_info_struct, goldrush_path.cpp:41-51 */

/* decision record for one processed read (for parity tests) */
enum
{
  ORC_DEC_SKIP_SHORT = 0, /* :907-918 */
  ORC_DEC_SKIP_FILTERED = 1, /* :919-932 */
  ORC_DEC_INSERT_WHOLE = 2, /* :978-1011 "_untrimmed" */
  ORC_DEC_ASSIGNED_ALL = 3, /* :1013-1023 complete assignment */
  ORC_DEC_INSERT_TRIMMED = 4, /* :1038-1080 "_trimmed" */
  ORC_DEC_ASSIGNED = 5 /* :1083-1088 wood path */
};

typedef struct
{
  int decision;
  size_t num_tiles;
  size_t num_assigned;
  size_t trim_start, trim_end; /* valid for INSERT_TRIMMED */
  uint32_t first_id;           /* ids_inserted after the ++ (first block id) */
  uint64_t path_at_write;      /* curr_path the record was written to */
  int finished;                /* 1 if the reference would have exit(0)'d */
} orc_decision;

typedef struct orc_path orc_path;

/* Everything main() does up to "assigning tiles" (:1096-1208), given already
 * loaded reads.  Returns NULL and sets *exit_code when the reference would
 * have exited. log goes where std::cerr went (may be NULL). */
orc_path* orc_path_open(const orc_opts* o, const orc_reads* reads, FILE* log, int* exit_code);
/* bench.py (like-for-like CPU baseline): OR these words into the bit vector of the NEXT path
 * that is opened (the filter of the whole data set, a path opened on a sample of the reads);
 * and continue a path from an imported state */
void orc_path_use_external_bits(const uint64_t* words, uint64_t n_words);
void orc_path_set_state(orc_path* p, uint32_t ids_inserted, uint64_t inserted_bases, uint32_t id);
/* process_read (:892-1094) for reads->rec[idx]; fills *dec */
void orc_path_process_read(orc_path* p, size_t idx, orc_decision* dec);
/* tail of main() (:1257-1273) and cleanup */
void orc_path_close(orc_path* p);
/* The reference's hashing producers (read_hashing.cpp:77-117; 6 threads, goldrush_path.cpp:1219): n threads hash the
 * reads [first, first + count) ahead of orc_path_process_read through an ordered ring of 64 reads; the reads of
 * the range must then be processed in order, each once.  Without them process_read hashes inline. */
int orc_path_start_producers(orc_path* p, int n, size_t first, size_t count);
void orc_path_stop_producers(orc_path* p);

/* accessors for tests */
orc_mibf* orc_path_mibf(orc_path* p);
const orc_log_info* orc_path_log_info(const orc_path* p);
uint32_t orc_path_phred_min(const orc_path* p);
const char* orc_path_seed(const orc_path* p, unsigned i);
uint64_t orc_path_filter_size(const orc_path* p);
int orc_path_is_filtered(const orc_path* p, size_t idx);
/* phase timers (seconds): [0] bit-vector fill, [1] assigning tiles so far */
void orc_path_timers(const orc_path* p, double out[2]);

/* full CLI (main, :1096-1275): returns the process exit code */
int orc_main(int argc, char** argv);

#ifdef __cplusplus
}
#endif
#endif
