/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_path.h header comment).
 * PARITY UNPINNED (no reference golden vectors; reference unbuildable here).
 *
 * Every function cites the reference lines it restates; paths are relative to
 * /root/reference/goldrush_path/.
 */
#define _GNU_SOURCE
#include "orc_path.h"
#include "orc_ntcard.h"

#include <ctype.h>
#include <getopt.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <time.h>
#if defined(_OPENMP)
#include <omp.h>
#include <pthread.h>
#endif

static double
now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ======================================================================== */
/* options (opt.cpp)                                                         */
/* ======================================================================== */

void
orc_opts_default(orc_opts* o)
{
  /* opt.cpp:5-34 */
  memset(o, 0, sizeof(*o));
  o->assigned_max = 1;
  o->unassigned_min = 5;
  o->tile_length = 1000;
  o->hash_universe = 0;
  o->genome_size = 0;
  o->kmer_size = 0;
  o->weight = 0;
  o->min_length = 20000;
  o->hash_num = 3;
  o->occupancy = 0.1;
  o->ratio = 0.9;
  o->jobs = 48;
  o->block_size = 10;
  o->max_paths = 1;
  o->threshold = 10;
  o->phred_min = 0;
  o->phred_delta = 5;
  strcpy(o->prefix_file, "goldrush_out");
}

static void
print_usage(const char* progname)
{
  /* opt.cpp:36-87 (text kept verbatim, including its stale defaults) */
  printf(
    "Usage:  %s"
    "  -k K -w W -i INPUT -g G [-p prefix] [-P PHRED_AVG] [-o O] [-t T] [-f "
    "F] [-h H] [-u U] [-m M] [-H HASH_UNIVERSE] [-s S] [-x X] [-M MAX_PATHS]"
    "[-a A] [-j J] [-b B] [-d D] [--silver_path] [--ntcard] [--help] \n\n"
    "  -i INPUT                find golden paths from INPUT [required]\n"
    "  -g G                    estimated genome size [required]\n"
    "  -b B                    during insertion, B number of consecutive "
    "tiles to be inserted with the same ID [10]\n"
    "  -d D                    remove reads with greater or equal then D "
    "phred average "
    "between first half and second half of the read [5]\n"
    "  -f F                    don't use reads from F. Expects one read per "
    "line\n"
    "  -o O                    use O as occupancy [0.1]\n"
    "  -h H                    use h as number of spaced seed patterns [1]\n"
    "  -H HASH_UNIVERSE        determine MiBF size based on HASH_UNIVERSE "
    "[Calculated based on W and h]\n"
    "  -t T                    tile length [1000]\n"
    "  -k K                    span of spaced seed [required]\n"
    "  -w W                    weight of spaced seed [required]\n"
    "  -m M                    use reads longer than M [20000]\n"
    "  -u U                    U minimum unassigned tiles for read to be "
    "unassigned "
    "[5]\n"
    "  -a A                    A maximum assigned tiles for read to be "
    "unassigned [1]\n"
    "  -p prefix               write output to files with prefix "
    "[goldrush_out]\n"
    "  -P PHRED_AVG            minimum average phred score for each read "
    "[0 (calculates phred score minimum automatically)]\n"
    "  -j J                    number of threads [48]\n"
    "  -s S                    use S seed preset. Must be consistent with k "
    "and w [n/a, "
    "generate one randomly based on k and w]\n"
    "  -x X                    require X hits for a tile to be assigned "
    "[10]\n"
    "  -M MAX_PATHS            output MAX_PATHS [5, used with "
    "--silver_path]\n"
    "  --ntcard                use ntcard to estimate genome size [false, "
    "assume max "
    "entries]\n"
    "  --silver_path           generate silver path(s) instead of golden "
    "path. "
    "Silver paths terminate when the number of bases recruited equals or "
    "exceeds T * r\n"
    " --verbose                print verbose messages [false]\n"
    "  --help                  display this help and exit\n",
    progname);
  fflush(stdout);
}

int
orc_process_options(orc_opts* o, int argc, char** argv)
{
  /* opt.cpp:89-217 */
  const struct option longopts[] = { { "debug", no_argument, &o->debug, 1 },
                                     { "verbose", no_argument, &o->verbose, 1 },
                                     { "silver_path", no_argument, &o->silver_path, 1 },
                                     { "help", no_argument, &o->help, 1 },
                                     { "ntcard", no_argument, &o->ntcard, 1 },
                                     { NULL, 0, NULL, 0 } };
  int optindex = 0;
  int c;
  char* end = NULL;
  optind = 0; /* glibc: full re-initialisation (the function may be called twice) */
  while ((c = getopt_long(argc, argv, "a:b:d:f:g:h:i:j:k:m:M:o:r:s:t:u:w:x:p:P:H:", longopts, &optindex)) != -1) {
    switch (c) {
      case 0:
        break;
      case 'a':
        o->assigned_max = strtoul(optarg, &end, 10);
        break;
      case 'b':
        o->block_size = strtoul(optarg, &end, 10);
        break;
      case 'd':
        o->phred_delta = (uint32_t)strtoul(optarg, &end, 10);
        break;
      case 'f':
        snprintf(o->filter_file, sizeof(o->filter_file), "%s", optarg);
        break;
      case 'H':
        o->hash_universe = strtoull(optarg, &end, 10);
        break;
      case 'h':
        o->hash_num = strtoul(optarg, &end, 10);
        break;
      case 'i':
        snprintf(o->input, sizeof(o->input), "%s", optarg);
        break;
      case 'j':
        o->jobs = strtoul(optarg, &end, 10);
        break;
      case 'k':
        o->kmer_size = strtoul(optarg, &end, 10);
        break;
      case 'm':
        o->min_length = strtoul(optarg, &end, 10);
        break;
      case 'M':
        o->max_paths = strtoul(optarg, &end, 10);
        break;
      case 'o':
        o->occupancy = strtod(optarg, &end);
        break;
      case 'r':
        o->ratio = strtod(optarg, &end);
        break;
      case 'p':
        snprintf(o->prefix_file, sizeof(o->prefix_file), "%s", optarg);
        break;
      case 'P':
        o->phred_min = (uint32_t)strtoul(optarg, &end, 10);
        break;
      case 's':
        snprintf(o->seed_preset, sizeof(o->seed_preset), "%s", optarg);
        break;
      case 't':
        o->tile_length = strtoul(optarg, &end, 10);
        break;
      case 'g':
        o->genome_size = (uint64_t)strtod(optarg, &end);
        break;
      case 'u':
        o->unassigned_min = strtoul(optarg, &end, 10);
        break;
      case 'w':
        o->weight = strtoul(optarg, &end, 10);
        break;
      case 'x':
        o->threshold = strtoul(optarg, &end, 10);
        break;
      default:
        return EXIT_FAILURE;
    }
  }

  if (o->help) {
    print_usage("goldrush_path");
    return 0;
  }
  if (!o->kmer_size) {
    fprintf(stderr, "span of spaced seed cannot be 0\n");
    print_usage("goldrush_path");
    return 1;
  }
  if (!o->weight) {
    fprintf(stderr, "weight of spaced seed cannot be 0\n");
    print_usage("goldrush_path");
    return 1;
  }
  if (o->genome_size == 0) {
    fprintf(stderr, "genome size cannot be 0\n");
    print_usage("goldrush_path");
    return 1;
  }
  if (o->seed_preset[0] != '\0') {
    if (o->kmer_size != strlen(o->seed_preset)) {
      fprintf(stderr, "seed preset must be the same size of k\n");
      print_usage("goldrush_path");
      return 1;
    }
    uint8_t num_1s_in_seed = 0;
    for (const char* c2 = o->seed_preset; *c2; ++c2) {
      if (*c2 == '1') {
        ++num_1s_in_seed;
      }
    }
    if (o->weight != num_1s_in_seed) {
      fprintf(stderr, "seed preset must have the same weight as w\n");
      print_usage("goldrush_path");
      return 1;
    }
  }
  return -1;
}

/* ======================================================================== */
/* small pure functions                                                      */
/* ======================================================================== */

static int
make_seed_pattern_log(const char* preset, unsigned k, unsigned weight, unsigned h, char* out, size_t stride, FILE* log)
{
  /* spaced_seeds.cpp:7-69 */
  char left[ORC_MAX_SPAN + 1];
  char right[ORC_MAX_SPAN + 1];
  if (preset == NULL || preset[0] == '\0') {
    srand(123);
    if (log) {
      fprintf(log, "Designing base symmetrical spaced seed\nUsing:\nspan: %u\nweight: %u\n", k, weight);
    }
    unsigned half = k / 2;
    if (half == 0 || half > ORC_MAX_SPAN) {
      return -1;
    }
    unsigned left_seed_vec[ORC_MAX_SPAN];
    memset(left_seed_vec, 0, sizeof(left_seed_vec));
    left_seed_vec[0] = 1; /* left most val in seed must be a 1 */
    size_t weight_count = 0;
    while (weight_count != weight / 2) {
      for (size_t i = 1; i < half; ++i) {
        left_seed_vec[i] = (unsigned)(rand() % 2);
      }
      weight_count = 0;
      for (size_t i = 0; i < half; ++i) {
        weight_count += (left_seed_vec[i] == 1);
      }
    }
    for (size_t i = 0; i < half; ++i) {
      left[i] = (char)('0' + left_seed_vec[i]);
    }
    left[half] = '\0';
    for (size_t i = 0; i < half; ++i) {
      right[i] = left[half - 1 - i];
    }
    right[half] = '\0';
  } else {
    size_t n = strlen(preset);
    size_t ones = 0;
    for (size_t i = 0; i < n; ++i) {
      ones += (preset[i] == '1');
    }
    if (log) {
      fprintf(log, "Using preset spaced seed\nwith:\n\tspan: %zu\n\tweight: %zu\n", n, ones);
    }
    size_t half = n / 2;
    if (half > ORC_MAX_SPAN) {
      return -1;
    }
    memcpy(left, preset, half);
    left[half] = '\0';
    /* substr(size/2, size/2): an odd-length preset drops its last char */
    memcpy(right, preset + half, half);
    right[half] = '\0';
  }
  size_t ll = strlen(left), rl = strlen(right);
  for (unsigned i = 0; i < h; ++i) {
    if (ll + i + rl + 1 > stride) {
      return -1;
    }
    char* dst = out + (size_t)i * stride;
    memcpy(dst, left, ll);
    memset(dst + ll, '0', i);
    memcpy(dst + ll + i, right, rl);
    dst[ll + i + rl] = '\0';
  }
  return 0;
}

int
orc_make_seed_pattern(const char* preset, unsigned k, unsigned weight, unsigned h, char* out, size_t stride)
{
  return make_seed_pattern_log(preset, k, weight, h, out, stride, NULL);
}

uint64_t
orc_hash_universe(uint64_t weight, uint64_t genome_size, uint64_t hash_num)
{
  /* goldrush_path.cpp:1113-1121.  size_t * const float * size_t is evaluated
   * in single precision (usual arithmetic conversions), then truncated. */
  static const uint8_t BASES = 4;
  static const float HASH_UNIVERSE_COEFFICIENT = 0.5;
  static const uint8_t GENOME_SIZE_MULTIPLIER = 2;
  uint64_t a = (uint64_t)(pow(BASES, (double)weight));
  uint64_t b = GENOME_SIZE_MULTIPLIER * genome_size;
  size_t hash_universe_base = a < b ? a : b;
  volatile float prod = (float)hash_universe_base * HASH_UNIVERSE_COEFFICIENT;
  prod = prod * (float)hash_num;
  return (uint64_t)prod;
}

void
orc_calc_phred_average(const char* qual, size_t n, uint32_t* avg, uint32_t* delta)
{
  /* calc_phred_average.cpp:8-43 */
  double phred_sum = 0.0;
  double first_avg = 0.0;
  double second_avg = 0.0;
  size_t qual_size = n;
  for (size_t i = 0; i < qual_size; ++i) {
    int phred_score = (int)(qual[i] - 33);
    double delog_phred = pow(10.0, -phred_score / 10.0);
    phred_sum += delog_phred;
    if (i == qual_size / 2 - 1) {
      first_avg = phred_sum;
    }
  }
  second_avg = phred_sum - first_avg;
  second_avg = second_avg / (qual_size * 0.5);
  first_avg = first_avg / (qual_size * 0.5);
  *avg = (uint32_t)(-10 * log10(phred_sum / qual_size));
  *delta = (uint32_t)abs((int32_t)(-10 * log10(first_avg)) - (int32_t)(-10 * log10(second_avg)));
}

double
orc_sum_phred(const char* qual, size_t n)
{
  /* calc_phred_average.cpp:45-58 */
  double phred_sum = 0;
  for (size_t i = 0; i < n; ++i) {
    int phred_score = (int)(qual[i] - 33);
    double delog_phred = pow(10.0, -phred_score / 10.0);
    phred_sum += delog_phred;
  }
  return phred_sum;
}

/* ======================================================================== */
/* FASTQ reader (btllib::SeqReader semantics)        */
/* ======================================================================== */

static char*
dup_range(const char* s, size_t n)
{
  char* d = (char*)malloc(n + 1);
  memcpy(d, s, n);
  d[n] = '\0';
  return d;
}

static size_t
rtrim_len(const char* s, size_t n)
{
  while (n > 0 && (s[n - 1] == '\r' || s[n - 1] == '\n' || s[n - 1] == ' ' || s[n - 1] == '\t')) {
    --n;
  }
  return n;
}

int
orc_reads_load(orc_reads* r, const char* path)
{
  memset(r, 0, sizeof(*r));
  FILE* fp = fopen(path, "rb");
  if (!fp) {
    return -1;
  }
  fseek(fp, 0, SEEK_END);
  long sz = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  char* buf = (char*)malloc((size_t)sz + 1);
  if (sz > 0 && fread(buf, 1, (size_t)sz, fp) != (size_t)sz) {
    fclose(fp);
    free(buf);
    return -1;
  }
  fclose(fp);
  buf[sz] = '\0';
  r->is_fastq = (sz > 0 && buf[0] == '@');
  if (!r->is_fastq) {
    free(buf);
    return 0;
  }
  size_t cap = 1024;
  r->rec = (orc_record*)malloc(cap * sizeof(orc_record));
  const char* p = buf;
  const char* endp = buf + sz;
  while (p < endp) {
    const char* line[4];
    size_t ln[4];
    int got = 0;
    for (int i = 0; i < 4 && p < endp; ++i) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(endp - p));
      size_t n = nl ? (size_t)(nl - p) : (size_t)(endp - p);
      line[i] = p;
      ln[i] = rtrim_len(p, n);
      p = nl ? nl + 1 : endp;
      ++got;
    }
    if (got < 4 || ln[0] == 0 || line[0][0] != '@') {
      break;
    }
    if (r->n == cap) {
      cap *= 2;
      r->rec = (orc_record*)realloc(r->rec, cap * sizeof(orc_record));
    }
    orc_record* rec = &r->rec[r->n++];
    size_t idn = 0;
    while (1 + idn < ln[0] && !isspace((unsigned char)line[0][1 + idn])) {
      ++idn;
    }
    rec->id = dup_range(line[0] + 1, idn);
    rec->seq = dup_range(line[1], ln[1]);
    for (size_t i = 0; i < ln[1]; ++i) {
      rec->seq[i] = (char)toupper((unsigned char)rec->seq[i]);
    }
    rec->len = ln[1];
    rec->qual = dup_range(line[3], ln[3]);
    rec->qlen = ln[3];
  }
  free(buf);
  return 0;
}

void
orc_reads_free(orc_reads* r)
{
  for (size_t i = 0; i < r->n; ++i) {
    free(r->rec[i].id);
    free(r->rec[i].seq);
    free(r->rec[i].qual);
  }
  free(r->rec);
  memset(r, 0, sizeof(*r));
}

/* ======================================================================== */
/* string set (std::unordered_set<std::string> filter_out_reads)             */
/* ======================================================================== */

typedef struct
{
  char** slot;
  size_t cap, n;
} strset;

static uint64_t
str_hash(const char* s)
{
  uint64_t h = 1469598103934665603ULL;
  for (; *s; ++s) {
    h = (h ^ (unsigned char)*s) * 1099511628211ULL;
  }
  return h;
}

static void strset_insert(strset* ss, const char* s);

static void
strset_grow(strset* ss)
{
  strset old = *ss;
  ss->cap = old.cap ? old.cap * 2 : 64;
  ss->slot = (char**)calloc(ss->cap, sizeof(char*));
  ss->n = 0;
  for (size_t i = 0; i < old.cap; ++i) {
    if (old.slot[i]) {
      strset_insert(ss, old.slot[i]);
      free(old.slot[i]);
    }
  }
  free(old.slot);
}

static void
strset_insert(strset* ss, const char* s)
{
  if ((ss->n + 1) * 2 > ss->cap) {
    strset_grow(ss);
  }
  size_t i = (size_t)(str_hash(s) & (ss->cap - 1));
  while (ss->slot[i]) {
    if (strcmp(ss->slot[i], s) == 0) {
      return;
    }
    i = (i + 1) & (ss->cap - 1);
  }
  ss->slot[i] = strdup(s);
  ss->n++;
}

static int
strset_has(const strset* ss, const char* s)
{
  if (ss->n == 0) {
    return 0;
  }
  size_t i = (size_t)(str_hash(s) & (ss->cap - 1));
  while (ss->slot[i]) {
    if (strcmp(ss->slot[i], s) == 0) {
      return 1;
    }
    i = (i + 1) & (ss->cap - 1);
  }
  return 0;
}

static void
strset_free(strset* ss)
{
  for (size_t i = 0; i < ss->cap; ++i) {
    free(ss->slot[i]);
  }
  free(ss->slot);
  memset(ss, 0, sizeof(*ss));
}

/* ======================================================================== */
/* tile hashing (read_hashing.cpp:29-54)                                     */
/* ======================================================================== */

void
orc_hash_read_tiles(orc_tile_hashes* th,
                    const char* seq,
                    size_t len,
                    size_t tile_size,
                    size_t k,
                    const orc_seed* seeds,
                    unsigned h,
                    int hashed)
{
  const size_t num_tiles = len / tile_size; /* :30 */
  th->num_tiles = num_tiles;
  th->tile_hashes = (uint64_t**)calloc(num_tiles ? num_tiles : 1, sizeof(uint64_t*));
  th->tile_sizes = (size_t*)calloc(num_tiles ? num_tiles : 1, sizeof(size_t));
  if (!hashed) {
    return;
  }
  for (size_t i = 0; i < num_tiles; ++i) {
    /* seq.substr(i * tile_size, tile_size + k - 1), clipped at the end (:44-45) */
    size_t start = i * tile_size;
    size_t tl = tile_size + k - 1;
    if (start + tl > len) {
      tl = len - start;
    }
    size_t frames = orc_multi_hash(seeds, h, seq + start, tl, NULL, 0);
    th->tile_hashes[i] = (uint64_t*)malloc((frames * h + 1) * sizeof(uint64_t));
    orc_multi_hash(seeds, h, seq + start, tl, th->tile_hashes[i], frames * h);
    th->tile_sizes[i] = frames * h;
  }
}

void
orc_tile_hashes_free(orc_tile_hashes* th)
{
  if (th->tile_hashes) {
    for (size_t i = 0; i < th->num_tiles; ++i) {
      free(th->tile_hashes[i]);
    }
  }
  free(th->tile_hashes);
  free(th->tile_sizes);
  memset(th, 0, sizeof(*th));
}

/* ======================================================================== */
/* per-tile query (goldrush_path.cpp:544-626)                                */
/* ======================================================================== */

static int
cmp_u32(const void* a, const void* b)
{
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return (x > y) - (x < y);
}

static int
cmp_idcount_desc(const void* a, const void* b)
{
  const orc_id_count* x = (const orc_id_count*)a;
  const orc_id_count* y = (const orc_id_count*)b;
  if (x->count != y->count) {
    return (x->count < y->count) - (x->count > y->count);
  }
  return (x->id > y->id) - (x->id < y->id);
}

/* the tile's vote over every (frame, unique id) occurrence (:597-622): std::map order = ascending IDs,
 * strict '>' keeps the smallest ID among equal counts, IDs seen more than twice go to the list,
 * sorted by count (sort_by_sec).  Sorts `occ` in place. */
static size_t
vote_occurrences(uint32_t* occ, size_t n_occ, uint32_t* top_id, uint32_t* top_count, orc_id_count* list, size_t list_cap)
{
  qsort(occ, n_occ, sizeof(uint32_t), cmp_u32);
  uint32_t curr_id = 0, curr_id_count = 0; /* :607-608 */
  size_t n_list = 0;
  for (size_t i = 0; i < n_occ;) {
    size_t j = i;
    while (j < n_occ && occ[j] == occ[i]) {
      ++j;
    }
    uint32_t cnt = (uint32_t)(j - i);
    if (cnt > curr_id_count) { /* :612-615 */
      curr_id = occ[i];
      curr_id_count = cnt;
    }
    if (cnt > 2) { /* :616-619 */
      if (n_list < list_cap) {
        list[n_list].id = occ[i];
        list[n_list].count = cnt;
      }
      ++n_list;
    }
    i = j;
  }
  qsort(list, n_list < list_cap ? n_list : list_cap, sizeof(orc_id_count), cmp_idcount_desc); /* :622 */
  *top_id = curr_id;
  *top_count = curr_id_count;
  return n_list;
}

/* The vote alone, given every frame's IDs (saturation bit stripped, no zeros; duplicates inside a
 * frame count once, the reference's std::set): what tests/test_reference_funcs.py holds against
 * the reference's own statements (oracle/_ref/libref_funcs.so: ref_vote_tile). */
size_t
orc_vote_tile(const uint32_t* ids, const uint64_t* frame_off, size_t n_frames, uint32_t* top_id, uint32_t* top_count, orc_id_count* list, size_t list_cap)
{
  const size_t n_all = (size_t)frame_off[n_frames];
  uint32_t* occ = (uint32_t*)malloc((n_all ? n_all : 1) * sizeof(uint32_t));
  size_t n_occ = 0;
  for (size_t fr = 0; fr < n_frames; ++fr) {
    const size_t first = n_occ;
    for (uint64_t i = frame_off[fr]; i < frame_off[fr + 1]; ++i) {
      int seen = 0;
      for (size_t u = first; u < n_occ; ++u) {
        if (occ[u] == ids[i]) {
          seen = 1;
        }
      }
      if (!seen) {
        occ[n_occ++] = ids[i];
      }
    }
  }
  const size_t n_list = vote_occurrences(occ, n_occ, top_id, top_count, list, list_cap);
  free(occ);
  return n_list;
}

size_t
orc_query_tile(const orc_mibf* f,
               const uint64_t* hashes,
               size_t n_values,
               unsigned h,
               uint32_t* top_id,
               uint32_t* top_count,
               orc_id_count* list,
               size_t list_cap,
               uint64_t counters[3])
{
  const size_t frames = n_values / h; /* :560 */
  /* every (frame, unique id) occurrence; sorted afterwards = std::map order */
  uint32_t* occ = (uint32_t*)malloc((n_values ? n_values : 1) * sizeof(uint32_t));
  size_t n_occ = 0;
  uint64_t rank_pos[ORC_MAX_SEEDS];
  uint64_t q = 0, hits = 0, misses = 0;
  for (size_t curr_frame = 0; curr_frame < frames; ++curr_frame) {
    const uint64_t* hv = hashes + curr_frame * h;
    ++q; /* :567-568 */
    uint32_t unique_ids[ORC_MAX_SEEDS];
    unsigned n_unique = 0;
    if (orc_mibf_at_rank(f, hv, rank_pos)) { /* :571 */
      for (unsigned m = 0; m < h; ++m) {
        uint32_t d = f->data[rank_pos[m]]; /* getData :572 */
        uint32_t the_id;
        if (d > ORC_S_MASK) { /* :574 */
          uint32_t new_id = d & ORC_S_ANTIMASK;
          if (new_id == 0) {
            ++misses;
            continue;
          }
          ++hits;
          the_id = new_id;
        } else {
          if (d == 0) {
            ++misses;
            continue;
          }
          ++hits;
          the_id = d;
        }
        int seen = 0;
        for (unsigned u = 0; u < n_unique; ++u) {
          if (unique_ids[u] == the_id) {
            seen = 1;
          }
        }
        if (!seen) {
          unique_ids[n_unique++] = the_id;
        }
      }
    }
    for (unsigned u = 0; u < n_unique; ++u) { /* :597-604 */
      occ[n_occ++] = unique_ids[u];
    }
  }
  const size_t n_list = vote_occurrences(occ, n_occ, top_id, top_count, list, list_cap);
  free(occ);
  if (counters) {
    counters[0] += q;
    counters[1] += hits;
    counters[2] += misses;
  }
  return n_list;
}

/* ======================================================================== */
/* threshold + smoothing passes (goldrush_path.cpp:628-889)                  */
/* ======================================================================== */

static void
log_tile_states(FILE* dbg, const uint32_t* ids, const uint8_t* bools, size_t n)
{
  /* :109-124 */
  if (!dbg) {
    return;
  }
  for (size_t i = 0; i < n; ++i) {
    fprintf(dbg, "%u\t", ids[i]);
  }
  fprintf(dbg, "\n");
  for (size_t i = 0; i < n; ++i) {
    fprintf(dbg, "%u\t", (unsigned)bools[i]);
  }
  fprintf(dbg, "\n");
}

typedef struct
{
  uint32_t id;
  uint32_t idx;
} id_idx;

static int
cmp_id_idx(const void* a, const void* b)
{
  const id_idx* x = (const id_idx*)a;
  const id_idx* y = (const id_idx*)b;
  if (x->id != y->id) {
    return (x->id > y->id) - (x->id < y->id);
  }
  return (x->idx > y->idx) - (x->idx < y->idx);
}

size_t
orc_smooth_tiles(size_t num_tiles,
                 uint32_t* ids,
                 uint8_t* bools,
                 const orc_id_count* const* lists,
                 const size_t* list_n,
                 size_t threshold,
                 FILE* dbg)
{
  size_t num_assigned_tiles = 0;
  /* :628-634 */
  for (size_t i = 0; i < num_tiles; ++i) {
    if (list_n[i] != 0) {
      if (lists[i][0].count > threshold) {
        bools[i] = 1;
      }
    }
  }
  if (num_tiles >= 3) {
    log_tile_states(dbg, ids, bools, num_tiles);

    /* P1 :646-661 */
    for (size_t i = 1; i < num_tiles; ++i) {
      uint32_t curr_id = ids[i];
      uint32_t prev_id = ids[i - 1];
      if (curr_id != prev_id) {
        for (size_t j = 0; j < list_n[i]; ++j) {
          if (lists[i][j].id == prev_id) {
            ids[i] = prev_id;
            if (lists[i][j].count > threshold) {
              bools[i] = 1;
            } else {
              bools[i] = 0;
            }
          }
        }
      }
    }
    log_tile_states(dbg, ids, bools, num_tiles);

    /* P2 :667-682 */
    for (ssize_t i = (ssize_t)num_tiles - 2; i >= 0; --i) {
      uint32_t curr_id = ids[i];
      uint32_t prev_id = ids[i + 1];
      if (curr_id != prev_id) {
        for (size_t j = 0; j < list_n[i]; ++j) {
          if (lists[i][j].id == prev_id) {
            ids[i] = prev_id;
            if (lists[i][j].count > threshold) {
              bools[i] = 1;
            } else {
              bools[i] = 0;
            }
          }
        }
      }
    }
    log_tile_states(dbg, ids, bools, num_tiles);

    /* P3 :688-710 ; uint32_t wrap-around on id +/- 1 as in the reference */
    for (size_t i = 1; i < num_tiles - 1; ++i) {
      uint8_t* curr_assign = &bools[i];
      uint32_t* curr_id = &ids[i];
      const uint8_t prev_assign = bools[i - 1];
      const uint32_t prev_id = ids[i - 1];
      const uint8_t next_assign = bools[i + 1];
      const uint32_t next_id = ids[i + 1];
      if (!*curr_assign) {
        if ((*curr_id == prev_id && prev_assign) || (*curr_id == next_id && next_assign)) {
          *curr_assign = 1;
        } else if ((*curr_id == prev_id + 1 && prev_assign) || (*curr_id == next_id + 1 && next_assign)) {
          *curr_assign = 1;
        } else if ((*curr_id == prev_id - 1 && prev_assign) || (*curr_id == next_id - 1 && next_assign)) {
          *curr_assign = 1;
        } else if (prev_id == next_id && prev_assign && next_assign) {
          bools[i] = prev_assign;
          *curr_id = prev_id;
        }
      }
    }

    /* P4 :712-734 */
    for (size_t i = num_tiles - 2; i >= 1; --i) {
      uint8_t* curr_assign = &bools[i];
      uint32_t* curr_id = &ids[i];
      const uint8_t prev_assign = bools[i - 1];
      const uint32_t prev_id = ids[i - 1];
      const uint8_t next_assign = bools[i + 1];
      const uint32_t next_id = ids[i + 1];
      if (!*curr_assign) {
        if ((*curr_id == prev_id && prev_assign) || (*curr_id == next_id && next_assign)) {
          *curr_assign = 1;
        } else if ((*curr_id == prev_id + 1 && prev_assign) || (*curr_id == next_id + 1 && next_assign)) {
          *curr_assign = 1;
        } else if ((*curr_id == prev_id - 1 && prev_assign) || (*curr_id == next_id - 1 && next_assign)) {
          *curr_assign = 1;
        } else if (prev_id == next_id && prev_assign && next_assign) {
          bools[i] = prev_assign;
          *curr_id = prev_id;
        }
      }
    }
    log_tile_states(dbg, ids, bools, num_tiles);

    /* P5 :739-766 */
    size_t start_idx = 0;
    size_t end_idx = 0;
    size_t n_coords = 0;
    size_t* coord_first = (size_t*)malloc(num_tiles * sizeof(size_t));
    size_t* coord_second = (size_t*)malloc(num_tiles * sizeof(size_t));
    for (size_t i = 1; i < num_tiles - 1; ++i) {
      const uint8_t curr_assign = bools[i];
      const uint8_t prev_assign = bools[i - 1];
      if (!curr_assign && prev_assign) {
        start_idx = i;
      } else if (curr_assign && !prev_assign) {
        end_idx = i - 1;
        coord_first[n_coords] = start_idx;
        coord_second[n_coords] = end_idx;
        ++n_coords;
      }
    }
    for (size_t c = 0; c < n_coords; ++c) {
      if (coord_first[c] == 0 || coord_second[c] == num_tiles - 1) {
        continue;
      }
      const uint32_t left = ids[coord_first[c] - 1];
      const uint32_t right = ids[coord_second[c] + 1];
      if (left == right || left == right + 1 || left == right - 1) {
        for (size_t i = coord_first[c]; i <= coord_second[c]; ++i) {
          bools[i] = 1;
          ids[i] = left;
        }
      }
    }
    log_tile_states(dbg, ids, bools, num_tiles);

    /* P6 :771-793 */
    if (num_tiles >= 3) {
      for (size_t i = 2; i < num_tiles - 2; ++i) {
        const uint8_t curr_assign = bools[i];
        const uint8_t prev_assign = bools[i - 1];
        const uint8_t next_assign = bools[i + 1];
        if (curr_assign) {
          if (!prev_assign && !next_assign) {
            bools[i] = 0;
          }
        }
      }
      for (size_t i = num_tiles - 3; i >= 2; --i) {
        const uint8_t curr_assign = bools[i];
        const uint8_t prev_assign = bools[i - 1];
        const uint8_t next_assign = bools[i + 1];
        if (curr_assign) {
          if (!prev_assign && !next_assign) {
            bools[i] = 0;
          }
        }
      }
    }
    log_tile_states(dbg, ids, bools, num_tiles);

    /* P7 :799-822 : std::map<id, vector<idx>> in ascending id order */
    id_idx* ii = (id_idx*)malloc(num_tiles * sizeof(id_idx));
    size_t n_ii = 0;
    for (size_t i = 0; i < num_tiles; ++i) {
      if (bools[i]) {
        ii[n_ii].id = ids[i];
        ii[n_ii].idx = (uint32_t)i;
        ++n_ii;
      }
    }
    qsort(ii, n_ii, sizeof(id_idx), cmp_id_idx);
    for (size_t g = 0; g < n_ii;) {
      size_t ge = g;
      while (ge < n_ii && ii[ge].id == ii[g].id) {
        ++ge;
      }
      for (size_t i = g + 1; i < ge; ++i) {
        uint32_t prev_idx = ii[i - 1].idx;
        uint32_t curr_idx = ii[i].idx;
        if (curr_idx > prev_idx + 1) {
          uint32_t prev_id = ids[prev_idx];
          for (size_t j = prev_idx + 1; j <= curr_idx; ++j) {
            ids[j] = prev_id;
          }
        }
      }
      g = ge;
    }
    free(ii);
    log_tile_states(dbg, ids, bools, num_tiles);

    /* P8 :827-838 (size_t arithmetic, as in the reference) */
    size_t last_id = ids[num_tiles - 1];
    size_t second_last_id = ids[num_tiles - 2];
    size_t start_id = ids[0];
    size_t second_start_id = ids[1];
    if (last_id == second_last_id || last_id == second_last_id + 1 || last_id == second_last_id - 1) {
      bools[num_tiles - 1] = 1;
    }
    if (start_id == second_start_id || start_id == second_start_id + 1 || start_id == second_start_id - 1) {
      bools[0] = 1;
    }

    /* P9 :840-850 (uint32_t arithmetic) */
    for (size_t i = 1; i < num_tiles - 1; ++i) {
      const uint32_t curr_id = ids[i];
      const uint32_t prev_id = ids[i - 1];
      const uint32_t next_id = ids[i + 1];
      if (curr_id != next_id && curr_id != next_id - 1 && curr_id != next_id + 1 && curr_id != prev_id &&
          curr_id != prev_id - 1 && curr_id != prev_id + 1) {
        bools[i] = 0;
      }
    }
    log_tile_states(dbg, ids, bools, num_tiles);

    /* P10 :856-877 */
    start_idx = 0;
    end_idx = 0;
    n_coords = 0;
    for (size_t i = 1; i < num_tiles - 1; ++i) {
      const uint8_t curr_assign = bools[i];
      const uint8_t prev_assign = bools[i - 1];
      if (curr_assign && !prev_assign) {
        start_idx = i;
      } else if (!curr_assign && prev_assign) {
        end_idx = i - 1;
        coord_first[n_coords] = start_idx;
        coord_second[n_coords] = end_idx;
        ++n_coords;
      }
    }
    for (size_t c = 0; c < n_coords; ++c) {
      if (coord_second[c] - coord_first[c] + 1 <= 5) {
        for (size_t i = coord_first[c]; i <= coord_second[c]; ++i) {
          bools[i] = 0;
        }
      }
    }
    free(coord_first);
    free(coord_second);
    log_tile_states(dbg, ids, bools, num_tiles);
  }
  /* :883-889 */
  num_assigned_tiles = 0;
  for (size_t i = 0; i < num_tiles; ++i) {
    if (bools[i]) {
      ++num_assigned_tiles;
    }
  }
  return num_assigned_tiles;
}

/* ======================================================================== */
/* find_longest_stretch (:195-233)                                           */
/* ======================================================================== */

void
orc_find_longest_stretch(const uint8_t* b, size_t num_tiles, long* start, long* end)
{
  size_t start_idx = 0;
  size_t end_idx = 0;
  ssize_t longest_start_idx = 0;
  ssize_t longest_end_idx = 0;
  size_t curr_stretch = 0;
  size_t longest_stretch = 0;
  /* the reference evaluates `i < num_tiles - 1` in size_t; num_tiles == 0
   * never reaches this function (complete-assignment return, :1013-1023) */
  for (size_t i = 1; num_tiles > 0 && i < num_tiles - 1; ++i) {
    if (!b[i] && b[i - 1]) {
      start_idx = i;
      curr_stretch = 1;
    } else if ((!b[i] && b[i] == b[i - 1]) && (i + 1 != num_tiles - 1)) {
      ++curr_stretch;
    } else if (b[i] && b[i] != b[i - 1]) {
      end_idx = i - 1;
      if (longest_stretch < curr_stretch) {
        longest_stretch = curr_stretch;
        longest_start_idx = (ssize_t)start_idx;
        longest_end_idx = (ssize_t)end_idx;
      }
    } else if (i + 1 == num_tiles - 1 && end_idx < start_idx) {
      end_idx = i;
      ++curr_stretch;
      if (longest_stretch < curr_stretch) {
        longest_stretch = curr_stretch;
        longest_start_idx = (ssize_t)start_idx;
        longest_end_idx = (ssize_t)end_idx;
      }
    }
  }
  *start = (long)longest_start_idx;
  *end = (long)longest_end_idx;
}

/* ======================================================================== */
/* eval_flanks (:341-527)                                                    */
/* ======================================================================== */

typedef struct
{
  size_t first, second;
} szpair;

/* std::map<size_t,size_t> built by repeated ++map[id] : returned ascending */
static size_t
flank_map_add(szpair* m, size_t n, size_t id)
{
  size_t i = 0;
  while (i < n && m[i].first < id) {
    ++i;
  }
  if (i < n && m[i].first == id) {
    ++m[i].second;
    return n;
  }
  memmove(&m[i + 1], &m[i], (n - i) * sizeof(szpair));
  m[i].first = id;
  m[i].second = 1;
  return n + 1;
}

/* sort(v.begin(), v.end(), sort_by_sec): libstdc++ std::sort on fewer than 16
 * elements is a plain insertion sort, i.e. stable for equal counts; the flank
 * maps hold at most 14 entries (num_tiles < 15 or MAX_TILES_TO_CHECK = 5). */
static void
sort_by_sec_desc(szpair* v, size_t n)
{
  for (size_t i = 1; i < n; ++i) {
    szpair val = v[i];
    size_t j = i;
    while (j > 0 && val.second > v[j - 1].second) {
      v[j] = v[j - 1];
      --j;
    }
    v[j] = val;
  }
}

int
orc_eval_flanks(long longest_start_idx, long longest_end_idx, const uint32_t* ids, size_t num_tiles, size_t* trim_start, size_t* trim_end)
{
  szpair* left = (szpair*)calloc(num_tiles + 16, sizeof(szpair));
  szpair* right = (szpair*)calloc(num_tiles + 16, sizeof(szpair));
  size_t n_left = 0, n_right = 0;

  size_t trim_start_idx = 0;
  if (longest_start_idx != 0) {
    trim_start_idx = (size_t)(longest_start_idx - 1);
  } else {
    trim_start_idx = (size_t)longest_start_idx;
  }
  size_t trim_end_idx = (size_t)(longest_end_idx + 1);

  static const uint8_t SMALL_READ_THRESHOLD = 15;
  static const uint8_t MAX_TILES_TO_CHECK = 5;
  static const uint8_t MIN_IDS_IN_FLANK = 2;

  int good_flank = 0;
  if (num_tiles < SMALL_READ_THRESHOLD) {
    int good_right_flank = 0;
    int good_left_flank = 0;

    for (ssize_t i = longest_start_idx - 1; i >= 0; --i) {
      n_left = flank_map_add(left, n_left, ids[i]);
    }
    sort_by_sec_desc(left, n_left);

    if (n_left != 0) {
      if (left[0].second >= MIN_IDS_IN_FLANK) {
        if (longest_start_idx != 0) {
          trim_start_idx = (size_t)(longest_start_idx - 1);
        } else {
          trim_start_idx = (size_t)longest_start_idx;
        }
        good_left_flank = 1;
      } else if (n_left >= 2 &&
                 (left[0].second + left[1].second > (size_t)MIN_IDS_IN_FLANK + 1 &&
                  (left[0].first - left[1].first == 1 || left[1].first - left[0].first == 1))) {
        if (longest_start_idx != 0) {
          trim_start_idx = (size_t)(longest_start_idx - 1);
        } else {
          trim_start_idx = (size_t)longest_start_idx;
        }
        good_left_flank = 1;
      }
    }

    if (trim_start_idx == 0) {
      good_left_flank = 1;
    }

    for (ssize_t i = longest_end_idx + 1; i < (ssize_t)num_tiles; ++i) {
      n_right = flank_map_add(right, n_right, ids[i]);
    }
    sort_by_sec_desc(right, n_right);
    if (n_right != 0) {
      if (right[0].second >= MIN_IDS_IN_FLANK) {
        trim_end_idx = (size_t)(longest_end_idx + 1);
        good_right_flank = 1;
      } else if (n_right >= 2 &&
                 (right[0].second + right[1].second > (size_t)MIN_IDS_IN_FLANK + 1 &&
                  (right[0].first - right[1].first == 1 || right[1].first - right[0].first == 1))) {
        trim_end_idx = (size_t)(longest_end_idx + 1);
        good_right_flank = 1;
      }
    }
    if (trim_end_idx == num_tiles - 1) {
      good_right_flank = 1;
    }

    if (good_left_flank && good_right_flank) {
      good_flank = 1;
    }
  } else {
    if (longest_start_idx - MAX_TILES_TO_CHECK >= 1) {
      for (ssize_t i = longest_start_idx - MAX_TILES_TO_CHECK; i < longest_start_idx; ++i) {
        n_left = flank_map_add(left, n_left, ids[i]);
      }
      sort_by_sec_desc(left, n_left);

      if (left[0].second >= MIN_IDS_IN_FLANK) {
        if (longest_start_idx != 0) {
          trim_start_idx = (size_t)(longest_start_idx - 1);
        } else {
          trim_start_idx = (size_t)longest_start_idx;
        }
        good_flank = 1;
      } else if (left[0].second + left[1].second > (size_t)MIN_IDS_IN_FLANK + 1 &&
                 (left[0].first - left[1].first == 1 || left[1].first - left[0].first == 1)) {
        /* n_left >= 2 here: 5 tiles, top count 1 => 5 distinct ids */
        if (longest_start_idx != 0) {
          trim_start_idx = (size_t)(longest_start_idx - 1);
        } else {
          trim_start_idx = (size_t)longest_start_idx;
        }
        good_flank = 1;
      }
    } else {
      good_flank = 1;
      trim_start_idx = 0;
    }

    if (longest_end_idx + MAX_TILES_TO_CHECK < (ssize_t)num_tiles - 1) {
      for (ssize_t i = longest_end_idx + MAX_TILES_TO_CHECK; i > longest_end_idx; --i) {
        n_right = flank_map_add(right, n_right, ids[i]);
      }
      sort_by_sec_desc(right, n_right);

      if (right[0].second >= MIN_IDS_IN_FLANK) {
        trim_end_idx = (size_t)(longest_end_idx + 1);
        good_flank = 1;
      } else if (right[0].second + right[1].second > (size_t)MIN_IDS_IN_FLANK + 1 &&
                 (right[0].first - right[1].first == 1 || right[1].first - right[0].first == 1)) {
        trim_end_idx = (size_t)(longest_end_idx + 1);
        good_flank = 1;
      }
    } else {
      good_flank = 1;
      trim_end_idx = (size_t)((ssize_t)num_tiles - 1);
    }
  }
  free(left);
  free(right);
  *trim_start = trim_start_idx;
  *trim_end = trim_end_idx;
  return good_flank;
}

/* ======================================================================== */
/* path state: main(), fill_bit_vector, process_read, silver_path_check      */
/* ======================================================================== */

struct orc_path
{
  orc_opts opt;
  const orc_reads* reads;
  FILE* log;
  char seeds_str[ORC_MAX_SEEDS][ORC_MAX_SPAN + ORC_MAX_SEEDS + 1];
  orc_seed seeds[ORC_MAX_SEEDS];
  unsigned h;
  uint64_t filter_size;
  orc_mibf* mibf;
  strset filter_out_reads;
  FILE* out;
  uint64_t inserted_bases;
  uint64_t target_bases;
  uint64_t curr_path;
  uint32_t id;
  uint32_t ids_inserted;
  orc_log_info log_info;
  int finished;
  double t_fill;
  double t_assign_start;
  struct orc_producers* prod; /* hashing threads running ahead of process_read (orc_path_start_producers), NULL: hashes are computed inline */
};

/* ---- the hashing producers of read_hashing.cpp:77-117 ------------------------------------------------------
 * The reference hashes with `worker_num` threads (6, goldrush_path.cpp:1219) that take reads from the reader's
 * queue and push {record, tile hashes} into an ORDER queue the main thread consumes (btllib::OrderQueueMPMC):
 * hashing runs ahead of, and beside, the serial process_read loop.  Here: n threads draw read indices from a
 * counter, hash, and put the result into a ring of Q slots in index order (slot = index % Q; a producer waits
 * until the consumer has passed index - Q).  Reads that process_read skips without hashing (too short, filtered)
 * get an empty entry. */
typedef struct
{
  orc_tile_hashes th;
  size_t idx;
  int ready, hashed;
} orc_prod_slot;

struct orc_producers
{
  orc_path* p;
  pthread_t* threads;
  int n;
  size_t first, end, next, consumed; /* next: next index to hash; consumed: every index below it has been taken */
  orc_prod_slot* ring;
  size_t Q;
  int stop;
  pthread_mutex_t mu;
  pthread_cond_t cv;
};

#define LOGF(p, ...)                                                                                                   \
  do {                                                                                                                 \
    if ((p)->log) {                                                                                                    \
      fprintf((p)->log, __VA_ARGS__);                                                                                  \
    }                                                                                                                  \
  } while (0)

static int
cmp_u32_desc(const void* a, const void* b)
{
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return (x < y) - (x > y);
}

static void
calc_min_phred_threshold(orc_path* p)
{
  /* goldrush_path.cpp:79-107 */
  enum
  {
    MEDIAN_SAMPLES_NEEDED = 50000
  };
  const uint32_t MINIMUM_PHRED_THRESHOLD = 10;
  if (p->opt.phred_min != 0) {
    return;
  }
  LOGF(p, "Calculating minimum phred score via median\n");
  uint32_t* phred_scores = (uint32_t*)calloc(MEDIAN_SAMPLES_NEEDED, sizeof(uint32_t));
  size_t num_reads = 0;
  size_t over = 0;
  for (size_t r = 0; r < p->reads->n; ++r) {
    const orc_record* rec = &p->reads->rec[r];
    if (rec->len < p->opt.min_length) {
      continue;
    }
    if (num_reads >= MEDIAN_SAMPLES_NEEDED) {
      /* every one of the `jobs` OpenMP threads performs exactly one more
       * fetch_add before it breaks (:93-96), if the file still has eligible
       * reads for it */
      ++over;
      if (over >= p->opt.jobs) {
        break;
      }
      continue;
    }
    uint32_t avg, delta;
    orc_calc_phred_average(rec->qual, rec->qlen, &avg, &delta);
    phred_scores[num_reads++] = avg;
  }
  size_t n = num_reads + over;
  qsort(phred_scores, MEDIAN_SAMPLES_NEEDED, sizeof(uint32_t), cmp_u32_desc);
  uint32_t med = phred_scores[n / 2]; /* calc_median :54-59 */
  p->opt.phred_min = med > MINIMUM_PHRED_THRESHOLD ? med : MINIMUM_PHRED_THRESHOLD;
  if (p->opt.debug) { /* log_phred_calculations :60-70 */
    LOGF(p, "Number of reads used to calculate median: %zu\nMedian array: ", n);
    for (size_t i = 0; i < MEDIAN_SAMPLES_NEEDED; ++i) {
      LOGF(p, "%u ", phred_scores[i]);
    }
    LOGF(p, "\n");
  }
  if (p->opt.verbose) {
    LOGF(p, "Minimum phred score calculated with median: %u\n", p->opt.phred_min);
  }
  free(phred_scores);
}

static int
fill_bit_vector(orc_path* p)
{
  /* goldrush_path.cpp:235-339 ; returns exit code or -1 */
  LOGF(p, "inserting bit vector\n");
  double s_time = now_s();
  if (!p->reads->is_fastq) {
    LOGF(p, "Gold Path requires fastq format\n");
    return 1;
  }
  size_t num_reads = 0;
  size_t num_passed_reads = 0;
  size_t num_reads_skipped_by_phred = 0;
  size_t num_reads_skipped_by_delta = 0;
  size_t num_reads_skipped_by_length = 0;
  size_t num_reads_skipped_by_invalid_bases = 0;
  const size_t n = p->reads->n;
  uint8_t* passed = (uint8_t*)calloc(n ? n : 1, 1);
  /* filters first (serial; they mutate the string set), then the hashing +
   * atomic-OR loop in parallel over reads like the reference's omp region */
  for (size_t r = 0; r < n; ++r) {
    const orc_record* rec = &p->reads->rec[r];
    ++num_reads;
    if (rec->len < p->opt.min_length) {
      ++num_reads_skipped_by_length;
      continue;
    }
    uint32_t avg, delta;
    orc_calc_phred_average(rec->qual, rec->qlen, &avg, &delta);
    if (p->opt.debug) {
      LOGF(p, "phred avg: %u\nphred delta: %u\n", avg, delta);
    }
    if (avg < p->opt.phred_min || delta >= p->opt.phred_delta) {
      if (p->opt.verbose) {
        if (avg < p->opt.phred_min) {
          ++num_reads_skipped_by_phred;
        }
        if (delta >= p->opt.phred_delta) {
          ++num_reads_skipped_by_delta;
        }
      }
      strset_insert(&p->filter_out_reads, rec->id);
      continue;
    }
    if (strspn(rec->seq, "ACGTacgt") != rec->len) {
      ++num_reads_skipped_by_invalid_bases;
      strset_insert(&p->filter_out_reads, rec->id);
      continue;
    }
    ++num_passed_reads;
    passed[r] = 1;
  }
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (size_t r = 0; r < n; ++r) {
    if (!passed[r]) {
      continue;
    }
    const orc_record* rec = &p->reads->rec[r];
    /* a read shorter than the longest seed span is outside the reference's defined
     * behaviour (btllib::SeedNtHash on a too-short string; only reachable with -m below
     * k + h - 1): it contributes nothing — the rule the product documents */
    if (rec->len < p->seeds[p->h - 1].span) {
      continue;
    }
    /* multiLensfrHashIterator itr(record.seq, seeds); insertBV(itr) :304-305 */
    size_t frames = orc_multi_hash(p->seeds, p->h, rec->seq, rec->len, NULL, 0);
    uint64_t* hv = (uint64_t*)malloc((frames * p->h + 1) * sizeof(uint64_t));
    orc_multi_hash(p->seeds, p->h, rec->seq, rec->len, hv, frames * p->h);
    orc_mibf_insert_bv(p->mibf, hv, frames * p->h);
    free(hv);
  }
  free(passed);

  if (p->opt.verbose) {
    LOGF(p,
         "num_passed_reads: %zu\nnum_reads: %zu\nnum_reads - num_passed_reads: %zu\n"
         "num_reads - num_passed_reads / num_reads: %.4f\nnum_reads_skipped_by_phred: %zu\n"
         "num_reads_skipped_by_delta: %zu\nnum_reads_skipped_by_length: %zu\n"
         "num_reads_skipped_by_invalid_bases: %zu\nTotal reads skipped: %zu\n",
         num_passed_reads,
         num_reads,
         num_reads - num_passed_reads,
         floor((double)(num_reads - num_passed_reads) / (double)num_reads),
         num_reads_skipped_by_phred,
         num_reads_skipped_by_delta,
         num_reads_skipped_by_length,
         num_reads_skipped_by_invalid_bases,
         num_reads_skipped_by_phred + num_reads_skipped_by_delta + num_reads_skipped_by_length +
           num_reads_skipped_by_invalid_bases);
  }
  if (num_passed_reads == 0) {
    LOGF(p,
         "Error: no reads passed the Phred score and min length requirements\n"
         "Try again with a lower Phred threshold or lower min length\n");
    return 1;
  }
  LOGF(p, "finished inserting bit vector\n");
  p->t_fill = now_s() - s_time;
  LOGF(p, "in %.4f\n", p->t_fill);
  return -1;
}

/* bench.py's like-for-like CPU baseline: the bit vector of the whole data set (filled on
 * the GPU from all reads) is OR-ed into the next path that is opened, so that a path opened
 * on a SAMPLE of the reads probes the same filter as the measured run */
static const uint64_t* g_external_bits = NULL;
static uint64_t g_external_words = 0;

void
orc_path_use_external_bits(const uint64_t* words, uint64_t n_words)
{
  g_external_bits = words;
  g_external_words = n_words;
}

/* the loop state in front of the next read (goldrush_path.cpp:1222-1227), for a path that
 * continues from an imported miBF state */
void
orc_path_set_state(orc_path* p, uint32_t ids_inserted, uint64_t inserted_bases, uint32_t id)
{
  p->ids_inserted = ids_inserted;
  p->inserted_bases = inserted_bases;
  p->id = id;
}

orc_path*
orc_path_open(const orc_opts* o, const orc_reads* reads, FILE* log, int* exit_code)
{
  orc_path* p = (orc_path*)calloc(1, sizeof(orc_path));
  p->opt = *o;
  p->reads = reads;
  p->log = log;
  *exit_code = -1;
#if defined(_OPENMP)
  omp_set_num_threads((int)o->jobs); /* :1101-1103 */
#endif
  /* :1106-1107 */
  p->h = (unsigned)o->hash_num;
  if (p->h == 0 || p->h > ORC_MAX_SEEDS ||
      make_seed_pattern_log(o->seed_preset, (unsigned)o->kmer_size, (unsigned)o->weight, p->h, &p->seeds_str[0][0], sizeof(p->seeds_str[0]), log) != 0) {
    LOGF(p, "oracle: unsupported seed configuration\n");
    *exit_code = 1;
    free(p);
    return NULL;
  }
  for (unsigned i = 0; i < p->h; ++i) {
    orc_seed_parse(&p->seeds[i], p->seeds_str[i]);
  }
  /* :1109-1123 */
  if (p->opt.hash_universe == 0) {
    if (p->opt.ntcard) {
      /* calc_ntcard_genome_size (ntcard.hpp:248-275) over every record of the input */
      const double nt_s = now_s();
      uint64_t input_bytes = 0;
      FILE* fsz = fopen(o->input, "rb");
      if (fsz) {
        fseek(fsz, 0, SEEK_END);
        input_bytes = (uint64_t)ftell(fsz); /* getInf (:44-49) */
        fclose(fsz);
      }
      LOGF(p, "Calculating expected entries\n");
      orc_ntcard* nc = orc_ntcard_new(p->h, input_bytes);
      for (size_t i = 0; i < reads->n; ++i) {
        orc_ntcard_add_read(nc, p->seeds, reads->rec[i].seq, reads->rec[i].len);
      }
      LOGF(p, "Reapeat profile estimated using ntCard in (sec): %.4f\n", now_s() - nt_s);
      uint64_t genome_size = 0;
      for (unsigned i = 0; i < p->h; ++i) {
        const uint64_t f0 = orc_ntcard_f0(nc, i);
        LOGF(p, "Expected entries for seed pattern %s : %llu\n", p->seeds_str[i], (unsigned long long)f0);
        genome_size += f0;
      }
      LOGF(p, "Total expected entries for seed patterns: %llu\n", (unsigned long long)genome_size);
      orc_ntcard_free(nc);
      p->opt.hash_universe = genome_size;
    } else {
      p->opt.hash_universe = orc_hash_universe(o->weight, o->genome_size, o->hash_num);
    }
  }
  char num_and_type_path_log[64];
  if (o->silver_path) {
    snprintf(num_and_type_path_log, sizeof(num_and_type_path_log), "%zu silver path(s)", o->max_paths);
  } else {
    snprintf(num_and_type_path_log, sizeof(num_and_type_path_log), "the golden path");
  }
  calc_min_phred_threshold(p); /* :1131 */
  /* getHist leaves std::cerr in setprecision(4) << fixed (ntcard.hpp:240-241) */
  char occupancy_str[64];
  snprintf(occupancy_str, sizeof(occupancy_str), (o->hash_universe == 0 && o->ntcard) ? "%.4f" : "%g", p->opt.occupancy);
  LOGF(p,
       "Calculating %s\nUsing:\n\ttile length: %zu\n\tblock size: %zu\n\tseed patterns: %zu\n\tthreshold: %zu\n"
       "\tbase seed pattern: %s\n\tminimum unassigned tiles: %zu\n\tmaximum assigned tiles: %zu\n"
       "\texpected hash space: %llu\n\tminimum average phred quality score: %u\n"
       "\tmaximum average phred delta between first and second half of read: %u\n\toccupancy: %s\n\tjobs: %zu\n",
       num_and_type_path_log,
       p->opt.tile_length,
       p->opt.block_size,
       p->opt.hash_num,
       p->opt.threshold,
       p->seeds_str[0],
       p->opt.unassigned_min,
       p->opt.assigned_max,
       (unsigned long long)p->opt.hash_universe,
       p->opt.phred_min,
       p->opt.phred_delta,
       occupancy_str,
       p->opt.jobs);
  /* :1161-1171 */
  if (o->filter_file[0] != '\0') {
    LOGF(p, "Using only reads not found in: %s\n", o->filter_file);
    FILE* ff = fopen(o->filter_file, "r");
    if (ff) {
      char name[8192];
      while (fscanf(ff, "%8191s", name) == 1) {
        strset_insert(&p->filter_out_reads, name);
      }
      fclose(ff);
    }
  }
  /* :1173-1179 */
  char path[8300];
  if (o->silver_path) {
    snprintf(path, sizeof(path), "%s_1.fq", o->prefix_file);
  } else {
    snprintf(path, sizeof(path), "%s.fa", o->prefix_file);
  }
  p->out = fopen(path, "w");
  double s_time = now_s();
  LOGF(p, "allocating bit vector\n");
  /* :1183-1191 */
  p->filter_size = orc_calc_optimal_size(p->opt.hash_universe, 1, p->opt.occupancy);
  LOGF(p, "m_filterSize: %llu\n", (unsigned long long)p->filter_size);
  p->mibf = orc_mibf_create(p->filter_size, p->h);
  LOGF(p, "finished allocating bit vector\nin %.4f\n", now_s() - s_time);
  LOGF(p, "opening: %s\n", o->input);
  int ec = fill_bit_vector(p); /* :1199-1200 */
  if (ec >= 0) {
    *exit_code = ec;
    orc_path_close(p);
    return NULL;
  }
  if (g_external_bits) {
    const uint64_t nw = g_external_words < p->mibf->n_words ? g_external_words : p->mibf->n_words;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static)
#endif
    for (uint64_t i = 0; i < nw; ++i) {
      p->mibf->bv[i] |= g_external_bits[i];
    }
    g_external_bits = NULL;
    g_external_words = 0;
  }
  orc_mibf_finalize(p->mibf); /* :1203-1205 */
  LOGF(p, "assigning tiles\n");
  p->t_assign_start = now_s();
  /* :1222-1227 */
  p->inserted_bases = 0;
  p->target_bases = (uint64_t)(p->opt.ratio * (double)p->opt.genome_size);
  p->curr_path = 1;
  p->id = 1;
  p->ids_inserted = 0;
  memset(&p->log_info, 0, sizeof(p->log_info));
  return p;
}

static void
log_path_stat(orc_path* p)
{
  /* :126-154 */
  const orc_log_info* li = &p->log_info;
  unsigned long long cp = (unsigned long long)p->curr_path;
  LOGF(p, "Visited %llu reads to generate %llu silver paths\n", (unsigned long long)li->valid_reads, cp);
  LOGF(p, "Saw: %llu tiles to generate %llu silver paths\n", (unsigned long long)li->total_tiles_per_path, cp);
  LOGF(p, "Assigned: %llu tiles to generate %llu silver paths\n", (unsigned long long)li->total_assigned_tiles_per_path, cp);
  LOGF(p, "Unassigned: %llu tiles to generate %llu silver paths\n", (unsigned long long)li->total_unassigned_tiles_per_path, cp);
  LOGF(p, "Total queries: %llu to generate %llu silver paths\n", (unsigned long long)li->total_queries_per_path, cp);
  LOGF(p, "Total hits: %llu to generate %llu silver paths\n", (unsigned long long)li->total_hits_per_path, cp);
  LOGF(p, "Total misses: %llu to generate %llu silver paths\n", (unsigned long long)li->total_misses_per_path, cp);
  LOGF(p, "Num reads: %llu in silver path %llu\n", (unsigned long long)li->num_reads_in_path, cp);
  uint32_t avg_phred = (uint32_t)(-10 * log10(li->phred_sum_in_path / (double)p->inserted_bases));
  LOGF(p, "Average Phred: %u in silver path %llu\n", avg_phred, cp);
}

static void
silver_path_check(orc_path* p)
{
  /* :156-187 */
  if (p->target_bases < p->inserted_bases) {
    if (p->opt.verbose) {
      log_path_stat(p);
    }
    ++p->curr_path;
    if (p->opt.max_paths < p->curr_path) {
      p->finished = 1; /* exit(0) in the reference */
      return;
    }
    p->inserted_bases = 0;
    p->log_info.num_reads_in_path = 0;
    p->log_info.phred_sum_in_path = 0;
    orc_mibf_reset_ids(p->mibf);
    fclose(p->out);
    char path[8300];
    snprintf(path, sizeof(path), "%s_%llu.fq", p->opt.prefix_file, (unsigned long long)p->curr_path);
    p->out = fopen(path, "w");
    p->ids_inserted = 0;
  }
}

static void
progress(orc_path* p)
{
  if (p->id % 10000 == 0) {
    LOGF(p, "processed %u reads\n", p->id);
  }
}


static int strset_has(const strset* ss, const char* s);

static void*
producer_main(void* arg)
{
  struct orc_producers* pr = (struct orc_producers*)arg;
  orc_path* p = pr->p;
  for (;;) {
    pthread_mutex_lock(&pr->mu);
    if (pr->stop || pr->next >= pr->end) {
      pthread_mutex_unlock(&pr->mu);
      return NULL;
    }
    const size_t idx = pr->next++;
    pthread_mutex_unlock(&pr->mu);
    const orc_record* record = &p->reads->rec[idx];
    orc_tile_hashes th;
    memset(&th, 0, sizeof(th));
    int hashed = 0;
    /* the reads process_read hashes: long enough and not filtered out (read_hashing.cpp:31-43 / goldrush_path.cpp:907-932) */
    if (record->len >= p->opt.min_length && !(p->filter_out_reads.n != 0 && strset_has(&p->filter_out_reads, record->id))) {
      orc_hash_read_tiles(&th, record->seq, record->len, p->opt.tile_length, p->opt.kmer_size, p->seeds, p->h, 1);
      hashed = 1;
    }
    pthread_mutex_lock(&pr->mu);
    while (!pr->stop && idx >= pr->consumed + pr->Q) { /* the ring is full: the consumer has not passed idx - Q yet */
      pthread_cond_wait(&pr->cv, &pr->mu);
    }
    if (pr->stop) {
      pthread_mutex_unlock(&pr->mu);
      if (hashed) {
        orc_tile_hashes_free(&th);
      }
      return NULL;
    }
    orc_prod_slot* sl = &pr->ring[idx % pr->Q];
    sl->th = th;
    sl->idx = idx;
    sl->hashed = hashed;
    sl->ready = 1;
    pthread_cond_broadcast(&pr->cv);
    pthread_mutex_unlock(&pr->mu);
  }
}

void
orc_path_stop_producers(orc_path* p)
{
  struct orc_producers* pr = p ? p->prod : NULL;
  if (!pr) {
    return;
  }
  pthread_mutex_lock(&pr->mu);
  pr->stop = 1;
  pthread_cond_broadcast(&pr->cv);
  pthread_mutex_unlock(&pr->mu);
  for (int i = 0; i < pr->n; ++i) {
    pthread_join(pr->threads[i], NULL);
  }
  for (size_t i = 0; i < pr->Q; ++i) {
    if (pr->ring[i].ready && pr->ring[i].hashed) {
      orc_tile_hashes_free(&pr->ring[i].th);
    }
  }
  pthread_mutex_destroy(&pr->mu);
  pthread_cond_destroy(&pr->cv);
  free(pr->ring);
  free(pr->threads);
  free(pr);
  p->prod = NULL;
}

/* n hashing threads for the reads [first, first + count), which must then be processed in order, each once */
int
orc_path_start_producers(orc_path* p, int n, size_t first, size_t count)
{
  if (!p || n <= 0 || first + count > p->reads->n) {
    return -1;
  }
  orc_path_stop_producers(p);
  struct orc_producers* pr = (struct orc_producers*)calloc(1, sizeof(*pr));
  pr->p = p;
  pr->n = n;
  pr->first = pr->next = pr->consumed = first;
  pr->end = first + count;
  pr->Q = 64; /* reads in flight between the producers and the consumer */
  pr->ring = (orc_prod_slot*)calloc(pr->Q, sizeof(orc_prod_slot));
  pr->threads = (pthread_t*)calloc((size_t)n, sizeof(pthread_t));
  pthread_mutex_init(&pr->mu, NULL);
  pthread_cond_init(&pr->cv, NULL);
  p->prod = pr;
  for (int i = 0; i < n; ++i) {
    pthread_create(&pr->threads[i], NULL, producer_main, pr);
  }
  return 0;
}

/* the consumer's side: the hashes of read idx (1), or 0 when idx is not the next read of the producers' range */
static int
producer_take(orc_path* p, size_t idx, orc_tile_hashes* th)
{
  struct orc_producers* pr = p->prod;
  if (!pr || idx != pr->consumed || idx >= pr->end) {
    return 0;
  }
  pthread_mutex_lock(&pr->mu);
  orc_prod_slot* sl = &pr->ring[idx % pr->Q];
  while (!(sl->ready && sl->idx == idx)) {
    pthread_cond_wait(&pr->cv, &pr->mu);
  }
  *th = sl->th;
  const int hashed = sl->hashed;
  sl->ready = 0;
  pr->consumed = idx + 1;
  pthread_cond_broadcast(&pr->cv);
  pthread_mutex_unlock(&pr->mu);
  return hashed ? 1 : 2;
}

void
orc_path_process_read(orc_path* p, size_t idx, orc_decision* dec)
{
  /* goldrush_path.cpp:892-1094 */
  const orc_opts* opt = &p->opt;
  const orc_record* record = &p->reads->rec[idx];
  orc_decision local;
  if (!dec) {
    dec = &local;
  }
  memset(dec, 0, sizeof(*dec));
  dec->path_at_write = p->curr_path;
  if (p->finished) {
    dec->finished = 1;
    return;
  }
  orc_tile_hashes th;
  const int taken = producer_take(p, idx, &th); /* 1: hashed ahead by a producer thread, 2: a read that is not hashed, 0: no producers */
  if (record->len < opt->min_length) { /* :907-918 */
    if (opt->debug) {
      LOGF(p, "too short\nskipping: %s\n", record->id);
    }
    dec->decision = ORC_DEC_SKIP_SHORT;
    ++p->id;
    progress(p);
    return;
  }
  if (p->filter_out_reads.n != 0) { /* :919-932 */
    if (strset_has(&p->filter_out_reads, record->id)) {
      if (opt->debug) {
        LOGF(p, "hairpin or quality too low or invalid bases\nskipping: %s\n", record->id);
      }
      dec->decision = ORC_DEC_SKIP_FILTERED;
      ++p->id;
      progress(p);
      return;
    }
  }
  size_t len = record->len;
  size_t num_tiles = len / opt->tile_length;
  p->log_info.total_tiles_per_path += num_tiles;
  dec->num_tiles = num_tiles;
  if (opt->debug) { /* :938-941 */
    LOGF(p, "name: %s\nnum tiles: %zu\n", record->id, num_tiles);
  }

  /* the producer side (read_hashing.cpp:29-54): hashes of this read's tiles — computed ahead by the hashing
   * threads when there are any (orc_path_start_producers), inline otherwise */
  if (taken != 1) {
    orc_hash_read_tiles(&th, record->seq, len, opt->tile_length, opt->kmer_size, p->seeds, p->h, 1);
  }

  int assigned = 1;
  uint32_t* ids = (uint32_t*)calloc(num_tiles ? num_tiles : 1, sizeof(uint32_t));
  uint8_t* bools = (uint8_t*)calloc(num_tiles ? num_tiles : 1, 1);

  /* calc_num_assigned_tiles (:529-890) */
  orc_id_count** lists = (orc_id_count**)calloc(num_tiles ? num_tiles : 1, sizeof(orc_id_count*));
  size_t* list_n = (size_t*)calloc(num_tiles ? num_tiles : 1, sizeof(size_t));
  uint64_t counters[3] = { 0, 0, 0 };
#if defined(_OPENMP)
#pragma omp parallel for
#endif
  for (size_t i = 0; i < num_tiles; ++i) {
    uint64_t local_counters[3] = { 0, 0, 0 };
    size_t cap = th.tile_sizes[i] ? th.tile_sizes[i] : 1;
    lists[i] = (orc_id_count*)malloc(cap * sizeof(orc_id_count));
    uint32_t top_id = 0, top_count = 0;
    list_n[i] = orc_query_tile(p->mibf, th.tile_hashes[i], th.tile_sizes[i], p->h, &top_id, &top_count, lists[i], cap, local_counters);
    ids[i] = top_id;
#if defined(_OPENMP)
#pragma omp critical(orc_counters)
#endif
    {
      counters[0] += local_counters[0];
      counters[1] += local_counters[1];
      counters[2] += local_counters[2];
    }
  }
  p->log_info.total_queries_per_path += counters[0];
  p->log_info.total_hits_per_path += counters[1];
  p->log_info.total_misses_per_path += counters[2];
  const size_t num_assigned_tiles = orc_smooth_tiles(
    num_tiles, ids, bools, (const orc_id_count* const*)lists, list_n, opt->threshold, opt->debug ? p->log : NULL);
  const size_t num_unassigned_tiles = num_tiles - num_assigned_tiles;
  if (opt->debug) { /* :957-964 */
    LOGF(p, "num assigned tiles: %zu\nnum unassigned tiles: %zu\n", num_assigned_tiles, num_unassigned_tiles);
  }
  dec->num_assigned = num_assigned_tiles;
  p->log_info.total_assigned_tiles_per_path += num_assigned_tiles;
  p->log_info.total_unassigned_tiles_per_path += num_unassigned_tiles;

  /* :967-971 */
  if (num_unassigned_tiles >= opt->unassigned_min && num_assigned_tiles <= opt->assigned_max) {
    assigned = 0;
  }
  const char* header_first_char = opt->silver_path ? "@" : ">";

  if (!assigned) {
    /* :978-1011 */
    if (opt->debug) {
      LOGF(p, "unassigned\n");
    }
    dec->decision = ORC_DEC_INSERT_WHOLE;
    ++p->ids_inserted;
    dec->first_id = p->ids_inserted;
    size_t block_start = 0;
    while (block_start < num_tiles) {
      size_t block_end = block_start + opt->block_size < num_tiles ? block_start + opt->block_size : num_tiles;
      uint32_t curr_ids_inserted = p->ids_inserted + (uint32_t)((block_start) / opt->block_size);
      orc_mibf_insert(p->mibf, (const uint64_t* const*)th.tile_hashes, th.tile_sizes, block_start, block_end, curr_ids_inserted);
      block_start = block_start + opt->block_size;
    }
    p->ids_inserted = p->ids_inserted + (uint32_t)(record->len / (opt->tile_length * opt->block_size));
    fprintf(p->out, "%s%s_untrimmed\n%s\n", header_first_char, record->id, record->seq);
    fflush(p->out);
    p->inserted_bases += record->len;
    ++p->log_info.num_reads_in_path;
    p->log_info.phred_sum_in_path += orc_sum_phred(record->qual, record->qlen);
    if (opt->silver_path) {
      fprintf(p->out, "+\n%s\n", record->qual);
      fflush(p->out);
      silver_path_check(p);
    }
  } else {
    if (num_assigned_tiles == num_tiles) {
      /* :1013-1023 */
      dec->decision = ORC_DEC_ASSIGNED_ALL;
      ++p->id;
      ++p->log_info.valid_reads;
      if (opt->debug) { /* :1016-1018 */
        LOGF(p, "complete assignment\n");
      }
      progress(p);
      goto cleanup;
    }
    long longest_start_idx, longest_end_idx;
    orc_find_longest_stretch(bools, num_tiles, &longest_start_idx, &longest_end_idx);
    size_t trim_start_idx, trim_end_idx;
    int good_flank = orc_eval_flanks(longest_start_idx, longest_end_idx, ids, num_tiles, &trim_start_idx, &trim_end_idx);
    if (good_flank) {
      /* :1038-1080 */
      assigned = 0;
      if (opt->debug) { /* :1037-1039 */
        LOGF(p, "trimmed\n");
      }
      dec->decision = ORC_DEC_INSERT_TRIMMED;
      dec->trim_start = trim_start_idx;
      dec->trim_end = trim_end_idx;
      ++p->ids_inserted;
      dec->first_id = p->ids_inserted;
      size_t block_start = trim_start_idx;
      while (block_start <= trim_end_idx) {
        size_t block_end = block_start + opt->block_size - 1 < trim_end_idx ? block_start + opt->block_size - 1 : trim_end_idx;
        uint32_t curr_ids_inserted = p->ids_inserted + (uint32_t)((block_start - trim_start_idx + 1) / opt->block_size);
        orc_mibf_insert(p->mibf, (const uint64_t* const*)th.tile_hashes, th.tile_sizes, block_start, block_end + 1, curr_ids_inserted);
        block_start = block_start + opt->block_size;
      }
      p->ids_inserted = p->ids_inserted + (uint32_t)((trim_end_idx - trim_start_idx) / opt->block_size);
      size_t off = trim_start_idx * opt->tile_length;
      size_t end_pos = (trim_end_idx == num_tiles - 1) ? (size_t)-1 : (trim_end_idx - trim_start_idx + 1) * opt->tile_length;
      /* std::string::substr(pos, n) clips n to size - pos */
      size_t seq_n = record->len - off;
      if (end_pos < seq_n) {
        seq_n = end_pos;
      }
      size_t qual_n = record->qlen >= off ? record->qlen - off : 0;
      if (end_pos < qual_n) {
        qual_n = end_pos;
      }
      p->inserted_bases += seq_n;
      fprintf(p->out, "%s%s_trimmed\n%.*s\n", header_first_char, record->id, (int)seq_n, record->seq + off);
      fflush(p->out);
      ++p->log_info.num_reads_in_path;
      p->log_info.phred_sum_in_path += orc_sum_phred(record->qual + (record->qlen >= off ? off : record->qlen), qual_n);
      if (opt->silver_path) {
        fprintf(p->out, "+\n%.*s\n", (int)qual_n, record->qual + (record->qlen >= off ? off : record->qlen));
        fflush(p->out);
        silver_path_check(p);
      }
    }
  }
  if (assigned) {
    if (opt->debug) { /* :1084-1086 */
      LOGF(p, "assigned\n");
    }
    dec->decision = ORC_DEC_ASSIGNED;
  }
  if (p->finished) {
    dec->finished = 1;
    goto cleanup; /* exit(0) happened inside silver_path_check: no ++id */
  }
  ++p->id;
  ++p->log_info.valid_reads;
  progress(p);
cleanup:
  for (size_t i = 0; i < num_tiles; ++i) {
    free(lists[i]);
  }
  free(lists);
  free(list_n);
  free(ids);
  free(bools);
  orc_tile_hashes_free(&th);
}

void
orc_path_close(orc_path* p)
{
  if (p) {
    orc_path_stop_producers(p);
  }
  if (!p) {
    return;
  }
  if (p->mibf && p->mibf->finalized && !p->finished) {
    /* :1257-1273 */
    if (p->opt.silver_path && p->opt.max_paths > p->curr_path) {
      LOGF(p,
           "WARNING: Expected %zu silver paths, but only %llu generated.\nPossible reasons include:\n"
           "\t- Input reads sorted by chromosome/position\n\t- Genome size set too large\n",
           p->opt.max_paths,
           (unsigned long long)p->curr_path);
    }
    if (p->opt.verbose) {
      log_path_stat(p);
    }
    LOGF(p, "assigned\nin %.4f\n", now_s() - p->t_assign_start);
  }
  if (p->out) {
    fclose(p->out);
  }
  orc_mibf_destroy(p->mibf);
  strset_free(&p->filter_out_reads);
  free(p);
}

orc_mibf*
orc_path_mibf(orc_path* p)
{
  return p->mibf;
}
const orc_log_info*
orc_path_log_info(const orc_path* p)
{
  return &p->log_info;
}
uint32_t
orc_path_phred_min(const orc_path* p)
{
  return p->opt.phred_min;
}
const char*
orc_path_seed(const orc_path* p, unsigned i)
{
  return p->seeds_str[i];
}
uint64_t
orc_path_filter_size(const orc_path* p)
{
  return p->filter_size;
}
int
orc_path_is_filtered(const orc_path* p, size_t idx)
{
  return strset_has(&p->filter_out_reads, p->reads->rec[idx].id);
}
void
orc_path_timers(const orc_path* p, double out[2])
{
  out[0] = p->t_fill;
  out[1] = now_s() - p->t_assign_start;
}

int
orc_main(int argc, char** argv)
{
  orc_opts o;
  orc_opts_default(&o);
  int ec = orc_process_options(&o, argc, argv);
  if (ec >= 0) {
    return ec;
  }
  orc_reads reads;
  if (orc_reads_load(&reads, o.input) != 0) {
    fprintf(stderr, "oracle: cannot read %s\n", o.input);
    return 1;
  }
  orc_path* p = orc_path_open(&o, &reads, stderr, &ec);
  if (!p) {
    orc_reads_free(&reads);
    return ec;
  }
  for (size_t i = 0; i < reads.n; ++i) {
    orc_decision d;
    orc_path_process_read(p, i, &d);
    if (d.finished) {
      break;
    }
  }
  orc_path_close(p);
  orc_reads_free(&reads);
  return 0;
}
