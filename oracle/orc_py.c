/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Flat helper entry points for the ctypes
 * binding in oracle/orc.py (tests, smoke check, bench cpu_baseline leg).
 * PARITY UNPINNED (see orc_path.h).
 */
#include "orc_path.h"

#include <stdlib.h>
#include <string.h>

orc_seed*
orcpy_seeds_new(const char* const* patterns, unsigned h)
{
  orc_seed* s = (orc_seed*)calloc(h ? h : 1, sizeof(orc_seed));
  for (unsigned i = 0; i < h; ++i) {
    if (orc_seed_parse(&s[i], patterns[i]) != 0) {
      free(s);
      return NULL;
    }
  }
  return s;
}

void
orcpy_seeds_free(orc_seed* s)
{
  free(s);
}

/* hashes of tile `tile_idx` of a read exactly as read_hashing.cpp:44-53 */
size_t
orcpy_tile_hashes(const orc_seed* seeds, unsigned h, const char* seq, size_t len, size_t tile_size, size_t k, size_t tile_idx, uint64_t* out, size_t cap)
{
  size_t start = tile_idx * tile_size;
  size_t tl = tile_size + k - 1;
  if (start + tl > len) {
    tl = len - start;
  }
  size_t frames = orc_multi_hash(seeds, h, seq + start, tl, out, cap);
  return frames * h;
}

/* whole-read insertBV (goldrush_path.cpp:304-305) */
void
orcpy_bv_insert_read(orc_mibf* f, const orc_seed* seeds, unsigned h, const char* seq, size_t len)
{
  size_t frames = orc_multi_hash(seeds, h, seq, len, NULL, 0);
  uint64_t* hv = (uint64_t*)malloc((frames * h + 1) * sizeof(uint64_t));
  orc_multi_hash(seeds, h, seq, len, hv, frames * h);
  orc_mibf_insert_bv(f, hv, frames * h);
  free(hv);
}

/* the same for n reads, OpenMP over the reads like the reference's fill loop
 * (goldrush_path.cpp:257-305: insertBV is called from an `omp parallel` region) */
void
orcpy_bv_insert_reads(orc_mibf* f, const orc_seed* seeds, unsigned h, const char* const* seqs, const size_t* lens, size_t n)
{
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (size_t i = 0; i < n; ++i) {
    orcpy_bv_insert_read(f, seeds, h, seqs[i], lens[i]);
  }
}

/* insertMIBF for tiles [start,end) of one read (goldrush_path.cpp:988-989) */
void
orcpy_insert_read_tiles(orc_mibf* f, const orc_seed* seeds, unsigned h, const char* seq, size_t len, size_t tile_size, size_t k, size_t start, size_t end, uint32_t id)
{
  orc_tile_hashes th;
  orc_hash_read_tiles(&th, seq, len, tile_size, k, seeds, h, 1);
  orc_mibf_insert(f, (const uint64_t* const*)th.tile_hashes, th.tile_sizes, start, end, id);
  orc_tile_hashes_free(&th);
}

uint64_t* orcpy_mibf_bv(orc_mibf* f) { return f->bv; }
uint64_t orcpy_mibf_n_words(const orc_mibf* f) { return f->n_words; }
uint64_t orcpy_mibf_pop(const orc_mibf* f) { return f->pop; }
uint32_t* orcpy_mibf_data(orc_mibf* f) { return f->data; }
uint32_t* orcpy_mibf_counts(orc_mibf* f) { return f->counts; }
uint64_t orcpy_mibf_m(const orc_mibf* f) { return f->m; }
size_t orcpy_sizeof_opts(void) { return sizeof(orc_opts); }
size_t orcpy_sizeof_decision(void) { return sizeof(orc_decision); }

#include "orc_ntcard.h"
uint16_t*
orcpy_ntcard_counters(orc_ntcard* nc)
{
  return nc->counters;
}
