/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_mibf.h header comment).
 * PARITY UNPINNED (no reference golden vectors; reference unbuildable here).
 */
#include "orc_mibf.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

uint64_t
orc_calc_optimal_size(uint64_t entries, unsigned hash_num, double occupancy)
{
  /* MIBloomFilter.hpp:94-101 — note it always adds 1..64 bits */
  size_t non64 = (size_t)(-(double)entries * (double)hash_num / log(1.0 - occupancy));
  return non64 + (64 - non64 % 64);
}

orc_mibf*
orc_mibf_create(uint64_t m, unsigned h)
{
  orc_mibf* f = (orc_mibf*)calloc(1, sizeof(orc_mibf));
  if (!f) {
    return NULL;
  }
  f->m = m;
  f->h = h;
  f->n_words = (m + 63) / 64;
  f->bv = (uint64_t*)calloc(f->n_words ? f->n_words : 1, sizeof(uint64_t));
  if (!f->bv) {
    free(f);
    return NULL;
  }
  return f;
}

void
orc_mibf_destroy(orc_mibf* f)
{
  if (!f) {
    return;
  }
  free(f->bv);
  free(f->rank512);
  free(f->data);
  free(f->counts);
  free(f);
}

void
orc_mibf_insert_bv(orc_mibf* f, const uint64_t* hashes, size_t n)
{
  /* MIBFConstructSupport.hpp:134-147 */
  for (size_t i = 0; i < n; ++i) {
    uint64_t pos = hashes[i] % f->m;
    uint64_t* data_index = f->bv + (pos >> 6);
    uint64_t bit_mask = (uint64_t)1 << (pos & 0x3F);
    (void)__sync_fetch_and_or(data_index, bit_mask);
  }
}

void
orc_mibf_finalize(orc_mibf* f)
{
  /* rank support: ones before each 512-bit block (sdsl rank semantics) */
  uint64_t n_blocks = (f->n_words + 7) / 8;
  f->rank512 = (uint64_t*)malloc((n_blocks + 1) * sizeof(uint64_t));
  uint64_t acc = 0;
  for (uint64_t b = 0; b < n_blocks; ++b) {
    f->rank512[b] = acc;
    for (uint64_t w = b * 8; w < b * 8 + 8 && w < f->n_words; ++w) {
      acc += (uint64_t)__builtin_popcountll(f->bv[w]);
    }
  }
  f->rank512[n_blocks] = acc;
  /* getPop (MIBloomFilter.hpp:538-546): rank(last set bit) + 1 */
  f->pop = acc;
  f->data = (uint32_t*)calloc(f->pop ? f->pop : 1, sizeof(uint32_t));
  f->counts = (uint32_t*)calloc(f->pop ? f->pop : 1, sizeof(uint32_t));
  f->finalized = 1;
}

int
orc_mibf_bit(const orc_mibf* f, uint64_t pos)
{
  return (int)((f->bv[pos >> 6] >> (pos & 63)) & 1);
}

uint64_t
orc_mibf_rank(const orc_mibf* f, uint64_t pos)
{
  uint64_t blk = pos >> 9;
  uint64_t r = f->rank512[blk];
  uint64_t w0 = blk * 8;
  uint64_t wl = pos >> 6;
  for (uint64_t w = w0; w < wl; ++w) {
    r += (uint64_t)__builtin_popcountll(f->bv[w]);
  }
  unsigned bit = (unsigned)(pos & 63);
  if (bit) {
    r += (uint64_t)__builtin_popcountll(f->bv[wl] & (((uint64_t)1 << bit) - 1));
  }
  return r;
}

int
orc_mibf_at_rank(const orc_mibf* f, const uint64_t* hashes, uint64_t* rank_pos)
{
  /* MIBloomFilter.hpp:465-476 */
  for (unsigned i = 0; i < f->h; ++i) {
    uint64_t pos = hashes[i] % f->m;
    if (orc_mibf_bit(f, pos)) {
      rank_pos[i] = orc_mibf_rank(f, pos);
    } else {
      return 0;
    }
  }
  return 1;
}

uint64_t
orc_mibf_get_rank_pos(const orc_mibf* f, uint64_t hash)
{
  return orc_mibf_rank(f, hash % f->m);
}

void
orc_mibf_set_data(orc_mibf* f, uint64_t pos, uint32_t id)
{
  /* MIBloomFilter.hpp:593-602 */
  uint32_t old_value;
  do {
    old_value = f->data[pos];
    if (old_value > ORC_S_MASK) {
      id |= ORC_S_MASK;
    }
  } while (!__sync_bool_compare_and_swap(&f->data[pos], old_value, id));
}

static int
cmp_u64(const void* a, const void* b)
{
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return (x > y) - (x < y);
}

void
orc_mibf_insert(orc_mibf* f,
                const uint64_t* const* tile_hashes,
                const size_t* tile_sizes,
                size_t start,
                size_t end,
                uint32_t id)
{
  /* MIBFConstructSupport.hpp:247-283 */
  size_t vec_size = tile_sizes[0];
  size_t num_elements = 0;
  for (size_t i = start; i < end; ++i) {
    num_elements += tile_sizes[i];
  }
  if (num_elements == 0 || vec_size == 0) {
    return;
  }
  uint64_t* values = (uint64_t*)malloc(num_elements * sizeof(uint64_t));
  for (size_t i = 0; i < num_elements; ++i) {
    size_t vec_num = i / vec_size;
    size_t hash_loc = i % vec_size;
    uint64_t hash = tile_hashes[start + vec_num][hash_loc];
    values[i] = orc_mibf_get_rank_pos(f, hash);
  }
  /* dense_hash_set -> unique values (iteration order is irrelevant: every
   * unique rank is updated independently, :274-282) */
  qsort(values, num_elements, sizeof(uint64_t), cmp_u64);
  size_t n_unique = 0;
  for (size_t i = 0; i < num_elements; ++i) {
    if (i == 0 || values[i] != values[i - 1]) {
      values[n_unique++] = values[i];
    }
  }
#if defined(_OPENMP)
#pragma omp parallel for
#endif
  for (size_t i = 0; i < n_unique; ++i) {
    uint64_t rank = values[i];
    uint64_t random_seed = rank ^ (uint64_t)id;
    uint32_t count = __sync_add_and_fetch(&f->counts[rank], 1);
    /* std::hash<uint32_t>{}(uint64) : argument truncated to 32 bits, identity */
    uint32_t random_num = (uint32_t)random_seed % count;
    if (random_num == count - 1) {
      orc_mibf_set_data(f, rank, id);
    }
  }
  free(values);
}

void
orc_mibf_reset_ids(orc_mibf* f)
{
  memset(f->counts, 0, (size_t)f->pop * sizeof(uint32_t));
  memset(f->data, 0, (size_t)f->pop * sizeof(uint32_t));
}
