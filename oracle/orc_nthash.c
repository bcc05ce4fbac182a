/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_nthash.h header comment).
 * PARITY UNPINNED: restates btllib::SeedNtHash (third-party, un-vendored,
 * "btllib >=1.6.2" at /root/reference/requirements.txt:8).
 */
#include "orc_nthash.h"

#include <string.h>

uint64_t
orc_srol1(uint64_t x)
{
  /* bit 63 -> bit 33, bit 32 -> bit 0, everything else one to the left */
  uint64_t m = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
  return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | m;
}

uint64_t
orc_sror1(uint64_t x)
{
  uint64_t m = ((x & 0x200000000ULL) << 30) | ((x & 1ULL) << 32);
  return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | m;
}

uint64_t
orc_srol(uint64_t x, unsigned d)
{
  const uint64_t M33 = 0x1FFFFFFFFULL; /* low 33 bits */
  const uint64_t M31 = 0x7FFFFFFFULL;  /* 31 bits */
  uint64_t lo = x & M33;
  uint64_t hi = (x >> 33) & M31;
  unsigned rl = d % 33;
  unsigned rh = d % 31;
  if (rl) {
    lo = ((lo << rl) | (lo >> (33 - rl))) & M33;
  }
  if (rh) {
    hi = ((hi << rh) | (hi >> (31 - rh))) & M31;
  }
  return (hi << 33) | lo;
}

uint64_t
orc_base_seed(unsigned char c)
{
  switch (c) {
    case 'A':
    case 'a':
      return ORC_SEED_A;
    case 'C':
    case 'c':
      return ORC_SEED_C;
    case 'G':
    case 'g':
      return ORC_SEED_G;
    case 'T':
    case 't':
      return ORC_SEED_T;
    default:
      return ORC_SEED_N;
  }
}

unsigned char
orc_complement(unsigned char c)
{
  switch (c) {
    case 'A':
    case 'a':
      return 'T';
    case 'C':
    case 'c':
      return 'G';
    case 'G':
    case 'g':
      return 'C';
    case 'T':
    case 't':
      return 'A';
    default:
      return 'N';
  }
}

int
orc_seed_parse(orc_seed* s, const char* pattern)
{
  size_t n = strlen(pattern);
  if (n == 0 || n > ORC_MAX_SPAN) {
    return -1;
  }
  s->span = (unsigned)n;
  s->weight = 0;
  for (size_t q = 0; q < n; ++q) {
    if (pattern[q] == '1') {
      s->care[s->weight++] = (unsigned)q;
    } else if (pattern[q] != '0') {
      return -1;
    }
  }
  return 0;
}

uint64_t
orc_seed_hash_at(const orc_seed* s, const char* seq, size_t p)
{
  uint64_t fwd = 0, rev = 0;
  const unsigned K = s->span;
  for (unsigned i = 0; i < s->weight; ++i) {
    unsigned q = s->care[i];
    unsigned char c = (unsigned char)seq[p + q];
    fwd ^= orc_srol(orc_base_seed(c), K - 1 - q);
    rev ^= orc_srol(orc_base_seed(orc_complement(c)), q);
  }
  return fwd + rev;
}

size_t
orc_multi_hash(const orc_seed* seeds,
               unsigned h,
               const char* seq,
               size_t len,
               uint64_t* out,
               size_t out_cap)
{
  /* multiLensfrHashIterator.hpp:29-43: each seed's first roll() positions it
   * at 0; :49-68: ++ rolls every seed, frame exists while any seed rolled. */
  uint64_t cur[ORC_MAX_SEEDS];
  size_t n_valid[ORC_MAX_SEEDS]; /* number of valid positions of seed s */
  size_t frames = 0;
  for (unsigned s = 0; s < h; ++s) {
    n_valid[s] = (len >= seeds[s].span) ? (len - seeds[s].span + 1) : 0;
    if (n_valid[s] > frames) {
      frames = n_valid[s];
    }
    cur[s] = 0;
  }
  if (out == NULL) {
    return frames;
  }
  for (size_t f = 0; f < frames; ++f) {
    for (unsigned s = 0; s < h; ++s) {
      if (f < n_valid[s]) {
        cur[s] = orc_seed_hash_at(&seeds[s], seq, f);
      } /* else: stale, keeps the last valid value */
      if (f * h + s < out_cap) {
        out[f * h + s] = cur[s];
      }
    }
  }
  return frames;
}
