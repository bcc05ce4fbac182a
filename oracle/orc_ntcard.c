/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_ntcard.h). */
#include "orc_ntcard.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>

orc_ntcard*
orc_ntcard_new(unsigned h, uint64_t input_bytes)
{
  orc_ntcard* nc = (orc_ntcard*)calloc(1, sizeof(orc_ntcard));
  nc->h = h;
  nc->sbits = input_bytes < 50000000000ULL ? 7u : 11u; /* ntcard.hpp:177-178 */
  nc->counters = (uint16_t*)calloc((size_t)h * ((size_t)ORC_NTC_NSAMP << ORC_NTC_RBITS), sizeof(uint16_t));
  return nc;
}

void
orc_ntcard_free(orc_ntcard* nc)
{
  if (nc) {
    free(nc->counters);
    free(nc);
  }
}

/* ntComp (ntcard.hpp:81-94) with a multiplicity */
static void
nt_comp(orc_ntcard* nc, unsigned seed, uint64_t hval, uint64_t times)
{
  const unsigned sbits = nc->sbits;
  const uint64_t smask = (((uint64_t)1) << (sbits - 1)) - 1; /* :182 */
  const uint64_t rbuck = ((uint64_t)1) << ORC_NTC_RBITS;
  uint64_t ind = ORC_NTC_NSAMP;
  if (hval >> (63 - sbits) == 1) {
    ind = 0;
  }
  if (hval >> (64 - sbits) == smask) {
    ind = 1;
  }
  if (ind < ORC_NTC_NSAMP) {
    uint16_t* t = nc->counters + (size_t)seed * ((size_t)ORC_NTC_NSAMP << ORC_NTC_RBITS);
    const uint64_t sh = hval & (rbuck - 1);
    t[ind * rbuck + sh] = (uint16_t)(t[ind * rbuck + sh] + times);
  }
}

static int
is_acgt(char c)
{
  switch (c) {
    case 'A': case 'C': case 'G': case 'T':
    case 'a': case 'c': case 'g': case 't':
      return 1;
    default:
      return 0;
  }
}

void
orc_ntcard_add_read(orc_ntcard* nc, const orc_seed* seeds, const char* seq, size_t len)
{
  /* valid windows per seed (see the header): a running count of clean characters */
  uint64_t V[ORC_MAX_SEEDS];
  uint64_t last_hash[ORC_MAX_SEEDS];
  uint64_t F = 0;
  for (unsigned s = 0; s < nc->h; ++s) {
    const size_t K = seeds[s].span;
    V[s] = 0;
    last_hash[s] = 0;
    size_t clean = 0;
    for (size_t e = 0; e < len; ++e) {
      clean = is_acgt(seq[e]) ? clean + 1 : 0;
      if (clean >= K) {
        const uint64_t hv = orc_seed_hash_at(&seeds[s], seq, e + 1 - K);
        nt_comp(nc, s, hv, 1);
        last_hash[s] = hv;
        ++V[s];
      }
    }
    if (V[s] > F) {
      F = V[s];
    }
  }
  for (unsigned s = 0; s < nc->h; ++s) {
    if (V[s] && F > V[s]) {
      nt_comp(nc, s, last_hash[s], F - V[s]); /* stale repeats of the iterator */
    }
    nc->tot_kmers[s] += F; /* ++totKmer once per frame and seed (:107) */
  }
}

uint64_t
orc_ntcard_zero_buckets(const orc_ntcard* nc, unsigned seed, unsigned samp)
{
  const size_t rbuck = ((size_t)1) << ORC_NTC_RBITS;
  const uint16_t* t = nc->counters + (size_t)seed * ((size_t)ORC_NTC_NSAMP << ORC_NTC_RBITS) + (size_t)samp * rbuck;
  uint64_t z = 0;
  for (size_t j = 0; j < rbuck; ++j) {
    z += t[j] == 0;
  }
  return z;
}

uint64_t
orc_ntcard_f0(const orc_ntcard* nc, unsigned seed)
{
  /* compEst: p[i][0] = zero buckets; pMean[0] = (p[0][0] + p[1][0]) / (1.0 * nSamp)  (:124-133) */
  double pmean0 = 0.0;
  for (unsigned i = 0; i < ORC_NTC_NSAMP; ++i) {
    pmean0 += (double)(unsigned)orc_ntcard_zero_buckets(nc, seed, i);
  }
  pmean0 /= 1.0 * ORC_NTC_NSAMP;
  /* :135-136 */
  const double f0mean = (double)(ssize_t)((ORC_NTC_RBITS * log(2) - log(pmean0)) * 1.0 * (double)((size_t)1 << (nc->sbits + ORC_NTC_RBITS)));
  return (uint64_t)(size_t)f0mean; /* histArray[1] = (size_t)F0Mean (:232) */
}
