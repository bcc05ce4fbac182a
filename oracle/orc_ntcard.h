/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * compile, link, import or execute anything under oracle/.
 *
 * PARITY UNPINNED (see orc_nthash.h): the reference holds no test or golden
 * vector for this path either.
 *
 * `--ntcard` of goldrush-path restated: the ntCard
 * cardinality estimator run on the spaced-seed hash stream of every read,
 * goldrush_path/ntcard.hpp:81-112 (ntComp, stRead), :114-154 (compEst),
 * :156-246 (getHist), :248-275 (calc_ntcard_genome_size), called from
 * goldrush_path.cpp:1109-1112.  Only histArray[i][1] (= F0, the estimated
 * number of distinct hashes of seed i) is consumed by the caller, and F0 only
 * depends on the number of ZERO buckets of the two sample tables.
 *
 * Hash stream: multiLensfrHashIterator over the whole record sequence
 * (ntcard.hpp:100-111) — no read filter applies here, so sequences with
 * non-ACGT characters reach btllib::SeedNtHash.  Restated from btllib's
 * published roll()/init(): a seed of span K visits,
 * in ascending order, every window [p, p+K) that holds only ACGT/acgt (a
 * non-ACGT character anywhere in the span, care position or not, invalidates
 * the window).  The iterator yields one frame per step while ANY seed still
 * rolls; a seed that cannot roll keeps its last value, which is therefore
 * counted again (stale repeats, as in the fill path).
 *   V_i = number of valid windows of seed i, F = max_i V_i frames;
 *   seed i contributes each valid window once, and its LAST valid window
 *   F - V_i more times.
 * Reference-undefined and defined here: a seed with V_i = 0 (sequence shorter
 * than its span, or no clean window) would expose btllib's uninitialised hash
 * array F times; here it contributes nothing.
 */
#ifndef ORC_NTCARD_H
#define ORC_NTCARD_H

#include "orc_nthash.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NTC_RBITS 27u /* nts::rBits, ntcard.hpp:34 */
#define ORC_NTC_NSAMP 2u  /* nts::nSamp, ntcard.hpp:38 */

typedef struct
{
  unsigned h;
  unsigned sbits;      /* nts::sBits: 7 below 50e9 input bytes, else 11 (ntcard.hpp:35,177-178) */
  uint16_t* counters;  /* [h][nSamp << rBits], wrap mod 2^16 like the reference's uint16_t */
  uint64_t tot_kmers[ORC_MAX_SEEDS]; /* histArray[i][0] */
} orc_ntcard;

orc_ntcard* orc_ntcard_new(unsigned h, uint64_t input_bytes);
void orc_ntcard_free(orc_ntcard* nc);
/* stRead (ntcard.hpp:96-112) */
void orc_ntcard_add_read(orc_ntcard* nc, const orc_seed* seeds, const char* seq, size_t len);
/* number of zero buckets of sample table `samp` of seed i */
uint64_t orc_ntcard_zero_buckets(const orc_ntcard* nc, unsigned seed, unsigned samp);
/* compEst's F0Mean, cast as getHist stores it (ntcard.hpp:135-136,232) */
uint64_t orc_ntcard_f0(const orc_ntcard* nc, unsigned seed);

#ifdef __cplusplus
}
#endif
#endif
