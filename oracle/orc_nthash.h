/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * compile, link, import or execute anything under oracle/.
 *
 * PARITY UNPINNED at this file: the arithmetic restated here lives in the
 * un-vendored third-party dependency btllib (requirements.txt:8
 * "btllib >=1.6.2", no exact pin; README.md:142), class btllib::SeedNtHash
 * (ntHash2 spaced-seed hash).  It is absent from /root/reference, not
 * installed in this image, and the reference holds no golden vectors or
 * known-answer tests for it.  What follows restates btllib's published
 * algorithm (nthash_consts.hpp / nthash_lowlevel.hpp / nthash.hpp,
 * v1.4.x..1.7.x); anchors are the reference's call sites
 * goldrush_path/multiLensfrHashIterator.hpp:39-41,54,60.
 *
 * HOW TO PIN IT (one command, for anyone with a real btllib install):
 *   python3 tools/make_btllib_kat.py
 * runs btllib.SeedNtHash exactly as multiLensfrHashIterator.hpp:39-41,54,60
 * does (one object per seed, roll() / hashes()[0]) over tests/golden/tiny.fq
 * with the pipeline's seeds at h = 3 and h = 5 and a family spanning 60..64
 * bases, and writes tests/golden/btllib_seed_kat.json.  With that file in
 * place tests/test_oracle.py::test_seed_hashes_match_a_real_btllib (this
 * restatement) and tests/test_gpu_parity.py::test_seed_hashes_match_a_real_btllib
 * (the HIP kernels) stop skipping and check every position against btllib's
 * values.  Until then: unpinned, as said above.
 *
 * Definition restated (closed form; btllib evaluates the same value
 * incrementally over "care blocks" and "monomers"):
 *   seed of span K with care set C = { q : seed[q] == '1' }
 *   fwd(p) = XOR_{q in C} srol( SEED[x[p+q]],        K-1-q )
 *   rev(p) = XOR_{q in C} srol( SEED[comp(x[p+q])],  q     )
 *   hash(p) = fwd(p) + rev(p)   (mod 2^64)   -- ntHash2 canonical = sum
 * srol = "split rotate left": bits [0,33) and bits [33,64) of the word are
 * rotated left independently (a 33-bit and a 31-bit rotation).
 */
#ifndef ORC_NTHASH_H
#define ORC_NTHASH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SEEDS 16
#define ORC_MAX_SPAN 256

/* btllib nthash_consts.hpp: per-base 64-bit seeds */
#define ORC_SEED_A 0x3c8bfbb395c60474ULL
#define ORC_SEED_C 0x3193c18562a02b4cULL
#define ORC_SEED_G 0x20323ed082572324ULL
#define ORC_SEED_T 0x295549f54be24456ULL
#define ORC_SEED_N 0x0000000000000000ULL

/* one-step split rotate left / right (btllib srol(x) / sror(x)) */
uint64_t orc_srol1(uint64_t x);
uint64_t orc_sror1(uint64_t x);
/* d-step split rotate left (btllib srol(x, d)), any d >= 0 */
uint64_t orc_srol(uint64_t x, unsigned d);

/* 64-bit seed value of an ASCII base (upper or lower case); 0 for non-ACGT */
uint64_t orc_base_seed(unsigned char c);
/* ASCII complement (A<->T, C<->G), other characters map to 'N' */
unsigned char orc_complement(unsigned char c);

/* One spaced seed, parsed. */
typedef struct
{
  unsigned span;               /* K = strlen(seed string) */
  unsigned weight;             /* |C| */
  unsigned care[ORC_MAX_SPAN]; /* care positions, ascending */
} orc_seed;

/* parse "1011..." ; returns 0 on success */
int orc_seed_parse(orc_seed* s, const char* pattern);

/* closed-form hash of the window starting at seq[p] (needs p + span <= len) */
uint64_t orc_seed_hash_at(const orc_seed* s, const char* seq, size_t p);

/*
 * multiLensfrHashIterator restated (multiLensfrHashIterator.hpp:29-73).
 * h independent seeds (spans may differ) over one sequence; per frame an
 * array of h values; iteration continues while ANY seed can still roll;
 * a seed that can no longer roll keeps its last value (stale hash).
 *
 * Writes frames*h values, frame-major [f*h + s] (the layout
 * read_hashing.cpp:47-53 produces) into out (capacity in values) and
 * returns the number of frames.  If out is NULL only counts frames.
 * Sequences shorter than the shortest span produce 0 frames.  Behaviour
 * for a sequence shorter than the LONGEST span is undefined in the
 * reference (SeedNtHash is constructed on a too-short string); here such
 * a seed contributes the value 0 until/unless it becomes valid, and this
 * case is never reached by the hot path (tiles are >= tile_length long).
 */
size_t orc_multi_hash(const orc_seed* seeds,
                      unsigned h,
                      const char* seq,
                      size_t len,
                      uint64_t* out,
                      size_t out_cap);

#ifdef __cplusplus
}
#endif
#endif
