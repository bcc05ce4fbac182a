/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * compile, link, import or execute anything under oracle/.
 *
 * CPU restatement of the LIVE subset of
 *   goldrush_path/MIBloomFilter.hpp        (query side)
 *   goldrush_path/MIBFConstructSupport.hpp (bit-vector fill, ID insert)
 * The sdsl-lite containers those files use (bit_vector, bit_vector_il<512>,
 * rank_support_il<1>; un-vendored, requirements.txt:4, no version) are
 * restated by their positional semantics only: bit i of a zero-initialised
 * m-bit vector; rank(i) = number of set bits in [0, i).
 * google::dense_hash_set (requirements.txt:7) is used by the reference only
 * as "the set of distinct ranks" and is restated as sort + unique.
 *
 * PARITY UNPINNED: the reference holds no unit tests, golden vectors or
 * fixtures for this path and cannot be built here.
 */
#ifndef ORC_MIBF_H
#define ORC_MIBF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MIBloomFilter.hpp:38-39 */
#define ORC_S_MASK 0x80000000u
#define ORC_S_ANTIMASK 0x7FFFFFFFu

typedef struct
{
  uint64_t m;        /* filter size in bits (MIBFConstructSupport m_filterSize) */
  unsigned h;        /* number of hash functions / seeds */
  uint64_t* bv;      /* plain bit vector, word i>>6, bit i&63 */
  uint64_t n_words;
  uint64_t* rank512; /* ones before each 512-bit block (built by finalize) */
  uint64_t pop;      /* MIBloomFilter::getPop() */
  uint32_t* data;    /* m_data[pop]  (IDs)     MIBloomFilter.hpp:758 */
  uint32_t* counts;  /* m_counts[pop]          MIBFConstructSupport.hpp:338 */
  int finalized;
} orc_mibf;

/* MIBloomFilter::calcOptimalSize (MIBloomFilter.hpp:94-101) */
uint64_t orc_calc_optimal_size(uint64_t entries, unsigned hash_num, double occupancy);

/* MIBFConstructSupport ctor with explicit filter size (:66-84) */
orc_mibf* orc_mibf_create(uint64_t m, unsigned h);
void orc_mibf_destroy(orc_mibf* f);

/* insertBV(H&) (MIBFConstructSupport.hpp:134-147) over frames*h hash values;
 * thread-safe (atomic OR) like the reference */
void orc_mibf_insert_bv(orc_mibf* f, const uint64_t* hashes, size_t n);

/* setup() + getEmptyMIBF() (:165-181), MIBloomFilter ctor + getPop
 * (MIBloomFilter.hpp:165-184, 538-546) */
void orc_mibf_finalize(orc_mibf* f);

/* bit test and rank(pos) = ones in [0,pos) */
int orc_mibf_bit(const orc_mibf* f, uint64_t pos);
uint64_t orc_mibf_rank(const orc_mibf* f, uint64_t pos);

/* atRank (MIBloomFilter.hpp:465-476): 1 if all h bits set, fills rank_pos */
int orc_mibf_at_rank(const orc_mibf* f, const uint64_t* hashes, uint64_t* rank_pos);
/* getRankPos(hash) (:488-491) */
uint64_t orc_mibf_get_rank_pos(const orc_mibf* f, uint64_t hash);
/* setData (:593-602) */
void orc_mibf_set_data(orc_mibf* f, uint64_t pos, uint32_t id);

/*
 * insertMIBF(miBF, hash_vec, start, end, id)
 * (MIBFConstructSupport.hpp:247-283).  tile_hashes[t] points at tile t's
 * flat hash array of tile_sizes[t] values.
 */
void orc_mibf_insert(orc_mibf* f,
                     const uint64_t* const* tile_hashes,
                     const size_t* tile_sizes,
                     size_t start,
                     size_t end,
                     uint32_t id);

/* reset_counts (:183-186) + reset_ID_vector (MIBloomFilter.hpp:679-682) */
void orc_mibf_reset_ids(orc_mibf* f);

#ifdef __cplusplus
}
#endif
#endif
