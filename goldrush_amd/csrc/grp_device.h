// Device-side data layout and helpers of the gfx950 GoldRush-Path engine.
//
// HBM layout (DESIGN.md "Data layout")
//
// phase 1 (bit-vector fill, before grp_finalize)
//   bv[ceil(m/32)]  uint32  plain bit vector, bit i = word i>>5, bit i&31
//                           (byte-identical to sdsl::bit_vector's 64-bit words,
//                           MIBFConstructSupport.hpp:140-142)
//
// phase 2 (after grp_finalize): the reference keeps the interleaved bit vector
// + rank support (MIBloomFilter.hpp:757-759) and a separate ID array indexed by
// rank (:758) — two dependent random DRAM accesses per probe.  Random 64-byte
// gathers are bound by the request rate of the memory system on MI355X (~48 G lines/s,
// tools/gather_bench.hip), so here the bits and THEIR IDs share one 64-byte
// line and a probe costs one line:
//   buckets[n_buckets]  128-byte UNITS (round 5), 128-byte aligned:
//     line 0, the QUERY's line (all a probe reads):
//       uint32 rel        ones before this bucket, relative to its superbucket
//       uint64 bitmap     W <= 64 consecutive filter bits (bit j = position b*W+j)
//       uint32 ids[13]    ID of the bucket's j-th set bit (MIBloomFilter m_data)
//     line 1, the INSERT's line (touched by inserts only):
//       uint64 cw[8]      count word of the bucket's j-th set bit, j < 8: bits 0..27 the insert count
//                         (MIBFConstructSupport m_counts, :338), bits 28..37 the epoch of the batch that claimed the
//                         rank last, bits 38..63 the claiming record of that batch (grp_batch.inc)
//   W is chosen at finalize from the measured occupancy so that a bucket holds
//   ~6 set bits on average; the rare bucket with more than 13 set bits keeps the
//   IDs of its 14th.. set bits in a small open-addressing table keyed by rank, and the count words of a bucket's
//   9th.. set bits live in a second one (`far`, keys written once at finalize: ~5 % of the ranks at W >= 13 and the
//   design occupancy).
//   Why a unit (round 5, tools/collect_matrix.sh -> profiles/r05_collect_matrix.txt): an insert dirties the ID (line 0)
//   and the count (rounds 1-4: an array indexed by rank, another random line).  Two lines of ONE 128-byte block cost
//   the memory system 0.7 x what two unrelated lines cost (17.6 against 12.3 G records/s; already 256 bytes apart the
//   gain is gone), and the query still reads one 64-byte line.
//   super[n_super]   uint64  absolute ones before each superbucket (2^22 buckets)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GRP_DEV_MAX_H 8
#define GRP_DEV_MAX_W 32
#define GRP_BUCKET_IDS 13u
#define GRP_NEAR_CW 8u           /* count words in the unit's insert line */
#define GRP_UNIT_U4 8u           /* uint4 per bucket unit (128 B) */
#define GRP_UNIT_DW 32u          /* dwords per unit: dword 3 + j of a unit = the ID of its j-th set bit */
#define GRP_UNIT_QW 16u          /* 64-bit words per unit: word 8 + j = the count word of its j-th set bit */
#define GRP_SUPER_SHIFT 22       /* buckets per superbucket = 2^22 (rel < 2^28) */
#define GRP_CHUNK_BUCKETS 4096u  /* rank-build chunk; divides the superbucket */

struct DevSeeds
{
  uint32_t h;
  uint32_t k;    // base span
  uint32_t wmax; // max weight over seeds
  // Seeds built by make_seed_pattern (spaced_seeds.cpp:58-66) are seed 0 with i don't-care
  // positions inserted at the centre: seed i = left || 0^i || right.  n_left > 0 says so (the
  // number of care positions of the left part): the H hashes of a frame then share their halves
  // (grp_kernels.inc, seed_halves), 16 table look-ups per frame instead of 16 per seed.
  uint32_t n_left;
  uint32_t wide; // 1: the longest span exceeds 32 bases (k + h - 1 <= 64): hashes read a 128-bit window (grp_kernels.inc, seed_hash)
  uint32_t weight[GRP_DEV_MAX_H];
  uint32_t span[GRP_DEV_MAX_H];
  uint32_t shift[GRP_DEV_MAX_H][GRP_DEV_MAX_W]; // 2*q for care position q
  // [seed][care index][2-bit base] = { srol(SEED[b], K-1-q), srol(SEED[comp b], q) }
  ulonglong2 tab[GRP_DEV_MAX_H][GRP_DEV_MAX_W][4];
};

struct DevFilter
{
  uint32_t* bv;                 // phase 1
  uint4* buckets;               // phase 2: units of 8 x uint4 (query line, insert line)
  const uint64_t* super;
  ulonglong2* far;              // {rank + 1, count word} of the ranks beyond a bucket's 8th set bit; keys immutable after grp_finalize
  uint64_t far_mask;
  unsigned long long* ovf_keys; // rank + 1, 0 = empty
  uint32_t* ovf_ids;
  uint64_t ovf_mask;
  uint64_t m;       // filter bits
  uint64_t m_inv;   // floor((2^64-1)/m)
  uint64_t w_magic; // floor(2^64 / W) + 1
  uint64_t n_buckets;
  uint64_t pop;
  uint32_t W;       // filter bits per bucket (13..64)
  uint32_t pad;
};

constexpr unsigned long long GRP_CNT_MASK = (1ull << 28) - 1ull;
constexpr uint32_t GRP_EPOCH_MAX = (1u << 10) - 1u; // epochs 1 .. 1023, 0: never claimed

__host__ __device__ inline uint32_t
grp_cnt(unsigned long long w)
{
  return (uint32_t)(w & GRP_CNT_MASK);
}

// the word with its count replaced (epoch and claim kept)
__host__ __device__ inline unsigned long long
grp_cnt_with(unsigned long long w, uint32_t count)
{
#if defined(__HIP_DEVICE_COMPILE__)
  if (count > (uint32_t)GRP_CNT_MASK) {
    __builtin_trap(); // 2^28 inserts on one rank: fail loudly rather than wrap
  }
#endif
  return (w & ~GRP_CNT_MASK) | (unsigned long long)count;
}

__host__ __device__ inline uint32_t
grp_claim_epoch(unsigned long long w)
{
  return (uint32_t)(w >> 28) & GRP_EPOCH_MAX;
}

__host__ __device__ inline uint32_t
grp_claim_record(unsigned long long w)
{
  return (uint32_t)(w >> 38);
}

__host__ __device__ inline unsigned long long
grp_claim_word(uint32_t record, uint32_t epoch, uint32_t count)
{
  return ((unsigned long long)record << 38) | ((unsigned long long)epoch << 28) | (unsigned long long)count;
}

struct DevReads
{
  const uint32_t* packed;
  const uint64_t* word_off;  // [n_reads+1]
  const uint32_t* len;       // [n_reads]
  const uint64_t* tile0;     // [n_reads+1] first tile of each read
  const uint32_t* tile_read; // [n_tiles]   read of each tile
  const uint64_t* chunk0;    // [n_reads+1] first fill chunk of each read
  const uint32_t* chunk_read; // [n_chunks]
};

// x mod m for a run-time 64-bit m (MIBloomFilter.hpp:468 `hashes[i] % m_bv.size()`).
// q' = mulhi(x, floor((2^64-1)/m)) is q or q-1, so one conditional subtract is exact.
__host__ __device__ inline uint64_t
grp_mod_m(uint64_t x, uint64_t m, uint64_t m_inv)
{
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t q = __umul64hi(x, m_inv);
#else
  uint64_t q = (uint64_t)(((unsigned __int128)x * m_inv) >> 64);
#endif
  uint64_t r = x - q * m;
  if (r >= m) {
    r -= m;
  }
  return r;
}

// pos / W for a run-time W in [13,64]: exact for pos < 2^58 with
// magic = floor(2^64 / W) + 1
__host__ __device__ inline uint64_t
grp_div_w(uint64_t pos, uint64_t w_magic)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(pos, w_magic);
#else
  return (uint64_t)(((unsigned __int128)pos * w_magic) >> 64);
#endif
}

// one probe, first access: bucket index, bit offset
struct Probe
{
  uint64_t b;
  uint32_t off;
};

__device__ inline Probe
grp_locate(const DevFilter& f, uint64_t hash)
{
  Probe p;
  const uint64_t pos = grp_mod_m(hash, f.m, f.m_inv);
  p.b = grp_div_w(pos, f.w_magic);
  p.off = (uint32_t)(pos - p.b * f.W);
  return p;
}

// dword index (into f.buckets) of the ID slot of a bucket's lr-th set bit; dword 0 of the unit: the ID lives in the
// overflow table (`loc & 31` tells)
__host__ __device__ inline unsigned long long
grp_id_loc(uint64_t b, uint32_t lr)
{
  return (lr < GRP_BUCKET_IDS) ? (unsigned long long)(b * GRP_UNIT_DW + 3u + lr) : (unsigned long long)(b * GRP_UNIT_DW);
}

// the count word of a rank beyond its bucket's 8th set bit (the key was written at finalize)
__device__ inline unsigned long long*
grp_far_ptr(const DevFilter& f, uint64_t rank)
{
  uint64_t slot = ((rank + 1) * 0x9E3779B97F4A7C15ULL) >> 19 & f.far_mask;
  for (;;) {
    const unsigned long long k = f.far[slot].x;
    if (k == rank + 1) {
      return &f.far[slot].y;
    }
    if (k == 0) {
      __builtin_trap(); // a rank without a count word: the far table is complete by construction
    }
    slot = (slot + 1) & f.far_mask;
  }
}

// the count word (count | epoch | claiming record) of a bucket's lr-th set bit
__device__ inline unsigned long long*
grp_cw_ptr(const DevFilter& f, uint64_t b, uint32_t lr, uint64_t rank)
{
  if (lr < GRP_NEAR_CW) {
    return reinterpret_cast<unsigned long long*>(f.buckets) + b * GRP_UNIT_QW + 8u + lr;
  }
  return grp_far_ptr(f, rank);
}

// ... from the rank's ID location (grp_id_loc)
__device__ inline unsigned long long*
grp_cw_ptr_loc(const DevFilter& f, unsigned long long loc, uint64_t rank)
{
  const uint32_t dw = (uint32_t)(loc & (GRP_UNIT_DW - 1u));
  return grp_cw_ptr(f, loc / GRP_UNIT_DW, dw ? dw - 3u : GRP_BUCKET_IDS, rank);
}

// header = first 16 bytes of a bucket: {rel, bitmap lo, bitmap hi, ids[0]}
__device__ inline uint64_t
grp_bitmap(const uint4& hd)
{
  return (uint64_t)hd.y | ((uint64_t)hd.z << 32);
}

__device__ inline uint32_t
grp_local_rank(uint64_t bm, uint32_t off)
{
  return (uint32_t)__popcll(bm & ((1ull << off) - 1ull)); // off < 64
}

// open-addressing table for the IDs of a bucket's 14th.. set bits
__device__ inline uint32_t
grp_ovf_get(const DevFilter& f, uint64_t rank)
{
  uint64_t slot = (rank * 0x9E3779B97F4A7C15ULL) >> 17 & f.ovf_mask;
  for (;;) {
    const unsigned long long k = f.ovf_keys[slot];
    if (k == rank + 1) {
      return f.ovf_ids[slot];
    }
    if (k == 0) {
      return 0u;
    }
    slot = (slot + 1) & f.ovf_mask;
  }
}

__device__ inline void
grp_ovf_put(const DevFilter& f, uint64_t rank, uint32_t id)
{
  uint64_t slot = (rank * 0x9E3779B97F4A7C15ULL) >> 17 & f.ovf_mask;
  for (;;) {
    unsigned long long k = f.ovf_keys[slot];
    if (k == 0) {
      k = atomicCAS(&f.ovf_keys[slot], 0ull, (unsigned long long)(rank + 1));
      if (k == 0) {
        k = rank + 1;
      }
    }
    if (k == rank + 1) {
      f.ovf_ids[slot] = id;
      return;
    }
    slot = (slot + 1) & f.ovf_mask;
  }
}
