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
// gathers are bound by DRAM row activations on MI355X (~55 G sectors/s,
// tools/gather_bench.hip), so here the bits and THEIR IDs share one 64-byte
// bucket and a probe costs one sector:
//   buckets[n_buckets]  64 B each:
//       uint32 rel        ones before this bucket, relative to its superbucket
//       uint64 bitmap     W <= 64 consecutive filter bits (bit j = position b*W+j)
//       uint32 ids[13]    ID of the bucket's j-th set bit (MIBloomFilter m_data)
//   W is chosen at finalize from the measured occupancy so that a bucket holds
//   ~6 set bits on average; the rare bucket with more than 13 set bits keeps the
//   IDs of its 14th.. set bits in a small open-addressing table keyed by rank.
//   super[n_super]   uint64  absolute ones before each superbucket (2^22 buckets)
//   counts[pop]      uint64  bits 0..27: insert count (MIBFConstructSupport m_counts, :338), indexed by
//                            global rank; touched by inserts only.  Round 4: bits 28..37 the epoch of the
//                            batch that claimed the rank last, bits 38..63 the claiming record of that batch
//                            (grp_batch.inc: the claim of the collect pass and the count share one word — one
//                            random line per record instead of two; a count that would not fit 28 bits traps)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GRP_DEV_MAX_H 8
#define GRP_DEV_MAX_W 32
#define GRP_BUCKET_IDS 13u
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
  uint4* buckets;               // phase 2: 4 x uint4 per bucket
  const uint64_t* super;
  unsigned long long* counts;   // per rank: insert count | batch epoch | claiming record (grp_cnt, grp_claim_* below)
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
