// Device-side data layout and helpers of the gfx950 GoldRush-Path engine.
//
// HBM layout (DESIGN.md "Data layout"):
//   blocks[nblk]  uint4   rank-interleaved bit vector: .x = ones before this
//                         block relative to its superblock, .y/.z/.w = 96 data
//                         bits.  One 16-byte load answers "bit set?" and
//                         "rank?" for a probe (the reference's
//                         sdsl::bit_vector_il<512> + rank_support_il<1>,
//                         MIBloomFilter.hpp:757-759, needs a 72-byte block).
//   super[nsb]    uint64  absolute ones before each superblock of 2^24 blocks.
//   idc[pop]      uint2   .x = ID (MIBloomFilter m_data, :758),
//                         .y = insert count (MIBFConstructSupport m_counts, :338).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GRP_DEV_MAX_H 8
#define GRP_DEV_MAX_W 32
#define GRP_BLOCK_BITS 96u
#define GRP_SUPER_SHIFT 24 /* blocks per superblock = 2^24 */
#define GRP_CHUNK_BLOCKS 4096u /* rank-build chunk; divides the superblock */

struct DevSeeds
{
  uint32_t h;
  uint32_t k;    // base span
  uint32_t wmax; // max weight over seeds
  uint32_t pad;
  uint32_t weight[GRP_DEV_MAX_H];
  uint32_t span[GRP_DEV_MAX_H];
  uint32_t shift[GRP_DEV_MAX_H][GRP_DEV_MAX_W]; // 2*q for care position q
  // [seed][care index][2-bit base] = { srol(SEED[b], K-1-q), srol(SEED[comp b], q) }
  ulonglong2 tab[GRP_DEV_MAX_H][GRP_DEV_MAX_W][4];
};

struct DevFilter
{
  uint4* blocks;
  const uint64_t* super;
  uint2* idc;
  uint64_t m;     // filter bits
  uint64_t m_inv; // floor((2^64-1)/m)
  uint64_t nblk;
  uint64_t pop;
};

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

// ones among the `off` lowest of a block's 96 data bits
__device__ inline uint32_t
grp_block_rank(const uint4& b, uint32_t off)
{
  uint32_t r = 0;
  uint32_t w = off >> 5;
  uint32_t bit = off & 31u;
  uint32_t lowmask = (1u << bit) - 1u; // bit < 32
  if (w == 0) {
    r = __popc(b.y & lowmask);
  } else if (w == 1) {
    r = __popc(b.y) + __popc(b.z & lowmask);
  } else {
    r = __popc(b.y) + __popc(b.z) + __popc(b.w & lowmask);
  }
  return r;
}

__device__ inline uint32_t
grp_block_bit(const uint4& b, uint32_t off)
{
  uint32_t w = off >> 5;
  uint32_t word = (w == 0) ? b.y : ((w == 1) ? b.z : b.w);
  return (word >> (off & 31u)) & 1u;
}
