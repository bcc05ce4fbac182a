// Synthetic ONT-like read generator on the GPU (include/grpath_synth.h).
// Measurement support only: lets bench.py build BASELINE-sized inputs
// (1 M reads x 25 kb = 25 Gbases) directly in HBM in about a second.
#include "../../include/grpath_synth.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

namespace {

thread_local std::string g_err;

__host__ __device__ inline uint64_t
mix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

__host__ __device__ inline uint32_t
unique_base(uint64_t seed, uint64_t i)
{
  return (uint32_t)(mix64(seed ^ (i * 0xD1342543DE82EF95ULL)) >> 62);
}

constexpr uint64_t REP_SLOT = 6144; // bases per slot of a repeat-rich genome (grpath_synth.h)

// base i of the genome: unique sequence, or — inside the unit of a slot that is a repeat copy — the family's
// consensus with this copy's substitutions
__host__ __device__ inline uint32_t
genome_base(const grp_synth_params& p, uint64_t i)
{
  if (!(p.repeat_frac > 0.0f)) {
    return unique_base(p.genome_seed, i);
  }
  const uint64_t slot = i / REP_SLOT, off = i - slot * REP_SLOT;
  const uint64_t h = mix64(p.genome_seed ^ 0x5EEDC0DE5EEDC0DEULL ^ (slot * 0xC2B2AE3D27D4EB4FULL));
  const float p_slot = p.repeat_frac * 1.5f < 1.0f ? p.repeat_frac * 1.5f : 1.0f; // mean unit 4096 of 6144 bases
  if ((float)(h & 0xFFFFFFu) >= p_slot * 16777216.0f) {
    return unique_base(p.genome_seed, i);
  }
  const uint32_t tier = (uint32_t)((h >> 24) % 3u);
  const uint64_t copies = tier == 0 ? 10000u : tier == 1 ? 1000u : 30u;
  const uint64_t tier_slots = (uint64_t)((double)(p.genome_len / REP_SLOT) * p_slot / 3.0);
  const uint64_t n_fam = tier_slots / copies > 0 ? tier_slots / copies : 1;
  const uint64_t fam = ((uint64_t)tier << 48) | ((h >> 28) % n_fam);
  const uint64_t fh = mix64(p.genome_seed ^ 0xFA111E5FA111E5ULL ^ (fam * 0x9E3779B97F4A7C15ULL));
  const uint64_t unit = 2048 + fh % 4097; // 2 .. 6 kb
  if (off >= unit) {
    return unique_base(p.genome_seed, i);
  }
  uint32_t b = (uint32_t)(mix64(fh ^ (off * 0xD6E8FEB86659FD93ULL)) >> 62); // the family's consensus
  const uint32_t div = 1u + (uint32_t)((fh >> 13) % 5u);                       // 1 .. 5 % of the positions differ in a copy
  const uint64_t m = mix64(p.genome_seed ^ 0xD17E26E7CEULL ^ (i * 0xA24BAED4963EE407ULL));
  if ((uint32_t)(m % 100u) < div) {
    b = (b + 1u + (uint32_t)((m >> 32) % 3u)) & 3u;
  }
  return b;
}

constexpr int T = 256;

__global__ void __launch_bounds__(T)
k_synth(grp_synth_params p,
        const uint64_t* __restrict__ start,
        const uint32_t* __restrict__ len,
        const uint8_t* __restrict__ strand,
        const uint64_t* __restrict__ word_off,
        uint32_t* __restrict__ out,
        uint32_t th_sub,
        uint32_t th_ins,
        uint32_t th_del,
        uint32_t r0)
{
  __shared__ uint8_t sBuf[16 + 2 * T + 16];
  __shared__ uint32_t sScan[T];
  const uint32_t r = r0 + blockIdx.x;
  const uint32_t L = len[r];
  const uint64_t s0 = start[r] % p.genome_len;
  const bool rev = strand[r] != 0;
  uint32_t* dst = out + word_off[r];
  const uint64_t rseed = mix64(p.error_seed ^ ((uint64_t)r * 0xA24BAED4963EE407ULL));

  uint32_t flushed = 0; // bases already written (multiple of 16)
  uint32_t carry = 0;   // codes waiting in sBuf[0..carry)
  for (uint64_t c0 = 0; flushed < L; c0 += T) {
    const uint64_t i = c0 + threadIdx.x;
    uint64_t coord = rev ? (s0 + p.genome_len - (i % p.genome_len)) % p.genome_len : (s0 + i) % p.genome_len;
    uint32_t b = genome_base(p, coord);
    if (rev) {
      b = 3u - b;
    }
    const uint64_t e = mix64(rseed ^ (i * 0x9FB21C651E98DF25ULL));
    const uint32_t u_del = (uint32_t)(e & 0xFFFFF);
    const uint32_t u_sub = (uint32_t)((e >> 20) & 0xFFFFF);
    const uint32_t u_ins = (uint32_t)((e >> 40) & 0xFFFFF);
    uint32_t n_emit = 0;
    uint32_t c1 = 0, c2 = 0;
    if (u_del >= th_del) {
      n_emit = 1;
      c1 = (u_sub < th_sub) ? ((b + 1u + (uint32_t)((e >> 60) % 3u)) & 3u) : b;
      if (u_ins < th_ins) {
        n_emit = 2;
        c2 = (uint32_t)((e >> 62) & 3u);
      }
    }
    // exclusive scan of n_emit over the workgroup
    sScan[threadIdx.x] = n_emit;
    __syncthreads();
    for (uint32_t o = 1; o < T; o <<= 1) {
      uint32_t t = (threadIdx.x >= o) ? sScan[threadIdx.x - o] : 0;
      __syncthreads();
      sScan[threadIdx.x] += t;
      __syncthreads();
    }
    const uint32_t off = carry + sScan[threadIdx.x] - n_emit;
    const uint32_t total = sScan[T - 1];
    if (n_emit >= 1) {
      sBuf[off] = (uint8_t)c1;
    }
    if (n_emit == 2) {
      sBuf[off + 1] = (uint8_t)c2;
    }
    __syncthreads();
    uint32_t n = carry + total;
    const uint32_t want = L - flushed;
    const bool last = n >= want;
    if (last) {
      n = want;
    }
    const uint32_t nw = last ? (n + 15u) / 16u : n / 16u;
    for (uint32_t w = threadIdx.x; w < nw; w += T) {
      uint32_t v = 0;
#pragma unroll
      for (uint32_t j = 0; j < 16; ++j) {
        uint32_t idx = w * 16u + j;
        uint32_t code = (idx < n) ? sBuf[idx] : 0u;
        v |= code << (2u * j);
      }
      dst[flushed / 16u + w] = v;
    }
    const uint32_t rem = last ? 0u : n - nw * 16u;
    uint8_t keep = 0;
    if (threadIdx.x < rem) {
      keep = sBuf[nw * 16u + threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x < rem) {
      sBuf[threadIdx.x] = keep;
    }
    carry = rem;
    flushed = last ? L : flushed + nw * 16u;
    __syncthreads();
  }
}

int
fail(const char* what, hipError_t e)
{
  char buf[512];
  snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  g_err = buf;
  return -3;
}

} // namespace

extern "C" {

const char*
grp_synth_last_error(void)
{
  return g_err.c_str();
}

int
grp_synth_reads(const grp_synth_params* p,
                const uint64_t* start,
                const uint32_t* len,
                const uint8_t* strand,
                const uint64_t* word_off,
                uint32_t n,
                void* d_out,
                void* stream_v)
{
  if (!p || !start || !len || !strand || !word_off || !d_out || p->genome_len == 0) {
    g_err = "grp_synth_reads: bad argument";
    return -1;
  }
  if (n == 0) {
    return 0;
  }
  hipStream_t stream = (hipStream_t)stream_v;
  uint64_t *d_start = nullptr, *d_off = nullptr;
  uint32_t* d_len = nullptr;
  uint8_t* d_strand = nullptr;
  hipError_t e;
  if ((e = hipMalloc(&d_start, (size_t)n * 8)) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMalloc(&d_off, ((size_t)n + 1) * 8)) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMalloc(&d_len, (size_t)n * 4)) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMalloc(&d_strand, (size_t)n)) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMemcpy(d_start, start, (size_t)n * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
  if ((e = hipMemcpy(d_off, word_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
  if ((e = hipMemcpy(d_len, len, (size_t)n * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
  if ((e = hipMemcpy(d_strand, strand, (size_t)n, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
  const float sc = 1048576.0f; // 2^20
  // a grid holds fewer than 2^32 work-items: slices of 2^22 reads
  for (uint32_t r0 = 0; r0 < n; r0 += (1u << 22)) {
    const uint32_t nr = n - r0 < (1u << 22) ? n - r0 : (1u << 22);
    k_synth<<<dim3(nr), dim3(T), 0, stream>>>(*p, d_start, d_len, d_strand, d_off, static_cast<uint32_t*>(d_out), (uint32_t)(p->p_sub * sc), (uint32_t)(p->p_ins * sc), (uint32_t)(p->p_del * sc), r0);
    if ((e = hipGetLastError()) != hipSuccess) return fail("k_synth launch", e);
  }
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("k_synth", e);
  (void)hipFree(d_start);
  (void)hipFree(d_off);
  (void)hipFree(d_len);
  (void)hipFree(d_strand);
  return 0;
}

void*
grp_synth_alloc(uint64_t bytes)
{
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, bytes ? bytes : 1);
  if (e != hipSuccess) {
    fail("hipMalloc", e);
    return nullptr;
  }
  return d;
}

void
grp_synth_free(void* d)
{
  (void)hipFree(d);
}

int
grp_synth_download(const void* d, uint64_t bytes, void* host)
{
  hipError_t e = hipMemcpy(host, d, bytes, hipMemcpyDeviceToHost);
  return e == hipSuccess ? 0 : fail("hipMemcpy", e);
}

} // extern "C"
