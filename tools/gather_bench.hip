// Micro-benchmark: random-gather ceilings of one MI355X for the access shapes
// of the miBF probe (calibration for the roofline discussion in DESIGN.md).
//   mode 0: independent random 16-B loads          (one sector per load)
//   mode 1: independent random 4-B loads
//   mode 2: two DEPENDENT random loads per item: 16 B from table A (small),
//           then 4/8 B from table B (large) at an index derived from the first
//   mode 3: like 2 but UNR items in flight per lane
//   modes 4-10: bucket-shaped accesses (10 = what k_query does: a quad reads one 64-B line)
// Round 4 (locality / cache-residency questions, DESIGN 4 "the ceiling of the steady state"):
//   mode 11: quad 64-B lines drawn PAGE-BINNED in address order: the grid sweeps table A once,
//            item n (grid order) falls at a random line of region n / L, regions of R bytes
//            (env GB_REGION bytes, GB_DENSITY lines drawn per line of the region, default 0.6)
//   mode 17: quad 64-B lines, the 16 lines of one wave step inside ONE random region of R bytes
//   mode 18: like 17 for the 64 lines of a workgroup step
//   mode 12: random atomicOr (u32) on table B        mode 13: random atomicCAS (u64) on table B
//   mode 14: random 1-byte stores on table B         mode 15: 8 lanes read a 128-B bucket
//   mode 16: quad reads a 64-B line of A, one lane writes 4 B of it back (a dirty random line)
//   mode 20: mode 10's random lines in the loop shape of modes 11 / 17 / 18 (their reference)
//   mode 19: random u32 load from table B, one per lane, keyed like a Bloom look-up (same as 1, kept for the tables)
// usage: gather_bench <tableA_MiB> <tableB_MiB> <items_per_lane> <mode> [unroll] [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }

template<int UNR>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ A, uint64_t nA, const uint2* __restrict__ B, uint64_t nB, int items, int mode, uint64_t* __restrict__ out, uint64_t region_lines, uint64_t lines_per_region)
{
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc = 0;
  for (int it = 0; it < items; it += UNR) {
    uint64_t ia[UNR];
    uint4 va[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      ia[u] = mix64(gid * 1315423911ULL + (uint64_t)(it + u) * 0x9E3779B97F4A7C15ULL);
    }
    if (mode == 0) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) va[u] = A[ia[u] % nA];
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += va[u].x + va[u].w;
    } else if (mode == 1) {
      uint32_t v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = reinterpret_cast<const uint32_t*>(B)[ia[u] % (nB * 2)];
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    } else if (mode == 8) { // 8-B header, then a DEPENDENT 4-B slot of the same 64-B bucket
      const uint64_t nb = nA / 4;
      uint2 h2[UNR];
      uint32_t v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) h2[u] = reinterpret_cast<const uint2*>(A)[(ia[u] % nb) * 8];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = reinterpret_cast<const uint32_t*>(A)[(ia[u] % nb) * 16 + 2 + ((h2[u].y ^ ia[u]) % 14)];
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += h2[u].x + v[u];
    } else if (mode == 20) { // mode 10's access (a quad reads one random 64-B line) in the loop shape of modes 11 / 17 / 18
      const uint64_t nbmask = nA / 4 - 1;
      const int lane = threadIdx.x & 63;
      const int sub = lane & 3;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const uint64_t idx = mix64((gid / 4) * 77ULL + (uint64_t)(it + u) * 0xD1B54A32D192ED03ULL) & nbmask;
        const uint4 t = A[idx * 4 + sub];
        const uint32_t got = __shfl(t.w, (lane & ~3) + (int)(idx & 3), 64);
        acc += got + t.x;
      }
    } else if (mode == 9 || mode == 10) {
      // cooperative: G lanes read one whole 64-B bucket with one coalesced access;
      // every lane still owns `items` probes, processed G at a time via shuffles
      const int G = (mode == 9) ? 8 : 4;
      const uint64_t nb = nA / 4;
      const int lane = threadIdx.x & 63;
      const int sub = lane % G;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        for (int g = 0; g < G; ++g) {
          // the probe of lane (group base + g)
          const uint64_t idx = __shfl(ia[u], (lane / G) * G + g, 64) % nb;
          if (mode == 9) {
            uint2 t = reinterpret_cast<const uint2*>(A)[idx * 8 + sub];
            const uint32_t want = (uint32_t)(idx >> 3) % 8u;
            uint32_t got = __shfl(t.y, (lane / G) * G + want, 64);
            if (sub == g) acc += got + t.x;
          } else {
            uint4 t = A[idx * 4 + sub];
            const uint32_t want = (uint32_t)(idx >> 3) % 4u;
            uint32_t got = __shfl(t.w, (lane / G) * G + want, 64);
            if (sub == g) acc += got + t.x;
          }
        }
      }
    } else if (mode == 11 || mode == 17 || mode == 18) {
      // (table and region sizes are powers of two: masks and one reciprocal multiply, no 64-bit division —
      // a first form spent its time in three `%` per item and measured the ALU, not the memory system)
      const uint64_t nb = nA / 4; // 64-B lines of table A
      const int lane = threadIdx.x & 63;
      const int sub = lane & 3;
      const uint32_t nquads = gridDim.x * blockDim.x / 4;
      const uint32_t inv = (uint32_t)(0x100000000ull / lines_per_region); // region of item n ~ (n * inv) >> 32
      const uint64_t rmask = region_lines - 1, nbmask = nb - 1, nregmask = nb / region_lines - 1;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        uint64_t idx;
        if (mode == 11) {
          const uint32_t n = (uint32_t)(it + u) * nquads + (uint32_t)(gid / 4); // grid order: the sweep
          const uint64_t region = ((uint64_t)n * inv) >> 32;
          idx = (region * region_lines + (mix64(n) & rmask)) & nbmask;
        } else {
          // the lines of one wave (17) / workgroup (18) step share a random region
          const uint64_t scope = (mode == 17) ? gid / 64 : gid / 256;
          const uint64_t region = mix64(scope * 0x9E3779B97F4A7C15ULL + (uint64_t)(it + u)) & nregmask;
          idx = region * region_lines + (mix64((gid / 4) * 77ULL + (uint64_t)(it + u) * 0xD1B54A32D192ED03ULL) & rmask);
        }
        const uint4 t = A[idx * 4 + sub];
        const uint32_t got = __shfl(t.w, (lane & ~3) + (int)(idx & 3), 64);
        acc += got + t.x;
      }
    } else if (mode == 12) {
      uint32_t* T = const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(B));
#pragma unroll
      for (int u = 0; u < UNR; ++u) atomicOr(&T[ia[u] % (nB * 2)], 1u << (ia[u] >> 59));
    } else if (mode == 13) {
      unsigned long long* T = const_cast<unsigned long long*>(reinterpret_cast<const unsigned long long*>(B));
      unsigned long long v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = atomicCAS(&T[ia[u] % nB], 0x0202020202020202ULL, ia[u] | 1ull);
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    } else if (mode == 14) {
      uint8_t* T = const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(B));
#pragma unroll
      for (int u = 0; u < UNR; ++u) T[ia[u] % (nB * 8)] = 1;
    } else if (mode == 15) {
      const uint64_t nb = nA / 8; // 128-B buckets
      const int lane = threadIdx.x & 63;
      const int sub = lane & 7;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        for (int g = 0; g < 8; ++g) {
          const uint64_t idx = __shfl(ia[u], (lane & ~7) + g, 64) % nb;
          const uint4 t = A[idx * 8 + sub];
          const uint32_t got = __shfl(t.w, (lane & ~7) + (int)(idx & 7), 64);
          if (sub == g) acc += got + t.x;
        }
      }
    } else if (mode == 16) {
      const uint64_t nb = nA / 4;
      const int lane = threadIdx.x & 63;
      const int sub = lane & 3;
      uint4* W = const_cast<uint4*>(A);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        for (int g = 0; g < 4; ++g) {
          const uint64_t idx = __shfl(ia[u], (lane & ~3) + g, 64) % nb;
          const uint4 t = A[idx * 4 + sub];
          if (sub == g) {
            acc += t.x;
            reinterpret_cast<uint32_t*>(W)[idx * 16 + 3 + (ia[u] >> 50) % 13] = (uint32_t)ia[u] | 1u;
          }
        }
      }
    } else if (mode == 19) {
      uint32_t v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = reinterpret_cast<const uint32_t*>(B)[(ia[u] >> 5) % (nB * 2)];
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += (v[u] >> (ia[u] & 31)) & 1u;
    } else if (mode == 4 || mode == 5 || mode == 6 || mode == 7) {
      // bucket-shaped accesses on table A viewed as 64-byte buckets
      const uint64_t nb = nA / 4;
      uint32_t v[UNR];
      if (mode == 4) { // 16-B header, then a DEPENDENT 4-B slot of the same 64-B bucket
#pragma unroll
        for (int u = 0; u < UNR; ++u) va[u] = A[(ia[u] % nb) * 4];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = reinterpret_cast<const uint32_t*>(A)[(ia[u] % nb) * 16 + 3 + ((va[u].y ^ ia[u]) % 13)];
      } else if (mode == 5) { // header and slot issued together (independent)
#pragma unroll
        for (int u = 0; u < UNR; ++u) va[u] = A[(ia[u] % nb) * 4];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = reinterpret_cast<const uint32_t*>(A)[(ia[u] % nb) * 16 + 3 + (ia[u] >> 40) % 13];
      } else if (mode == 6) { // one 8-B load per bucket
#pragma unroll
        for (int u = 0; u < UNR; ++u) { uint2 t = reinterpret_cast<const uint2*>(A)[(ia[u] % nb) * 8]; va[u] = make_uint4(t.x, t.y, 0, 0); v[u] = 0; }
      } else { // one 16-B load per bucket
#pragma unroll
        for (int u = 0; u < UNR; ++u) { va[u] = A[(ia[u] % nb) * 4]; v[u] = 0; }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += va[u].x + va[u].w + v[u];
    } else {
#pragma unroll
      for (int u = 0; u < UNR; ++u) va[u] = A[ia[u] % nA];
      uint32_t v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        uint64_t ib = (ia[u] ^ ((uint64_t)va[u].x << 7) ^ va[u].y) % nB; // depends on the first load
        v[u] = B[ib].x;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    }
  }
  if (acc == 0x123456789ULL) out[0] = acc;
}

int main(int argc, char** argv)
{
  uint64_t aMiB = argc > 1 ? atoll(argv[1]) : 512, bMiB = argc > 2 ? atoll(argv[2]) : 20480;
  int items = argc > 3 ? atoi(argv[3]) : 64, mode = argc > 4 ? atoi(argv[4]) : 2, unr = argc > 5 ? atoi(argv[5]) : 4;
  int wgs = argc > 6 ? atoi(argv[6]) : 256 * 8 * 4;
  uint64_t nA = aMiB * (1ull << 20) / 16, nB = bMiB * (1ull << 20) / 8;
  uint4* A; uint2* B; uint64_t* out;
  CK(hipMalloc(&A, nA * 16)); CK(hipMalloc(&B, nB * 8)); CK(hipMalloc(&out, 8));
  CK(hipMemset(A, 1, nA * 16)); CK(hipMemset(B, 2, nB * 8));
  const uint64_t region_bytes = getenv("GB_REGION") ? strtoull(getenv("GB_REGION"), nullptr, 10) : 2048;
  const double density = getenv("GB_DENSITY") ? atof(getenv("GB_DENSITY")) : 0.6;
  const uint64_t region_lines = region_bytes / 64 ? region_bytes / 64 : 1;
  uint64_t lines_per_region = (uint64_t)(region_lines * density + 0.5);
  if (lines_per_region == 0) lines_per_region = 1;
  if (mode == 11) { // one sweep of the table: items per lane from the table size
    const uint64_t nquads = (uint64_t)wgs * 256 / 4;
    const uint64_t total = (nA / 4) / region_lines * lines_per_region;
    items = (int)(total / nquads / unr * unr);
    if (items < unr) items = unr;
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    if (unr == 1) k_gather<1><<<wgs, 256>>>(A, nA, B, nB, items, mode, out, region_lines, lines_per_region);
    else if (unr == 2) k_gather<2><<<wgs, 256>>>(A, nA, B, nB, items, mode, out, region_lines, lines_per_region);
    else if (unr == 4) k_gather<4><<<wgs, 256>>>(A, nA, B, nB, items, mode, out, region_lines, lines_per_region);
    else if (unr == 8) k_gather<8><<<wgs, 256>>>(A, nA, B, nB, items, mode, out, region_lines, lines_per_region);
    else k_gather<16><<<wgs, 256>>>(A, nA, B, nB, items, mode, out, region_lines, lines_per_region);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double n = (double)wgs * 256 * items;
    if (mode == 11 || mode == 17 || mode == 18 || mode == 20) n /= 4; // one line per QUAD and item here
    if (rep == 2 && (mode == 11 || mode == 17 || mode == 18)) printf("region=%lluB density=%.2f items=%d ", (unsigned long long)region_bytes, density, items);
    if (rep == 2) printf("mode %d A=%lluMiB B=%lluMiB unr=%d wgs=%d: %.2f ms, %.2f G items/s, %.2f G sector-loads/s -> %.2f TB/s at 64 B/sector\n", mode, (unsigned long long)aMiB, (unsigned long long)bMiB, unr, wgs, ms, n / ms / 1e6, n * ((mode == 2 || mode == 3) ? 2 : 1) / ms / 1e6, n * ((mode == 2 || mode == 3) ? 2 : 1) * 64 / ms / 1e9);
  }
  return 0;
}
