// Micro-benchmark: latencies of the memory-side primitives the commit loop is built from
// (one MI355X): dependent loads (plain / agent-scope), atomics with return on private and
// on contended lines, flag ping-pong between two workgroups, barriers of N workgroups.
// usage: latency_bench            (prints one line per measurement)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline uint32_t ldc(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void stc(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0 plain dependent loads, 1 agent-scope dependent loads, 2 atomicAdd(0) with return (dependent)
__global__ void k_chase(uint32_t* table, uint32_t n_mask, int iters, int mode, unsigned long long* out)
{
  if (threadIdx.x != 0) return;
  uint32_t idx = (blockIdx.x * 2654435761u) & n_mask;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    uint32_t v;
    if (mode == 0) v = table[idx];
    else if (mode == 1) v = ldc(table + idx);
    else v = atomicAdd(table + idx, 0u);
    idx = (v + i * 40503u) & n_mask;
  }
  const unsigned long long t1 = wall_clock64();
  out[blockIdx.x] = (t1 - t0) + (idx == 0xFFFFFFFFu);
}

// every workgroup: `iters` atomicAdd-with-return on ONE shared word
__global__ void k_hot(uint32_t* word, int iters, unsigned long long* out)
{
  if (threadIdx.x != 0) return;
  uint32_t acc = 0;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) acc += atomicAdd(word, 1u);
  const unsigned long long t1 = wall_clock64();
  out[blockIdx.x] = (t1 - t0) + (acc == 0xFFFFFFFFu);
}

// N workgroups, `iters` barriers (arrive with a no-return atomic, spin on the counter)
__global__ void k_barrier(uint32_t* counter, int iters, int sleep, unsigned long long* out)
{
  const uint32_t n = gridDim.x;
  unsigned long long t0 = 0;
  for (int i = 0; i < iters; ++i) {
    if (i == 1) t0 = wall_clock64(); // the first one absorbs the launch skew
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t target = (uint32_t)(i + 1) * n;
      while ((int32_t)(ldc(counter) - target) < 0) {
        if (sleep) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = wall_clock64() - t0;
}

// two workgroups bounce a flag (workgroup 0 waits for even, 1 for odd values)
__global__ void k_pingpong(uint32_t* flag, int iters, unsigned long long* out)
{
  if (threadIdx.x != 0) return;
  const uint32_t me = blockIdx.x;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    const uint32_t want = 2u * i + me;
    while (ldc(flag) != want) {}
    stc(flag, want + 1u);
  }
  out[blockIdx.x] = wall_clock64() - t0;
}

// burst: every lane of `wgs` x 1024-lane workgroups performs ONE random operation of the kind
// on a table of n 8-byte words; the time until the slowest workgroup is done (phase latency)
// kind 0 plain load, 1 sc1 load, 2 atomicCAS (with return), 3 atomicOr (no return) + wait,
// 4 sc1 store + wait, 5 sc1 load then DEPENDENT sc1 load
__global__ void __launch_bounds__(1024) k_burst(unsigned long long* table, unsigned long long n_mask, int kind, int lanes, unsigned long long* out, unsigned long long* sink)
{
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long x = gid * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
  const unsigned long long idx = x & n_mask;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  unsigned long long v = 0;
  if ((int)threadIdx.x < lanes) {
    if (kind == 0) v = table[idx];
    else if (kind == 1) v = __hip_atomic_load(table + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (kind == 2) v = atomicCAS(table + idx, 0ull, gid + 1);
    else if (kind == 3) atomicOr(table + idx, 1ull << (gid & 63));
    else if (kind == 4) __hip_atomic_store(table + idx, gid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else {
      v = __hip_atomic_load(table + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = __hip_atomic_load(table + ((v ^ x) & n_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  if (v == 0xDEADBEEFCAFEull) sink[0] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = wall_clock64() - t0;
}

int main()
{
  const uint32_t n = 1u << 28; // 1 GiB of uint32
  uint32_t* table; unsigned long long* out; uint32_t* word;
  CK(hipMalloc(&table, (size_t)n * 4)); CK(hipMalloc(&out, 4096 * 8)); CK(hipMalloc(&word, 4096));
  std::vector<uint32_t> h(n);
  uint64_t x = 88172645463325252ull;
  for (uint32_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)x; }
  CK(hipMemcpy(table, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  std::vector<unsigned long long> ho(4096);
  auto report = [&](const char* what, int wgs, int iters) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ho.data(), out, wgs * 8, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < wgs; ++i) s += (double)ho[i];
    printf("%-58s wgs=%4d  %8.3f us per op\n", what, wgs, s / wgs / iters / 100.0);
  };
  const int it = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int wgs : { 1, 32, 256 }) {
      k_chase<<<wgs, 64>>>(table, n - 1, it, mode, out);
      report(mode == 0 ? "dependent plain load, random over 1 GiB" : mode == 1 ? "dependent agent-scope (sc1) load, random over 1 GiB" : "dependent atomicAdd with return, random over 1 GiB", wgs, it);
    }
  }
  for (int wgs : { 1, 8, 32, 64, 256 }) {
    CK(hipMemset(word, 0, 4096));
    k_hot<<<wgs, 64>>>(word, it, out);
    report("atomicAdd with return on ONE shared word (per op and workgroup)", wgs, it);
  }
  for (int sleep = 0; sleep < 2; ++sleep) {
    for (int wgs : { 2, 8, 32, 64, 128, 256 }) {
      CK(hipMemset(word, 0, 4096));
      k_barrier<<<wgs, 64>>>(word, it + 1, sleep, out);
      report(sleep ? "barrier: no-return arrive + spin with s_sleep(1)" : "barrier: no-return arrive + spin", wgs, it);
    }
  }
  {
    unsigned long long* big; unsigned long long* sink;
    const unsigned long long n_big = 1ull << 31; // 16 GiB of 8-byte words
    CK(hipMalloc(&big, n_big * 8)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(big, 0, n_big * 8));
    const char* kinds[6] = { "plain load", "sc1 load", "atomicCAS", "atomicOr (no return)", "sc1 store", "2 dependent sc1 loads" };
    for (unsigned long long n_tab : { 1ull << 19, 1ull << 31 }) { // 4 MiB, 16 GiB
      for (int kind = 0; kind < 6; ++kind) {
        for (int wgs : { 32, 256 }) {
          for (int lanes : { 128, 1024 }) {
            double worst = 0, mean = 0;
            for (int rep = 0; rep < 5; ++rep) {
              CK(hipMemset(big, 0, (n_tab < (1ull << 24) ? n_tab : (1ull << 24)) * 8));
              k_burst<<<wgs, 1024>>>(big, n_tab - 1, kind, lanes, out, sink);
              CK(hipDeviceSynchronize());
              CK(hipMemcpy(ho.data(), out, wgs * 8, hipMemcpyDeviceToHost));
              if (rep == 0) continue;
              double w = 0, m = 0;
              for (int i = 0; i < wgs; ++i) { w = ho[i] > w ? (double)ho[i] : w; m += (double)ho[i]; }
              worst += w / 4; mean += m / wgs / 4;
            }
            printf("burst %-24s table %6s  wgs=%3d lanes=%4d (%6d ops): slowest workgroup %7.2f us, mean %7.2f us\n", kinds[kind], n_tab == (1ull << 19) ? "4 MiB" : "16 GiB", wgs, lanes, wgs * lanes, worst / 100.0, mean / 100.0);
          }
        }
      }
    }
  }
  CK(hipMemset(word, 0, 4096));
  k_pingpong<<<2, 64>>>(word, it, out);
  report("flag ping-pong between two workgroups (one way = half)", 2, it);
  return 0;
}
