// All-gather of small host buffers between the ranks of ONE node through a file in
// /dev/shm (include/grpath_host.h, gr_shm_allgather_*): what the ranks of a multi-GPU
// classification exchange are 32-byte decision records, a few KB per call, already in host
// memory — a GPU collective would need free compute units beside the persistent query launch
// and two PCIe copies per call (RCCL is used where bulk data moves: the bit vectors).
// Every rank writes its block into the round's buffer, then publishes the round number in
// its own cache line; readers spin on the round numbers.  Two buffers alternate, so a rank
// one round ahead never overwrites what a slower rank still reads.
#include "../../../include/grpath_host.h"

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr uint64_t kSlot = 1u << 20;     // bytes per rank and buffer
constexpr uint64_t kHeader = 64 * 1024;  // one 64-byte line per rank: [0] round, [1] attached, [2] detached
constexpr uint64_t kMagic = 0x47525053484d3031ull; // "GRPSHM01"

struct Shm
{
  uint32_t world = 0, rank = 0;
  uint64_t round = 0;
  std::string path;
  uint8_t* base = nullptr;
  size_t size = 0;
  std::atomic<uint64_t>* word(uint32_t r, uint32_t k) const { return reinterpret_cast<std::atomic<uint64_t>*>(base + 64 + (size_t)r * 64 + k * 8); }
  std::atomic<uint64_t>* magic() const { return reinterpret_cast<std::atomic<uint64_t>*>(base); }
  uint8_t* data(uint32_t buf, uint32_t r) const { return base + kHeader + ((size_t)buf * world + r) * kSlot; }
};

} // namespace

extern "C" {

void*
gr_shm_allgather_open(uint32_t world, uint32_t rank, const char* key, double timeout_s)
{
  if (world == 0 || rank >= world || !key || (uint64_t)world * 64 + 64 > kHeader) {
    return nullptr;
  }
  Shm* s = new Shm();
  s->world = world;
  s->rank = rank;
  s->path = std::string("/dev/shm/grp_") + key;
  s->size = kHeader + 2 * (size_t)world * kSlot;
  const auto t0 = std::chrono::steady_clock::now();
  auto expired = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
  int fd = -1;
  if (rank == 0) {
    (void)unlink(s->path.c_str()); // a stale file of a crashed run
    fd = open(s->path.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)s->size) != 0) {
      if (fd >= 0) {
        close(fd);
      }
      delete s;
      return nullptr;
    }
  } else {
    for (;;) { // rank 0 creates and sizes the file
      fd = open(s->path.c_str(), O_RDWR);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == s->size) {
        break;
      }
      if (fd >= 0) {
        close(fd);
        fd = -1;
      }
      if (expired()) {
        delete s;
        return nullptr;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
  }
  void* p = mmap(nullptr, s->size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete s;
    return nullptr;
  }
  s->base = static_cast<uint8_t*>(p);
  if (rank == 0) {
    s->magic()->store(kMagic, std::memory_order_release);
  } else {
    while (s->magic()->load(std::memory_order_acquire) != kMagic) { // not a stale file: rank 0 of THIS run has set it up
      if (expired()) {
        munmap(s->base, s->size);
        delete s;
        return nullptr;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
  }
  s->word(rank, 1)->store(1, std::memory_order_release);
  for (uint32_t r = 0; r < world; ++r) { // everybody is attached before the first round
    while (s->word(r, 1)->load(std::memory_order_acquire) != 1) {
      if (expired()) {
        munmap(s->base, s->size);
        delete s;
        return nullptr;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
  }
  return s;
}

// signature of gr_allgather_fn (user = the handle): rank r's block lands at recv + r * bytes
int
gr_shm_allgather(void* handle, const void* send, uint64_t bytes, void* recv)
{
  Shm* s = static_cast<Shm*>(handle);
  if (!s || bytes > kSlot) {
    return -1;
  }
  s->round += 1;
  const uint32_t b = (uint32_t)(s->round & 1);
  memcpy(s->data(b, s->rank), send, bytes);
  s->word(s->rank, 0)->store(s->round, std::memory_order_release);
  for (uint32_t r = 0; r < s->world; ++r) {
    uint64_t spins = 0;
    while (s->word(r, 0)->load(std::memory_order_acquire) < s->round) {
      if (++spins > 200000) {
        std::this_thread::yield(); // a rank that is far behind: do not burn its core
        if (spins > 2000000000ull) {
          return -2;
        }
      } else {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
    }
    memcpy(static_cast<uint8_t*>(recv) + (size_t)r * bytes, s->data(b, r), bytes);
  }
  return 0;
}

void
gr_shm_allgather_close(void* handle)
{
  Shm* s = static_cast<Shm*>(handle);
  if (!s) {
    return;
  }
  s->word(s->rank, 2)->store(1, std::memory_order_release);
  if (s->rank == 0) { // the last one out removes the file: wait (bounded) for the others
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t r = 0; r < s->world; ++r) {
      while (s->word(r, 2)->load(std::memory_order_acquire) != 1 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 10.0) {
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
    }
    (void)unlink(s->path.c_str());
  }
  munmap(s->base, s->size);
  delete s;
}

} // extern "C"
