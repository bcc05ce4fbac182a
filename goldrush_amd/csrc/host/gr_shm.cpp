// All-gather of small host buffers between the ranks of ONE node through a file in
// /dev/shm (include/grpath_host.h, gr_shm_allgather_*): what the ranks of a multi-GPU
// classification exchange are 32-byte decision records, a few KB per call, already in host
// memory — a GPU collective would need free compute units beside the persistent query launch
// and two PCIe copies per call (RCCL is used where bulk data moves: the bit vectors).
// Every rank writes its block into the round's buffer, then publishes the round number in
// its own cache line; readers spin on the round numbers.  Two buffers alternate, so a rank
// one round ahead never overwrites what a slower rank still reads.  A waiting rank notices a
// peer that has detached (its close()) or stopped answering (GRP_SHM_TIMEOUT_S, default 120 s).
#include "../../../include/grpath_host.h"

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr uint64_t kSlot = 1u << 20;     // bytes per rank and buffer
constexpr uint64_t kHeader = 64 * 1024;  // one 64-byte line per rank: [0] round, [1] token of the attached rank, [2] detached, [3] rank 0's echo of the token
constexpr uint64_t kMagic = 0x47525053484d3032ull; // "GRPSHM02"

struct Shm
{
  uint32_t world = 0, rank = 0;
  uint64_t round = 0;
  double timeout_s = 120.0; // what a rank waits for a peer inside gr_shm_allgather
  std::string path;
  uint8_t* base = nullptr;
  size_t size = 0;
  std::atomic<uint64_t>* word(uint32_t r, uint32_t k) const { return reinterpret_cast<std::atomic<uint64_t>*>(base + 64 + (size_t)r * 64 + k * 8); }
  std::atomic<uint64_t>* magic() const { return reinterpret_cast<std::atomic<uint64_t>*>(base); }
  uint8_t* data(uint32_t buf, uint32_t r) const { return base + kHeader + ((size_t)buf * world + r) * kSlot; }
};

// a value no earlier run has used: what a rank proves its presence with (a file left behind by a
// crashed run holds the tokens and echoes of THAT run — they never match)
uint64_t
fresh_token()
{
  uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  t ^= (uint64_t)getpid() << 32;
  timespec ts{};
  clock_gettime(CLOCK_REALTIME, &ts);
  t ^= (uint64_t)ts.tv_nsec * 0x9E3779B97F4A7C15ull + (uint64_t)ts.tv_sec;
  return t | 1ull; // never 0
}

} // namespace

extern "C" {

// Rank 0 builds the file under a private name and renames it into place (an atomic replacement of
// whatever a crashed run left under the key); every other rank opens the path, announces itself
// with a fresh token and waits for rank 0's echo of it — a stale file never echoes, and a rank
// that mapped one notices that the path has changed hands and opens it again.
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
  if (const char* e = getenv("GRP_SHM_TIMEOUT_S")) {
    s->timeout_s = std::max(0.05, atof(e));
  }
  const auto t0 = std::chrono::steady_clock::now();
  auto expired = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
  auto fail = [&]() -> void* {
    if (s->base) {
      munmap(s->base, s->size);
    }
    delete s;
    return nullptr;
  };
  if (rank == 0) {
    const std::string tmp = s->path + ".new." + std::to_string((long)getpid());
    (void)unlink(tmp.c_str());
    int fd = open(tmp.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)s->size) != 0) { // a new file: all zeros
      if (fd >= 0) {
        close(fd);
        (void)unlink(tmp.c_str());
      }
      return fail();
    }
    void* p = mmap(nullptr, s->size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
      (void)unlink(tmp.c_str());
      return fail();
    }
    s->base = static_cast<uint8_t*>(p);
    s->magic()->store(kMagic, std::memory_order_release);
    if (rename(tmp.c_str(), s->path.c_str()) != 0) {
      (void)unlink(tmp.c_str());
      return fail();
    }
    // everybody is attached before the first round: echo every token
    s->word(0, 1)->store(fresh_token(), std::memory_order_release);
    for (uint32_t r = 1; r < world; ++r) {
      uint64_t tok;
      while ((tok = s->word(r, 1)->load(std::memory_order_acquire)) == 0) {
        if (expired()) {
          (void)unlink(s->path.c_str());
          return fail();
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
      s->word(r, 3)->store(tok, std::memory_order_release);
    }
    return s;
  }
  const uint64_t token = fresh_token();
  for (;;) {
    if (expired()) {
      return fail();
    }
    int fd = open(s->path.c_str(), O_RDWR);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size != s->size) { // rank 0 has not put the file there yet
      if (fd >= 0) {
        close(fd);
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
      continue;
    }
    void* p = mmap(nullptr, s->size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
      return fail();
    }
    s->base = static_cast<uint8_t*>(p);
    bool ours = false, retry = false;
    if (s->magic()->load(std::memory_order_acquire) == kMagic) {
      s->word(rank, 1)->store(token, std::memory_order_release);
      while (!expired()) {
        if (s->word(rank, 3)->load(std::memory_order_acquire) == token) {
          ours = true; // rank 0 of THIS run has seen this process
          break;
        }
        struct stat now;
        if (stat(s->path.c_str(), &now) != 0 || now.st_ino != st.st_ino || now.st_dev != st.st_dev) {
          retry = true; // the file mapped here was a crashed run's: rank 0 has replaced it meanwhile
          break;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
    } else {
      retry = true;
    }
    if (ours) {
      break;
    }
    munmap(s->base, s->size);
    s->base = nullptr;
    if (!retry) {
      return fail();
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  for (uint32_t r = 1; r < world; ++r) { // every rank has been echoed before the first round
    while (s->word(r, 3)->load(std::memory_order_acquire) == 0) {
      if (expired()) {
        return fail();
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
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t r = 0; r < s->world; ++r) {
    uint64_t spins = 0;
    while (s->word(r, 0)->load(std::memory_order_acquire) < s->round) {
      if (++spins > 200000) {
        std::this_thread::yield(); // a rank that is far behind: do not burn its core
        if ((spins & 0x3FFu) == 0) {
          // a peer that has left (an error on its side: its close marks it detached) or that has
          // not answered for timeout_s: report it instead of spinning for hours
          if (s->word(r, 2)->load(std::memory_order_acquire) != 0) {
            return -3;
          }
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s->timeout_s) {
            return -2;
          }
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
