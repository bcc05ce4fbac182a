// The fill of several ranks of one node merged into every rank's bit vector (the fill is order-free, its
// merge a bitwise OR) — one implementation for the goldrush-path binary (gr_path.cpp) and for bench.py's N > 1 runs, so
// that what the driver's scaling run exercises is what ships (round 5; bench.py used to merge through torch.distributed).
// Reference: the whole-node OpenMP fill, goldrush_path.cpp:257-306 (every thread ORs into ONE shared vector,
// MIBFConstructSupport.hpp:134-147); here every rank has its own vector and the ranks' vectors are ORed once.
#include "../../../include/grpath_host.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

// every rank says `mine`; true if all of them said true (false also when the exchange failed)
bool
all_agree(void* shm, uint32_t world, bool mine)
{
  std::vector<uint8_t> one(64, mine ? 1 : 0), all((size_t)64 * world);
  if (gr_shm_allgather(shm, one.data(), 64, all.data()) != 0) {
    return false;
  }
  bool ok = true;
  for (uint32_t p = 0; p < world; ++p) {
    ok = ok && all[(size_t)p * 64] != 0;
  }
  return ok;
}

} // namespace

extern "C" {

// How the ranks will merge: GR_MERGE_RCCL inside the engine (one GPU per rank, grp_comm_* / grp_bv_merge_ranks),
// GR_MERGE_STAGED through host memory and /dev/shm (ranks sharing a device, engines without RCCL), GR_MERGE_NONE
// (neither: every rank fills every read).  Every decision is taken by ALL ranks alike — flags all-gathered through
// `shm` — so the ranks never diverge; for GR_MERGE_RCCL the communicator is up when this returns.  Collective.
int
gr_fill_merge_plan(const grp_engine_vt* vt, void* ctx, void* shm, uint32_t world, uint32_t rank, int device)
{
  if (!vt || !shm || world < 2 || rank >= world) {
    return GR_MERGE_NONE;
  }
  const bool staged_ok = all_agree(shm, world, vt->bv_words && vt->bv_export_words && vt->bv_or_words);
  bool rccl_ok = all_agree(shm, world, vt->comm_unique_id && vt->comm_init && vt->bv_merge_ranks && !getenv("GRP_NO_RCCL"));
  if (rccl_ok) { // one device per rank?  (RCCL refuses ranks that share a GPU)
    std::vector<int32_t> dev(16, device), devs((size_t)16 * world);
    rccl_ok = gr_shm_allgather(shm, dev.data(), 64, devs.data()) == 0;
    for (uint32_t a = 0; a < world && rccl_ok; ++a) {
      for (uint32_t b = a + 1; b < world; ++b) {
        rccl_ok = rccl_ok && devs[(size_t)a * 16] != devs[(size_t)b * 16];
      }
    }
    rccl_ok = all_agree(shm, world, rccl_ok);
  }
  if (rccl_ok) {
    std::vector<char> id(128, 0), ids((size_t)128 * world);
    const bool got = rank != 0 || vt->comm_unique_id(id.data(), id.size()) == GRP_OK;
    const bool shared = gr_shm_allgather(shm, id.data(), 128, ids.data()) == 0;
    rccl_ok = all_agree(shm, world, got && shared);
    if (rccl_ok) {
      rccl_ok = all_agree(shm, world, vt->comm_init(ctx, ids.data(), world, rank) == GRP_OK); // (rank 0's id is block 0)
    }
  }
  return rccl_ok ? GR_MERGE_RCCL : staged_ok ? GR_MERGE_STAGED : GR_MERGE_NONE;
}

// The merge itself, between the ranks' fills and grp_finalize.  0: done; 1: an engine call failed (vt->last_error
// says which); 2: the exchange between the ranks failed.  Collective.
int
gr_fill_merge_run(const grp_engine_vt* vt, void* ctx, void* shm, uint32_t world, uint32_t rank, int plan)
{
  if (plan == GR_MERGE_NONE || world < 2) {
    return 0;
  }
  if (plan == GR_MERGE_RCCL) {
    return vt->bv_merge_ranks(ctx) == GRP_OK ? 0 : 1;
  }
  uint64_t n_words = 0;
  if (vt->bv_words(ctx, &n_words) != GRP_OK) {
    return 1;
  }
  const uint64_t chunk = (1u << 20) / 4; // words per exchange: one slot of the shared-memory all-gather
  std::vector<uint32_t> mine(chunk), all((size_t)chunk * world);
  for (uint64_t w0 = 0; w0 < n_words; w0 += chunk) {
    const uint64_t n = std::min<uint64_t>(chunk, n_words - w0);
    if (vt->bv_export_words(ctx, w0, n, mine.data()) != GRP_OK) {
      return 1;
    }
    if (gr_shm_allgather(shm, mine.data(), n * 4, all.data()) != 0) {
      return 2;
    }
    for (uint32_t p = 0; p < world; ++p) {
      if (p != rank && vt->bv_or_words(ctx, w0, n, all.data() + (size_t)p * n) != GRP_OK) {
        return 1;
      }
    }
  }
  return 0;
}

// 1: at least two ranks of the node run on the same device (the plumbing runs of a one-GPU box), 0: every rank has its
// own, -1: the exchange failed.  The ranks' persistent launches must then all be resident on that device TOGETHER (their
// in-launch inserts wait grid-wide): the caller limits them to one workgroup per CU (GRP_STREAM_WGS_PER_CU).  Collective.
int
gr_ranks_share_device(void* shm, uint32_t world, int device)
{
  std::vector<int32_t> dev(16, device), devs((size_t)16 * world);
  if (!shm || gr_shm_allgather(shm, dev.data(), 64, devs.data()) != 0) {
    return -1;
  }
  for (uint32_t a = 0; a < world; ++a) {
    for (uint32_t b = a + 1; b < world; ++b) {
      if (devs[(size_t)a * 16] == devs[(size_t)b * 16]) {
        return 1;
      }
    }
  }
  return 0;
}

// 1: every rank holds `value` (the filter's population behind the merge: the replicas are the same filter), 0: they
// differ, -1: the exchange failed.  Collective.
int
gr_ranks_same_u64(void* shm, uint32_t world, uint64_t value)
{
  std::vector<uint64_t> one(8, value), all((size_t)8 * world);
  if (!shm || gr_shm_allgather(shm, one.data(), 64, all.data()) != 0) {
    return -1;
  }
  for (uint32_t p = 0; p < world; ++p) {
    if (all[(size_t)p * 8] != value) {
      return 0;
    }
  }
  return 1;
}

} // extern "C"
