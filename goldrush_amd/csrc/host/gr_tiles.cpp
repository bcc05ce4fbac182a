// See gr_tiles.hpp.  Thin host wrappers around the shared host/device core
// (gr_tiles_core.hpp), which restates goldrush_path.cpp:195-233
// (find_longest_stretch), :341-527 (eval_flanks), :628-889 (threshold + passes
// P1..P10 of calc_num_assigned_tiles) and :960-1040 (decision).
#include "gr_tiles.hpp"

#include "gr_tiles_core.hpp"

#include <cstring>

#include <algorithm>

namespace gr {

static void
reserve(TileWorkspace& ws, size_t n)
{
  if (ws.ids.size() < n + 1) {
    ws.ids.resize(n + 1);
    ws.asg.resize(n + 1);
    ws.scratch.resize(std::max(n + 1, core::GR_MIN_SCRATCH));
  }
}

// log_tile_states (goldrush_path.cpp:109-124): IDs, then flags, tab-separated
static void
print_tile_states(void* out, const uint32_t* ids, const uint8_t* flags, size_t n)
{
  FILE* f = static_cast<FILE*>(out);
  for (size_t i = 0; i < n; ++i) {
    fprintf(f, "%u\t", ids[i]);
  }
  fprintf(f, "\n");
  for (size_t i = 0; i < n; ++i) {
    fprintf(f, "%u\t", (unsigned)flags[i]);
  }
  fprintf(f, "\n");
}

size_t
smooth_tiles(size_t n, const grp_tile_summary* tiles, const grp_id_count* lists, size_t x, TileWorkspace& ws, FILE* dbg)
{
  reserve(ws, n);
  core::DebugPtrState s;
  s.out = dbg;
  s.print = dbg ? print_tile_states : nullptr;
  s.tiles = tiles;
  s.lists = lists;
  s.ids = ws.ids.data();
  s.flags = ws.asg.data();
  s.scratch = ws.scratch.data();
  return core::smooth(n, x, s);
}

void
find_longest_stretch(const std::vector<uint8_t>& asg, size_t n, long& start, long& end)
{
  core::PtrState s;
  s.flags = const_cast<uint8_t*>(asg.data());
  core::longest_stretch(n, s, start, end);
}

bool
eval_flanks(long ls, long le, const uint32_t* ids, size_t n, size_t& trim_start, size_t& trim_end)
{
  uint64_t hist[core::GR_MIN_SCRATCH];
  core::PtrState s;
  s.ids = const_cast<uint32_t*>(ids);
  s.scratch = hist;
  return core::flanks(ls, le, n, s, trim_start, trim_end);
}

void
decide_read(const DecideParams& p, size_t n, const grp_tile_summary* tiles, const grp_id_count* lists, TileWorkspace& ws, ReadDecision& out, FILE* dbg)
{
  reserve(ws, n);
  gr_read_decision d;
  if (dbg) {
    core::DebugPtrState s;
    s.tiles = tiles;
    s.lists = lists;
    s.ids = ws.ids.data();
    s.flags = ws.asg.data();
    s.scratch = ws.scratch.data();
    s.out = dbg;
    s.print = print_tile_states;
    core::decide(p.threshold, p.unassigned_min, p.assigned_max, n, s, d);
  } else {
    core::decide(p.threshold, p.unassigned_min, p.assigned_max, n, tiles, lists, ws.ids.data(), ws.asg.data(), ws.scratch.data(), d);
  }
  static_assert(sizeof(ReadDecision) == sizeof(gr_read_decision), "decision layout");
  std::memcpy(&out, &d, sizeof(d));
}

} // namespace gr
