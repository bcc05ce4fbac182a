// Host side of the tile classification: everything calc_num_assigned_tiles does
// after the per-tile query (threshold + the ten smoothing passes), the longest
// unassigned stretch, the flank test and the read decision.
// Reference: goldrush_path/goldrush_path.cpp:195-233, 341-527, 628-889, 960-1040.
//
// Input is what the query kernel returns per tile (grp_tile_summary + the
// count>2 lists); the functions here are pure (no global state), so a read's
// decision can be computed on any rank / thread and only the commit
// (gr_builder) is ordered.
#pragma once
#include "../../../include/grpath.h"

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <utility>
#include <vector>

namespace gr {

enum DecisionKind : uint8_t
{
  DEC_INSERT_WHOLE = 2,   // "_untrimmed"  (goldrush_path.cpp:978-1011)
  DEC_ASSIGNED_ALL = 3,   // complete assignment (:1013-1023)
  DEC_INSERT_TRIMMED = 4, // "_trimmed"    (:1038-1080)
  DEC_ASSIGNED = 5        // wood path     (:1083-1088)
};

// fixed-size, trivially copyable: this is what ranks exchange
struct ReadDecision
{
  uint32_t kind;
  uint32_t num_tiles;
  uint32_t num_assigned;
  uint32_t trim_start;
  uint32_t trim_end;
  uint32_t hits;   // sum over the read's tiles
  uint32_t misses; // sum over the read's tiles
  uint32_t pad;
};

struct DecideParams
{
  size_t threshold;      // -x
  size_t unassigned_min; // -u
  size_t assigned_max;   // -a
};

// scratch reused from read to read (no allocation in steady state)
struct TileWorkspace
{
  std::vector<uint32_t> ids;
  std::vector<uint8_t> asg;
  std::vector<uint64_t> scratch;
};

// threshold + smoothing passes; ids/asg are outputs (size num_tiles).
// Returns the number of assigned tiles.  dbg != NULL reproduces --debug's dumps.
size_t smooth_tiles(size_t num_tiles,
                    const grp_tile_summary* tiles,
                    const grp_id_count* lists,
                    size_t threshold,
                    TileWorkspace& ws,
                    FILE* dbg = nullptr);

void find_longest_stretch(const std::vector<uint8_t>& asg, size_t num_tiles, long& start, long& end);

bool eval_flanks(long longest_start, long longest_end, const uint32_t* ids, size_t num_tiles, size_t& trim_start, size_t& trim_end);

// full decision of one read from its tile summaries
void decide_read(const DecideParams& p,
                 size_t num_tiles,
                 const grp_tile_summary* tiles,
                 const grp_id_count* lists,
                 TileWorkspace& ws,
                 ReadDecision& out,
                 FILE* dbg = nullptr);

} // namespace gr
