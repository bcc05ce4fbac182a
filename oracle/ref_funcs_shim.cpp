// ORACLE — TEST INFRASTRUCTURE ONLY.
// C wrappers around pieces of the reference's host logic that are libstdc++-only code:
// oracle/extract_ref_funcs.py cuts them out of /root/reference at build time into the
// git-ignored oracle/_ref/extract/*.inc (whole functions: log_tile_states, sort_by_sec,
// find_longest_stretch, eval_flanks, MIBloomFilter::calcOptimalSize; the tail of
// calc_num_assigned_tiles behind its per-tile query loop and, inside that loop, the two statement
// ranges that only touch std::set / std::map locals — a frame's unique IDs tabulated into the tile's
// count table, the selection of the tile's ID and list; the hash-universe statements of main) and
// this file compiles them unchanged.  The reference's own opt.cpp is compiled next
// to it for the opt:: variables those lines read.  This file contains no reference code.
#include "opt.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <map>
#include <set>
#include <string>
#include <sys/types.h>
#include <tuple>
#include <utility>
#include <vector>

using namespace std; // goldrush_path.cpp names pair / vector unqualified

#include "_ref/extract/log_tile_states.inc"
#include "_ref/extract/sort_by_sec.inc"
#include "_ref/extract/find_longest_stretch.inc"
#include "_ref/extract/eval_flanks.inc"

struct RefMIBloomFilter
{
#include "_ref/extract/calc_optimal_size.inc"
};

// the tail of calc_num_assigned_tiles (goldrush_path.cpp: from the threshold test behind the
// per-tile loop to the function's closing brace): the parameters are the locals it uses
static size_t
ref_smooth_tail(size_t num_tiles, std::vector<uint32_t>& tiles_assigned_id_vec, std::vector<uint8_t>& tiles_assigned_bool_vec, std::vector<std::vector<std::pair<uint32_t, uint32_t>>>& tiles_assigned_all_id_vec)
{
  size_t num_assigned_tiles = 0;
#include "_ref/extract/smooth_tail.inc"

extern "C" {

void
ref_find_longest_stretch(const uint8_t* bools, size_t n, long* start, long* end)
{
  const std::vector<uint8_t> v(bools, bools + n);
  const auto r = find_longest_stretch(v);
  *start = (long)r.first;
  *end = (long)r.second;
}

int
ref_eval_flanks(long longest_start, long longest_end, const uint32_t* ids, size_t n, size_t* trim_start, size_t* trim_end)
{
  const auto r = eval_flanks((ssize_t)longest_start, (ssize_t)longest_end, std::vector<uint32_t>(ids, ids + n));
  *trim_start = std::get<1>(r);
  *trim_end = std::get<2>(r);
  return std::get<0>(r) ? 1 : 0;
}

// ids: in = the tiles' top IDs, out = after the passes; lists: tile i owns pairs [list_off[i], list_off[i+1])
// of (id, count), count descending; bools_out: assigned flags; returns the number of assigned tiles
size_t
ref_smooth_tiles(size_t n, uint32_t* ids, uint8_t* bools_out, const uint64_t* list_off, const uint32_t* list_ids, const uint32_t* list_counts, size_t threshold)
{
  opt::threshold = threshold;
  opt::debug = false;
  std::vector<uint32_t> id_vec(ids, ids + n);
  std::vector<uint8_t> bool_vec(n, 0);
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> all(n);
  for (size_t i = 0; i < n; ++i) {
    for (uint64_t j = list_off[i]; j < list_off[i + 1]; ++j) {
      all[i].emplace_back(list_ids[j], list_counts[j]);
    }
  }
  const size_t r = ref_smooth_tail(n, id_vec, bool_vec, all);
  for (size_t i = 0; i < n; ++i) {
    ids[i] = id_vec[i];
    bools_out[i] = bool_vec[i];
  }
  return r;
}

// The per-tile vote given every frame's IDs (goldrush_path.cpp: the statements of the per-tile loop
// behind getData): frame f holds ids[frame_off[f] .. frame_off[f+1]) — already stripped of the
// saturation bit, no zeros; duplicates inside a frame allowed (the std::set removes them, as in the
// reference).  Returns the tile's ID; *top_count its count; the (id, count) pairs with count > 2 in
// the reference's order (std::sort with sort_by_sec) in list_ids / list_counts.
uint32_t
ref_vote_tile(const uint32_t* ids, const uint64_t* frame_off, size_t n_frames, uint32_t* top_count, uint32_t* list_ids, uint32_t* list_counts, size_t list_cap, size_t* list_n)
{
  std::map<uint32_t, std::pair<uint32_t, uint32_t>> id_counts;
  for (size_t fr = 0; fr < n_frames; ++fr) {
    std::set<uint32_t> unique_ids(ids + frame_off[fr], ids + frame_off[fr + 1]);
#include "_ref/extract/vote_tabulate.inc"
  }
#include "_ref/extract/vote_select.inc"
  *top_count = curr_id_count;
  *list_n = id_counts_vec.size();
  for (size_t i = 0; i < id_counts_vec.size() && i < list_cap; ++i) {
    list_ids[i] = id_counts_vec[i].first;
    list_counts[i] = id_counts_vec[i].second;
  }
  return curr_id;
}

uint64_t
ref_calc_optimal_size(uint64_t entries, unsigned hash_num, double occupancy)
{
  return (uint64_t)RefMIBloomFilter::calcOptimalSize((size_t)entries, hash_num, occupancy);
}

uint64_t
ref_hash_universe(uint64_t weight, uint64_t genome_size, uint64_t hash_num)
{
  opt::weight = weight;
  opt::genome_size = genome_size;
  opt::hash_num = hash_num;
  opt::hash_universe = 0;
  {
#include "_ref/extract/hash_universe.inc"
  }
  return opt::hash_universe;
}

}
