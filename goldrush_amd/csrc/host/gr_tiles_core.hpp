// Tile classification core, shared by the host (g++) and the device (hipcc):
// threshold + the ten smoothing passes of calc_num_assigned_tiles
// (goldrush_path.cpp:628-889), find_longest_stretch (:195-233), eval_flanks
// (:341-527) and the read decision (:960-1040).  No allocation, no STL: the
// caller provides the scratch arrays (host vectors or LDS).
//
// Integer widths and wrap-around follow the reference expression by expression
// (uint32_t +-1 in P3/P4/P5/P9, size_t in P8 and in the flank tests).
#pragma once
#include "../../../include/grpath.h"
#include "../../../include/grpath_host.h"

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GR_HD __host__ __device__
#else
#define GR_HD
#endif

namespace gr {
namespace core {

enum : uint32_t
{
  KIND_INSERT_WHOLE = 2,
  KIND_ASSIGNED_ALL = 3,
  KIND_INSERT_TRIMMED = 4,
  KIND_ASSIGNED = 5
};

// P1 / P2 (:646-661, :667-682): tile i takes over its neighbour's ID when that
// ID is in tile i's own count>2 list; its flag becomes (count > x).
GR_HD inline void
adopt_neighbour(size_t i, size_t nb, const grp_tile_summary* tiles, const grp_id_count* lists, size_t x, uint32_t* ids, uint8_t* asg)
{
  const uint32_t want = ids[nb];
  if (ids[i] == want) {
    return;
  }
  const grp_id_count* l = lists + tiles[i].list_off;
  for (uint32_t j = 0; j < tiles[i].list_n; ++j) {
    if (l[j].id == want) {
      ids[i] = want;
      asg[i] = l[j].count > x ? 1 : 0;
    }
  }
}

// P3 / P4 (:688-710, :712-734)
GR_HD inline void
neighbour_fill(size_t i, uint32_t* ids, uint8_t* asg)
{
  if (asg[i]) {
    return;
  }
  const uint32_t cur = ids[i];
  const uint32_t pid = ids[i - 1], nid = ids[i + 1];
  const uint8_t pa = asg[i - 1], na = asg[i + 1];
  if ((cur == pid && pa) || (cur == nid && na)) {
    asg[i] = 1;
  } else if ((cur == (uint32_t)(pid + 1u) && pa) || (cur == (uint32_t)(nid + 1u) && na)) {
    asg[i] = 1;
  } else if ((cur == (uint32_t)(pid - 1u) && pa) || (cur == (uint32_t)(nid - 1u) && na)) {
    asg[i] = 1;
  } else if (pid == nid && pa && na) {
    asg[i] = pa;
    ids[i] = pid;
  }
}

// run discovery shared by P5 (:742-753, unassigned runs) and P10 (:859-869,
// assigned runs): only indices 1..n-2 are scanned, a run's start defaults to 0,
// a run still open at the end of the scan is dropped.  runs[r] = start<<32 | end.
template<bool kAssigned>
GR_HD inline size_t
collect_runs(const uint8_t* asg, size_t n, uint64_t* runs)
{
  size_t nr = 0;
  size_t start = 0;
  for (size_t i = 1; i + 1 < n; ++i) {
    const bool c = asg[i] != 0, p = asg[i - 1] != 0;
    if (c == kAssigned && p != kAssigned) {
      start = i;
    } else if (c != kAssigned && p == kAssigned) {
      runs[nr++] = ((uint64_t)start << 32) | (uint64_t)(i - 1);
    }
  }
  return nr;
}

// ids / asg: outputs, n entries; scratch: n uint64 entries.
// Returns the number of assigned tiles.
GR_HD inline size_t
smooth(size_t n, const grp_tile_summary* tiles, const grp_id_count* lists, size_t x, uint32_t* ids, uint8_t* asg, uint64_t* scratch)
{
  for (size_t i = 0; i < n; ++i) {
    ids[i] = tiles[i].top_id;
    // :628-634 — the list is non-empty iff some count is > 2, and then its largest
    // count (list[0] after the reference's sort) is the tile's top count; the
    // device-side lists are unsorted, so top_count is used
    asg[i] = (tiles[i].list_n != 0 && tiles[i].top_count > x) ? 1 : 0;
  }
  if (n >= 3) {
    for (size_t i = 1; i < n; ++i) { // P1
      adopt_neighbour(i, i - 1, tiles, lists, x, ids, asg);
    }
    for (size_t i = n - 1; i-- > 0;) { // P2: i = n-2 .. 0
      adopt_neighbour(i, i + 1, tiles, lists, x, ids, asg);
    }
    for (size_t i = 1; i + 1 < n; ++i) { // P3
      neighbour_fill(i, ids, asg);
    }
    for (size_t i = n - 2; i >= 1; --i) { // P4
      neighbour_fill(i, ids, asg);
    }
    // P5 (:739-766): interior unassigned runs whose flanking IDs differ by <= 1
    size_t nr = collect_runs<false>(asg, n, scratch);
    for (size_t r = 0; r < nr; ++r) {
      const size_t first = (size_t)(scratch[r] >> 32), second = (size_t)(scratch[r] & 0xFFFFFFFFu);
      if (first == 0 || second == n - 1) {
        continue;
      }
      const uint32_t left = ids[first - 1];
      const uint32_t right = ids[second + 1];
      if (left == right || left == (uint32_t)(right + 1u) || left == (uint32_t)(right - 1u)) {
        for (size_t i = first; i <= second; ++i) {
          asg[i] = 1;
          ids[i] = left;
        }
      }
    }
    // P6 (:771-793): isolated assigned tiles, forward then backward, 2..n-3
    for (size_t i = 2; i + 2 < n; ++i) {
      if (asg[i] && !asg[i - 1] && !asg[i + 1]) {
        asg[i] = 0;
      }
    }
    for (size_t i = n - 3; i >= 2; --i) {
      if (asg[i] && !asg[i - 1] && !asg[i + 1]) {
        asg[i] = 0;
      }
    }
    // P7 (:799-822): per ID in ascending order (std::map), between two
    // non-adjacent assigned occurrences every tile gets the ID found at the
    // earlier occurrence *at that moment* (earlier groups may have rewritten it).
    // scratch[e] = id<<32 | idx, insertion-sorted (indices arrive ascending).
    size_t ne = 0;
    for (size_t i = 0; i < n; ++i) {
      if (asg[i]) {
        const uint64_t key = ((uint64_t)ids[i] << 32) | (uint64_t)i;
        size_t j = ne++;
        while (j > 0 && scratch[j - 1] > key) {
          scratch[j] = scratch[j - 1];
          --j;
        }
        scratch[j] = key;
      }
    }
    for (size_t g = 1; g < ne; ++g) {
      if ((scratch[g] >> 32) != (scratch[g - 1] >> 32)) {
        continue;
      }
      const uint32_t a = (uint32_t)scratch[g - 1], b = (uint32_t)scratch[g];
      if (b > a + 1) {
        const uint32_t v = ids[a];
        for (size_t j = (size_t)a + 1; j <= b; ++j) {
          ids[j] = v;
        }
      }
    }
    // P8 (:827-838): end tiles, compared in size_t (no 32-bit wrap)
    {
      const size_t last = ids[n - 1], last2 = ids[n - 2], first = ids[0], second = ids[1];
      if (last == last2 || last == last2 + 1 || last == last2 - 1) {
        asg[n - 1] = 1;
      }
      if (first == second || first == second + 1 || first == second - 1) {
        asg[0] = 1;
      }
    }
    // P9 (:840-850): a tile unrelated (uint32_t +-1) to both neighbours
    for (size_t i = 1; i + 1 < n; ++i) {
      const uint32_t c = ids[i], p = ids[i - 1], q = ids[i + 1];
      if (c != q && c != (uint32_t)(q - 1u) && c != (uint32_t)(q + 1u) && c != p && c != (uint32_t)(p - 1u) && c != (uint32_t)(p + 1u)) {
        asg[i] = 0;
      }
    }
    // P10 (:856-877): assigned runs of length <= 5
    nr = collect_runs<true>(asg, n, scratch);
    for (size_t r = 0; r < nr; ++r) {
      const size_t first = (size_t)(scratch[r] >> 32), second = (size_t)(scratch[r] & 0xFFFFFFFFu);
      if (second - first + 1 <= 5) {
        for (size_t i = first; i <= second; ++i) {
          asg[i] = 0;
        }
      }
    }
  }
  size_t n_assigned = 0;
  for (size_t i = 0; i < n; ++i) {
    n_assigned += asg[i] ? 1 : 0;
  }
  return n_assigned;
}

// goldrush_path.cpp:195-233, branch for branch
GR_HD inline void
longest_stretch(const uint8_t* b, size_t n, long& out_start, long& out_end)
{
  size_t start = 0, end = 0, cur = 0, best = 0;
  long best_start = 0, best_end = 0;
  for (size_t i = 1; i + 1 < n; ++i) {
    const bool c = b[i] != 0, p = b[i - 1] != 0;
    const bool at_last = (i + 1 == n - 1);
    if (!c && p) {
      start = i;
      cur = 1;
    } else if (!c && !p && !at_last) {
      ++cur;
    } else if (c && !p) {
      end = i - 1;
      if (best < cur) {
        best = cur;
        best_start = (long)start;
        best_end = (long)end;
      }
    } else if (at_last && end < start) {
      end = i;
      ++cur;
      if (best < cur) {
        best = cur;
        best_start = (long)start;
        best_end = (long)end;
      }
    }
  }
  out_start = best_start;
  out_end = best_end;
}

struct FlankCount
{
  size_t id, n;
};

// ID histogram of tiles [lo, hi) as std::map<size_t,size_t> -> vector sorted with
// sort_by_sec.  At most 14 entries, where libstdc++'s std::sort is a plain
// (stable) insertion sort: equal counts stay in ascending-ID order.
GR_HD inline size_t
flank_histogram(const uint32_t* ids, long lo, long hi, FlankCount* out)
{
  size_t m = 0;
  for (long i = lo; i < hi; ++i) {
    const size_t id = ids[i];
    size_t j = 0;
    while (j < m && out[j].id != id) {
      ++j;
    }
    if (j == m) {
      out[m].id = id;
      out[m].n = 1;
      ++m;
    } else {
      ++out[j].n;
    }
  }
  // order: count descending, then ID ascending
  for (size_t i = 1; i < m; ++i) {
    const FlankCount v = out[i];
    size_t j = i;
    while (j > 0 && (out[j - 1].n < v.n || (out[j - 1].n == v.n && out[j - 1].id > v.id))) {
      out[j] = out[j - 1];
      --j;
    }
    out[j] = v;
  }
  return m;
}

// the two acceptance rules shared by all four flank tests (:384-403 etc.)
GR_HD inline bool
flank_ok(const FlankCount* v, size_t m, bool need_two_checked)
{
  const size_t MIN_IDS_IN_FLANK = 2;
  if (v[0].n >= MIN_IDS_IN_FLANK) {
    return true;
  }
  if (need_two_checked && m < 2) {
    return false;
  }
  return v[0].n + v[1].n > MIN_IDS_IN_FLANK + 1 && (v[0].id - v[1].id == 1 || v[1].id - v[0].id == 1);
}

GR_HD inline bool
flanks(long ls, long le, const uint32_t* ids, size_t n, size_t& trim_start, size_t& trim_end)
{
  const size_t SMALL_READ_THRESHOLD = 15;
  const long MAX_TILES_TO_CHECK = 5;
  const size_t default_start = (ls != 0) ? (size_t)(ls - 1) : (size_t)ls;
  size_t ts = default_start;
  size_t te = (size_t)(le + 1);
  bool good = false;
  FlankCount hist[16];

  if (n < SMALL_READ_THRESHOLD) {
    // :364-447 — whole flanks on both sides, both must pass
    bool left_ok = false, right_ok = false;
    size_t m = (ls > 0) ? flank_histogram(ids, 0, ls, hist) : 0;
    if (m != 0 && flank_ok(hist, m, true)) {
      left_ok = true;
    }
    if (ts == 0) {
      left_ok = true;
    }
    m = (le + 1 < (long)n) ? flank_histogram(ids, le + 1, (long)n, hist) : 0;
    if (m != 0 && flank_ok(hist, m, true)) {
      right_ok = true;
    }
    if (te == n - 1) {
      right_ok = true;
    }
    good = left_ok && right_ok;
  } else {
    // :448-525 — up to 5 tiles on each side, either side passing is enough;
    // a stretch too close to an end extends the trim to that end
    if (ls - MAX_TILES_TO_CHECK >= 1) {
      size_t m = flank_histogram(ids, ls - MAX_TILES_TO_CHECK, ls, hist);
      if (flank_ok(hist, m, false)) {
        good = true; // ts keeps its default
      }
    } else {
      good = true;
      ts = 0;
    }
    if (le + MAX_TILES_TO_CHECK < (long)n - 1) {
      size_t m = flank_histogram(ids, le + 1, le + MAX_TILES_TO_CHECK + 1, hist);
      if (flank_ok(hist, m, false)) {
        good = true; // te keeps its default
      }
    } else {
      good = true;
      te = (size_t)((long)n - 1);
    }
  }
  trim_start = ts;
  trim_end = te;
  return good;
}

// full decision of one read; ids / asg / scratch hold n entries each
GR_HD inline void
decide(size_t threshold, size_t unassigned_min, size_t assigned_max, size_t n, const grp_tile_summary* tiles, const grp_id_count* lists, uint32_t* ids, uint8_t* asg, uint64_t* scratch, gr_read_decision& out)
{
  out.kind = 0;
  out.num_tiles = (uint32_t)n;
  out.num_assigned = 0;
  out.trim_start = 0;
  out.trim_end = 0;
  out.hits = 0;
  out.misses = 0;
  out.pad = 0;
  for (size_t i = 0; i < n; ++i) {
    out.hits += tiles[i].hits;
    out.misses += tiles[i].misses;
  }
  const size_t na = smooth(n, tiles, lists, threshold, ids, asg, scratch);
  out.num_assigned = (uint32_t)na;
  const size_t nu = n - na;
  if (nu >= unassigned_min && na <= assigned_max) { // :967-971
    out.kind = KIND_INSERT_WHOLE;
    return;
  }
  if (na == n) { // :1013
    out.kind = KIND_ASSIGNED_ALL;
    return;
  }
  long ls = 0, le = 0;
  longest_stretch(asg, n, ls, le);
  size_t ts = 0, te = 0;
  if (flanks(ls, le, ids, n, ts, te)) {
    out.kind = KIND_INSERT_TRIMMED;
    out.trim_start = (uint32_t)ts;
    out.trim_end = (uint32_t)te;
  } else {
    out.kind = KIND_ASSIGNED;
  }
}

} // namespace core
} // namespace gr
