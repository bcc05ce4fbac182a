// Tile classification core, shared by the host (g++) and the device (hipcc):
// threshold + the ten smoothing passes of calc_num_assigned_tiles
// (goldrush_path.cpp:628-889), find_longest_stretch (:195-233), eval_flanks
// (:341-527) and the read decision (:960-1040).  No allocation, no STL.
//
// The algorithm is written once against a small "state" interface S (per-tile summaries,
// the working IDs / flags, a 64-bit scratch array, the count>2 lists):
//   PtrState   plain arrays — host vectors, LDS or global memory on the device
//   LaneState  (grpath_hip.hip) tile i lives in lane i of a few VGPRs and every access is
//              a v_readlane / v_writelane: the whole decision runs as wave-uniform scalar
//              code without a single LDS round trip (~20x faster than one lane walking
//              LDS arrays)
//
// Integer widths and wrap-around follow the reference expression by expression
// (uint32_t +-1 in P3/P4/P5/P9, size_t in P8 and in the flank tests).
#pragma once
#include "../../../include/grpath.h"
#include "../../../include/grpath_host.h"

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
// always_inline: the device keeps the state of a read in REGISTERS (LaneState / LaneStateN, grp_kernels.inc), which only
// works while every function below is inlined into the kernel — a call takes the state by reference, i.e. through
// scratch memory (round 4: k_decide's four-tiles-per-lane form first compiled to 304 bytes of scratch and 1.7 ms per
// decision launch)
#define GR_HD __host__ __device__ __attribute__((always_inline))
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

// the 64-bit scratch holds max(n, GR_MIN_SCRATCH) words: the run table of the smoothing
// passes, later the flank histograms (16 entries of {id, count})
constexpr size_t GR_MIN_SCRATCH = 32;

// One entry of a tile's count>2 list.  On the device the load is coherent at agent scope:
// in a streaming window (k_query<.., true>) the entry was written by another workgroup —
// possibly on another XCD, behind another L2 — during the SAME launch.
GR_HD inline grp_id_count
load_list_entry(const grp_id_count* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  grp_id_count e;
  e.id = (uint32_t)v;
  e.count = (uint32_t)(v >> 32);
  return e;
#else
  return *p;
#endif
}

// state over plain arrays; pointers that a caller does not need may stay null
struct PtrState
{
  const grp_tile_summary* tiles = nullptr;
  const grp_id_count* lists = nullptr;
  uint32_t* ids = nullptr;
  uint8_t* flags = nullptr;
  uint64_t* scratch = nullptr;

  GR_HD uint32_t top_id(size_t i) const { return tiles[i].top_id; }
  GR_HD uint32_t top_count(size_t i) const { return tiles[i].top_count; }
  GR_HD uint32_t list_n(size_t i) const { return tiles[i].list_n; }
  GR_HD uint32_t list_off(size_t i) const { return tiles[i].list_off; }
  GR_HD uint32_t hits(size_t i) const { return tiles[i].hits; }
  GR_HD uint32_t misses(size_t i) const { return tiles[i].misses; }
  GR_HD grp_id_count list_entry(size_t k) const { return load_list_entry(lists + k); }
  // entry j of tile i's count>2 list
  GR_HD grp_id_count tile_list_entry(size_t i, uint32_t j) const { return load_list_entry(lists + tiles[i].list_off + j); }
  // is `want` in tile i's count>2 list, and with which count (:646-661: the reference walks the whole list; an ID is in
  // it at most once).  The device's wave-resident states search with the whole wave (grp_kernels.inc): on a
  // repeat-rich genome a tile's list holds hundreds of IDs and the entry-by-entry walk — one dependent load each — made
  // the decisions the largest kernel of the run (round 5, bench.py --repeat-frac)
  GR_HD bool find_in_list(size_t i, uint32_t want, uint32_t& count) const
  {
    const uint32_t ln = list_n(i);
    bool found = false;
    for (uint32_t j = 0; j < ln; ++j) {
      const grp_id_count e = tile_list_entry(i, j);
      if (e.id == want) {
        count = e.count;
        found = true;
      }
    }
    return found;
  }
  GR_HD uint32_t id(size_t i) const { return ids[i]; }
  GR_HD void set_id(size_t i, uint32_t v) { ids[i] = v; }
  GR_HD uint32_t asg(size_t i) const { return flags[i]; }
  GR_HD void set_asg(size_t i, uint32_t v) { flags[i] = (uint8_t)v; }
  GR_HD uint64_t scr(size_t i) const { return scratch[i]; }
  GR_HD void set_scr(size_t i, uint64_t v) { scratch[i] = v; }
  // --debug's dump of the tile states after a pass (log_tile_states, goldrush_path.cpp:109-124):
  // nothing here; DebugPtrState (gr_tiles.cpp) prints
  GR_HD void log_pass(size_t) const {}
  // whole-state operations (one loop here; a few cross-lane instructions in LaneState)
  GR_HD void init_from_top(size_t n, size_t x)
  {
    for (size_t i = 0; i < n; ++i) {
      ids[i] = tiles[i].top_id;
      // :628-634 — the list is non-empty iff some count is > 2, and then its largest
      // count (list[0] after the reference's sort) is the tile's top count; the
      // device-side lists are unsorted, so top_count is used
      flags[i] = (tiles[i].list_n != 0 && tiles[i].top_count > x) ? 1u : 0u;
    }
  }
  GR_HD size_t count_asg(size_t n) const
  {
    size_t c = 0;
    for (size_t i = 0; i < n; ++i) {
      c += flags[i] ? 1 : 0;
    }
    return c;
  }
  // P7's work list: scratch[0 .. ne) = id << 32 | index of the assigned tiles, ascending; returns ne.
  // Insertion sort (indices arrive ascending).
  GR_HD size_t sort_assigned(size_t n)
  {
    size_t ne = 0;
    for (size_t i = 0; i < n; ++i) {
      if (flags[i]) {
        const uint64_t key = ((uint64_t)ids[i] << 32) | (uint64_t)i;
        size_t j = ne++;
        while (j > 0 && scratch[j - 1] > key) {
          scratch[j] = scratch[j - 1];
          --j;
        }
        scratch[j] = key;
      }
    }
    return ne;
  }
  GR_HD void sum_hits_misses(size_t n, uint32_t& h, uint32_t& m) const
  {
    h = 0;
    m = 0;
    for (size_t i = 0; i < n; ++i) {
      h += tiles[i].hits;
      m += tiles[i].misses;
    }
  }
  // summaries of the state, used to skip passes that cannot change anything
  GR_HD bool any_asg(size_t n) const
  {
    for (size_t i = 0; i < n; ++i) {
      if (flags[i]) {
        return true;
      }
    }
    return false;
  }
  GR_HD bool any_list(size_t n) const
  {
    for (size_t i = 0; i < n; ++i) {
      if (tiles[i].list_n) {
        return true;
      }
    }
    return false;
  }
};

// P1 / P2 (:646-661, :667-682): tile i takes over its neighbour's ID when that
// ID is in tile i's own count>2 list; its flag becomes (count > x).
template<class S>
GR_HD inline void
adopt_neighbour(size_t i, size_t nb, size_t x, S& s)
{
  const uint32_t want = s.id(nb);
  if (s.id(i) == want) {
    return;
  }
  uint32_t count = 0;
  if (s.find_in_list(i, want, count)) {
    s.set_id(i, want);
    s.set_asg(i, count > x ? 1u : 0u);
  }
}

// P3 / P4 (:688-710, :712-734)
template<class S>
GR_HD inline void
neighbour_fill(size_t i, S& s)
{
  if (s.asg(i)) {
    return;
  }
  const uint32_t cur = s.id(i);
  const uint32_t pid = s.id(i - 1), nid = s.id(i + 1);
  const uint32_t pa = s.asg(i - 1), na = s.asg(i + 1);
  if ((cur == pid && pa) || (cur == nid && na)) {
    s.set_asg(i, 1);
  } else if ((cur == (uint32_t)(pid + 1u) && pa) || (cur == (uint32_t)(nid + 1u) && na)) {
    s.set_asg(i, 1);
  } else if ((cur == (uint32_t)(pid - 1u) && pa) || (cur == (uint32_t)(nid - 1u) && na)) {
    s.set_asg(i, 1);
  } else if (pid == nid && pa && na) {
    s.set_asg(i, pa);
    s.set_id(i, pid);
  }
}

// run discovery shared by P5 (:742-753, unassigned runs) and P10 (:859-869,
// assigned runs): only indices 1..n-2 are scanned, a run's start defaults to 0,
// a run still open at the end of the scan is dropped.  scratch[r] = start<<32 | end.
template<bool kAssigned, class S>
GR_HD inline size_t
collect_runs(size_t n, S& s)
{
  size_t nr = 0;
  size_t start = 0;
  for (size_t i = 1; i + 1 < n; ++i) {
    const bool c = s.asg(i) != 0, p = s.asg(i - 1) != 0;
    if (c == kAssigned && p != kAssigned) {
      start = i;
    } else if (c != kAssigned && p == kAssigned) {
      s.set_scr(nr++, ((uint64_t)start << 32) | (uint64_t)(i - 1));
    }
  }
  return nr;
}

// Fills the working IDs / flags of the n tiles; returns the number of assigned tiles.
template<class S>
GR_HD inline size_t
smooth(size_t n, size_t x, S& s)
{
  s.init_from_top(n, x);
  if (n >= 3) {
    s.log_pass(n); // :636
    // Passes that provably leave the state alone are skipped (the decision of a read is on
    // the latency path of the device-side commit loop): P1 / P2 only act through a tile's
    // count>2 list; P3 .. P7 only act next to / between / on ASSIGNED tiles; P9 / P10 only
    // clear flags.  The start of every silver path — no hit anywhere — skips everything
    // but P8.
    const bool lists = s.any_list(n);
    if (lists) {
      for (size_t i = 1; i < n; ++i) { // P1
        adopt_neighbour(i, i - 1, x, s);
      }
    }
    s.log_pass(n); // :663
    if (lists) {
      for (size_t i = n - 1; i-- > 0;) { // P2: i = n-2 .. 0
        adopt_neighbour(i, i + 1, x, s);
      }
    }
    s.log_pass(n); // :684
    const bool assigned_1_7 = s.any_asg(n);
    if (assigned_1_7) {
      for (size_t i = 1; i + 1 < n; ++i) { // P3
        neighbour_fill(i, s);
      }
      for (size_t i = n - 2; i >= 1; --i) { // P4
        neighbour_fill(i, s);
      }
    }
    s.log_pass(n); // :736
    // P5 (:739-766): interior unassigned runs whose flanking IDs differ by <= 1
    size_t nr = assigned_1_7 ? collect_runs<false>(n, s) : 0;
    for (size_t r = 0; r < nr; ++r) {
      const uint64_t run = s.scr(r);
      const size_t first = (size_t)(run >> 32), second = (size_t)(run & 0xFFFFFFFFu);
      if (first == 0 || second == n - 1) {
        continue;
      }
      const uint32_t left = s.id(first - 1);
      const uint32_t right = s.id(second + 1);
      if (left == right || left == (uint32_t)(right + 1u) || left == (uint32_t)(right - 1u)) {
        for (size_t i = first; i <= second; ++i) {
          s.set_asg(i, 1);
          s.set_id(i, left);
        }
      }
    }
    s.log_pass(n); // :768
    // P6 (:771-793): isolated assigned tiles, forward then backward, 2..n-3
    if (assigned_1_7) {
      for (size_t i = 2; i + 2 < n; ++i) {
        if (s.asg(i) && !s.asg(i - 1) && !s.asg(i + 1)) {
          s.set_asg(i, 0);
        }
      }
      for (size_t i = n - 3; i >= 2; --i) {
        if (s.asg(i) && !s.asg(i - 1) && !s.asg(i + 1)) {
          s.set_asg(i, 0);
        }
      }
    }
    s.log_pass(n); // :795
    // P7 (:799-822): per ID in ascending order (std::map), between two
    // non-adjacent assigned occurrences every tile gets the ID found at the
    // earlier occurrence *at that moment* (earlier groups may have rewritten it).
    // scratch[e] = id<<32 | idx, sorted ascending (State::sort_assigned; the keys are distinct).
    // (the device states sort with the whole wave: the insertion sort is quadratic for a read on the other
    // strand — IDs falling along the read — and was most of a long read's decision time, round 4)
    const size_t ne = assigned_1_7 ? s.sort_assigned(n) : 0;
    for (size_t g = 1; g < ne; ++g) {
      const uint64_t e0 = s.scr(g - 1), e1 = s.scr(g);
      if ((e1 >> 32) != (e0 >> 32)) {
        continue;
      }
      const uint32_t a = (uint32_t)e0, b = (uint32_t)e1;
      if (b > a + 1) {
        const uint32_t v = s.id(a);
        for (size_t j = (size_t)a + 1; j <= b; ++j) {
          s.set_id(j, v);
        }
      }
    }
    s.log_pass(n); // :823
    // P8 (:827-838): end tiles, compared in size_t (no 32-bit wrap)
    {
      const size_t last = s.id(n - 1), last2 = s.id(n - 2), first = s.id(0), second = s.id(1);
      if (last == last2 || last == last2 + 1 || last == last2 - 1) {
        s.set_asg(n - 1, 1);
      }
      if (first == second || first == second + 1 || first == second - 1) {
        s.set_asg(0, 1);
      }
    }
    // P9 (:840-850): a tile unrelated (uint32_t +-1) to both neighbours
    const bool assigned_9_10 = s.any_asg(n);
    for (size_t i = 1; assigned_9_10 && i + 1 < n; ++i) {
      const uint32_t c = s.id(i), p = s.id(i - 1), q = s.id(i + 1);
      if (c != q && c != (uint32_t)(q - 1u) && c != (uint32_t)(q + 1u) && c != p && c != (uint32_t)(p - 1u) && c != (uint32_t)(p + 1u)) {
        s.set_asg(i, 0);
      }
    }
    s.log_pass(n); // :852
    // P10 (:856-877): assigned runs of length <= 5
    nr = assigned_9_10 ? collect_runs<true>(n, s) : 0;
    for (size_t r = 0; r < nr; ++r) {
      const uint64_t run = s.scr(r);
      const size_t first = (size_t)(run >> 32), second = (size_t)(run & 0xFFFFFFFFu);
      if (second - first + 1 <= 5) {
        for (size_t i = first; i <= second; ++i) {
          s.set_asg(i, 0);
        }
      }
    }
    s.log_pass(n); // :879
  }
  return s.count_asg(n);
}

// goldrush_path.cpp:195-233, branch for branch
template<class S>
GR_HD inline void
longest_stretch(size_t n, const S& s, long& out_start, long& out_end)
{
  size_t start = 0, end = 0, cur = 0, best = 0;
  long best_start = 0, best_end = 0;
  for (size_t i = 1; i + 1 < n; ++i) {
    const bool c = s.asg(i) != 0, p = s.asg(i - 1) != 0;
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

// ID histogram of tiles [lo, hi) as std::map<size_t,size_t> -> vector sorted with
// sort_by_sec.  At most 14 entries, where libstdc++'s std::sort is a plain
// (stable) insertion sort: equal counts stay in ascending-ID order.
// Entry e lives in scratch[2e] (id) and scratch[2e+1] (count).
template<class S>
GR_HD inline size_t
flank_histogram(long lo, long hi, S& s)
{
  size_t m = 0;
  for (long i = lo; i < hi; ++i) {
    const size_t id = s.id((size_t)i);
    size_t j = 0;
    while (j < m && s.scr(2 * j) != id) {
      ++j;
    }
    if (j == m) {
      s.set_scr(2 * m, id);
      s.set_scr(2 * m + 1, 1);
      ++m;
    } else {
      s.set_scr(2 * j + 1, s.scr(2 * j + 1) + 1);
    }
  }
  // order: count descending, then ID ascending
  for (size_t i = 1; i < m; ++i) {
    const uint64_t vid = s.scr(2 * i), vn = s.scr(2 * i + 1);
    size_t j = i;
    while (j > 0 && (s.scr(2 * j - 1) < vn || (s.scr(2 * j - 1) == vn && s.scr(2 * j - 2) > vid))) {
      s.set_scr(2 * j, s.scr(2 * j - 2));
      s.set_scr(2 * j + 1, s.scr(2 * j - 1));
      --j;
    }
    s.set_scr(2 * j, vid);
    s.set_scr(2 * j + 1, vn);
  }
  return m;
}

// the two acceptance rules shared by all four flank tests (:384-403 etc.), on the
// histogram left in the scratch by flank_histogram
template<class S>
GR_HD inline bool
flank_ok(const S& s, size_t m, bool need_two_checked)
{
  const size_t MIN_IDS_IN_FLANK = 2;
  const size_t n0 = (size_t)s.scr(1);
  if (n0 >= MIN_IDS_IN_FLANK) {
    return true;
  }
  if (need_two_checked && m < 2) {
    return false;
  }
  const size_t id0 = (size_t)s.scr(0), id1 = (size_t)s.scr(2), n1 = (size_t)s.scr(3);
  return n0 + n1 > MIN_IDS_IN_FLANK + 1 && (id0 - id1 == 1 || id1 - id0 == 1);
}

template<class S>
GR_HD inline bool
flanks(long ls, long le, size_t n, S& s, size_t& trim_start, size_t& trim_end)
{
  const size_t SMALL_READ_THRESHOLD = 15;
  const long MAX_TILES_TO_CHECK = 5;
  const size_t default_start = (ls != 0) ? (size_t)(ls - 1) : (size_t)ls;
  size_t ts = default_start;
  size_t te = (size_t)(le + 1);
  bool good = false;

  if (n < SMALL_READ_THRESHOLD) {
    // :364-447 — whole flanks on both sides, both must pass
    bool left_ok = false, right_ok = false;
    size_t m = (ls > 0) ? flank_histogram(0, ls, s) : 0;
    if (m != 0 && flank_ok(s, m, true)) {
      left_ok = true;
    }
    if (ts == 0) {
      left_ok = true;
    }
    m = (le + 1 < (long)n) ? flank_histogram(le + 1, (long)n, s) : 0;
    if (m != 0 && flank_ok(s, m, true)) {
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
      size_t m = flank_histogram(ls - MAX_TILES_TO_CHECK, ls, s);
      if (flank_ok(s, m, false)) {
        good = true; // ts keeps its default
      }
    } else {
      good = true;
      ts = 0;
    }
    if (le + MAX_TILES_TO_CHECK < (long)n - 1) {
      size_t m = flank_histogram(le + 1, le + MAX_TILES_TO_CHECK + 1, s);
      if (flank_ok(s, m, false)) {
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

// full decision of one read
template<class S>
GR_HD inline void
decide(size_t threshold, size_t unassigned_min, size_t assigned_max, size_t n, S& s, gr_read_decision& out)
{
  out.kind = 0;
  out.num_tiles = (uint32_t)n;
  out.num_assigned = 0;
  out.trim_start = 0;
  out.trim_end = 0;
  out.hits = 0;
  out.misses = 0;
  out.pad = 0;
  s.sum_hits_misses(n, out.hits, out.misses);
  const size_t na = smooth(n, threshold, s);
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
  longest_stretch(n, s, ls, le);
  size_t ts = 0, te = 0;
  if (flanks(ls, le, n, s, ts, te)) {
    out.kind = KIND_INSERT_TRIMMED;
    out.trim_start = (uint32_t)ts;
    out.trim_end = (uint32_t)te;
  } else {
    out.kind = KIND_ASSIGNED;
  }
}

// PtrState that prints --debug's dumps (host only; `out` is a FILE*)
struct DebugPtrState : PtrState
{
  void* out = nullptr;
  void (*print)(void* out, const uint32_t* ids, const uint8_t* flags, size_t n) = nullptr;
  void log_pass(size_t n) const
  {
    if (print) {
      print(out, ids, flags, n);
    }
  }
};

// pointer form (ids / flags hold n entries, scratch max(n, GR_MIN_SCRATCH))
GR_HD inline void
decide(size_t threshold, size_t unassigned_min, size_t assigned_max, size_t n, const grp_tile_summary* tiles, const grp_id_count* lists, uint32_t* ids, uint8_t* asg, uint64_t* scratch, gr_read_decision& out)
{
  PtrState s;
  s.tiles = tiles;
  s.lists = lists;
  s.ids = ids;
  s.flags = asg;
  s.scratch = scratch;
  decide(threshold, unassigned_min, assigned_max, n, s, out);
}

} // namespace core
} // namespace gr
