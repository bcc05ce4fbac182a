// See gr_tiles.hpp.  Behavioural restatement of goldrush_path.cpp:195-233
// (find_longest_stretch), :341-527 (eval_flanks), :628-889 (threshold + passes
// P1..P10 of calc_num_assigned_tiles) and :960-1040 (decision); integer widths
// and wrap-around follow the reference expression by expression.
#include "gr_tiles.hpp"

#include <algorithm>

namespace gr {

namespace {

void
dump_states(FILE* dbg, const std::vector<uint32_t>& ids, const std::vector<uint8_t>& asg, size_t n)
{
  // log_tile_states (goldrush_path.cpp:109-124)
  if (!dbg) {
    return;
  }
  for (size_t i = 0; i < n; ++i) {
    fprintf(dbg, "%u\t", ids[i]);
  }
  fputc('\n', dbg);
  for (size_t i = 0; i < n; ++i) {
    fprintf(dbg, "%u\t", (unsigned)asg[i]);
  }
  fputc('\n', dbg);
}

// P1 / P2 (:646-661, :667-682): tile i takes over its neighbour's ID when that
// ID is in tile i's own count>2 list; its flag becomes (count > x).
inline void
adopt_neighbour(size_t i, size_t nb, const grp_tile_summary* tiles, const grp_id_count* lists, size_t x, std::vector<uint32_t>& ids, std::vector<uint8_t>& asg)
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

// P3 / P4 (:688-710, :712-734): an unassigned tile becomes assigned when its ID
// equals, or is one off, an assigned neighbour's ID; or is bridged when both
// neighbours are assigned to the same ID.  uint32_t arithmetic (wraps).
inline void
neighbour_fill(size_t i, std::vector<uint32_t>& ids, std::vector<uint8_t>& asg)
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

// run discovery shared by P5 (:742-753, runs of unassigned tiles) and P10
// (:859-869, runs of assigned tiles): only indices 1..n-2 are scanned, a run's
// start defaults to 0 and a run still open at the end of the scan is dropped.
template<bool kAssigned>
void
collect_runs(const std::vector<uint8_t>& asg, size_t n, std::vector<std::pair<size_t, size_t>>& runs)
{
  runs.clear();
  size_t start = 0;
  for (size_t i = 1; i + 1 < n; ++i) {
    const bool c = asg[i] != 0, p = asg[i - 1] != 0;
    if (c == kAssigned && p != kAssigned) {
      start = i;
    } else if (c != kAssigned && p == kAssigned) {
      runs.emplace_back(start, i - 1);
    }
  }
}

} // namespace

size_t
smooth_tiles(size_t n, const grp_tile_summary* tiles, const grp_id_count* lists, size_t x, TileWorkspace& ws, FILE* dbg)
{
  std::vector<uint32_t>& ids = ws.ids;
  std::vector<uint8_t>& asg = ws.asg;
  ids.resize(n);
  asg.resize(n);
  for (size_t i = 0; i < n; ++i) {
    ids[i] = tiles[i].top_id;
    // :628-634 — list[0] is the largest count of the tile
    asg[i] = (tiles[i].list_n != 0 && lists[tiles[i].list_off].count > x) ? 1 : 0;
  }
  if (n >= 3) {
    dump_states(dbg, ids, asg, n);
    for (size_t i = 1; i < n; ++i) { // P1
      adopt_neighbour(i, i - 1, tiles, lists, x, ids, asg);
    }
    dump_states(dbg, ids, asg, n);
    for (size_t i = n - 1; i-- > 0;) { // P2: i = n-2 .. 0
      adopt_neighbour(i, i + 1, tiles, lists, x, ids, asg);
    }
    dump_states(dbg, ids, asg, n);
    for (size_t i = 1; i + 1 < n; ++i) { // P3
      neighbour_fill(i, ids, asg);
    }
    for (size_t i = n - 2; i >= 1; --i) { // P4
      neighbour_fill(i, ids, asg);
    }
    dump_states(dbg, ids, asg, n);

    // P5 (:739-766): interior unassigned runs whose flanking IDs differ by <= 1
    collect_runs<false>(asg, n, ws.runs);
    for (const auto& r : ws.runs) {
      if (r.first == 0 || r.second == n - 1) {
        continue;
      }
      const uint32_t left = ids[r.first - 1];
      const uint32_t right = ids[r.second + 1];
      if (left == right || left == (uint32_t)(right + 1u) || left == (uint32_t)(right - 1u)) {
        for (size_t i = r.first; i <= r.second; ++i) {
          asg[i] = 1;
          ids[i] = left;
        }
      }
    }
    dump_states(dbg, ids, asg, n);

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
    dump_states(dbg, ids, asg, n);

    // P7 (:799-822): per ID in ascending order (std::map), between two
    // non-adjacent assigned occurrences every tile gets the ID found at the
    // earlier occurrence *at that moment* (earlier groups may have rewritten it)
    ws.by_id.clear();
    for (size_t i = 0; i < n; ++i) {
      if (asg[i]) {
        ws.by_id.emplace_back(ids[i], (uint32_t)i);
      }
    }
    std::sort(ws.by_id.begin(), ws.by_id.end());
    for (size_t g = 1; g < ws.by_id.size(); ++g) {
      if (ws.by_id[g].first != ws.by_id[g - 1].first) {
        continue;
      }
      const uint32_t a = ws.by_id[g - 1].second, b = ws.by_id[g].second;
      if (b > a + 1) {
        const uint32_t v = ids[a];
        for (size_t j = (size_t)a + 1; j <= b; ++j) {
          ids[j] = v;
        }
      }
    }
    dump_states(dbg, ids, asg, n);

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
    dump_states(dbg, ids, asg, n);

    // P10 (:856-877): assigned runs of length <= 5
    collect_runs<true>(asg, n, ws.runs);
    for (const auto& r : ws.runs) {
      if (r.second - r.first + 1 <= 5) {
        for (size_t i = r.first; i <= r.second; ++i) {
          asg[i] = 0;
        }
      }
    }
    dump_states(dbg, ids, asg, n);
  }
  size_t n_assigned = 0;
  for (size_t i = 0; i < n; ++i) {
    n_assigned += asg[i] ? 1 : 0;
  }
  return n_assigned;
}

void
find_longest_stretch(const std::vector<uint8_t>& b, size_t n, long& out_start, long& out_end)
{
  // goldrush_path.cpp:195-233, branch for branch (the last interior index
  // i + 1 == n - 1 has its own rules)
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

namespace {

struct FlankCount
{
  size_t id, n;
};

// ID histogram of tiles [lo, hi) as std::map<size_t,size_t> -> vector sorted
// with sort_by_sec.  At most 14 entries, where libstdc++'s std::sort is a
// plain (stable) insertion sort: equal counts stay in ascending-ID order.
size_t
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
      out[m++] = { id, 1 };
    } else {
      ++out[j].n;
    }
  }
  std::sort(out, out + m, [](const FlankCount& a, const FlankCount& b) { return a.id < b.id; });
  std::stable_sort(out, out + m, [](const FlankCount& a, const FlankCount& b) { return a.n > b.n; });
  return m;
}

// the two acceptance rules shared by all four flank tests (:384-403 etc.):
// top ID seen >= 2 times, or the two top IDs are consecutive and together > 3
inline bool
flank_ok(const FlankCount* v, size_t m, bool need_two_checked)
{
  constexpr size_t MIN_IDS_IN_FLANK = 2;
  if (v[0].n >= MIN_IDS_IN_FLANK) {
    return true;
  }
  if (need_two_checked && m < 2) {
    return false;
  }
  return v[0].n + v[1].n > MIN_IDS_IN_FLANK + 1 && (v[0].id - v[1].id == 1 || v[1].id - v[0].id == 1);
}

} // namespace

bool
eval_flanks(long ls, long le, const uint32_t* ids, size_t n, size_t& trim_start, size_t& trim_end)
{
  constexpr size_t SMALL_READ_THRESHOLD = 15;
  constexpr long MAX_TILES_TO_CHECK = 5;
  const size_t default_start = (ls != 0) ? (size_t)(ls - 1) : (size_t)ls;
  size_t ts = default_start;
  size_t te = (size_t)(le + 1);
  bool good = false;
  FlankCount hist[SMALL_READ_THRESHOLD + 1];

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

void
decide_read(const DecideParams& p, size_t n, const grp_tile_summary* tiles, const grp_id_count* lists, TileWorkspace& ws, ReadDecision& out, FILE* dbg)
{
  out = ReadDecision{};
  out.num_tiles = (uint32_t)n;
  for (size_t i = 0; i < n; ++i) {
    out.hits += tiles[i].hits;
    out.misses += tiles[i].misses;
  }
  const size_t na = smooth_tiles(n, tiles, lists, p.threshold, ws, dbg);
  out.num_assigned = (uint32_t)na;
  const size_t nu = n - na;
  if (nu >= p.unassigned_min && na <= p.assigned_max) { // :967-971
    out.kind = DEC_INSERT_WHOLE;
    return;
  }
  if (na == n) { // :1013
    out.kind = DEC_ASSIGNED_ALL;
    return;
  }
  long ls = 0, le = 0;
  find_longest_stretch(ws.asg, n, ls, le);
  size_t ts = 0, te = 0;
  if (eval_flanks(ls, le, ws.ids.data(), n, ts, te)) {
    out.kind = DEC_INSERT_TRIMMED;
    out.trim_start = (uint32_t)ts;
    out.trim_end = (uint32_t)te;
  } else {
    out.kind = DEC_ASSIGNED;
  }
}

} // namespace gr
