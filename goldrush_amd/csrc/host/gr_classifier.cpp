// See gr_classifier.hpp.  commit() is process_read() after the tile query
// (goldrush_path.cpp:960-1094); silver_path_check() is :156-187.
#include "gr_classifier.hpp"
#include "gr_params.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#if defined(_OPENMP)
#include <omp.h>
#endif

namespace gr {

Classifier::Classifier(const gr_classifier_params& p, const grp_engine_vt& vt, void* ctx)
  : p_(p)
  , vt_(vt)
  , ctx_(ctx)
{
  // the behaviour switches are read ONCE per classifier, not per window
  auto env = [](const char* name) { const char* e = getenv(name); return std::string(e ? e : ""); };
  env_.pipeline = env("GRP_PIPELINE");
  env_.stream = env("GRP_STREAM");
  env_.batch = env("GRP_BATCH");
  {
    const std::string t = env("GRP_MAX_WINDOW_TILES");
    env_.max_window_tiles = t.empty() ? 0 : (uint64_t)std::max(1l, atol(t.c_str()));
  }
  {
    const std::string t = env("GRP_BATCH_OVERLAP");
    if (t == "off") {
      env_.overlap_samples = 0;
    } else if (!t.empty()) {
      env_.overlap_samples = (uint32_t)std::max(1l, atol(t.c_str()));
      env_.overlap_fixed = t.find('+') == std::string::npos; // "4+": the base threshold, still adapting (developer: with GRP_BATCH_OVERLAP_SHIFT)
    }
    const std::string pmin = env("GRP_BATCH_OVERLAP_P");
    if (!pmin.empty()) {
      env_.overlap_min_insert = atof(pmin.c_str());
    }
  }
  if (p_.world == 0) {
    p_.world = 1;
  }
  if (p_.max_window == 0) {
    p_.max_window = 8192;
  }
  if (p_.max_window < p_.world) {
    p_.max_window = p_.world;
  }
}

void
Classifier::set_callbacks(gr_commit_fn commit, gr_rollover_fn rollover, gr_allgather_fn allgather, void* user)
{
  commit_cb_ = commit;
  rollover_cb_ = rollover;
  allgather_cb_ = allgather;
  user_ = user;
  ag_user_ = user;
}

// every committed read goes to the host's callback (the CLI writes its files there) and, inside the ranges
// asked for with keep_commits, into kept_: a witness of what was decided, cheap enough to leave on in a
// measured run (bench.py compares it with the oracle's serial loop on the same reads)
double
Classifier::emit_commit(const gr_commit& ev)
{
  for (int i = 0; i < 2; ++i) {
    if (ev.read >= keep_first_[i] && ev.read - keep_first_[i] < keep_count_[i]) {
      kept_.push_back(ev);
      break;
    }
  }
  return commit_cb_ ? commit_cb_(user_, &ev) : 0.0;
}

void
Classifier::keep_commits(uint32_t first0, uint32_t count0, uint32_t first1, uint32_t count1)
{
  keep_first_[0] = first0;
  keep_count_[0] = count0;
  keep_first_[1] = first1;
  keep_count_[1] = count1;
  kept_.clear();
}

void
Classifier::set_allgather(gr_allgather_fn allgather, void* allgather_user)
{
  allgather_cb_ = allgather;
  ag_user_ = allgather_user;
}

void
Classifier::get_state(gr_classifier_state& s) const
{
  s = gr_classifier_state{};
  s.valid_reads = valid_reads_;
  s.total_tiles = total_tiles_;
  s.assigned_tiles = assigned_tiles_;
  s.unassigned_tiles = unassigned_tiles_;
  s.queries = queries_;
  s.hits = hits_;
  s.misses = misses_;
  s.num_reads_in_path = num_reads_in_path_;
  s.phred_sum_in_path = phred_sum_in_path_;
  s.inserted_bases = inserted_bases_;
  s.curr_path = curr_path_;
  s.id = id_;
  s.ids_inserted = ids_inserted_;
  s.windows = n_windows_;
  s.reads_queried = n_queried_;
  s.reads_committed = n_committed_;
  s.inserts = n_inserts_;
  s.seconds_windows = t_windows_;
  s.seconds_commit = t_commit_;
  s.batches = n_batches_;
  s.batches_undone = n_batch_undone_;
  s.batch_reads = n_batch_reads_;
  s.batches_refused = n_batch_refused_;
  s.batches_fused = n_batch_fused_;
  s.stream_inserts = n_stream_inserts_;
  s.stream_insert_fallbacks = n_stream_insert_fallbacks_;
  s.stream_relaunches = n_stream_relaunches_;
  s.stream_handbacks = n_stream_handbacks_;
  s.stream_rollovers = n_stream_rollovers_;
  s.batch_overlap_cuts = n_batch_overlap_cuts_;
  s.overlap_calls = n_overlap_calls_;
}

void
Classifier::log_path_stat() const
{
  // goldrush_path.cpp:126-154
  const uint64_t cp = curr_path_;
  std::cerr << "Visited " << valid_reads_ << " reads to generate " << cp << " silver paths" << std::endl;
  std::cerr << "Saw: " << total_tiles_ << " tiles to generate " << cp << " silver paths" << std::endl;
  std::cerr << "Assigned: " << assigned_tiles_ << " tiles to generate " << cp << " silver paths" << std::endl;
  std::cerr << "Unassigned: " << unassigned_tiles_ << " tiles to generate " << cp << " silver paths" << std::endl;
  std::cerr << "Total queries: " << queries_ << " to generate " << cp << " silver paths" << std::endl;
  std::cerr << "Total hits: " << hits_ << " to generate " << cp << " silver paths" << std::endl;
  std::cerr << "Total misses: " << misses_ << " to generate " << cp << " silver paths" << std::endl;
  std::cerr << "Num reads: " << num_reads_in_path_ << " in silver path " << cp << std::endl;
  const uint32_t avg_phred = (uint32_t)(-10 * std::log10(phred_sum_in_path_ / inserted_bases_));
  std::cerr << "Average Phred: " << avg_phred << " in silver path " << cp << std::endl;
}

void
Classifier::bump_id()
{
  ++id_;
  if (id_ % 10000 == 0 && p_.rank == 0) {
    std::cerr << "processed " << id_ << " reads" << std::endl;
  }
}

void
Classifier::skip_reads(uint32_t n)
{
  // too short / filtered reads only advance the counter (goldrush_path.cpp:907-932)
  for (uint32_t i = 0; i < n; ++i) {
    bump_id();
  }
}

Classifier::Plan
Classifier::window_plan() const
{
  if (p_.debug) {
    return Plan{ p_.world, false, false }; // one read per rank and round, decided on the host (the dumps are per read, in file order)
  }
  // Choose the speculation window S that maximises committed reads per second
  // under a cost model of one round.  p = insert probability per read, estimated
  // on two time scales (a burst of inserts shrinks the window at once; the long
  // average keeps it from growing to the cap between rare inserts, where every
  // insert would discard half a huge window).
  //   committed(S) = (1 - (1-p)^S) / p        reads up to and including the first insert
  // synchronous round (query, decide, fetch, commit one after the other):
  //   time(S) = t_fixed(S) + S * (t_read / world + t_host)
  // pipelined rounds (the next window runs on the GPU during decide + commit): the
  // slower of GPU and host per window, plus the abandoned next window whenever the
  // current one holds an insert:
  //   time(S) = max(g_fix + S * t_read / world, h_fix + S * t_host) * (1 + P_ins(S))
  static const double factor = [] {
    const char* e = getenv("GRP_SPEC_FACTOR");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 1.0;
  }();
  // GRP_PIPELINE=off: synchronous rounds only; =force: pipelined whenever the window
  // cap allows (tests)
  const bool no_pipeline = env_.pipeline == "off";
  const bool force_pipeline = env_.pipeline == "force";
  const bool force_stream = env_.stream == "force";
  // With the batches of batch_round taking every stretch where more than ~1 % of the reads insert,
  // the windows only see isolated inserts: the 32-read average (one insert = 3 %) would answer
  // each of them with a handful of small synchronous windows (~0.8 ms per insert, measured on
  // C2); the 256-read average follows a real change of regime within a few inserts.
  static const bool fast_p = getenv("GRP_PLAN_FAST_P") != nullptr; // developer hook: the 32-read average as before
  const double p_recent = (can_batch() && !fast_p) ? p_insert_mid_ : p_insert_;
  const double p = std::min(1.0, std::max(std::max(p_recent, p_insert_slow_), 1e-7));
  const double world = (double)p_.world;
  // measured on MI355X (bench.py --trace): a synchronous round costs ~70 us when it
  // takes the latency path (host decision, < 16 reads), ~170 us with the decision
  // kernel, plus ~0.15 us of ordered host commit per read; multi-rank rounds add the
  // all-gather.  Pipelined: ~40 us of ramp-up / tail per query launch on the GPU,
  // ~60 us of launches and waits per window on the host.
  const double t_gather = p_.world > 1 ? 200e-6 : 0.0;
  const double t_small = 70e-6, t_large = 170e-6 + t_gather;
  const double g_fix = 40e-6, h_fix = 60e-6 + t_gather;
  const double t_host = 0.15e-6;
  const double t_read = std::max(1e-9, avg_probes_per_read_ / 46e9); // query kernel ~46 G probes/s
  const bool can_pipe = vt_.classify_begin && vt_.classify_end && !no_pipeline;
  double best_rate = 0.0;
  uint32_t best = 1;
  bool best_pipe = false;
  const double lq = std::log1p(-std::min(p, 0.999999));
  for (uint32_t S = 1;; S = (S < 8) ? S + 1 : S + S / 4) {
    if (S > p_.max_window) {
      S = p_.max_window;
    }
    const double p_ins = 1.0 - std::exp(lq * S);
    const double committed = p_ins / p;
    const double t_sync = (S < 16 * p_.world ? t_small : t_large) + S * (t_read / world + t_host);
    if (committed / t_sync > best_rate && !(force_pipeline && best_pipe)) {
      best_rate = committed / t_sync;
      best = S;
      best_pipe = false;
    }
    if (can_pipe && S >= (force_pipeline ? 1 : 16) * p_.world) {
      const double t_pipe = std::max(g_fix + S * t_read / world, h_fix + S * t_host) * (1.0 + p_ins);
      if (committed / t_pipe > best_rate || (force_pipeline && !best_pipe)) {
        best_rate = committed / t_pipe;
        best = S;
        best_pipe = true;
      }
    }
    if (S == p_.max_window) {
      break;
    }
  }
  uint32_t w = (uint32_t)std::min<double>(std::max(best * factor, 1.0), (double)p_.max_window);
  if (p_.world > 1) {
    w = ((w + p_.world - 1) / p_.world) * p_.world;
  }
  if (can_stream() && p_.max_window >= 32) {
    // streaming launch: no per-window round trip and a stale window is cut short by the
    // abort flag, so an insert costs one drain + relaunch (~120 us: resident workgroups
    // finish, insert kernels, ramp-up) whatever the window size
    static const double t_abort_env = [] { // developer hook: what an insert costs a streaming launch (us)
      const char* e = getenv("GRP_T_ABORT_US");
      return e ? atof(e) * 1e-6 : 0.0;
    }();
    // a window that ends at the insert: drain + insert kernels + relaunch, ~400 us on C2 (tools/abort_matrix.sh);
    // a window that applies the insert itself (round 3): ~240 us until the first record behind it, ~100 us of
    // device time — 200 us fitted on C2's transition zone (tools/dev/r3_thresholds.sh)
    // (several ranks: what the RANKS agreed on for the last window, not this rank's own view — the plan must come out the same on every rank)
    const bool resume_plan = p_.world == 1 ? can_resume() : (ranks_resume_ok_ && vt_.stream_begin_striped_resumable && vt_.stream_resumable && vt_.stream_insert_done && vt_.stream_insert && vt_.insert_read);
    const double t_abort = t_abort_env > 0.0 ? t_abort_env : (resume_plan ? 200e-6 : 400e-6);
    // the window size does not matter to an abort (only the resident workgroups are lost),
    // so the launches are as long as allowed.  Several ranks: every rank works on its own
    // stripe of the current group, an insert also discards about half a group; the
    // stripe exchange (~150 us per group) is hidden behind the launches unless it is the
    // longer of the two.
    // (round 5: a window of several ranks that applies inserts itself only leaves on the host's word — the GPUs idle for
    // the host's lag behind them at every window's end, ~100 us — and a rank's launch covers 1 / world of it: the
    // windows grow with the number of ranks)
    const uint32_t S = (p_.world > 1 && resume_plan) ? (uint32_t)std::min<uint64_t>((uint64_t)p_.max_window * std::min<uint32_t>(p_.world, 8u), 1u << 16) : p_.max_window;
    // a record handed back (a tile needed the worst-case table) costs an abort plus a
    // synchronous single-read round: where that is frequent (large h on repeat-rich
    // data) the windows that redo flagged tiles in bulk win
    const double t_redo = t_abort + t_small + t_read;
    double per_read = t_read / world + t_host + p * t_abort + p_redo_ * t_redo + g_fix / S;
    if (p_.world > 1) {
      const double group = (double)stripe_reads() * world;
      per_read = std::max(t_read / world, 150e-6 / group + t_host) + p * (t_abort + 0.5 * group * t_read / world) + p_redo_ * t_redo + g_fix / S;
    }
    if (force_stream || 1.0 / per_read > best_rate) {
      return Plan{ S, false, true };
    }
  }
  return Plan{ w, best_pipe, false };
}

bool
Classifier::can_stream() const
{
  return vt_.stream_begin && vt_.stream_abort && vt_.stream_poll && vt_.stream_end && (p_.world == 1 || allgather_cb_) && env_.stream != "off";
}

uint32_t
Classifier::stripe_reads() const
{
  if (p_.world == 1) {
    return 0;
  }
  // Ranks exchange finished stripes (one all-gather per group of `world` stripes) while
  // their launches run: long enough to hide the exchange behind the stripe's GPU time
  // (~1.6 us per 25 kb read), short enough that an insert discards little.
  // The all-gather of a group takes ~100 us + ~30 us per rank on a CPU group; a stripe
  // must cover it with GPU time: 128 reads for 2 ranks, 224 for 8.
  static const uint32_t env_stripe = [] {
    const char* e = getenv("GRP_STRIPE");
    const long v = e ? atol(e) : 0;
    return v > 0 ? (uint32_t)v : 0u;
  }();
  const uint32_t stripe = env_stripe ? env_stripe : 96u + 16u * p_.world;
  return stripe;
}

// The engine takes at most 2^22 tiles per synchronous / pipelined window or batch (a HIP grid holds
// fewer than 2^32 work-items) and 2^30 per streaming window: windows are capped in reads by the
// plan, this caps them in tiles (small -t, very long reads).  At least one read.
uint32_t
Classifier::clamp_tiles(uint32_t pos, uint32_t S, uint64_t max_tiles) const
{
  if (env_.max_window_tiles) { // tests: a tiny cap (GRP_MAX_WINDOW_TILES)
    max_tiles = std::min<uint64_t>(max_tiles, env_.max_window_tiles);
  }
  if (S <= 1 || tile0_[(size_t)pos + S] - tile0_[pos] <= max_tiles) {
    return S;
  }
  const auto first = tile0_.begin() + pos;
  const auto it = std::upper_bound(first, first + S + 1, tile0_[pos] + max_tiles); // first prefix value beyond the cap
  return (uint32_t)std::max<ptrdiff_t>(1, (it - first) - 1);
}

int
Classifier::launch_stream(void* reads, uint32_t pos, uint32_t S, uint32_t slot, StreamFlight& f)
{
  S = clamp_tiles(pos, S, 1ull << 30);
  const grp_decide_params dp{ p_.threshold, p_.unassigned_min, p_.assigned_max, 0 };
  const gr_read_decision* dec = nullptr;
  // One rank: a window that waits where it parks and applies the insert the host commits inside its own
  // launch (stream_round; a silver-path run resets the ID array at a rollover: commit() ends the launches in
  // front of that insert); several ranks share a window in stripes: those windows end where they park.
  bool resumable = can_resume();
  int rc;
  if (p_.world == 1) {
    rc = resumable ? vt_.stream_begin_resumable(ctx_, reads, base_ + pos, S, &dp, slot, &dec) : vt_.stream_begin(ctx_, reads, base_ + pos, S, &dp, slot, 0, 1, 0, &dec);
  } else {
    rc = resumable ? vt_.stream_begin_striped_resumable(ctx_, reads, base_ + pos, S, &dp, slot, stripe_reads(), p_.world, p_.rank, &dec)
                   : vt_.stream_begin(ctx_, reads, base_ + pos, S, &dp, slot, stripe_reads(), p_.world, p_.rank, &dec);
    if (vt_.stream_begin_striped_resumable && vt_.stream_resumable) {
      // What the window can do is the RANKS' matter: one rank whose launch ends where it parks (the runtime refused the
      // cooperative launch, parked windows switched off after a refused insert) or that cannot begin the window now
      // (GRP_ERR_BUSY) — and every rank takes that form, or they would walk different stripes of different windows.
      // Every rank calls this at the same point of the same sequence of records, whatever its own call returned.
      uint32_t mine[2] = { rc == GRP_OK ? 0u : rc == GRP_ERR_BUSY ? 1u : 2u, (rc == GRP_OK && resumable && vt_.stream_resumable(ctx_, slot) == 1) ? 1u : 0u };
      std::vector<uint32_t> all((size_t)2 * p_.world);
      if (!allgather_cb_ || allgather_cb_(ag_user_, mine, sizeof(mine), all.data()) != 0) {
        err_ = "allgather callback failed";
        if (rc == GRP_OK) {
          (void)vt_.stream_abort(ctx_, slot);
          (void)vt_.stream_end(ctx_, slot, nullptr);
        }
        return GRP_ERR_INVALID;
      }
      uint32_t worst = 0;
      bool all_resumable = true;
      for (uint32_t q = 0; q < p_.world; ++q) {
        worst = std::max(worst, all[(size_t)2 * q]);
        all_resumable = all_resumable && all[(size_t)2 * q + 1] != 0;
      }
      if (worst != 0) {
        if (rc == GRP_OK) { // another rank could not: this rank's window goes as well
          (void)vt_.stream_abort(ctx_, slot);
          (void)vt_.stream_end(ctx_, slot, nullptr);
        }
        if (worst == 1 && (rc == GRP_OK || rc == GRP_ERR_BUSY)) {
          return GRP_ERR_BUSY;
        }
        if (rc == GRP_OK) {
          err_ = "stream_begin failed on another rank";
          return GRP_ERR_STATE;
        }
      }
      resumable = all_resumable;
      ranks_resume_ok_ = all_resumable;
    }
  }
  if (rc != GRP_OK) {
    err_ = std::string("stream_begin: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
    return rc;
  }
  f.active = true;
  f.pos = pos;
  f.S = S;
  f.slot = slot;
  f.dec = dec;
  f.resumable = resumable;
  f.gen = 1;
  f.ins_posted = false;
  ++n_windows_;
  return GRP_OK;
}

int
Classifier::end_stream(StreamFlight& f)
{
  if (!f.active) {
    return GRP_OK;
  }
  f.active = false;
  uint32_t decided = 0;
  int rc = vt_.stream_end(ctx_, f.slot, &decided);
  if (rc == 1 || rc == 2) {
    // the launch ended without the insert it had been handed: nothing was inserted; the classic call applies it,
    // stream-ordered behind whatever is queued.  2 (one rank: also 1): its workgroups were not all resident — a shared
    // device — and windows end at inserts again for a while; 1 with several ranks: the launch had left before the
    // command came (its own stripes were decided, the inserting read was another rank's) — nothing is wrong
    if (rc == 2 || p_.world == 1) {
      resume_disabled_ = true;
      resume_clean_windows_ = 0;
    }
    ++n_stream_insert_fallbacks_;
    rc = GRP_OK;
    if (f.ins_posted) {
      rc = vt_.insert_read(ctx_, rg_.reads, f.ins_read, f.ins_ts, f.ins_te, p_.block_size, f.ins_first_id, f.ins_off);
      if (rc != GRP_OK) {
        err_ = std::string("insert_read: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
      }
    }
  } else if (rc != GRP_OK) {
    err_ = std::string("stream_end: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
    return rc;
  }
  f.ins_posted = false;
  n_queried_ += decided;
  if (resume_disabled_ && rc == GRP_OK && ++resume_clean_windows_ >= 64) {
    resume_disabled_ = false; // 64 windows ended cleanly since the last refused insert: parked windows are tried again
  }
  return rc;
}

static constexpr uint64_t kMaxWindowTiles = 1ull << 22; // grp_classify_reads / grp_query_tiles / grp_batch_* take no more per call

// spin until record j of the window is complete
static constexpr int STREAM_LOST = 100;

int
Classifier::wait_record(const StreamFlight& f, uint32_t j)
{
  const uint32_t* flag = &f.dec[j].pad;
  uint32_t spins = 0;
  bool finished_seen = false;
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != f.gen) { // (an older generation: decided before an insert in front of the read, stale)
    __builtin_ia32_pause();
    if ((++spins & 0x3FFFu) == 0) {
      if (finished_seen) {
        // The launch is over and this record never came.  With an insert posted: the launch gave up on it
        // (end_stream applies it).  Without: a parked window that was told nothing for its idle limit (a host
        // stopped by a debugger, SIGSTOP, a stalled file system) has left by itself — nothing was modified, the
        // window is ended and begun again at this read.  Twice in a row at the same read is an error.
        if (!f.ins_posted && lost_at_ == (uint64_t)base_ + f.pos + j) {
          err_ = "streaming window finished without deciding one of its reads";
          return GRP_ERR_STATE;
        }
        lost_at_ = f.ins_posted ? UINT64_MAX : (uint64_t)base_ + f.pos + j;
        if (!f.ins_posted) {
          ++n_stream_relaunches_;
        }
        return STREAM_LOST;
      }
      const int st = vt_.stream_poll(ctx_, f.slot);
      if (st < 0) {
        err_ = std::string("stream_poll: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
        return st;
      }
      finished_seen = st == 1; // one more round of spinning for the last store to land
    }
  }
  if (lost_at_ == (uint64_t)base_ + f.pos + j) {
    lost_at_ = UINT64_MAX; // the window begun again has decided the read: a later idle exit at the same index of ANOTHER run() is a new event
  }
  return GRP_OK;
}

int
Classifier::gather_decisions(uint32_t q)
{
  const uint32_t world = p_.world;
  if (world > 1) {
    dec_all_.resize((size_t)q * world);
    if (!allgather_cb_) {
      err_ = "world > 1 but no allgather callback";
      return GRP_ERR_INVALID;
    }
    int rc = allgather_cb_(ag_user_, dec_.data(), (uint64_t)q * sizeof(gr_read_decision), dec_all_.data());
    if (rc != 0) {
      err_ = "allgather callback failed";
      return GRP_ERR_INVALID;
    }
  } else {
    dec_all_.swap(dec_);
  }
  ++n_windows_;
  return GRP_OK;
}

// enqueue this rank's slice of the window [pos, pos+S) in an engine slot
int
Classifier::launch_window(void* reads, uint32_t pos, uint32_t S, uint32_t slot, Flight& f)
{
  const uint32_t world = p_.world;
  const uint32_t q = (S + world - 1) / world;
  const uint32_t my_first = std::min<uint64_t>((uint64_t)pos + (uint64_t)p_.rank * q, (uint64_t)pos + S);
  const uint32_t my_count = std::min<uint32_t>(q, pos + S - my_first);
  const grp_decide_params dp{ p_.threshold, p_.unassigned_min, p_.assigned_max, 0 };
  int rc = vt_.classify_begin(ctx_, reads, base_ + my_first, my_count, &dp, slot);
  if (rc != GRP_OK) {
    err_ = std::string("classify_begin: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
    return rc;
  }
  n_queried_ += my_count;
  f.active = true;
  f.pos = pos;
  f.S = S;
  f.slot = slot;
  f.q = q;
  f.my_count = my_count;
  return GRP_OK;
}

int
Classifier::finish_window(Flight& f)
{
  f.active = false;
  dec_.assign(f.q, gr_read_decision{});
  int rc = vt_.classify_end(ctx_, f.slot, dec_.data());
  if (rc != GRP_OK) {
    err_ = std::string("classify_end: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
    return rc;
  }
  return gather_decisions(f.q);
}

void
Classifier::abandon_window(Flight& f)
{
  if (f.active) {
    (void)vt_.classify_end(ctx_, f.slot, nullptr); // results are stale: do not wait for them
    f.active = false;
  }
}

int
Classifier::query_window(void* reads, const uint32_t* lens, uint32_t first, uint32_t count)
{
  (void)lens;
  const uint32_t world = p_.world;
  const uint32_t q = (count + world - 1) / world; // reads per rank
  const uint32_t my_first = std::min<uint64_t>((uint64_t)first + (uint64_t)p_.rank * q, (uint64_t)first + count);
  const uint32_t my_count = std::min<uint32_t>(q, first + count - my_first);
  dec_.assign(q, gr_read_decision{});
  // tiny windows (insert-heavy phases) are latency-bound: one kernel + host decision
  // is shorter than two kernels
  if (my_count >= 16 && vt_.classify_reads) {
    // query + decision on the device: 32 bytes per read come back
    const grp_decide_params dp{ p_.threshold, p_.unassigned_min, p_.assigned_max, 0 };
    int rc = vt_.classify_reads(ctx_, reads, base_ + my_first, my_count, &dp, dec_.data());
    if (rc != GRP_OK) {
      err_ = std::string("classify_reads: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
      return rc;
    }
    n_queried_ += my_count;
  } else if (my_count) {
    const uint64_t nt = tile0_[my_first + my_count] - tile0_[my_first];
    tiles_.resize(nt ? nt : 1);
    if (lists_.size() < 4 * nt + 1024) {
      lists_.resize(4 * nt + 1024);
    }
    for (;;) {
      uint64_t used = 0;
      int rc = vt_.query_tiles(ctx_, reads, base_ + my_first, my_count, tiles_.data(), lists_.data(), lists_.size(), &used, nullptr);
      if (rc == GRP_ERR_NOMEM && used > lists_.size()) {
        lists_.resize(used + used / 4);
        continue;
      }
      if (rc != GRP_OK) {
        err_ = std::string("query_tiles: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
        return rc;
      }
      break;
    }
    n_queried_ += my_count;
    const DecideParams dp{ p_.threshold, p_.unassigned_min, p_.assigned_max };
    const uint64_t t_base = tile0_[my_first];
#if defined(_OPENMP)
    static const int kCpus = (int)std::min<unsigned>(effective_cpus(), 32);
    const int nthreads = (my_count >= 256) ? std::max(1, std::min(omp_get_max_threads(), kCpus)) : 1;
#else
    const int nthreads = 1;
#endif
    if ((int)ws_.size() < nthreads) {
      ws_.resize(nthreads);
    }
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (uint32_t j = 0; j < my_count; ++j) {
#if defined(_OPENMP)
      TileWorkspace& ws = ws_[omp_get_thread_num()];
#else
      TileWorkspace& ws = ws_[0];
#endif
      const uint64_t a = tile0_[my_first + j] - t_base;
      const uint64_t b = tile0_[my_first + j + 1] - t_base;
      ReadDecision rd;
      if (p_.debug) { // single-threaded here (windows of one read): the dumps are kept until the read is committed
        char* buf = nullptr;
        size_t len = 0;
        FILE* mem = open_memstream(&buf, &len);
        decide_read(dp, (size_t)(b - a), tiles_.data() + a, lists_.data(), ws, rd, mem);
        fclose(mem);
        if (debug_text_.size() <= j) {
          debug_text_.resize((size_t)j + 1);
        }
        debug_text_[j].assign(buf ? buf : "", len);
        free(buf);
      } else {
        decide_read(dp, (size_t)(b - a), tiles_.data() + a, lists_.data(), ws, rd);
      }
      static_assert(sizeof(ReadDecision) == sizeof(gr_read_decision), "decision layout");
      std::memcpy(&dec_[j], &rd, sizeof(rd));
    }
  }
  return gather_decisions(q);
}

void
Classifier::silver_path_check(int& rc)
{
  // goldrush_path.cpp:156-187
  if (p_.target_bases < inserted_bases_) {
    if (p_.verbose && p_.rank == 0) {
      log_path_stat();
    }
    ++curr_path_;
    if (p_.max_paths < curr_path_) {
      finished_ = true; // exit(0) in the reference
      return;
    }
    inserted_bases_ = 0;
    num_reads_in_path_ = 0;
    phred_sum_in_path_ = 0;
    int e = vt_.reset_ids(ctx_);
    if (e != GRP_OK) {
      rc = e;
      err_ = std::string("reset_ids: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
      return;
    }
    if (rollover_cb_) {
      rollover_cb_(user_, curr_path_);
    }
    ids_inserted_ = 0;
    last_insert_shares_id_ = false;
  }
}

// returns true when the read changed the miBF (everything queried after it is stale)
bool
Classifier::commit(void* reads, const uint32_t* lens, uint32_t r, const gr_read_decision& d, int& rc, bool engine_inserted, uint32_t engine_first_id)
{
  const uint32_t len = lens[r];
  const uint32_t tile = p_.tile_length, block = p_.block_size, k = p_.kmer_size;
  const uint32_t nt = d.num_tiles;
  total_tiles_ += nt;
  if (nt) {
    // one query per frame (:567-568); only the last tile can be clipped
    const uint32_t start = (nt - 1) * tile;
    const uint32_t Lp = std::min(tile + k - 1, len - start);
    queries_ += (uint64_t)(nt - 1) * tile + (Lp >= k ? Lp - k + 1 : 0);
  }
  hits_ += d.hits;
  misses_ += d.misses;
  assigned_tiles_ += d.num_assigned;
  unassigned_tiles_ += nt - d.num_assigned;

  gr_commit ev{};
  ev.read = base_ + r;
  ev.dec = d;
  ev.path = curr_path_;
  bool inserted = false;

  auto insert_block = [&](uint32_t ts, uint32_t te, uint32_t id) {
    if (rc != GRP_OK || engine_inserted) {
      return;
    }
    int e = vt_.insert_tiles(ctx_, reads, base_ + r, ts, te, id);
    if (e != GRP_OK) {
      rc = e;
      err_ = std::string("insert_tiles: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
    }
  };

  // all ID blocks of the read in one engine call (same result as the block loop)
  // `bases`: what the insert adds to the path (silver mode: the host knows in front of the insert whether the path
  // rolls over behind it, :156-187 — the ID array is reset there, a parked window has nothing to carry on with)
  auto insert_read = [&](uint32_t ts, uint32_t te_excl, uint32_t id_offset, uint64_t bases) {
    if (engine_inserted) {
      // the batch applied the insert ahead of this commit, with the ID this counter allocates now
      if (engine_first_id != ids_inserted_) {
        rc = GRP_ERR_STATE;
        err_ = "batch: the engine inserted block ID " + std::to_string(engine_first_id) + ", the host allocates " + std::to_string(ids_inserted_);
      }
      return;
    }
    if (stream_ins_) {
      // the record comes out of a streaming window that is parked at it: the launch applies the
      // insert itself and carries on behind the read (no launch boundary)
      StreamFlight& f = *stream_ins_;
      uint32_t gen = 0;
      const bool rolls_over = p_.silver_path && p_.target_bases < inserted_bases_ + bases;
      if (rolls_over) {
        ++n_stream_rollovers_;
      }
      if (!rolls_over && vt_.stream_insert(ctx_, f.slot, base_ + r, ts, te_excl, block, ids_inserted_, id_offset, &gen) == GRP_OK) {
        f.gen = gen;
        f.ins_posted = true;
        f.ins_read = base_ + r;
        f.ins_ts = ts;
        f.ins_te = te_excl;
        f.ins_first_id = ids_inserted_;
        f.ins_off = id_offset;
        stream_ins_ok_ = true;
        ++n_stream_inserts_;
        return;
      }
      // this window cannot: end the launches, then the insert behind them as before
      (void)vt_.stream_abort(ctx_, scur_.slot);
      if (snext_.active) {
        (void)vt_.stream_abort(ctx_, snext_.slot);
      }
    }
    int e = vt_.insert_read(ctx_, reads, base_ + r, ts, te_excl, block, ids_inserted_, id_offset);
    if (e != GRP_OK) {
      rc = e;
      err_ = std::string("insert_read: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
    }
  };

  if (p_.debug && p_.rank == 0) { // :979, :1016, :1037, :1084
    std::cerr << (d.kind == DEC_INSERT_WHOLE ? "unassigned" : d.kind == DEC_ASSIGNED_ALL ? "complete assignment" : d.kind == DEC_INSERT_TRIMMED ? "trimmed" : "assigned") << std::endl;
  }
  switch (d.kind) {
    case DEC_INSERT_WHOLE: {
      // :978-1011
      ++ids_inserted_;
      ev.first_id = ids_inserted_;
      if (vt_.insert_read || engine_inserted) {
        insert_read(0, nt, 0, len);
      } else {
        for (uint32_t bs = 0; bs < nt; bs += block) {
          insert_block(bs, std::min(bs + block, nt), ids_inserted_ + (uint32_t)(bs / block));
        }
      }
      ids_inserted_ = ids_inserted_ + (uint32_t)(len / ((size_t)tile * block));
      last_insert_shares_id_ = false;
      const double ph = emit_commit(ev);
      inserted_bases_ += len;
      ++num_reads_in_path_;
      phred_sum_in_path_ += ph;
      if (p_.silver_path) {
        silver_path_check(rc);
      }
      inserted = true;
      break;
    }
    case DEC_ASSIGNED_ALL: {
      // :1013-1023
      (void)emit_commit(ev);
      ++valid_reads_;
      bump_id();
      return false;
    }
    case DEC_INSERT_TRIMMED: {
      // :1038-1080
      const uint32_t ts = d.trim_start, te = d.trim_end;
      ++ids_inserted_;
      ev.first_id = ids_inserted_;
      // new_seq = seq.substr(ts*tile, te == nt-1 ? npos : (te-ts+1)*tile)
      const uint64_t off = (uint64_t)ts * tile;
      uint64_t n_out = len - off;
      if (te != nt - 1) {
        n_out = std::min<uint64_t>(n_out, (uint64_t)(te - ts + 1) * tile);
      }
      if (vt_.insert_read || engine_inserted) {
        insert_read(ts, te + 1, 1, n_out);
      } else {
        for (uint64_t bs = ts; bs <= te; bs += block) {
          const uint64_t be = std::min<uint64_t>(bs + block - 1, te);
          insert_block((uint32_t)bs, (uint32_t)be + 1, ids_inserted_ + (uint32_t)((bs - ts + 1) / block));
        }
      }
      ids_inserted_ = ids_inserted_ + (uint32_t)((te - ts) / block);
      // The blocks of a trimmed read are numbered from (0 + 1) / block (:1048-1049) while the counter
      // advances by (te - ts) / block (:1074): with block == 1 the LAST block holds ID ids_inserted_ + 1,
      // the next insert's first; with block > 1 block j holds first + j and the counter ends on the
      // last of them, nothing is shared (the rule used to be (te - ts + 1) % block == 0,
      // exact but needlessly wide — every probe with ID == floor then took the slow look-up)
      last_insert_shares_id_ = block == 1;
      inserted_bases_ += n_out;
      const double ph = emit_commit(ev);
      ++num_reads_in_path_;
      phred_sum_in_path_ += ph;
      if (p_.silver_path) {
        silver_path_check(rc);
      }
      inserted = true;
      break;
    }
    default: // DEC_ASSIGNED (:1083-1088)
      (void)emit_commit(ev);
      break;
  }
  if (finished_) {
    return inserted; // exit(0) inside silver_path_check: no ++id
  }
  ++valid_reads_;
  bump_id();
  return inserted;
}

// one read of the current range: the reads filtered out before it only advance the
// counter, then process_read's tail; keeps the insert-rate estimates up to date
bool
Classifier::commit_one(uint32_t r, const gr_read_decision& d, int& rc, bool engine_inserted, uint32_t engine_first_id)
{
  if (rg_.skipped_before) {
    skip_reads(rg_.skipped_before[r]);
  }
  if (p_.debug && p_.rank == 0) {
    if (debug_cb_) {
      debug_cb_(user_, base_ + r); // skipped records in front of it, "name:", "num tiles:" (:907-941)
    }
    if (debug_window_pos_ <= r && r - debug_window_pos_ < debug_text_.size()) {
      std::cerr << debug_text_[r - debug_window_pos_];
    }
    std::cerr << "num assigned tiles: " << d.num_assigned << "\n"
              << "num unassigned tiles: " << d.num_tiles - d.num_assigned << std::endl; // :957-964
  }
  const uint64_t path_before = curr_path_;
  const bool ins = commit(rg_.reads, rg_.lens, r, d, rc, engine_inserted, engine_first_id);
  if (rc != GRP_OK) {
    return ins;
  }
  ++n_committed_;
  p_insert_ += (1.0 / 32.0) * ((ins ? 1.0 : 0.0) - p_insert_);
  p_insert_mid_ += (1.0 / 256.0) * ((ins ? 1.0 : 0.0) - p_insert_mid_);
  if (curr_path_ != path_before) {
    p_insert_ = p_insert_mid_ = 1.0; // a new silver path starts on an empty ID array: every read inserts
  }
  p_insert_slow_ += (1.0 / 8192.0) * ((ins ? 1.0 : 0.0) - p_insert_slow_);
  p_redo_ -= p_redo_ / 4096.0; // forgotten slowly where no streaming window measures it
  n_inserts_ += ins ? 1 : 0;
  return ins;
}

int
Classifier::drop_streams()
{
  int rc = GRP_OK;
  for (StreamFlight* f : { &scur_, &snext_ }) { // both flags first: the window queued behind must not start on a state an insert is still missing from
    if (f->active) {
      (void)vt_.stream_abort(ctx_, f->slot);
    }
  }
  for (StreamFlight* f : { &scur_, &snext_ }) {
    if (f->active) {
      const int e = end_stream(*f);
      if (rc == GRP_OK) {
        rc = e;
      }
    }
  }
  return rc;
}

// Decision of read j of the streaming window scur_.  One rank reads its own record as
// soon as it is complete; several ranks exchange the stripes of a group (`world`
// consecutive stripes = world * stripe consecutive reads) once each has finished its own.
int
Classifier::stream_decision(uint32_t j, gr_read_decision& d)
{
  const uint32_t W = p_.world, S = scur_.S;
  if (W == 1) {
    const int e = wait_record(scur_, j);
    if (e == GRP_OK) {
      d = scur_.dec[j];
      scur_.ins_posted = false; // a record of the generation behind the posted insert: the launch has applied it
    }
    return e;
  }
  // A rank's block of the exchange: its stripe's C records behind one header record — .kind of the header != 0: this
  // rank's launch has left without a record it owes (wait_record: STREAM_LOST).  Every rank then ends the round at this
  // read, together: the ranks walk one sequence of records, whatever happens to one launch (round 5; a window of
  // several ranks used to end at every insert, so a launch could not be lost behind one).
  const uint32_t C = stripe_reads(), GW = C * W, CB = C + 1;
  if (group_base_ == UINT32_MAX || j < group_base_ || j >= group_base_ + GW) {
    group_base_ = (j / GW) * GW;
    const uint32_t lo = std::min(group_base_ + p_.rank * C, S), hi = std::min(lo + C, S);
    stripe_send_.assign(CB, gr_read_decision{});
    int lost = GRP_OK;
    if (ins_lost_) {
      stripe_send_[0].kind = 1;
      lost = STREAM_LOST;
    }
    for (uint32_t q = std::max(lo, group_from_); q < hi && lost == GRP_OK; ++q) { // (reads below group_from_ are committed: the group is gathered again behind an insert the launches applied)
      const int e = wait_record(scur_, q);
      if (e == STREAM_LOST) {
        lost = e;
        stripe_send_[0].kind = 1;
        break;
      }
      if (e != GRP_OK) {
        return e;
      }
      stripe_send_[1 + q - lo] = scur_.dec[q];
      const uint32_t kind = scur_.dec[q].kind;
      if (kind == DEC_INSERT_WHOLE || kind == DEC_INSERT_TRIMMED || kind == 0) {
        break; // the launch parks itself behind such a record: later records of the stripe never come, and no rank reads them
      }
    }
    stripe_recv_.resize((size_t)CB * W);
    if (allgather_cb_(ag_user_, stripe_send_.data(), (uint64_t)CB * sizeof(gr_read_decision), stripe_recv_.data()) != 0) {
      err_ = "allgather callback failed";
      return GRP_ERR_INVALID;
    }
    ins_unconfirmed_ = false;
    for (uint32_t q = 0; q < W; ++q) {
      if (stripe_recv_[(size_t)q * CB].kind != 0) {
        lost = STREAM_LOST;
      }
    }
    if (lost != GRP_OK) {
      group_base_ = UINT32_MAX;
      return lost;
    }
  }
  const uint32_t g = j - group_base_;
  d = stripe_recv_[(size_t)(g / C) * CB + 1 + g % C];
  return GRP_OK;
}

// ---- windows committed as batches ------------------------------------------------------
// Where many reads insert, a classic window ends at its first insert: one round trip (~80 us)
// per inserting read.  Two reads of a window hardly ever influence each other (they would have
// to overlap on the genome), so the whole window is decided against the state in front of it,
// its inserts are applied at once, and a second query — every read against the state in front of
// its own insert, through the engine's log — confirms the decisions or names the first read
// where the serial loop would have gone another way (include/grpath.h, grp_batch_*).
bool
Classifier::can_batch() const
{
  return vt_.batch_insert && vt_.batch_classify && vt_.batch_undo && vt_.batch_end && vt_.classify_reads && vt_.insert_read && !p_.debug && p_.max_window >= 2 && env_.batch != "off";
}

bool
Classifier::want_batch() const
{
  if (!can_batch() || batch_bypass_) {
    return false;
  }
  if (env_.batch == "force") {
    return true;
  }
  // a batch costs two queries per read whatever the insert rate (~4.5 us on C2); the classic
  // windows cost one query per read plus ~80-400 us per insert: measured on C2 (bench.py --trace)
  // the batch wins down to ~1 % inserting reads
  static const double p_in_env = [] {
    const char* v = getenv("GRP_BATCH_ENTER");
    return v ? atof(v) : 0.0;
  }();
  static const double p_out_env = [] {
    const char* v = getenv("GRP_BATCH_LEAVE");
    return v ? atof(v) : 0.0;
  }();
  // (windows that apply inserts themselves are cheaper per insert: the batches take over later)
  const double p_in = p_in_env > 0.0 ? p_in_env : (can_resume() ? 0.020 : 0.012);
  const double p_out = p_out_env > 0.0 ? p_out_env : (can_resume() ? 0.012 : 0.007);
  return p_insert_mid_ >= (in_batch_ ? p_out : p_in);
}

// one rank, an engine with the entry points: streaming windows apply inserts inside their launch (round 4: in silver
// mode too — the insert behind which the path rolls over is known to the host in front of it and ends the launches)
bool
Classifier::can_resume() const
{
  // (round 5: several ranks too — every rank's launch applies the insert on its replica, parked by the command where the
  // inserting read is another rank's; the ranks agree per window that all of them can, launch_stream)
  static const bool ranks_off = getenv("GRP_STREAM_RESUME_RANKS") && !strcmp(getenv("GRP_STREAM_RESUME_RANKS"), "off"); // developer switch: round 4's form, windows of several ranks end at inserts
  const bool entry = p_.world == 1 ? vt_.stream_begin_resumable != nullptr : (vt_.stream_begin_striped_resumable && vt_.stream_resumable && vt_.stream_insert_done && !ranks_off);
  return entry && vt_.stream_insert && vt_.insert_read && !resume_disabled_;
}

// The size of the next batch.  A read decides differently in a batch when it overlaps a read
// INSERTED in front of it in the same batch: with a fraction p of inserting reads the chance that
// read x is the first one grows like c * p * x, so a batch of B reads is confirmed with
// probability exp(-c p B^2 / 2).  c is estimated from what the batches did (first differing
// reads over (inserted read, later read) pairs exposed, both decaying); B maximises reads
// committed per unit of time, a batch costing a fixed part (launches, waits: worth ~70 reads,
// measured) plus its reads, and a batch taken back its undo pass on top.
void
Classifier::batch_feedback(uint32_t reads, uint32_t bad, double exposure)
{
  const bool failed = bad < reads;
  bf_fail_ = 0.98 * bf_fail_ + (failed ? 1.0 : 0.0);
  bf_expo_ = 0.98 * bf_expo_ + exposure;
  static const uint32_t fixed = [] { // developer hook: a fixed batch size
    const char* e = getenv("GRP_BATCH_READS");
    return e ? (uint32_t)std::max(2l, atol(e)) : 0u;
  }();
  if (fixed) {
    batch_reads_ = fixed;
    return;
  }
  const double p = std::min(1.0, std::max(p_insert_mid_, 1e-3));
  const double c = std::max(bf_fail_, 0.05) / std::max(bf_expo_, 1.0) * p;
  const double fixed_cost = 70.0;
  double best_rate = 0.0, area = 0.0, prev = 0.0;
  uint32_t best = 8;
  for (double b = 8.0; b <= 16384.0; b *= 1.125) {
    // area = integral of the survival function up to b = expected reads confirmed
    const int steps = 8;
    for (int i = 0; i < steps; ++i) {
      const double t = prev + (b - prev) * (i + 0.5) / steps;
      area += std::exp(-0.5 * c * t * t) * (b - prev) / steps;
    }
    prev = b;
    const double p_fail = 1.0 - std::exp(-0.5 * c * b * b);
    const double rate = (area + p_fail) / (fixed_cost + b * (1.0 + 0.35 * p_fail));
    if (rate > best_rate) {
      best_rate = rate;
      best = (uint32_t)b;
    }
  }
  batch_reads_ = best;
}

int
Classifier::batch_round(uint32_t& pos)
{
  const uint32_t n = rg_.n;
  const uint32_t tile = p_.tile_length, block = p_.block_size;
  static const uint32_t max_batch = [] { // developer hook
    const char* e = getenv("GRP_BATCH_MAX");
    return e ? (uint32_t)std::max(2l, atol(e)) : 4096u;
  }();
  uint32_t B = clamp_tiles(pos, std::min<uint32_t>({ batch_reads_, n - pos, p_.max_window, max_batch }), kMaxWindowTiles);
  // The second query of the previous batch ran on past its window (round 3): the reads behind
  // it were decided in the same launch, against the state that batch left — this window's first
  // decisions, if nothing has touched the filter since (the batch was confirmed in full).
  const bool have_first = bnext_valid_ && bnext_base_ == base_ && bnext_pos_ == pos && bnext_inserts_ == n_inserts_ && bnext_path_ == curr_path_ && !bnext_.empty();
  bnext_valid_ = false;
  if (have_first) {
    B = std::min<uint32_t>(B, (uint32_t)bnext_.size());
  }
  const grp_decide_params dp{ p_.threshold, p_.unassigned_min, p_.assigned_max, 0 };
  auto fail = [&](const char* what, int rc) {
    err_ = std::string(what) + ": " + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
    return rc;
  };
  // Where most reads insert, a read that overlaps ANY read in front of it in its window will decide differently
  // behind that read's insert and take the rest of the batch back.  The engine tells for blocks of 4096 reads which
  // reads overlap which read in front of them (a hash-only pass, grp_window_overlap: one call per block, nothing per
  // batch) and a window ends in front of the first read that overlaps one of ITS reads — it becomes the next
  // window's first read.  Only a hint; one rank (the ranks would have to agree on it).
  const bool ask_overlap = vt_.window_overlap && env_.overlap_samples && p_.world == 1 && p_insert_mid_ >= env_.overlap_min_insert;
  auto ends_at_overlap = [&](uint32_t at, uint32_t count) -> uint32_t { // reads [at, at + count) of the range: how many to keep
    if (!ask_overlap || count < 3) {
      return count;
    }
    if (ovl_base_ != base_ || at < ovl_lo_ || at >= ovl_hi_ || (at + count > ovl_hi_ && ovl_hi_ < n)) {
      // the block that starts at this window (overlaps with reads in front of `at` do not matter to it); a window
      // that would reach beyond the block's end begins the next block (its tail would go unexamined)
      constexpr uint32_t kBlock = 4096;
      const uint32_t nb = clamp_tiles(at, std::min<uint32_t>(kBlock, n - at), 1ull << 19);
      ovl_prev_.assign(nb, UINT32_MAX);
      ovl_base_ = base_;
      ovl_lo_ = at;
      ovl_hi_ = at + nb;
      ++n_overlap_calls_;
      // How many shared samples end a window: 4 (of the one frame in 32 the engine samples; rounds 4 - 5: 8 of one in 16)
      // tells every true overlap of a uniform genome (an error-free kilobase is ~30 samples, unrelated reads share ~0.005) — and, on a genome with repeats, nearly every read: the windows
      // end after ~50 reads and the batches' fixed cost (worth ~70 reads) is what the run pays (bench.py --repeat-frac
      // 0.4, measured with one frame in 16: 75 k reads/s at 8, 103 k at 64; uniform genome: 127 k at 8, 123 k at 64).  So the threshold climbs the
      // block-by-block efficiency — reads committed per (70 x batches + reads queried), from the classifier's own
      // counters: deterministic — while the batches are short; where they are long already (>= 128 reads at the base
      // threshold) it stays.  A hint either way: every batch is still confirmed by its second decisions.
      if (!env_.overlap_fixed) {
        if (ovl_thr_ == 0) {
          ovl_thr_ = env_.overlap_samples;
          ovl_s_batches_ = n_batches_;
          ovl_s_reads_ = n_batch_reads_;
          ovl_s_queried_ = n_queried_;
        } else {
          const uint64_t db = n_batches_ - ovl_s_batches_, dr = n_batch_reads_ - ovl_s_reads_, dq = n_queried_ - ovl_s_queried_;
          if (db >= 8) {
            const double eff = (double)dr / (70.0 * (double)db + (double)dq);
            if (!(ovl_thr_ == env_.overlap_samples && dr >= 128 * db)) {
              if (eff < ovl_last_eff_) {
                ovl_dir_ = -ovl_dir_;
              }
              uint32_t next = ovl_dir_ > 0 ? ovl_thr_ * 2u : ovl_thr_ / 2u;
              if (next < env_.overlap_samples) {
                next = env_.overlap_samples;
                ovl_dir_ = 1;
              } else if (next > 32u * env_.overlap_samples) {
                next = 32u * env_.overlap_samples;
                ovl_dir_ = -1;
              }
              ovl_thr_ = next;
            }
            ovl_last_eff_ = eff;
            // (the snapshots move only when a block of >= 8 batches has been judged — short hint windows
            // accumulate until there are 8, instead of never adapting)
            ovl_s_batches_ = n_batches_;
            ovl_s_reads_ = n_batch_reads_;
            ovl_s_queried_ = n_queried_;
          }
        }
      }
      const uint32_t threshold = env_.overlap_fixed || ovl_thr_ == 0 ? env_.overlap_samples : ovl_thr_;
      if (nb >= 3 && vt_.window_overlap(ctx_, rg_.reads, base_ + at, nb, threshold, ovl_prev_.data()) != GRP_OK) {
        ovl_prev_.assign(nb, UINT32_MAX); // no hints from this engine call
      }
    }
    const uint32_t end = std::min(at + count, ovl_hi_);
    for (uint32_t j = at + 1; j < end; ++j) {
      const uint32_t p = ovl_prev_[j - ovl_lo_];
      if (p != UINT32_MAX && ovl_lo_ + p >= at) {
        ++n_batch_overlap_cuts_;
        return j - at;
      }
    }
    return count;
  };
  if (!have_first) {
    B = ends_at_overlap(pos, B);
  }
  in_batch_ = true;
  ++n_windows_;
  // Several ranks (every one holds a replica and applies the whole batch to it): the two queries
  // of a batch are striped over the ranks like any other window, the 32-byte decisions
  // all-gathered; small batches are not worth the two exchanges.
  static const uint32_t stripe_min = [] { // developer hook / tests: reads per rank from which a batch's queries are striped
    const char* e = getenv("GRP_BATCH_STRIPE_MIN");
    return e ? (uint32_t)std::max(1l, atol(e)) : 32u;
  }();
  const uint32_t world = p_.world;
  // reads [lo, lo + my) of a window of `count` reads are this rank's; q = reads per rank
  auto my_stripe = [&](uint32_t count, uint32_t& q, uint32_t& lo, uint32_t& my) {
    q = (count + world - 1) / world;
    lo = std::min<uint64_t>((uint64_t)p_.rank * q, count);
    my = std::min<uint32_t>(q, count - lo);
  };
  bdec0_.resize(B);
  int rc = GRP_OK;
  const bool striped = world > 1 && allgather_cb_ && B >= stripe_min * world;
  if (have_first) {
    std::copy(bnext_.begin(), bnext_.begin() + B, bdec0_.begin());
    ++n_batch_fused_;
  } else if (striped) {
    uint32_t q, lo, my;
    my_stripe(B, q, lo, my);
    dec_.assign(q, gr_read_decision{});
    if (my) {
      rc = vt_.classify_reads(ctx_, rg_.reads, base_ + pos + lo, my, &dp, dec_.data());
    }
    if (rc != GRP_OK) {
      return fail("classify_reads", rc);
    }
    n_queried_ += my;
    rc = gather_decisions(q);
    if (rc != GRP_OK) {
      return rc;
    }
    --n_windows_; // counted above
    std::copy(dec_all_.begin(), dec_all_.begin() + B, bdec0_.begin()); // rank r's block starts at read r * q
  } else {
    rc = vt_.classify_reads(ctx_, rg_.reads, base_ + pos, B, &dp, bdec0_.data());
    if (rc != GRP_OK) {
      return fail("classify_reads", rc);
    }
    n_queried_ += B;
  }

  // the inserts these decisions ask for, with the IDs commit() will allocate
  bins_.clear();
  bfloor_.assign(B, 0);
  bfirst_.assign(B, 0);
  uint32_t ids = ids_inserted_;
  uint64_t bases = inserted_bases_;
  uint32_t cnt = B, first_ins = UINT32_MAX;
  // the last insert so far was a trimmed read whose last ID block carries the next first ID (:1048-1049, :1074; block == 1 only)
  bool shared_id = last_insert_shares_id_;
  // a batch holds 2^26 (frame, seed) records (frames in units of 256 per tile): the window ends
  // in front of the insert that would not fit (the engine refuses larger batches, GRP_ERR_NOMEM)
  const uint64_t rec_per_tile = (uint64_t)((tile + 255u) / 256u) * 256u * p_.hash_num;
  uint64_t rec = 0;
  for (uint32_t j = 0; j < B; ++j) {
    const gr_read_decision& d = bdec0_[j];
    const uint32_t len = rg_.lens[pos + j];
    bfloor_[j] = (ids + 1) | (shared_id ? 0x80000000u : 0u);
    bool ins = false;
    uint32_t ts = 0, te_excl = 0, off = 0, next_ids = ids;
    if (d.kind == DEC_INSERT_WHOLE) {
      ins = true;
      te_excl = d.num_tiles;
      next_ids = ids + 1 + (uint32_t)(len / ((size_t)tile * block));
      bases += len;
    } else if (d.kind == DEC_INSERT_TRIMMED) {
      ins = true;
      ts = d.trim_start;
      te_excl = d.trim_end + 1;
      off = 1;
      next_ids = ids + 1 + (d.trim_end - d.trim_start) / block;
      uint64_t n_out = len - (uint64_t)ts * tile;
      if (d.trim_end != d.num_tiles - 1) {
        n_out = std::min<uint64_t>(n_out, (uint64_t)(te_excl - ts) * tile);
      }
      bases += n_out;
    } else if (d.kind != DEC_ASSIGNED_ALL && d.kind != DEC_ASSIGNED) {
      cnt = j; // a read the engine hands back to the host's decision: the classic path takes it
      break;
    }
    if (ins) {
      if ((te_excl - ts + block - 1) / block > 256 || te_excl <= ts) {
        cnt = j; // more ID blocks than a batch entry holds
        break;
      }
      rec += (uint64_t)(te_excl - ts) * rec_per_tile;
      if (rec > (60ull << 20) && !bins_.empty()) {
        cnt = j;
        break;
      }
      bins_.push_back(grp_batch_insert{ base_ + pos + j, ts, te_excl, ids + 1, off });
      bfirst_[j] = ids + 1;
      ids = next_ids;
      shared_id = off == 1 && block == 1; // see commit(): only ID blocks of one tile share the next insert's first ID
      if (first_ins == UINT32_MAX) {
        first_ins = j;
      }
      if (p_.silver_path && p_.target_bases < bases) {
        cnt = j + 1; // the silver path rolls over behind this read: the ID array is reset there
        break;
      }
    }
  }
  if (cnt == 0) {
    batch_bypass_ = true;
    return GRP_OK;
  }
  // up to and including the first insert the decisions ARE the serial loop's
  auto commit_classic = [&](uint32_t upto) {
    uint32_t j = 0;
    bool stale = false;
    while (j < upto && !stale && !finished_ && rc == GRP_OK) {
      stale = commit_one(pos + j, bdec0_[j], rc);
      ++j;
    }
    pos += j;
    return rc;
  };
  if (bins_.size() <= 1) {
    return commit_classic(bins_.empty() ? cnt : first_ins + 1);
  }
  rc = vt_.batch_insert(ctx_, rg_.reads, bins_.data(), (uint32_t)bins_.size(), block, base_ + pos);
  if (rc == GRP_ERR_NOMEM) {
    batch_reads_ = std::max<uint32_t>(2, cnt / 2); // more records than a batch holds
    ++n_batch_refused_;
    rc = GRP_OK;
    return commit_classic(first_ins + 1);
  }
  if (rc != GRP_OK) {
    return fail("batch_insert", rc);
  }
  // One rank: the second query runs on past the window — the reads behind it are decided in the
  // same launch (their floor is the largest ID: they see the filter as it is, with this batch's
  // inserts) and become the next window's first decisions when this batch is confirmed in full.
  // The launch is twice as large (better filled), and a confirmed batch costs one round trip less.
  static const bool fuse_off = [] {
    const char* e = getenv("GRP_BATCH_FUSE");
    return e && !strcmp(e, "off");
  }();
  uint32_t extra = 0;
  if (!striped && world == 1 && !fuse_off && cnt == B && pos + cnt < n) {
    extra = clamp_tiles(pos, std::min<uint32_t>({ batch_reads_, n - pos - cnt, p_.max_window, max_batch }) + cnt, kMaxWindowTiles);
    extra = extra > cnt ? extra - cnt : 0;
    // worth it while the query time thrown away with a batch that ends early (probability from the
    // decayed count of such batches, ~50 batches of memory) stays below the round trip saved (~0.1 ms)
    const double p_fail = std::min(1.0, bf_fail_ / 50.0);
    const double t_query = (double)extra * avg_probes_per_read_ / 40e9;
    if (p_fail * t_query > 1.0e-4) {
      extra = 0;
    }
    if (extra) {
      extra = ends_at_overlap(pos + cnt, extra); // the reads behind the batch are the next window
    }
  }
  bdec1_.resize((size_t)cnt + extra);
  if (extra) {
    bfloor_.resize((size_t)cnt + extra);
    std::fill(bfloor_.begin() + cnt, bfloor_.end(), 0x7FFFFFFFu);
  }
  uint32_t queried = cnt + extra;
  if (striped) {
    uint32_t q, lo, my;
    my_stripe(cnt, q, lo, my);
    dec_.assign(q, gr_read_decision{});
    // (also with an empty stripe: the call tells every rank alike whether the batch was refused)
    rc = vt_.batch_classify(ctx_, rg_.reads, base_ + pos + lo, my, &dp, bfloor_.data() + lo, dec_.data());
    queried = my;
    if (rc == GRP_OK) {
      const int grc = gather_decisions(q);
      --n_windows_;
      if (grc != GRP_OK) {
        (void)vt_.batch_undo(ctx_, base_ + pos, bfloor_[0] & 0x7FFFFFFFu);
        return grc;
      }
      std::copy(dec_all_.begin(), dec_all_.begin() + cnt, bdec1_.begin());
    }
  } else if (vt_.batch_verify) {
    // round 4: the inserted tiles are patched from the batch's own records, only the tiles the batch did
    // not insert are queried again; the reads behind the batch take the plain query (include/grpath.h)
    rc = vt_.batch_verify(ctx_, rg_.reads, base_ + pos, cnt, extra, &dp, bfloor_.data(), bdec1_.data());
  } else {
    rc = vt_.batch_classify(ctx_, rg_.reads, base_ + pos, cnt + extra, &dp, bfloor_.data(), bdec1_.data());
  }
  if (rc == GRP_ERR_NOMEM) {
    // found on the device: the window's reads share too many ranks (they overlap each other);
    // nothing was inserted, the batch is over — a smaller one next time
    batch_reads_ = std::max<uint32_t>(2, cnt / 2);
    ++n_batch_refused_;
    rc = GRP_OK;
    return commit_classic(first_ins + 1);
  }
  if (rc != GRP_OK) {
    (void)vt_.batch_undo(ctx_, base_ + pos, bfloor_[0] & 0x7FFFFFFFu);
    return fail("batch_classify", rc);
  }
  n_queried_ += queried;
  ++n_batches_;
  uint32_t bad = cnt;
  bool bad_undecided = false; // the engine handed the read back in its second decision: the classic path takes it
  for (uint32_t j = 0; j < cnt; ++j) {
    const gr_read_decision &a = bdec0_[j], &b = bdec1_[j];
    if (a.kind != b.kind || a.num_tiles != b.num_tiles || (a.kind == DEC_INSERT_TRIMMED && (a.trim_start != b.trim_start || a.trim_end != b.trim_end))) {
      bad = j;
      bad_undecided = b.kind != DEC_INSERT_WHOLE && b.kind != DEC_INSERT_TRIMMED && b.kind != DEC_ASSIGNED_ALL && b.kind != DEC_ASSIGNED;
      break;
    }
  }
  const uint32_t confirmed = bad;
  if (bad == cnt) {
    rc = vt_.batch_end(ctx_);
    if (rc != GRP_OK) {
      return fail("batch_end", rc);
    }
  } else {
    // from read `bad` on the batch was not the serial loop: its insert and the ones behind it are
    // taken back, the confirmed ones in front of it stay
    ++n_batch_undone_;
    rc = vt_.batch_undo(ctx_, base_ + pos + bad, bfloor_[bad] & 0x7FFFFFFFu);
    if (rc != GRP_OK) {
      return fail("batch_undo", rc);
    }
  }
  {
    // (inserted read, later read) pairs that had the chance to differ
    double exposure = 0.0;
    uint32_t seen = 0, k = 0;
    const uint32_t upto = bad < cnt ? bad + 1 : cnt;
    for (uint32_t j = 0; j < upto; ++j) {
      exposure += seen;
      if (k < bins_.size() && bins_[k].read == base_ + pos + j) {
        ++seen;
        ++k;
      }
    }
    batch_feedback(cnt, bad, exposure);
  }
  const uint64_t path_at_query = curr_path_; // a rollover while committing resets the ID array: the decisions behind the window would be stale
  // the second decisions are the records (hits / misses against the state in front of each read)
  for (uint32_t j = 0; j < confirmed && rc == GRP_OK && !finished_; ++j) {
    (void)commit_one(pos + j, bdec1_[j], rc, bfirst_[j] != 0, bfirst_[j]);
  }
  if (rc != GRP_OK) {
    return rc;
  }
  n_batch_reads_ += confirmed;
  pos += confirmed;
  if (extra && bad == cnt && !finished_ && curr_path_ == path_at_query) {
    bnext_.assign(bdec1_.begin() + cnt, bdec1_.end());
    bnext_valid_ = true;
    bnext_base_ = base_;
    bnext_pos_ = pos;
    bnext_inserts_ = n_inserts_;
    bnext_path_ = curr_path_;
  }
  if (bad != cnt && !finished_) {
    if (bad_undecided) {
      batch_bypass_ = true;
    } else {
      // `bad` itself: its second decision was taken against exactly the state it now sees
      (void)commit_one(pos, bdec1_[bad], rc);
      ++pos;
    }
  }
  return rc;
}

// ---- streaming window: consume the decisions while the launch is running ----------
int
Classifier::stream_round(uint32_t& pos)
{
  const uint32_t n = rg_.n;
  int rc = GRP_OK;
  const auto tr_enter = std::chrono::steady_clock::now();
  if (snext_.active) {
    scur_ = snext_; // starts at pos: an insert would have aborted it
    snext_.active = false;
  } else {
    const Plan plan = window_plan();
    rc = launch_stream(rg_.reads, pos, std::min<uint32_t>(plan.S, n - pos), 0, scur_);
    if (rc != GRP_OK) {
      return rc;
    }
  }
  const uint32_t S = scur_.S;
  group_base_ = UINT32_MAX;
  group_from_ = 0;
  ins_lost_ = false;
  ins_unconfirmed_ = false;
  uint32_t j = 0;
  bool stale = false, redo = false, lost = false, next_refused = false;
  // developer hook: where the time of an insert goes (launch call, first record, drain + insert)
  static const bool trace_abort = getenv("GRP_TRACE_ABORT") != nullptr;
  static double t_launch = 0, t_first = 0, t_commit_ins = 0, t_drop = 0, t_resume = 0;
  static uint64_t n_rounds = 0, n_stale = 0, n_resumed = 0;
  std::chrono::steady_clock::time_point tr_resume0{};
  bool tr_resume_pending = false;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const auto tr_after_launch = now();
  if (trace_abort) {
    t_launch += secs(tr_enter, tr_after_launch);
    ++n_rounds;
  }
  while (j < S) {
    // the next window goes in shortly before this launch runs out of work: early
    // enough to start back to back (the GPU is at most ~100 reads ahead of the host),
    // late enough that an insert rarely has to abort it
    if (!snext_.active && !next_refused && j + 256 * p_.world >= S && pos + S < n) {
      const Plan plan = window_plan();
      if (plan.streaming) { // queued right behind the current launch
        rc = launch_stream(rg_.reads, pos + S, std::min<uint32_t>(plan.S, n - pos - S), scur_.slot ^ 1u, snext_);
        if (rc == GRP_ERR_BUSY) {
          // the engine would have to wait for the device (buffers to grow) while the current window
          // may be waiting for this thread: the next window is begun when this one has ended
          rc = GRP_OK;
          next_refused = true;
        }
        if (rc != GRP_OK) {
          break;
        }
      }
    }
    gr_read_decision d;
    rc = stream_decision(j, d);
    if (rc == STREAM_LOST) {
      rc = GRP_OK;
      lost = true;
      break;
    }
    if (rc != GRP_OK) {
      break;
    }
    if (trace_abort && j == 0) {
      t_first += secs(tr_after_launch, now());
    }
    if (trace_abort && tr_resume_pending) { // the first record behind an insert the launch applied itself
      tr_resume_pending = false;
      t_resume += secs(tr_resume0, now());
      static const bool each = getenv("GRP_TRACE_ABORT") && getenv("GRP_TRACE_ABORT")[0] == '2';
      if (each) {
        fprintf(stderr, "insert behind read %u of a window of %u (launched %.1f us ago): %.1f us to the first record behind it\n", j - 1, S, 1e6 * secs(tr_after_launch, tr_resume0), 1e6 * secs(tr_resume0, now()));
      }
      if ((++n_resumed & 511u) == 0) {
        fprintf(stderr, "in-launch inserts %llu: %.1f us from the insert record to the first record behind it\n", (unsigned long long)n_resumed, 1e6 * t_resume / n_resumed);
      }
    }
    d.pad = 0;
    // (1 / 1024: a record handed back is priced as what it is over the last ~1000 reads.  With 1 / 64 a SINGLE read of
    // more than 256 tiles — one in 70 000 with a realistic ONT length tail — made the streaming windows look 2.6 x as
    // expensive as they are and the plan left them for ~45 synchronous windows: 146 such reads cost the C2 stream 5 s,
    // bench.py --len-sigma 0.6, profiles/r04_long_tail.txt)
    p_redo_ += (1.0 / 1024.0) * ((d.kind == 0 ? 1.0 : 0.0) - p_redo_);
    n_stream_handbacks_ += d.kind == 0 ? 1 : 0;
    if (d.kind == 0) {
      redo = true; // needs the worst-case table / a larger arena: synchronous path below
      break;
    }
    bool resume = false;
    if (d.kind == DEC_INSERT_WHOLE || d.kind == DEC_INSERT_TRIMMED) {
      // One rank: the parked launch applies the insert itself and carries on behind the read
      // (commit() hands it over).  Otherwise everything behind this read is
      // stale: stop the launches before the insert is queued.
      resume = scur_.resumable;
      if (resume) {
        stream_ins_ = &scur_;
        stream_ins_ok_ = false;
      } else {
        (void)vt_.stream_abort(ctx_, scur_.slot);
        if (snext_.active) {
          (void)vt_.stream_abort(ctx_, snext_.slot);
        }
        stale = true;
      }
    }
    const auto tr_c0 = now();
    commit_one(pos + j, d, rc);
    if (resume) {
      stream_ins_ = nullptr;
      stale = !stream_ins_ok_; // the engine refused (the launches were aborted, the insert queued behind them)
      if (!stale && p_.world > 1) {
        // the launches carry on behind the read under the next generation: the rest of its stripe group is exchanged again
        group_base_ = UINT32_MAX;
        group_from_ = j + 1;
        ins_unconfirmed_ = true;
        // This rank's launch may have left before the command came (its own stripes were decided; the read is another
        // rank's): then the window queued behind it must not go on unnoticed on a replica without the insert.  The launch
        // answers within the time the owner's launch takes to apply it (the ranks wait for its next record anyway).
        for (;;) {
          const int st = vt_.stream_insert_done(ctx_, scur_.slot);
          if (st == 1) {
            break;
          }
          if (st == 2) {
            ins_lost_ = true; // said in the next exchange: every rank ends the round there (end_stream applies the insert the classic way)
            break;
          }
          if (st < 0) {
            rc = st;
            err_ = std::string("stream_insert_done: ") + (vt_.last_error ? vt_.last_error(ctx_) : "failed");
            break;
          }
          __builtin_ia32_pause();
        }
      }
      if (trace_abort && !stale) {
        tr_resume0 = now();
        tr_resume_pending = true;
      }
    }
    if (trace_abort && stale) {
      t_commit_ins += secs(tr_c0, now());
    }
    if (rc != GRP_OK) {
      break;
    }
    ++j;
    if (stale || finished_) {
      break;
    }
  }
  pos += j;
  if (rc == GRP_OK && p_.world > 1 && ins_unconfirmed_ && !lost) {
    // the window's last record was an insert the launches took: one more exchange, of the headers alone — a rank whose
    // launch had left before the command must not begin the next window by itself while the others carry on with theirs
    gr_read_decision mine{};
    mine.kind = ins_lost_ ? 1u : 0u;
    std::vector<gr_read_decision> all(p_.world);
    if (allgather_cb_(ag_user_, &mine, sizeof(mine), all.data()) != 0) {
      err_ = "allgather callback failed";
      rc = GRP_ERR_INVALID;
    }
    for (uint32_t q = 0; q < p_.world && rc == GRP_OK; ++q) {
      lost = lost || all[q].kind != 0;
    }
    ins_unconfirmed_ = false;
  }
  if (rc != GRP_OK || stale || finished_ || redo || lost) {
    const auto tr_d0 = now();
    const int drc = drop_streams();
    if (rc == GRP_OK) {
      rc = drc;
    }
    if (trace_abort) {
      t_drop += secs(tr_d0, now());
      ++n_stale;
      if ((n_stale & 1023u) == 0) {
        fprintf(stderr, "stream rounds %llu, ended by an insert %llu: per round launch call %.1f us, first record %.1f us; per insert commit+insert call %.1f us, drop (drain + insert kernels) %.1f us\n",
                (unsigned long long)n_rounds, (unsigned long long)n_stale, 1e6 * t_launch / n_rounds, 1e6 * t_first / n_rounds, 1e6 * t_commit_ins / n_stale, 1e6 * t_drop / n_stale);
      }
    }
  } else {
    if (p_.world > 1) {
      (void)vt_.stream_abort(ctx_, scur_.slot); // every read of the window is committed: a striped window that takes inserts stays until it is told
    }
    rc = end_stream(scur_); // completed: returns at once
    if (rc != GRP_OK) {
      drop_streams();
    }
  }
  if (rc == GRP_OK && redo && !finished_) {
    // one read through the synchronous path (it redoes flagged tiles / grows the arena)
    rc = query_window(rg_.reads, rg_.lens, pos, 1);
    if (rc == GRP_OK) {
      commit_one(pos, dec_all_[0], rc);
      ++pos;
    }
  }
  return rc;
}

// ---- synchronous / pipelined window: all decisions of the window, then the commit -----
int
Classifier::window_round(uint32_t& pos)
{
  const uint32_t n = rg_.n;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = GRP_OK;
  Flight cur;
  uint32_t S;
  if (next_.active) {
    cur = next_; // starts at pos: an insert would have abandoned it
    next_.active = false;
    S = cur.S;
  } else {
    const Plan plan = window_plan();
    S = clamp_tiles(pos, std::min<uint32_t>(plan.S, n - pos), kMaxWindowTiles);
    debug_window_pos_ = pos;
    rc = plan.pipelined ? launch_window(rg_.reads, pos, S, 0, cur) : query_window(rg_.reads, rg_.lens, pos, S);
    if (rc != GRP_OK) {
      return rc;
    }
  }
  if (cur.active) {
    if (pos + S < n) {
      const Plan plan = window_plan();
      const uint32_t S2 = clamp_tiles(pos + S, std::min<uint32_t>(plan.S, n - pos - S), kMaxWindowTiles);
      if (plan.pipelined && (S2 >= 16 * p_.world || S2 == plan.S)) {
        rc = launch_window(rg_.reads, pos + S, S2, cur.slot ^ 1u, next_);
      }
    }
    if (rc == GRP_OK) {
      rc = finish_window(cur);
    }
    if (rc != GRP_OK) {
      abandon_window(next_);
      return rc;
    }
  }
  const auto t1 = std::chrono::steady_clock::now();
  t_windows_ += std::chrono::duration<double>(t1 - t0).count();
  uint32_t j = 0;
  bool stale = false;
  while (j < S && !stale && !finished_) {
    stale = commit_one(pos + j, dec_all_[j], rc); // an insert changes the miBF: later speculative results are stale
    if (rc != GRP_OK) {
      abandon_window(next_);
      return rc;
    }
    ++j;
  }
  if (stale || finished_) {
    abandon_window(next_);
  }
  pos += j;
  t_commit_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  return GRP_OK;
}

int
Classifier::run(void* reads, const uint32_t* lens, uint32_t first, uint32_t n, const uint32_t* skipped_before, uint32_t skipped_after, bool& finished)
{
  finished = finished_;
  if (finished_) {
    return GRP_OK;
  }
  base_ = first;
  lost_at_ = UINT64_MAX;  // (reads are numbered from the range's own first: an index of the range before means nothing here)
  ovl_base_ = UINT64_MAX; // (the overlap hints of the range before are another batch's)
  rg_.reads = reads;
  rg_.lens = lens + first;
  rg_.skipped_before = skipped_before ? skipped_before + first : nullptr;
  rg_.n = n;
  tile0_.resize((size_t)n + 1);
  tile0_[0] = 0;
  for (uint32_t i = 0; i < n; ++i) {
    tile0_[i + 1] = tile0_[i] + rg_.lens[i] / p_.tile_length;
  }
  if (n) {
    // frames x seeds per read of this range (every tile has ~tile_length frames)
    avg_probes_per_read_ = (double)tile0_[n] * p_.tile_length * p_.hash_num / n;
  }
  uint32_t pos = 0;
  int rc = GRP_OK;
  while (rc == GRP_OK && pos < n && !finished_) {
    if (!next_.active && !snext_.active && want_batch()) {
      const auto t0 = std::chrono::steady_clock::now();
      rc = batch_round(pos);
      t_windows_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      continue;
    }
    in_batch_ = false;
    batch_bypass_ = false;
    if (!next_.active && (snext_.active || window_plan().streaming)) {
      const auto t0 = std::chrono::steady_clock::now();
      rc = stream_round(pos);
      t_windows_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } else {
      rc = window_round(pos);
    }
  }
  abandon_window(next_);
  {
    const int drc = drop_streams();
    if (rc == GRP_OK) {
      rc = drc;
    }
  }
  if (rc != GRP_OK) {
    return rc;
  }
  if (!finished_) {
    skip_reads(skipped_after);
  }
  finished = finished_;
  return GRP_OK;
}

} // namespace gr
