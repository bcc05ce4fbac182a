// Order-exact read classification on top of the engine ABI.
//
// The reference classifies reads strictly one after the other
// (goldrush_path.cpp:1229-1256): read N is queried after the inserts of every
// accepted read < N.  Here a *window* of consecutive reads is queried
// speculatively in one kernel launch, the decisions are committed in file
// order, and as soon as one read inserts (which changes the miBF) the rest of
// the window is discarded and queried again.  The window size adapts to the
// observed insert rate, so the result is identical to the serial loop while
// the insert-free stretches run as large GPU batches.  With an engine that has
// classify_begin / classify_end the window after the current one is already on
// the GPU while the host commits (two slots); an insert abandons it.
#pragma once
#include "../../../include/grpath_host.h"
#include "gr_tiles.hpp"

#include <string>
#include <vector>

namespace gr {

class Classifier
{
public:
  Classifier(const gr_classifier_params& p, const grp_engine_vt& vt, void* ctx);
  void set_callbacks(gr_commit_fn commit, gr_rollover_fn rollover, gr_allgather_fn allgather, void* user);
  void set_allgather(gr_allgather_fn allgather, void* allgather_user);
  void set_debug(gr_debug_fn fn) { debug_cb_ = fn; }
  // commits of the reads [first0, first0 + count0) and [first1, first1 + count1) (numbers in the batch of reads) are kept
  void keep_commits(uint32_t first0, uint32_t count0, uint32_t first1, uint32_t count1);
  const std::vector<gr_commit>& kept_commits() const { return kept_; }
  // classifies reads [first, first+n) of the batch; lens / skipped_before are indexed by absolute read number
  int run(void* reads, const uint32_t* lens, uint32_t first, uint32_t n, const uint32_t* skipped_before, uint32_t skipped_after, bool& finished);
  void get_state(gr_classifier_state& s) const;
  // tail of main(): verbose per-path statistics (goldrush_path.cpp:1266-1270)
  void log_path_stat() const;
  const std::string& error() const { return err_; }
  uint64_t curr_path() const { return curr_path_; }
  bool finished() const { return finished_; }

private:
  int query_window(void* reads, const uint32_t* lens, uint32_t first, uint32_t count);
  // engine_inserted: the engine has applied the read's ID blocks already (a window committed as a batch: grp_batch_insert_reads)
  bool commit(void* reads, const uint32_t* lens, uint32_t r, const gr_read_decision& d, int& rc, bool engine_inserted = false, uint32_t engine_first_id = 0);
  void silver_path_check(int& rc);
  void skip_reads(uint32_t n);
  void bump_id();
  struct Plan
  {
    uint32_t S;     // reads in the next window
    bool pipelined; // worth keeping a second window in flight
    bool streaming; // one long launch, decisions consumed while it runs (single rank)
  };
  Plan window_plan() const;
  struct Flight
  {
    bool active = false;
    uint32_t pos = 0, S = 0, slot = 0, q = 0, my_count = 0;
  };
  int launch_window(void* reads, uint32_t pos, uint32_t S, uint32_t slot, Flight& f);
  int finish_window(Flight& f);
  void abandon_window(Flight& f);
  int gather_decisions(uint32_t q);
  struct StreamFlight
  {
    bool active = false;
    uint32_t pos = 0, S = 0, slot = 0;
    const gr_read_decision* dec = nullptr;
    bool resumable = false;   // begun with stream_begin_resumable: it waits where it parks, for stream_abort or stream_insert
    uint32_t gen = 1;         // generation (.pad) of the records that count: +1 per insert the window applied itself
    bool ins_posted = false;  // an insert was handed to the launch (stream_insert); its arguments, should the launch end without it:
    uint32_t ins_read = 0, ins_ts = 0, ins_te = 0, ins_first_id = 0, ins_off = 0;
  };
  uint32_t clamp_tiles(uint32_t pos, uint32_t S, uint64_t max_tiles) const;
  int launch_stream(void* reads, uint32_t pos, uint32_t S, uint32_t slot, StreamFlight& f);
  int end_stream(StreamFlight& f);
  int wait_record(const StreamFlight& f, uint32_t j);
  bool can_stream() const;
  bool can_resume() const;
  bool commit_one(uint32_t r, const gr_read_decision& d, int& rc, bool engine_inserted = false, uint32_t engine_first_id = 0);
  bool can_batch() const;
  bool want_batch() const; // the insert rate calls for windows committed as batches
  int batch_round(uint32_t& pos);
  void batch_feedback(uint32_t reads, uint32_t bad, double exposure);
  int drop_streams();
  int stream_decision(uint32_t j, gr_read_decision& d);
  int stream_round(uint32_t& pos);
  int window_round(uint32_t& pos);
  uint32_t stripe_reads() const; // reads per stripe of a window shared by several ranks (0: one rank)

  gr_classifier_params p_;
  grp_engine_vt vt_;
  void* ctx_;
  struct
  {
    std::string pipeline, stream, batch; // GRP_PIPELINE / GRP_STREAM / GRP_BATCH as found when the classifier was created
    uint64_t max_window_tiles = 0;             // GRP_MAX_WINDOW_TILES (0: unset)
    bool overlap_fixed = false;                // GRP_BATCH_OVERLAP was given: the threshold stays what it says (else it adapts, batch_round)
    uint32_t overlap_samples = 4;              // GRP_BATCH_OVERLAP=<n> / off: windows of batches end in front of a read sharing >= n sampled k-mers with a read in front of it (0: not asked)
    double overlap_min_insert = 0.3;           // GRP_BATCH_OVERLAP_P: ... where at least this share of the reads inserts
  } env_;
  gr_commit_fn commit_cb_ = nullptr;
  uint32_t keep_first_[2] = { 0, 0 }, keep_count_[2] = { 0, 0 };
  std::vector<gr_commit> kept_;
  double emit_commit(const gr_commit& ev);
  gr_rollover_fn rollover_cb_ = nullptr;
  gr_allgather_fn allgather_cb_ = nullptr;
  void* user_ = nullptr;
  void* ag_user_ = nullptr;
  gr_debug_fn debug_cb_ = nullptr;
  std::vector<std::string> debug_text_; // --debug: the tile-state dumps of the window's reads
  uint32_t debug_window_pos_ = 0;       // ... and the window's first read

  // main()'s loop state (goldrush_path.cpp:1222-1227) + log_info_struct (:41-51)
  uint64_t inserted_bases_ = 0;
  uint64_t curr_path_ = 1;
  uint32_t id_ = 1;
  uint32_t ids_inserted_ = 0;
  uint64_t valid_reads_ = 0, total_tiles_ = 0, assigned_tiles_ = 0, unassigned_tiles_ = 0;
  uint64_t queries_ = 0, hits_ = 0, misses_ = 0, num_reads_in_path_ = 0;
  double phred_sum_in_path_ = 0;
  bool finished_ = false;

  // speculation control / statistics
  double p_insert_ = 1.0;        // EMA over ~32 reads
  double p_insert_slow_ = 0.0;   // EMA over ~8192 reads
  double p_insert_mid_ = 1.0;    // EMA over ~256 reads: chooses between the batches and the windows
  bool last_insert_shares_id_ = false; // the last insert was a trimmed read whose last ID block carries the next insert's first ID
  bool in_batch_ = false;        // the last round was a batch (hysteresis)
  bool batch_bypass_ = false;    // the read in front cannot be part of a batch: one classic round
  uint32_t batch_reads_ = 128;   // reads per batch, chosen by batch_feedback
  double bf_fail_ = 1.0, bf_expo_ = 5e4; // first reads that decided differently / (inserted read, later read) pairs exposed (decaying sums)
  uint64_t n_batches_ = 0, n_batch_undone_ = 0, n_batch_reads_ = 0, n_batch_refused_ = 0;
  double p_redo_ = 0.0;          // streaming records handed back to the synchronous path (EMA over ~64 reads)
  double avg_probes_per_read_ = 75000.0;
  uint64_t n_windows_ = 0, n_queried_ = 0, n_committed_ = 0, n_inserts_ = 0;
  double t_windows_ = 0, t_commit_ = 0;

  // the range being classified (run) and what is in flight on the engine
  struct Range
  {
    void* reads = nullptr;
    const uint32_t* lens = nullptr;           // indexed from the range's first read
    const uint32_t* skipped_before = nullptr; // same, or null
    uint32_t n = 0;
  } rg_;
  Flight next_;              // pipelined: the window after the current one
  StreamFlight scur_, snext_; // streaming: the current window and the one queued behind it
  StreamFlight* stream_ins_ = nullptr; // set while stream_round commits an insert record: commit() hands the insert to this window's launch
  bool stream_ins_ok_ = false;         // ... and the engine took it
  bool resume_disabled_ = false;       // a launch could not apply an insert itself (shared device): windows end at inserts again
  uint64_t n_stream_inserts_ = 0;
  uint32_t resume_clean_windows_ = 0;  // windows ended without a refused insert since resume_disabled_ was set (64: tried again)
  uint64_t n_stream_insert_fallbacks_ = 0, n_stream_relaunches_ = 0, n_stream_handbacks_ = 0;
  uint64_t n_batch_overlap_cuts_ = 0; // windows of batches ended in front of a read grp_window_overlap named
  uint64_t n_overlap_calls_ = 0;
  std::vector<uint32_t> ovl_prev_;    // grp_window_overlap's answer for reads [ovl_lo_, ovl_hi_) of the range that starts at base_ == ovl_base_
  uint32_t ovl_lo_ = 0, ovl_hi_ = 0;
  uint64_t ovl_base_ = UINT64_MAX;
  // the overlap threshold adapts (round 5): on a repeat-rich genome nearly every read shares a few sampled k-mers with a
  // read in front of it, the windows end after ~50 reads and the batches' fixed cost (~70 reads) is what the run pays
  uint32_t ovl_thr_ = 0;             // the threshold in use (0: not started, env_.overlap_samples)
  int ovl_dir_ = 1;                  // the way the last change went (x 2 / / 2)
  double ovl_last_eff_ = 0.0;        // committed reads per unit of cost over the block before
  uint64_t ovl_s_batches_ = 0, ovl_s_reads_ = 0, ovl_s_queried_ = 0; // counters at the last block's start
  uint64_t n_stream_rollovers_ = 0; // silver mode: inserts kept out of a parked launch because the path rolls over behind them
  uint64_t lost_at_ = UINT64_MAX;      // read at which a window was last begun again because its launch had left without deciding it
  uint32_t group_base_ = UINT32_MAX; // first read of the stripe group held in stripe_recv_
  bool ranks_resume_ok_ = true;      // several ranks: the last window begun was one every rank's launch applies inserts in (what the window plan prices an insert with: the same on every rank)
  bool ins_unconfirmed_ = false;     // several ranks: an insert went to the launches and no exchange has followed yet (the ranks have not told each other whether every launch took it)
  bool ins_lost_ = false;            // several ranks: this rank's launch had left before the insert command reached it — said in the next exchange, the ranks end the round together
  uint32_t group_from_ = 0;          // reads of the window below this one are committed: a group gathered again behind an insert the launches applied themselves starts here

  // scratch
  std::vector<uint64_t> tile0_; // tile prefix of the current range, relative to base_
  uint32_t base_ = 0;
  std::vector<grp_tile_summary> tiles_;
  std::vector<grp_id_count> lists_;
  std::vector<gr_read_decision> dec_, dec_all_, stripe_send_, stripe_recv_, bdec0_, bdec1_;
  std::vector<grp_batch_insert> bins_;
  // decisions of the reads behind the last batch, taken by its second query's launch (batch_round)
  std::vector<gr_read_decision> bnext_;
  bool bnext_valid_ = false;
  uint32_t bnext_base_ = 0, bnext_pos_ = 0;
  uint64_t bnext_inserts_ = 0, bnext_path_ = 0, n_batch_fused_ = 0;
  std::vector<uint32_t> bfloor_, bfirst_;
  std::vector<TileWorkspace> ws_;
  std::string err_;
};

} // namespace gr
