// libgrpath_hip.so — hand-written gfx950 (CDNA4) kernels + the C ABI of
// include/grpath.h.  Integer / bit-vector work, HBM-transaction bound: no MFMA.
// This file: context, launch logic and the exported functions; the kernels live in
// grp_kernels.inc (hot path), grp_ingest.inc (FASTQ), grp_ntcard.inc (--ntcard).
//
// Kernels
//   k_fill        spaced-seed ntHash of every read position + test-and-set of
//                 bit (hash % m)           (goldrush_path.cpp:304-305,
//                 MIBFConstructSupport.hpp:134-147)
//   k_bucket_*    popcount prefix scan -> 64-byte buckets {rank, bitmap, IDs}
//                 (MIBFConstructSupport.hpp:165-181)
//   k_query       fused hash -> probe (bit+rank, then ID) -> per-frame ID
//                 dedup -> per-tile LDS count table -> top ID + count>2 list
//                 (read_hashing.cpp:29-54, goldrush_path.cpp:544-626)
//   k_insert      hash -> rank -> exact per-call rank dedup -> reservoir rule
//                 (MIBFConstructSupport.hpp:247-283)
#include "grp_device.h"
#include "host/gr_tiles_core.hpp"

#include "../../include/grpath.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <chrono>
#include <cmath>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ---------------------------------------------------------------------------
// host-side context
// ---------------------------------------------------------------------------

namespace {

thread_local std::string g_create_error;

struct EventPair
{
  hipEvent_t a, b;
  int kind;
};

} // namespace


// per-window device / pinned buffers of the query + decision kernels
struct QuerySlot
{
  grp_tile_summary* d_tiles = nullptr;
  uint64_t d_tiles_cap = 0;
  grp_id_count* d_lists = nullptr;
  uint64_t d_lists_cap = 0;
  uint64_t* d_qctr = nullptr; // [3] list arena cursor, [4] flagged tiles (kept zero between calls)
  uint64_t* h_qctr = nullptr; // pinned: the first 64 bytes of h_dec_raw (one copy brings counters and decisions, slot_dec_alloc)
  uint32_t* d_flag_idx = nullptr;
  uint64_t d_flag_cap = 0;
  // decide scratch
  uint32_t* d_dec_ids = nullptr;
  uint8_t* d_dec_asg = nullptr;
  uint64_t* d_dec_scratch = nullptr;
  std::vector<uint32_t> h_long_reads; // reads of the window of more than 256 tiles: decided by a launch of their own (classify_enqueue_decide)
  uint32_t* d_long_reads = nullptr;
  uint64_t long_reads_cap = 0;
  uint64_t d_dec_cap = 0;
  grp_read_decision* d_dec = nullptr; // = d_dec_raw + 2: two entries (64 bytes) in front of the decisions carry the window's counters
  grp_read_decision* h_dec = nullptr; // = h_dec_raw + 2, pinned
  grp_read_decision* d_dec_raw = nullptr;
  grp_read_decision* h_dec_raw = nullptr;
  uint64_t dec_cap = 0;
  // a window in flight (grp_classify_reads_begin)
  hipEvent_t done = nullptr;   // decisions of the window are in h_dec
  hipEvent_t qdone = nullptr;  // the window's query kernel has finished (decide stream waits on it)
  bool side_used = false;      // `done` was last recorded on the decide stream
  // streaming window (grp_classify_stream_*)
  grp_read_decision* h_sdec = nullptr;    // mapped, coherent: the kernel publishes decisions here
  grp_read_decision* dmap_sdec = nullptr;
  uint64_t sdec_cap = 0;
  // abort flag: the host raises the one in mapped host memory (a plain store); every
  // 64th workgroup reads that one when it starts and relays it to the flag in DEVICE
  // memory, which is what all workgroups test (a PCIe round trip in every workgroup
  // costs the kernel 2.4x, measured; a 4-byte hipMemcpyAsync takes ~1.5 ms to land
  // beside a full-device kernel, measured)
  uint32_t* d_abort = nullptr;
  uint32_t* h_abort = nullptr;            // mapped, coherent
  uint32_t* dmap_abort = nullptr;
  uint32_t* d_tiles_done = nullptr;
  uint64_t tiles_done_cap = 0;
  unsigned long long* d_executed = nullptr;
  unsigned long long* h_executed = nullptr; // pinned
  bool streaming = false;
  // a streaming window that applies inserts itself (grp_classify_stream_insert)
  uint32_t* d_tile_ver = nullptr; // round 6: per tile of a resumable window, the generation its summary belongs to (DevStreamCtl::tile_ver)
  uint64_t tile_ver_cap = 0;
  uint32_t* d_rel = nullptr;    // release words of a resumable window's grid-wide waits: a 128-byte line per workgroup (STREAM_REL_WGS of them)
  uint32_t* d_dbg = nullptr;    // developer (GRP_STREAM_DEBUG): a state word per workgroup of the streaming launch
  uint32_t* d_sctl = nullptr;   // SCT_* control block
  uint32_t* h_cmd = nullptr;    // mapped, coherent: [0] sequence number, [1..10] the command
  uint32_t* dmap_cmd = nullptr;
  uint32_t* h_ack = nullptr;    // mapped, coherent: [0] inserts applied by the launch, [1] failure code
  uint32_t* dmap_ack = nullptr;
  bool resumable = false;       // the window in flight takes grp_classify_stream_insert
  uint32_t stripe_reads = 0, n_owners = 1, owner = 0; // its stripes (0: the whole window is this launch's)
  uint32_t gen = 1;             // generation of the records the host is waiting for
  uint32_t cmd_seq = 0;         // commands posted to the window in flight
  uint32_t* h_stripe_tiles = nullptr; // pinned: this rank's tiles of a striped window
  uint32_t* d_stripe_tiles = nullptr;
  uint64_t stripe_tiles_cap = 0;
  bool busy = false;
  const grp_reads* reads = nullptr;
  uint32_t first = 0, count = 0;
  grp_decide_params dp{};
  uint64_t list_cap = 0;
  uint64_t t0 = 0, nt = 0, probes = 0;
};

struct DevBatchView; // grp_kernels.inc

// state of a batch of inserts applied ahead of their reads' queries (grp_batch_*, grp_batch.inc)
struct BatchRun
{
  bool active = false;
  uint32_t n_ins = 0, block_tiles = 0;
  uint64_t n_units = 0, n_rec = 0;
  uint32_t* h_counters = nullptr; // page-locked [4]: records that own a chain, chained touches, log entries, error
  std::vector<uint32_t> h_ins;    // [n_ins][6]: read, tile_start, tile_end, first_id, id_offset, first unit
  uint32_t* h_ins_stage = nullptr; // page-locked copy of h_ins: the upload does not wait for the stream
  uint64_t ins_stage_cap = 0;
  uint32_t* d_counters = nullptr; // two sets of 4: a batch's clean-up zeroes the NEXT batch's set (k_batch_clear)
  uint32_t ctr_set = 0;
  uint32_t* d_ins = nullptr;
  uint64_t ins_cap = 0;
  uint32_t epoch = 0; // of the current / last batch: the claim field of the count words (grp_device.h), 1 .. GRP_EPOCH_MAX
  uint64_t tab_cap = 0, cur_cap = 0;    // allocated slots / slots the current batch uses (both tables)
  unsigned long long* d_rec_key = nullptr; // records, one per (frame, seed) of the inserted tiles
  unsigned long long* d_rec_loc = nullptr;
  unsigned long long* d_rec_old = nullptr;
  uint32_t* d_rec_chain = nullptr;
  uint64_t rec_cap = 0;
  uint32_t* d_ovf_next = nullptr; // touches of a rank by other (read, block)s than its owner's, chained per record
  uint32_t* d_ovf_jb = nullptr;
  uint32_t* d_chained = nullptr; // the records that own a chain (k_batch_apply walks these, not all records)
  uint64_t ovf_cap = 0;
  unsigned long long* d_log_keys = nullptr;
  uint32_t* d_log_head = nullptr;
  uint32_t* d_log_bits = nullptr; // 1 bit per log table slot: something was logged with this home slot
  uint64_t log_tab_cap = 0;
  uint32_t* d_log_old = nullptr;
  uint32_t* d_log_writer = nullptr;
  uint32_t* d_log_next = nullptr;
  uint32_t* d_log_slot = nullptr; // log entry -> its slot in the log table (to clean it)
  uint64_t log_cap = 0;
  uint32_t* d_floor = nullptr;    // grp_batch_classify: per read of the window
  uint64_t floor_cap = 0;
  // grp_batch_verify: the window's tiles with records, then the tiles queried again; per read its insert entry
  uint32_t* h_vf_stage = nullptr; // page-locked: per read of the batch its insert entry, then the floors
  uint64_t vf_stage_cap = 0;
  uint32_t* d_vf_stage = nullptr;
  uint64_t vf_stage_dev_cap = 0;
  uint32_t first_read = 0;        // reads are numbered from here in the log
};

constexpr int FQ_TEXT_SLOTS = 3; // device text buffers of the FASTQ ingest (grp_ingest.inc)

struct grp_ctx
{
  int device = 0;
  std::string arch;
  bool coherent_arch = false; // gfx942 / gfx950: agent-scope accesses are served by the memory side
  BatchRun batch;
  hipStream_t stream = nullptr;
  // decision kernel + copy-back of a pipelined window run here, next to the following
  // window's query kernel on `stream`
  hipStream_t stream2 = nullptr;
  hipStream_t stream3 = nullptr; // the uploads of coming FASTQ chunks (grp_fastq_prefetch): a copy must not stand in front of the parse of the chunk before it
  grp_params params{};
  std::vector<std::string> seeds;
  DevSeeds h_seeds{};
  DevSeeds* d_seeds = nullptr;
  uint32_t* d_gtab = nullptr; // count tables of k_query<..., GT> (query_geom: tiles whose worst case does not fit the LDS)
  uint64_t gtab_cap = 0;
  DevFilter f{};
  uint64_t nsb = 0;      // superbuckets
  uint64_t n_bv_words = 0;
  uint64_t n_ovf = 0;    // IDs living in the overflow table
  uint64_t n_far = 0;    // count words living in the far table
  uint32_t uniform_weight = 0; // weight shared by all seeds, 0 if they differ

  // developer switches, read once at grp_create (not on every call of a latency path)
  bool env_no_direct = false, env_stream_resume_off = false, env_no_early_park = false, env_trace_abort = false;
  uint32_t env_small_hist = 0; // GRP_SMALL_HIST (developer hook / tests): slots of the first-step count table
  // grp_window_overlap (grp_batch.inc): samples per tile, their table, the result per read
  struct OverlapBuf
  {
    unsigned long long* d_samples = nullptr;
    uint64_t samples_cap = 0; // in tiles
    uint32_t* d_n = nullptr;
    uint64_t n_cap = 0;
    unsigned long long* d_tab = nullptr;
    uint64_t tab_cap = 0;
    uint32_t* d_prev = nullptr;
    uint64_t prev_cap = 0;
  } ovl;
  uint64_t n_overlap_calls = 0;
  uint32_t batch_epochs = GRP_EPOCH_MAX; // batches between two sweeps of the claims (GRP_BATCH_EPOCHS: tests)
  uint64_t n_batch_sweeps = 0; // times the batch epochs wrapped and the claims were swept out of the count words
  uint64_t n_stream_keep[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; // grp_debug_stream_stats
  uint64_t n_stream_idle_exits = 0, n_stream_coop_refused = 0; // parked windows that left by themselves (idle limit); resumable windows whose cooperative launch was refused
  uint64_t n_flagged_tiles = 0, n_flagged_distinct = 0, n_flagged_list = 0; // statistics: tiles redone with the worst-case table; why (distinct IDs / list length)
  uint64_t n_verify_impossible = 0;
  uint64_t n_verify_tiles = 0, n_verify_queried = 0, n_verify_flagged = 0, n_verify_fallbacks = 0, n_verify_uncertified = 0, n_verify_unpatched = 0; // grp_batch_verify: tiles patched from records / queried again / patched tiles redone / calls that took the second query
  uint64_t n_direct_windows = 0, n_direct_fallbacks = 0, n_general_windows = 0, n_redo_launches = 0; // GRP_DEBUG_STATS
  uint64_t n_chunks = 0; // rank-build chunks
  uint64_t* d_super = nullptr;
  // grp_set_occupancy_hint: the phase-2 tables allocated by a helper thread while the fill runs (grp_finalize joins it)
  struct Prealloc
  {
    double occupancy = 0.0;
    bool started = false;
    std::thread worker;
    uint4* units = nullptr;
    uint64_t units_buckets = 0; // capacity in 128-byte units
    ulonglong2* far = nullptr;
    uint64_t far_cap = 0;       // slots (a power of two)
  } pre;
  double finalize_times[6] = { 0, 0, 0, 0, 0, 0 };
  bool finalized = false;
  int n_cus = 0;
  std::mutex fill_mutex; // grp_bv_insert may be called from several host threads (the reference calls insertBV under `omp parallel`)
  // --ntcard pass (grp_ntcard.inc)
  uint32_t* d_ntc = nullptr;
  uint32_t ntc_sbits = 0;
  uint2* d_ntc_chunks = nullptr;
  uint64_t ntc_chunks_cap = 0;
  uint32_t* d_ntc_extra = nullptr;
  uint64_t ntc_extra_cap = 0;
  // query scratch: two slots so that a second window can be in flight while the
  // host commits the first (grp_classify_reads_begin / _end)
  QuerySlot slot[2];
  QuerySlot* q = &slot[0];
  // first decisions kept for grp_batch_verify: slot `slot` holds, from tile `tile_off` of its buffers on, the plain
  // tile summaries of reads [first, first + count) of `reads` against the filter as it is (as it was in front of the
  // batch, between grp_batch_insert_reads and grp_batch_end / _undo).  Dropped by whatever changes the filter or the slot.
  struct Carry
  {
    bool valid = false;
    uint32_t slot = 0;
    const grp_reads* reads = nullptr;
    uint32_t first = 0, count = 0;
    uint64_t tile_off = 0;
  } carry;
  const DevBatchView* view = nullptr; // set while grp_batch_classify enqueues its window
  grp_tile_summary* h_tiles = nullptr; // pinned staging
  uint64_t h_tiles_cap = 0;
  grp_id_count* h_lists = nullptr; // pinned staging of the list prefix
  // small windows: kernel writes straight into mapped host memory
  grp_tile_summary* h_small_tiles = nullptr;
  grp_id_count* h_small_lists = nullptr;
  grp_tile_summary* dmap_small_tiles = nullptr;
  grp_id_count* dmap_small_lists = nullptr;
  // insert scratch
  unsigned long long* d_dedup = nullptr;
  uint64_t dedup_cap = 0;
  uint32_t epoch = 0;
  // whole-read insert table
  unsigned long long* d_ir_keys = nullptr;
  unsigned long long* d_ir_masks = nullptr;
  unsigned long long* d_ir_locs = nullptr;
  uint32_t* d_ir_slots = nullptr;
  uint32_t* d_ir_counter = nullptr;
  uint32_t dbg_stream_lds = 0, dbg_stream_grid = 0; // developer (GRP_STREAM_DEBUG): the last streaming launch
  uint32_t* d_fp_tab[2] = { nullptr, nullptr }; // changed-slot sets of a streaming window's last two in-launch inserts (ir_cap words each)
  uint64_t ir_cap = 0;
  uint32_t ir_parity = 0;
  // RCCL communicator of a multi-GPU fill (grp_comm.inc), NULL on one GPU
  void* comm = nullptr;
  uint32_t comm_world = 1, comm_rank = 0;
  uint32_t n_comm_merges = 0; // grp_bv_merge_ranks calls that succeeded
  double* d_delog = nullptr; // 10^(-Q/10) table for the FASTQ ingest
  // buffers of the FASTQ ingest, kept between chunks (round 4: a chunk used to pay seven hipMalloc / hipFree pairs,
  // 256 MiB of text among them — every hipFree waits for the device)
  struct IngestPool
  {
    uint8_t* text[FQ_TEXT_SLOTS] = {}; // the chunk being packed, the next one (uploaded ahead) and the one after (uploading)
    uint64_t text_cap[FQ_TEXT_SLOTS] = {};
    bool text_used[FQ_TEXT_SLOTS] = {};
    int last_slot = FQ_TEXT_SLOTS - 1; // the slot handed out last (the next one takes the one after it)
    // grp_fastq_prefetch: bodies of coming chunks on their way into slots, oldest first
    struct Pre
    {
      int slot = -1;
      const char* body = nullptr;
      uint64_t n_body = 0;
    } pre[2];
    int n_pre = 0;
    uint64_t n_dropped = 0; // prefetched bodies that were not followed by their parse (uploaded for nothing)
    uint32_t* d_counts = nullptr;
    uint64_t counts_cap = 0;
    uint64_t *d_base = nullptr, *d_super = nullptr, *d_total = nullptr, *d_nl = nullptr;
    uint64_t base_cap = 0, super_cap = 0, total_cap = 0, nl_cap = 0;
    void* d_rec = nullptr; // grp_fastq_record[]
    uint64_t rec_bytes = 0;
    uint64_t *d_so = nullptr, *d_wo = nullptr; // grp_fastq_pack: sequence offsets, word offsets, lengths of the selection
    uint32_t* d_len = nullptr;
    uint64_t so_cap = 0, wo_cap = 0, len_cap = 0;
    hipEvent_t uploaded = nullptr; // the chunk's text has arrived (copied on the side stream, beside the fill of the chunk before)
    // page-locked, device-mapped staging of a parse (grp_ingest.inc: ingest_stage): [0] newlines, [1] end of the last record
    uint64_t* h_scal = nullptr;
    uint64_t* dm_scal = nullptr;
    uint8_t* h_front = nullptr; // the bytes in front of a prefetched body
    uint8_t* dm_front = nullptr;
    uint8_t* h_rec = nullptr; // the record table
    uint8_t* dm_rec = nullptr;
    uint64_t h_rec_bytes = 0;
    hipEvent_t text_up[FQ_TEXT_SLOTS] = {};   // the slot's prefetched body has arrived (recorded on the copy stream)
    hipEvent_t text_done[FQ_TEXT_SLOTS] = {}; // the main stream's last use of the slot's text (grp_fastq_free records it: the next upload into the slot waits for it, not the host)
  } ingest;
  const char* reg_text = nullptr; // the caller's text buffer, page-locked by grp_fastq_pin
  size_t reg_bytes = 0;
  uint32_t timing_mask = (1u << GRP_K_FILL) | (1u << GRP_K_RANK) | (1u << GRP_K_QUERY) | (1u << GRP_K_DECIDE) | (1u << GRP_K_QUERY_LAT) | (1u << GRP_K_VERIFY) | (1u << GRP_K_BATCH);
  // timing
  bool timing = true;
  std::vector<EventPair> pending;
  std::vector<EventPair> free_events;
  grp_kernel_stat kstat[GRP_K_COUNT]{};
  mutable std::string err;
};

struct grp_reads
{
  grp_ctx* ctx = nullptr;
  uint32_t n_reads = 0;
  uint64_t n_words = 0;
  bool owns_packed = false;
  uint32_t* d_packed = nullptr;
  uint64_t* d_word_off = nullptr;
  uint32_t* d_len = nullptr;
  uint64_t* d_tile0 = nullptr;
  uint32_t* d_tile_read = nullptr;
  uint64_t* d_chunk0 = nullptr;
  uint32_t* d_chunk_read = nullptr;
  std::vector<uint64_t> tile0;  // host copy
  std::vector<uint64_t> chunk0; // host copy
  std::vector<uint32_t> len;    // host copy
  // grp_fastq_pack (round 5): one device allocation behind all index arrays, filled by asynchronous copies out of these
  // vectors — they live as long as the batch, so no call has to wait for a copy
  void* d_slab = nullptr;
  std::vector<uint64_t> h_word_off, h_seq_off;
  std::vector<uint32_t> h_tile_read, h_chunk_read;
  DevReads dev{};
};

namespace {

void comm_release(grp_ctx* c); // grp_comm.inc

int
set_err(const grp_ctx* ctx, int code, const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) {
    ctx->err = buf;
  } else {
    g_create_error = buf;
  }
  return code;
}

#define HIP_TRY(ctx, expr)                                                                                             \
  do {                                                                                                                 \
    hipError_t e_ = (expr);                                                                                            \
    if (e_ != hipSuccess) {                                                                                            \
      return set_err(ctx, GRP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);     \
    }                                                                                                                  \
  } while (0)

constexpr uint32_t FILL_CHUNK = 2048; // read positions per fill workgroup
constexpr uint64_t LIST_PREFIX = 8192; // list entries copied back together with the tile summaries
constexpr uint32_t SMALL_TILES = 512;  // windows up to this many tiles use the direct (zero-copy) path
constexpr uint32_t SMALL_STRIDE = 32;  // list entries per tile in the direct path
constexpr int THREADS = 256;
// A HIP grid holds fewer than 2^32 work-items per dimension (larger ones are truncated
// modulo 2^32 without an error): no launch of 256-lane workgroups exceeds this many
constexpr uint32_t MAX_GRID_WGS = 1u << 22;

// ---- ntHash (btllib::SeedNtHash, restated; see DESIGN.md "Hash") -----------

uint64_t
srol_n(uint64_t x, unsigned d)
{
  const uint64_t M33 = 0x1FFFFFFFFULL, M31 = 0x7FFFFFFFULL;
  uint64_t lo = x & M33, hi = (x >> 33) & M31;
  unsigned rl = d % 33, rh = d % 31;
  if (rl) {
    lo = ((lo << rl) | (lo >> (33 - rl))) & M33;
  }
  if (rh) {
    hi = ((hi << rh) | (hi >> (31 - rh))) & M31;
  }
  return (hi << 33) | lo;
}

const uint64_t BASE_SEED[4] = { 0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL, 0x20323ed082572324ULL, 0x295549f54be24456ULL };

} // namespace

#include "grp_kernels.inc"

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

namespace {

struct Timer
{
  grp_ctx* c;
  EventPair ep{};
  bool on = false;
  hipStream_t st;
  Timer(grp_ctx* ctx, int kind, uint64_t units, hipStream_t on_stream = nullptr)
    : c(ctx)
    , st(on_stream ? on_stream : ctx->stream)
  {
    c->kstat[kind].launches += 1;
    c->kstat[kind].units += units;
    if (!c->timing || !(c->timing_mask & (1u << kind))) {
      return;
    }
    if (!c->free_events.empty()) {
      ep = c->free_events.back();
      c->free_events.pop_back();
    } else {
      if (hipEventCreate(&ep.a) != hipSuccess || hipEventCreate(&ep.b) != hipSuccess) {
        return;
      }
    }
    ep.kind = kind;
    on = true;
    (void)hipEventRecord(ep.a, st);
  }
  ~Timer()
  {
    if (on) {
      (void)hipEventRecord(ep.b, st);
      c->pending.push_back(ep);
    }
  }
};

// accumulate the durations of the launches that have completed; launches still in
// flight (a pipelined next window) stay pending
void
drain_events(grp_ctx* c)
{
  std::vector<EventPair> keep;
  for (auto& ep : c->pending) {
    if (hipEventQuery(ep.b) != hipSuccess) {
      keep.push_back(ep);
      continue;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
      c->kstat[ep.kind].ms += (double)ms;
    }
    c->free_events.push_back(ep);
  }
  c->pending.swap(keep);
}

uint64_t
next_pow2_64(uint64_t v)
{
  uint64_t p = 1;
  while (p < v) {
    p <<= 1;
  }
  return p;
}

uint32_t
next_pow2(uint64_t v)
{
  uint32_t p = 1;
  while (p < v) {
    p <<= 1;
  }
  return p;
}

size_t
tab_bytes(const grp_ctx* c)
{
  return (size_t)c->h_seeds.h * c->h_seeds.wmax * 4u * sizeof(ulonglong2);
}

size_t
bases_bytes(uint32_t nbases)
{
  // words covering nbases at any 16-base phase + 2 pad words, rounded to 16 B
  size_t words = (nbases + 15u) / 16u + 1u + 4u;
  return ((words * 4u + 15u) / 16u) * 16u;
}

template<typename K>
int
ensure_lds(grp_ctx* c, K kernel, size_t bytes)
{
  if (bytes > 64 * 1024) {
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  }
  return GRP_OK;
}

// k_decide's LDS work space: reads of more than 256 tiles (register forms above) up to DECIDE_LDS_MAX_TILES
inline uint32_t
decide_lds_tiles(const uint64_t* tile0, uint32_t count)
{
  uint64_t most = 0;
  for (uint32_t i = 0; i < count; ++i) {
    most = std::max(most, tile0[i + 1] - tile0[i]);
  }
  return most <= 4 * LANE_TILES ? 0u : (uint32_t)std::min<uint64_t>((most + 63) / 64 * 64, DECIDE_LDS_MAX_TILES);
}

inline size_t
decide_lds_bytes(uint32_t lds_tiles)
{
  return (size_t)lds_tiles * DECIDE_LDS_BYTES_PER_TILE;
}

#define DISPATCH_H(hval, CALL)                                                                                         \
  switch (hval) {                                                                                                      \
    case 1: { constexpr int HH = 1; CALL; } break;                                                                     \
    case 2: { constexpr int HH = 2; CALL; } break;                                                                     \
    case 3: { constexpr int HH = 3; CALL; } break;                                                                     \
    case 4: { constexpr int HH = 4; CALL; } break;                                                                     \
    case 5: { constexpr int HH = 5; CALL; } break;                                                                     \
    case 6: { constexpr int HH = 6; CALL; } break;                                                                     \
    case 7: { constexpr int HH = 7; CALL; } break;                                                                     \
    default: { constexpr int HH = 8; CALL; } break;                                                                    \
  }

// decision buffers of a slot: [64 bytes of counters][cap decisions], on the device and page-locked on the host
int
slot_dec_alloc(grp_ctx* c, QuerySlot& sl, uint64_t cap)
{
  static_assert(2 * sizeof(grp_read_decision) == 8 * sizeof(uint64_t), "the counters take two decision entries");
  (void)hipFree(sl.d_dec_raw);
  sl.d_dec_raw = sl.d_dec = nullptr;
  if (sl.h_dec_raw) {
    (void)hipHostFree(sl.h_dec_raw);
    sl.h_dec_raw = sl.h_dec = nullptr;
  }
  sl.h_qctr = nullptr;
  sl.dec_cap = 0;
  HIP_TRY(c, hipMalloc(&sl.d_dec_raw, (cap + 2) * sizeof(grp_read_decision)));
  HIP_TRY(c, hipHostMalloc(&sl.h_dec_raw, (cap + 2) * sizeof(grp_read_decision), hipHostMallocDefault));
  memset(sl.h_dec_raw, 0, 2 * sizeof(grp_read_decision));
  sl.d_dec = sl.d_dec_raw + 2;
  sl.h_dec = sl.h_dec_raw + 2;
  sl.h_qctr = reinterpret_cast<uint64_t*>(sl.h_dec_raw);
  sl.dec_cap = cap;
  return GRP_OK;
}

template<typename T>
int ensure_dev(grp_ctx* c, T*& p, uint64_t& cap, uint64_t want);

constexpr size_t LDS_PER_WORKGROUP = 160 * 1024 - 512; // gfx950, less the kernels' static shared variables
constexpr uint32_t GT_SLICE = 256;                     // workgroups of one launch with global count tables

struct QueryGeom
{
  uint32_t hist_cap;       // slots of the per-tile count table (even; 6 bytes each: a key and a 16-bit count)
  uint32_t distinct_limit; // distinct IDs it takes (the rest is room for the claims in flight)
  size_t lds;
  bool global_table = false; // the worst-case table does not fit the LDS: it lives in global memory (k_query<..., GT>; the redo launches only)
};

// The per-tile count table (k_query).  full = worst case: every probe of the tile a different ID — tile * h slots
// plus one claim per lane and seed in flight.  Round 4: at 6 bytes per slot and without power-of-two rounding the
// worst case of the default geometries fits the LDS share that lets three workgroups (h = 5: 48 KB) or four (h = 3:
// 27 KB) onto a CU, so it is what every launch uses and no tile is flagged; only geometries beyond that (h = 8 with
// tiles of 1000: 69 KB; long tiles) keep the two-step scheme: a table of the budget's size first, the tiles whose
// distinct IDs do not fit flagged and recomputed by the same kernel with the worst-case table.
QueryGeom
query_geom(const grp_ctx* c, bool full)
{
  const uint32_t h = c->params.h, tile = c->params.tile;
  const uint64_t max_ids = (uint64_t)tile * h;
  auto lds_of = [&](uint32_t cap) { return tab_bytes(c) + (size_t)cap * 6 + bases_bytes(tile + c->params.k + h); };
  const uint32_t cap_full = (uint32_t)((max_ids + (uint64_t)THREADS * h + 2 + 1023) / 1024 * 1024);
  const uint32_t forced_small = c->env_small_hist; // developer hook / tests (GRP_SMALL_HIST, read when the context is created): slots of the first-step table — forces the two-step scheme
  QueryGeom g;
  g.hist_cap = cap_full;
  if (!full) {
    if (forced_small && forced_small < cap_full && forced_small > (uint32_t)THREADS * h + 64) {
      g.hist_cap = forced_small;
    } else if (lds_of(cap_full) > 52 * 1024) {
      // the largest table that still lets three workgroups share a CU's 160 KB
      const size_t fixed = lds_of(0);
      const uint32_t cap = (uint32_t)((52 * 1024 - std::min<size_t>(fixed, 40 * 1024)) / 6 / 1024 * 1024);
      if (cap > (uint32_t)THREADS * h + 1024 && cap < cap_full) {
        g.hist_cap = cap;
      }
    }
  }
  g.distinct_limit = g.hist_cap - THREADS * h - 1;
  g.lds = lds_of(g.hist_cap);
  if (full && g.lds > LDS_PER_WORKGROUP) {
    // Round 5: tiles beyond ~9 000 frames at h = 3 (the reference has no such limit).  The first step's table is LDS-sized
    // (above: the three-workgroups budget); what it cannot hold is redone with the worst-case table in global memory.
    g.global_table = true;
    g.lds = lds_of(0);
  }
  return g;
}

template<int HH>
int
launch_query(grp_ctx* c, const grp_reads* r, uint64_t n_launch, uint64_t t0, const uint32_t* d_tile_idx, const QueryGeom& g, uint64_t list_cap, grp_tile_summary* out_tiles = nullptr, grp_id_count* out_lists = nullptr, uint32_t direct_stride = 0, const DevStreamCtl* stream_ctl = nullptr,
             uint32_t blk0 = 0,          // no tile list: the launch covers tiles [blk0, blk0 + n_launch) of the window
             bool list_flags = false,    // a tile list that is NOT the redo of flagged tiles: flagged tiles are collected as without a list
             bool plain = false)         // the plain query even while a batch view is set (the reads behind the batch)
{
  if (!out_tiles) {
    out_tiles = c->q->d_tiles;
    out_lists = c->q->d_lists;
  }
  size_t launch_lds = g.lds; // dynamic LDS of the launch (the streaming form adds its fingerprint buffers)
  auto go = [&](auto kern) -> int {
    int rc = ensure_lds(c, kern, launch_lds);
    if (rc != GRP_OK) {
      return rc;
    }
    if (stream_ctl && stream_ctl->ctl) {
      // A window that applies inserts itself: its workgroups wait for each other inside the launch.  The grid is sized
      // from the occupancy query below and launched cooperatively — the runtime checks that it fits the device (or
      // refuses: another process on it).  That is a check of sizes, not of residency: now and then a few
      // workgroups begin only when others leave, so the waits count the workgroups that have begun (stream_register).
      // GRP_ERR_BUSY: refused — the caller begins the window in its classic form (it ends where it parks).
      {
        DevFilter a_f = c->f;
        DevReads a_rd = r->dev;
        const DevSeeds* a_sd = c->d_seeds;
        uint32_t a_tile = c->params.tile;
        uint64_t a_t0 = t0;
        const uint32_t* a_idx = d_tile_idx;
        uint32_t a_cap = g.hist_cap, a_lim = g.distinct_limit;
        grp_tile_summary* a_tiles = out_tiles;
        grp_id_count* a_lists = out_lists;
        uint64_t a_lcap = list_cap;
        unsigned long long* a_ctr = reinterpret_cast<unsigned long long*>(c->q->d_qctr);
        uint32_t* a_flag = (d_tile_idx && !list_flags) ? nullptr : c->q->d_flag_idx;
        uint32_t a_fcap = (uint32_t)c->q->d_flag_cap, a_ds = direct_stride, a_blk0 = blk0;
        DevStreamCtl a_sc = *stream_ctl;
        DevBatchView a_bv{};
        uint32_t* a_gtab = nullptr;
        void* args[] = { &a_f, &a_rd, &a_sd, &a_tile, &a_t0, &a_idx, &a_cap, &a_lim, &a_tiles, &a_lists, &a_lcap, &a_ctr, &a_flag, &a_fcap, &a_ds, &a_blk0, &a_sc, &a_bv, &a_gtab };
        const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3((uint32_t)n_launch), dim3(THREADS), args, (unsigned int)launch_lds, c->stream);
        if (e == hipSuccess) {
          return GRP_OK;
        }
        (void)hipGetLastError();
        return set_err(c, GRP_ERR_BUSY, "cooperative launch of a resumable window refused: %s", hipGetErrorString(e));
      }
    }
    kern<<<dim3((uint32_t)n_launch), dim3(THREADS), launch_lds, c->stream>>>(c->f, r->dev, c->d_seeds, c->params.tile, t0, d_tile_idx, g.hist_cap, g.distinct_limit, out_tiles, out_lists, list_cap, reinterpret_cast<unsigned long long*>(c->q->d_qctr), (d_tile_idx && !list_flags) ? nullptr : c->q->d_flag_idx, (uint32_t)c->q->d_flag_cap, direct_stride, blk0, stream_ctl ? *stream_ctl : DevStreamCtl{}, (c->view && !plain) ? *c->view : DevBatchView{}, nullptr);
    return GRP_OK;
  };
  // The synchronous forms (large windows, the two queries of a batch): two frames per lane and pass
  // up to h = 3; from h = 4 on one frame per lane with the software-pipelined pass (222 -> 171 VGPRs
  // at h = 5; C4 geometry +4 %, h = 3 -3 %: measured, tools/dev/r3_fr1.sh of round 3; round 4 again with the 24 KB count table:
  // the head of C2 2.94 - 3.04 s against 2.94 - 2.96 s, no gain, the switch is gone)
  constexpr int SFR = (HH <= 3) ? 2 : 1;
  if (g.global_table) {
    // the worst-case table in global memory, GT_SLICE workgroups (and tables) per launch
    if (stream_ctl || direct_stride) {
      return set_err(c, GRP_ERR_STATE, "launch_query: a global count table in a streaming / zero-copy launch");
    }
    const uint64_t words = (uint64_t)g.hist_cap + g.hist_cap / 2u;
    if (const int rc = ensure_dev(c, c->d_gtab, c->gtab_cap, words * GT_SLICE); rc != GRP_OK) {
      return rc;
    }
    const bool ver = c->view && !plain;
    for (uint64_t off = 0; off < n_launch; off += GT_SLICE) {
      const uint32_t nb = (uint32_t)std::min<uint64_t>(GT_SLICE, n_launch - off);
      const uint32_t* idx = d_tile_idx ? d_tile_idx + off : nullptr;
      const uint32_t b0 = d_tile_idx ? blk0 : blk0 + (uint32_t)off;
      uint32_t* const flag_out = (d_tile_idx && !list_flags) ? nullptr : c->q->d_flag_idx;
      if (ver) {
        DevBatchView v = *c->view;
        if (v.redo_count && off) {
          break; // (a launch sized before its list was known covers one slice: longer lists take the host's way)
        }
        auto kern = k_query<HH, 1, 0, false, true, true>;
        if (const int rc = ensure_lds(c, kern, g.lds); rc != GRP_OK) {
          return rc;
        }
        kern<<<dim3(nb), dim3(THREADS), g.lds, c->stream>>>(c->f, r->dev, c->d_seeds, c->params.tile, t0, idx, g.hist_cap, g.distinct_limit, out_tiles, out_lists, list_cap, reinterpret_cast<unsigned long long*>(c->q->d_qctr), flag_out, (uint32_t)c->q->d_flag_cap, 0u, b0, DevStreamCtl{}, v, c->d_gtab);
      } else {
        auto kern = k_query<HH, 1, 0, false, false, true>;
        if (const int rc = ensure_lds(c, kern, g.lds); rc != GRP_OK) {
          return rc;
        }
        kern<<<dim3(nb), dim3(THREADS), g.lds, c->stream>>>(c->f, r->dev, c->d_seeds, c->params.tile, t0, idx, g.hist_cap, g.distinct_limit, out_tiles, out_lists, list_cap, reinterpret_cast<unsigned long long*>(c->q->d_qctr), flag_out, (uint32_t)c->q->d_flag_cap, 0u, b0, DevStreamCtl{}, DevBatchView{}, c->d_gtab);
      }
    }
    return GRP_OK;
  }
  if (c->view && !stream_ctl && !plain) { // grp_batch_classify: every read sees the state in front of its own insert
    return go(k_query<HH, SFR, 0, false, true>);
  }
  if (stream_ctl) {
    // persistent workgroups: exactly what is resident at once
    // one frame per lane and pass: fewer registers, more resident workgroups — measured
    // better than two for the persistent form (h = 3: +3 %, h = 5: +19 %)
    auto kern = k_query<HH, 1, 0, true>;
    int per_cu = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, THREADS, g.lds));
    // Round 6: a window that applies inserts itself keeps the probes' fingerprints of the tile in progress and of the
    // tile before it in LDS (DevStreamCtl::n_fp) — as many of the two buffers as fit WITHOUT costing a resident workgroup
    // (three per CU is what the launch runs at, below): C2's geometry 27.9 + 2 x 12.3 KB, three of them in the CU's 160 KB;
    // h = 5 with tiles of 1000 (48 KB of count table) has no room: such a window hands everything behind an insert out again.
    DevStreamCtl sc_fit = *stream_ctl;
    QueryGeom g_fit = g;
    sc_fit.n_fp = 0;
    if (stream_ctl->ctl && stream_ctl->n_fp) {
      static const int keep_env = [] { // developer switch / tests: fingerprint buffers per workgroup (0: rounds 3 - 5's form)
        const char* e = getenv("GRP_STREAM_KEEP");
        return e ? std::max(0, std::min(2, atoi(e))) : 2;
      }();
      const uint32_t fb = 3u * (64u - ((uint32_t)HH - 1u)) + 64u; // frames per pass at least (k_query: the helper-lane layout)
      const uint32_t passes = (c->params.tile + fb - 1u) / fb;
      const uint32_t words = passes * (uint32_t)THREADS * (uint32_t)HH;
      const int want_cu = std::min(per_cu, 3);
      for (int nf = std::min<int>(keep_env, (int)stream_ctl->n_fp); nf >= 1; --nf) {
        const size_t lds = g.lds + (size_t)nf * words * 4u;
        int fit = 0;
        if (lds <= LDS_PER_WORKGROUP && ensure_lds(c, kern, lds) == GRP_OK && hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, kern, THREADS, lds) == hipSuccess && fit >= want_cu) {
          sc_fit.n_fp = (uint32_t)nf;
          sc_fit.fp_off = (uint32_t)g.lds;
          sc_fit.fp_words = words;
          g_fit.lds = lds;
          break;
        }
        (void)hipGetLastError();
      }
    }
    stream_ctl = &sc_fit;
    launch_lds = g_fit.lds; // (the launch takes the geometry with the fingerprint buffers)
    int rc = ensure_lds(c, kern, launch_lds);
    if (rc != GRP_OK) {
      return rc;
    }
    static const int cap_per_cu = [] { // developer hook
      const char* e = getenv("GRP_STREAM_WGS_PER_CU");
      return e ? atoi(e) : 0;
    }();
    // 3 .. 6 resident workgroups per CU run equally fast (the kernel is bound by the DRAM
    // access rate, not by occupancy); 4 keeps the amount of work in flight — what an abort
    // throws away — small and leaves room on every CU for other kernels (copies, a
    // collective) beside the persistent launch
    // Round 4: THREE, not four.  With the 24 KB count table a fourth workgroup fits a CU by LDS and VGPRs (128: four
    // waves per SIMD, every wave slot of the device), and the occupancy query — and hipLaunchCooperativeKernel — say
    // so; but the launch then is NOT fully resident in practice (the kernel keeps 4 VGPRs in scratch, and the
    // dispatcher admits fewer scratch waves than wave slots): on the C1 stream the grid-wide waits of the in-launch
    // insert timed out (inserts handed back to the host: 243 k reads/s instead of 489 k, or code 2 with the
    // cooperative launch), profiles/r04_c1_workgroups_per_cu.txt.  Rounds 1 - 3 ran three (the 43.6 KB table allowed
    // no more); three leave a wave slot per SIMD free, and are as fast (C1 559 k reads/s in the steady state, two: 500 k).
    per_cu = std::min(per_cu, cap_per_cu > 0 ? cap_per_cu : 3);
    const uint64_t resident = std::min<uint64_t>((uint64_t)std::max(per_cu, 1) * (uint64_t)std::max(c->n_cus, 1), STREAM_REL_WGS);
    n_launch = std::min<uint64_t>(n_launch, resident);
    c->dbg_stream_lds = (uint32_t)launch_lds;
    c->dbg_stream_grid = (uint32_t)n_launch;
    return go(kern);
  }
  // (round 2 had a second form for windows of a few reads, the care loop unrolled for weight-16
  // seeds: 346 VGPRs + 14 spilled at h = 3.  With the shared halves the hash is a third of the
  // instructions it was and the batches took the insert-heavy phases over: the form is gone — the
  // CLI's silver run takes the same time with and without it, tools/dev/r3_latform.sh.)
  return go(k_query<HH, SFR, 0, false>);
}

int
build_seed_tables(grp_ctx* c)
{
  DevSeeds& sd = c->h_seeds;
  memset(&sd, 0, sizeof(sd));
  sd.h = c->params.h;
  sd.k = c->params.k;
  sd.wmax = 0;
  for (uint32_t s = 0; s < sd.h; ++s) {
    const std::string& p = c->seeds[s];
    if (p.size() != (size_t)c->params.k + s) {
      return set_err(c, GRP_ERR_INVALID, "seed %u has span %zu, expected k+%u = %u", s, p.size(), s, c->params.k + s);
    }
    sd.span[s] = (uint32_t)p.size();
    uint32_t w = 0;
    for (uint32_t q = 0; q < p.size(); ++q) {
      if (p[q] == '1') {
        if (w >= GRP_DEV_MAX_W) {
          return set_err(c, GRP_ERR_INVALID, "seed weight > %d unsupported", GRP_DEV_MAX_W);
        }
        sd.shift[s][w] = 2u * q;
        for (uint32_t b = 0; b < 4; ++b) {
          sd.tab[s][w][b].x = srol_n(BASE_SEED[b], sd.span[s] - 1u - q);
          sd.tab[s][w][b].y = srol_n(BASE_SEED[3u - b], q);
        }
        ++w;
      } else if (p[q] != '0') {
        return set_err(c, GRP_ERR_INVALID, "seed %u contains '%c'", s, p[q]);
      }
    }
    sd.weight[s] = w;
    sd.wmax = std::max(sd.wmax, w);
  }
  if (sd.wmax == 0) {
    return set_err(c, GRP_ERR_INVALID, "seeds have weight 0");
  }
  c->uniform_weight = sd.wmax;
  for (uint32_t s2 = 0; s2 < sd.h; ++s2) {
    if (sd.weight[s2] != sd.wmax) {
      c->uniform_weight = 0;
    }
  }
  // spans beyond 32 bases (round 4): the generic hash reads a second 64-bit window; the unrolled weight-16 form and
  // the shared halves are 32-base forms and stay off
  sd.wide = (c->params.k + sd.h - 1 > 32) ? 1u : 0u;
  if (sd.wide) {
    c->uniform_weight = 0;
  }
  // make_seed_pattern's family (spaced_seeds.cpp:58-66): seed i = left || "0" x i || right with
  // left = the first k/2 positions of seed 0.  The query kernel then evaluates the halves once per
  // frame (GRP_SHARED_HALVES=off: developer switch, every seed on its own as before).
  sd.n_left = 0;
  {
    const char* e = getenv("GRP_SHARED_HALVES");
    const std::string& s0 = c->seeds[0];
    const size_t cut = s0.size() / 2;
    bool family = sd.h >= 2 && !(e && !strcmp(e, "off"));
    for (uint32_t s2 = 1; s2 < sd.h && family; ++s2) {
      family = c->seeds[s2] == s0.substr(0, cut) + std::string(s2, '0') + s0.substr(cut);
    }
    uint32_t nl = 0;
    for (size_t q = 0; q < cut; ++q) {
      nl += s0[q] == '1';
    }
    if (family && nl > 0 && nl < sd.weight[0] && !sd.wide) {
      sd.n_left = nl;
    }
  }
  return GRP_OK;
}

} // namespace

extern "C" {

const char*
grp_last_error(const grp_ctx* ctx)
{
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int
grp_create(const grp_params* p, grp_ctx** out)
{
  if (!p || !out || p->struct_size != sizeof(grp_params)) {
    return set_err(nullptr, GRP_ERR_INVALID, "grp_create: bad params / struct_size");
  }
  *out = nullptr;
  if (p->h < 1 || p->h > GRP_MAX_SEEDS) {
    return set_err(nullptr, GRP_ERR_INVALID, "h=%u outside [1,%d]", p->h, GRP_MAX_SEEDS);
  }
  if (p->k < 1 || p->k + p->h - 1 > GRP_MAX_SPAN) {
    return set_err(nullptr, GRP_ERR_INVALID, "k+h-1=%u exceeds the %d-base window of this implementation", p->k + p->h - 1, GRP_MAX_SPAN);
  }
  if (p->tile < p->k + p->h - 1) {
    return set_err(nullptr, GRP_ERR_INVALID, "tile length %u shorter than the longest seed span %u", p->tile, p->k + p->h - 1);
  }
  if (p->tile > GRP_MAX_TILE) {
    return set_err(nullptr, GRP_ERR_INVALID, "tile length %u: a tile holds at most 65 535 frames (an ID's count per tile is 16 bits)", p->tile);
  }
  // m = 0: the size is not known yet (--ntcard estimates it from the reads);
  // grp_set_filter_size() must follow before the first grp_bv_insert
  if (p->m != 0 && (p->m < 64 || p->m >= (1ULL << 50))) {
    return set_err(nullptr, GRP_ERR_INVALID, "filter size m=%llu unsupported", (unsigned long long)p->m);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    return set_err(nullptr, GRP_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  }
  grp_ctx* c = new grp_ctx();
  c->params = *p;
  for (uint32_t s = 0; s < p->h; ++s) {
    c->seeds.emplace_back(p->seeds[s] ? p->seeds[s] : "");
  }
  c->params.seeds = nullptr;
  auto fail = [&](int code) {
    g_create_error = c->err;
    grp_destroy(c);
    return code;
  };
  if (p->device >= 0) {
    if (hipSetDevice(p->device) != hipSuccess) {
      set_err(c, GRP_ERR_NO_DEVICE, "hipSetDevice(%d) failed", p->device);
      return fail(GRP_ERR_NO_DEVICE);
    }
    c->device = p->device;
  } else if (hipGetDevice(&c->device) != hipSuccess) {
    set_err(c, GRP_ERR_NO_DEVICE, "hipGetDevice failed");
    return fail(GRP_ERR_NO_DEVICE);
  }
  {
    const char* e = getenv("GRP_STREAM_RESUME");
    c->env_stream_resume_off = e && !strcmp(e, "off");
    c->env_no_direct = getenv("GRP_NO_DIRECT") != nullptr;
    if (const char* e = getenv("GRP_BATCH_EPOCHS")) { // tests: the claims are swept every n batches instead of every 1023
      c->batch_epochs = (uint32_t)std::min<long>(std::max<long>(atol(e), 1), (long)GRP_EPOCH_MAX);
    }
    c->env_no_early_park = getenv("GRP_NO_EARLY_PARK") != nullptr;
    if (const char* e = getenv("GRP_SMALL_HIST")) {
      c->env_small_hist = (uint32_t)atoi(e) / 2u * 2u;
    }
    c->env_trace_abort = getenv("GRP_TRACE_ABORT") != nullptr;
  }
  int rc = build_seed_tables(c);
  if (rc != GRP_OK) {
    return fail(rc);
  }
  // LDS geometry of the query kernel's worst-case launch must fit one workgroup
  {
    const QueryGeom g = query_geom(c, true), g1 = query_geom(c, false);
    if (g.lds > LDS_PER_WORKGROUP || g1.lds > LDS_PER_WORKGROUP || g1.hist_cap <= (uint32_t)THREADS * p->h + 64) {
      set_err(c, GRP_ERR_INVALID, "tile=%u h=%u needs %zu B of LDS per workgroup beside the count table (limit 160 KiB)", p->tile, p->h, std::max(g.lds, g1.lds));
      return fail(GRP_ERR_INVALID);
    }
  }
#define CREATE_TRY(expr)                                                                                               \
  do {                                                                                                                 \
    hipError_t e_ = (expr);                                                                                            \
    if (e_ != hipSuccess) {                                                                                            \
      set_err(c, GRP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                                          \
      return fail(e_ == hipErrorOutOfMemory ? GRP_ERR_NOMEM : GRP_ERR_HIP);                                            \
    }                                                                                                                  \
  } while (0)
  {
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, c->device));
    c->n_cus = prop.multiProcessorCount;
    c->arch = prop.gcnArchName;
    c->coherent_arch = c->arch.compare(0, 5, "gfx94") == 0 || c->arch.compare(0, 5, "gfx95") == 0;
  }
  CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  {
    // The side stream runs what must not queue behind a device-filling launch of the main stream: the parse of the
    // next FASTQ chunk beside the fill of the chunk before (the fill's grid holds ~10^6 workgroups: at equal priority
    // the parse kernels were dispatched behind them and the chunk's chain was fill + parse, round 5), the decisions and
    // copy-back of a pipelined window beside the next window's query.  Highest priority the device offers.
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) {
      lo = hi = 0;
      (void)hipGetLastError();
    }
    CREATE_TRY(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi));
    // the copy stream at the LOWEST priority: streams of one priority may share a hardware queue, and the barrier packets
    // around a 5 ms DMA copy then hold back whatever else that queue carries (round 5: with both side streams at the
    // highest priority the parse kernels of a chunk started exactly when the copy of the chunk after next ended)
    CREATE_TRY(hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, lo));
  }
  CREATE_TRY(hipMalloc(&c->d_seeds, sizeof(DevSeeds)));
  CREATE_TRY(hipMemcpyAsync(c->d_seeds, &c->h_seeds, sizeof(DevSeeds), hipMemcpyHostToDevice, c->stream));
  if (p->m != 0) {
    c->f.m = p->m;
    c->f.m_inv = ~0ULL / p->m;
    // phase 1: plain bit vector (+3 zero pad words for the bucket builder)
    c->n_bv_words = (p->m + 31) / 32;
    CREATE_TRY(hipMalloc(&c->f.bv, (c->n_bv_words + 3) * sizeof(uint32_t)));
    CREATE_TRY(hipMemsetAsync(c->f.bv, 0, (c->n_bv_words + 3) * sizeof(uint32_t), c->stream));
  }
  for (QuerySlot& sl : c->slot) {
    CREATE_TRY(hipMalloc(&sl.d_qctr, 8 * sizeof(uint64_t)));
    if (const int arc = slot_dec_alloc(c, sl, 1024); arc != GRP_OK) {
      return fail(arc);
    }
    CREATE_TRY(hipMemsetAsync(sl.d_qctr, 0, 8 * sizeof(uint64_t), c->stream));
    CREATE_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&sl.qdone, hipEventDisableTiming));
    CREATE_TRY(hipMalloc(&sl.d_abort, 256)); // [0] abort flag, [32] tile dispenser (its own 128-byte line), [48] park
    CREATE_TRY(hipMemsetAsync(sl.d_abort, 0, 256, c->stream));
    CREATE_TRY(hipHostMalloc(&sl.h_abort, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&sl.dmap_abort), sl.h_abort, 0));
    *sl.h_abort = 0;
    CREATE_TRY(hipMalloc(&sl.d_sctl, SCT_WORDS * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(&sl.d_rel, (size_t)STREAM_REL_WGS * 32u * sizeof(uint32_t)));
    CREATE_TRY(hipHostMalloc(&sl.h_cmd, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&sl.dmap_cmd), sl.h_cmd, 0));
    CREATE_TRY(hipHostMalloc(&sl.h_ack, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&sl.dmap_ack), sl.h_ack, 0));
    memset(sl.h_cmd, 0, 64);
    memset(sl.h_ack, 0, 64);
    CREATE_TRY(hipMalloc(&sl.d_executed, 8 * sizeof(unsigned long long)));
    CREATE_TRY(hipHostMalloc(&sl.h_executed, 8 * sizeof(unsigned long long), hipHostMallocDefault));
  }
  CREATE_TRY(hipHostMalloc(&c->h_lists, LIST_PREFIX * sizeof(grp_id_count), hipHostMallocDefault));
  CREATE_TRY(hipHostMalloc(&c->h_small_tiles, SMALL_TILES * sizeof(grp_tile_summary), hipHostMallocMapped | hipHostMallocCoherent));
  CREATE_TRY(hipHostMalloc(&c->h_small_lists, (size_t)SMALL_TILES * SMALL_STRIDE * sizeof(grp_id_count), hipHostMallocMapped | hipHostMallocCoherent));
  CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->dmap_small_tiles), c->h_small_tiles, 0));
  CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->dmap_small_lists), c->h_small_lists, 0));
  CREATE_TRY(hipStreamSynchronize(c->stream));
#undef CREATE_TRY
  *out = c;
  return GRP_OK;
}

void
grp_destroy(grp_ctx* c)
{
  if (!c) {
    return;
  }
  if (getenv("GRP_DEBUG_STATS")) {
    fprintf(stderr, "grp stats: direct windows %llu (fallbacks %llu), general windows %llu, redo launches %llu, flagged tiles %llu, FASTQ prefetches dropped %llu\n", (unsigned long long)c->n_direct_windows,
            (unsigned long long)c->n_direct_fallbacks, (unsigned long long)c->n_general_windows, (unsigned long long)c->n_redo_launches, (unsigned long long)c->n_flagged_tiles, (unsigned long long)c->ingest.n_dropped);
  }
  if (c->stream3) {
    (void)hipStreamSynchronize(c->stream3);
  }
  if (c->stream2) {
    (void)hipStreamSynchronize(c->stream2);
  }
  if (c->stream) {
    (void)hipStreamSynchronize(c->stream);
  }
  if (c->pre.worker.joinable()) {
    c->pre.worker.join();
  }
  (void)hipFree(c->pre.units);
  (void)hipFree(c->pre.far);
  comm_release(c);
  if (c->reg_text) {
    (void)hipHostUnregister(const_cast<char*>(c->reg_text));
  }
  drain_events(c);
  for (auto& ep : c->free_events) {
    (void)hipEventDestroy(ep.a);
    (void)hipEventDestroy(ep.b);
  }
  for (QuerySlot& sl : c->slot) {
    (void)hipFree(sl.d_tiles);
    (void)hipFree(sl.d_lists);
    (void)hipFree(sl.d_qctr);
    (void)hipFree(sl.d_flag_idx);
    (void)hipFree(sl.d_dec_ids);
    (void)hipFree(sl.d_dec_asg);
    (void)hipFree(sl.d_dec_scratch);
    (void)hipFree(sl.d_long_reads);
    (void)hipFree(sl.d_dec_raw);
    if (sl.h_dec_raw) {
      (void)hipHostFree(sl.h_dec_raw);
    }
    if (sl.done) {
      (void)hipEventDestroy(sl.done);
    }
    if (sl.qdone) {
      (void)hipEventDestroy(sl.qdone);
    }
    if (sl.h_sdec) {
      (void)hipHostFree(sl.h_sdec);
    }
    if (sl.h_abort) {
      (void)hipHostFree(sl.h_abort);
    }
    if (sl.h_stripe_tiles) {
      (void)hipHostFree(sl.h_stripe_tiles);
    }
    (void)hipFree(sl.d_stripe_tiles);
    (void)hipFree(sl.d_abort);
    if (sl.h_executed) {
      (void)hipHostFree(sl.h_executed);
    }
    (void)hipFree(sl.d_tiles_done);
    (void)hipFree(sl.d_executed);
    (void)hipFree(sl.d_sctl);
    (void)hipFree(sl.d_tile_ver);
    (void)hipFree(sl.d_dbg);
    (void)hipFree(sl.d_rel);
    if (sl.h_cmd) {
      (void)hipHostFree(sl.h_cmd);
    }
    if (sl.h_ack) {
      (void)hipHostFree(sl.h_ack);
    }
  }
  {
    BatchRun& b = c->batch;
    (void)hipFree(b.d_counters);
    (void)hipFree(b.d_ins);
    (void)hipFree(b.d_rec_key);
    (void)hipFree(b.d_rec_loc);
    (void)hipFree(b.d_rec_old);
    (void)hipFree(b.d_rec_chain);
    (void)hipFree(b.d_ovf_next);
    (void)hipFree(b.d_ovf_jb);
    (void)hipFree(b.d_chained);
    (void)hipFree(b.d_log_keys);
    (void)hipFree(b.d_log_head);
    (void)hipFree(b.d_log_bits);
    (void)hipFree(b.d_log_old);
    (void)hipFree(b.d_log_writer);
    (void)hipFree(b.d_log_next);
    (void)hipFree(b.d_log_slot);
    (void)hipFree(b.d_floor);
    (void)hipFree(b.d_vf_stage);
    if (b.h_vf_stage) {
      (void)hipHostFree(b.h_vf_stage);
    }
    if (b.h_counters) {
      (void)hipHostFree(b.h_counters);
    }
    if (b.h_ins_stage) {
      (void)hipHostFree(b.h_ins_stage);
    }
  }
  (void)hipFree(c->d_ntc);
  (void)hipFree(c->d_ntc_chunks);
  (void)hipFree(c->d_ntc_extra);
  (void)hipFree(c->d_seeds);
  (void)hipFree(c->d_gtab);
  (void)hipFree(c->f.bv);
  (void)hipFree(c->f.buckets);
  (void)hipFree(c->d_super);
  (void)hipFree(c->f.far);
  (void)hipFree(c->f.ovf_keys);
  (void)hipFree(c->f.ovf_ids);
  if (c->h_tiles) {
    (void)hipHostFree(c->h_tiles);
  }
  if (c->h_lists) {
    (void)hipHostFree(c->h_lists);
  }
  if (c->h_small_tiles) {
    (void)hipHostFree(c->h_small_tiles);
  }
  if (c->h_small_lists) {
    (void)hipHostFree(c->h_small_lists);
  }
  (void)hipFree(c->d_delog);
  (void)hipFree(c->ovl.d_samples);
  (void)hipFree(c->ovl.d_n);
  (void)hipFree(c->ovl.d_tab);
  (void)hipFree(c->ovl.d_prev);
  for (int i = 0; i < FQ_TEXT_SLOTS; ++i) {
    (void)hipFree(c->ingest.text[i]);
  }
  (void)hipFree(c->ingest.d_counts);
  (void)hipFree(c->ingest.d_base);
  (void)hipFree(c->ingest.d_super);
  (void)hipFree(c->ingest.d_total);
  (void)hipFree(c->ingest.d_nl);
  (void)hipFree(c->ingest.d_rec);
  (void)hipFree(c->ingest.d_so);
  (void)hipFree(c->ingest.d_wo);
  (void)hipFree(c->ingest.d_len);
  for (hipEvent_t& e : c->ingest.text_done) {
    if (e) {
      (void)hipEventDestroy(e);
      e = nullptr;
    }
  }
  if (c->ingest.h_scal) {
    (void)hipHostFree(c->ingest.h_scal);
  }
  if (c->ingest.h_front) {
    (void)hipHostFree(c->ingest.h_front);
  }
  if (c->ingest.h_rec) {
    (void)hipHostFree(c->ingest.h_rec);
  }
  for (hipEvent_t& e : c->ingest.text_up) {
    if (e) {
      (void)hipEventDestroy(e);
      e = nullptr;
    }
  }
  if (c->ingest.uploaded) {
    (void)hipEventDestroy(c->ingest.uploaded);
  }
  (void)hipFree(c->d_dedup);
  (void)hipFree(c->d_ir_keys);
  (void)hipFree(c->d_ir_masks);
  (void)hipFree(c->d_ir_locs);
  (void)hipFree(c->d_ir_slots);
  (void)hipFree(c->d_ir_counter);
  (void)hipFree(c->d_fp_tab[0]);
  (void)hipFree(c->d_fp_tab[1]);
  if (c->stream3) {
    (void)hipStreamDestroy(c->stream3);
  }
  if (c->stream2) {
    (void)hipStreamDestroy(c->stream2);
  }
  if (c->stream) {
    (void)hipStreamDestroy(c->stream);
  }
  delete c;
}

// ---- reads -------------------------------------------------------------------------

static int
reads_build(grp_ctx* c, grp_reads* r, const uint64_t* word_off, const uint32_t* len, uint32_t n)
{
  const uint32_t tile = c->params.tile;
  const uint32_t k = c->params.k;
  const uint32_t min_len = c->params.k + c->params.h - 1;
  r->ctx = c;
  r->n_reads = n;
  r->n_words = word_off[n];
  r->len.assign(len, len + n);
  r->tile0.resize((size_t)n + 1);
  r->chunk0.resize((size_t)n + 1);
  uint64_t t = 0, ch = 0;
  for (uint32_t i = 0; i < n; ++i) {
    r->tile0[i] = t;
    r->chunk0[i] = ch;
    t += len[i] / tile;
    // reads shorter than the longest span are outside the reference's defined
    // behaviour (SeedNtHash on a too-short string); they contribute nothing
    if (len[i] >= min_len) {
      uint64_t npos = (uint64_t)len[i] - k + 1;
      ch += (npos + FILL_CHUNK - 1) / FILL_CHUNK;
    }
    if ((uint64_t)(len[i] + 15u) / 16u > word_off[i + 1] - word_off[i]) {
      return set_err(c, GRP_ERR_INVALID, "read %u: %u bases do not fit its %llu words", i, len[i], (unsigned long long)(word_off[i + 1] - word_off[i]));
    }
  }
  r->tile0[n] = t;
  r->chunk0[n] = ch;
  std::vector<uint32_t> tile_read(t), chunk_read(ch);
  for (uint32_t i = 0; i < n; ++i) {
    for (uint64_t j = r->tile0[i]; j < r->tile0[i + 1]; ++j) {
      tile_read[j] = i;
    }
    for (uint64_t j = r->chunk0[i]; j < r->chunk0[i + 1]; ++j) {
      chunk_read[j] = i;
    }
  }
  HIP_TRY(c, hipMalloc(&r->d_word_off, ((size_t)n + 1) * 8));
  HIP_TRY(c, hipMalloc(&r->d_len, std::max<size_t>(n, 1) * 4));
  HIP_TRY(c, hipMalloc(&r->d_tile0, ((size_t)n + 1) * 8));
  HIP_TRY(c, hipMalloc(&r->d_chunk0, ((size_t)n + 1) * 8));
  HIP_TRY(c, hipMalloc(&r->d_tile_read, std::max<size_t>(t, 1) * 4));
  HIP_TRY(c, hipMalloc(&r->d_chunk_read, std::max<size_t>(ch, 1) * 4));
  HIP_TRY(c, hipMemcpy(r->d_word_off, word_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(r->d_len, len, (size_t)n * 4, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(r->d_tile0, r->tile0.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(r->d_chunk0, r->chunk0.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(r->d_tile_read, tile_read.data(), t * 4, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(r->d_chunk_read, chunk_read.data(), ch * 4, hipMemcpyHostToDevice));
  r->dev.packed = r->d_packed;
  r->dev.word_off = r->d_word_off;
  r->dev.len = r->d_len;
  r->dev.tile0 = r->d_tile0;
  r->dev.tile_read = r->d_tile_read;
  r->dev.chunk0 = r->d_chunk0;
  r->dev.chunk_read = r->d_chunk_read;
  return GRP_OK;
}

// the host half of reads_build for grp_fastq_pack: lengths, tile / fill-chunk prefixes and the tile -> read, chunk -> read
// maps, kept in the batch (its asynchronous uploads read them)
int
reads_index(grp_ctx* c, grp_reads* r, const uint64_t* word_off, const uint32_t* len, uint32_t n)
{
  const uint32_t tile = c->params.tile;
  const uint32_t k = c->params.k;
  const uint32_t min_len = c->params.k + c->params.h - 1;
  r->ctx = c;
  r->n_reads = n;
  r->n_words = word_off[n];
  r->len.assign(len, len + n);
  r->tile0.resize((size_t)n + 1);
  r->chunk0.resize((size_t)n + 1);
  uint64_t t = 0, ch = 0;
  for (uint32_t i = 0; i < n; ++i) {
    r->tile0[i] = t;
    r->chunk0[i] = ch;
    t += len[i] / tile;
    if (len[i] >= min_len) { // (see reads_build)
      const uint64_t npos = (uint64_t)len[i] - k + 1;
      ch += (npos + FILL_CHUNK - 1) / FILL_CHUNK;
    }
  }
  r->tile0[n] = t;
  r->chunk0[n] = ch;
  r->h_tile_read.resize(t);
  r->h_chunk_read.resize(ch);
  for (uint32_t i = 0; i < n; ++i) {
    for (uint64_t j = r->tile0[i]; j < r->tile0[i + 1]; ++j) {
      r->h_tile_read[j] = i;
    }
    for (uint64_t j = r->chunk0[i]; j < r->chunk0[i + 1]; ++j) {
      r->h_chunk_read[j] = i;
    }
  }
  return GRP_OK;
}

int
grp_reads_upload(grp_ctx* c, const uint32_t* packed, const uint64_t* word_off, const uint32_t* len, uint32_t n, grp_reads** out)
{
  if (!c || !out || !word_off || (!len && n) || (!packed && n && word_off[n])) {
    return set_err(c, GRP_ERR_INVALID, "grp_reads_upload: null argument");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  grp_reads* r = new grp_reads();
  r->owns_packed = true;
  hipError_t e = hipMalloc(&r->d_packed, std::max<uint64_t>(word_off[n], 1) * 4 + 16);
  if (e != hipSuccess) {
    delete r;
    return set_err(c, GRP_ERR_NOMEM, "hipMalloc(%llu words) failed: %s", (unsigned long long)word_off[n], hipGetErrorString(e));
  }
  e = hipMemcpy(r->d_packed, packed, word_off[n] * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    grp_reads_free(r);
    return set_err(c, GRP_ERR_HIP, "hipMemcpy(packed) failed: %s", hipGetErrorString(e));
  }
  int rc = reads_build(c, r, word_off, len, n);
  if (rc != GRP_OK) {
    grp_reads_free(r);
    return rc;
  }
  *out = r;
  return GRP_OK;
}

int
grp_reads_wrap_device(grp_ctx* c, const void* d_packed, const uint64_t* word_off, const uint32_t* len, uint32_t n, grp_reads** out)
{
  if (!c || !out || !word_off || (!len && n) || !d_packed) {
    return set_err(c, GRP_ERR_INVALID, "grp_reads_wrap_device: null argument");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  grp_reads* r = new grp_reads();
  r->owns_packed = false;
  r->d_packed = const_cast<uint32_t*>(static_cast<const uint32_t*>(d_packed));
  int rc = reads_build(c, r, word_off, len, n);
  if (rc != GRP_OK) {
    grp_reads_free(r);
    return rc;
  }
  *out = r;
  return GRP_OK;
}

void
grp_reads_free(grp_reads* r)
{
  if (!r) {
    return;
  }
  if (r->ctx) {
    // a pipelined window that was abandoned may still run its decision kernel and copy-back on
    // the second stream; both read this batch's arrays
    if (r->ctx->stream2) {
      (void)hipStreamSynchronize(r->ctx->stream2);
    }
    if (r->ctx->stream) {
      (void)hipStreamSynchronize(r->ctx->stream);
    }
    for (QuerySlot& sl : r->ctx->slot) {
      if (sl.reads == r && !sl.busy) {
        sl.reads = nullptr;
      }
    }
  }
  if (r->owns_packed) {
    (void)hipFree(r->d_packed);
  }
  if (r->d_slab) { // the index arrays are parts of one allocation
    (void)hipFree(r->d_slab);
    delete r;
    return;
  }
  (void)hipFree(r->d_word_off);
  (void)hipFree(r->d_len);
  (void)hipFree(r->d_tile0);
  (void)hipFree(r->d_tile_read);
  (void)hipFree(r->d_chunk0);
  (void)hipFree(r->d_chunk_read);
  delete r;
}

const uint64_t*
grp_reads_tile0(const grp_reads* r)
{
  return r ? r->tile0.data() : nullptr;
}

// ---- fill ---------------------------------------------------------------------------

// buckets (128-byte units) and far-table slots a filter of m bits at occupancy `occ` needs: W as grp_finalize picks it
// from the measured occupancy, the far entries from the binomial bucket population (E[max(c - 8, 0)] per bucket)
static void
prealloc_sizes(uint64_t m, double occ, uint64_t& n_buckets, uint64_t& far_cap)
{
  // a W slightly below what the hinted occupancy gives: a measured occupancy a few percent above the hint still fits
  const double w = 6.0 / (occ * 1.04);
  const uint32_t W = std::max<uint32_t>(GRP_BUCKET_IDS, std::min<uint32_t>(w >= 64.0 ? 64u : (uint32_t)w, 64u));
  n_buckets = (m + W - 1) / W;
  double pk = std::pow(1.0 - occ, (double)W), tail = 0.0; // P(c = 0)
  for (uint32_t k = 0; k <= W; ++k) {
    if (k > GRP_NEAR_CW) {
      tail += pk * (double)(k - GRP_NEAR_CW);
    }
    pk = pk * (double)(W - k) / (double)(k + 1) * occ / (1.0 - occ);
  }
  const uint64_t expect_far = (uint64_t)(tail * (double)n_buckets * 1.10) + 1024;
  far_cap = 1024;
  while (far_cap < 2 * expect_far) {
    far_cap <<= 1;
  }
}

// called with the first fill launches enqueued: the allocations run beside them
static void
prealloc_start(grp_ctx* c)
{
  if (c->pre.started || c->pre.occupancy <= 0.0 || c->f.m == 0 || c->finalized) {
    return;
  }
  c->pre.started = true;
  uint64_t nb = 0, fc = 0;
  prealloc_sizes(c->f.m, c->pre.occupancy, nb, fc);
  const int device = c->device;
  grp_ctx::Prealloc* p = &c->pre;
  const uint64_t bv_bytes = c->f.m / 8;
  c->pre.worker = std::thread([p, nb, fc, device, bv_bytes] {
    if (hipSetDevice(device) != hipSuccess) {
      return;
    }
    // Only where the tables fit BESIDE what phase 1 may still need: two more bit vectors (the ranks' merge,
    // grp_bv_merge_ranks) and 16 GB for the batches of reads that pass through — C2 has the room, C4's 246 GB of tables
    // have not (grp_finalize then allocates when the fill is through, as it did until round 5).
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    const uint64_t need = nb * (uint64_t)GRP_UNIT_U4 * 16 + fc * sizeof(ulonglong2);
    if ((uint64_t)free_b < need + 2 * bv_bytes + (16ull << 30)) {
      return;
    }
    uint4* u = nullptr;
    if (hipMalloc(&u, nb * (uint64_t)GRP_UNIT_U4 * 16) != hipSuccess) {
      (void)hipGetLastError();
      return; // grp_finalize allocates what it needs (and reports the failure if there is one)
    }
    ulonglong2* f = nullptr;
    if (hipMalloc(&f, fc * sizeof(ulonglong2)) != hipSuccess) {
      (void)hipGetLastError();
      f = nullptr;
    }
    p->units = u;
    p->units_buckets = nb;
    p->far = f;
    p->far_cap = f ? fc : 0;
  });
}

int
grp_set_occupancy_hint(grp_ctx* c, double occupancy)
{
  if (!c || !(occupancy > 0.0 && occupancy < 1.0)) {
    return set_err(c, GRP_ERR_INVALID, "grp_set_occupancy_hint: occupancy must lie in (0, 1)");
  }
  if (c->finalized || c->pre.started) {
    return set_err(c, GRP_ERR_STATE, "grp_set_occupancy_hint: the tables are being prepared already (call it before the first grp_bv_insert)");
  }
  c->pre.occupancy = occupancy;
  return GRP_OK;
}

int
grp_debug_finalize_times(const grp_ctx* c, double out[6])
{
  if (!c || !out) {
    return GRP_ERR_INVALID;
  }
  memcpy(out, c->finalize_times, sizeof(c->finalize_times));
  return GRP_OK;
}

int
grp_set_filter_size(grp_ctx* c, uint64_t m)
{
  if (!c || m < 64 || m >= (1ULL << 50)) {
    return set_err(c, GRP_ERR_INVALID, "grp_set_filter_size: filter size m=%llu unsupported", (unsigned long long)m);
  }
  if (c->f.m != 0) {
    return set_err(c, GRP_ERR_STATE, "grp_set_filter_size: the filter already has m=%llu bits", (unsigned long long)c->f.m);
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t words = (m + 31) / 32;
  HIP_TRY(c, hipMalloc(&c->f.bv, (words + 3) * sizeof(uint32_t)));
  HIP_TRY(c, hipMemsetAsync(c->f.bv, 0, (words + 3) * sizeof(uint32_t), c->stream));
  c->params.m = m;
  c->f.m = m;
  c->f.m_inv = ~0ULL / m;
  c->n_bv_words = words;
  return GRP_OK;
}

int
grp_bv_insert(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count)
{
  if (!c || !r || r->ctx != c || (uint64_t)first + count > r->n_reads) {
    return set_err(c, GRP_ERR_INVALID, "grp_bv_insert: bad range");
  }
  if (c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_bv_insert after grp_finalize: the bit vector is immutable");
  }
  if (c->f.m == 0) {
    return set_err(c, GRP_ERR_STATE, "grp_bv_insert: the filter size is not set (grp_set_filter_size)");
  }
  // re-entrant: the enqueue (stream, event bookkeeping) is serialised here; the launches of
  // different callers may interleave on the stream in any order — the fill is order-free
  std::lock_guard<std::mutex> lock(c->fill_mutex);
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t ch0 = r->chunk0[first], ch1 = r->chunk0[first + count];
  if (ch1 == ch0) {
    return GRP_OK;
  }
  const size_t lds = tab_bytes(c) + bases_bytes(FILL_CHUNK + c->params.k + c->params.h);
  uint64_t probes = 0;
  for (uint32_t i = first; i < first + count; ++i) {
    for (uint32_t s = 0; s < c->params.h; ++s) {
      uint32_t span = c->params.k + s;
      if (r->len[i] >= c->params.k + c->params.h - 1) {
        probes += r->len[i] - span + 1;
      }
    }
  }
  // launch in slices of at most MAX_GRID_WGS workgroups
  for (uint64_t b = ch0; b < ch1;) {
    uint32_t nb = (uint32_t)std::min<uint64_t>(ch1 - b, MAX_GRID_WGS);
    Timer t(c, GRP_K_FILL, b == ch0 ? probes : 0);
    DISPATCH_H(c->params.h, (k_fill<HH><<<dim3(nb), dim3(THREADS), lds, c->stream>>>(c->f, r->dev, c->d_seeds, b)));
    HIP_TRY(c, hipGetLastError());
    b += nb;
  }
  prealloc_start(c); // (the first call with work: the phase-2 tables are allocated beside the fill, grp_set_occupancy_hint)
  return GRP_OK;
}

int
grp_bv_words(const grp_ctx* c, uint64_t* n_words32)
{
  if (!c || !n_words32) {
    return GRP_ERR_INVALID;
  }
  *n_words32 = c->n_bv_words;
  return GRP_OK;
}

int
grp_bv_export_device(grp_ctx* c, void* d_dst)
{
  if (!c || !d_dst) {
    return GRP_ERR_INVALID;
  }
  if (c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_bv_export_device after grp_finalize");
  }
  if (c->f.m == 0) {
    return set_err(c, GRP_ERR_STATE, "grp_bv_export_device: the filter size is not set (grp_set_filter_size)");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(d_dst, c->f.bv, c->n_bv_words * 4, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return GRP_OK;
}

int
grp_bv_merge_device(grp_ctx* c, const void* d_src)
{
  if (!c || !d_src) {
    return GRP_ERR_INVALID;
  }
  if (c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_bv_merge_device after grp_finalize: the bit vector is immutable");
  }
  if ((reinterpret_cast<uintptr_t>(d_src) & 15u) != 0) {
    return set_err(c, GRP_ERR_INVALID, "grp_bv_merge_device: source must be 16-byte aligned");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t n4 = c->n_bv_words / 4;
  const uint32_t tail = (uint32_t)(c->n_bv_words - n4 * 4);
  k_bv_or<<<dim3(4096), dim3(THREADS), 0, c->stream>>>(reinterpret_cast<uint4*>(c->f.bv), static_cast<const uint4*>(d_src), n4, c->f.bv + n4 * 4, static_cast<const uint32_t*>(d_src) + n4 * 4, tail);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return GRP_OK;
}

int
grp_words_or_device(grp_ctx* c, void* d_dst, const void* d_src, uint64_t n_words32)
{
  if (!c || !d_dst || !d_src || ((reinterpret_cast<uintptr_t>(d_dst) | reinterpret_cast<uintptr_t>(d_src)) & 15u) != 0) {
    return set_err(c, GRP_ERR_INVALID, "grp_words_or_device: null or unaligned buffer");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t n4 = n_words32 / 4;
  const uint32_t tail = (uint32_t)(n_words32 - n4 * 4);
  k_bv_or<<<dim3(4096), dim3(THREADS), 0, c->stream>>>(static_cast<uint4*>(d_dst), static_cast<const uint4*>(d_src), n4, static_cast<uint32_t*>(d_dst) + n4 * 4, static_cast<const uint32_t*>(d_src) + n4 * 4, tail);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return GRP_OK;
}

int
grp_bv_import_device(grp_ctx* c, const void* d_src)
{
  if (!c || !d_src) {
    return GRP_ERR_INVALID;
  }
  if (c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_bv_import_device after grp_finalize: the bit vector is immutable");
  }
  if (c->f.m == 0) {
    return set_err(c, GRP_ERR_STATE, "grp_bv_import_device: the filter size is not set (grp_set_filter_size)");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(c->f.bv, d_src, c->n_bv_words * 4, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return GRP_OK;
}

int
grp_finalize(grp_ctx* c, uint64_t* pop)
{
  if (!c) {
    return GRP_ERR_INVALID;
  }
  if (c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_finalize called twice");
  }
  if (c->f.m == 0) {
    return set_err(c, GRP_ERR_STATE, "grp_finalize: the filter size is not set (grp_set_filter_size)");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const auto t_begin = now();
  for (double& v : c->finalize_times) {
    v = 0.0;
  }
  unsigned long long* d_scalars = nullptr; // [0] pop (popcount), [1] pop (scan), [2] overflow entries (IDs), [3] far entries (count words)
  HIP_TRY(c, hipMalloc(&d_scalars, 4 * sizeof(unsigned long long)));
  HIP_TRY(c, hipMemsetAsync(d_scalars, 0, 4 * sizeof(unsigned long long), c->stream));
  Timer* t = new Timer(c, GRP_K_RANK, c->n_bv_words);
  k_popcount<<<dim3(4096), dim3(THREADS), 0, c->stream>>>(c->f.bv, c->n_bv_words, d_scalars);
  unsigned long long h_scalars[4] = { 0, 0, 0, 0 };
  HIP_TRY(c, hipMemcpyAsync(h_scalars, d_scalars, sizeof(h_scalars), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const auto t_pop = now();
  c->finalize_times[0] = secs(t_begin, t_pop);
  const uint64_t h_pop = h_scalars[0];
  // bucket width: ~6 set bits per bucket on average at the measured occupancy;
  // with W <= 13 a bucket can never overflow its 13 ID slots
  const double occ = (double)h_pop / (double)c->f.m;
  uint32_t W = 64;
  if (occ > 0) {
    const double w = 6.0 / occ;
    W = (w >= 64.0) ? 64u : (uint32_t)w;
  }
  W = std::max<uint32_t>(GRP_BUCKET_IDS, std::min<uint32_t>(W, 64u));
  c->f.W = W;
  c->f.w_magic = (uint64_t)((((unsigned __int128)1) << 64) / W) + 1;
  c->f.n_buckets = (c->f.m + W - 1) / W;
  for (uint64_t q : { (uint64_t)0, (uint64_t)1, c->f.n_buckets / 2, c->f.n_buckets - 1 }) { // the division shortcut is exact
    for (uint64_t d : { (uint64_t)0, (uint64_t)W - 1 }) {
      const uint64_t pos = q * W + d;
      if (grp_div_w(pos, c->f.w_magic) != q) {
        delete t;
        return set_err(c, GRP_ERR_INVALID, "internal: bucket division is not exact for W=%u", W);
      }
    }
  }
  c->n_chunks = (c->f.n_buckets + GRP_CHUNK_BUCKETS - 1) / GRP_CHUNK_BUCKETS;
  c->nsb = ((c->f.n_buckets - 1) >> GRP_SUPER_SHIFT) + 1;
  if (c->n_chunks > MAX_GRID_WGS) {
    delete t;
    return set_err(c, GRP_ERR_INVALID, "filter of %llu buckets exceeds the rank builder's launch size", (unsigned long long)c->f.n_buckets);
  }
  uint32_t* d_chunk_sum = nullptr;
  uint64_t* d_chunk_base = nullptr;
  // a bucket is a 128-byte unit since round 5: its query line {rel, bitmap, 13 IDs} and its insert line (8 count words);
  // the count words of the ranks beyond a bucket's 8th set bit come on top (the far table below, ~5 % of the ranks x 32 B)
  const uint64_t unit_bytes = (uint64_t)GRP_UNIT_U4 * 16;
  // Round 6: the tables prepared beside the fill (grp_set_occupancy_hint) are taken where they are large enough — the
  // measured occupancy picks W and the bucket count as ever, a prepared table only has to hold them.  hipMalloc of
  // 130 GB + 8 GB took ~4 s of host time here, with the device idle.
  if (c->pre.worker.joinable()) {
    c->pre.worker.join();
  }
  bool used_prepared = false;
  if (c->pre.units && c->pre.units_buckets >= c->f.n_buckets) {
    c->f.buckets = c->pre.units;
    c->pre.units = nullptr;
    used_prepared = true;
  } else {
    if (c->pre.units) { // prepared for another geometry (the occupancy came out far above the hint): released first
      (void)hipFree(c->pre.units);
      c->pre.units = nullptr;
    }
    const hipError_t e = hipMalloc(&c->f.buckets, c->f.n_buckets * unit_bytes);
    if (e != hipSuccess) {
      delete t;
      return set_err(c, GRP_ERR_NOMEM, "hipMalloc of %llu bucket units (%.1f GB: 128 B per %u filter bits, IDs and insert counts included) failed: %s", (unsigned long long)c->f.n_buckets, c->f.n_buckets * unit_bytes / 1e9, W,
                     hipGetErrorString(e));
    }
  }
  HIP_TRY(c, hipMalloc(&c->d_super, c->nsb * sizeof(uint64_t)));
  c->f.super = c->d_super;
  HIP_TRY(c, hipMalloc(&d_chunk_sum, c->n_chunks * 4));
  HIP_TRY(c, hipMalloc(&d_chunk_base, c->n_chunks * 8));
  const auto t_alloc = now();
  c->finalize_times[1] = secs(t_pop, t_alloc);
  k_bucket_chunk_sums<<<dim3((uint32_t)c->n_chunks), dim3(THREADS), 0, c->stream>>>(c->f.bv, c->f.m, W, c->f.n_buckets, d_chunk_sum);
  k_scan_chunks<<<dim3(1), dim3(1024), 0, c->stream>>>(d_chunk_sum, c->n_chunks, d_chunk_base, c->d_super, reinterpret_cast<uint64_t*>(d_scalars + 1));
  k_bucket_write<<<dim3((uint32_t)c->n_chunks), dim3(THREADS), 0, c->stream>>>(c->f.bv, c->f.m, W, c->f.n_buckets, c->f.buckets, d_chunk_base, c->d_super, d_scalars + 2);
  delete t;
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(h_scalars, d_scalars, sizeof(h_scalars), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (h_scalars[1] != h_pop) {
    return set_err(c, GRP_ERR_HIP, "internal: bucket scan counted %llu set bits, popcount %llu", h_scalars[1], (unsigned long long)h_pop);
  }
  {
    // the bits now live in the buckets.  (Measured in round 6 before deciding where this belongs: hipFree waits for the
    // device — idle here — and unmaps 7.6 GB at C2; finalize_times[5] has what it takes.  A helper thread would free it
    // behind the caller's back, but hipFree synchronises with EVERY later launch too: a parked streaming window that waits
    // for the host while the host's next HIP call waits for the free is a deadlock this engine has met before)
    const auto t_free = now();
    (void)hipFree(c->f.bv);
    c->f.bv = nullptr;
    (void)hipFree(d_chunk_sum);
    (void)hipFree(d_chunk_base);
    (void)hipFree(d_scalars);
    c->finalize_times[5] = secs(t_free, now());
  }
  c->f.pop = h_pop;
  const auto t_rank = now();
  c->finalize_times[2] = secs(t_alloc, t_rank);
  {
    // the count words beyond a bucket's 8th set bit: {rank + 1, word}, open addressing at load <= 1/2, keys written now
    const uint64_t far_cap = next_pow2_64(std::max<uint64_t>(2 * h_scalars[3], 1024));
    if (c->pre.far && c->pre.far_cap >= far_cap) {
      // (a prepared table larger than needed keeps its size: the load only drops)
      c->f.far = c->pre.far;
      c->f.far_mask = c->pre.far_cap - 1;
      c->pre.far = nullptr;
    } else {
      used_prepared = false;
      if (c->pre.far) {
        (void)hipFree(c->pre.far);
        c->pre.far = nullptr;
      }
      const hipError_t e = hipMalloc(&c->f.far, far_cap * sizeof(ulonglong2));
      if (e != hipSuccess) {
        return set_err(c, GRP_ERR_NOMEM, "hipMalloc of the far count table (%llu ranks beyond their bucket's 8th set bit, %.1f GB) failed: %s", h_scalars[3], far_cap * sizeof(ulonglong2) / 1e9, hipGetErrorString(e));
      }
      c->f.far_mask = far_cap - 1;
    }
    HIP_TRY(c, hipMemsetAsync(c->f.far, 0, (c->f.far_mask + 1) * sizeof(ulonglong2), c->stream));
    c->n_far = h_scalars[3];
    if (h_scalars[3]) {
      k_far_build<<<dim3((uint32_t)std::min<uint64_t>((c->f.n_buckets + THREADS - 1) / THREADS, 65536)), dim3(THREADS), 0, c->stream>>>(c->f);
      HIP_TRY(c, hipGetLastError());
    }
  }
  const uint64_t ovf_cap = next_pow2_64(std::max<uint64_t>(2 * h_scalars[2], 1024));
  HIP_TRY(c, hipMalloc(&c->f.ovf_keys, ovf_cap * sizeof(unsigned long long)));
  HIP_TRY(c, hipMalloc(&c->f.ovf_ids, ovf_cap * sizeof(uint32_t)));
  HIP_TRY(c, hipMemsetAsync(c->f.ovf_keys, 0, ovf_cap * sizeof(unsigned long long), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->f.ovf_ids, 0, ovf_cap * sizeof(uint32_t), c->stream));
  c->f.ovf_mask = ovf_cap - 1;
  c->n_ovf = h_scalars[2];
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->finalize_times[3] = secs(t_rank, now());
  c->finalize_times[4] = used_prepared ? 1.0 : 0.0;
  c->finalized = true;
  if (pop) {
    *pop = h_pop;
  }
  return GRP_OK;
}

// ---- query --------------------------------------------------------------------------

} // extern "C"

namespace {

struct QueryRun
{
  uint64_t t0 = 0, nt = 0, probes = 0;
  uint64_t used = 0;    // list arena entries
  uint64_t flagged = 0; // tiles redone with the worst-case geometry
};

template<typename T>
int
ensure_dev(grp_ctx* c, T*& p, uint64_t& cap, uint64_t want)
{
  if (want > cap) {
    (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const uint64_t n = want + want / 4 + 64;
    HIP_TRY(c, hipMalloc(&p, n * sizeof(T)));
    cap = n;
  }
  return GRP_OK;
}

uint64_t
count_probes(const grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count)
{
  const uint32_t tile = c->params.tile, k = c->params.k;
  uint64_t probes = 0;
  for (uint32_t i = first; i < first + count; ++i) {
    const uint32_t ntile = r->len[i] / tile;
    if (ntile == 0) {
      continue;
    }
    // all tiles but the last have `tile` frames; the last may be clipped
    const uint32_t start = (ntile - 1) * tile;
    const uint32_t Lp = std::min(tile + k - 1, r->len[i] - start);
    probes += ((uint64_t)(ntile - 1) * tile + (Lp - k + 1)) * c->params.h;
  }
  return probes;
}

// Enqueue the query kernel for the window (results stay in c->q->d_tiles /
// c->q->d_lists).  No synchronisation.
int
enqueue_query(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, uint64_t list_cap, QueryRun& q)
{
  q.t0 = r->tile0[first];
  q.nt = r->tile0[first + count] - q.t0;
  q.probes = count_probes(c, r, first, count);
  int rc = ensure_dev(c, c->q->d_tiles, c->q->d_tiles_cap, q.nt);
  if (rc == GRP_OK) {
    rc = ensure_dev(c, c->q->d_lists, c->q->d_lists_cap, std::max<uint64_t>(list_cap, 1));
  }
  if (rc == GRP_OK) {
    rc = ensure_dev(c, c->q->d_flag_idx, c->q->d_flag_cap, q.nt);
  }
  if (rc != GRP_OK) {
    return rc;
  }
  if (c->q->side_used) {
    // the slot's previous window may still be in its decision kernel on the other stream
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->q->done, 0));
    c->q->side_used = false;
  }
  const QueryGeom g = query_geom(c, false);
  Timer t(c, GRP_K_QUERY, q.probes);
  int lrc = GRP_OK;
  DISPATCH_H(c->params.h, lrc = launch_query<HH>(c, r, q.nt, q.t0, nullptr, g, list_cap));
  if (lrc != GRP_OK) {
    return lrc;
  }
  HIP_TRY(c, hipGetLastError());
  return GRP_OK;
}

// tiles whose ID set did not fit the small LDS table (indices collected on the
// device): redo them with the worst-case geometry — same kernel, same arithmetic.
// Expects h_qctr to hold the counters of the first launch.
int
enqueue_redo_flagged(grp_ctx* c, const grp_reads* r, uint64_t list_cap, QueryRun& q)
{
  q.flagged = c->q->h_qctr[4];
  c->n_flagged_tiles += q.flagged;
  c->n_flagged_distinct += c->q->h_qctr[1];
  c->n_flagged_list += c->q->h_qctr[2];
  ++c->n_redo_launches;
  // continue the list arena where the first launch stopped
  uint64_t cursor[8] = { 0, 0, 0, c->q->h_qctr[3], 0, 0, 0, 0 };
  HIP_TRY(c, hipMemcpyAsync(c->q->d_qctr, cursor, sizeof(cursor), hipMemcpyHostToDevice, c->stream));
  const QueryGeom g = query_geom(c, true);
  Timer t(c, GRP_K_QUERY, 0);
  int lrc = GRP_OK;
  DISPATCH_H(c->params.h, lrc = launch_query<HH>(c, r, q.flagged, q.t0, c->q->d_flag_idx, g, list_cap));
  if (lrc != GRP_OK) {
    return lrc;
  }
  HIP_TRY(c, hipGetLastError());
  return GRP_OK;
}

} // namespace

extern "C" {

int
grp_query_tiles(grp_ctx* c,
                const grp_reads* r,
                uint32_t first,
                uint32_t count,
                grp_tile_summary* tiles_out,
                grp_id_count* lists_out,
                uint64_t list_cap,
                uint64_t* list_used,
                grp_query_stats* stats)
{
  if (c) {
    c->carry.valid = false;
  }
  if (!c || !r || r->ctx != c || (uint64_t)first + count > r->n_reads) {
    return set_err(c, GRP_ERR_INVALID, "grp_query_tiles: bad range");
  }
  if (!c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_query_tiles before grp_finalize");
  }
  if (c->slot[0].busy) {
    return set_err(c, GRP_ERR_STATE, "grp_query_tiles while an asynchronous window is in flight in slot 0");
  }
  c->q = &c->slot[0];
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->q->side_used) {
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->q->done, 0));
    c->q->side_used = false;
  }
  const uint64_t nt = r->tile0[first + count] - r->tile0[first];
  if (list_used) {
    *list_used = 0;
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
  }
  if (nt == 0) {
    return GRP_OK;
  }
  if (nt > MAX_GRID_WGS) {
    return set_err(c, GRP_ERR_INVALID, "grp_query_tiles: %llu tiles in one call (limit 2^22)", (unsigned long long)nt);
  }
  if (!tiles_out) {
    return set_err(c, GRP_ERR_INVALID, "grp_query_tiles: tiles_out is NULL");
  }
  if (nt <= SMALL_TILES && !c->env_no_direct) { // (GRP_NO_DIRECT, developer hook: force the general path)
    // latency path (insert-heavy phases query one read at a time): the kernel
    // writes summaries and lists straight into mapped host memory, no copies,
    // no counters; one stream synchronisation
    const uint64_t t0 = r->tile0[first];
    const uint64_t probes = count_probes(c, r, first, count);
    constexpr uint32_t PENDING = 0xFFFFFFFEu; // list_n of a summary the kernel has not written yet
    for (uint64_t i = 0; i < nt; ++i) {
      c->h_small_tiles[i].list_n = PENDING;
    }
    {
      const QueryGeom g = query_geom(c, false);
      Timer t(c, GRP_K_QUERY_LAT, probes);
      int lrc = GRP_OK;
      DISPATCH_H(c->params.h, lrc = launch_query<HH>(c, r, nt, t0, nullptr, g, (uint64_t)SMALL_TILES * SMALL_STRIDE, c->dmap_small_tiles, c->dmap_small_lists, SMALL_STRIDE));
      if (lrc != GRP_OK) {
        return lrc;
      }
    }
    HIP_TRY(c, hipGetLastError());
    // the summaries arrive in host memory while the launch runs (lists first, the word with
    // list_n last): spin on them instead of waiting for the end-of-kernel signal; a launch
    // that never delivers falls back to the stream synchronisation
    {
      uint64_t i = 0;
      uint32_t spins = 0;
      while (i < nt) {
        if (__atomic_load_n(&c->h_small_tiles[i].list_n, __ATOMIC_ACQUIRE) != PENDING) {
          ++i;
          continue;
        }
        __builtin_ia32_pause();
        if ((++spins & 0xFFFFu) == 0 && hipStreamQuery(c->stream) != hipErrorNotReady) {
          HIP_TRY(c, hipStreamSynchronize(c->stream));
          for (uint64_t j = i; j < nt; ++j) {
            if (__atomic_load_n(&c->h_small_tiles[j].list_n, __ATOMIC_ACQUIRE) == PENDING) {
              return set_err(c, GRP_ERR_HIP, "grp_query_tiles: the launch ended without delivering tile %llu", (unsigned long long)j);
            }
          }
          break;
        }
      }
    }
    ++c->n_direct_windows;
    bool any_flagged = false;
    uint64_t used = 0;
    for (uint64_t i = 0; i < nt; ++i) {
      any_flagged = any_flagged || c->h_small_tiles[i].list_n == GRP_TILE_FLAGGED;
      used += c->h_small_tiles[i].list_n;
    }
    if (!any_flagged) {
      if (list_used) {
        *list_used = used;
      }
      if (used > list_cap) {
        return set_err(c, GRP_ERR_NOMEM, "grp_query_tiles: %llu list entries needed, capacity %llu", (unsigned long long)used, (unsigned long long)list_cap);
      }
      uint64_t off = 0;
      for (uint64_t i = 0; i < nt; ++i) {
        grp_tile_summary ts = c->h_small_tiles[i];
        grp_id_count* src = c->h_small_lists + i * SMALL_STRIDE;
        if (ts.list_n) {
          memcpy(lists_out + off, src, ts.list_n * sizeof(grp_id_count));
          if (ts.list_n > 1) {
            std::sort(lists_out + off, lists_out + off + ts.list_n, [](const grp_id_count& a, const grp_id_count& b) {
              return a.count != b.count ? a.count > b.count : a.id < b.id;
            });
          }
        }
        ts.list_off = (uint32_t)off;
        off += ts.list_n;
        tiles_out[i] = ts;
        if (stats) {
          stats->hits += ts.hits;
          stats->misses += ts.misses;
        }
      }
      if (stats) {
        stats->queries = probes / c->params.h;
      }
      return GRP_OK;
    }
    // a tile needs the worst-case table or a longer list: take the general path
    ++c->n_direct_fallbacks;
  }
  if (nt > c->h_tiles_cap) {
    if (c->h_tiles) {
      (void)hipHostFree(c->h_tiles);
      c->h_tiles = nullptr;
      c->h_tiles_cap = 0;
    }
    const uint64_t cap = std::max<uint64_t>(nt + nt / 4, 4096);
    HIP_TRY(c, hipHostMalloc(&c->h_tiles, cap * sizeof(grp_tile_summary), hipHostMallocDefault));
    c->h_tiles_cap = cap;
  }
  ++c->n_general_windows;
  QueryRun q;
  int rc = enqueue_query(c, r, first, count, list_cap, q);
  if (rc != GRP_OK) {
    return rc;
  }
  const uint64_t prefix = std::min<uint64_t>(list_cap, LIST_PREFIX);
  // one round trip: counters, tile summaries and the first LIST_PREFIX list
  // entries come back together; the counters are re-zeroed for the next call
  auto fetch = [&]() -> int {
    HIP_TRY(c, hipMemcpyAsync(c->q->h_qctr, c->q->d_qctr, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_tiles, c->q->d_tiles, nt * sizeof(grp_tile_summary), hipMemcpyDeviceToHost, c->stream));
    if (prefix) {
      HIP_TRY(c, hipMemcpyAsync(c->h_lists, c->q->d_lists, prefix * sizeof(grp_id_count), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(c->q->d_qctr, 0, 8 * sizeof(uint64_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GRP_OK;
  };
  rc = fetch();
  if (rc != GRP_OK) {
    return rc;
  }
  if (c->q->h_qctr[4]) {
    rc = enqueue_redo_flagged(c, r, list_cap, q);
    if (rc == GRP_OK) {
      rc = fetch();
    }
    if (rc != GRP_OK) {
      return rc;
    }
  }
  drain_events(c);
  memcpy(tiles_out, c->h_tiles, nt * sizeof(grp_tile_summary));
  const uint64_t used = c->q->h_qctr[3];
  if (list_used) {
    *list_used = used;
  }
  if (stats) {
    stats->queries = q.probes / c->params.h; /* one query per frame (goldrush_path.cpp:567-568) */
    for (uint64_t i = 0; i < nt; ++i) {
      stats->hits += tiles_out[i].hits;
      stats->misses += tiles_out[i].misses;
    }
  }
  if (used > list_cap) {
    return set_err(c, GRP_ERR_NOMEM, "grp_query_tiles: %llu list entries needed, capacity %llu", (unsigned long long)used, (unsigned long long)list_cap);
  }
  if (used) {
    if (!lists_out) {
      return set_err(c, GRP_ERR_INVALID, "grp_query_tiles: lists_out is NULL");
    }
    if (used <= prefix) {
      memcpy(lists_out, c->h_lists, used * sizeof(grp_id_count));
    } else {
      HIP_TRY(c, hipMemcpy(lists_out, c->q->d_lists, used * sizeof(grp_id_count), hipMemcpyDeviceToHost));
    }
    // canonical order inside each tile's list: count descending, id ascending
    for (uint64_t i = 0; i < nt; ++i) {
      grp_tile_summary& ts = tiles_out[i];
      if (ts.list_n > 1) {
        std::sort(lists_out + ts.list_off, lists_out + ts.list_off + ts.list_n, [](const grp_id_count& a, const grp_id_count& b) {
          return a.count != b.count ? a.count > b.count : a.id < b.id;
        });
      }
    }
  }
  return GRP_OK;
}

} // extern "C"

namespace {

int
classify_enqueue_decide(grp_ctx* c, QuerySlot& sl, hipStream_t st)
{
  Timer t(c, GRP_K_DECIDE, sl.count, st);
  // Reads of more than 256 tiles keep their state in LDS (up to 148 KB at 4096 tiles): sized for the longest read and
  // applied to the whole launch, ONE such read used to drop every workgroup of the window to about one per CU (ADVICE
  // r04).  They get a launch of their own over an index list; the common launch carries no dynamic LDS.
  const uint64_t* tile0 = sl.reads->tile0.data() + sl.first;
  std::vector<uint32_t>& longs = sl.h_long_reads;
  longs.clear();
  uint64_t most = 0;
  for (uint32_t i = 0; i < sl.count; ++i) {
    const uint64_t n = tile0[i + 1] - tile0[i];
    if (n > 4 * LANE_TILES) {
      longs.push_back(i);
      most = std::max(most, n);
    }
  }
  k_decide<<<dim3(sl.count), dim3(DECIDE_THREADS), 0, st>>>(sl.reads->dev, sl.first, sl.count, sl.dp, sl.d_tiles, sl.d_lists, sl.list_cap, sl.d_dec_ids, sl.d_dec_asg, sl.d_dec_scratch, sl.d_dec, 0u, nullptr,
                                                            longs.empty() ? 0u : 4u * LANE_TILES, reinterpret_cast<unsigned long long*>(sl.d_qctr), reinterpret_cast<unsigned long long*>(sl.d_dec_raw));
  HIP_TRY(c, hipGetLastError());
  if (!longs.empty()) {
    const uint32_t lds_tiles = (uint32_t)std::min<uint64_t>((most + 63) / 64 * 64, DECIDE_LDS_MAX_TILES);
    const size_t lds = decide_lds_bytes(lds_tiles);
    if (const int rc = ensure_lds(c, k_decide, lds); rc != GRP_OK) {
      return rc;
    }
    if (const int rc = ensure_dev(c, sl.d_long_reads, sl.long_reads_cap, longs.size()); rc != GRP_OK) {
      return rc;
    }
    HIP_TRY(c, hipMemcpyAsync(sl.d_long_reads, longs.data(), longs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st)); // (pageable: staged before the call returns)
    k_decide<<<dim3((uint32_t)longs.size()), dim3(DECIDE_THREADS), lds, st>>>(sl.reads->dev, sl.first, sl.count, sl.dp, sl.d_tiles, sl.d_lists, sl.list_cap, sl.d_dec_ids, sl.d_dec_asg, sl.d_dec_scratch, sl.d_dec, lds_tiles,
                                                                            sl.d_long_reads, 0u);
    HIP_TRY(c, hipGetLastError());
  }
  return GRP_OK;
}

// counters + decisions to pinned memory, completion event.  ONE copy: the window's k_decide has moved the counters into the
// 64 bytes in front of the decisions and zeroed them (round 5: this was two copies and a fill, each with its gap on the stream,
// behind every window and every batch).
int
classify_enqueue_fetch(grp_ctx* c, QuerySlot& sl, hipStream_t st)
{
  if (sl.count == 0) { // no decide launch: the counters alone
    HIP_TRY(c, hipMemcpyAsync(sl.h_qctr, sl.d_qctr, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemsetAsync(sl.d_qctr, 0, 8 * sizeof(uint64_t), st));
  } else {
    HIP_TRY(c, hipMemcpyAsync(sl.h_dec_raw, sl.d_dec_raw, ((size_t)sl.count + 2) * sizeof(grp_read_decision), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(c, hipEventRecord(sl.done, st));
  sl.side_used = (st != c->stream);
  return GRP_OK;
}

// query + decide + fetch of the slot's window, all asynchronous
int
classify_enqueue(grp_ctx* c, QuerySlot& sl, bool side)
{
  c->q = &sl;
  QueryRun q;
  if (sl.nt) {
    int rc = enqueue_query(c, sl.reads, sl.first, sl.count, sl.list_cap, q);
    if (rc != GRP_OK) {
      return rc;
    }
  }
  sl.t0 = q.t0;
  // optimistic: decide right behind the query, one round trip in the common case.
  // `side`: decision + copy-back go to the second stream, so that the next window's
  // query kernel (already queued on the main stream) starts at once.
  hipStream_t st = c->stream;
  if (side) {
    HIP_TRY(c, hipEventRecord(sl.qdone, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream2, sl.qdone, 0));
    st = c->stream2;
  }
  int rc = classify_enqueue_decide(c, sl, st);
  if (rc == GRP_OK) {
    rc = classify_enqueue_fetch(c, sl, st);
  }
  return rc;
}

} // namespace

extern "C" {

// the slot's buffers and fields for a window of reads [first, first + count); nothing is enqueued
static int
classify_setup(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot)
{
  if (!c || !r || r->ctx != c || (uint64_t)first + count > r->n_reads || !dp || slot > 1) {
    return set_err(c, GRP_ERR_INVALID, "grp_classify_reads_begin: bad argument");
  }
  if (!c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_reads before grp_finalize");
  }
  QuerySlot& sl = c->slot[slot];
  if (sl.busy) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_reads_begin: slot %u still has a window in flight", slot);
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t nt = r->tile0[first + count] - r->tile0[first];
  if (nt > MAX_GRID_WGS) {
    return set_err(c, GRP_ERR_INVALID, "grp_classify_reads: %llu tiles in one call (limit 2^22)", (unsigned long long)nt);
  }
  sl.reads = r;
  sl.first = first;
  sl.count = count;
  sl.dp = *dp;
  sl.nt = nt;
  sl.busy = true;
  if (count == 0) {
    return GRP_OK;
  }
  c->q = &sl;
  // decision buffers + scratch for reads too long for the LDS path
  if (count > sl.dec_cap) {
    if (const int rc = slot_dec_alloc(c, sl, std::max<uint64_t>((uint64_t)count + count / 4, 1024)); rc != GRP_OK) {
      return rc;
    }
  }
  if (nt + 1 > sl.d_dec_cap) {
    (void)hipFree(sl.d_dec_ids);
    (void)hipFree(sl.d_dec_asg);
    (void)hipFree(sl.d_dec_scratch);
    sl.d_dec_ids = nullptr;
    sl.d_dec_asg = nullptr;
    sl.d_dec_scratch = nullptr;
    sl.d_dec_cap = 0;
    const uint64_t cap = nt + nt / 4 + 64;
    HIP_TRY(c, hipMalloc(&sl.d_dec_ids, cap * 4));
    HIP_TRY(c, hipMalloc(&sl.d_dec_asg, cap));
    HIP_TRY(c, hipMalloc(&sl.d_dec_scratch, cap * 8));
    sl.d_dec_cap = cap;
  }
  sl.list_cap = std::max<uint64_t>(sl.d_lists_cap, 4 * nt + 4096);
  return GRP_OK;
}

static int
classify_begin(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot, bool side)
{
  int rc = classify_setup(c, r, first, count, dp, slot);
  if (rc != GRP_OK || count == 0) {
    return rc;
  }
  if (c) {
    c->carry.valid = false; // the slot's summaries are about to be replaced
  }
  return classify_enqueue(c, c->slot[slot], side);
}

int
grp_classify_reads_begin(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot)
{
  return classify_begin(c, r, first, count, dp, slot, true);
}


int
grp_classify_reads_end(grp_ctx* c, uint32_t slot, grp_read_decision* out)
{
  if (!c || slot > 1 || !c->slot[slot].busy || c->slot[slot].streaming) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_reads_end: no window in flight in this slot");
  }
  QuerySlot& sl = c->slot[slot];
  sl.busy = false;
  if (sl.count == 0) {
    return GRP_OK;
  }
  if (!out) {
    // abandoned window: nothing to wait for, the slot's next use is ordered behind it
    return GRP_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  c->q = &sl;
  HIP_TRY(c, hipEventSynchronize(sl.done));
  for (int attempt = 0; attempt < 8; ++attempt) {
    if (sl.h_qctr[3] > sl.list_cap) { // list arena too small: grow (with room for the lists of tiles that are redone) and redo the window
      sl.list_cap = 2 * sl.h_qctr[3] + 4096;
      int rc = classify_enqueue(c, sl, false);
      if (rc != GRP_OK) {
        return rc;
      }
      HIP_TRY(c, hipEventSynchronize(sl.done));
      continue;
    }
    if (sl.h_qctr[4]) { // some tiles needed the worst-case table: redo them, decide again
      QueryRun q;
      q.t0 = sl.t0;
      int rc = enqueue_redo_flagged(c, sl.reads, sl.list_cap, q);
      if (rc == GRP_OK) {
        rc = classify_enqueue_decide(c, sl, c->stream);
      }
      if (rc == GRP_OK) {
        rc = classify_enqueue_fetch(c, sl, c->stream);
      }
      if (rc != GRP_OK) {
        return rc;
      }
      HIP_TRY(c, hipEventSynchronize(sl.done));
      continue;
    }
    if (c->pending.size() > 64) {
      drain_events(c);
    }
    memcpy(out, sl.h_dec, (size_t)sl.count * sizeof(grp_read_decision));
    return GRP_OK;
  }
  return set_err(c, GRP_ERR_NOMEM, "grp_classify_reads: list arena kept overflowing");
}

int
grp_classify_reads(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, const grp_decide_params* dp, grp_read_decision* out)
{
  if (!out && count) {
    return set_err(c, GRP_ERR_INVALID, "grp_classify_reads: out is NULL");
  }
  int rc = classify_begin(c, r, first, count, dp, 0, false);
  if (rc != GRP_OK) {
    if (c && r && r->ctx == c) {
      c->slot[0].busy = false;
    }
    return rc;
  }
  rc = grp_classify_reads_end(c, 0, out);
  if (rc == GRP_OK && !c->view && count) {
    // plain summaries of [first, first + count) stay in slot 0: grp_batch_verify patches them instead of asking again
    c->carry.valid = true;
    c->carry.slot = 0;
    c->carry.reads = r;
    c->carry.first = first;
    c->carry.count = count;
    c->carry.tile_off = 0;
  }
  return rc;
}

// ---- streaming window --------------------------------------------------------------

// the scratch table of a whole-read insert (k_insert_collect / k_insert_apply and the in-launch
// insert of a streaming window): room for `max_ranks` distinct ranks at load <= 1/2.  Growing it
// waits for the stream (never while a window that holds its address is in flight: see the caller).
static int
ensure_insert_table(grp_ctx* c, uint64_t max_ranks)
{
  const uint64_t want = next_pow2_64(max_ranks * 2);
  if (want <= c->ir_cap) {
    return GRP_OK;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipFree(c->d_ir_keys);
  (void)hipFree(c->d_ir_masks);
  (void)hipFree(c->d_ir_locs);
  (void)hipFree(c->d_ir_slots);
  (void)hipFree(c->d_fp_tab[0]);
  (void)hipFree(c->d_fp_tab[1]);
  c->d_ir_keys = c->d_ir_masks = c->d_ir_locs = nullptr;
  c->d_ir_slots = nullptr;
  c->d_fp_tab[0] = c->d_fp_tab[1] = nullptr;
  c->ir_cap = 0;
  HIP_TRY(c, hipMalloc(&c->d_fp_tab[0], want * 4));
  HIP_TRY(c, hipMalloc(&c->d_fp_tab[1], want * 4));
  HIP_TRY(c, hipMalloc(&c->d_ir_keys, want * 8));
  HIP_TRY(c, hipMalloc(&c->d_ir_masks, want * 8));
  HIP_TRY(c, hipMalloc(&c->d_ir_locs, want * 8));
  HIP_TRY(c, hipMalloc(&c->d_ir_slots, want * 4));
  HIP_TRY(c, hipMemsetAsync(c->d_ir_keys, 0, want * 8, c->stream));
  HIP_TRY(c, hipMemsetAsync(c->d_ir_masks, 0, want * 8, c->stream));
  if (!c->d_ir_counter) {
    HIP_TRY(c, hipMalloc(&c->d_ir_counter, 2 * sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(c->d_ir_counter, 0, 2 * sizeof(uint32_t), c->stream));
  }
  c->ir_cap = want;
  return GRP_OK;
}

static int
stream_begin_impl(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot, uint32_t stripe_reads, uint32_t n_owners, uint32_t owner, const grp_read_decision** decisions, bool want_resumable);

int
grp_classify_stream_begin(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot, const grp_read_decision** decisions)
{
  return stream_begin_impl(c, r, first, count, dp, slot, 0, 1, 0, decisions, false);
}

int
grp_classify_stream_begin_striped(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot, uint32_t stripe_reads, uint32_t n_owners, uint32_t owner, const grp_read_decision** decisions)
{
  return stream_begin_impl(c, r, first, count, dp, slot, stripe_reads, n_owners, owner, decisions, false);
}

int
grp_classify_stream_begin_resumable(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot, const grp_read_decision** decisions)
{
  return stream_begin_impl(c, r, first, count, dp, slot, 0, 1, 0, decisions, true);
}

int
grp_classify_stream_begin_striped_resumable(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot, uint32_t stripe_reads, uint32_t n_owners, uint32_t owner, const grp_read_decision** decisions)
{
  return stream_begin_impl(c, r, first, count, dp, slot, stripe_reads, n_owners, owner, decisions, true);
}

int
grp_classify_stream_insert_done(grp_ctx* c, uint32_t slot)
{
  if (!c || slot > 1 || !c->slot[slot].busy || !c->slot[slot].streaming) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_stream_insert_done: no streaming window in this slot");
  }
  QuerySlot& sl = c->slot[slot];
  if (__atomic_load_n(&sl.h_ack[0], __ATOMIC_ACQUIRE) == sl.cmd_seq) {
    return 1;
  }
  if (__atomic_load_n(&sl.h_ack[1], __ATOMIC_ACQUIRE) != 0) {
    return 2;
  }
  const hipError_t e = hipEventQuery(sl.done);
  if (e == hipSuccess) {
    return __atomic_load_n(&sl.h_ack[0], __ATOMIC_ACQUIRE) == sl.cmd_seq ? 1 : 2;
  }
  if (e != hipErrorNotReady) {
    return set_err(c, GRP_ERR_HIP, "grp_classify_stream_insert_done: %s", hipGetErrorString(e));
  }
  return 0;
}

int
grp_classify_stream_resumable(grp_ctx* c, uint32_t slot)
{
  if (!c || slot > 1 || !c->slot[slot].busy || !c->slot[slot].streaming) {
    return 0;
  }
  return c->slot[slot].resumable ? 1 : 0;
}

static int
stream_begin_impl(grp_ctx* c, const grp_reads* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot, uint32_t stripe_reads, uint32_t n_owners, uint32_t owner, const grp_read_decision** decisions, bool want_resumable)
{
  if (c) {
    c->carry.valid = false;
  }
  if (!c || !r || r->ctx != c || (uint64_t)first + count > r->n_reads || !dp || slot > 1 || !decisions || n_owners == 0 || owner >= n_owners ||
      (n_owners > 1 && stripe_reads == 0)) {
    return set_err(c, GRP_ERR_INVALID, "grp_classify_stream_begin: bad argument");
  }
  if (!c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_stream_begin before grp_finalize");
  }
  if (!c->coherent_arch) {
    // the hand-over between workgroups inside the launch relies on agent-scope accesses being
    // served by the memory side, in issue order (gfx942 / gfx950); not a portable guarantee
    return set_err(c, GRP_ERR_INVALID, "grp_classify_stream_begin: streaming windows are only enabled on gfx942 / gfx950 (this device is %s)", c->arch.c_str());
  }
  QuerySlot& sl = c->slot[slot];
  if (sl.busy) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_stream_begin: slot %u still has a window in flight", slot);
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t t0 = r->tile0[first];
  const uint64_t nt = r->tile0[first + count] - t0;
  if (nt > (1u << 30)) {
    return set_err(c, GRP_ERR_INVALID, "grp_classify_stream_begin: %llu tiles in one call (limit 2^30)", (unsigned long long)nt);
  }
  {
    // Growing a buffer frees the old one, and hipFree / hipHostFree wait for the device.  A
    // resumable window in the other slot may be parked, waiting for THIS thread's answer — the
    // wait would never end (found on C2: a window with a record number of tiles, begun while
    // the current one still had an insert to take; the launch gave up after its idle limit).
    // Not now, then: the caller begins this window once the other one has ended.
    const QuerySlot& other = c->slot[slot ^ 1u];
    const bool grows = count > sl.sdec_cap || std::max<uint64_t>(count, 1) > sl.tiles_done_cap || nt + 1 > sl.d_dec_cap || std::max<uint64_t>(nt, 1) > sl.d_tiles_cap ||
                       std::max<uint64_t>(sl.d_lists_cap, (want_resumable ? 8 : 4) * nt + 4096) > sl.d_lists_cap || std::max<uint64_t>(nt, 1) > sl.d_flag_cap ||
                       (want_resumable && std::max<uint64_t>(nt, 1) > sl.tile_ver_cap);
    if (grows && other.busy && other.streaming && other.resumable) {
      return set_err(c, GRP_ERR_BUSY, "grp_classify_stream_begin: the slot's buffers must grow while a resumable window is in flight in the other slot: begin this window when that one has ended");
    }
  }
  c->q = &sl;
  if (count > sl.sdec_cap) {
    if (sl.h_sdec) {
      (void)hipHostFree(sl.h_sdec);
      sl.h_sdec = nullptr;
    }
    sl.sdec_cap = 0;
    const uint64_t cap = std::max<uint64_t>((uint64_t)count + count / 2, 1024);
    HIP_TRY(c, hipHostMalloc(&sl.h_sdec, cap * sizeof(grp_read_decision), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&sl.dmap_sdec), sl.h_sdec, 0));
    sl.sdec_cap = cap;
  }
  int rc = ensure_dev(c, sl.d_tiles_done, sl.tiles_done_cap, std::max<uint64_t>(count, 1));
  if (rc != GRP_OK) {
    return rc;
  }
  if (nt + 1 > sl.d_dec_cap) {
    (void)hipFree(sl.d_dec_ids);
    (void)hipFree(sl.d_dec_asg);
    (void)hipFree(sl.d_dec_scratch);
    sl.d_dec_ids = nullptr;
    sl.d_dec_asg = nullptr;
    sl.d_dec_scratch = nullptr;
    sl.d_dec_cap = 0;
    const uint64_t cap = nt + nt / 4 + 64;
    HIP_TRY(c, hipMalloc(&sl.d_dec_ids, cap * 4));
    HIP_TRY(c, hipMalloc(&sl.d_dec_asg, cap));
    HIP_TRY(c, hipMalloc(&sl.d_dec_scratch, cap * 8));
    sl.d_dec_cap = cap;
  }
  // (a window that keeps tiles across its inserts does not start its list arena over at each of them: twice the room)
  sl.list_cap = std::max<uint64_t>(sl.d_lists_cap, (want_resumable ? 8 : 4) * nt + 4096);
  rc = ensure_dev(c, sl.d_tiles, sl.d_tiles_cap, std::max<uint64_t>(nt, 1));
  if (rc == GRP_OK) {
    rc = ensure_dev(c, sl.d_lists, sl.d_lists_cap, sl.list_cap);
  }
  if (rc == GRP_OK) {
    rc = ensure_dev(c, sl.d_flag_idx, sl.d_flag_cap, std::max<uint64_t>(nt, 1));
  }
  if (rc == GRP_OK && want_resumable) { // (round 6) what a window keeps across an in-launch insert: a mark per tile (DevStreamCtl::tile_ver)
    rc = ensure_dev(c, sl.d_tile_ver, sl.tile_ver_cap, std::max<uint64_t>(nt, 1));
  }
  if (rc != GRP_OK) {
    return rc;
  }
  // striped window (several ranks share it): stripe t = reads [t*stripe, (t+1)*stripe) of
  // the window belongs to rank t % n_owners; this launch draws only the owner's tiles
  uint64_t n_mine = nt;
  const bool striped = n_owners > 1;
  if (striped) {
    n_mine = 0;
    for (uint32_t j = 0; j < count; ++j) {
      if ((j / stripe_reads) % n_owners == owner) {
        n_mine += r->tile0[first + j + 1] - r->tile0[first + j];
      }
    }
    if (n_mine > sl.stripe_tiles_cap) {
      if (sl.h_stripe_tiles) {
        (void)hipHostFree(sl.h_stripe_tiles);
        sl.h_stripe_tiles = nullptr;
      }
      (void)hipFree(sl.d_stripe_tiles);
      sl.d_stripe_tiles = nullptr;
      sl.stripe_tiles_cap = 0;
      const uint64_t cap = n_mine + n_mine / 4 + 1024;
      HIP_TRY(c, hipHostMalloc(&sl.h_stripe_tiles, cap * sizeof(uint32_t), hipHostMallocDefault));
      HIP_TRY(c, hipMalloc(&sl.d_stripe_tiles, cap * sizeof(uint32_t)));
      sl.stripe_tiles_cap = cap;
    }
    uint64_t w = 0;
    for (uint32_t j = 0; j < count; ++j) {
      if ((j / stripe_reads) % n_owners == owner) {
        for (uint64_t t = r->tile0[first + j]; t < r->tile0[first + j + 1]; ++t) {
          sl.h_stripe_tiles[w++] = (uint32_t)(t - t0);
        }
      }
    }
  }
  // reads without a single tile are never completed by a workgroup: decided here
  memset(sl.h_sdec, 0, (size_t)count * sizeof(grp_read_decision));
  uint32_t n_decidable = 0; // ... of THIS launch: a striped window completes when its own stripes' reads are decided
  uint64_t max_tiles_read = 0;
  for (uint32_t j = 0; j < count; ++j) {
    const uint64_t ntj = r->tile0[first + j + 1] - r->tile0[first + j];
    max_tiles_read = std::max(max_tiles_read, ntj);
    if (ntj == 0) {
      grp_read_decision d{};
      gr::core::decide(dp->threshold, dp->unassigned_min, dp->assigned_max, 0, nullptr, nullptr, nullptr, nullptr, nullptr, d);
      d.pad = 1;
      sl.h_sdec[j] = d;
    } else if (!striped || (j / stripe_reads) % n_owners == owner) {
      ++n_decidable;
    }
  }
  // A resumable window (grp_classify_stream_begin_resumable) does not end where it parks: it
  // waits for the host's word — _abort, or _insert: the insert the host has committed, applied
  // inside the launch.  Not when the scratch table would have to grow while the other slot's
  // window holds its address (the window is then an ordinary one; _insert says so).
  const bool resume_off = c->env_stream_resume_off; // GRP_STREAM_RESUME=off (developer switch / tests, read at grp_create): every window ends where it parks
  sl.resumable = false;
  if (want_resumable && !resume_off && n_mine != 0) {
    const uint64_t need_ranks = max_tiles_read * c->params.tile * c->params.h;
    const bool other_streaming = c->slot[slot ^ 1u].busy && c->slot[slot ^ 1u].streaming;
    if (next_pow2_64(need_ranks * 2) <= c->ir_cap || !other_streaming) {
      rc = ensure_insert_table(c, need_ranks);
      if (rc != GRP_OK) {
        return rc;
      }
      sl.resumable = true;
    }
  }
  sl.gen = 1;
  sl.cmd_seq = 0;
  sl.stripe_reads = striped ? stripe_reads : 0;
  sl.n_owners = n_owners;
  sl.owner = owner;
  __atomic_store_n(sl.h_abort, 0u, __ATOMIC_RELEASE);
  sl.reads = r;
  sl.first = first;
  sl.count = count;
  sl.dp = *dp;
  sl.nt = nt;
  sl.t0 = t0;
  sl.busy = true;
  sl.streaming = true;
  *decisions = sl.h_sdec;
  if (sl.side_used) {
    HIP_TRY(c, hipStreamWaitEvent(c->stream, sl.done, 0));
    sl.side_used = false;
  }
  if (n_mine) {
    if (striped) {
      HIP_TRY(c, hipMemcpyAsync(sl.d_stripe_tiles, sl.h_stripe_tiles, n_mine * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(sl.d_abort, 0, 256, c->stream));
    HIP_TRY(c, hipMemsetAsync(sl.d_abort + 48, 0xFF, sizeof(uint32_t), c->stream)); // park: nothing stale yet
    HIP_TRY(c, hipMemsetAsync(sl.d_tiles_done, 0, (size_t)count * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(sl.d_executed, 0, 8 * sizeof(unsigned long long), c->stream));
    if (sl.resumable) {
      HIP_TRY(c, hipMemsetAsync(sl.d_sctl, 0, SCT_WORDS * sizeof(uint32_t), c->stream));
      HIP_TRY(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(sl.d_sctl + SCT_GEN), 1, 1, c->stream));
      HIP_TRY(c, hipMemsetAsync(sl.d_tile_ver, 0, std::max<uint64_t>(nt, 1) * sizeof(uint32_t), c->stream));
      HIP_TRY(c, hipMemsetAsync(sl.d_rel, 0, (size_t)STREAM_REL_WGS * 32u * sizeof(uint32_t), c->stream));
      // the changed-slot sets of the window's inserts start empty (the sets of the window before may hold its last two)
      HIP_TRY(c, hipMemsetAsync(c->d_fp_tab[0], 0, c->ir_cap * sizeof(uint32_t), c->stream));
      HIP_TRY(c, hipMemsetAsync(c->d_fp_tab[1], 0, c->ir_cap * sizeof(uint32_t), c->stream));
      __atomic_store_n(&sl.h_cmd[0], 0u, __ATOMIC_RELEASE);
      sl.h_ack[0] = 0;
      sl.h_ack[1] = 0;
    }
    DevStreamCtl sc;
    sc.first = first;
    sc.tiles_done = sl.d_tiles_done;
    sc.abort = sl.d_abort;
    sc.abort_host = sl.dmap_abort;
    sc.next_tile = sl.d_abort + 32;
    sc.park = sl.d_abort + 48;
    sc.early_park = c->env_no_early_park ? 0u : 1u; // GRP_NO_EARLY_PARK, developer hook (the engine-level API contract assumes 1)
    sc.n_tiles = (uint32_t)n_mine;
    sc.dec = sl.dmap_sdec;
    sc.executed = sl.d_executed;
    sc.dp = *dp;
    sc.g_ids = sl.d_dec_ids;
    sc.g_asg = sl.d_dec_asg;
    sc.g_scratch = sl.d_dec_scratch;
    sc.n_reads = count;
    sc.n_decidable = n_decidable;
    if (sl.resumable) {
      sc.ctl = sl.d_sctl;
      sc.cmd_while_busy = striped ? 1u : 0u;
      sc.cmd_host = sl.dmap_cmd;
      sc.ack_host = sl.dmap_ack;
      sc.tb = InsertTable{ c->d_ir_keys, c->d_ir_masks, c->d_ir_locs, c->d_ir_slots, c->d_ir_counter, c->ir_cap - 1 };
      static const bool dbg_on = getenv("GRP_STREAM_DEBUG") != nullptr;
      if (dbg_on) {
        if (!sl.d_dbg) {
          HIP_TRY(c, hipMalloc(&sl.d_dbg, 8192 * sizeof(uint32_t)));
        }
        HIP_TRY(c, hipMemsetAsync(sl.d_dbg, 0, 8192 * sizeof(uint32_t), c->stream));
        sc.dbg = sl.d_dbg;
      }
      sc.rel = sl.d_rel;
      sc.tile_ver = sl.d_tile_ver;
      sc.fp_tab[0] = c->d_fp_tab[0];
      sc.fp_tab[1] = c->d_fp_tab[1];
      sc.fp_mask = (uint32_t)(c->ir_cap - 1);
      sc.n_fp = 2; // asked for; launch_query gives what this geometry's LDS has room for beside three workgroups per CU (0: nothing is kept)
      static const double wait_s = [] { // developer hook: time limit of a grid-wide wait (seconds)
        const char* e = getenv("GRP_STREAM_WAIT_S");
        return e ? atof(e) : 2.0;
      }();
      sc.timeout_ticks = (unsigned long long)(wait_s * 1e8); // wall_clock64 counts at 100 MHz
    }
    const QueryGeom g = query_geom(c, false);
    {
      Timer t(c, GRP_K_QUERY, 0); // the probes actually executed are added at _end
      int lrc = GRP_OK;
      DISPATCH_H(c->params.h, lrc = launch_query<HH>(c, r, n_mine, t0, striped ? sl.d_stripe_tiles : nullptr, g, sl.list_cap, nullptr, nullptr, 0, &sc));
      if (lrc == GRP_ERR_BUSY && sl.resumable) {
        // the runtime cannot keep every workgroup of the window resident (the device is shared): the window is
        // begun in its classic form — it ends where it parks, grp_classify_stream_insert says GRP_ERR_STATE
        sl.resumable = false;
        sc.ctl = nullptr;
        ++c->n_stream_coop_refused;
        DISPATCH_H(c->params.h, lrc = launch_query<HH>(c, r, n_mine, t0, striped ? sl.d_stripe_tiles : nullptr, g, sl.list_cap, nullptr, nullptr, 0, &sc));
      }
      if (lrc != GRP_OK) {
        sl.busy = false;
        sl.streaming = false;
        return lrc;
      }
      HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipMemcpyAsync(sl.h_executed, sl.d_executed, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemsetAsync(sl.d_qctr, 0, 8 * sizeof(uint64_t), c->stream));
  } else {
    memset(sl.h_executed, 0, 8 * sizeof(unsigned long long));
  }
  HIP_TRY(c, hipEventRecord(sl.done, c->stream));
  return GRP_OK;
}

int
grp_classify_stream_abort(grp_ctx* c, uint32_t slot)
{
  if (!c || slot > 1 || !c->slot[slot].busy || !c->slot[slot].streaming) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_stream_abort: no streaming window in this slot");
  }
  __atomic_store_n(c->slot[slot].h_abort, 1u, __ATOMIC_RELEASE);
  return GRP_OK;
}

int
grp_classify_stream_poll(grp_ctx* c, uint32_t slot)
{
  if (!c || slot > 1 || !c->slot[slot].busy || !c->slot[slot].streaming) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_stream_poll: no streaming window in this slot");
  }
  hipError_t e = hipEventQuery(c->slot[slot].done);
  if (e == hipSuccess) {
    return 1;
  }
  if (e == hipErrorNotReady) {
    return 0;
  }
  return set_err(c, GRP_ERR_HIP, "grp_classify_stream_poll: %s", hipGetErrorString(e));
}

int
grp_classify_stream_end(grp_ctx* c, uint32_t slot, uint32_t* reads_decided)
{
  if (!c || slot > 1 || !c->slot[slot].busy || !c->slot[slot].streaming) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_stream_end: no streaming window in this slot");
  }
  QuerySlot& sl = c->slot[slot];
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipEventSynchronize(sl.done));
  sl.busy = false;
  sl.streaming = false;
  c->kstat[GRP_K_QUERY].units += *sl.h_executed;
  for (int i = 1; i < 8; ++i) { // what the window's in-launch inserts kept and redid (grp_debug_stream_stats)
    c->n_stream_keep[i] += sl.h_executed[i];
  }
  if (reads_decided) {
    uint32_t n = 0;
    for (uint32_t j = 0; j < sl.count; ++j) {
      n += sl.h_sdec[j].pad != 0;
    }
    *reads_decided = n;
  }
  if (c->pending.size() > 64) {
    drain_events(c);
  }
  if (sl.resumable) {
    sl.resumable = false;
    const uint32_t code = __atomic_load_n(&sl.h_ack[1], __ATOMIC_ACQUIRE);
    const uint32_t applied = __atomic_load_n(&sl.h_ack[0], __ATOMIC_ACQUIRE);
    if ((code == 2 || code == 1) && sl.d_dbg) { // developer: where every workgroup was
      std::vector<uint32_t> st(8192);
      if (hipMemcpy(st.data(), sl.d_dbg, st.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
        std::string line;
        for (size_t i = 0; i < st.size(); ++i) {
          if (i < 1024 && st[i] != 0 && ((st[i] >> 28) != 5u || (st[i] & 0xFFFFu) != ((sl.cmd_seq << 1) & 0xFFFFu))) {
            char b[64];
            snprintf(b, sizeof(b), " %zu:%08x", i, st[i]);
            line += b;
          }
        }
        {
          // when the workgroups came to the event and reached the first wait (100 MHz ticks, relative to the earliest)
          const uint32_t grid = std::min<uint32_t>(st[4095], 1000u);
          uint32_t t_min = 0xFFFFFFFFu;
          for (uint32_t i = 0; i < grid; ++i) {
            t_min = std::min(t_min, st[2048 + i]);
          }
          std::vector<std::pair<uint32_t, uint32_t>> late;
          for (uint32_t i = 0; i < grid; ++i) {
            late.push_back({ st[2048 + i] - t_min, i });
          }
          std::sort(late.begin(), late.end());
          char b[200];
          snprintf(b, sizeof(b), " | grid %u; reached the first wait (us behind the earliest): median %.1f", st[4095], late[grid / 2].first * 1e-2);
          line += b;
          for (uint32_t q = grid >= 6 ? grid - 6 : 0; q < grid; ++q) {
            const uint32_t w = late[q].second;
            snprintf(b, sizeof(b), "; wg %u %.1f us (came to the event %.1f us earlier; began %.1f us before that; xcc %u se %u cu %u)", w, late[q].first * 1e-2, (st[2048 + w] - st[1024 + w]) * 1e-2,
                     (st[1024 + w] - st[4096 + w]) * 1e-2, st[5120 + w] >> 16, (st[5120 + w] >> 13) & 7u, (st[5120 + w] >> 8) & 15u);
            line += b;
          }
          uint32_t t_start = 0xFFFFFFFFu, t_last = 0;
          for (uint32_t i = 0; i < grid; ++i) {
            t_start = std::min(t_start, st[4096 + i]);
          }
          uint32_t n_late = 0;
          for (uint32_t i = 0; i < grid; ++i) {
            t_last = std::max(t_last, st[4096 + i] - t_start);
            n_late += (st[4096 + i] - t_start) > 100000u ? 1u : 0u; // more than a millisecond behind the first
          }
          snprintf(b, sizeof(b), " | the last workgroup began %.1f us behind the first, %u of them more than 1 ms behind", t_last * 1e-2, n_late);
          line += b;
        }
        fprintf(stderr, "  last streaming launch: %u workgroups, %u bytes of LDS each\n", c->dbg_stream_grid, c->dbg_stream_lds);
        fprintf(stderr, "  window: reads [%u, %u), %llu tiles; command: read %u tiles [%u, %u) block %u first id %u offset %u resume read %u tile %u decided base %u gen %u\n", sl.first, sl.first + sl.count, (unsigned long long)sl.nt,
                sl.h_cmd[1], sl.h_cmd[2], sl.h_cmd[3], sl.h_cmd[4], sl.h_cmd[5], sl.h_cmd[6], sl.h_cmd[7], sl.h_cmd[8], sl.h_cmd[9], sl.h_cmd[10]);
        fprintf(stderr, "grp_classify_stream_end: code %u, first to give up: wait %u, workgroup %u, %u of the insert's %u workgroups had arrived; workgroups not inside insert %u (index:state):%s\n", code, sl.h_ack[14] & 0xFFFFu, sl.h_ack[14] >> 16, sl.h_ack[15] & 0xFFFFu, sl.h_ack[15] >> 16, sl.cmd_seq, line.c_str());
      }
    }
    if (code == 2) {
      return set_err(c, GRP_ERR_STATE, "grp_classify_stream_end: a grid-wide wait timed out in the middle of an insert (the ID array may be inconsistent); first to give up: wait %u, workgroup %u, %u of the insert's %u workgroups had arrived",
                     sl.h_ack[14] & 0xFFFFu, sl.h_ack[14] >> 16, sl.h_ack[15] & 0xFFFFu, sl.h_ack[15] >> 16);
    }
    if (code == 3) {
      // The parked window was told nothing for its idle limit (16 x GRP_STREAM_WAIT_S: a host stopped by a
      // debugger or SIGSTOP, a stalled file system) and left by itself.  It had modified nothing: not an error
      //  — the records it did not produce stay open and the caller begins the window again behind
      // the last one it has; what the launch was waiting on is kept as the context's last message.
      uint32_t first_open = sl.count;
      for (uint32_t j = 0; j < sl.count; ++j) {
        if (sl.h_sdec[j].pad != sl.gen) {
          first_open = j;
          break;
        }
      }
      ++c->n_stream_idle_exits;
      (void)set_err(c, GRP_OK,
                    "grp_classify_stream_end: the parked window was never told how to go on (idle time limit) and left; launch: park %u decided %u next tile %u / %llu event %u command seen %u "
                    "generation %u; host: %u reads, generation %u, commands posted %u applied %u, first open record %u",
                    sl.h_ack[8], sl.h_ack[9], sl.h_ack[10], (unsigned long long)sl.nt, sl.h_ack[11], sl.h_ack[12], sl.h_ack[13], sl.count, sl.gen, sl.cmd_seq, applied, first_open);
      if (c->env_trace_abort) {
        fprintf(stderr, "%s\n", c->err.c_str());
      }
    }
    if (applied != sl.cmd_seq) {
      // The launch ended without the insert posted last (an abort overtook it or the launch had left already: 1; the
      // first grid-wide wait timed out, the device is shared: 2).  Only the scratch table may have been
      // touched; the caller applies the insert with grp_insert_read.
      HIP_TRY(c, hipMemsetAsync(c->d_ir_keys, 0, c->ir_cap * 8, c->stream));
      HIP_TRY(c, hipMemsetAsync(c->d_ir_masks, 0, c->ir_cap * 8, c->stream));
      return code == 1 ? 2 : 1;
    }
  }
  return GRP_OK;
}

int
grp_classify_stream_insert(grp_ctx* c, uint32_t slot, uint32_t read_idx, uint32_t tile_start, uint32_t tile_end, uint32_t block_tiles, uint32_t first_id, uint32_t id_offset, uint32_t* generation)
{
  if (c) {
    c->carry.valid = false;
  }
  if (!c || slot > 1 || !c->slot[slot].busy || !c->slot[slot].streaming || block_tiles == 0 || id_offset > 1 || !generation) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_stream_insert: no streaming window in this slot / bad argument");
  }
  QuerySlot& sl = c->slot[slot];
  const grp_reads* r = sl.reads;
  if (!sl.resumable) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_stream_insert: this window ends where it parks (not begun with grp_classify_stream_begin_resumable, or GRP_STREAM_RESUME=off)");
  }
  if (read_idx < sl.first || read_idx >= sl.first + sl.count) {
    return set_err(c, GRP_ERR_INVALID, "grp_classify_stream_insert: read %u is not in the window", read_idx);
  }
  const uint64_t ntile = r->tile0[read_idx + 1] - r->tile0[read_idx];
  if (tile_start >= tile_end || tile_end > ntile) {
    return set_err(c, GRP_ERR_INVALID, "grp_classify_stream_insert: tiles [%u,%u) outside the read's %llu tiles", tile_start, tile_end, (unsigned long long)ntile);
  }
  if ((tile_end - tile_start + block_tiles - 1) / block_tiles > 64) {
    return set_err(c, GRP_ERR_STATE, "grp_classify_stream_insert: more than 64 ID blocks (the caller ends the window and uses grp_insert_read)");
  }
  if (sl.cmd_seq != 0 && c->env_trace_abort) { // GRP_TRACE_ABORT, developer trace: phase times of the previous in-launch insert (100 MHz ticks)
    static double t_ph[4] = { 0, 0, 0, 0 };
    static uint64_t n_tr = 0;
    for (int i = 0; i < 4; ++i) {
      t_ph[i] += sl.h_ack[4 + i] * 1e-2;
    }
    if (getenv("GRP_TRACE_ABORT")[0] == '2') {
      fprintf(stderr, "  previous in-launch insert, workgroup 0: collect %.1f us, first wait %.1f us, apply %.1f us, second wait %.1f us\n", sl.h_ack[4] * 1e-2, sl.h_ack[5] * 1e-2, sl.h_ack[6] * 1e-2, sl.h_ack[7] * 1e-2);
    }
    if ((++n_tr & 511u) == 0) {
      fprintf(stderr, "in-launch inserts %llu: workgroup 0 spent %.1f us collecting, %.1f us in the first wait, %.1f us applying, %.1f us in the second wait\n", (unsigned long long)n_tr, t_ph[0] / n_tr,
              t_ph[1] / n_tr, t_ph[2] / n_tr, t_ph[3] / n_tr);
    }
  }
  // (records of the new generation can overtake the launch's acknowledgement of the insert in front of them by a moment)
  for (uint32_t spins = 0; __atomic_load_n(&sl.h_ack[0], __ATOMIC_ACQUIRE) != sl.cmd_seq; ++spins) {
    if (spins > (1u << 26) || __atomic_load_n(&sl.h_ack[1], __ATOMIC_ACQUIRE) != 0) {
      return set_err(c, GRP_ERR_STATE, "grp_classify_stream_insert: the previous insert of this window has not been applied");
    }
    __builtin_ia32_pause();
  }
  const uint32_t x = read_idx - sl.first;
  uint32_t decided_base = 0; // this launch's decidable reads up to the insert, and its tiles up to there (= where its dispenser starts over)
  uint64_t resume_tile = 0;
  for (uint32_t j = 0; j <= x; ++j) {
    if (sl.stripe_reads == 0 || (j / sl.stripe_reads) % sl.n_owners == sl.owner) {
      const uint64_t ntj = r->tile0[sl.first + j + 1] - r->tile0[sl.first + j];
      decided_base += ntj != 0;
      resume_tile += ntj;
    }
  }
  const uint32_t gen = ++sl.gen;
  for (uint32_t j = x + 1; j < sl.count; ++j) { // the host's own records (reads without tiles) move to the new generation
    if (r->tile0[sl.first + j + 1] == r->tile0[sl.first + j]) {
      __atomic_store_n(&sl.h_sdec[j].pad, gen, __ATOMIC_RELEASE);
    }
  }
  uint32_t* w = sl.h_cmd + 1;
  w[0] = read_idx;
  w[1] = tile_start;
  w[2] = tile_end;
  w[3] = block_tiles;
  w[4] = first_id;
  w[5] = id_offset;
  w[6] = x + 1;
  w[7] = (uint32_t)resume_tile;
  w[8] = decided_base;
  w[9] = gen;
  __atomic_store_n(&sl.h_cmd[0], ++sl.cmd_seq, __ATOMIC_RELEASE);
  c->kstat[GRP_K_INSERT].launches += 1;
  c->kstat[GRP_K_INSERT].units += (uint64_t)(tile_end - tile_start) * c->params.tile * c->params.h;
  *generation = gen;
  return GRP_OK;
}

// ---- insert -------------------------------------------------------------------------

int
grp_insert_tiles(grp_ctx* c, const grp_reads* r, uint32_t read_idx, uint32_t tile_start, uint32_t tile_end, uint32_t id)
{
  if (c) {
    c->carry.valid = false;
  }
  if (!c || !r || r->ctx != c || read_idx >= r->n_reads) {
    return set_err(c, GRP_ERR_INVALID, "grp_insert_tiles: bad read index");
  }
  if (!c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_insert_tiles before grp_finalize");
  }
  const uint64_t ntile = r->tile0[read_idx + 1] - r->tile0[read_idx];
  if (tile_start > tile_end || tile_end > ntile) {
    return set_err(c, GRP_ERR_INVALID, "grp_insert_tiles: tiles [%u,%u) outside the read's %llu tiles", tile_start, tile_end, (unsigned long long)ntile);
  }
  if (tile_start == tile_end) {
    return GRP_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t nt = tile_end - tile_start;
  const uint64_t max_ranks = (uint64_t)nt * c->params.tile * c->params.h;
  const uint64_t want = next_pow2(max_ranks * 2);
  if (want > c->dedup_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_dedup);
    c->d_dedup = nullptr;
    c->dedup_cap = 0;
    HIP_TRY(c, hipMalloc(&c->d_dedup, want * 8));
    HIP_TRY(c, hipMemsetAsync(c->d_dedup, 0, want * 8, c->stream));
    c->dedup_cap = want;
    c->epoch = 0;
  }
  c->epoch += 1;
  if (c->epoch >= (1u << 24)) {
    HIP_TRY(c, hipMemsetAsync(c->d_dedup, 0, c->dedup_cap * 8, c->stream));
    c->epoch = 1;
  }
  const unsigned long long epoch_tag = (unsigned long long)c->epoch << 40;
  const size_t lds = tab_bytes(c) + bases_bytes(c->params.tile + c->params.k + c->params.h);
  {
    Timer t(c, GRP_K_INSERT, max_ranks);
    DISPATCH_H(c->params.h, (k_insert<HH><<<dim3(nt * ((c->params.tile + THREADS - 1) / THREADS)), dim3(THREADS), lds, c->stream>>>(c->f, r->dev, c->d_seeds, c->params.tile, read_idx, tile_start, id, c->d_dedup, c->dedup_cap - 1, epoch_tag)));
  }
  HIP_TRY(c, hipGetLastError());
  return GRP_OK;
}

int
grp_insert_read(grp_ctx* c, const grp_reads* r, uint32_t read_idx, uint32_t tile_start, uint32_t tile_end, uint32_t block_tiles, uint32_t first_id, uint32_t id_offset)
{
  if (c) {
    c->carry.valid = false;
  }
  if (!c || !r || r->ctx != c || read_idx >= r->n_reads || block_tiles == 0 || id_offset > 1) {
    return set_err(c, GRP_ERR_INVALID, "grp_insert_read: bad argument");
  }
  if (!c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_insert_read before grp_finalize");
  }
  const uint64_t ntile = r->tile0[read_idx + 1] - r->tile0[read_idx];
  if (tile_start > tile_end || tile_end > ntile) {
    return set_err(c, GRP_ERR_INVALID, "grp_insert_read: tiles [%u,%u) outside the read's %llu tiles", tile_start, tile_end, (unsigned long long)ntile);
  }
  if (tile_start == tile_end) {
    return GRP_OK;
  }
  const uint32_t nt = tile_end - tile_start;
  const uint32_t n_blocks = (nt + block_tiles - 1) / block_tiles;
  if (n_blocks > 64) { // more ID blocks than mask bits: one launch per block
    for (uint32_t j = 0; j < n_blocks; ++j) {
      const uint32_t bs = tile_start + j * block_tiles;
      int rc = grp_insert_tiles(c, r, read_idx, bs, std::min(bs + block_tiles, tile_end), first_id + (j * block_tiles + id_offset) / block_tiles);
      if (rc != GRP_OK) {
        return rc;
      }
    }
    return GRP_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t max_ranks = (uint64_t)nt * c->params.tile * c->params.h;
  {
    const int trc = ensure_insert_table(c, max_ranks);
    if (trc != GRP_OK) {
      return trc;
    }
  }
  InsertTable tb{ c->d_ir_keys, c->d_ir_masks, c->d_ir_locs, c->d_ir_slots, c->d_ir_counter, c->ir_cap - 1 };
  const uint32_t parity = c->ir_parity;
  c->ir_parity ^= 1u;
  const size_t lds = tab_bytes(c) + bases_bytes(c->params.tile + c->params.k + c->params.h);
  {
    Timer t(c, GRP_K_INSERT, max_ranks);
    if (c->uniform_weight == 16) { // make_seed_pattern's default weight: care loop unrolled (the launch is latency-bound)
      DISPATCH_H(c->params.h, (k_insert_collect<HH, 16><<<dim3(nt * ((c->params.tile + THREADS - 1) / THREADS)), dim3(THREADS), lds, c->stream>>>(c->f, r->dev, c->d_seeds, c->params.tile, read_idx, tile_start, block_tiles, tb, parity)));
    } else {
      DISPATCH_H(c->params.h, (k_insert_collect<HH, 0><<<dim3(nt * ((c->params.tile + THREADS - 1) / THREADS)), dim3(THREADS), lds, c->stream>>>(c->f, r->dev, c->d_seeds, c->params.tile, read_idx, tile_start, block_tiles, tb, parity)));
    }
    k_insert_apply<<<dim3((uint32_t)((max_ranks + THREADS - 1) / THREADS)), dim3(THREADS), 0, c->stream>>>(c->f, tb, parity, block_tiles, first_id, id_offset);
  }
  HIP_TRY(c, hipGetLastError());
  return GRP_OK;
}

int
grp_reset_ids(grp_ctx* c)
{
  if (c) {
    c->carry.valid = false;
  }
  if (!c || !c->finalized) {
    return set_err(c, GRP_ERR_STATE, "grp_reset_ids before grp_finalize");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  k_reset_bucket_ids<<<dim3(65536), dim3(THREADS), 0, c->stream>>>(c->f.buckets, c->f.n_buckets);
  HIP_TRY(c, hipGetLastError());
  k_far_reset<<<dim3(4096), dim3(THREADS), 0, c->stream>>>(c->f.far, c->f.far_mask + 1); // (the insert lines went with the IDs above)
  c->batch.epoch = 0; // (the claims are gone with the counts)
  HIP_TRY(c, hipMemsetAsync(c->f.ovf_keys, 0, (c->f.ovf_mask + 1) * sizeof(unsigned long long), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->f.ovf_ids, 0, (c->f.ovf_mask + 1) * sizeof(uint32_t), c->stream));
  return GRP_OK;
}

int
grp_sync(grp_ctx* c)
{
  if (!c) {
    return GRP_ERR_INVALID;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream3));
  HIP_TRY(c, hipStreamSynchronize(c->stream2));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return GRP_OK;
}

// ---- inspection -------------------------------------------------------------------

uint64_t
grp_filter_bits(const grp_ctx* c)
{
  return c ? c->f.m : 0;
}

uint64_t
grp_pop(const grp_ctx* c)
{
  return c ? c->f.pop : 0;
}

int
grp_export_bits(grp_ctx* c, uint64_t* words, uint64_t n_words)
{
  if (!c || !words || n_words != (c->f.m + 63) / 64) {
    return set_err(c, GRP_ERR_INVALID, "grp_export_bits: n_words must be ceil(m/64)");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  memset(words, 0, n_words * 8);
  if (!c->finalized) {
    // phase 1: the device words ARE the sdsl layout (little-endian 32-bit halves)
    HIP_TRY(c, hipMemcpyAsync(words, c->f.bv, c->n_bv_words * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GRP_OK;
  }
  uint32_t* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, (n_words * 2 + 3) * 4));
  HIP_TRY(c, hipMemsetAsync(d, 0, (n_words * 2 + 3) * 4, c->stream));
  k_export_bits<<<dim3((uint32_t)std::min<uint64_t>((c->f.n_buckets + 255) / 256, MAX_GRID_WGS)), dim3(256), 0, c->stream>>>(c->f.buckets, c->f.n_buckets, c->f.W, d);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(words, d, n_words * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  return GRP_OK;
}

int
grp_rank(grp_ctx* c, const uint64_t* pos, uint64_t n, uint8_t* bit, uint64_t* rank)
{
  if (!c || !c->finalized || !pos || !bit || !rank) {
    return set_err(c, GRP_ERR_STATE, "grp_rank: needs a finalized filter and non-null arrays");
  }
  for (uint64_t i = 0; i < n; ++i) {
    if (pos[i] >= c->f.m) {
      return set_err(c, GRP_ERR_INVALID, "grp_rank: position %llu >= m", (unsigned long long)pos[i]);
    }
  }
  if (n == 0) {
    return GRP_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  uint64_t *d_pos = nullptr, *d_rank = nullptr;
  uint8_t* d_bit = nullptr;
  HIP_TRY(c, hipMalloc(&d_pos, n * 8));
  HIP_TRY(c, hipMalloc(&d_rank, n * 8));
  HIP_TRY(c, hipMalloc(&d_bit, n));
  HIP_TRY(c, hipMemcpyAsync(d_pos, pos, n * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_rank_positions, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, c->f, d_pos, n, d_bit, d_rank);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(bit, d_bit, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(rank, d_rank, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d_pos);
  (void)hipFree(d_rank);
  (void)hipFree(d_bit);
  return GRP_OK;
}

int
grp_export_ids(grp_ctx* c, uint64_t first, uint64_t n, uint32_t* ids, uint32_t* counts)
{
  if (!c || !c->finalized || first + n > c->f.pop || !ids || !counts) {
    return set_err(c, GRP_ERR_INVALID, "grp_export_ids: bad range");
  }
  if (n == 0) {
    return GRP_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t *d_ids = nullptr, *d_cnt = nullptr;
  HIP_TRY(c, hipMalloc(&d_ids, n * 4));
  HIP_TRY(c, hipMalloc(&d_cnt, n * 4));
  k_export_ids<<<dim3((uint32_t)std::min<uint64_t>((c->f.n_buckets + 255) / 256, MAX_GRID_WGS)), dim3(256), 0, c->stream>>>(c->f, first, n, d_ids, d_cnt);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(ids, d_ids, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(counts, d_cnt, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d_ids);
  (void)hipFree(d_cnt);
  return GRP_OK;
}

int
grp_import_ids(grp_ctx* c, uint64_t first, uint64_t n, const uint32_t* ids, const uint32_t* counts)
{
  if (c) {
    c->carry.valid = false;
  }
  if (!c || !c->finalized || first + n > c->f.pop) {
    return set_err(c, GRP_ERR_INVALID, "grp_import_ids: bad range");
  }
  if (n == 0 || (!ids && !counts)) {
    return GRP_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t *d_ids = nullptr, *d_cnt = nullptr;
  if (ids) {
    HIP_TRY(c, hipMalloc(&d_ids, n * 4));
    HIP_TRY(c, hipMemcpyAsync(d_ids, ids, n * 4, hipMemcpyHostToDevice, c->stream));
  }
  if (counts) {
    HIP_TRY(c, hipMalloc(&d_cnt, n * 4));
    HIP_TRY(c, hipMemcpyAsync(d_cnt, counts, n * 4, hipMemcpyHostToDevice, c->stream));
  }
  k_import_ids<<<dim3((uint32_t)std::min<uint64_t>((c->f.n_buckets + 255) / 256, MAX_GRID_WGS)), dim3(256), 0, c->stream>>>(c->f, first, n, d_ids, d_cnt);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d_ids);
  (void)hipFree(d_cnt);
  return GRP_OK;
}

int
grp_debug_tile_hashes(grp_ctx* c, const grp_reads* r, uint32_t read_idx, uint32_t tile_idx, uint64_t* out, uint64_t cap, uint64_t* n_values)
{
  if (!c || !r || r->ctx != c || read_idx >= r->n_reads || !out || !n_values) {
    return set_err(c, GRP_ERR_INVALID, "grp_debug_tile_hashes: bad argument");
  }
  const uint32_t tile = c->params.tile, k = c->params.k;
  if (tile_idx >= r->len[read_idx] / tile) {
    return set_err(c, GRP_ERR_INVALID, "grp_debug_tile_hashes: tile index out of range");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t start = tile_idx * tile;
  const uint32_t Lp = std::min(tile + k - 1, r->len[read_idx] - start);
  const uint64_t nv = (uint64_t)(Lp - k + 1) * c->params.h;
  *n_values = nv;
  if (nv > cap) {
    return set_err(c, GRP_ERR_NOMEM, "grp_debug_tile_hashes: %llu values, capacity %llu", (unsigned long long)nv, (unsigned long long)cap);
  }
  uint64_t* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, nv * 8));
  const size_t lds = tab_bytes(c) + bases_bytes(tile + k + c->params.h);
  DISPATCH_H(c->params.h, (k_debug_tile_hashes<HH><<<dim3(1), dim3(THREADS), lds, c->stream>>>(r->dev, c->d_seeds, tile, read_idx, tile_idx, d, nv)));
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, d, nv * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  return GRP_OK;
}

int
grp_debug_locate(grp_ctx* c, const uint64_t* x, uint64_t n, uint64_t m, uint32_t W, int on_device, uint64_t* mod_out, uint64_t* div_out)
{
  if (!x || !mod_out || !div_out || m == 0 || W < GRP_BUCKET_IDS || W > 64 || (on_device && !c)) {
    return set_err(c, GRP_ERR_INVALID, "grp_debug_locate: bad argument");
  }
  const uint64_t m_inv = ~0ULL / m;                                               // as grp_create
  const uint64_t w_magic = (uint64_t)((((unsigned __int128)1) << 64) / W) + 1;   // as grp_finalize
  if (!on_device) {
    for (uint64_t i = 0; i < n; ++i) {
      mod_out[i] = grp_mod_m(x[i], m, m_inv);
      div_out[i] = grp_div_w(mod_out[i], w_magic);
    }
    return GRP_OK;
  }
  if (n == 0) {
    return GRP_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  uint64_t *d_x = nullptr, *d_mod = nullptr, *d_div = nullptr;
  HIP_TRY(c, hipMalloc(&d_x, n * 8));
  HIP_TRY(c, hipMalloc(&d_mod, n * 8));
  HIP_TRY(c, hipMalloc(&d_div, n * 8));
  HIP_TRY(c, hipMemcpyAsync(d_x, x, n * 8, hipMemcpyHostToDevice, c->stream));
  k_debug_locate<<<dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream>>>(d_x, n, m, m_inv, W, w_magic, d_mod, d_div);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(mod_out, d_mod, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(div_out, d_div, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d_x);
  (void)hipFree(d_mod);
  (void)hipFree(d_div);
  return GRP_OK;
}

int
grp_debug_tile_states(grp_ctx* c, uint64_t n_tiles, uint32_t* ids, uint8_t* assigned)
{
  if (!c || !ids || !assigned || n_tiles > c->q->d_dec_cap) { // c->q: the slot of the last window
    return set_err(c, GRP_ERR_INVALID, "grp_debug_tile_states: no such window");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpy(ids, c->q->d_dec_ids, n_tiles * 4, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(assigned, c->q->d_dec_asg, n_tiles, hipMemcpyDeviceToHost));
  return GRP_OK;
}

// ---- measurement ---------------------------------------------------------------

// k_decide on caller-provided tile summaries (inspection: the decision kernel against the
// reference's own smoothing / stretch / flank vectors, tests/golden/reference_funcs.json)
int
grp_debug_decide(grp_ctx* c, uint32_t n_reads, const uint64_t* tile0, const grp_tile_summary* tiles, const grp_id_count* lists, uint64_t n_lists, const grp_decide_params* dp, grp_read_decision* out, uint32_t* ids_out, uint8_t* asg_out)
{
  if (!c || !tile0 || !dp || !out || n_reads == 0) {
    return set_err(c, GRP_ERR_INVALID, "grp_debug_decide: bad argument");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t nt = tile0[n_reads];
  uint64_t* d_tile0 = nullptr;
  grp_tile_summary* d_tiles = nullptr;
  grp_id_count* d_lists = nullptr;
  uint32_t* d_ids = nullptr;
  uint8_t* d_asg = nullptr;
  uint64_t* d_scr = nullptr;
  grp_read_decision* d_out = nullptr;
  auto cleanup = [&] {
    (void)hipFree(d_tile0);
    (void)hipFree(d_tiles);
    (void)hipFree(d_lists);
    (void)hipFree(d_ids);
    (void)hipFree(d_asg);
    (void)hipFree(d_scr);
    (void)hipFree(d_out);
  };
#define DBG_TRY(expr)                                                                                                  \
  do {                                                                                                                 \
    hipError_t e_ = (expr);                                                                                            \
    if (e_ != hipSuccess) {                                                                                            \
      cleanup();                                                                                                       \
      return set_err(c, GRP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                                   \
    }                                                                                                                  \
  } while (0)
  DBG_TRY(hipMalloc(&d_tile0, (n_reads + 1) * sizeof(uint64_t)));
  DBG_TRY(hipMalloc(&d_tiles, std::max<uint64_t>(nt, 1) * sizeof(grp_tile_summary)));
  DBG_TRY(hipMalloc(&d_lists, std::max<uint64_t>(n_lists, 1) * sizeof(grp_id_count)));
  DBG_TRY(hipMalloc(&d_ids, std::max<uint64_t>(nt, 1) * 4));
  DBG_TRY(hipMalloc(&d_asg, std::max<uint64_t>(nt, 1)));
  DBG_TRY(hipMalloc(&d_scr, std::max<uint64_t>(nt, 1) * 8));
  DBG_TRY(hipMalloc(&d_out, n_reads * sizeof(grp_read_decision)));
  DBG_TRY(hipMemcpy(d_tile0, tile0, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
  if (nt) {
    DBG_TRY(hipMemcpy(d_tiles, tiles, nt * sizeof(grp_tile_summary), hipMemcpyHostToDevice));
  }
  if (n_lists) {
    DBG_TRY(hipMemcpy(d_lists, lists, n_lists * sizeof(grp_id_count), hipMemcpyHostToDevice));
  }
  DBG_TRY(hipMemset(d_ids, 0, std::max<uint64_t>(nt, 1) * 4));
  DBG_TRY(hipMemset(d_asg, 0, std::max<uint64_t>(nt, 1)));
  DevReads rd{};
  rd.tile0 = d_tile0;
  const uint32_t lds_tiles = decide_lds_tiles(tile0, n_reads);
  DBG_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_decide), hipFuncAttributeMaxDynamicSharedMemorySize, (int)decide_lds_bytes(DECIDE_LDS_MAX_TILES)));
  k_decide<<<dim3(n_reads), dim3(DECIDE_THREADS), decide_lds_bytes(lds_tiles), c->stream>>>(rd, 0, n_reads, *dp, d_tiles, d_lists, std::max<uint64_t>(n_lists, 1), d_ids, d_asg, d_scr, d_out, lds_tiles);
  DBG_TRY(hipGetLastError());
  DBG_TRY(hipStreamSynchronize(c->stream));
  DBG_TRY(hipMemcpy(out, d_out, n_reads * sizeof(grp_read_decision), hipMemcpyDeviceToHost));
  if (ids_out && nt) {
    DBG_TRY(hipMemcpy(ids_out, d_ids, nt * 4, hipMemcpyDeviceToHost));
  }
  if (asg_out && nt) {
    DBG_TRY(hipMemcpy(asg_out, d_asg, nt, hipMemcpyDeviceToHost));
  }
#undef DBG_TRY
  cleanup();
  return GRP_OK;
}

int
grp_debug_stream_stats(const grp_ctx* c, uint64_t out[8])
{
  if (!c || !out) {
    return GRP_ERR_INVALID;
  }
  memcpy(out, c->n_stream_keep, sizeof(c->n_stream_keep));
  out[0] = c->n_stream_coop_refused;
  return GRP_OK;
}

int
grp_set_timing(grp_ctx* c, int enabled)
{
  if (!c) {
    return GRP_ERR_INVALID;
  }
  c->timing = enabled != 0;
  if (enabled >= 2) {
    c->timing_mask = ~0u; // also the latency-critical insert launches
  }
  return GRP_OK;
}

int
grp_get_kernel_stats(grp_ctx* c, grp_kernel_stat out[GRP_K_COUNT])
{
  if (!c || !out) {
    return GRP_ERR_INVALID;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream2));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_events(c);
  memcpy(out, c->kstat, sizeof(c->kstat));
  return GRP_OK;
}

int
grp_reset_kernel_stats(grp_ctx* c)
{
  if (!c) {
    return GRP_ERR_INVALID;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream2));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_events(c);
  memset(c->kstat, 0, sizeof(c->kstat));
  return GRP_OK;
}

void*
grp_stream(grp_ctx* c)
{
  return c ? (void*)c->stream : nullptr;
}

} // extern "C"

extern "C" int
grp_dev_hooks(void)
{
#ifdef GRP_DEV_HOOKS
  return 1;
#else
  return 0;
#endif
}
#include "grp_batch.inc"
#include "grp_verify.inc"
#include "grp_ingest.inc"
#include "grp_ntcard.inc"
#include "grp_comm.inc"
#ifdef GRP_DEV_HOOKS // priced-and-rejected prototypes, measurement only (make DEV=1; include/grpath_dev.h)
#include "grp_pshard.inc"
#endif
