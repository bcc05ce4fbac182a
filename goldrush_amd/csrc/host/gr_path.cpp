// goldrush-path as a function: main() of goldrush_path/goldrush_path.cpp:1096-1275
// with the miBF work going through the engine ABI (grpath.h) instead of
// MIBloomFilter / MIBFConstructSupport / multiLensfrHashIterator.
//
// Same flags, same stderr messages, same output files (<prefix>_<n>.fq in
// --silver_path mode, <prefix>.fa otherwise), same exit codes.
#include "../../../include/grpath_host.h"
#include "gr_classifier.hpp"
#include "gr_fastq.hpp"
#include "gr_opts.hpp"
#include "gr_params.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_set>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#if defined(_OPENMP)
#include <omp.h>
#endif

namespace gr {
namespace {

double
now_s()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// records / bases per host batch (one grp_reads upload each).  GRP_BATCH_RECORDS
// overrides the record count (tests use it to exercise the batch boundaries).
size_t
batch_records()
{
  static const size_t n = [] {
    const char* e = getenv("GRP_BATCH_RECORDS");
    const long v = e ? atol(e) : 0;
    return v > 0 ? (size_t)v : (size_t)16384;
  }();
  return n;
}
#define BATCH_RECORDS batch_records()
constexpr size_t BATCH_BASES = size_t(384) << 20;

size_t
ingest_chunk_bytes()
{
  static const size_t n = [] {
    const char* e = getenv("GRP_INGEST_CHUNK");
    const long long v = e ? atoll(e) : 0;
    return v > 0 ? (size_t)v : (size_t)256 << 20;
  }();
  return n;
}

struct PathRun
{
  Opts opt;
  grp_engine_vt vt{};
  void* ctx = nullptr;
  std::vector<std::string> seeds;
  std::unordered_set<std::string> filter_out_reads;
  std::ofstream out;
  uint32_t world = 1, rank = 0; // ranks sharing the classification windows (one process per GPU)
  // several ranks: every rank hashes the reads of its share of the batches into its bit vector, the
  // vectors are OR-merged before the rank build (round 3: the CLI's fill is no longer replicated)
  bool shard_fill = false;   // batch b belongs to rank b % world
  bool merge_rccl = false;   // ... merged by RCCL inside the engine (else staged through host memory and /dev/shm)
  void* shm = nullptr;       // gr_shm_allgather handle
  int32_t device = -1;       // HIP device ordinal of this rank

  int fail_engine(const char* what)
  {
    std::cerr << "goldrush-path: " << what << ": " << (vt.last_error ? vt.last_error(ctx) : "engine error") << std::endl;
    return 1;
  }
};

// ---- record sources -----------------------------------------------------------
// A batch of consecutive FASTQ records; offsets are relative to `text`.
struct Rec
{
  size_t id_off, id_len, seq_off, seq_len, qual_off, qual_len;
};

struct Batch
{
  const char* text = nullptr;
  uint64_t base_off = 0;            // offset of text[0] in the (decompressed) input stream
  std::vector<Rec> rec;
  std::vector<uint32_t> avg, delta; // calc_phred_average, valid where stats() ran
  std::vector<uint8_t> non_acgt;    // find_first_not_of("ACGTacgt") != npos
  bool seq_is_upper = false;        // the host reader folds case while copying
  std::string id_str(size_t i) const { return std::string(text + rec[i].id_off, rec[i].id_len); }
};

class RecordSource
{
public:
  virtual ~RecordSource() {}
  virtual bool ok() const = 0;
  virtual bool is_fastq() = 0;
  virtual bool next(Batch& b) = 0;
  // Phred statistics (and the ACGT check) of the records with seq_len >= min_len
  virtual void stats(Batch& b, size_t min_len, bool need_acgt) = 0;
  // the selected records as a batch of packed reads on the device
  virtual int upload(Batch& b, const std::vector<uint32_t>& sel, std::vector<uint32_t>& lens, void** reads) = 0;
};

// host reader + host packing (engines without the ingest entry points)
class HostSource : public RecordSource
{
public:
  HostSource(PathRun& run)
    : run_(run)
    , fq_(run.opt.input)
  {}
  bool ok() const override { return fq_.ok(); }
  bool is_fastq() override { return fq_.is_fastq(); }
  bool next(Batch& b) override
  {
    if (!fq_.next_batch(rb_, BATCH_RECORDS, BATCH_BASES)) {
      return false;
    }
    b.text = rb_.text.data();
    b.rec.resize(rb_.rec.size());
    for (size_t i = 0; i < rb_.rec.size(); ++i) {
      const RecordRef& r = rb_.rec[i];
      b.rec[i] = Rec{ r.id_off, r.id_len, r.seq_off, r.seq_len, r.qual_off, r.qual_len };
    }
    b.seq_is_upper = true;
    return true;
  }
  void stats(Batch& b, size_t min_len, bool need_acgt) override
  {
    const size_t n = b.rec.size();
    b.avg.assign(n, 0);
    b.delta.assign(n, 0);
    b.non_acgt.assign(n, 0);
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 8)
#endif
    for (size_t i = 0; i < n; ++i) {
      const Rec& r = b.rec[i];
      if (r.seq_len < min_len) {
        continue;
      }
      calc_phred_average(b.text + r.qual_off, r.qual_len, b.avg[i], b.delta[i]);
      if (need_acgt) {
        const char* s = b.text + r.seq_off; // already upper-cased
        for (size_t j = 0; j < r.seq_len; ++j) {
          const char c = s[j];
          if (c != 'A' && c != 'C' && c != 'G' && c != 'T') {
            b.non_acgt[i] = 1;
            break;
          }
        }
      }
    }
  }
  int upload(Batch& b, const std::vector<uint32_t>& sel, std::vector<uint32_t>& lens, void** reads) override
  {
    const size_t n = sel.size();
    lens.resize(n);
    word_off_.resize(n + 1);
    uint64_t w = 0;
    for (size_t i = 0; i < n; ++i) {
      word_off_[i] = w;
      lens[i] = (uint32_t)b.rec[sel[i]].seq_len;
      w += (b.rec[sel[i]].seq_len + 15) / 16;
    }
    word_off_[n] = w;
    packed_.resize(w ? w : 1);
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 16)
#endif
    for (size_t i = 0; i < n; ++i) {
      pack_2bit(b.text + b.rec[sel[i]].seq_off, b.rec[sel[i]].seq_len, packed_.data() + word_off_[i]);
    }
    return run_.vt.reads_upload(run_.ctx, packed_.data(), word_off_.data(), lens.data(), (uint32_t)n, reads);
  }

private:
  PathRun& run_;
  FastqStream fq_;
  RecordBatch rb_;
  std::vector<uint32_t> packed_;
  std::vector<uint64_t> word_off_;
};

// raw text chunks parsed, filtered and packed on the GPU (grpath_ingest.h).
// A reader thread keeps the NEXT chunk of the file coming (InputFile: parallel pread of a plain
// file, zlib for gzip data) while the caller works on the current one — upload, parse, pack, fill
// or classify — so a pass costs max(read, GPU) per chunk instead of their sum.  Four slots in ONE
// page-locked buffer; every slot has room in front of the bytes read for the unconsumed tail of
// the chunk before it (a partial record), which is copied there before the parse.
class GpuSource : public RecordSource
{
public:
  GpuSource(PathRun& run)
    : run_(run)
    , in_(run.opt.input)
  {
    chunk_ = std::max<size_t>(ingest_chunk_bytes(), 64);
    if (in_.plain_size() != 0) { // a small file: one slot holds it all, no 2 x 256 MiB to allocate and page-lock
      chunk_ = std::min<size_t>(chunk_, std::max<size_t>((size_t)in_.plain_size() + 1, size_t(1) << 16));
    }
    front_ = std::min<size_t>(std::max<size_t>(chunk_ / 16, 4096), (size_t)GRP_FASTQ_PREFETCH_FRONT); // (what a prefetched body may have in front of it, grpath_ingest.h)
    buf_.resize((size_t)kSlots * (front_ + chunk_));
    const char* pin_min = getenv("GRP_PIN_MIN_BYTES"); // tests: page-lock small buffers too
    if (run_.vt.fastq_pin && run_.vt.fastq_unpin && buf_.size() >= (pin_min ? (size_t)atoll(pin_min) : (size_t(8) << 20))) {
      pinned_ = run_.vt.fastq_pin(run_.ctx, buf_.data(), buf_.size()) == GRP_OK;
    }
  }
  ~GpuSource() override
  {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (reader_.joinable()) {
      reader_.join();
    }
    release();
    if (run_.vt.fastq_prefetch) {
      (void)run_.vt.fastq_prefetch(run_.ctx, nullptr, 0); // bodies handed over ahead of a parse that never came (an early end)
    }
    if (getenv("GRP_TRACE_INGEST")) {
      std::cerr << "GRP_TRACE_INGEST source: " << n_pre_[0] << " chunks, next chunk not ready at " << n_pre_[1] << ", uploads started ahead " << n_pre_[2] << "; seconds waiting for the reader " << t_tr_[0]
                << ", in fastq_parse " << t_tr_[1] << ", in fastq_prefetch " << t_tr_[2] << ", copying tails " << t_tr_[3] << std::endl;
    }
    if (pinned_) {
      run_.vt.fastq_unpin(run_.ctx); // before the buffer goes away: the engine cannot know when the host frees memory
    }
  }
  bool ok() const override { return in_.ok(); }
  bool is_fastq() override { return in_.peek() == '@'; } // (asked before the first chunk is read)
  bool next(Batch& b) override
  {
    release();
    if (!reader_.joinable() && !done_) {
      reader_ = std::thread([this] { read_loop(); });
    }
    while (!done_) {
      // the slot handed out by the previous call goes back to the reader, the next one is awaited
      Slot* sl = nullptr;
      {
        std::unique_lock<std::mutex> g(mu_);
        if (held_ >= 0) {
          slot_[held_].uploaded = false;
          slot_[held_].state = Slot::FREE;
          held_ = -1;
          cv_.notify_all();
        }
        const double tw0 = now_s();
        cv_.wait(g, [&] { return slot_[take_].state == Slot::READY; });
        t_tr_[0] += now_s() - tw0;
        sl = &slot_[take_];
        held_ = take_;
        take_ = (take_ + 1) % kSlots;
      }
      const bool eof = sl->eof;
      char* data = slot_data(held_);
      // the tail of the chunk before, in front of what was read
      char* text = nullptr;
      size_t fill = carry_.size() + sl->n;
      if (carry_.size() <= front_) {
        text = data - carry_.size();
        if (!carry_.empty()) { // (a prefetch reads the slot's body only: the room in front of it is free to write)
          const double tc0 = now_s();
          memcpy(text, carry_.data(), carry_.size());
          t_tr_[3] += now_s() - tc0;
        }
      } else {
        // a record longer than a chunk: assembled in a buffer of its own (pageable; rare)
        big_.resize(fill);
        memcpy(big_.data(), carry_.data(), carry_.size());
        memcpy(big_.data() + carry_.size(), data, sl->n);
        text = big_.data();
      }
      const uint64_t text_off = sl->file_off - carry_.size(); // stream offset of text[0]
      if (fill == 0) {
        done_ = true;
        break;
      }
      uint64_t n_rec = 0, used = 0;
      int stopped = 0;
      const double tp0 = now_s();
      const int prc = run_.vt.fastq_parse(run_.ctx, text, fill, eof ? 1 : 0, &fq_, &n_rec, &used, &stopped);
      t_tr_[1] += now_s() - tp0;
      if (prc != GRP_OK) {
        std::cerr << "goldrush-path: FASTQ ingest: " << (run_.vt.last_error ? run_.vt.last_error(run_.ctx) : "failed") << std::endl;
        failed_ = true;
        done_ = true;
        break;
      }
      if (n_rec == 0 && !eof && !stopped) {
        // not one complete record yet: everything is carried into the next chunk
        release();
        carry_.assign(text, text + fill);
        continue;
      }
      carry_.assign(text + used, text + fill);
      if (eof || stopped) {
        done_ = true;
      }
      if (n_rec == 0) {
        release();
        continue;
      }
      meta_.resize(n_rec);
      run_.vt.fastq_records(fq_, meta_.data());
      b.text = text;
      b.base_off = text_off;
      b.rec.resize(n_rec);
      for (size_t i = 0; i < n_rec; ++i) {
        const grp_fastq_record& m = meta_[i];
        b.rec[i] = Rec{ (size_t)m.id_off, m.id_len, (size_t)m.seq_off, m.seq_len, (size_t)m.qual_off, m.qual_len };
      }
      b.seq_is_upper = false;
      prefetch_next();
      return true;
    }
    return false;
  }
  // The uploads of the next TWO chunks' bodies, started as soon as the reader has them (round 5).  A copy issued behind the
  // fill of the current chunk only begins when that has ended (tools/dev/r5_ingest_timeline.sh: fill, then copy, then parse,
  // one after the other); issued one chunk ahead — with the tail of the chunk before copied in front first — it still lay on
  // the path of every chunk (copy 5.5 ms + parse + the host's share against a fill of 7 ms).  The body does not depend on
  // that tail: it goes up two chunks ahead, the engine leaves room in front of it and the parse uploads the tail alone
  // (grpath_ingest.h: grp_fastq_prefetch).
  void prefetch_next()
  {
    ++n_pre_[0];
    if (!run_.vt.fastq_prefetch || done_) {
      return;
    }
    for (int ahead = 0; ahead < 2; ++ahead) {
      const int j = (take_ + ahead) % kSlots;
      size_t n = 0;
      {
        std::lock_guard<std::mutex> g(mu_);
        if (slot_[j].state != Slot::READY) {
          n_pre_[1] += ahead == 0;
          return; // the reader is still at it (and the slots behind it are not ready either)
        }
        if (slot_[j].eof && slot_[j].n == 0) {
          return;
        }
        n = slot_[j].n; // (a READY slot is the consumer's until it releases it)
      }
      if (slot_[j].uploaded || n == 0) {
        continue;
      }
      const double tq0 = now_s();
      const int qrc = run_.vt.fastq_prefetch(run_.ctx, slot_data(j), n);
      t_tr_[2] += now_s() - tq0;
      if (qrc != GRP_OK) {
        return; // no free buffer on the device: later
      }
      slot_[j].uploaded = true;
      ++n_pre_[2];
    }
  }
  void stats(Batch& b, size_t min_len, bool) override
  {
    const size_t n = b.rec.size();
    b.avg.assign(n, 0);
    b.delta.assign(n, 0);
    b.non_acgt.assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
      b.non_acgt[i] = (meta_[i].flags & GRP_FQ_NON_ACGT) ? 1 : 0;
      if (b.rec[i].seq_len >= min_len) {
        // the device summed left to right in double; log10 / truncation happen here
        phred_from_sums(meta_[i].phred_sum, meta_[i].phred_first, b.rec[i].qual_len, b.avg[i], b.delta[i]);
      }
    }
  }
  int upload(Batch& b, const std::vector<uint32_t>& sel, std::vector<uint32_t>& lens, void** reads) override
  {
    lens.resize(sel.size());
    for (size_t i = 0; i < sel.size(); ++i) {
      lens[i] = (uint32_t)b.rec[sel[i]].seq_len;
    }
    return run_.vt.fastq_pack(run_.ctx, fq_, sel.data(), (uint32_t)sel.size(), reads);
  }
  bool failed() const { return failed_; }

private:
  struct Slot
  {
    enum State { FREE, READY } state = FREE;
    size_t n = 0;          // bytes read
    uint64_t file_off = 0; // stream offset of the first of them
    bool eof = false;      // the data ends with this chunk
    bool uploaded = false; // consumer's: the engine has been handed the body ahead of its parse (prefetch_next)
  };
  char* slot_data(int i) { return buf_.data() + (size_t)i * (front_ + chunk_) + front_; }
  void read_loop()
  {
    uint64_t off = 0;
    for (int i = 0;; i = (i + 1) % kSlots) {
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return stop_ || slot_[i].state == Slot::FREE; });
        if (stop_) {
          return;
        }
      }
      size_t got = 0;
      bool eof = false;
      while (!eof && got < chunk_) {
        const size_t k = in_.read(slot_data(i) + got, chunk_ - got);
        if (k == 0) {
          eof = true;
        }
        got += k;
      }
      {
        std::lock_guard<std::mutex> g(mu_);
        slot_[i].n = got;
        slot_[i].file_off = off;
        slot_[i].eof = eof;
        slot_[i].state = Slot::READY;
      }
      cv_.notify_all();
      off += got;
      if (eof) {
        return;
      }
    }
  }
  void release()
  {
    if (fq_) {
      run_.vt.fastq_free(fq_);
      fq_ = nullptr;
    }
  }
  PathRun& run_;
  InputFile in_;
  std::vector<char> buf_;  // kSlots x [front | chunk]
  uint64_t n_pre_[3] = { 0, 0, 0 }; // developer trace: chunks returned, next slot not ready, prefetches the engine took
  double t_tr_[4] = { 0, 0, 0, 0 };  // developer trace: seconds waiting for the reader / in the parse call / in the prefetch calls / copying tails
  std::vector<char> carry_; // unconsumed tail of the chunk before
  std::vector<char> big_;
  size_t chunk_ = 0, front_ = 0;
  bool pinned_ = false;
  std::thread reader_;
  std::mutex mu_;
  std::condition_variable cv_;
  // four chunk buffers (round 5; two before): the caller's batch lives in one, the bodies of the next two have been handed to
  // the engine ahead (prefetch_next) and the reader fills the fourth
  static constexpr int kSlots = 4;
  Slot slot_[kSlots];
  int take_ = 0;  // the slot the next chunk arrives in
  int held_ = -1; // the slot the caller's current batch lives in
  bool stop_ = false;
  bool done_ = false, failed_ = false;
  void* fq_ = nullptr;
  std::vector<grp_fastq_record> meta_;
};

std::unique_ptr<RecordSource>
open_source(PathRun& run)
{
  const bool gpu = run.vt.fastq_parse && run.vt.fastq_records && run.vt.fastq_pack && run.vt.fastq_free && !getenv("GRP_HOST_INGEST");
  if (gpu) {
    return std::unique_ptr<RecordSource>(new GpuSource(run));
  }
  return std::unique_ptr<RecordSource>(new HostSource(run));
}

// goldrush_path.cpp:1109-1112 -> calc_ntcard_genome_size (ntcard.hpp:248-275): the
// hash stream of EVERY record (no read filter) sampled on the device; the host only
// turns the zero-bucket counts into F0.  Returns -1 to continue.
int
calc_ntcard_genome_size(PathRun& run, uint64_t& genome_size)
{
  const Opts& opt = run.opt;
  const unsigned k = (unsigned)opt.kmer_size, h = (unsigned)opt.hash_num;
  std::cerr << "Calculating expected entries" << std::endl;
  const double s_time = now_s();
  uint64_t input_bytes = 0;
  {
    std::ifstream in(opt.input, std::ifstream::ate | std::ifstream::binary); // getInf (:44-49)
    if (in) {
      input_bytes = (uint64_t)in.tellg();
    }
  }
  const unsigned sbits = ntcard_sbits(input_bytes);
  if (run.vt.ntcard_begin(run.ctx, sbits) != GRP_OK) {
    return run.fail_engine("ntcard tables");
  }
  auto src = open_source(run);
  Batch b;
  std::vector<uint32_t> sel, lens, extra, run_extra, packed;
  std::vector<uint64_t> word_off;
  std::vector<std::pair<size_t, size_t>> runs;
  // (another format fails in fill_bit_vector, as in the reference; nothing is counted here)
  const bool readable = src->ok() && src->is_fastq();
  while (readable && src->next(b)) {
    src->stats(b, 0, true);
    sel.clear();
    // records with other characters go in as their ACGT runs (rare: host packing)
    lens.clear();
    extra.clear();
    word_off.assign(1, 0);
    packed.clear();
    for (size_t i = 0; i < b.rec.size(); ++i) {
      if (!b.non_acgt[i]) {
        if (b.rec[i].seq_len >= k) {
          sel.push_back((uint32_t)i);
        }
        continue;
      }
      const char* seq = b.text + b.rec[i].seq_off;
      ntcard_split(seq, b.rec[i].seq_len, k, h, runs, run_extra);
      for (size_t r = 0; r < runs.size(); ++r) {
        const size_t w0 = packed.size();
        packed.resize(w0 + (runs[r].second + 15) / 16);
        pack_2bit(seq + runs[r].first, runs[r].second, packed.data() + w0);
        lens.push_back((uint32_t)runs[r].second);
        word_off.push_back(packed.size());
      }
      extra.insert(extra.end(), run_extra.begin(), run_extra.end());
    }
    if (!lens.empty()) {
      void* hr = nullptr;
      if (packed.empty()) {
        packed.push_back(0);
      }
      if (run.vt.reads_upload(run.ctx, packed.data(), word_off.data(), lens.data(), (uint32_t)lens.size(), &hr) != GRP_OK) {
        return run.fail_engine("uploading reads");
      }
      const int rc = run.vt.ntcard_add(run.ctx, hr, 0, (uint32_t)lens.size(), extra.data());
      run.vt.reads_free(hr);
      if (rc != GRP_OK) {
        return run.fail_engine("ntcard");
      }
    }
    if (!sel.empty()) {
      void* hr = nullptr;
      if (src->upload(b, sel, lens, &hr) != GRP_OK) {
        return run.fail_engine("uploading reads");
      }
      // stRead (:96-112): multiLensfrHashIterator over the whole sequence, ntComp per seed
      const int rc = run.vt.ntcard_add(run.ctx, hr, 0, (uint32_t)sel.size(), nullptr);
      run.vt.reads_free(hr);
      if (rc != GRP_OK) {
        return run.fail_engine("ntcard");
      }
    }
  }
  std::vector<uint64_t> zeros((size_t)h * 2, 0);
  if (run.vt.ntcard_finish(run.ctx, zeros.data()) != GRP_OK) {
    return run.fail_engine("ntcard");
  }
  std::cerr << "Reapeat profile estimated using ntCard in (sec): " << std::setprecision(4) << std::fixed << now_s() - s_time << "\n";
  // (setprecision(4) << fixed stay set on std::cerr, as in the reference: the banner
  // below prints "occupancy: 0.1000" after an ntcard pass)
  genome_size = 0;
  for (unsigned i = 0; i < h; ++i) {
    const uint64_t f0 = ntcard_f0(zeros[2 * i], zeros[2 * i + 1], sbits);
    std::cerr << "Expected entries for seed pattern " << run.seeds[i] << " : " << f0 << std::endl;
    genome_size += f0;
  }
  std::cerr << "Total expected entries for seed patterns: " << genome_size << std::endl;
  return -1;
}

// goldrush_path.cpp:79-107.  Deterministic form of the OpenMP loop: the first
// 50000 eligible reads in file order fill the sample; every one of the `jobs`
// threads performs one more fetch_add before it breaks, which only moves the
// median index (calc_median takes vec[n/2] of the descending sort).
int
calc_min_phred_threshold(PathRun& run)
{
  constexpr size_t MEDIAN_SAMPLES_NEEDED = 50000;
  constexpr uint32_t MINIMUM_PHRED_THRESHOLD = 10;
  Opts& opt = run.opt;
  if (opt.phred_min != 0) {
    return -1;
  }
  std::cerr << "Calculating minimum phred score via median" << std::endl;
  std::vector<uint32_t> scores(MEDIAN_SAMPLES_NEEDED, 0);
  size_t taken = 0, over = 0;
  auto src = open_source(run);
  Batch b;
  bool done = false;
  while (!done && src->ok() && src->next(b)) {
    src->stats(b, opt.min_length, false);
    for (size_t i = 0; i < b.rec.size(); ++i) {
      if (b.rec[i].seq_len < opt.min_length) {
        continue;
      }
      if (taken >= MEDIAN_SAMPLES_NEEDED) {
        if (++over >= opt.jobs) {
          done = true;
          break;
        }
        continue;
      }
      scores[taken++] = b.avg[i];
    }
  }
  const size_t n = taken + over;
  std::sort(scores.begin(), scores.end(), std::greater<uint32_t>());
  opt.phred_min = std::max(MINIMUM_PHRED_THRESHOLD, scores[n / 2]);
  if (opt.debug) { // log_phred_calculations (:60-70)
    std::cerr << "Number of reads used to calculate median: " << n << std::endl;
    std::cerr << "Median array: ";
    for (const uint32_t v : scores) {
      std::cerr << v << " ";
    }
    std::cerr << std::endl;
  }
  if (opt.verbose) {
    std::cerr << "Minimum phred score calculated with median: " << opt.phred_min << std::endl;
  }
  return -1;
}

// goldrush_path.cpp:235-339
// ---- reads kept on the device between the passes (round 3) -----------------------
// The reference reads its input three times (Phred median, fill, classification:
// goldrush_path.cpp:79-107, 235-339, 1210-1256).  Here the fill pass is the only full parse:
// the packed reads it uploads stay in HBM (2 bits per base: 62.5 GB for C2's 10 M reads) and
// the classification pass runs over those batches; the text of a read that is written out
// (an inserted read) is fetched from the input file by offset.  Plain, seekable input without
// a -f list only (gzip / pipes / --debug take the two-pass form); GRP_RESIDENT=off switches it
// off, GRP_RESIDENT_MAX_GB caps the packed bytes kept (default 100).
struct RecLoc
{
  uint64_t off;     // offset of the record's id in the input file
  uint32_t id_len;
  uint32_t seq_rel; // seq / qual relative to `off`
  uint32_t seq_len;
  uint64_t qual_rel;
  uint32_t qual_len;
};

struct ResidentBatch
{
  void* reads = nullptr;
  std::vector<uint32_t> lens, skipped_before;
  uint32_t skipped_after = 0;
  std::vector<RecLoc> loc;
  std::vector<uint64_t> name_hash; // FNV-1a of the id: candidates for filter_out_reads (names are compared through the file)
};

uint64_t
fnv1a(const char* p, size_t n)
{
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
  }
  return h;
}

struct Resident
{
  bool on = false;
  int fd = -1;
  uint64_t packed_bytes = 0, max_bytes = 0;
  std::vector<ResidentBatch> batches;
  void drop(PathRun& run)
  {
    for (ResidentBatch& rb : batches) {
      if (rb.reads) {
        run.vt.reads_free(rb.reads);
      }
    }
    batches.clear();
    on = false;
  }
};

// plain regular file (no gzip magic)?  Returns a descriptor for pread, -1 otherwise.
int
open_plain_input(const std::string& path)
{
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) {
    return -1;
  }
  struct stat st;
  unsigned char magic[2] = { 0, 0 };
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b)) {
    close(fd);
    return -1;
  }
  return fd;
}

int
fill_bit_vector(PathRun& run, Resident& res)
{
  const Opts& opt = run.opt;
  std::cerr << "inserting bit vector" << std::endl;
  const double s_time = now_s();
  auto src = open_source(run);
  if (!src->ok() || !src->is_fastq()) {
    std::cerr << "Gold Path requires fastq format" << std::endl;
    return 1;
  }
  size_t num_reads = 0, num_passed_reads = 0, by_phred = 0, by_delta = 0, by_length = 0, by_bases = 0;
  Batch b;
  void* prev = nullptr;
  std::vector<uint32_t> sel, lens;
  uint32_t lead_skipped = 0;
  uint64_t n_batches = 0;
  // developer trace (GRP_TRACE_INGEST): where the host's time of this pass goes, per call site
  static const bool trace = getenv("GRP_TRACE_INGEST") != nullptr;
  double t_next = 0, t_filter = 0, t_upload = 0, t_fill = 0, t_keep = 0;
  for (;;) {
    const double tn0 = trace ? now_s() : 0.0;
    if (!src->next(b)) {
      break;
    }
    const double tn1 = trace ? now_s() : 0.0;
    t_next += tn1 - tn0;
    src->stats(b, opt.min_length, true);
    sel.clear();
    ResidentBatch keep;
    uint32_t skipped = 0;
    for (size_t i = 0; i < b.rec.size(); ++i) {
      ++num_reads;
      ++skipped; // (taken back below when the record is selected)
      if (b.rec[i].seq_len < opt.min_length) { // :261-265
        ++by_length;
        continue;
      }
      if (opt.debug) { // :267-273 (file order here; the reference prints from an OpenMP region)
        std::cerr << "phred avg: " << b.avg[i] << "\n"
                  << "phred delta: " << b.delta[i] << std::endl;
      }
      const bool low = b.avg[i] < opt.phred_min, hairpin = b.delta[i] >= opt.phred_delta;
      if (low || hairpin) { // :275-292
        if (opt.verbose) {
          by_phred += low ? 1 : 0;
          by_delta += hairpin ? 1 : 0;
        }
        run.filter_out_reads.insert(b.id_str(i));
        continue;
      }
      if (b.non_acgt[i]) { // :293-301
        ++by_bases;
        run.filter_out_reads.insert(b.id_str(i));
        continue;
      }
      ++num_passed_reads;
      sel.push_back((uint32_t)i);
      if (res.on) {
        // what the classification pass needs of this read: it is classified iff it is long enough and
        // not in filter_out_reads (read_hashing.cpp:35-42) — without a -f list exactly the reads selected here
        const Rec& r = b.rec[i];
        keep.skipped_before.push_back(skipped - 1);
        keep.loc.push_back(RecLoc{ b.base_off + r.id_off, (uint32_t)r.id_len, (uint32_t)(r.seq_off - r.id_off), (uint32_t)r.seq_len, (uint64_t)(r.qual_off - r.id_off), (uint32_t)r.qual_len });
        keep.name_hash.push_back(fnv1a(b.text + r.id_off, r.id_len));
      }
      skipped = 0;
    }
    if (res.on && sel.empty() && !res.batches.empty()) {
      res.batches.back().skipped_after += skipped; // a batch without a single selected read
      skipped = 0;
    }
    const double tf1 = trace ? now_s() : 0.0;
    t_filter += tf1 - tn1;
    if (!sel.empty()) {
      void* h = nullptr;
      if (src->upload(b, sel, lens, &h) != GRP_OK) {
        return run.fail_engine("uploading reads");
      }
      const double tu1 = trace ? now_s() : 0.0;
      t_upload += tu1 - tf1;
      if (prev) {
        run.vt.reads_free(prev); // waits for the previous batch's kernel
        prev = nullptr;
      }
      // multiLensfrHashIterator itr(record.seq, seeds); miBFCS.insertBV(itr)  (:304-305)
      static const bool trace_nofill = trace && getenv("GRP_TRACE_NOFILL") != nullptr; // developer: the pass without its fill kernels (timing only: the filter stays empty)
      const bool mine = !trace_nofill && (!run.shard_fill || (n_batches % run.world) == run.rank);
      ++n_batches;
      if (mine && run.vt.bv_insert(run.ctx, h, 0, (uint32_t)sel.size()) != GRP_OK) {
        return run.fail_engine("bit-vector insert");
      }
      const double tb1 = trace ? now_s() : 0.0;
      t_fill += tb1 - tu1;
      if (res.on) {
        for (uint32_t l : lens) {
          res.packed_bytes += ((uint64_t)l + 15) / 16 * 4;
        }
        if (res.packed_bytes > res.max_bytes) {
          res.drop(run); // too much to keep: the classification pass parses the input again
          prev = h;
        } else {
          keep.reads = h;
          keep.lens = lens;
          keep.skipped_after = skipped;
          res.batches.push_back(std::move(keep));
        }
      } else {
        prev = h;
      }
    } else if (res.on && res.batches.empty()) {
      lead_skipped += skipped; // records in front of the first selected read of the input
    }
  }
  if (res.on && !res.batches.empty() && lead_skipped) {
    res.batches.front().skipped_before.front() += lead_skipped;
  }
  if (prev) {
    run.vt.reads_free(prev);
  }
  if (opt.verbose) {
    std::cerr << "num_passed_reads: " << num_passed_reads << "\n"
              << "num_reads: " << num_reads << "\n"
              << "num_reads - num_passed_reads: " << num_reads - num_passed_reads << "\n"
              << "num_reads - num_passed_reads / num_reads: " << floor((double)(num_reads - num_passed_reads) / num_reads) << "\n"
              << "num_reads_skipped_by_phred: " << by_phred << "\n"
              << "num_reads_skipped_by_delta: " << by_delta << "\n"
              << "num_reads_skipped_by_length: " << by_length << "\n"
              << "num_reads_skipped_by_invalid_bases: " << by_bases << "\n"
              << "Total reads skipped: " << by_phred + by_delta + by_length + by_bases << std::endl;
  }
  if (num_passed_reads == 0) {
    std::cerr << "Error: no reads passed the Phred score and min length requirements\n"
              << "Try again with a lower Phred threshold or lower min length" << std::endl;
    return 1;
  }
  const double ts0 = trace ? now_s() : 0.0;
  if (run.vt.sync(run.ctx) != GRP_OK) {
    return run.fail_engine("bit-vector insert");
  }
  if (trace) {
    (void)t_keep;
    std::cerr << "GRP_TRACE_INGEST fill pass: " << n_batches << " batches; host seconds in next() (read + parse) " << t_next << ", filter " << t_filter << ", upload (pack) " << t_upload << ", bv_insert call " << t_fill
              << ", final sync " << now_s() - ts0 << std::endl;
  }
  if (run.shard_fill) {
    // the bitwise OR of the ranks' vectors (the fill is order-free; csrc/host/gr_ranks.cpp)
    const int mrc = gr_fill_merge_run(&run.vt, run.ctx, run.shm, run.world, run.rank, run.merge_rccl ? GR_MERGE_RCCL : GR_MERGE_STAGED);
    if (mrc == 1) {
      return run.fail_engine(run.merge_rccl ? "merging the bit vectors of the ranks (RCCL)" : "merging the bit vectors of the ranks");
    }
    if (mrc != 0) {
      std::cerr << "goldrush-path: the exchange between the ranks failed" << std::endl;
      return 1;
    }
  }
  std::cerr << "finished inserting bit vector" << std::endl;
  std::cerr << "in " << std::setprecision(4) << std::fixed << now_s() - s_time << "\n";
  return -1;
}

// classification sink: writes one committed read (goldrush_path.cpp:996-1002, 1055-1070)
struct SinkState
{
  PathRun* run = nullptr;
  const Batch* batch = nullptr;
  const std::vector<uint32_t>* sel = nullptr;
  // reads kept on the device: the record's text comes from the input file
  const ResidentBatch* resident = nullptr;
  int fd = -1;
  std::vector<char> text;
  std::string upper;
  // --debug: the records that are not classified, in file order ({record, 1 = too short / 2 = filtered}),
  // and for every classified read the number of them in front of it
  const std::vector<std::pair<uint32_t, uint8_t>>* skipped = nullptr;
  const std::vector<uint32_t>* skipped_before = nullptr;
  size_t skipped_done = 0;
};

// --debug: what process_read prints for the records it skips (goldrush_path.cpp:907-932)
void
debug_skipped(const SinkState& st, size_t from, size_t to)
{
  const Batch& b = *st.batch;
  for (size_t i = from; i < to; ++i) {
    const auto& sk = (*st.skipped)[i];
    std::cerr << (sk.second == 1 ? "too short" : "hairpin or quality too low or invalid bases") << std::endl;
    std::cerr << "skipping: " << b.id_str(sk.first) << std::endl;
  }
}

// --debug: in front of a classified read (:907-941)
void
debug_sink(void* user, uint32_t read)
{
  SinkState& st = *static_cast<SinkState*>(user);
  const size_t n = (*st.skipped_before)[read];
  debug_skipped(st, st.skipped_done, st.skipped_done + n);
  st.skipped_done += n;
  const Batch& b = *st.batch;
  const uint32_t i = (*st.sel)[read];
  std::cerr << "name: " << b.id_str(i) << std::endl;
  std::cerr << "num tiles: " << b.rec[i].seq_len / st.run->opt.tile_length << std::endl;
}

double
commit_sink(void* user, const gr_commit* c)
{
  SinkState& st = *static_cast<SinkState*>(user);
  PathRun& run = *st.run;
  if (c->dec.kind != DEC_INSERT_WHOLE && c->dec.kind != DEC_INSERT_TRIMMED) {
    return 0.0;
  }
  if (input_failed()) {
    return 0.0;
  }
  Batch fetched; // resident mode: one record read back from the file
  Rec fetched_rec{};
  if (st.resident) {
    const RecLoc& l = st.resident->loc[c->read];
    const size_t n = (size_t)l.qual_rel + l.qual_len;
    st.text.resize(n);
    size_t got = 0;
    while (got < n) {
      const ssize_t k = pread(st.fd, st.text.data() + got, n - got, (off_t)(l.off + got));
      if (k <= 0) {
        // not exit(1) from inside the commit callback (a streaming window may be parked on this thread's word,
        // ADVICE r03): the failure is noted, this and the later records are not written, and the program ends with
        // an error behind the pass
        note_input_failure("cannot read the record back from " + run.opt.input);
        return 0.0;
      }
      got += (size_t)k;
    }
    fetched.text = st.text.data();
    fetched.seq_is_upper = false;
    fetched_rec = Rec{ 0, l.id_len, l.seq_rel, l.seq_len, (size_t)l.qual_rel, l.qual_len };
  }
  const Batch& b = st.resident ? fetched : *st.batch;
  const Rec& r = st.resident ? fetched_rec : b.rec[(*st.sel)[c->read]];
  const char first = run.opt.silver_path ? '@' : '>';
  const char* seq = b.text + r.seq_off;
  const char* qual = b.text + r.qual_off;
  size_t off = 0, n_seq = r.seq_len, n_qual = r.qual_len;
  const char* suffix = "_untrimmed\n";
  if (c->dec.kind == DEC_INSERT_TRIMMED) {
    suffix = "_trimmed\n";
    const size_t tile = run.opt.tile_length;
    off = (size_t)c->dec.trim_start * tile;
    const size_t end_pos = (c->dec.trim_end == c->dec.num_tiles - 1) ? std::string::npos : (size_t)(c->dec.trim_end - c->dec.trim_start + 1) * tile;
    n_seq = std::min(end_pos, r.seq_len - off);
    n_qual = (off <= r.qual_len) ? std::min(end_pos, r.qual_len - off) : 0;
  }
  std::ofstream& o = run.out;
  o.put(first);
  o.write(b.text + r.id_off, (std::streamsize)r.id_len);
  o << suffix;
  if (b.seq_is_upper) {
    o.write(seq + off, (std::streamsize)n_seq);
  } else {
    // SeqReader folds the case of what the reference later writes
    st.upper.assign(seq + off, n_seq);
    for (char& ch : st.upper) {
      if (ch >= 'a' && ch <= 'z') {
        ch = (char)(ch - 32);
      }
    }
    o.write(st.upper.data(), (std::streamsize)n_seq);
  }
  o << std::endl;
  const size_t qoff = std::min(off, r.qual_len);
  if (run.opt.silver_path) {
    o << "+\n";
    o.write(qual + qoff, (std::streamsize)n_qual);
    o << std::endl;
  }
  return sum_phred(qual + qoff, n_qual);
}

void
rollover_sink(void* user, uint64_t new_path)
{
  // golden_path_vec.pop_back(); emplace_back(ofstream(prefix + "_" + path + ".fq"))  (:182-184)
  PathRun& run = *static_cast<SinkState*>(user)->run;
  run.out.close();
  run.out.open(run.rank == 0 ? run.opt.prefix_file + "_" + std::to_string(new_path) + ".fq" : std::string("/dev/null"));
}

} // namespace
} // namespace gr

extern "C" int
gr_path_main(int argc, char** argv, const grp_engine_vt* vt)
{
  using namespace gr;
  PathRun run;
  run.vt = *vt;
  Opts& opt = run.opt;
  // Several GPUs of one node: one process per GPU (GRP_WORLD / GRP_RANK, or the launcher's
  // WORLD_SIZE / RANK; LOCAL_RANK picks the device).  Every rank runs the whole program on
  // its own replica of the miBF — the same input, the same passes, every insert — and the
  // ranks share the QUERY work of each classification window (gr_classifier: the window's
  // reads are split / striped over the ranks, the 32-byte decisions all-gathered through
  // /dev/shm).  Rank 0 writes the files and the log; the others are silent.
  auto env_u32 = [](const char* a, const char* b, uint32_t dflt) {
    const char* e = getenv(a);
    if (!e || !*e) {
      e = getenv(b);
    }
    return (e && *e) ? (uint32_t)strtoul(e, nullptr, 10) : dflt;
  };
  run.world = std::max<uint32_t>(1, env_u32("GRP_WORLD", "WORLD_SIZE", 1));
  run.rank = env_u32("GRP_RANK", "RANK", 0);
  if (run.rank >= run.world) {
    std::cerr << "goldrush-path: rank " << run.rank << " outside the world of " << run.world << std::endl;
    return 1;
  }
  struct Quiet // ranks > 0: nothing on stdout / stderr
  {
    std::streambuf *cerr_buf = nullptr, *cout_buf = nullptr;
    std::ofstream null;
    ~Quiet()
    {
      if (cerr_buf) {
        std::cerr.rdbuf(cerr_buf);
        std::cout.rdbuf(cout_buf);
      }
    }
  } quiet;
  if (run.rank != 0) {
    quiet.null.open("/dev/null");
    quiet.cerr_buf = std::cerr.rdbuf(quiet.null.rdbuf());
    quiet.cout_buf = std::cout.rdbuf(quiet.null.rdbuf());
  }
  clear_input_failure();
  // an input that could not be read to its end is an error, not a shorter input (gr_fastq.hpp): checked behind every pass
  auto input_error = [] {
    std::string what;
    if (!input_failed(&what)) {
      return false;
    }
    std::cerr << "ERROR: " << what << std::endl;
    return true;
  };
  int ec = process_options(opt, argc, argv);
  if (ec >= 0) {
    return ec;
  }
  struct ShmGuard
  {
    void* h = nullptr;
    ~ShmGuard() { gr_shm_allgather_close(h); }
  } shm;
  if (run.world > 1) {
    const char* key = getenv("GRP_SHM_KEY");
    std::string k = key && *key ? key : std::string("path_") + (getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0") + "_" + std::to_string((unsigned long)getppid());
    shm.h = gr_shm_allgather_open(run.world, run.rank, k.c_str(), 120.0);
    if (!shm.h) {
      std::cerr << "goldrush-path: cannot set up the exchange between the " << run.world << " ranks (/dev/shm/grp_" << k << ")" << std::endl;
      return 1;
    }
  }
#if defined(_OPENMP)
  // the reference uses -j for its OpenMP regions; here it only drives the host
  // side (parsing, Phred, packing, decisions) and is capped at the machine size
  omp_set_num_threads((int)std::max<size_t>(1, std::min<size_t>(opt.jobs, (size_t)effective_cpus())));
#endif
  run.seeds = make_seed_pattern(opt.seed_preset, (unsigned)opt.kmer_size, (unsigned)opt.weight, (unsigned)opt.hash_num, true);
  const bool use_ntcard = opt.hash_universe == 0 && opt.ntcard;
  if (use_ntcard && !(run.vt.ntcard_begin && run.vt.ntcard_add && run.vt.ntcard_finish && run.vt.set_filter_size)) {
    std::cerr << "goldrush-path: --ntcard needs an engine with the ntcard entry points" << std::endl;
    return 1;
  }
  if (opt.hash_universe == 0 && !use_ntcard) {
    opt.hash_universe = hash_universe(opt.weight, opt.genome_size, opt.hash_num);
  }
  const std::string what = opt.silver_path ? std::to_string(opt.max_paths) + " silver path(s)" : std::string("the golden path");
  // the engine is set up before the first pass over the input (the Phred median
  // pass already uses the GPU ingest); the reference's messages keep their order.
  // With --ntcard the filter size is only known after the ntcard pass.
  uint64_t filter_size = use_ntcard ? 0 : calc_optimal_size(opt.hash_universe, 1, opt.occupancy);
  {
    std::vector<const char*> sp;
    for (const auto& s : run.seeds) {
      sp.push_back(s.c_str());
    }
    grp_params gp{};
    gp.struct_size = sizeof(gp);
    gp.k = (uint32_t)opt.kmer_size;
    gp.h = (uint32_t)opt.hash_num;
    gp.tile = (uint32_t)opt.tile_length;
    gp.m = filter_size;
    gp.seeds = sp.data();
    gp.device = run.world > 1 ? (int32_t)env_u32("GRP_LOCAL_RANK", "LOCAL_RANK", run.rank) : -1;
    run.device = gp.device;
    if (run.vt.create(&gp, &run.ctx) != GRP_OK) {
      std::cerr << "goldrush-path: cannot set up the MI355X engine: " << (run.vt.last_error ? run.vt.last_error(nullptr) : "") << std::endl;
      return 1;
    }
  }
  struct CtxGuard
  {
    PathRun& r;
    ~CtxGuard()
    {
      if (r.ctx) {
        r.vt.destroy(r.ctx);
      }
    }
  } guard{ run };
  if (use_ntcard) {
    uint64_t genome_size = 0;
    ec = calc_ntcard_genome_size(run, genome_size);
    if (ec >= 0) {
      return ec;
    }
    if (input_error()) {
      return 1;
    }
    opt.hash_universe = genome_size;
    filter_size = calc_optimal_size(opt.hash_universe, 1, opt.occupancy);
    if (run.vt.set_filter_size(run.ctx, filter_size) != GRP_OK) {
      return run.fail_engine("allocating the bit vector");
    }
  }
  if (run.vt.occupancy_hint && opt.occupancy > 0.0 && opt.occupancy < 1.0) {
    // the ID / count tables of phase 2 are sized by the occupancy the fill will reach: -o, by construction of the filter
    // size (:1183-1184) — the engine allocates them beside the fill instead of between the passes (a hint: ignored where wrong)
    (void)run.vt.occupancy_hint(run.ctx, opt.occupancy);
  }
  if (run.world > 1 && shm.h && gr_ranks_share_device(shm.h, run.world, run.device) == 1) {
    // ranks on one device (a test box): their windows' launches must all fit it at once — the in-launch inserts of
    // the ranks' windows wait grid-wide (read at the engine's first streaming launch; an explicit setting stands)
    setenv("GRP_STREAM_WGS_PER_CU", "1", 0);
  }
  if (run.world > 1 && !getenv("GRP_REPLICATED_FILL")) {
    // A sharded fill needs a way to merge: RCCL inside the engine (one GPU per rank), else the
    // host-staged form (ranks sharing a device, engines without RCCL).  Every decision below is
    // taken by ALL ranks alike (flags all-gathered through /dev/shm).
    run.shm = shm.h;
    const int plan = gr_fill_merge_plan(&run.vt, run.ctx, shm.h, run.world, run.rank, run.device);
    run.merge_rccl = plan == GR_MERGE_RCCL;
    run.shard_fill = plan != GR_MERGE_NONE;
  }
  ec = calc_min_phred_threshold(run);
  if (ec >= 0) {
    return ec;
  }
  if (input_error()) {
    return 1;
  }
  std::cerr << "Calculating " << what << "\n"
            << "Using:\n"
            << "\ttile length: " << opt.tile_length << "\n"
            << "\tblock size: " << opt.block_size << "\n"
            << "\tseed patterns: " << opt.hash_num << "\n"
            << "\tthreshold: " << opt.threshold << "\n"
            << "\tbase seed pattern: " << run.seeds[0] << "\n"
            << "\tminimum unassigned tiles: " << opt.unassigned_min << "\n"
            << "\tmaximum assigned tiles: " << opt.assigned_max << "\n"
            << "\texpected hash space: " << opt.hash_universe << "\n"
            << "\tminimum average phred quality score: " << opt.phred_min << "\n"
            << "\tmaximum average phred delta between first and second half of read: " << opt.phred_delta << "\n"
            << "\toccupancy: " << opt.occupancy << "\n"
            << "\tjobs: " << opt.jobs << std::endl;
  if (!opt.filter_file.empty()) {
    std::cerr << "Using only reads not found in: " << opt.filter_file << std::endl;
    std::ifstream in(opt.filter_file);
    std::string name;
    while (in >> name) {
      run.filter_out_reads.insert(name);
    }
  }
  run.out.open(run.rank != 0 ? std::string("/dev/null") : opt.silver_path ? opt.prefix_file + "_1.fq" : opt.prefix_file + ".fa");
  double s_time = now_s();
  std::cerr << "allocating bit vector" << std::endl;
  std::cerr << "m_filterSize: " << filter_size << std::endl;
  std::cerr << "finished allocating bit vector" << std::endl;
  std::cerr << "in " << std::setprecision(4) << std::fixed << now_s() - s_time << "\n";
  std::cerr << "opening: " << opt.input << std::endl;

  Resident res;
  struct ResGuard
  {
    PathRun& run;
    Resident& res;
    ~ResGuard()
    {
      res.drop(run);
      if (res.fd >= 0) {
        close(res.fd);
      }
    }
  } res_guard{ run, res };
  {
    const char* e = getenv("GRP_RESIDENT");
    const bool gpu_ingest = run.vt.fastq_parse && run.vt.fastq_records && run.vt.fastq_pack && run.vt.fastq_free && !getenv("GRP_HOST_INGEST");
    if (!(e && !strcmp(e, "off")) && gpu_ingest && !opt.debug) {
      res.fd = open_plain_input(opt.input);
      res.on = res.fd >= 0;
      const char* cap = getenv("GRP_RESIDENT_MAX_GB");
      res.max_bytes = (uint64_t)((cap ? atof(cap) : 100.0) * (double)(1ull << 30));
    }
  }
  ec = fill_bit_vector(run, res);
  if (ec >= 0) {
    return ec;
  }
  if (input_error()) {
    return 1;
  }
  uint64_t pop = 0;
  if (run.vt.finalize(run.ctx, &pop) != GRP_OK) {
    return run.fail_engine("building the rank structure");
  }
  if (run.world > 1 && run.shm) {
    // every rank classifies on its own replica: they must hold the same filter (a merge that failed on one rank
    // would otherwise only show as diverging decisions much later)
    if (gr_ranks_same_u64(run.shm, run.world, pop) != 1) {
      std::cerr << "ERROR: the ranks' filters differ after the merge (rank " << run.rank << ": " << pop << " set bits)" << std::endl;
      return 1;
    }
  }

  std::cerr << "assigning tiles" << std::endl;
  s_time = now_s();
  gr_classifier_params cp{};
  cp.struct_size = sizeof(cp);
  cp.tile_length = (uint32_t)opt.tile_length;
  cp.block_size = (uint32_t)opt.block_size;
  cp.threshold = (uint32_t)opt.threshold;
  cp.unassigned_min = (uint32_t)opt.unassigned_min;
  cp.assigned_max = (uint32_t)opt.assigned_max;
  cp.kmer_size = (uint32_t)opt.kmer_size;
  cp.hash_num = (uint32_t)opt.hash_num;
  cp.target_bases = (uint64_t)(opt.ratio * opt.genome_size); // :1223
  cp.max_paths = opt.max_paths;
  cp.silver_path = opt.silver_path;
  cp.verbose = opt.verbose;
  cp.world = run.world;
  cp.rank = run.rank;
  cp.debug = opt.debug;
  if (opt.debug && run.world > 1) {
    std::cerr << "goldrush-path: --debug runs on one rank" << std::endl;
    return 1;
  }
  Classifier cls(cp, run.vt, run.ctx);
  SinkState sink;
  sink.run = &run;
  cls.set_callbacks(commit_sink, rollover_sink, nullptr, &sink);
  if (run.world > 1) {
    cls.set_allgather(gr_shm_allgather, shm.h);
  }
  if (opt.debug) {
    cls.set_debug(debug_sink);
  }

  if (res.on) {
    // the classification pass over the batches the fill pass left in HBM: no second parse
    sink.fd = res.fd;
    std::unordered_set<uint64_t> filtered_hash;
    for (const std::string& name : run.filter_out_reads) {
      filtered_hash.insert(fnv1a(name.data(), name.size()));
    }
    bool finished = false;
    std::vector<uint32_t> sb;
    std::string name;
    for (size_t bi = 0; bi < res.batches.size() && !finished; ++bi) {
      ResidentBatch& rb = res.batches[bi];
      const uint32_t n = (uint32_t)rb.lens.size();
      // reads named in filter_out_reads (a -f list, or a failing read of the same name) are not
      // classified (read_hashing.cpp:35-42): they become skipped records of the run behind them
      std::vector<uint8_t> drop(n, 0);
      bool any_drop = false;
      if (!filtered_hash.empty()) {
        for (uint32_t i = 0; i < n; ++i) {
          if (filtered_hash.count(rb.name_hash[i])) {
            name.resize(rb.loc[i].id_len);
            if (pread(res.fd, &name[0], name.size(), (off_t)rb.loc[i].off) == (ssize_t)name.size() && run.filter_out_reads.count(name)) {
              drop[i] = 1;
              any_drop = true;
            }
          }
        }
      }
      sink.resident = &rb;
      int rc = GRP_OK;
      if (!any_drop) {
        rc = cls.run(rb.reads, rb.lens.data(), 0, n, rb.skipped_before.data(), rb.skipped_after, finished);
      } else {
        // kept reads with the records skipped in front of each (dropped reads included), runs of
        // consecutive kept reads go to the classifier one after the other
        sb.assign(n, 0);
        std::vector<uint32_t> idx;
        uint32_t carry = 0;
        for (uint32_t i = 0; i < n; ++i) {
          if (drop[i]) {
            carry += rb.skipped_before[i] + 1;
          } else {
            sb[i] = rb.skipped_before[i] + carry;
            carry = 0;
            idx.push_back(i);
          }
        }
        const uint32_t trailing = carry + rb.skipped_after;
        if (idx.empty()) {
          rc = cls.run(rb.reads, rb.lens.data(), 0, 0, sb.data(), trailing, finished);
        }
        for (size_t a = 0; a < idx.size() && rc == GRP_OK && !finished;) {
          size_t e = a + 1;
          while (e < idx.size() && idx[e] == idx[e - 1] + 1) {
            ++e;
          }
          rc = cls.run(rb.reads, rb.lens.data(), idx[a], (uint32_t)(e - a), sb.data(), e == idx.size() ? trailing : 0u, finished);
          a = e;
        }
      }
      run.vt.reads_free(rb.reads);
      rb.reads = nullptr;
      if (rc != GRP_OK) {
        std::cerr << "goldrush-path: " << cls.error() << std::endl;
        return 1;
      }
    }
    sink.resident = nullptr;
    if (finished) {
      run.out.flush();
      return 0; // exit(0) inside silver_path_check (:173-176)
    }
  } else {
    auto src = open_source(run);
    Batch b;
    std::vector<uint32_t> sel, skipped_before, lens;
    std::vector<std::pair<uint32_t, uint8_t>> skipped_recs;
    bool finished = false;
    while (!finished && src->ok() && src->next(b)) {
      // read_hashing.cpp:35-42 / goldrush_path.cpp:907-932: a read is classified
      // iff it is long enough and not in filter_out_reads
      sel.clear();
      skipped_before.clear();
      skipped_recs.clear();
      uint32_t skipped = 0;
      for (size_t i = 0; i < b.rec.size(); ++i) {
        bool eligible = b.rec[i].seq_len >= opt.min_length;
        uint8_t why = eligible ? 0 : 1;
        if (eligible && !run.filter_out_reads.empty() && run.filter_out_reads.count(b.id_str(i))) {
          eligible = false;
          why = 2;
        }
        if (!eligible && opt.debug) {
          skipped_recs.emplace_back((uint32_t)i, why);
        }
        if (eligible) {
          sel.push_back((uint32_t)i);
          skipped_before.push_back(skipped);
          skipped = 0;
        } else {
          ++skipped;
        }
      }
      void* h = nullptr;
      if (src->upload(b, sel, lens, &h) != GRP_OK) {
        return run.fail_engine("uploading reads");
      }
      sink.batch = &b;
      sink.sel = &sel;
      sink.skipped = &skipped_recs;
      sink.skipped_before = &skipped_before;
      sink.skipped_done = 0;
      const int rc = cls.run(h, lens.data(), 0, (uint32_t)sel.size(), skipped_before.data(), skipped, finished);
      if (opt.debug && rc == GRP_OK && !finished) {
        debug_skipped(sink, sink.skipped_done, skipped_recs.size()); // the records behind the batch's last classified read
      }
      run.vt.reads_free(h);
      if (rc != GRP_OK) {
        std::cerr << "goldrush-path: " << cls.error() << std::endl;
        return 1;
      }
    }
    if (finished) {
      run.out.flush();
      return input_error() ? 1 : 0; // exit(0) inside silver_path_check (:173-176)
    }
  }
  if (input_error()) {
    return 1;
  }
  if (opt.silver_path && opt.max_paths > cls.curr_path()) {
    std::cerr << "WARNING: Expected " << std::to_string(opt.max_paths) << " silver paths, but only " << std::to_string(cls.curr_path()) << " generated.\n"
              << "Possible reasons include:\n"
              << "\t- Input reads sorted by chromosome/position\n"
              << "\t- Genome size set too large\n";
  }
  if (opt.verbose) {
    cls.log_path_stat();
  }
  std::cerr << "assigned" << std::endl;
  std::cerr << "in " << std::setprecision(4) << std::fixed << now_s() - s_time << "\n";
  return 0;
}
