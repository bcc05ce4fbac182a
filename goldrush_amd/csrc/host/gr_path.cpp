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
#include <memory>
#include <unordered_set>
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

struct PackedBatch
{
  std::vector<uint32_t> packed;
  std::vector<uint64_t> word_off;
  std::vector<uint32_t> len;
  std::vector<uint32_t> src; // record index in the RecordBatch
};

// pack the selected records (2 bits / base); returns false if one of them is not ACGT
void
pack_selected(const RecordBatch& rb, const std::vector<uint32_t>& sel, PackedBatch& pb)
{
  const size_t n = sel.size();
  pb.src = sel;
  pb.len.resize(n);
  pb.word_off.resize(n + 1);
  uint64_t w = 0;
  for (size_t i = 0; i < n; ++i) {
    pb.word_off[i] = w;
    pb.len[i] = (uint32_t)rb.rec[sel[i]].seq_len;
    w += (rb.rec[sel[i]].seq_len + 15) / 16;
  }
  pb.word_off[n] = w;
  pb.packed.resize(w ? w : 1);
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 16)
#endif
  for (size_t i = 0; i < n; ++i) {
    pack_2bit(rb.seq(sel[i]), rb.rec[sel[i]].seq_len, pb.packed.data() + pb.word_off[i]);
  }
}

struct PathRun
{
  Opts opt;
  grp_engine_vt vt{};
  void* ctx = nullptr;
  std::vector<std::string> seeds;
  std::unordered_set<std::string> filter_out_reads;
  std::ofstream out;
  // classification sink
  const RecordBatch* cur_batch = nullptr;
  const PackedBatch* cur_packed = nullptr;

  int fail_engine(const char* what)
  {
    std::cerr << "goldrush-path: " << what << ": " << (vt.last_error ? vt.last_error(ctx) : "engine error") << std::endl;
    return 1;
  }
};

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
  FastqStream fq(opt.input);
  RecordBatch rb;
  bool done = false;
  while (!done && fq.ok() && fq.next_batch(rb, BATCH_RECORDS, BATCH_BASES)) {
    // phred of the eligible records of this batch, in parallel, then consumed in order
    std::vector<uint32_t> avg(rb.rec.size(), 0);
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 8)
#endif
    for (size_t i = 0; i < rb.rec.size(); ++i) {
      if (rb.rec[i].seq_len >= opt.min_length) {
        uint32_t a, d;
        calc_phred_average(rb.qual(i), rb.rec[i].qual_len, a, d);
        avg[i] = a;
      }
    }
    for (size_t i = 0; i < rb.rec.size(); ++i) {
      if (rb.rec[i].seq_len < opt.min_length) {
        continue;
      }
      if (taken >= MEDIAN_SAMPLES_NEEDED) {
        if (++over >= opt.jobs) {
          done = true;
          break;
        }
        continue;
      }
      scores[taken++] = avg[i];
    }
  }
  const size_t n = taken + over;
  std::sort(scores.begin(), scores.end(), std::greater<uint32_t>());
  opt.phred_min = std::max(MINIMUM_PHRED_THRESHOLD, scores[n / 2]);
  if (opt.debug) {
    std::cerr << "Number of reads used to calculate median: " << n << std::endl;
  }
  if (opt.verbose) {
    std::cerr << "Minimum phred score calculated with median: " << opt.phred_min << std::endl;
  }
  return -1;
}

// goldrush_path.cpp:235-339
int
fill_bit_vector(PathRun& run)
{
  const Opts& opt = run.opt;
  std::cerr << "inserting bit vector" << std::endl;
  const double s_time = now_s();
  FastqStream fq(opt.input);
  if (!fq.ok() || !fq.is_fastq()) {
    std::cerr << "Gold Path requires fastq format" << std::endl;
    return 1;
  }
  size_t num_reads = 0, num_passed_reads = 0, by_phred = 0, by_delta = 0, by_length = 0, by_bases = 0;
  RecordBatch rb;
  PackedBatch pb;
  void* prev = nullptr;
  std::vector<uint8_t> verdict; // 0 pass, 1 short, 2 phred/delta, 3 invalid bases
  std::vector<uint8_t> why;     // bit0 phred, bit1 delta
  while (fq.next_batch(rb, BATCH_RECORDS, BATCH_BASES)) {
    const size_t n = rb.rec.size();
    verdict.assign(n, 0);
    why.assign(n, 0);
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 8)
#endif
    for (size_t i = 0; i < n; ++i) {
      const RecordRef& r = rb.rec[i];
      if (r.seq_len < opt.min_length) {
        verdict[i] = 1;
        continue;
      }
      uint32_t avg, delta;
      calc_phred_average(rb.qual(i), r.qual_len, avg, delta);
      if (avg < opt.phred_min || delta >= opt.phred_delta) {
        verdict[i] = 2;
        why[i] = (uint8_t)((avg < opt.phred_min ? 1 : 0) | (delta >= opt.phred_delta ? 2 : 0));
        continue;
      }
      // seq is already upper-cased; find_first_not_of("ACGTacgt")
      const char* s = rb.seq(i);
      bool ok = true;
      for (size_t j = 0; j < r.seq_len; ++j) {
        const char c = s[j];
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') {
          ok = false;
          break;
        }
      }
      if (!ok) {
        verdict[i] = 3;
      }
    }
    std::vector<uint32_t> sel;
    for (size_t i = 0; i < n; ++i) {
      ++num_reads;
      switch (verdict[i]) {
        case 0:
          ++num_passed_reads;
          sel.push_back((uint32_t)i);
          break;
        case 1:
          ++by_length;
          break;
        case 2:
          if (opt.verbose) {
            by_phred += (why[i] & 1) ? 1 : 0;
            by_delta += (why[i] & 2) ? 1 : 0;
          }
          run.filter_out_reads.insert(rb.id_str(i));
          break;
        default:
          ++by_bases;
          run.filter_out_reads.insert(rb.id_str(i));
          break;
      }
    }
    if (!sel.empty()) {
      pack_selected(rb, sel, pb);
      if (prev) {
        run.vt.reads_free(prev); // waits for the previous batch's kernel
        prev = nullptr;
      }
      void* h = nullptr;
      if (run.vt.reads_upload(run.ctx, pb.packed.data(), pb.word_off.data(), pb.len.data(), (uint32_t)sel.size(), &h) != GRP_OK) {
        return run.fail_engine("uploading reads");
      }
      // multiLensfrHashIterator itr(record.seq, seeds); miBFCS.insertBV(itr)  (:304-305)
      if (run.vt.bv_insert(run.ctx, h, 0, (uint32_t)sel.size()) != GRP_OK) {
        return run.fail_engine("bit-vector insert");
      }
      prev = h;
    }
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
  if (run.vt.sync(run.ctx) != GRP_OK) {
    return run.fail_engine("bit-vector insert");
  }
  std::cerr << "finished inserting bit vector" << std::endl;
  std::cerr << "in " << std::setprecision(4) << std::fixed << now_s() - s_time << "\n";
  return -1;
}

// writes one committed read (goldrush_path.cpp:996-1002, 1055-1070)
double
commit_sink(void* user, const gr_commit* c)
{
  PathRun& run = *static_cast<PathRun*>(user);
  if (c->dec.kind != DEC_INSERT_WHOLE && c->dec.kind != DEC_INSERT_TRIMMED) {
    return 0.0;
  }
  const RecordBatch& rb = *run.cur_batch;
  const size_t i = run.cur_packed->src[c->read];
  const RecordRef& r = rb.rec[i];
  const char first = run.opt.silver_path ? '@' : '>';
  const char* seq = rb.seq(i);
  const char* qual = rb.qual(i);
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
  o.write(rb.id(i), (std::streamsize)r.id_len);
  o << suffix;
  o.write(seq + off, (std::streamsize)n_seq);
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
  PathRun& run = *static_cast<PathRun*>(user);
  run.out.close();
  run.out.open(run.opt.prefix_file + "_" + std::to_string(new_path) + ".fq");
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
  int ec = process_options(opt, argc, argv);
  if (ec >= 0) {
    return ec;
  }
#if defined(_OPENMP)
  // the reference uses -j for its OpenMP regions; here it only drives the host
  // side (parsing, Phred, packing, decisions) and is capped at the machine size
  omp_set_num_threads((int)std::max<size_t>(1, std::min<size_t>(opt.jobs, (size_t)effective_cpus())));
#endif
  run.seeds = make_seed_pattern(opt.seed_preset, (unsigned)opt.kmer_size, (unsigned)opt.weight, (unsigned)opt.hash_num, true);
  if (opt.hash_universe == 0) {
    if (opt.ntcard) {
      std::cerr << "goldrush-path: --ntcard is not available in this build (it is never passed by bin/goldrush)" << std::endl;
      return 1;
    }
    opt.hash_universe = hash_universe(opt.weight, opt.genome_size, opt.hash_num);
  }
  const std::string what = opt.silver_path ? std::to_string(opt.max_paths) + " silver path(s)" : std::string("the golden path");
  ec = calc_min_phred_threshold(run);
  if (ec >= 0) {
    return ec;
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
  run.out.open(opt.silver_path ? opt.prefix_file + "_1.fq" : opt.prefix_file + ".fa");
  double s_time = now_s();
  std::cerr << "allocating bit vector" << std::endl;
  const uint64_t filter_size = calc_optimal_size(opt.hash_universe, 1, opt.occupancy);
  std::cerr << "m_filterSize: " << filter_size << std::endl;
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
    gp.device = -1;
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
  std::cerr << "finished allocating bit vector" << std::endl;
  std::cerr << "in " << std::setprecision(4) << std::fixed << now_s() - s_time << "\n";
  std::cerr << "opening: " << opt.input << std::endl;

  ec = fill_bit_vector(run);
  if (ec >= 0) {
    return ec;
  }
  uint64_t pop = 0;
  if (run.vt.finalize(run.ctx, &pop) != GRP_OK) {
    return run.fail_engine("building the rank structure");
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
  cp.world = 1;
  cp.rank = 0;
  Classifier cls(cp, run.vt, run.ctx);
  cls.set_callbacks(commit_sink, rollover_sink, nullptr, &run);

  {
    FastqStream fq(opt.input);
    RecordBatch rb;
    PackedBatch pb;
    bool finished = false;
    while (!finished && fq.ok() && fq.next_batch(rb, BATCH_RECORDS, BATCH_BASES)) {
      // read_hashing.cpp:35-42 / goldrush_path.cpp:907-932: a read is classified
      // iff it is long enough and not in filter_out_reads
      std::vector<uint32_t> sel, skipped_before;
      uint32_t skipped = 0;
      for (size_t i = 0; i < rb.rec.size(); ++i) {
        bool eligible = rb.rec[i].seq_len >= opt.min_length;
        if (eligible && !run.filter_out_reads.empty() && run.filter_out_reads.count(rb.id_str(i))) {
          eligible = false;
        }
        if (eligible) {
          sel.push_back((uint32_t)i);
          skipped_before.push_back(skipped);
          skipped = 0;
        } else {
          ++skipped;
        }
      }
      pack_selected(rb, sel, pb);
      void* h = nullptr;
      if (run.vt.reads_upload(run.ctx, pb.packed.data(), pb.word_off.data(), pb.len.data(), (uint32_t)sel.size(), &h) != GRP_OK) {
        return run.fail_engine("uploading reads");
      }
      run.cur_batch = &rb;
      run.cur_packed = &pb;
      const int rc = cls.run(h, pb.len.data(), 0, (uint32_t)sel.size(), skipped_before.data(), skipped, finished);
      run.vt.reads_free(h);
      if (rc != GRP_OK) {
        std::cerr << "goldrush-path: " << cls.error() << std::endl;
        return 1;
      }
    }
    if (finished) {
      run.out.flush();
      return 0; // exit(0) inside silver_path_check (:173-176)
    }
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
