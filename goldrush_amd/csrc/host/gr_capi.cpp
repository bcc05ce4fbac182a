// extern "C" surface of libgrpath_host.so (include/grpath_host.h).
#include "../../../include/grpath_host.h"
#include "gr_classifier.hpp"
#include "gr_params.hpp"
#include "gr_opts.hpp"
#include <algorithm>
#include <sstream>
#include "gr_tiles.hpp"
#include "gr_fastq.hpp"

#include <cstring>
#include <new>

extern "C" {

int
gr_make_seed_pattern(const char* preset, unsigned k, unsigned weight, unsigned h, char* out, size_t stride, int log_to_stderr)
{
  if (!out || k / 2 == 0) {
    return -1;
  }
  auto seeds = gr::make_seed_pattern(preset ? preset : "", k, weight, h, log_to_stderr != 0);
  for (unsigned i = 0; i < h; ++i) {
    if (seeds[i].size() + 1 > stride) {
      return -1;
    }
    std::memcpy(out + (size_t)i * stride, seeds[i].c_str(), seeds[i].size() + 1);
  }
  return 0;
}

uint64_t
gr_hash_universe(uint64_t weight, uint64_t genome_size, uint64_t hash_num)
{
  return gr::hash_universe(weight, genome_size, hash_num);
}

uint64_t
gr_calc_optimal_size(uint64_t entries, unsigned hash_num, double occupancy)
{
  return gr::calc_optimal_size(entries, hash_num, occupancy);
}

void
gr_calc_phred_average(const char* qual, size_t n, uint32_t* avg, uint32_t* delta)
{
  gr::calc_phred_average(qual, n, *avg, *delta);
}

double
gr_sum_phred(const char* qual, size_t n)
{
  return gr::sum_phred(qual, n);
}

int
gr_process_options_dump(int argc, char** argv, char* out, size_t cap)
{
  gr::Opts o;
  const int rc = gr::process_options(o, argc, argv);
  std::ostringstream ss;
  ss << "assigned_max=" << o.assigned_max << "\n"
     << "unassigned_min=" << o.unassigned_min << "\n"
     << "tile_length=" << o.tile_length << "\n"
     << "block_size=" << o.block_size << "\n"
     << "hash_universe=" << o.hash_universe << "\n"
     << "genome_size=" << o.genome_size << "\n"
     << "kmer_size=" << o.kmer_size << "\n"
     << "phred_min=" << o.phred_min << "\n"
     << "phred_delta=" << o.phred_delta << "\n"
     << "weight=" << o.weight << "\n"
     << "min_length=" << o.min_length << "\n"
     << "hash_num=" << o.hash_num << "\n"
     << "occupancy=" << o.occupancy << "\n"
     << "ratio=" << o.ratio << "\n"
     << "jobs=" << o.jobs << "\n"
     << "max_paths=" << o.max_paths << "\n"
     << "threshold=" << o.threshold << "\n"
     << "prefix_file=" << o.prefix_file << "\n"
     << "input=" << o.input << "\n"
     << "seed_preset=" << o.seed_preset << "\n"
     << "filter_file=" << o.filter_file << "\n"
     << "help=" << o.help << "\n"
     << "ntcard=" << o.ntcard << "\n"
     << "silver_path=" << o.silver_path << "\n"
     << "verbose=" << o.verbose << "\n"
     << "debug=" << o.debug << "\n";
  const std::string t = ss.str();
  if (out && cap) {
    const size_t n = std::min(cap - 1, t.size());
    std::memcpy(out, t.data(), n);
    out[n] = '\0';
  }
  return rc;
}

unsigned
gr_ntcard_sbits(uint64_t input_bytes)
{
  return gr::ntcard_sbits(input_bytes);
}

uint64_t
gr_ntcard_f0(uint64_t zero0, uint64_t zero1, unsigned sbits)
{
  return gr::ntcard_f0(zero0, zero1, sbits);
}

size_t
gr_ntcard_split(const char* seq, size_t n, unsigned k, unsigned h, uint64_t* run_off, uint64_t* run_len, uint32_t* extra, size_t cap)
{
  std::vector<std::pair<size_t, size_t>> runs;
  std::vector<uint32_t> ex;
  gr::ntcard_split(seq, n, k, h, runs, ex);
  for (size_t r = 0; r < runs.size() && r < cap; ++r) {
    run_off[r] = runs[r].first;
    run_len[r] = runs[r].second;
    for (unsigned s = 0; s < h; ++s) {
      extra[r * h + s] = ex[r * h + s];
    }
  }
  return runs.size();
}

unsigned
gr_effective_cpus(void)
{
  return gr::effective_cpus();
}

int
gr_pack_2bit(const char* seq, size_t n, uint32_t* out_words)
{
  return gr::pack_2bit(seq, n, out_words) ? 0 : -1;
}

void
gr_decide_read(size_t threshold, size_t unassigned_min, size_t assigned_max, size_t num_tiles, const grp_tile_summary* tiles, const grp_id_count* lists, gr_read_decision* out)
{
  gr::TileWorkspace ws;
  gr::ReadDecision d;
  gr::decide_read(gr::DecideParams{ threshold, unassigned_min, assigned_max }, num_tiles, tiles, lists, ws, d);
  std::memcpy(out, &d, sizeof(d));
}

size_t
gr_smooth_tiles(size_t num_tiles, const grp_tile_summary* tiles, const grp_id_count* lists, size_t threshold, uint32_t* ids_out, uint8_t* bools_out)
{
  gr::TileWorkspace ws;
  size_t n = gr::smooth_tiles(num_tiles, tiles, lists, threshold, ws);
  for (size_t i = 0; i < num_tiles; ++i) {
    ids_out[i] = ws.ids[i];
    bools_out[i] = ws.asg[i];
  }
  return n;
}

void
gr_find_longest_stretch(const uint8_t* bools, size_t num_tiles, long* start, long* end)
{
  std::vector<uint8_t> b(bools, bools + num_tiles);
  gr::find_longest_stretch(b, num_tiles, *start, *end);
}

int
gr_eval_flanks(long ls, long le, const uint32_t* ids, size_t num_tiles, size_t* trim_start, size_t* trim_end)
{
  return gr::eval_flanks(ls, le, ids, num_tiles, *trim_start, *trim_end) ? 1 : 0;
}

struct gr_classifier
{
  gr::Classifier impl;
  gr_classifier(const gr_classifier_params& p, const grp_engine_vt& vt, void* ctx)
    : impl(p, vt, ctx)
  {}
};

int
gr_classifier_create(const gr_classifier_params* p, const grp_engine_vt* vt, void* engine_ctx, gr_classifier** out)
{
  if (!p || !vt || !out || p->struct_size != sizeof(gr_classifier_params) || p->tile_length == 0 || p->block_size == 0) {
    return GRP_ERR_INVALID;
  }
  *out = new (std::nothrow) gr_classifier(*p, *vt, engine_ctx);
  return *out ? GRP_OK : GRP_ERR_NOMEM;
}

void
gr_classifier_destroy(gr_classifier* c)
{
  delete c;
}

void
gr_classifier_keep_commits(gr_classifier* c, uint32_t first0, uint32_t count0, uint32_t first1, uint32_t count1)
{
  if (c) {
    c->impl.keep_commits(first0, count0, first1, count1);
  }
}

size_t
gr_classifier_kept_commits(const gr_classifier* c, gr_commit* out, size_t cap)
{
  if (!c) {
    return 0;
  }
  const auto& k = c->impl.kept_commits();
  for (size_t i = 0; i < k.size() && i < cap; ++i) {
    out[i] = k[i];
  }
  return k.size();
}

void
gr_classifier_set_debug(gr_classifier* c, gr_debug_fn fn)
{
  if (c) {
    c->impl.set_debug(fn);
  }
}

void
gr_classifier_set_allgather(gr_classifier* c, gr_allgather_fn allgather, void* allgather_user)
{
  if (c) {
    c->impl.set_allgather(allgather, allgather_user);
  }
}

void
gr_classifier_set_callbacks(gr_classifier* c, gr_commit_fn commit, gr_rollover_fn rollover, gr_allgather_fn allgather, void* user)
{
  c->impl.set_callbacks(commit, rollover, allgather, user);
}

int
gr_classifier_run(gr_classifier* c, void* reads, const uint32_t* lens, uint32_t n_reads, const uint32_t* skipped_before, uint32_t skipped_after, int* finished)
{
  return gr_classifier_run_range(c, reads, lens, 0, n_reads, skipped_before, skipped_after, finished);
}

int
gr_classifier_run_range(gr_classifier* c, void* reads, const uint32_t* lens, uint32_t first, uint32_t count, const uint32_t* skipped_before, uint32_t skipped_after, int* finished)
{
  bool fin = false;
  int rc = c->impl.run(reads, lens, first, count, skipped_before, skipped_after, fin);
  if (finished) {
    *finished = fin ? 1 : 0;
  }
  return rc;
}

const char*
gr_classifier_error(const gr_classifier* c)
{
  return c->impl.error().c_str();
}

void
gr_classifier_get_state(const gr_classifier* c, gr_classifier_state* out)
{
  c->impl.get_state(*out);
}

uint64_t
gr_input_read(const char* path, uint64_t request_bytes, char* dst, uint64_t cap)
{
  gr::InputFile in(path ? path : "");
  if (!in.ok()) {
    return UINT64_MAX;
  }
  uint64_t got = 0;
  while (got < cap) {
    const size_t k = in.read(dst + got, (size_t)std::min<uint64_t>(std::max<uint64_t>(request_bytes, 1), cap - got));
    if (k == 0) {
      break;
    }
    got += k;
  }
  return got;
}

} // extern "C"
