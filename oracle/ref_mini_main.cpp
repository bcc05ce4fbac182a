// ORACLE — TEST INFRASTRUCTURE ONLY.
// "Mini reference": everything GoldRush-Path does with the HASHES of its reads, compiled from the
// reference's own text (cut out of /root/reference at build time by oracle/extract_ref_funcs.py into
// the git-ignored oracle/_ref/extract/*.inc and included below, unchanged):
//   goldrush_path.cpp      calc_num_assigned_tiles (whole: atRank / getData per frame, the saturation
//                          bit, hits / misses, the per-tile vote, the smoothing passes), the BODY of
//                          process_read (decision, ID allocation per block, the insert calls, the output
//                          record and its substrings, the counters), silver_path_check, log_path_stat,
//                          log_tile_states, find_longest_stretch, eval_flanks, sort_by_sec, log_info_struct
//   MIBloomFilter.hpp      s_mask / s_antiMask, atRank, getRankPos, getHashNum, size, getData, setData,
//                          reset_ID_vector — as members of the capture class of the same name below
//   MIBFConstructSupport.hpp  insertMIBF (the hash_vec overload, whole: the vec_size indexing, the unique
//                          ranks, count / reservoir test / setData), reset_counts
//   opt.cpp, calc_phred_average.cpp   compiled whole next to this file
// What is OURS here, and therefore still a restatement (the libraries are neither vendored nor installed):
//   * the hashes themselves — the scenario file holds them (the oracle's ntHash restatement writes it);
//   * the bit vector and its rank (sdsl::bit_vector_il / rank_support_il): MiniBits / MiniRank,
//     positional semantics only — bit i, and the number of ones in [0, i);
//   * a set of uint64 (google::dense_hash_set): MiniSet — uniqueness only, each rank is then processed
//     independently of the others (MIBFConstructSupport.hpp:274-282);
//   * the record type (btllib::SeqReader::Record): three strings;
//   * main: the scenario reader, the output file names of goldrush_path.cpp:1174-1179, the loop.
// The oracle's whole path (its own hashing, then orc_path.c) must write the same files and end in the
// same counters, IDs and counts on the same reads: tests/test_reference_mini.py,
// tests/golden/reference_mini.json (generator: tests/golden/make_reference_fixtures.py).
// This file contains no reference code.
#include "calc_phred_average.hpp"
#include "opt.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <sys/types.h>
#include <tuple>
#include <unordered_set>
#include <utility>
#include <vector>

using namespace std; // MIBloomFilter.hpp:32; goldrush_path.cpp names pair / vector unqualified

// ---- ours: what stands where sdsl / sparsehash / btllib types stand in the reference ----------------
struct MiniBits
{
  std::vector<uint64_t> words;
  size_t n = 0;
  size_t size() const { return n; }
  bool operator[](size_t i) const { return (words[i >> 6] >> (i & 63)) & 1ull; }
};

struct MiniRank // ones in [0, i)
{
  const MiniBits* bits = nullptr;
  std::vector<uint64_t> before; // ones in front of word w
  void build(const MiniBits& b)
  {
    bits = &b;
    before.assign(b.words.size() + 1, 0);
    for (size_t w = 0; w < b.words.size(); ++w) {
      before[w + 1] = before[w] + (uint64_t)__builtin_popcountll(b.words[w]);
    }
  }
  uint64_t operator()(size_t i) const
  {
    const uint64_t mask = (i & 63) ? ((1ull << (i & 63)) - 1ull) : 0ull;
    return before[i >> 6] + (uint64_t)__builtin_popcountll(bits->words[i >> 6] & mask);
  }
};

struct MiniSet
{
  std::unordered_set<uint64_t> s;
  void set_empty_key(uint64_t) {}
  void insert(uint64_t v) { s.insert(v); }
  std::unordered_set<uint64_t>::const_iterator begin() const { return s.begin(); }
  std::unordered_set<uint64_t>::const_iterator end() const { return s.end(); }
};

struct MiniRecord
{
  std::string id, seq, qual;
};

class multiLensfrHashIterator; // a template argument's name only

// ---- capture classes named like the reference's: every member function below is the reference's text ----
template<typename T>
class MIBloomFilter
{
public:
#include "_ref/extract/s_mask_decl.inc"
#include "_ref/extract/s_antimask_decl.inc"
  MIBloomFilter(MiniBits& bv, MiniRank& rs, unsigned hash_num, size_t pop)
    : m_bv(bv)
    , m_data(pop, 0)
    , m_rankSupport(rs)
    , m_hashNum(hash_num)
  {}
#include "_ref/extract/at_rank_vec.inc"
#include "_ref/extract/get_rank_pos_hash.inc"
#include "_ref/extract/get_hash_num.inc"
#include "_ref/extract/size_fn.inc"
#include "_ref/extract/get_data_vec.inc"
#include "_ref/extract/set_data.inc"
#include "_ref/extract/reset_id_vector.inc"
  MiniBits& m_bv;
  std::vector<T> m_data;
  MiniRank& m_rankSupport;
  unsigned m_hashNum;
};

template<typename T, class H>
class MIBFConstructSupport
{
public:
  typedef MiniSet hashSet;
  explicit MIBFConstructSupport(size_t pop)
    : m_counts(pop, 0)
  {}
#include "_ref/extract/insert_mibf_whole.inc"
#include "_ref/extract/reset_counts.inc"
  vector<T> m_counts;
};

#include "_ref/extract/log_info_struct.inc"
#include "_ref/extract/log_tile_states.inc"
#include "_ref/extract/log_path_stat.inc"
#include "_ref/extract/silver_path_check.inc"
#include "_ref/extract/sort_by_sec.inc"
#include "_ref/extract/find_longest_stretch.inc"
#include "_ref/extract/calc_num_assigned_tiles.inc"
#include "_ref/extract/eval_flanks.inc"

// the body of process_read (goldrush_path.cpp:906-1094) under its own parameter names
static void
ref_process_read(const MiniRecord& record,
                 const std::vector<std::vector<uint64_t>>& hashed_values,
                 std::vector<std::ofstream>& golden_path_vec,
                 std::vector<std::unique_ptr<MIBloomFilter<uint32_t>>>& mibf_vec,
                 MIBFConstructSupport<uint32_t, multiLensfrHashIterator>& miBFCS,
                 uint64_t& inserted_bases,
                 uint64_t& target_bases,
                 uint64_t& curr_path,
                 uint32_t& id,
                 uint32_t& ids_inserted,
                 const size_t min_seq_len,
                 const std::unordered_set<std::string>& filter_out_reads,
                 log_info_struct& log_info)
{
#include "_ref/extract/process_read_body.inc"

// ---- ours: scenario in, end state out ----------------------------------------------------------------
namespace {

struct State
{
  std::string prefix;
  uint64_t inserted_bases = 0, target_bases = 0, curr_path = 1;
  uint32_t id = 1, ids_inserted = 0;
  log_info_struct log_info;
  std::vector<std::unique_ptr<MIBloomFilter<uint32_t>>>* mibf_vec = nullptr;
  MIBFConstructSupport<uint32_t, multiLensfrHashIterator>* cs = nullptr;
  std::vector<std::ofstream>* out = nullptr;
  bool exited_in_check = true; // cleared when main reaches its end
  bool dumped = false;
} g;

uint64_t
rd64(FILE* f)
{
  uint64_t v = 0;
  if (fread(&v, 8, 1, f) != 1) {
    fprintf(stderr, "ref_mini: scenario truncated\n");
    _Exit(3);
  }
  return v;
}

std::string
rdstr(FILE* f)
{
  std::string s(rd64(f), '\0');
  if (!s.empty() && fread(&s[0], 1, s.size(), f) != s.size()) {
    fprintf(stderr, "ref_mini: scenario truncated\n");
    _Exit(3);
  }
  return s;
}

// what the run left: counters as JSON, the ID and count arrays as raw uint32 — also when the reference's
// exit(0) inside silver_path_check ends the process
void
dump_state()
{
  if (g.dumped) {
    return;
  }
  g.dumped = true;
  for (auto& o : *g.out) {
    o.flush();
  }
  FILE* f = fopen((g.prefix + ".mini.json").c_str(), "w");
  fprintf(f,
          "{\"ids_inserted\": %u, \"inserted_bases\": %llu, \"curr_path\": %llu, \"id\": %u, \"exit_in_silver_path_check\": %s, \"valid_reads\": %llu, \"total_tiles\": %llu, \"assigned_tiles\": %llu, "
          "\"unassigned_tiles\": %llu, \"queries\": %llu, \"hits\": %llu, \"misses\": %llu, \"num_reads_in_path\": %llu, \"phred_sum_in_path_bits\": %llu}\n",
          g.ids_inserted,
          (unsigned long long)g.inserted_bases,
          (unsigned long long)g.curr_path,
          g.id,
          g.exited_in_check ? "true" : "false",
          (unsigned long long)g.log_info.valid_reads,
          (unsigned long long)g.log_info.total_tiles_per_path,
          (unsigned long long)g.log_info.total_assigned_tiles_per_path,
          (unsigned long long)g.log_info.total_unassigned_tiles_per_path,
          (unsigned long long)g.log_info.total_queries_per_path,
          (unsigned long long)g.log_info.total_hits_per_path,
          (unsigned long long)g.log_info.total_misses_per_path,
          (unsigned long long)g.log_info.num_reads_in_path,
          [] {
            unsigned long long b;
            memcpy(&b, &g.log_info.phred_sum_in_path, 8);
            return b;
          }());
  fclose(f);
  const auto& ids = (*g.mibf_vec)[0]->m_data;
  f = fopen((g.prefix + ".mini.ids").c_str(), "wb");
  fwrite(ids.data(), 4, ids.size(), f);
  fclose(f);
  f = fopen((g.prefix + ".mini.counts").c_str(), "wb");
  fwrite(g.cs->m_counts.data(), 4, g.cs->m_counts.size(), f);
  fclose(f);
}

} // namespace

int
main(int argc, char** argv)
{
  if (argc != 2) {
    fprintf(stderr, "usage: ref_mini <scenario>\n");
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  char magic[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "GRMINI1\n", 8) != 0) {
    fprintf(stderr, "ref_mini: not a scenario file\n");
    return 2;
  }
  opt::tile_length = rd64(f);
  opt::block_size = rd64(f);
  opt::threshold = rd64(f);
  opt::unassigned_min = rd64(f);
  opt::assigned_max = rd64(f);
  opt::silver_path = (int)rd64(f);
  opt::verbose = (int)rd64(f);
  g.target_bases = rd64(f);
  opt::max_paths = rd64(f);
  opt::min_length = rd64(f);
  opt::hash_num = rd64(f);
  const uint64_t m = rd64(f);
  const uint64_t n_reads = rd64(f);
  const uint64_t n_words = rd64(f);
  opt::debug = 0;
  g.prefix = rdstr(f);
  opt::prefix_file = g.prefix;

  MiniBits bits;
  bits.n = m;
  bits.words.resize(n_words);
  if (n_words && fread(bits.words.data(), 8, n_words, f) != n_words) {
    fprintf(stderr, "ref_mini: scenario truncated\n");
    return 3;
  }
  MiniRank rank;
  rank.build(bits);
  const size_t pop = rank.before.back();

  std::vector<std::unique_ptr<MIBloomFilter<uint32_t>>> mibf_vec;
  mibf_vec.emplace_back(new MIBloomFilter<uint32_t>(bits, rank, (unsigned)opt::hash_num, pop));
  MIBFConstructSupport<uint32_t, multiLensfrHashIterator> miBFCS(pop);
  // the output files of goldrush_path.cpp:1174-1179
  std::vector<std::ofstream> golden_path_vec;
  golden_path_vec.emplace_back(std::ofstream(opt::silver_path ? g.prefix + "_1.fq" : g.prefix + ".fa"));
  g.mibf_vec = &mibf_vec;
  g.cs = &miBFCS;
  g.out = &golden_path_vec;
  atexit(dump_state);

  std::unordered_set<std::string> filter_out_reads;
  std::vector<MiniRecord> records(n_reads);
  std::vector<std::vector<std::vector<uint64_t>>> hashes(n_reads);
  for (uint64_t r = 0; r < n_reads; ++r) {
    records[r].id = rdstr(f);
    records[r].seq = rdstr(f);
    records[r].qual = rdstr(f);
    if (rd64(f)) {
      filter_out_reads.insert(records[r].id);
    }
    hashes[r].resize(rd64(f));
    for (auto& tile : hashes[r]) {
      tile.resize(rd64(f));
      if (!tile.empty() && fread(tile.data(), 8, tile.size(), f) != tile.size()) {
        fprintf(stderr, "ref_mini: scenario truncated\n");
        return 3;
      }
    }
  }
  fclose(f);

  // the loop of goldrush_path.cpp:1229-1256 (records and their hashes in file order)
  for (uint64_t r = 0; r < n_reads; ++r) {
    ref_process_read(records[r], hashes[r], golden_path_vec, mibf_vec, miBFCS, g.inserted_bases, g.target_bases, g.curr_path, g.id, g.ids_inserted, opt::min_length, filter_out_reads, g.log_info);
  }
  g.exited_in_check = false;
  if (opt::verbose) { // main's last words (goldrush_path.cpp:1266-1270): the reference's own log_path_stat
    log_path_stat(g.curr_path, g.log_info, g.inserted_bases);
  }
  dump_state(); // (the objects it reads are gone by the time atexit handlers run behind main)
  return 0;
}
