// Command-line surface of goldrush-path (goldrush_path/opt.cpp, opt.hpp): same
// flag letters, defaults, validation messages and exit codes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace gr {

struct Opts
{
  size_t assigned_max = 1;    // -a
  size_t unassigned_min = 5;  // -u
  size_t tile_length = 1000;  // -t   (NOT threads)
  uint64_t hash_universe = 0; // -H
  uint64_t genome_size = 0;   // -g   (strtod: "3e9" works)
  size_t kmer_size = 0;       // -k
  size_t weight = 0;          // -w
  size_t min_length = 20000;  // -m
  size_t hash_num = 3;        // -h   (NOT help)
  double occupancy = 0.1;     // -o
  double ratio = 0.9;         // -r
  size_t jobs = 48;           // -j
  size_t block_size = 10;     // -b
  size_t max_paths = 1;       // -M
  size_t threshold = 10;      // -x
  uint32_t phred_min = 0;     // -P
  uint32_t phred_delta = 5;   // -d
  std::string prefix_file = "goldrush_out"; // -p
  std::string input;          // -i
  std::string seed_preset;    // -s
  std::string filter_file;    // -f
  int help = 0, ntcard = 0, silver_path = 0, verbose = 0, debug = 0;
};

void print_usage(const std::string& progname);
// returns -1 to continue, otherwise the exit code the reference exits with
int process_options(Opts& o, int argc, char** argv);

} // namespace gr
