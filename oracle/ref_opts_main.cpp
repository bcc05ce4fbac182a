// ORACLE — TEST INFRASTRUCTURE ONLY.
// Runs the reference's own process_options (goldrush_path/opt.cpp, compiled in place from
// /root/reference by oracle/Makefile, target `ref`) on argv and prints every option it
// left behind, one "name=value" per line.  The reference exits by itself on --help and on
// invalid input; the exit code is the reference's.  No reference code in this file.
#include "opt.hpp"

#include <iostream>

int
main(int argc, char** argv)
{
  process_options(argc, argv);
  std::cout << "assigned_max=" << opt::assigned_max << "\n"
            << "unassigned_min=" << opt::unassigned_min << "\n"
            << "tile_length=" << opt::tile_length << "\n"
            << "block_size=" << opt::block_size << "\n"
            << "hash_universe=" << opt::hash_universe << "\n"
            << "genome_size=" << opt::genome_size << "\n"
            << "kmer_size=" << opt::kmer_size << "\n"
            << "phred_min=" << opt::phred_min << "\n"
            << "phred_delta=" << opt::phred_delta << "\n"
            << "weight=" << opt::weight << "\n"
            << "min_length=" << opt::min_length << "\n"
            << "hash_num=" << opt::hash_num << "\n"
            << "occupancy=" << opt::occupancy << "\n"
            << "ratio=" << opt::ratio << "\n"
            << "jobs=" << opt::jobs << "\n"
            << "max_paths=" << opt::max_paths << "\n"
            << "threshold=" << opt::threshold << "\n"
            << "prefix_file=" << opt::prefix_file << "\n"
            << "input=" << opt::input << "\n"
            << "seed_preset=" << opt::seed_preset << "\n"
            << "filter_file=" << opt::filter_file << "\n"
            << "help=" << opt::help << "\n"
            << "ntcard=" << opt::ntcard << "\n"
            << "silver_path=" << opt::silver_path << "\n"
            << "verbose=" << opt::verbose << "\n"
            << "debug=" << opt::debug << std::endl;
  return 0;
}
