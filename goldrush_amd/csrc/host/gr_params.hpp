// Small pure host functions of GoldRush-Path: seed design, filter sizing,
// Phred statistics, 2-bit packing.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace gr {

// make_seed_pattern (spaced_seeds.cpp:7-69)
std::vector<std::string> make_seed_pattern(const std::string& preset, unsigned k, unsigned weight, unsigned h, bool log);
// goldrush_path.cpp:1113-1121
uint64_t hash_universe(uint64_t weight, uint64_t genome_size, uint64_t hash_num);
// MIBloomFilter::calcOptimalSize (MIBloomFilter.hpp:94-101)
uint64_t calc_optimal_size(uint64_t entries, unsigned hash_num, double occupancy);
// calc_phred_average.cpp:8-43 / :45-58
void calc_phred_average(const char* qual, size_t n, uint32_t& avg, uint32_t& delta);
double sum_phred(const char* qual, size_t n);
// the tail of calc_phred_average (:32-42) from its two left-to-right sums
void phred_from_sums(double total, double first, size_t n, uint32_t& avg, uint32_t& delta);
// 2 bits per base, A=0 C=1 G=2 T=3 (either case), 16 bases per word; false if
// the read holds anything else
bool pack_2bit(const char* seq, size_t n, uint32_t* out);

// CPUs this process may actually use: min(affinity mask, cgroup v2 cpu.max quota)
unsigned effective_cpus();

} // namespace gr
