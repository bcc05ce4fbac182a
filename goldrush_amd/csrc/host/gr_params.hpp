// Small pure host functions of GoldRush-Path: seed design, filter sizing,
// Phred statistics, 2-bit packing.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
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

// ---- --ntcard (goldrush_path/ntcard.hpp) ----------------------------------------
// nts::sBits for an input of this many bytes (:35, :177-178)
unsigned ntcard_sbits(uint64_t input_bytes);
// compEst's F0Mean as getHist stores it (:124-136, :232), from the zero buckets of
// the two sample tables of one seed
uint64_t ntcard_f0(uint64_t zero0, uint64_t zero1, unsigned sbits);
// A record with non-ACGT characters as the engine wants it (grp_ntcard_add): its
// maximal ACGT runs of at least k bases (offset, length), and per run and seed the
// number of extra counts of the run's last window — the stale repeats of
// multiLensfrHashIterator (seed s of span k+s visits its V_s clean windows, the
// iterator runs max_s V_s frames; multiLensfrHashIterator.hpp:49-68).
void ntcard_split(const char* seq, size_t n, unsigned k, unsigned h, std::vector<std::pair<size_t, size_t>>& runs, std::vector<uint32_t>& extra);

} // namespace gr
