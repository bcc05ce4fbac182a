// ORACLE — TEST INFRASTRUCTURE ONLY.
// C wrappers around the three first-party translation units of the reference that
// build from their own sources with nothing but libstdc++ (compiled IN PLACE from
// /root/reference by oracle/Makefile, target `ref`; the output goes to oracle/_ref/):
//   goldrush_path/spaced_seeds.cpp        make_seed_pattern
//   goldrush_path/calc_phred_average.cpp  calc_phred_average, sum_phred
// (the rest of the path needs btllib / sdsl-lite / sparsehash and cannot be built here).
// This file contains no reference code: it only calls it.
#include "calc_phred_average.hpp"
#include "spaced_seeds.hpp"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

extern "C" {

// h patterns written `stride` bytes apart; returns h
int
ref_make_seed_pattern(const char* preset, unsigned k, unsigned weight, unsigned h, char* out, size_t stride)
{
  const std::vector<std::string> v = make_seed_pattern(preset, k, weight, h);
  for (size_t i = 0; i < v.size(); ++i) {
    std::strncpy(out + i * stride, v[i].c_str(), stride - 1);
    out[i * stride + stride - 1] = '\0';
  }
  return (int)v.size();
}

void
ref_calc_phred_average(const char* qual, size_t n, uint32_t* avg, uint32_t* delta)
{
  const std::pair<uint32_t, uint32_t> r = calc_phred_average(std::string(qual, n));
  *avg = r.first;
  *delta = r.second;
}

double
ref_sum_phred(const char* qual, size_t n)
{
  return sum_phred(std::string(qual, n));
}

}
