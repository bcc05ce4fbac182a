#include "gr_params.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sched.h>
#include <string>

namespace gr {

std::vector<std::string>
make_seed_pattern(const std::string& preset, unsigned k, unsigned weight, unsigned h, bool log)
{
  // spaced_seeds.cpp:7-69.  Without a preset the left half is drawn with glibc
  // srand(123)/rand()%2 until it has weight/2 ones (first position forced to 1);
  // the right half is its mirror image.  Seed i = left + i zeros + right.
  std::string left, right;
  if (preset.empty()) {
    srand(123);
    if (log) {
      std::cerr << "Designing base symmetrical spaced seed\nUsing:\nspan: " << k << "\nweight: " << weight << std::endl;
    }
    std::vector<unsigned> v(k / 2, 0);
    if (!v.empty()) {
      v[0] = 1;
    }
    size_t ones = 0;
    while (ones != weight / 2) {
      for (size_t i = 1; i < v.size(); ++i) {
        v[i] = (unsigned)(rand() % 2);
      }
      ones = (size_t)std::count(v.begin(), v.end(), 1u);
    }
    for (unsigned b : v) {
      left.push_back(b ? '1' : '0');
    }
    right.assign(left.rbegin(), left.rend());
  } else {
    if (log) {
      std::cerr << "Using preset spaced seed\nwith:\n\tspan: " << preset.size() << "\n\tweight: " << std::count(preset.begin(), preset.end(), '1') << std::endl;
    }
    left = preset.substr(0, preset.size() / 2);
    right = preset.substr(preset.size() / 2, preset.size() / 2);
  }
  std::vector<std::string> out;
  for (unsigned i = 0; i < h; ++i) {
    out.push_back(left + std::string(i, '0') + right);
  }
  return out;
}

uint64_t
hash_universe(uint64_t weight, uint64_t genome_size, uint64_t hash_num)
{
  // goldrush_path.cpp:1113-1121: size_t * const float * size_t -> float math
  const uint64_t pow4 = (uint64_t)std::pow(4.0, (double)weight);
  const uint64_t base = std::min<uint64_t>(pow4, 2 * genome_size);
  volatile float f = (float)base * 0.5f;
  f = f * (float)hash_num;
  return (uint64_t)f;
}

uint64_t
calc_optimal_size(uint64_t entries, unsigned hash_num, double occupancy)
{
  const size_t approx = (size_t)(-(double)entries * (double)hash_num / std::log(1.0 - occupancy));
  return approx + (64 - approx % 64);
}

namespace {
// pow(10.0, -(c - 33) / 10.0) for every char value: the same libm call the
// reference makes per base (calc_phred_average.cpp:21-24), made once
struct DelogTable
{
  double v[256];
  DelogTable()
  {
    for (int c = 0; c < 256; ++c) {
      const int q = (int)((char)c - 33);
      v[c] = std::pow(10.0, -q / 10.0);
    }
  }
};
const DelogTable kDelog;
} // namespace

void
calc_phred_average(const char* qual, size_t n, uint32_t& avg, uint32_t& delta)
{
  // same left-to-right double summation as the reference => bit-identical sums
  double total = 0.0, first = 0.0;
  for (size_t i = 0; i < n; ++i) {
    total += kDelog.v[(unsigned char)qual[i]];
    if (i == n / 2 - 1) {
      first = total;
    }
  }
  phred_from_sums(total, first, n, avg, delta);
}

void
phred_from_sums(double total, double first, size_t n, uint32_t& avg, uint32_t& delta)
{
  double second = total - first;
  second = second / (n * 0.5);
  first = first / (n * 0.5);
  avg = (uint32_t)(-10 * std::log10(total / n));
  delta = (uint32_t)std::abs((int32_t)(-10 * std::log10(first)) - (int32_t)(-10 * std::log10(second)));
}

double
sum_phred(const char* qual, size_t n)
{
  double total = 0;
  for (size_t i = 0; i < n; ++i) {
    total += kDelog.v[(unsigned char)qual[i]];
  }
  return total;
}

namespace {
struct CodeTable
{
  uint8_t t[256];
  CodeTable()
  {
    for (auto& x : t) {
      x = 4;
    }
    t['A'] = t['a'] = 0;
    t['C'] = t['c'] = 1;
    t['G'] = t['g'] = 2;
    t['T'] = t['t'] = 3;
  }
};
const CodeTable kCode;
} // namespace

bool
pack_2bit(const char* seq, size_t n, uint32_t* out)
{
  uint32_t bad = 0;
  size_t w = 0;
  size_t i = 0;
  for (; i + 16 <= n; i += 16, ++w) {
    uint32_t v = 0;
    for (unsigned j = 0; j < 16; ++j) {
      const uint32_t c = kCode.t[(unsigned char)seq[i + j]];
      bad |= c;
      v |= (c & 3u) << (2 * j);
    }
    out[w] = v;
  }
  if (i < n) {
    uint32_t v = 0;
    for (unsigned j = 0; i + j < n; ++j) {
      const uint32_t c = kCode.t[(unsigned char)seq[i + j]];
      bad |= c;
      v |= (c & 3u) << (2 * j);
    }
    out[w] = v;
  }
  return (bad & 4u) == 0;
}

unsigned
ntcard_sbits(uint64_t input_bytes)
{
  return input_bytes < 50000000000ULL ? 7u : 11u;
}

uint64_t
ntcard_f0(uint64_t zero0, uint64_t zero1, unsigned sbits)
{
  const unsigned r_bits = 27; // nts::rBits
  const size_t n_samp = 2;    // nts::nSamp
  // pMean[0] = (p[0][0] + p[1][0]) / (1.0 * nSamp), p[][] unsigned
  double p_mean0 = 0.0;
  p_mean0 += (unsigned)zero0;
  p_mean0 += (unsigned)zero1;
  p_mean0 /= 1.0 * n_samp;
  const double f0_mean = (ssize_t)((r_bits * log(2) - log(p_mean0)) * 1.0 * ((size_t)1 << (sbits + r_bits)));
  return (size_t)f0_mean;
}

void
ntcard_split(const char* seq, size_t n, unsigned k, unsigned h, std::vector<std::pair<size_t, size_t>>& runs, std::vector<uint32_t>& extra)
{
  runs.clear();
  extra.clear();
  size_t start = 0;
  for (size_t i = 0; i <= n; ++i) {
    const bool clean = i < n && (kCode.t[(unsigned char)seq[i]] & 4u) == 0;
    if (!clean) {
      if (i - start >= k) {
        runs.emplace_back(start, i - start);
      }
      start = i + 1;
    }
  }
  extra.assign(runs.size() * h, 0);
  std::vector<uint64_t> V(h, 0);
  std::vector<size_t> last(h, (size_t)-1);
  for (size_t r = 0; r < runs.size(); ++r) {
    for (unsigned s = 0; s < h; ++s) {
      if (runs[r].second >= k + s) {
        V[s] += runs[r].second - (k + s) + 1;
        last[s] = r;
      }
    }
  }
  const uint64_t F = h ? V[0] : 0; // the shortest span has the most windows
  for (unsigned s = 0; s < h; ++s) {
    if (V[s]) {
      extra[last[s] * h + s] = (uint32_t)(F - V[s]);
    }
  }
}

unsigned
effective_cpus()
{
  unsigned n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    n = (unsigned)CPU_COUNT(&set);
  }
  std::ifstream f("/sys/fs/cgroup/cpu.max");
  std::string quota;
  double period = 0;
  if (f >> quota >> period && quota != "max" && period > 0) {
    const double q = std::atof(quota.c_str()) / period;
    if (q >= 1.0 && q < n) {
      n = (unsigned)q;
    }
  }
  return n ? n : 1;
}

} // namespace gr
