/* Synthetic ONT-like FASTQ, fast (tools/cli_end_to_end.py's generator wrote 51 GB in 227 s: most of a GPU call).
 *   gcc -O3 -fopenmp -o fqgen tools/fqgen.c -lm
 *   fqgen <out.fq> <reads> <genome_len> [seed]
 * The model of SURVEY 8(d) / goldrush_amd/synth.py: uniform random genome, reads at uniform places on either strand,
 * log-normal lengths (mean 25 kb, sigma 0.25, floor 20 kb), i.i.d. 3 % substitutions, 1 % insertions, 1 % deletions,
 * quality string of '5' (Q20).  Counter-based generators: read i is a function of (seed, i) alone, blocks of reads are
 * made by all threads and written in order.  Measurement support: not the reads of bench.py (another generator), the
 * same statistics. */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t mix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

static inline char gbase(uint64_t seed, uint64_t i) { return "ACGT"[mix64(seed ^ (i * 0xD1342543DE82EF95ULL)) >> 62]; }
static inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }

int main(int argc, char** argv)
{
  if (argc < 4) {
    fprintf(stderr, "usage: fqgen <out.fq> <reads> <genome_len> [seed]\n");
    return 2;
  }
  const uint64_t n_reads = strtoull(argv[2], 0, 10), G = (uint64_t)strtod(argv[3], 0), seed = argc > 4 ? strtoull(argv[4], 0, 10) : 1;
  FILE* f = fopen(argv[1], "wb");
  if (!f) {
    perror(argv[1]);
    return 1;
  }
  setvbuf(f, 0, _IOFBF, 1 << 24);
  enum { BLOCK = 2048 };
  char** buf = calloc(BLOCK, sizeof(char*));
  size_t* len = calloc(BLOCK, sizeof(size_t));
  size_t* cap = calloc(BLOCK, sizeof(size_t));
  const double sigma = 0.25, mu = log(25000.0) - 0.5 * sigma * sigma;
  uint64_t bytes = 0;
  for (uint64_t r0 = 0; r0 < n_reads; r0 += BLOCK) {
    const int nb = (int)(n_reads - r0 < BLOCK ? n_reads - r0 : BLOCK);
#pragma omp parallel for schedule(dynamic, 8)
    for (int b = 0; b < nb; ++b) {
      const uint64_t i = r0 + b, h = mix64(seed * 0x100000001B3ULL + i);
      /* log-normal length: Box-Muller on two hashed uniforms */
      const double u1 = ((mix64(h ^ 1) >> 11) + 1.0) / 9007199254740993.0, u2 = (mix64(h ^ 2) >> 11) / 9007199254740992.0;
      const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
      uint64_t L = (uint64_t)exp(mu + sigma * z);
      if (L < 20000) L = 20000;
      if (L > G) L = G;
      const uint64_t start = mix64(h ^ 3) % (G - L + 1);
      const int rev = (int)(mix64(h ^ 4) & 1);
      const size_t need = 2 * (L + L / 16 + 64) + 64;
      if (cap[b] < need) {
        free(buf[b]);
        buf[b] = malloc(need);
        cap[b] = need;
      }
      char* p = buf[b];
      p += sprintf(p, "@r%llu\n", (unsigned long long)i);
      char* s = p;
      for (uint64_t j = 0; j < L; ++j) {
        const uint64_t e = mix64(h ^ (0xABCDEF12345ULL + j * 0x9FB21C651E98DF25ULL));
        char c = rev ? comp(gbase(seed, start + L - 1 - j)) : gbase(seed, start + j);
        if ((e & 0xFFFFF) < 10486) continue;                                        /* 1 % deleted */
        if (((e >> 20) & 0xFFFFF) < 31457) c = "ACGT"[(strchr("ACGT", c) - "ACGT" + 1 + ((e >> 60) % 3)) & 3]; /* 3 % substituted */
        *p++ = c;
        if (((e >> 40) & 0xFFFFF) < 10486) *p++ = "ACGT"[(e >> 62) & 3];            /* 1 % followed by an inserted base */
      }
      const size_t n = (size_t)(p - s);
      *p++ = '\n';
      *p++ = '+';
      *p++ = '\n';
      memset(p, '5', n);
      p += n;
      *p++ = '\n';
      len[b] = (size_t)(p - buf[b]);
    }
    for (int b = 0; b < nb; ++b) {
      if (fwrite(buf[b], 1, len[b], f) != len[b]) {
        perror("write");
        return 1;
      }
      bytes += len[b];
    }
  }
  fclose(f);
  fprintf(stderr, "fqgen: %llu reads, %llu bytes\n", (unsigned long long)n_reads, (unsigned long long)bytes);
  return 0;
}
