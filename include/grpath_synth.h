/*
 * grpath_synth.h — synthetic ONT-like read generator on the GPU (measurement
 * support for bench.py; not part of the reference's path and not used by the
 * goldrush-path CLI).  Exported by libgrpath_hip.so.
 *
 * Model: uniform random genome of `genome_len` bases defined
 * by a counter-based generator (base i = f(genome_seed, i); optionally repeat-rich,
 * repeat_frac below); read r is the
 * substring starting at start[r] (genome coordinates wrap), reverse-
 * complemented when strand[r] != 0, passed through i.i.d. errors: each source
 * base is deleted with probability p_del, otherwise emitted (substituted by a
 * different base with probability p_sub) and followed by one random inserted
 * base with probability p_ins.  Exactly len[r] bases are produced per read and
 * written 2-bit packed (grpath.h encoding) at word_off[r].
 */
#ifndef GRPATH_SYNTH_H
#define GRPATH_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
  uint64_t genome_len;
  uint64_t genome_seed;
  uint64_t error_seed;
  float p_sub, p_ins, p_del;
  /* Round 5: the share of the genome made of REPEATS, 0 = none (the uniform genome of rounds
   * 1-4: two reads share k-mers only where they overlap).  With f > 0 the genome is cut into slots of 6144 bases; a
   * slot is a repeat copy with probability 1.5 f and then begins with its family's unit — 2 to 6 kb long (per family),
   * the family's consensus with a copy-specific substitution at 1 to 5 % of the positions (per family) — the rest of
   * the slot and every other slot are unique sequence.  Families come in three tiers of ~10 000, ~1 000 and ~30 copies
   * (a third of the repeat slots each; fewer copies where the genome is too small for that many). */
  float repeat_frac;
} grp_synth_params;

/*
 * d_packed_out: device memory of word_off[n_reads] 32-bit words (caller
 * allocates; e.g. a torch tensor or grp_synth_alloc).  start / len / strand /
 * word_off are HOST arrays.  `stream` is a hipStream_t (NULL = default stream);
 * the call returns after the kernel has completed.  Returns 0 on success.
 */
int grp_synth_reads(const grp_synth_params* p,
                    const uint64_t* start,
                    const uint32_t* len,
                    const uint8_t* strand,
                    const uint64_t* word_off,
                    uint32_t n_reads,
                    void* d_packed_out,
                    void* stream);

/* plain hipMalloc / hipFree / device->host copy, for callers without torch */
void* grp_synth_alloc(uint64_t bytes);
void grp_synth_free(void* d);
int grp_synth_download(const void* d, uint64_t bytes, void* host);
const char* grp_synth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
