/*
 * grpath_dev.h — measurement-only entry points of DEVELOPER builds of libgrpath_hip.so
 * (make -C goldrush_amd/csrc DEV=1; grp_dev_hooks() == 1).  The product library does not
 * export them and nothing on the product's path calls them: they are the prototypes that
 * priced designs DESIGN.md records as measured and rejected (tools/pshard_bench.py).
 */
#ifndef GRPATH_DEV_H
#define GRPATH_DEV_H

#include "grpath.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Measurement (round 5): grp_query_tiles' result through the POSITION-SHARDED form of the query — the filter cut into
 * `n_owners` ranges of buckets, every probe a record in its owner's bin (partition), the bins gathered in owner order,
 * the IDs handed back and voted on per tile (csrc/grp_pshard.inc) — with the owners on ONE device: what the
 * partition and return passes cost beside the gather, before any xGMI traffic.  Same tile summaries as
 * grp_query_tiles (lists in any order); times_ms[3] = partition, gather, vote (HIP events).  Not on the product's
 * path (DESIGN.md 7 has the decision it feeds).
 */
/* Measurement (round 5): the hash-and-test pass of a hashed filter of the ranks a batch touched —
 * `table_mib` MiB (a power of two), a share `fill` of its bits set, `n_hash` (1 or 2) bits tested per probe — over the
 * tiles of reads [first, first + count): *ms (best of three), *dirty_frames (frames with a probe that hits: what the
 * second decisions would still evaluate through the log), *frames.  csrc/grp_pshard.inc; tools/touch_filter_bench.py. */
int grp_debug_touch_filter(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, uint32_t table_mib, double fill, uint32_t n_hash, float* ms, uint64_t* dirty_frames, uint64_t* frames);
int grp_pshard_query(grp_ctx* ctx, const grp_reads* reads, uint32_t first, uint32_t count, uint32_t n_owners, grp_tile_summary* tiles, grp_id_count* lists, uint64_t list_cap, uint64_t* list_used, float* times_ms);

#ifdef __cplusplus
}
#endif
#endif /* GRPATH_DEV_H */
