// goldrush-path — drop-in for the reference binary of the same name
// (goldrush_path/meson.build:6): same flags, same outputs; the miBF work runs
// on an MI355X through libgrpath_hip.so.  There is no CPU path.
#include "../../../include/grpath.h"
#include "../../../include/grpath_host.h"
#include "../../../include/grpath_ingest.h"

static grp_engine_vt
hip_engine()
{
  grp_engine_vt vt{};
  vt.create = [](const grp_params* p, void** out) { return grp_create(p, reinterpret_cast<grp_ctx**>(out)); };
  vt.destroy = [](void* c) { grp_destroy(static_cast<grp_ctx*>(c)); };
  vt.last_error = [](const void* c) { return grp_last_error(static_cast<const grp_ctx*>(c)); };
  vt.reads_upload = [](void* c, const uint32_t* packed, const uint64_t* off, const uint32_t* len, uint32_t n, void** out) {
    return grp_reads_upload(static_cast<grp_ctx*>(c), packed, off, len, n, reinterpret_cast<grp_reads**>(out));
  };
  vt.reads_free = [](void* r) { grp_reads_free(static_cast<grp_reads*>(r)); };
  vt.bv_insert = [](void* c, const void* r, uint32_t first, uint32_t count) { return grp_bv_insert(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count); };
  vt.finalize = [](void* c, uint64_t* pop) { return grp_finalize(static_cast<grp_ctx*>(c), pop); };
  vt.query_tiles = [](void* c, const void* r, uint32_t first, uint32_t count, grp_tile_summary* t, grp_id_count* l, uint64_t cap, uint64_t* used, grp_query_stats* st) {
    return grp_query_tiles(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count, t, l, cap, used, st);
  };
  vt.insert_tiles = [](void* c, const void* r, uint32_t ri, uint32_t ts, uint32_t te, uint32_t id) { return grp_insert_tiles(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), ri, ts, te, id); };
  vt.reset_ids = [](void* c) { return grp_reset_ids(static_cast<grp_ctx*>(c)); };
  vt.sync = [](void* c) { return grp_sync(static_cast<grp_ctx*>(c)); };
  vt.classify_reads = [](void* c, const void* r, uint32_t first, uint32_t count, const grp_decide_params* dp, grp_read_decision* out) {
    return grp_classify_reads(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count, dp, out);
  };
  vt.insert_read = [](void* c, const void* r, uint32_t ri, uint32_t ts, uint32_t te, uint32_t block, uint32_t first_id, uint32_t off) {
    return grp_insert_read(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), ri, ts, te, block, first_id, off);
  };
  vt.classify_begin = [](void* c, const void* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot) {
    return grp_classify_reads_begin(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count, dp, slot);
  };
  vt.classify_end = [](void* c, uint32_t slot, grp_read_decision* out) { return grp_classify_reads_end(static_cast<grp_ctx*>(c), slot, out); };
  vt.stream_begin = [](void* c, const void* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot, uint32_t stripe, uint32_t n_owners, uint32_t owner, const grp_read_decision** dec) {
    return grp_classify_stream_begin_striped(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count, dp, slot, stripe, n_owners, owner, dec);
  };
  vt.stream_abort = [](void* c, uint32_t slot) { return grp_classify_stream_abort(static_cast<grp_ctx*>(c), slot); };
  vt.stream_poll = [](void* c, uint32_t slot) { return grp_classify_stream_poll(static_cast<grp_ctx*>(c), slot); };
  vt.stream_end = [](void* c, uint32_t slot, uint32_t* n) { return grp_classify_stream_end(static_cast<grp_ctx*>(c), slot, n); };
  vt.batch_insert = [](void* c, const void* r, const grp_batch_insert* ins, uint32_t n, uint32_t block, uint32_t first) {
    return grp_batch_insert_reads(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), ins, n, block, first);
  };
  vt.batch_classify = [](void* c, const void* r, uint32_t first, uint32_t count, const grp_decide_params* dp, const uint32_t* id_floor, grp_read_decision* out) {
    return grp_batch_classify(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count, dp, id_floor, out);
  };
  vt.batch_verify = [](void* c, const void* r, uint32_t first, uint32_t count, uint32_t extra, const grp_decide_params* dp, const uint32_t* id_floor, grp_read_decision* out) {
    return grp_batch_verify(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count, extra, dp, id_floor, out);
  };
  vt.window_overlap = [](void* c, const void* r, uint32_t first, uint32_t count, uint32_t threshold, uint32_t* prev_out) {
    return grp_window_overlap(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count, threshold, prev_out);
  };
  vt.batch_undo = [](void* c, uint32_t from_read, uint32_t id_floor) { return grp_batch_undo(static_cast<grp_ctx*>(c), from_read, id_floor); };
  vt.batch_end = [](void* c) { return grp_batch_end(static_cast<grp_ctx*>(c)); };
  vt.ntcard_begin = [](void* c, uint32_t sbits) { return grp_ntcard_begin(static_cast<grp_ctx*>(c), sbits); };
  vt.ntcard_add = [](void* c, const void* r, uint32_t first, uint32_t count, const uint32_t* extra) {
    return grp_ntcard_add(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count, extra);
  };
  vt.ntcard_finish = [](void* c, uint64_t* z) { return grp_ntcard_finish(static_cast<grp_ctx*>(c), z); };
  vt.set_filter_size = [](void* c, uint64_t m) { return grp_set_filter_size(static_cast<grp_ctx*>(c), m); };
  vt.fastq_parse = [](void* c, const char* text, uint64_t n, int fin, void** out, uint64_t* nrec, uint64_t* used, int* stopped) {
    return grp_fastq_parse(static_cast<grp_ctx*>(c), text, n, fin, reinterpret_cast<grp_fastq**>(out), nrec, used, stopped);
  };
  vt.fastq_records = [](void* fq, grp_fastq_record* out) { return grp_fastq_records(static_cast<grp_fastq*>(fq), out); };
  vt.fastq_pack = [](void* c, void* fq, const uint32_t* sel, uint32_t n, void** out) {
    return grp_fastq_pack(static_cast<grp_ctx*>(c), static_cast<grp_fastq*>(fq), sel, n, reinterpret_cast<grp_reads**>(out));
  };
  vt.fastq_free = [](void* fq) { grp_fastq_free(static_cast<grp_fastq*>(fq)); };
  vt.comm_unique_id = grp_comm_unique_id;
  vt.comm_init = [](void* c, const void* id, uint32_t world, uint32_t rank) { return grp_comm_init(static_cast<grp_ctx*>(c), id, world, rank); };
  vt.bv_merge_ranks = [](void* c) { return grp_bv_merge_ranks(static_cast<grp_ctx*>(c)); };
  vt.bv_words = [](const void* c, uint64_t* n) { return grp_bv_words(static_cast<const grp_ctx*>(c), n); };
  vt.bv_export_words = [](void* c, uint64_t first, uint64_t n, uint32_t* w) { return grp_bv_export_words(static_cast<grp_ctx*>(c), first, n, w); };
  vt.bv_or_words = [](void* c, uint64_t first, uint64_t n, const uint32_t* w) { return grp_bv_or_words(static_cast<grp_ctx*>(c), first, n, w); };
  vt.fastq_pin = [](void* c, const char* b, uint64_t n) { return grp_fastq_pin(static_cast<grp_ctx*>(c), b, n); };
  vt.fastq_unpin = [](void* c) { return grp_fastq_unpin(static_cast<grp_ctx*>(c)); };
  vt.stream_begin_resumable = [](void* c, const void* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot, const grp_read_decision** dec) {
    return grp_classify_stream_begin_resumable(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count, dp, slot, dec);
  };
  vt.stream_insert = [](void* c, uint32_t slot, uint32_t read, uint32_t ts, uint32_t te, uint32_t block, uint32_t first_id, uint32_t off, uint32_t* gen) {
    return grp_classify_stream_insert(static_cast<grp_ctx*>(c), slot, read, ts, te, block, first_id, off, gen);
  };
  vt.stream_begin_striped_resumable = [](void* c, const void* r, uint32_t first, uint32_t count, const grp_decide_params* dp, uint32_t slot, uint32_t stripe, uint32_t n_owners, uint32_t owner, const grp_read_decision** dec) {
    return grp_classify_stream_begin_striped_resumable(static_cast<grp_ctx*>(c), static_cast<const grp_reads*>(r), first, count, dp, slot, stripe, n_owners, owner, dec);
  };
  vt.stream_resumable = [](void* c, uint32_t slot) { return grp_classify_stream_resumable(static_cast<grp_ctx*>(c), slot); };
  vt.stream_insert_done = [](void* c, uint32_t slot) { return grp_classify_stream_insert_done(static_cast<grp_ctx*>(c), slot); };
  vt.fastq_prefetch = [](void* c, const char* text, uint64_t n) { return grp_fastq_prefetch(static_cast<grp_ctx*>(c), text, n); };
  vt.occupancy_hint = [](void* c, double o) { return grp_set_occupancy_hint(static_cast<grp_ctx*>(c), o); };
  return vt;
}

int
main(int argc, char** argv)
{
  const grp_engine_vt vt = hip_engine();
  return gr_path_main(argc, argv, &vt);
}
