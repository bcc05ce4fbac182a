/*
 * grpath_ingest.h — FASTQ ingest on the GPU: record
 * splitting, read filters' inputs and 2-bit packing for goldrush-path's three
 * passes over the input (phred median goldrush_path.cpp:79-107, bit-vector fill
 * :235-339, classification :1229-1256).  Exported by libgrpath_hip.so.
 *
 * The reference gets its records from btllib::SeqReader (external): id = header
 * up to the first whitespace, 4-line records, sequence case-folded to upper case.
 * Here a chunk of raw FASTQ text is uploaded once and parsed on the device; the
 * host keeps the text for the output files and only receives 56 bytes per record.
 *
 * Exactness of the Phred filter: calc_phred_average (calc_phred_average.cpp:8-43)
 * sums 10^(-Q/10) left to right in double precision and truncates
 * -10*log10(sum/n).  The device performs the SAME left-to-right double sums from
 * a 256-entry table of the host's pow() values (IEEE additions, no reassociation),
 * so the sums are bit-identical; log10 and the truncation stay on the host.
 */
#ifndef GRPATH_INGEST_H
#define GRPATH_INGEST_H

#include "grpath.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct grp_fastq grp_fastq; /* a parsed chunk of FASTQ text resident in HBM */

typedef struct
{
  uint64_t id_off;    /* byte offset of the id (after '@') inside the chunk */
  uint64_t seq_off;
  uint64_t qual_off;
  uint32_t id_len;    /* up to the first whitespace of the header */
  uint32_t seq_len;   /* trailing CR / blanks trimmed */
  uint32_t qual_len;
  uint32_t flags;     /* GRP_FQ_NON_ACGT: find_first_not_of("ACGTacgt") != npos (goldrush_path.cpp:293) */
  double phred_sum;   /* sum_{i<qual_len} 10^(-(q_i-33)/10), left to right */
  double phred_first; /* the same sum after i = qual_len/2 - 1 (0 when never reached) */
} grp_fastq_record;

#define GRP_FQ_NON_ACGT 1u

/*
 * Upload and parse `n_bytes` of FASTQ text (host memory; pinned memory — grp_fastq_pin — is faster).
 * final_chunk != 0: the text ends the file (a last line without newline is a line).
 * Returns the number of complete records and how many bytes they span; the caller
 * re-submits the unconsumed tail at the start of its next chunk.  *stopped is set
 * when a record's header does not start with '@' (records before it are returned;
 * the reader ends there, like a reader at end of input).
 */
int grp_fastq_parse(grp_ctx* ctx, const char* text, uint64_t n_bytes, int final_chunk, grp_fastq** out, uint64_t* n_records, uint64_t* bytes_consumed, int* stopped);
/*
 * Optional (round 5): start the upload of a COMING chunk now.  `text` is that chunk's body: the later grp_fastq_parse is
 * given a text that ENDS with exactly these bytes at this address — the body itself, or the body with up to GRP_FASTQ_PREFETCH_FRONT bytes (16 MiB) in front
 * of it (the unconsumed tail of the chunk before, which is only known once that chunk has been parsed); it finds the body on
 * the device and uploads what is in front of it alone.  Up to two prefetches may be pending, parsed in the order they were
 * issued.  Why: the copy of a 256 MiB chunk takes ~5 ms and, issued from inside grp_fastq_parse, only began when the fill
 * of the chunk before had ended (the timeline: tools/dev/r5_ingest_timeline.sh); two chunks ahead it is off the path of
 * every chunk.  The body must stay unchanged until its parse returns.  A prefetch that is not followed by the matching
 * parse is simply dropped (with every prefetch behind it).  GRP_ERR_BUSY: no device buffer is free for it right now (not
 * an error: the parse uploads as before).  (NULL, 0): every pending prefetch is forgotten and its copy waited for — call it
 * before the buffers the bodies live in are freed.
 */
#define GRP_FASTQ_PREFETCH_FRONT (16ull << 20) /* bytes a later parse may put in front of a prefetched body (the engine leaves this much room) */
int grp_fastq_prefetch(grp_ctx* ctx, const char* text, uint64_t n_bytes);
/* copy the record table (n_records entries) to the host */
int grp_fastq_records(grp_fastq* fq, grp_fastq_record* out);
/*
 * 2-bit pack the selected records (ascending indices; they must be pure ACGT) into a
 * read batch that lives on the device — no host packing, no second upload.  The
 * batch is independent of `fq` afterwards.
 */
int grp_fastq_pack(grp_ctx* ctx, grp_fastq* fq, const uint32_t* sel, uint32_t n_sel, grp_reads** out);
void grp_fastq_free(grp_fastq* fq);
/*
 * Optional: page-lock the caller's chunk buffer (hipHostRegister) so that grp_fastq_parse's upload
 * of text inside [buffer, buffer + n_bytes) is one DMA.  One buffer per context; _pin replaces the
 * previous one.  The caller MUST call grp_fastq_unpin (or grp_destroy) BEFORE it frees or
 * reallocates the buffer.  GRP_ERR_HIP: the buffer could not be locked — parse works all the same.
 */
int grp_fastq_pin(grp_ctx* ctx, const char* buffer, uint64_t n_bytes);
int grp_fastq_unpin(grp_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
