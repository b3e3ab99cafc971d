/*
 * groot_host.h -- C ABI of libgroot_host.so: the host-side (no GPU) pieces either side of the
 * `groot align` hot path.  Plain C types only, so that a cgo / ctypes binding is a direct
 * transliteration (INTEGRATION.md shows the cgo stub).
 *
 *   index side   : what `groot index` produces and `groot align` loads
 *                  (cmd/index.go:57-133, src/pipeline/index.go:37-211, src/graph/graph.go:37-396)
 *   output side  : graph weighting / pruning / GFA + BAM writing after the device path returns
 *                  (src/graph/graph.go:401-525, src/graph/graphio.go:19-154, src/pipeline/boss.go:45-105)
 *
 * Every function returns 0 on success or a negative GROOT_E_* code; groot_host_last_error() gives
 * the message for the calling thread.
 */
#ifndef GROOT_HOST_H
#define GROOT_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "groot_index.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GROOT_OK 0
#define GROOT_E_INVALID (-1)   /* bad argument                                                    */
#define GROOT_E_IO (-2)        /* file could not be read / written                                */
#define GROOT_E_FORMAT (-3)    /* malformed MSA / GFA / FASTQ / index file                        */
#define GROOT_E_NOMEM (-4)
#define GROOT_E_DEVICE (-5)    /* HIP runtime error (device library only)                         */
#define GROOT_E_NOSPACE (-6)   /* caller buffer too small; the needed size is reported            */
#define GROOT_E_SHORT_READ (-7) /* read shorter than k: the reference panics (boss.go:164-166)    */
#define GROOT_E_REVCOMP (-8)   /* read byte > 'T' reached RevComplement: reference panics (seqio.go:126) */
#define GROOT_E_STATE (-9)     /* call sequence error (collect without submit, ...)               */
#define GROOT_E_UNSUPPORTED (-10)

typedef struct groot_index groot_index; /* owning handle; groot_index_view() borrows from it */

const char *groot_host_last_error(void);
/* CPUs this process may really use = min(affinity mask, cgroup CPU quota), or $GROOT_THREADS: what "0 = all cores"
 * means throughout this library (a container may show 256 hardware threads and grant 16) */
uint32_t groot_host_usable_cpus(void);
const char *groot_host_version(void); /* "1.1.2": must equal Info.Version (cmd/align.go:96) */

/* ---- index: `groot index` (cmd/index.go:44-52 defaults k=31 s=21 w=100 x=8 y=4) ---------------- */
typedef struct groot_index_params {
    uint32_t kmer_size;       /* -k */
    uint32_t sketch_size;     /* -s */
    uint32_t window_size;     /* -w */
    uint32_t num_part;        /* -x */
    uint32_t max_k;           /* -y */
    uint32_t max_sketch_span; /* --maxSketchSpan (never enforced by the reference: graph.go:33,225) */
    uint32_t n_threads;       /* -p ; 0 = all cores */
    uint32_t reserved;
} groot_index_params;

void groot_index_params_default(groot_index_params *p);

/* MSAconverter + GraphSketcher + SketchIndexer (src/pipeline/index.go:37-211) over every
 * cluster*.msa in msa_dir, graph ids = position in the lexically sorted file list
 * (filepath.Glob, cmd/index.go:143). */
int groot_index_build_msa_dir(const char *msa_dir, const groot_index_params *p, groot_index **out);
/* Same, with the window sketches (Sequence.RunMinHash on every path window, graph.go:292-296) computed by a
 * caller-supplied batch function -- e.g. groot_hip_sketch, which makes `groot index` sketch on the GPU while this
 * library stays free of device code.  fn gets n sequences (seq_off has n+1 entries) and fills out[n*sketch_size];
 * it is called from one thread at a time and returns 0 on success. */
typedef int (*groot_sketch_fn)(void *user, const uint8_t *seq_concat, const uint64_t *seq_off, uint32_t n, uint64_t *out);
int groot_index_build_msa_dir_with(const char *msa_dir, const groot_index_params *p, groot_sketch_fn fn, void *user,
                                   groot_index **out);
int groot_index_build_msa_files(const char *const *files, uint32_t n_files, const groot_index_params *p,
                                groot_index **out);
/* same pipeline starting from GFA files (graph.LoadGFA + CreateGrootGraph, graphio.go:115-138,
 * graph.go:37-147); used with the reference's src/graph/test.gfa fixture */
int groot_index_build_gfa_files(const char *const *files, uint32_t n_files, const groot_index_params *p,
                                groot_index **out);
int groot_index_save(const groot_index *idx, const char *path);   /* flat little-endian .gidx file */
int groot_index_load(const char *path, groot_index **out);
/* replaces Info.Load + ContainmentIndex.Load (src/pipeline/runtime.go:75-91, src/lshe/lshe.go:95-120; called at
 * cmd/align.go:93-107): reads an index directory written by the reference's `groot index` -- the Go encoding/gob
 * streams <dir>/groot.gg (pipeline.Info incl. graph.Store) and <dir>/groot.lshe (lshe.ContainmentIndex with its
 * WindowLookup map) -- into the flat index.  Window ids are assigned in the canonical order (GraphID, Key.Node,
 * Key.OffSet, the "-<i>" suffix of the WindowLookup key, src/pipeline/index.go:195-203).  GROOT_E_FORMAT for a stream
 * that does not decode or whose cross references do not resolve. */
int groot_index_load_gob(const char *gg_path, const char *lshe_path, groot_index **out);
/* the reverse: Info.Dump + ContainmentIndex.Dump (src/pipeline/runtime.go:64-72, src/lshe/lshe.go:71-92; cmd/index.go:96-106,
 * 130-131) -- writes <dir>/groot.gg and <dir>/groot.lshe with the fields `groot index` sets, so that the reference's own
 * subcommands can load an index built here.  The directory must exist. */
int groot_index_save_gob(const groot_index *idx, const char *dir, uint32_t max_sketch_span);
/* renders every top-level value of a gob stream as JSON text (structs as objects holding the fields present on the
 * wire, maps as [[key,value],...]): the decoder behind groot_index_load_gob, exposed for inspection and for the
 * known-answer tests on the byte vectors of the gob documentation.  *needed = bytes incl. the terminating NUL; the
 * text is written only when cap >= *needed. */
int groot_gob_to_json(const uint8_t *data, uint64_t n, char *out, uint64_t cap, uint64_t *needed);
void groot_index_get_view(const groot_index *idx, groot_index_view *view);
/* One O(n) consistency pass over a view (every node / edge / path / window index in range, offset arrays monotone and
 * ending at their payload, path_words wide enough): GROOT_E_FORMAT with the first inconsistency in
 * groot_host_last_error().  groot_index_load / groot_index_load_gob run it on what they read, groot_hip_open runs the
 * same pass on the view it is given before anything is uploaded. */
int groot_index_view_check(const groot_index_view *view);
void groot_index_free(groot_index *idx);

/* fine-grained mirror of Sequence.RunMinHash(k, s, false, nil) (src/seqio/seqio.go:40-68) used by
 * the index builder for graph windows (graph.go:292-296).  Host arithmetic; the align path uses
 * the device kernels in libgroot_hip.so instead. */
int groot_host_window_sketch(const uint8_t *seq, uint32_t len, uint32_t k, uint32_t s, uint64_t *sketch);

/* One successful traversal of performAlignment (alignment.go:162-193) together with the path ids
 * processTraversal (alignment.go:263-317) assigns to it: bit p of mask = local path id p.
 * AlignRead emits one sam.Record per set bit, ascending p, traversals in `ord` order; the record's
 * Pos = Position[p] of `node` + offset.  groot_host_expand_alns() does that expansion. */
typedef struct groot_trav {
    uint32_t read_id;
    uint32_t graph_id;
    uint32_t node;    /* global node index of the first node of the traversal            */
    uint32_t offset;  /* offset in that node where the alignment starts                   */
    uint16_t ord;     /* emission order within the read (graphs ascending, DFS order)     */
    uint8_t  flags;   /* GROOT_TRAV_* */
    uint8_t  reserved;
} groot_trav;
#define GROOT_TRAV_RC 1u          /* read.RC: the reverse complement aligned (sam.Reverse)           */
#define GROOT_TRAV_START_CLIP 2u  /* 1H before the M op (alignment.go:72-85)                          */
#define GROOT_TRAV_END_CLIP 4u    /* 1H after the M op (alignment.go:87-103)                          */
#define GROOT_TRAV_FIRST 8u       /* first traversal of its (read, graph) AlignRead call              */

/* one sam.Record of AlignRead in id form (alignment.go:114-156) */
typedef struct groot_aln {
    uint32_t read_id;
    uint32_t graph_id;
    uint32_t path_id;   /* local path id: record.Ref = references[ID]                      */
    uint32_t ref_id;    /* global path index = position of its @SQ line (graph order)       */
    uint32_t pos;       /* record.Pos, 0-based                                              */
    uint8_t start_clip, end_clip, rc, secondary;
} groot_aln;

/* ---- traversal records -> alignment records ----------------------------------------------------- */
/* For every traversal, ascending set bit p of its mask: one record with Pos = Position[p] of the
 * traversal's first node + offset (alignment.go:296); Secondary on all but the first record of each
 * AlignRead call (alignment.go:147-149).  *n_out = records available; at most cap are written. */
int groot_host_expand_alns(const groot_index_view *idx, const groot_trav *travs, const uint64_t *masks, uint64_t n_trav,
                           groot_aln *out, uint64_t cap, uint64_t *n_out);

/* Path sets as groot_hip_collect hands them out (compact: max(1, ceil(paths of the traversal's graph / 8)) bytes per
 * traversal, back to back) -> path_words words per traversal, the layout groot_host_expand_alns takes. */
int groot_host_unpack_masks(const groot_index_view *idx, const groot_trav *travs, uint64_t n_trav, const uint8_t *compact_masks,
                            uint64_t *masks /*[n_trav * path_words]*/);

/* ---- graph weighting after alignment ------------------------------------------------------------ */
/* Replays GrootGraph.IncrementSubPath (graph.go:401-451) from the exact per-(kmerCount, window)
 * call counts the device accumulated: attempts[q * n_windows + w], q in [0, n_q).  Canonical order:
 * window ascending, kmerCount ascending, one floating-point add per call. */
int groot_host_weights(const groot_index_view *idx, const uint32_t *attempts, uint32_t n_q,
                       double *node_kmer_freq /*[n_nodes]*/, uint64_t *graph_kmer_total /*[n_graphs]*/);
/* The same replay from the compact table of groot_hip_attempts_export: row r holds the call counts of kmerCount
 * q_values[r] (strictly ascending), counts[r * n_windows + w]. */
int groot_host_weights_rows(const groot_index_view *idx, const uint32_t *q_values, uint32_t n_rows, const uint32_t *counts,
                            double *node_kmer_freq /*[n_nodes]*/, uint64_t *graph_kmer_total /*[n_graphs]*/);
/* GrootGraph.Prune (graph.go:455-525) over every graph */
int groot_host_prune(const groot_index_view *idx, const double *node_kmer_freq, double min_kmer_cov,
                     uint8_t *graph_kept /*[n_graphs]*/, uint8_t *path_kept /*[n_paths]*/,
                     uint8_t *node_removed /*[n_nodes]*/);
/* GrootGraph.SaveGraphAsGFA (graphio.go:19-112) for graph g after pruning; writes nothing and sets
 * *written=0 if no node has KmerFreq>0.  timestamp may be NULL (uses now). */
int groot_host_save_gfa(const groot_index_view *idx, uint32_t graph, const double *node_kmer_freq,
                        const uint8_t *path_kept, const uint8_t *node_removed, uint64_t total_kmers,
                        const char *timestamp, const char *file_name, int *written);

/* ---- FASTQ in (src/pipeline/sketch.go:41-77,175-238; seqio.go:173-188) -------------------------- */
typedef struct groot_fastq groot_fastq;
/* paths may end in .gz (sketch.go:60-68); n_files==0 reads stdin */
int groot_fastq_open(const char *const *files, uint32_t n_files, groot_fastq **out);
/* Fills caller buffers with up to max_reads reads: seq/qual/name are concatenations with offsets
 * (n+1 entries each).  Returns the number of reads (0 at end of input) or a negative error. */
int64_t groot_fastq_next_batch(groot_fastq *fq, uint32_t max_reads, uint8_t *seq, uint8_t *qual, uint64_t *seq_off,
                               uint64_t seq_cap, char *names, uint64_t *name_off, uint64_t name_cap);
void groot_fastq_close(groot_fastq *fq);

/* ---- BAM out (src/pipeline/boss.go:45-105,225-240; alignment.go:113-156) ------------------------ */
typedef struct groot_bam groot_bam;
typedef struct groot_aln_record {  /* one sam.Record of alignment.go:118-155 */
    const char *name; uint32_t name_len;        /* read.ID[1:]                                   */
    const uint8_t *seq; const uint8_t *qual;    /* read.Seq[0:seq_len], read.Qual[0:seq_len] raw  */
    uint32_t seq_len;
    uint32_t ref_id;                            /* index into the @SQ list (global path index)    */
    uint32_t pos;                               /* 0-based                                        */
    uint8_t start_clip, end_clip, reverse, secondary;
} groot_aln_record;
/* header: @HD VN:1.5, one @SQ per path (graphio.go:141-154), @PG ID:1 PN:groot CL:"groot align"
 * VN:1.1.2, @RG ID:readsID ... (boss.go:55-84).  path NULL or "-" = stdout.  date NULL = now. */
int groot_bam_open(const char *path, const groot_index_view *idx, const char *date, groot_bam **out);
int groot_bam_write(groot_bam *bam, const groot_aln_record *recs, uint64_t n);
/* BGZF write concurrency for large groot_bam_write calls (bam.NewWriter's third argument, boss.go:99); 0 = all cores */
int groot_bam_set_threads(groot_bam *bam, uint32_t n_threads);
/* Fast path of the collector: straight from the device's traversal records of one batch to BAM records (the
 * expansion of groot_host_expand_alns, the record fields of alignment.go:113-156 and the BGZF compression run in
 * parallel over chunks of traversals; output order = traversal order = read order).  The batch arrays are the ones
 * groot_fastq_next_batch filled. */
typedef struct groot_read_batch {
    const uint8_t *seq, *qual;     /* concatenated Seq / Qual (same offsets)           */
    const uint64_t *seq_off;       /* [n_reads+1]                                       */
    const char *names;             /* concatenated read.ID[1:]                          */
    const uint64_t *name_off;      /* [n_reads+1]                                       */
    uint32_t n_reads, first_read_id;
} groot_read_batch;
int groot_bam_write_travs(groot_bam *bam, const groot_index_view *idx, const groot_read_batch *batch, const groot_trav *travs,
                          const uint64_t *masks, uint64_t n_trav, uint64_t *n_records);
int groot_bam_close(groot_bam *bam);
/* BGZF compression level: -1 = zlib default = what bgzf.NewWriter uses in the reference, 0 = stored .. 9;
 * -2 = structural: the records of a read (one per path of a traversal, alignment.go:113-156) are written as deflate back-references
 * to the first one with the differing header bytes as literals -- no match search; the inflated BAM is byte for byte the same,
 * the file about 4x larger than at level 1, the writer an order of magnitude faster */
int groot_bam_set_level(groot_bam *bam, int level);
uint64_t groot_bam_bytes_written(const groot_bam *bam);

/* ---- parallel FASTQ ingest (src/pipeline/sketch.go:41-77,213-236; seqio.go:173-188) --------------------------------
 * One reader thread per input file (the next few files are opened ahead, so several gzip streams inflate at once), the
 * calling thread frames the text at record boundaries, and n_threads workers find the line breaks, parse the records
 * and pack the bases into the wire format of groot_hip_submit_packed16.  Batches come out in input order; lines of
 * consecutive files form one stream; a trailing partial record is dropped (FastqHandler.Run).  Records hold positions
 * into the batch's own copy of the FASTQ text: nothing is copied per read. */
typedef struct groot_reads groot_reads;
typedef struct groot_reads_batch groot_reads_batch;
typedef struct groot_reads_view {
    uint32_t n_reads, max_len;
    uint64_t n_bases, n_exc;
    const uint8_t *packed;         /* 2-bit bases, (n_bases+3)/4 bytes                      */
    const uint16_t *seq_len;       /* [n_reads]                                             */
    const uint64_t *exc_pos;       /* [n_exc] bytes other than ACGT, ascending position     */
    const uint8_t *exc_byte;
    const uint8_t *text;           /* FASTQ text the positions below point into             */
    const uint32_t *name_pos, *name_len;   /* read.ID[1:] (the record name, alignment.go:119) */
    const uint32_t *seq_pos;       /* read.Seq, seq_len[i] bytes                            */
    const uint32_t *qual_pos, *qual_len;   /* read.Qual as it came                           */
} groot_reads_view;
/* n_files == 0 reads stdin; n_threads 0 = all cores; block_bytes 0 = 256 MB of text per block; a batch holds at most
 * max_batch_reads reads (0 = 1<<20) and max_batch_bases bases (0 = 256 per read) */
int groot_reads_open(const char *const *files, uint32_t n_files, uint32_t n_threads, uint64_t block_bytes, uint32_t max_batch_reads,
                     uint64_t max_batch_bases, groot_reads **out);
int groot_reads_next(groot_reads *r, groot_reads_batch **out);   /* *out = NULL at the end of the input */
void groot_reads_batch_view(const groot_reads_batch *b, groot_reads_view *view);
void groot_reads_batch_free(groot_reads_batch *b);
uint64_t groot_reads_count(const groot_reads *r);
void groot_reads_close(groot_reads *r);
/* collector for such a batch: traversal records -> sam.Records -> BGZF, parallel like groot_bam_write_travs */
/* mask_ckpt != NULL: masks are the compact path sets of groot_hip_collect (bytes) with their checkpoints (every 256th traversal);
 * NULL: masks points at uint64_t words, path_words of them per traversal */
int groot_bam_write_batch(groot_bam *bam, const groot_index_view *idx, const groot_reads_view *reads, uint32_t first_read_id,
                          const groot_trav *travs, const void *masks, const uint32_t *mask_ckpt, uint64_t n_trav, uint64_t *n_records);

/* ---- packing reads for groot_hip_submit_packed (include/groot_hip.h) ------------------------------------ */
/* packed[(n_bases+3)/4]; exceptions (bytes other than A C G T, e.g. N or lower case) in ascending position, at most
 * exc_cap of them: GROOT_E_NOSPACE with *n_exc = the number needed otherwise.  n_threads 0 = all cores. */
int groot_host_pack_reads(const uint8_t *seq_concat, uint64_t n_bases, uint8_t *packed, uint64_t *exc_pos, uint8_t *exc_byte,
                          uint64_t exc_cap, uint64_t *n_exc, uint32_t n_threads);

/* ---- after the hot path: the reference's own consumer of the BAM ---------------------------------- */
/* `groot report` (src/reporting/reporting.go:33-173, cmd/report.go:104-129): breadth of coverage per reference from the
 * BAM of `groot align` (bam_path NULL = stdin).  One line "name\tread count\tlength\tcoverage cigar" per reference whose
 * covered fraction is >= cov_cutoff, written to out_path (NULL = stdout) in BAM header order; low_cov != 0 uses the
 * cutoff 0.97 and drops references with internal uncovered stretches (cmd/report.go:119-122, reporting.go:151-153). */
int groot_host_report(const char *bam_path, double cov_cutoff, int low_cov, const char *out_path, uint64_t *n_reported);

#ifdef __cplusplus
}
#endif
#endif
