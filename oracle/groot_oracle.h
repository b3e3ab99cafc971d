/*
 * groot_oracle.h -- CPU restatement of GROOT's `align` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker.  The product (libgroot_hip.so / libgroot_host.so) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the third-party arithmetic.  The reference
 * (will-rowe/groot v1.1.2) cannot be built here (no Go toolchain) and the arithmetic cores live in
 * un-vendored Go modules that are absent from /root/reference:
 *     github.com/will-rowe/nthash   v0.2.0   (ntHash rolling hash + MultiHash)
 *     github.com/ekzhu/lshensemble  v1.1.0   (LSH Ensemble: LshForest32, OptimalKL, Containment)
 * Their published algorithms are restated below from the module sources as recalled; the reference
 * tree holds no numeric golden vector for them (src/minhash/minhash_test.go:111-157 only checks
 * reverse-complement invariance), so the oracle is pinned on the reference's own property tests,
 * fixtures with truth-in-name reads and end-to-end assertions (tests/test_oracle_*.py), not on
 * hash-value vectors.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef GROOT_ORACLE_H
#define GROOT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Flat, read-only view of a GROOT index (graph.Store + lshe.ContainmentIndex), all arrays
 * little-endian POD.  Field-for-field the same layout as groot_index_view in include/groot_hip.h
 * so a test can hand one buffer to both sides; declared separately on purpose. */
typedef struct {
    uint32_t kmer_size, sketch_size, window_size, num_part, max_k, num_window_kmers;
    uint32_t path_words, reserved0;
    uint32_t n_graphs, n_nodes, n_edges, n_paths, n_windows, reserved1;
    uint64_t n_bases, n_np, n_cn, n_wref, n_name_bytes;
    const uint32_t *graph_node_off;  /* [n_graphs+1] nodes of graph g, in SortedNodes order     */
    const uint32_t *graph_path_off;  /* [n_graphs+1] global path index of local pathID 0        */
    const uint8_t  *graph_masked;    /* [n_graphs]                                              */
    const uint32_t *node_seg_id;     /* [n_nodes]   GrootGraphNode.SegmentID                    */
    const uint32_t *node_seq_off;    /* [n_nodes+1] into bases                                  */
    const uint32_t *node_edge_off;   /* [n_nodes+1] into edges (OutEdges order)                 */
    const uint32_t *node_np_off;     /* [n_nodes+1] into np_path/np_pos (PathIDs order)         */
    const uint64_t *node_mask;       /* [n_nodes*path_words] bitset of local path ids           */
    const uint8_t  *bases;           /* [n_bases] upper-case ACGTN                              */
    const uint32_t *edges;           /* [n_edges] global node index                             */
    const uint32_t *np_path;         /* [n_np] local path id                                    */
    const uint32_t *np_pos;          /* [n_np] GrootGraphNode.Position[pathID]                  */
    const uint32_t *path_len;        /* [n_paths]                                               */
    const uint32_t *path_name_off;   /* [n_paths+1]                                             */
    const char     *path_names;
    const uint32_t *win_graph;       /* [n_windows] Key.GraphID                                 */
    const uint32_t *win_node;        /* [n_windows] global node index of Key.Node               */
    const uint32_t *win_offset;      /* [n_windows] Key.OffSet                                  */
    const uint32_t *win_merge_span;  /* [n_windows] Key.MergeSpan                               */
    const uint32_t *win_cn_off;      /* [n_windows+1]                                           */
    const uint32_t *cn_node;         /* [n_cn] global node index, ascending SegmentID           */
    const uint32_t *cn_count;        /* [n_cn] Key.ContainedNodes value (integral)              */
    const uint32_t *win_ref_off;     /* [n_windows+1]                                           */
    const uint32_t *win_ref;         /* [n_wref] Key.Ref                                        */
    const uint64_t *win_sketch;      /* [n_windows*sketch_size]                                 */
} oracle_index;

/* one alignment record in canonical, ID-free-comparable form */
typedef struct {
    uint32_t read_id;     /* index of the read in the submitted batch                   */
    uint32_t graph_id;
    uint32_t path_id;     /* local path id (sam.Reference = references[ID])             */
    uint32_t ref_id;      /* global path index = graph_path_off[graph]+path_id          */
    uint32_t pos;         /* 0-based start (record.Pos)                                 */
    uint8_t  start_clip, end_clip, rc, secondary;
} oracle_aln;

typedef struct {
    uint32_t read_id;
    uint32_t window_id;
} oracle_seed;

typedef struct {
    uint64_t received, mapped, multimapped, alignments, seeds;
    uint64_t revcomp_panics; /* reads on which the reference would panic in RevComplement */
} oracle_counts;

typedef struct oracle_lshe oracle_lshe;
typedef struct oracle_run oracle_run;

/* ---- T1: will-rowe/nthash v0.2.0 + src/minhash/khf.go:18-60 ---- */
/* returns 0, or -1 if k > len (NewHasher error -> khf.go:39-41) or k==0/k>64/s==0 */
int oracle_khf_sketch(const uint8_t *seq, uint32_t len, uint32_t k, uint32_t s, uint64_t *sketch);
/* canonical ntHash of every k-mer (for unit tests); out has len-k+1 entries */
int oracle_nthash_canonical(const uint8_t *seq, uint32_t len, uint32_t k, uint64_t *out);
/* src/seqio/seqio.go:120-133; returns number of bytes that would panic (>84) */
uint32_t oracle_revcomp(uint8_t *seq, uint8_t *qual, uint32_t len);

/* ---- T2: ekzhu/lshensemble v1.1.0 ---- */
void   oracle_optimal_kl(int max_k, int max_l, int x, int q, double t, int *opt_k, int *opt_l);
double oracle_containment(const uint64_t *q, const uint64_t *x, int s, int q_size, int x_size);
oracle_lshe *oracle_lshe_build(const uint64_t *sketches, uint32_t n_windows, uint32_t s,
                               uint32_t num_part, uint32_t max_k, uint32_t num_window_kmers);
void   oracle_lshe_free(oracle_lshe *);
/* src/lshe/lshe.go:153-175: hits after the containment re-check, ascending window id.
 * returns the number of hits (may exceed cap; only cap are written) */
uint32_t oracle_lshe_query(const oracle_lshe *, const uint64_t *sig, int query_size, double threshold,
                           uint32_t *out, uint32_t cap);

/* ---- src/graph/alignment.go:13-159 AlignRead for one (already oriented) read against one seed window
 * (the call alignment_test.go:70-94 makes).  Returns the number of records; at most cap are written. */
uint32_t oracle_align_read(const oracle_index *idx, const uint8_t *read, uint32_t len, int rc, uint32_t window,
                           oracle_aln *out, uint32_t cap);

/* ---- whole path: boss.go:108-242 + graphminion.go:46-102 + alignment.go:13-317 ---- */
oracle_run *oracle_run_new(const oracle_index *idx, double containment_threshold, int no_exact_align);
void oracle_run_free(oracle_run *);
/* process a batch of reads (seq_off has n+1 entries into seq); appends to the run's outputs.
 * returns 0, or -1 on a read shorter than k (reference: panic, boss.go:164-166) */
int oracle_run_batch(oracle_run *, const uint8_t *seq, const uint64_t *seq_off, uint32_t n_reads,
                     uint32_t first_read_id);
void oracle_run_counts(const oracle_run *, oracle_counts *);
/* streaming use (bench.py's cpu_baseline): drop the seeds / records / sketches collected so far; counters, call counts stay */
void oracle_run_drop_records(oracle_run *);
uint64_t oracle_run_seeds(const oracle_run *, const oracle_seed **out);
uint64_t oracle_run_alns(const oracle_run *, const oracle_aln **out);
uint64_t oracle_run_sketches(const oracle_run *, const uint64_t **out); /* n_reads*s */
/* per-(kmerCount, window) IncrementSubPath call counts: out[q*n_windows+w], q in [0,max_q] */
uint32_t oracle_run_attempts(const oracle_run *, const uint32_t **out); /* returns max_q+1 */
/* graph weights (graph.go:401-451).  order=0: reference order with one sketching minion (reads
 * in input order); order=1: canonical replay (window asc, kmerCount asc, repeated adds). */
void oracle_run_weights(const oracle_run *, int order, double *node_kmer_freq, uint64_t *graph_kmer_total);
/* graph.go:455-525: per graph keep flag, per path kept flag, per node removed flag */
void oracle_prune(const oracle_index *idx, const double *node_kmer_freq, double min_kmer_cov,
                  uint8_t *graph_kept, uint8_t *path_kept, uint8_t *node_removed);

#ifdef __cplusplus
}
#endif
#endif
