/*
 * groot_hip.h -- C ABI of libgroot_hip.so: the MI355X (gfx950) device path of `groot align`.
 *
 * The reference has no FFI seam; the drop-in boundary is the body of
 *     func (b *theBoss) mapReads() error                      src/pipeline/boss.go:108-242
 * i.e. everything between "reads arrive on a channel" and "sam.Records + weighted graphs":
 *     Sequence.RunMinHash          src/seqio/seqio.go:40-68  (+ src/minhash/khf.go:35-55, nthash)
 *     ContainmentIndex.Query       src/lshe/lshe.go:153-175  (+ lshensemble Query/Containment)
 *     graphMinion loop             src/pipeline/graphminion.go:46-102
 *     GrootGraph.IncrementSubPath  src/graph/graph.go:401-451   (as exact integer call counts)
 *     GrootGraph.AlignRead         src/graph/alignment.go:13-317
 * The reference streams reads continuously through that loop (boss.go:145-203).  Here a host drains its read channel
 * into batches and keeps SEVERAL of them in flight in one ctx: groot_hip_submit* enqueues a batch (copy to pinned
 * staging -> H2D -> kernels -> D2H, each on its own HIP stream), groot_hip_collect blocks for the OLDEST batch and
 * hands back its traversal records, which the host turns into sam.Records (it keeps ID/Seq/Qual).  After the last batch
 * the host pulls the IncrementSubPath call counts (summed over GPUs by groot_hip_attempts_allreduce when there are
 * several) and replays the float formula (groot_host_weights_rows).  cgo/ has the Go binding, INTEGRATION.md the patch.
 *
 * Plain C: pointers and sizes only.  One ctx per GPU; a ctx is used from one host thread at a time, different ctxs
 * may be used concurrently.  No call retains a caller pointer after it returns (cgo rule), except buffers the ctx itself
 * handed out (groot_hip_acquire) and the caller-owned table of groot_hip_attempts_layout.  Every call returns 0 or a
 * negative GROOT_E_* (include/groot_host.h); groot_hip_last_error(ctx) has the text.  There is no CPU fallback: without
 * a usable HIP device groot_hip_open fails with GROOT_E_DEVICE.
 */
#ifndef GROOT_HIP_H
#define GROOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "groot_host.h"
#include "groot_index.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct groot_ctx groot_ctx;

typedef struct groot_params {
    double   containment_threshold; /* -t / --contThresh (cmd/align.go:47), default 0.99            */
    uint32_t no_exact_align;        /* --noAlign (cmd/align.go:46)                                   */
    uint32_t max_read_len;          /* longest read accepted (sizes LDS staging + DFS stacks); 0=256 */
    uint32_t max_batch_reads;       /* capacity of one submit; 0 = 1<<20                             */
    uint32_t max_seeds_per_read;    /* initial per-read seed slots (grown automatically); 0 = 8      */
    uint64_t max_batch_bases;       /* 0 = max_batch_reads * max_read_len                            */
    uint32_t keep_sketches;         /* 1: keep every read's KHF sketch on the device (tests)         */
    uint32_t pipeline_depth;        /* batches that may be in flight (submitted, not yet released); 0 = 3 */
    uint32_t results_on_device;     /* 1: traversal records stay in HBM (groot_batch_result.d_*), no D2H unless
                                       groot_hip_read_travs asks; 0 (default): D2H into pinned host memory, overlapped
                                       with the next batch's kernels                                 */
    uint32_t memo_budget_mb;        /* the memo of groot_hip_open (every WindowSize-mer of every indexed path run through the ctx's
                                       own pipeline once; reads that equal one are answered from it): 0 = default budget
                                       (GROOT_MEMO_DEFAULT_MB of HBM, and about as much host memory while it is built),
                                       GROOT_MEMO_OFF = no memo, else the budget in MiB.  An index whose strings need more than
                                       the budget is opened without the memo (groot_open_stats.memo_strings == 0); a caller
                                       with little input should switch it off: it costs ~0.3 s per GB of path bases at open
                                       and pays for itself only after some hundred million reads       */
} groot_params;
#define GROOT_MEMO_OFF 0xFFFFFFFFu
#define GROOT_MEMO_DEFAULT_MB 8192u

void groot_params_default(groot_params *p);

/* one seed = one lshe.Key returned by ContainmentIndex.Query for a read (lshe.go:165-171) */
typedef struct groot_seed {
    uint32_t read_id;   /* first_read_id + position in the batch */
    uint32_t window_id; /* window index in the groot_index_view  */
} groot_seed;

/* groot_trav (one traversal + its path set) is declared in groot_host.h: the host expands it to records */

typedef struct groot_counts {
    uint64_t received;       /* theBoss.receivedReadCount */
    uint64_t mapped;         /* theBoss.mappedCount       */
    uint64_t multimapped;    /* theBoss.multimappedCount  */
    uint64_t alignments;     /* theBoss.alignmentCount (= number of sam.Records)                     */
    uint64_t seeds;          /* total lshe.Keys returned by Query                                    */
    uint64_t travs;          /* groot_trav records of the batch                                      */
    uint64_t revcomp_panics; /* reads on which the reference panics in RevComplement (seqio.go:126)  */
    uint64_t short_reads;    /* reads shorter than k (reference panics, boss.go:164-166)             */
    uint64_t full_sketch_reads; /* reads whose seeds the full-width sketch kernel decided: all of them, or -- when the text
                                 * lookup / the signature kernel runs in front of it -- those it could not decide (diagnostic) */
    uint64_t walked_reads;      /* reads that went through the align stage's graph walk (the others: no seed window, or their
                                 * whole outcome came from the memo of groot_hip_open) (diagnostic) */
    uint64_t lean_reads;        /* ... of them, reads the align stage's first pass finished (one seed window, a walk that never
                                 * has two neighbours to choose from); the others took the state-machine kernel (diagnostic) */
} groot_counts;

/* per-stage device time of a batch, HIP events (ms); 0 if profiling off.
 * h2d = input copy on the copy-in stream; sketch_seed = the sketch+seed kernel alone; schedule = radix sort of the
 * processing order (or its stream compaction); align = the align kernel; sort = ordering of the traversal records into
 * (read, ord) order -- for reads answered from the outcome table this is where their records are written; total = first kernel .. last kernel; d2h = result copy on the copy-out stream */
typedef struct groot_stage_ms {
    float h2d, sketch_seed, align, sort, total, schedule, d2h, unpack;
    /* single kernels inside the stages above (for roofline figures): first_seed_kernel = the kernel that sees every read of the
     * batch first (text lookup / signature kernel / full-width kernel), inside sketch_seed; order_kernel = order_first_kernel,
     * inside sort */
    float first_seed_kernel, order_kernel;
    /* list_pass = what runs behind the first seed kernel inside sketch_seed: the full-width hashing kernel on the reads the first
     * kernel could not decide (LIST instance) + the wavefront-per-read LSH-Forest walk; 0 when the full-width kernel ran alone.
     * wall = first kernel of the batch .. last kernel of the batch on the wall clock of the device: the seed stage of the next
     * batch runs beside this batch's align stage, so wall < sketch_seed + schedule + align + sort of two neighbouring batches */
    float list_pass, wall;
    /* lean_pass = the first pass of the align stage (align_lean_kernel + the stream compaction of what it leaves), inside align; 0 when it did not run */
    float lean_pass;
} groot_stage_ms;

int groot_hip_device_count(int *n);
const char *groot_hip_last_error(const groot_ctx *ctx); /* ctx may be NULL: error of a failed open */

/* Uploads (replicates) the index into this GPU's HBM and builds the device lookup structures
 * (what ContainmentIndex.Load / BootstrapLshEnsembleEquiDepth do, lshe.go:95-147).  The view is checked first
 * (groot_index_view_check's pass): GROOT_E_FORMAT for one whose indices or offsets do not resolve. */
int groot_hip_open(groot_ctx **out, int device_id, const groot_index_view *idx, const groot_params *p);
/* The same with flags.  GROOT_OPEN_BACKGROUND: return as soon as the ctx can take batches (graphs, window arrays, exact and
 * LSH-Forest tables in HBM) and build the rest -- per-window prefix tables, the signature index: 0.6 of the 0.75 s an open without
 * memo takes on arg-annot.90 -- on a thread of the ctx while the first batches run through the full-width kernels; results are the
 * same whichever kernels a batch met.  The one exception to "no call retains a caller pointer": `idx` must stay valid until
 * groot_hip_open_wait (or groot_hip_close) has returned.  Ignored when the memo is wanted (it needs everything at once).
 * groot_hip_open_wait blocks until the background part is in place (0 at once if there is none); its failure is reported there
 * or by the next submit.  Replaces nothing in the reference: `groot align` rebuilds its LSH forests before the first read
 * (cmd/align.go:93-111). */
#define GROOT_OPEN_BACKGROUND 1u
int groot_hip_open_flags(groot_ctx **out, int device_id, const groot_index_view *idx, const groot_params *p, uint32_t flags);
int groot_hip_open_wait(groot_ctx *ctx);
/* Tell a background open to stop at its next checkpoint (a tenth of a second at most): for a caller whose input has ended before
 * the tables were there -- they would only be built to be freed (groot_hip_close asks the same before it joins the thread).  The ctx
 * keeps working, through the full-width kernels.  No-op without a background build in progress. */
int groot_hip_open_abandon(groot_ctx *ctx);
void groot_hip_close(groot_ctx *ctx);

/* What groot_hip_open built besides the uploaded index (diagnostic; times in ms, host wall clock).  The memo: every
 * WindowSize-mer of every indexed path went through the ctx's own pipeline once; reads that equal one of those strings are answered
 * from it (DESIGN.md "memo"). */
typedef struct groot_open_stats {
    double open_ms;            /* the whole of groot_hip_open                                            */
    double memo_ms;            /* of which: enumerating the strings, running the pipeline on them, tables */
    uint64_t memo_strings;     /* distinct path strings                                                  */
    uint64_t memo_tabulated;   /* of them with a stored outcome                                          */
    uint64_t memo_entries;     /* outcome-table entries (one per traversal, at least one per string)      */
    uint64_t text_entries;     /* strings in the text table (0: the text lookup is not used)              */
    uint64_t memo_hbm_bytes;   /* HBM held by outcome table + text table + sig_info                       */
} groot_open_stats;
int groot_hip_open_stats(const groot_ctx *ctx, groot_open_stats *out);

/* The SEED stage of every batch (decode, hashing, containment look-up, processing order) is enqueued on a caller-owned
 * hipStream_t (e.g. torch's current stream), so that it starts behind whatever the caller enqueued there (the kernels that
 * produced an IN_DEVICE batch); NULL = the ctx's own.  Only while nothing is in flight.
 * The align and order stages, and everything behind them, run on a stream PRIVATE to the ctx beside the next batch's seed
 * stage (boss.go:134-203: the reference's sketching and graph minions work side by side too).  Work the caller enqueues on
 * its stream after a submit is therefore NOT ordered behind the batch's results, and the inputs of groot_hip_submit_device
 * are still being read when the caller's stream has drained.  Three ways to order against a batch: groot_hip_wait /
 * groot_hip_collect (host side), or groot_hip_stream_join (device side, no host wait). */
int groot_hip_set_stream(groot_ctx *ctx, void *hip_stream);
/* Make `hip_stream` (NULL = the stream given to groot_hip_set_stream) wait -- on the device, the call returns at once --
 * until every batch submitted so far is through its kernels: records, path sets, counters and call counts of its FIRST PASS
 * are in place and that pass no longer reads the batch's inputs.  For callers that consume results_on_device buffers or
 * recycle groot_hip_submit_device inputs from kernels of their own.
 * The first pass is the only one unless a growable buffer of the ctx was too small for the batch (more seeds per read,
 * traversals, overflow records or kmerCount rows than it had room for): then groot_hip_wait / groot_hip_collect grow the
 * buffer on the HOST and run the batch again -- reading its inputs again, rewriting (possibly reallocating) its results.  A
 * consumer that is ordered by the join alone must therefore look at the batch's status word first (groot_hip_redo_status):
 * non-zero = this batch will be redone at collect, its device results are not final and its inputs are still needed. */
int groot_hip_stream_join(groot_ctx *ctx, void *hip_stream);
/* *d_status = device address of the newest submitted batch's status word (valid until that batch is released), *redo_mask = the bits of it
 * that mean "a buffer overflowed: collect will redo the batch".  A kernel (or a 4-byte copy) ordered behind groot_hip_stream_join reads
 * `*d_status & redo_mask`: zero = the results the join made visible are final.  GROOT_E_STATE when nothing has been submitted. */
int groot_hip_redo_status(groot_ctx *ctx, const uint32_t **d_status, uint32_t *redo_mask);
int groot_hip_set_profiling(groot_ctx *ctx, int enable);

/* ---- submitting batches ------------------------------------------------------------------------------------------
 * Every submit takes a free pipeline slot (GROOT_E_STATE "pipeline full" when pipeline_depth batches are submitted and
 * not yet released: collect + release first), copies the caller's buffers into the slot's pinned staging, enqueues
 * H2D -> kernels -> D2H on the ctx's three streams and returns at once.  Batches complete in submission order.
 * first_read_id only labels the records (groot_trav.read_id = first_read_id + position in the batch); a host that
 * streams more than 2^32 reads passes 0 and keeps its own 64-bit base.                                              */

/* read.Seq bytes back to back, seq_off[i]..seq_off[i+1] = read i (n_reads+1 entries). */
int groot_hip_submit(groot_ctx *ctx, const uint8_t *seq_concat, const uint64_t *seq_off, uint32_t n_reads,
                     uint32_t first_read_id);
/* Same batch, bases packed 2 bits each: base b of seq_concat sits in bits 2*(b%4).. of packed[b/4] as
 * (byte >> 1) & 3 (A=0 C=1 T=2 G=3); every byte that is not one of ACGT is listed in exc_pos (its index in
 * seq_concat) / exc_byte and may carry any code in `packed`.  A quarter of the PCIe traffic of groot_hip_submit for
 * the same result (the device unpacks to the byte layout first); groot_host_pack_reads builds the arguments. */
int groot_hip_submit_packed(groot_ctx *ctx, const uint8_t *packed, const uint64_t *seq_off, uint32_t n_reads,
                            uint32_t first_read_id, const uint64_t *exc_pos, const uint8_t *exc_byte, uint64_t n_exc);
/* The wire format proper: packed bases + one u16 LENGTH per read (the device scans them into offsets) + exceptions:
 * 27 bytes per 100 bp read over PCIe instead of 108. */
int groot_hip_submit_packed16(groot_ctx *ctx, const uint8_t *packed, const uint16_t *seq_len, uint32_t n_reads,
                              uint32_t first_read_id, const uint64_t *exc_pos, const uint8_t *exc_byte, uint64_t n_exc);
/* Zero-copy producer side of the same format: acquire hands out the pinned staging of a free slot, the caller (a FASTQ
 * parser) packs straight into it, submit_acquired enqueues it.  Capacities: packed (max_batch_bases+3)/4 bytes,
 * seq_len max_batch_reads entries, exceptions exc_cap entries (a batch with more goes through
 * groot_hip_submit_packed16, which grows the staging). */
typedef struct groot_batch_buffers {
    uint64_t ticket;       /* pass back to groot_hip_submit_acquired / groot_hip_release */
    uint8_t *packed;
    uint16_t *seq_len;
    uint64_t *exc_pos;
    uint8_t *exc_byte;
    uint64_t packed_cap, exc_cap;
    uint32_t reads_cap, reserved;
} groot_batch_buffers;
int groot_hip_acquire(groot_ctx *ctx, groot_batch_buffers *out);
int groot_hip_submit_acquired(groot_ctx *ctx, uint64_t ticket, uint32_t n_reads, uint64_t n_exc, uint32_t first_read_id);
/* Inputs already resident in HBM (no staging, no H2D).  d_seq must be 16-byte aligned and readable for 16 bytes past
 * the last base (the kernels load 16-byte / 8-byte words); max_len = longest read of the batch
 * (0 = params.max_read_len), taken as THE read length of the batch unless GROOT_MAXLEN_MIXED is set.  The buffers must stay valid until the batch is collected. */
#define GROOT_MAXLEN_MIXED 0x80000000u /* OR into max_len: the reads of the batch differ in length (scheduling hint only) */
int groot_hip_submit_device(groot_ctx *ctx, const void *d_seq, const void *d_seq_off, uint32_t n_reads,
                            uint32_t first_read_id, uint32_t max_len);

/* ---- collecting results -------------------------------------------------------------------------------------------
 * groot_hip_collect blocks until the OLDEST submitted batch is finished (its D2H included) and describes it; the
 * pointers stay valid until groot_hip_release(ticket), which frees the slot for another submit.  Several collected
 * batches may be held at once (e.g. while BAM writer threads work on them).  status = what the reference does with
 * the batch: GROOT_E_SHORT_READ / GROOT_E_REVCOMP where it panics (results remain readable), GROOT_E_NOSPACE for a
 * read longer than max_read_len; collect itself returns that status. */
typedef struct groot_batch_result {
    uint64_t ticket;
    uint32_t first_read_id, n_reads;
    groot_counts counts;
    const groot_trav *travs;   /* [n_travs] in (read, ord) order; pinned host memory (NULL with results_on_device) */
    /* Path sets, COMPACT: traversal i owns max(1, ceil(paths of graph travs[i].graph_id / 8)) consecutive BYTES (path p = bit
     * p % 8 of byte p / 8), traversal after traversal; mask_ckpt[j] = offset of the first byte of traversal 256*j.  (1 to 2
     * instead of 24 bytes per traversal over PCIe on arg-annot.90.)  groot_host_unpack_masks widens them to path_words words each. */
    const uint8_t *masks;
    const uint32_t *mask_ckpt;
    uint64_t n_mask_bytes;
    uint64_t n_travs;
    const void *d_travs;       /* in HBM: the records and their path sets at path_words words per traversal */
    const void *d_masks;
    uint32_t path_words;
    int32_t status;
    groot_stage_ms ms;
} groot_batch_result;
int groot_hip_collect(groot_ctx *ctx, groot_batch_result *out);
int groot_hip_release(groot_ctx *ctx, uint64_t ticket);
int groot_hip_in_flight(groot_ctx *ctx, uint32_t *submitted_not_collected, uint32_t *free_slots);

/* One-batch-at-a-time convenience over the same machinery (tests, simple hosts): wait = collect, the batch is released
 * by the next submit / wait; read_* copy out of it. */
int groot_hip_wait(groot_ctx *ctx, groot_counts *counts);
int groot_hip_read_travs(groot_ctx *ctx, groot_trav *out, uint64_t *masks /* [cap*path_words] */, uint64_t cap,
                         uint64_t *n);
/* seeds / sketches live in one of the ctx's two work sets (batches take them in turn): readable until the second batch
 * submitted after the waited one has started, i.e. always for the newest batch and the one before it (GROOT_E_STATE otherwise) */
int groot_hip_read_seeds(groot_ctx *ctx, groot_seed *out, uint64_t cap, uint64_t *n);
int groot_hip_read_sketches(groot_ctx *ctx, uint64_t *out /* [cap_reads*sketch_size] */, uint64_t cap_reads,
                            uint64_t *n_reads);
int groot_hip_stage_ms(groot_ctx *ctx, groot_stage_ms *out);

/* ---- IncrementSubPath call counts ---------------------------------------------------------------------------------
 * Accumulated over every batch since open/reset in a table [rows][n_windows] of uint32, one row per kmerCount
 * (len-k+1, graphminion.go:60) that occurred among seeded reads -- 1.3 MB per distinct read length on arg-annot.90,
 * not (max_read_len-k+2) rows.  Rows are created on the device as kmerCounts show up.                               */
/* rows in ascending kmerCount order: q_values[i] and counts[i*n_windows ..]; *n_rows = rows available, at most
 * cap_rows are written.  Waits for everything in flight. */
int groot_hip_attempts_export(groot_ctx *ctx, uint32_t *q_values, uint32_t *counts, uint32_t cap_rows, uint32_t *n_rows,
                              uint32_t *n_windows);
/* Fixes the row layout: row i holds kmerCount q_values[i] (strictly ascending; must include every kmerCount that has
 * counts).  d_table NULL: ctx-owned storage (still grows if a new kmerCount appears).  d_table != NULL: the table
 * lives in that caller-owned device buffer of n_q*n_windows uint32 (e.g. a torch tensor that is all-reduced over RCCL
 * afterwards); existing counts are copied into it; a kmerCount outside the layout then fails the batch with
 * GROOT_E_NOSPACE.  Only while nothing is in flight. */
int groot_hip_attempts_layout(groot_ctx *ctx, const uint32_t *q_values, uint32_t n_q, void *d_table);
/* Adds exported rows back (same shapes as groot_hip_attempts_export): a host that re-opens its ctx, e.g. for longer
 * reads, carries the counts over.  kmerCounts the table lacks get rows.  Only while nothing is in flight. */
int groot_hip_attempts_import(groot_ctx *ctx, const uint32_t *q_values, const uint32_t *counts, uint32_t n_rows);
int groot_hip_attempts_device(groot_ctx *ctx, void **d_table, uint32_t *n_rows, uint32_t *n_windows);
int groot_hip_attempts_reset(groot_ctx *ctx);
/* SURVEY 8e / north_star: the one exchange of a multi-GPU run.  Brings every ctx to the union row layout and sums the
 * tables in place over RCCL (ncclAllReduce, one communicator over the ctxs' devices, xGMI on an MI355X node); ctxs that
 * share a device are summed by a kernel instead.  Afterwards every ctx holds the totals.  All ctxs must be idle. */
int groot_hip_attempts_allreduce(groot_ctx *const *ctxs, int n_ctx);
/* dense compatibility view: counts[q * n_windows + w], q in [0, max_read_len-k+2) */
int groot_hip_attempts_shape(groot_ctx *ctx, uint32_t *n_q, uint32_t *n_windows);
int groot_hip_attempts_read(groot_ctx *ctx, uint32_t *out, uint64_t n_elems);

/* Fine-grained mirror of Sequence.RunMinHash(k, s, false, nil) (seqio.go:40-68) for a batch of
 * sequences in host memory: out[i*s .. (i+1)*s) = KHF sketch of sequence i.  Only while nothing is in flight. */
int groot_hip_sketch(groot_ctx *ctx, const uint8_t *seq_concat, const uint64_t *seq_off, uint32_t n, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
