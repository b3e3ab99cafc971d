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
 * A Go host drains its read channel into batches, calls submit/wait/read_*, turns the returned
 * traversal records into sam.Records (it keeps ID/Seq/Qual) and, after the last batch, pulls the
 * call counts and replays IncrementSubPath (groot_host_weights).  INTEGRATION.md has the cgo stub.
 *
 * Plain C: pointers and sizes only.  One ctx per GPU; a ctx is used from one host thread at a
 * time, different ctxs may be used concurrently.  Every call returns 0 or a negative GROOT_E_*
 * (include/groot_host.h); groot_hip_last_error(ctx) has the text.  There is no CPU fallback: without
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
    uint32_t reserved;
} groot_params;

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
} groot_counts;

/* per-stage device time of the last batch, HIP events on the ctx stream (ms); 0 if profiling off.
 * sketch_seed = the sketch+seed kernel alone; schedule = radix sort of the processing order + record gather;
 * align = the align kernel; sort = ordering of the traversal records into (read, ord) order */
typedef struct groot_stage_ms {
    float h2d, sketch_seed, align, sort, total, schedule;
} groot_stage_ms;

int groot_hip_device_count(int *n);
const char *groot_hip_last_error(const groot_ctx *ctx); /* ctx may be NULL: error of a failed open */

/* Uploads (replicates) the index into this GPU's HBM and builds the device lookup structures
 * (what ContainmentIndex.Load / BootstrapLshEnsembleEquiDepth do, lshe.go:95-147). */
int groot_hip_open(groot_ctx **out, int device_id, const groot_index_view *idx, const groot_params *p);
void groot_hip_close(groot_ctx *ctx);

/* Run work on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the ctx's own. */
int groot_hip_set_stream(groot_ctx *ctx, void *hip_stream);
int groot_hip_set_profiling(groot_ctx *ctx, int enable);

/* Submit one batch of reads held in host memory: seq_concat = read.Seq bytes back to back,
 * seq_off[i]..seq_off[i+1] = read i (n_reads+1 entries).  Copies H2D and launches; asynchronous. */
int groot_hip_submit(groot_ctx *ctx, const uint8_t *seq_concat, const uint64_t *seq_off, uint32_t n_reads,
                     uint32_t first_read_id);
/* Same batch, bases packed 2 bits each: base b of seq_concat sits in bits 2*(b%4).. of packed[b/4] as
 * (byte >> 1) & 3 (A=0 C=1 T=2 G=3); every byte that is not one of ACGT is listed in exc_pos (its index in
 * seq_concat) / exc_byte and may carry any code in `packed`.  A quarter of the PCIe traffic of groot_hip_submit for
 * the same result (the device unpacks to the byte layout first); groot_host_pack_reads builds the arguments. */
int groot_hip_submit_packed(groot_ctx *ctx, const uint8_t *packed, const uint64_t *seq_off, uint32_t n_reads,
                            uint32_t first_read_id, const uint64_t *exc_pos, const uint8_t *exc_byte, uint64_t n_exc);
/* Same, inputs already resident in HBM.  d_seq must be 16-byte aligned and readable for 16 bytes past
 * the last base (the kernels load 16-byte / 8-byte words); max_len = longest read of the batch
 * (0 = params.max_read_len). */
int groot_hip_submit_device(groot_ctx *ctx, const void *d_seq, const void *d_seq_off, uint32_t n_reads,
                            uint32_t first_read_id, uint32_t max_len);
/* Blocks until the submitted batch is finished; counts are for that batch. */
int groot_hip_wait(groot_ctx *ctx, groot_counts *counts);

/* Results of the finished batch.  *n = number available; at most cap are written. */
int groot_hip_read_seeds(groot_ctx *ctx, groot_seed *out, uint64_t cap, uint64_t *n);
int groot_hip_read_travs(groot_ctx *ctx, groot_trav *out, uint64_t *masks /* [cap*path_words] */, uint64_t cap,
                         uint64_t *n);
int groot_hip_read_sketches(groot_ctx *ctx, uint64_t *out /* [cap_reads*sketch_size] */, uint64_t cap_reads,
                            uint64_t *n_reads);
int groot_hip_stage_ms(groot_ctx *ctx, groot_stage_ms *out);

/* IncrementSubPath call counts accumulated over every batch since open/reset:
 * counts[q * n_windows + w], q = kmerCount of the read in [0, n_q).  The device pointer variant lets
 * a host reduce them across GPUs (RCCL all-reduce) without a round trip. */
int groot_hip_attempts_shape(groot_ctx *ctx, uint32_t *n_q, uint32_t *n_windows);
int groot_hip_attempts_device(groot_ctx *ctx, void **d_counts_u32, uint64_t *n_elems);
int groot_hip_attempts_read(groot_ctx *ctx, uint32_t *out, uint64_t n_elems);
int groot_hip_attempts_reset(groot_ctx *ctx);
/* Accumulate into a caller-owned device buffer of n_q*n_windows uint32 (e.g. a torch tensor that is
 * all-reduced over RCCL afterwards) instead of the ctx's own; NULL = back to the ctx's buffer. */
int groot_hip_attempts_bind(groot_ctx *ctx, void *d_counts_u32, uint64_t n_elems);

/* Fine-grained mirror of Sequence.RunMinHash(k, s, false, nil) (seqio.go:40-68) for a batch of
 * sequences in host memory: out[i*s .. (i+1)*s) = KHF sketch of sequence i. */
int groot_hip_sketch(groot_ctx *ctx, const uint8_t *seq_concat, const uint64_t *seq_off, uint32_t n, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
