/* ctest.c -- the call sequence of package groothip (cgo/groothip/groothip.go), replayed in C.
 *
 * cgo/ has never met a Go compiler; what a Go host does at the C ABI can still be run: this harness makes exactly the calls
 * groothip makes, in the same order, with buffers laid out the way its Batch builds them:
 *   LoadGob   groot_index_load_gob + groot_index_get_view
 *   Open      groot_params_default + groot_hip_open            (one ctx per GPU asked for)
 *   Submit    groot_hip_submit_packed16 on caller memory that is scribbled over right after the call returns (the cgo rule:
 *             nothing may be referenced afterwards), up to PipelineDepth batches in flight, GROOT_E_STATE = ErrFull
 *   Collect   groot_hip_collect + groot_host_unpack_masks, Release groot_hip_release
 *   Reopen    groot_hip_attempts_export -> groot_hip_open -> groot_hip_attempts_import -> groot_hip_close (after the first batch)
 *   Weights   groot_hip_attempts_allreduce + groot_hip_attempts_export + groot_host_weights_rows
 * It prints one JSON line with the counters, the number of records and checksums of the records and the weights; the -m gpu test
 * tests/test_pipeline.py::test_cgo_call_sequence compares it with the Python binding on the same reads.
 * Replaces: theBoss.mapReads, src/pipeline/boss.go:108-242 (through cgo/patch/boss_hip.go).
 *
 *   ctest <dir with groot.gg + groot.lshe> <reads.txt: one read per line> <batch reads> [n ctxs]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "groot_hip.h"

typedef struct {
    uint8_t *packed;
    uint16_t *lens;
    uint64_t *exc_pos;
    uint8_t *exc_byte;
    uint64_t n_bases, n_exc, cap_bases, cap_exc;
    uint32_t n, cap, max_len;
} batch;

static void batch_add(batch *b, const char *seq, size_t len)      /* Batch.Add */
{
    if (b->n == b->cap) {
        b->cap = b->cap ? 2 * b->cap : 1024;
        b->lens = realloc(b->lens, b->cap * sizeof *b->lens);
    }
    while ((b->n_bases + len + 3) / 4 + 1 > b->cap_bases) {
        const uint64_t nc = b->cap_bases ? 2 * b->cap_bases : 1 << 16;
        b->packed = realloc(b->packed, nc);
        memset(b->packed + b->cap_bases, 0, nc - b->cap_bases);
        b->cap_bases = nc;
    }
    for (size_t i = 0; i < len; i++) {
        const uint8_t ch = (uint8_t)seq[i];
        b->packed[b->n_bases >> 2] |= (uint8_t)(((ch >> 1) & 3) << (2 * (b->n_bases & 3)));
        if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') {
            if (b->n_exc == b->cap_exc) {
                b->cap_exc = b->cap_exc ? 2 * b->cap_exc : 256;
                b->exc_pos = realloc(b->exc_pos, b->cap_exc * sizeof *b->exc_pos);
                b->exc_byte = realloc(b->exc_byte, b->cap_exc);
            }
            b->exc_pos[b->n_exc] = b->n_bases;
            b->exc_byte[b->n_exc++] = ch;
        }
        b->n_bases++;
    }
    b->lens[b->n++] = (uint16_t)len;
    if (len > b->max_len) b->max_len = (uint32_t)len;
}

static void batch_reset(batch *b)
{
    if (b->packed) memset(b->packed, 0, b->cap_bases);
    b->n = 0; b->n_bases = 0; b->n_exc = 0; b->max_len = 0;
}

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)

static uint64_t mix(uint64_t h, uint64_t v) { h = (h ^ v) * 0x9E3779B97F4A7C15ULL; return h ^ (h >> 29); }

typedef struct { uint64_t received, mapped, multimapped, alignments, travs, records, rec_hash; } totals;

static groot_index_view view;

static void collect_one(groot_ctx *c, totals *t)                  /* Ctx.Collect + writeTraversal's loop + Ctx.Release */
{
    groot_batch_result r;
    if (groot_hip_collect(c, &r)) DIE("groot_hip_collect: %s", groot_hip_last_error(c));
    t->received += r.counts.received; t->mapped += r.counts.mapped; t->multimapped += r.counts.multimapped;
    t->alignments += r.counts.alignments; t->travs += r.n_travs;
    if (r.n_travs) {
        uint64_t *masks = calloc(r.n_travs * r.path_words, sizeof *masks);
        if (groot_host_unpack_masks(&view, r.travs, r.n_travs, r.masks, masks)) DIE("groot_host_unpack_masks: %s", groot_host_last_error());
        for (uint64_t i = 0; i < r.n_travs; i++) {
            const groot_trav *tr = &r.travs[i];
            for (uint32_t w = 0; w < r.path_words; w++)
                for (uint64_t word = masks[i * r.path_words + w]; word; word &= word - 1) {
                    const uint32_t id = 64 * w + (uint32_t)__builtin_ctzll(word);      /* one sam.Record per path id */
                    t->rec_hash = mix(mix(mix(mix(mix(t->rec_hash, tr->read_id), tr->graph_id), id), tr->node), ((uint64_t)tr->offset << 8) | tr->flags);
                    t->records++;
                }
        }
        free(masks);
    }
    if (groot_hip_release(c, r.ticket)) DIE("groot_hip_release: %s", groot_hip_last_error(c));
}

static groot_ctx *open_ctx(int device, uint32_t max_read_len, uint32_t batch_reads)
{
    groot_params prm;
    groot_params_default(&prm);
    prm.containment_threshold = 0.99;
    prm.max_read_len = max_read_len;
    prm.max_batch_reads = batch_reads;
    prm.pipeline_depth = 3;
    groot_ctx *c = NULL;
    if (groot_hip_open(&c, device, &view, &prm)) DIE("groot_hip_open: %s", groot_hip_last_error(NULL));
    return c;
}

static groot_ctx *reopen(groot_ctx *c, int device, uint32_t max_read_len, uint32_t batch_reads)      /* Ctx.Reopen */
{
    uint32_t n_rows = 0, n_win = 0;
    if (groot_hip_attempts_export(c, NULL, NULL, 0, &n_rows, &n_win)) DIE("groot_hip_attempts_export: %s", groot_hip_last_error(c));
    uint32_t *q = calloc(n_rows + 1, sizeof *q), *counts = calloc((size_t)n_rows * n_win + 1, sizeof *counts);
    if (groot_hip_attempts_export(c, q, counts, n_rows, &n_rows, &n_win)) DIE("groot_hip_attempts_export: %s", groot_hip_last_error(c));
    groot_ctx *bigger = open_ctx(device, max_read_len, batch_reads);
    if (groot_hip_attempts_import(bigger, q, counts, n_rows)) DIE("groot_hip_attempts_import: %s", groot_hip_last_error(bigger));
    groot_hip_close(c);
    free(q); free(counts);
    return bigger;
}

int main(int argc, char **argv)
{
    if (argc < 4) DIE("usage: ctest <gob dir> <reads.txt> <batch reads> [n ctxs]");
    const uint32_t batch_reads = (uint32_t)atoi(argv[3]);
    const int n_ctx = argc > 4 ? atoi(argv[4]) : 1;
    char gg[4096], lshe[4096];
    snprintf(gg, sizeof gg, "%s/groot.gg", argv[1]);
    snprintf(lshe, sizeof lshe, "%s/groot.lshe", argv[1]);
    groot_index *idx = NULL;                                       /* LoadGob */
    if (groot_index_load_gob(gg, lshe, &idx)) DIE("groot_index_load_gob: %s", groot_host_last_error());
    groot_index_get_view(idx, &view);
    int n_dev = 0;
    if (groot_hip_device_count(&n_dev) || n_dev == 0) DIE("no HIP device available (no CPU fallback)");
    groot_ctx *ctxs[8];
    uint32_t max_read_len = 128;
    for (int d = 0; d < n_ctx; d++) ctxs[d] = open_ctx(d % n_dev, max_read_len, batch_reads);
    int pending[8] = {0};
    totals t;
    memset(&t, 0, sizeof t);

    FILE *f = fopen(argv[2], "r");
    if (!f) DIE("cannot open %s", argv[2]);
    static char line[70000];
    batch b;
    memset(&b, 0, sizeof b);
    int next = 0, batches = 0, reopened = 0;
    for (;;) {
        const int have = fgets(line, sizeof line, f) != NULL;
        if (have) {
            size_t len = strlen(line);
            while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) len--;
            if (len) batch_add(&b, line, len);
        }
        if ((b.n == batch_reads || (!have && b.n)) ) {
            /* a read longer than the ctxs take: every ctx grows (boss_hip.go submit) */
            if (b.max_len > max_read_len) {
                while (max_read_len < b.max_len) max_read_len *= 2;
                for (int d = 0; d < n_ctx; d++) {
                    while (pending[d]) { collect_one(ctxs[d], &t); pending[d]--; }
                    ctxs[d] = reopen(ctxs[d], d % n_dev, max_read_len, batch_reads);
                }
                reopened++;
            }
            while (pending[next] >= 3) { collect_one(ctxs[next], &t); pending[next]--; }
            int rc = groot_hip_submit_packed16(ctxs[next], b.packed, b.lens, b.n, 0, b.n_exc ? b.exc_pos : NULL, b.n_exc ? b.exc_byte : NULL, b.n_exc);
            if (rc == GROOT_E_STATE) DIE("pipeline full although a slot was freed");
            if (rc) DIE("groot_hip_submit_packed16: %s", groot_hip_last_error(ctxs[next]));
            /* the cgo rule: the ctx keeps no pointer into the caller's buffers -- scribble over them at once */
            memset(b.packed, 0xFF, (size_t)((b.n_bases + 3) / 4));
            memset(b.lens, 0xFF, b.n * sizeof *b.lens);
            pending[next]++;
            batches++;
            next = (next + 1) % n_ctx;
            batch_reset(&b);
        }
        if (!have) break;
    }
    fclose(f);
    for (int d = 0; d < n_ctx; d++)
        while (pending[d]) { collect_one(ctxs[d], &t); pending[d]--; }

    /* Weights */
    if (groot_hip_attempts_allreduce(ctxs, n_ctx)) DIE("groot_hip_attempts_allreduce: %s", groot_hip_last_error(ctxs[0]));
    uint32_t n_rows = 0, n_win = 0;
    if (groot_hip_attempts_export(ctxs[0], NULL, NULL, 0, &n_rows, &n_win)) DIE("groot_hip_attempts_export: %s", groot_hip_last_error(ctxs[0]));
    uint32_t *q = calloc(n_rows + 1, sizeof *q), *counts = calloc((size_t)n_rows * n_win + 1, sizeof *counts);
    if (groot_hip_attempts_export(ctxs[0], q, counts, n_rows, &n_rows, &n_win)) DIE("groot_hip_attempts_export: %s", groot_hip_last_error(ctxs[0]));
    double *kf = calloc(view.n_nodes + 1, sizeof *kf);
    uint64_t *kt = calloc(view.n_graphs + 1, sizeof *kt);
    if (groot_host_weights_rows(&view, q, n_rows, counts, kf, kt)) DIE("groot_host_weights_rows: %s", groot_host_last_error());
    uint64_t wh = 0, kt_sum = 0;
    for (uint32_t i = 0; i < view.n_nodes; i++) { uint64_t bits; memcpy(&bits, &kf[i], 8); wh = mix(wh, bits); }
    for (uint32_t i = 0; i < view.n_graphs; i++) kt_sum += kt[i];
    printf("{\"batches\": %d, \"reopened\": %d, \"received\": %llu, \"mapped\": %llu, \"multimapped\": %llu, \"alignments\": %llu, \"travs\": %llu, "
           "\"records\": %llu, \"record_hash\": \"%016llx\", \"weights_hash\": \"%016llx\", \"kmer_total\": %llu, \"rows\": %u}\n",
           batches, reopened, (unsigned long long)t.received, (unsigned long long)t.mapped, (unsigned long long)t.multimapped,
           (unsigned long long)t.alignments, (unsigned long long)t.travs, (unsigned long long)t.records, (unsigned long long)t.rec_hash,
           (unsigned long long)wh, (unsigned long long)kt_sum, n_rows);
    for (int d = 0; d < n_ctx; d++) groot_hip_close(ctxs[d]);
    groot_index_free(idx);
    return 0;
}
