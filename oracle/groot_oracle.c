/*
 * groot_oracle.c -- CPU restatement (plain C11, single thread, deterministic) of the reference's
 * `groot align` hot path.  TEST INFRASTRUCTURE ONLY -- see groot_oracle.h for the rules and for
 * the "parity unpinned" statement about the two un-vendored Go modules.
 *
 * Citations are file:line under /root/reference (will-rowe/groot v1.1.2).
 */
#include "groot_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small growable vector helper                                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    void *p;
    size_t n, cap, esz;
} vec;

static void vec_init(vec *v, size_t esz) { v->p = NULL; v->n = 0; v->cap = 0; v->esz = esz; }
static void *vec_push(vec *v)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 64;
        v->p = realloc(v->p, v->cap * v->esz);
        if (!v->p) abort();
    }
    return (char *)v->p + (v->n++) * v->esz;
}
static void vec_free(vec *v) { free(v->p); v->p = NULL; v->n = v->cap = 0; }

/* ------------------------------------------------------------------------------------------ */
/* T1  github.com/will-rowe/nthash v0.2.0 (not in /root/reference; restated from the module)   */
/*     call sites: src/minhash/khf.go:38 (NewHasher) and khf.go:44 (MultiHash)                 */
/* ------------------------------------------------------------------------------------------ */
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL
#define MULTI_SEED 0x90b45d39fb6da1faULL
#define MULTI_SHIFT 27
#define CP_OFF 0x07 /* "offset" mask used to fetch a base's complement from the same table */

static uint64_t seed_tab[256];
static int seed_tab_ready = 0;

/* nthash.go seedTab: indexed by the raw byte; slots 0..7 double as the complement table
 * (A=0x41&7=1 ->T, C=0x43&7=3 ->G, T=0x54&7=4 ->A, U=0x55&7=5 ->A, G=0x47&7=7 ->C, N=0x4e&7=6 ->0) */
static void seed_tab_init(void)
{
    if (seed_tab_ready) return;
    memset(seed_tab, 0, sizeof seed_tab);
    seed_tab[1] = SEED_T; seed_tab[3] = SEED_G; seed_tab[4] = SEED_A; seed_tab[5] = SEED_A; seed_tab[7] = SEED_C;
    seed_tab['A'] = seed_tab['a'] = SEED_A;
    seed_tab['C'] = seed_tab['c'] = SEED_C;
    seed_tab['G'] = seed_tab['g'] = SEED_G;
    seed_tab['T'] = seed_tab['t'] = SEED_T;
    seed_tab['U'] = seed_tab['u'] = SEED_T;
    seed_tab_ready = 1;
}

static inline uint64_t rol64(uint64_t v, unsigned n) { n &= 63; return n ? (v << n) | (v >> (64 - n)) : v; }
static inline uint64_t ror64(uint64_t v, unsigned n) { n &= 63; return n ? (v >> n) | (v << (64 - n)) : v; }

typedef struct {
    const uint8_t *seq;
    uint32_t len, k, idx, max_idx;
    uint64_t fh, rh;
} nthi;

/* nthash.NewHasher: ntf64 / ntr64 over the first k-mer */
static int nthi_init(nthi *h, const uint8_t *seq, uint32_t len, uint32_t k)
{
    seed_tab_init();
    if (k == 0 || k > 64 || k > len) return -1;
    h->seq = seq; h->len = len; h->k = k; h->idx = 0; h->max_idx = len - (k - 1);
    uint64_t fh = 0, rh = 0;
    for (uint32_t i = 0; i < k; i++) { fh = rol64(fh, 1); fh ^= seed_tab[seq[i]]; }
    for (uint32_t i = 0; i < k; i++) { rh = rol64(rh, 1); rh ^= seed_tab[seq[k - 1 - i] & CP_OFF]; }
    h->fh = fh; h->rh = rh;
    return 0;
}

/* nthash.(*NTHi).Next(canonical=true): roll (ntHash paper alg. 3), return min(fh, rh) */
static int nthi_next(nthi *h, uint64_t *out)
{
    if (h->idx >= h->max_idx) return 0;
    if (h->idx != 0) {
        uint8_t prev = h->seq[h->idx - 1];
        uint8_t end = h->seq[h->idx + h->k - 1];
        h->fh = rol64(h->fh, 1);
        h->fh ^= rol64(seed_tab[prev], h->k);
        h->fh ^= seed_tab[end];
        h->rh = ror64(h->rh, 1);
        h->rh ^= ror64(seed_tab[prev & CP_OFF], 1);
        h->rh ^= rol64(seed_tab[end & CP_OFF], h->k - 1);
    }
    h->idx++;
    *out = h->fh < h->rh ? h->fh : h->rh;
    return 1;
}

int oracle_nthash_canonical(const uint8_t *seq, uint32_t len, uint32_t k, uint64_t *out)
{
    nthi h;
    if (nthi_init(&h, seq, len, k)) return -1;
    uint64_t v; uint32_t n = 0;
    while (nthi_next(&h, &v)) out[n++] = v;
    return 0;
}

/* src/minhash/khf.go:18-32 (init MaxUint64), :35-55 (AddSequence), nthash MultiHash:
 *   h[0] = canonical hash; h[i] = t ^ (t >> 27), t = h[0] * (i ^ (k * multiSeed))   (wrapping u64) */
int oracle_khf_sketch(const uint8_t *seq, uint32_t len, uint32_t k, uint32_t s, uint64_t *sketch)
{
    nthi h;
    if (s == 0) return -1;
    if (nthi_init(&h, seq, len, k)) return -1;
    for (uint32_t i = 0; i < s; i++) sketch[i] = UINT64_MAX;
    uint64_t hv;
    while (nthi_next(&h, &hv)) {
        if (hv < sketch[0]) sketch[0] = hv;
        for (uint64_t i = 1; i < (uint64_t)s; i++) {
            uint64_t t = hv * (i ^ ((uint64_t)k * MULTI_SEED));
            t ^= t >> MULTI_SHIFT;
            if (t < sketch[i]) sketch[i] = t;
        }
    }
    return 0;
}

/* src/seqio/seqio.go:17-23 (complementBases, len 'T'+1) and :120-133 (RevComplement) */
uint32_t oracle_revcomp(uint8_t *seq, uint8_t *qual, uint32_t len)
{
    uint32_t panics = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint8_t b = seq[i], c;
        switch (b) {
        case 'A': c = 'T'; break;
        case 'T': c = 'A'; break;
        case 'C': c = 'G'; break;
        case 'G': c = 'C'; break;
        case 'N': c = 'N'; break;
        default:
            if (b > 'T') panics++; /* Go: index out of range -> panic */
            c = 0;
        }
        seq[i] = c;
    }
    if (len) {
        for (uint32_t i = 0, j = len - 1; i < j; i++, j--) {
            uint8_t t = seq[i]; seq[i] = seq[j]; seq[j] = t;
            if (qual) { t = qual[i]; qual[i] = qual[j]; qual[j] = t; }
        }
    }
    return panics;
}

/* ------------------------------------------------------------------------------------------ */
/* T2  github.com/ekzhu/lshensemble v1.1.0 (not in /root/reference; restated from the module)  */
/*     call sites: src/lshe/lshe.go:134-145 (bootstrap), :157 (Query), :165 (Containment)      */
/* ------------------------------------------------------------------------------------------ */
#define INTEGRATION_PRECISION 0.01

typedef struct { int x, q, l, k; } fpfn_ctx;

static double prob_inner(const fpfn_ctx *c, double t)
{
    return 1.0 - pow(1.0 - pow(t / (1.0 + (double)c->x / (double)c->q - t), (double)c->k), (double)c->l);
}
static double f_false_positive(const fpfn_ctx *c, double t) { return prob_inner(c, t); }
static double f_false_negative(const fpfn_ctx *c, double t) { return 1.0 - prob_inner(c, t); }

/* probability.go integral(): midpoint rectangles, x accumulates by += precision */
static double integral(double (*f)(const fpfn_ctx *, double), const fpfn_ctx *c, double a, double b, double precision)
{
    double area = 0.0;
    for (double x = a; x < b; x += precision) area += f(c, x + 0.5 * precision) * precision;
    return area;
}
static double prob_false_negative(int x, int q, int l, int k, double t, double precision)
{
    fpfn_ctx c = { x, q, l, k };
    double xq = (double)x / (double)q;
    if (xq >= 1.0) return integral(f_false_negative, &c, t, 1.0, precision);
    if (xq >= t) return integral(f_false_negative, &c, t, xq, precision);
    return 0.0;
}
static double prob_false_positive(int x, int q, int l, int k, double t, double precision)
{
    fpfn_ctx c = { x, q, l, k };
    double xq = (double)x / (double)q;
    if (xq >= 1.0) return integral(f_false_positive, &c, 0.0, t, precision);
    if (xq >= t) return integral(f_false_positive, &c, 0.0, t, precision);
    return 0.0;
}

/* lshforest.go (*LshForest).OptimalKL: l outer, k inner, strict '>' keeps the first minimum */
void oracle_optimal_kl(int max_k, int max_l, int x, int q, double t, int *opt_k, int *opt_l)
{
    double min_error = 1.7976931348623157e308; /* math.MaxFloat64 */
    *opt_k = 0; *opt_l = 0;
    for (int l = 1; l <= max_l; l++) {
        for (int k = 1; k <= max_k; k++) {
            double fp = prob_false_positive(x, q, l, k, t, INTEGRATION_PRECISION);
            double fn = prob_false_negative(x, q, l, k, t, INTEGRATION_PRECISION);
            double err = fn + fp;
            if (min_error > err) { min_error = err; *opt_k = k; *opt_l = l; }
        }
    }
}

/* lshensemble.Containment */
double oracle_containment(const uint64_t *q, const uint64_t *x, int s, int q_size, int x_size)
{
    if (q_size == 0 || x_size == 0) return 0.0;
    int eq = 0;
    for (int i = 0; i < s; i++) if (x[i] == q[i]) eq++;
    if (eq == 0) return 0.0;
    double jaccard = (double)eq / (double)s;
    return ((double)x_size / (double)q_size + 1.0) * jaccard / (1.0 + jaccard);
}

/* One LshForest32(k=maxK, l=numHash/maxK) per partition; bucket key of band i = the low 4 bytes
 * (little endian) of each of sig[i*k .. (i+1)*k), concatenated (hashKeyFuncGen(4)); per band a
 * table sorted by key bytes. */
typedef struct { uint32_t id; } band_item; /* key bytes live in a parallel array */

typedef struct {
    uint32_t n;          /* records in this partition */
    int lower, upper;    /* Partition{Lower,Upper} */
    uint8_t **keys;      /* [l] -> n*key_bytes, sorted */
    uint32_t **ids;      /* [l] -> n ids, same order */
} forest;

struct oracle_lshe {
    uint32_t s, max_k, l, num_part, n_windows, key_bytes;
    int num_window_kmers;
    const uint64_t *sketches;
    forest *parts;
    /* paramCache: (x, q, t) -> (k, l) */
    struct { int x, q; double t; int k, l; } cache[64];
    int n_cache;
};

static uint32_t g_key_bytes; /* qsort context */
static const uint8_t *g_keys;
static int cmp_band(const void *a, const void *b)
{
    uint32_t ia = *(const uint32_t *)a, ib = *(const uint32_t *)b;
    int c = memcmp(g_keys + (size_t)ia * g_key_bytes, g_keys + (size_t)ib * g_key_bytes, g_key_bytes);
    if (c) return c;
    return ia < ib ? -1 : ia > ib;
}

static void put_key(uint8_t *dst, const uint64_t *sig, uint32_t n)
{
    for (uint32_t j = 0; j < n; j++) {
        uint64_t v = sig[j];
        dst[4 * j + 0] = (uint8_t)(v);
        dst[4 * j + 1] = (uint8_t)(v >> 8);
        dst[4 * j + 2] = (uint8_t)(v >> 16);
        dst[4 * j + 3] = (uint8_t)(v >> 24);
    }
}

/* lshensemble.BootstrapLshEnsembleEquiDepth + bootstrapEquiDepth: records arrive in (Go map)
 * arbitrary order, every record has Size = NumWindowKmers (lshe.go:134-138); we deal them in
 * window-id order -- the union over partitions is order independent. */
oracle_lshe *oracle_lshe_build(const uint64_t *sketches, uint32_t n_windows, uint32_t s,
                               uint32_t num_part, uint32_t max_k, uint32_t num_window_kmers)
{
    if (!max_k || !num_part || s < max_k) return NULL;
    oracle_lshe *e = calloc(1, sizeof *e);
    e->s = s; e->max_k = max_k; e->l = s / max_k; e->num_part = num_part; e->n_windows = n_windows;
    e->key_bytes = 4 * max_k; e->num_window_kmers = (int)num_window_kmers; e->sketches = sketches;
    e->parts = calloc(num_part, sizeof(forest));
    uint32_t depth = n_windows / num_part;
    uint32_t *part_of = malloc(sizeof(uint32_t) * (n_windows ? n_windows : 1));
    uint32_t curr_depth = 0, curr_part = 0;
    for (uint32_t w = 0; w < n_windows; w++) {
        part_of[w] = curr_part;
        forest *f = &e->parts[curr_part];
        f->n++;
        curr_depth++;
        f->upper = (int)num_window_kmers;
        if (curr_depth == 1) f->lower = (int)num_window_kmers;
        if (curr_depth >= depth && curr_part < num_part - 1) { curr_part++; curr_depth = 0; }
    }
    for (uint32_t p = 0; p < num_part; p++) {
        forest *f = &e->parts[p];
        f->keys = calloc(e->l, sizeof(uint8_t *));
        f->ids = calloc(e->l, sizeof(uint32_t *));
        uint32_t *members = malloc(sizeof(uint32_t) * (f->n ? f->n : 1));
        uint32_t m = 0;
        for (uint32_t w = 0; w < n_windows; w++) if (part_of[w] == p) members[m++] = w;
        for (uint32_t b = 0; b < e->l; b++) {
            uint8_t *raw = malloc((size_t)(f->n ? f->n : 1) * e->key_bytes);
            for (uint32_t i = 0; i < f->n; i++)
                put_key(raw + (size_t)i * e->key_bytes, sketches + (size_t)members[i] * s + (size_t)b * max_k, max_k);
            uint32_t *order = malloc(sizeof(uint32_t) * (f->n ? f->n : 1));
            for (uint32_t i = 0; i < f->n; i++) order[i] = i;
            g_key_bytes = e->key_bytes; g_keys = raw;
            qsort(order, f->n, sizeof(uint32_t), cmp_band);
            f->keys[b] = malloc((size_t)(f->n ? f->n : 1) * e->key_bytes);
            f->ids[b] = malloc(sizeof(uint32_t) * (f->n ? f->n : 1));
            for (uint32_t i = 0; i < f->n; i++) {
                memcpy(f->keys[b] + (size_t)i * e->key_bytes, raw + (size_t)order[i] * e->key_bytes, e->key_bytes);
                f->ids[b][i] = members[order[i]];
            }
            free(raw); free(order);
        }
        free(members);
    }
    free(part_of);
    return e;
}

void oracle_lshe_free(oracle_lshe *e)
{
    if (!e) return;
    for (uint32_t p = 0; p < e->num_part; p++) {
        for (uint32_t b = 0; b < e->l; b++) { free(e->parts[p].keys[b]); free(e->parts[p].ids[b]); }
        free(e->parts[p].keys); free(e->parts[p].ids);
    }
    free(e->parts); free(e);
}

static void params_for(oracle_lshe *e, int x, int q, double t, int *k, int *l)
{
    for (int i = 0; i < e->n_cache; i++)
        if (e->cache[i].x == x && e->cache[i].q == q && e->cache[i].t == t) { *k = e->cache[i].k; *l = e->cache[i].l; return; }
    oracle_optimal_kl((int)e->max_k, (int)e->l, x, q, t, k, l);
    if (e->n_cache < 64) {
        e->cache[e->n_cache].x = x; e->cache[e->n_cache].q = q; e->cache[e->n_cache].t = t;
        e->cache[e->n_cache].k = *k; e->cache[e->n_cache].l = *l; e->n_cache++;
    }
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* (*LshEnsemble).Query -> per partition (*LshForest).Query(sig, K, L): binary search for the first
 * bucket whose key[:4K] >= query prefix, walk while equal, dedupe; then lshe.go:165 keeps a hit iff
 * Containment(sig, key.Sketch, querySize, NumWindowKmers) > threshold (strict). */
uint32_t oracle_lshe_query(const oracle_lshe *ce, const uint64_t *sig, int query_size, double threshold,
                           uint32_t *out, uint32_t cap)
{
    oracle_lshe *e = (oracle_lshe *)ce;
    vec cand; vec_init(&cand, sizeof(uint32_t));
    uint8_t hk[4 * 64];
    for (uint32_t p = 0; p < e->num_part; p++) {
        forest *f = &e->parts[p];
        int K, L;
        params_for(e, f->upper, query_size, threshold, &K, &L);
        size_t first_of_part = cand.n;
        uint32_t prefix = 4u * (uint32_t)K;
        for (int i = 0; i < L; i++) {
            put_key(hk, sig + (size_t)i * e->max_k, (uint32_t)K);
            const uint8_t *keys = f->keys[i];
            /* sort.Search: smallest index with key[:prefix] >= hk */
            uint32_t lo = 0, hi = f->n;
            while (lo < hi) {
                uint32_t mid = lo + (hi - lo) / 2;
                if (memcmp(keys + (size_t)mid * e->key_bytes, hk, prefix) >= 0) hi = mid; else lo = mid + 1;
            }
            for (uint32_t j = lo; j < f->n && memcmp(keys + (size_t)j * e->key_bytes, hk, prefix) == 0; j++) {
                uint32_t id = f->ids[i][j];
                int seen = 0;
                for (size_t c = first_of_part; c < cand.n; c++) if (((uint32_t *)cand.p)[c] == id) { seen = 1; break; }
                if (!seen) *(uint32_t *)vec_push(&cand) = id;
            }
        }
    }
    uint32_t n = 0;
    uint32_t *c = cand.p;
    qsort(c, cand.n, sizeof(uint32_t), cmp_u32);
    for (size_t i = 0; i < cand.n; i++) {
        if (oracle_containment(sig, e->sketches + (size_t)c[i] * e->s, (int)e->s, query_size, e->num_window_kmers) > threshold) {
            if (n < cap) out[n] = c[i];
            n++;
        }
    }
    vec_free(&cand);
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* graph alignment: src/graph/alignment.go                                                     */
/* ------------------------------------------------------------------------------------------ */
struct oracle_run {
    const oracle_index *idx;
    oracle_lshe *lshe;
    double threshold;
    int no_align;
    vec seeds, alns, sketches;
    uint32_t *attempts;  /* [(max_q+1) * n_windows] */
    uint32_t max_q;
    int have_attempts;
    oracle_counts counts;
    double *kf_direct;       /* reference-order node KmerFreq */
    uint64_t *kt_direct;     /* reference-order graph KmerTotal */
    /* scratch */
    vec trav_nodes;          /* concatenated node lists of successful traversals */
    vec trav_off;            /* start offsets into trav_nodes (n+1) */
    uint32_t *path_buf;      /* DFS path */
    uint32_t path_cap;
};

typedef struct {
    oracle_run *r;
    const uint8_t *read;
    int read_len;
} dfs_ctx;

/* alignment.go:196-254 dfsRecursive */
static int dfs_recursive(dfs_ctx *c, uint32_t node, int distance, uint32_t depth, int offset)
{
    const oracle_index *ix = c->r->idx;
    uint32_t s0 = ix->node_seq_off[node], s1 = ix->node_seq_off[node + 1];
    int node_len = (int)(s1 - s0);
    if (offset >= node_len) return 0;                                  /* :199-201 */
    for (int i = offset; i < node_len; i++) {                          /* :204 */
        uint8_t base = ix->bases[s0 + (uint32_t)i];
        if (distance == c->read_len) break;                            /* :207-209 */
        if (base == 'N') { distance++; continue; }                     /* :212-215 */
        if (base == c->read[distance]) distance++;                     /* :218-219 */
        else return 0;                                                 /* :221 */
    }
    if (depth >= c->r->path_cap) {
        c->r->path_cap = c->r->path_cap ? c->r->path_cap * 2 : 256;
        c->r->path_buf = realloc(c->r->path_buf, sizeof(uint32_t) * c->r->path_cap);
    }
    c->r->path_buf[depth] = node;                                       /* :226 */
    uint32_t e0 = ix->node_edge_off[node], e1 = ix->node_edge_off[node + 1];
    if (distance == c->read_len || e0 == e1) {                          /* :229-236 */
        for (uint32_t i = 0; i <= depth; i++) *(uint32_t *)vec_push(&c->r->trav_nodes) = c->r->path_buf[i];
        *(uint32_t *)vec_push(&c->r->trav_off) = (uint32_t)c->r->trav_nodes.n;
        return 1;
    }
    int aligned = 0;
    for (uint32_t e = e0; e < e1; e++)                                  /* :242-252 */
        if (dfs_recursive(c, ix->edges[e], distance, depth + 1, 0)) aligned = 1;
    return aligned;
}

typedef struct { uint32_t id; uint32_t pos; } id_pos;

/* alignment.go:162-193 performAlignment + :263-317 processTraversal.
 * Appends the assigned path IDs (with duplicates, traversal order; within a traversal ascending
 * path id = our canonical choice for Go's map order) to `ids`.  Returns their number. */
static uint32_t perform_alignment(oracle_run *r, uint32_t graph, uint32_t node, const uint8_t *read, int read_len,
                                  int offset, vec *ids)
{
    const oracle_index *ix = r->idx;
    r->trav_nodes.n = 0; r->trav_off.n = 0;
    *(uint32_t *)vec_push(&r->trav_off) = 0;
    dfs_ctx c = { r, read, read_len };
    dfs_recursive(&c, node, 0, 0, offset);
    uint32_t n_trav = (uint32_t)r->trav_off.n - 1;
    if (!n_trav) return 0;
    uint32_t n_paths = ix->graph_path_off[graph + 1] - ix->graph_path_off[graph];
    uint32_t *count = calloc(n_paths ? n_paths : 1, sizeof(uint32_t));
    uint32_t before = (uint32_t)ids->n;
    const uint32_t *tn = r->trav_nodes.p, *to = r->trav_off.p;
    for (uint32_t t = 0; t < n_trav; t++) {
        uint32_t path_length = to[t + 1] - to[t];
        memset(count, 0, sizeof(uint32_t) * n_paths);
        for (uint32_t i = 0; i < path_length; i++) {                    /* :275-299 */
            uint32_t nd = tn[to[t] + i];
            for (uint32_t j = ix->node_np_off[nd]; j < ix->node_np_off[nd + 1]; j++) count[ix->np_path[j]]++;
        }
        uint32_t n0 = tn[to[t]];
        for (uint32_t id = 0; id < n_paths; id++) {                     /* :301-307 */
            if (count[id] < path_length) continue;
            uint32_t pos = 0;
            for (uint32_t j = ix->node_np_off[n0]; j < ix->node_np_off[n0 + 1]; j++)
                if (ix->np_path[j] == id) { pos = ix->np_pos[j] + (uint32_t)offset; break; }   /* :296 */
            id_pos *ip = vec_push(ids);
            ip->id = id; ip->pos = pos;
        }
    }
    free(count);
    return (uint32_t)ids->n - before;
}

/* alignment.go:13-159 AlignRead.  read/qual are the (possibly reverse-complemented) read. */
static uint32_t align_read(oracle_run *r, uint32_t read_id, const uint8_t *read, int read_len, int rc, uint32_t w)
{
    const oracle_index *ix = r->idx;
    const int max_clip = 1;                                             /* :16 */
    uint32_t graph = ix->win_graph[w];
    uint32_t seed_node = ix->win_node[w];                               /* :19-25 */
    int orig_off = (int)ix->win_offset[w];
    int start_clipped = 0, end_clipped = 0;
    vec ids; vec_init(&ids, sizeof(id_pos));
    uint32_t n = 0;
    /* 1. exact alignment and seed offset shuffling (:34-45) */
    int off = orig_off;
    for (int sh = 0; sh <= (int)(ix->win_merge_span[w] + ix->window_size); sh++) {
        n = perform_alignment(r, graph, seed_node, read, read_len, off, &ids);
        if (n) break;
        off++;
    }
    /* 2. seed node shuffling (:47-70); Go map order -> ascending SegmentID (canonical) */
    if (!n) {
        for (uint32_t c = ix->win_cn_off[w]; c < ix->win_cn_off[w + 1] && !n; c++) {
            off = 0;
            for (int sh = 0; sh <= 10; sh++) {
                n = perform_alignment(r, graph, ix->cn_node[c], read, read_len, off, &ids);
                if (n) break;
                off++;
            }
        }
    }
    /* 3. hard clip the start (:72-85) -- seed node, original offset */
    if (!n) {
        const uint8_t *clipped = read; int clen = read_len;
        for (int i = 1; i <= max_clip; i++) {
            clipped += i; clen -= i;
            n = perform_alignment(r, graph, seed_node, clipped, clen, orig_off, &ids);
            start_clipped++;
            if (n) break;
        }
    }
    /* 4. hard clip the end (:87-103) */
    if (!n) {
        start_clipped = 0;
        int clen = read_len;
        for (int i = max_clip; i > 0; i--) {
            clen -= 1;
            n = perform_alignment(r, graph, seed_node, read, clen, orig_off, &ids);
            end_clipped++;
            if (n) break;
        }
    }
    if (!n) { vec_free(&ids); return 0; }                               /* :108-110 */
    const id_pos *ip = ids.p;
    for (uint32_t i = 0; i < n; i++) {                                  /* :114-156 */
        oracle_aln *a = vec_push(&r->alns);
        a->read_id = read_id; a->graph_id = graph; a->path_id = ip[i].id;
        a->ref_id = ix->graph_path_off[graph] + ip[i].id;
        a->pos = ip[i].pos;
        a->start_clip = (uint8_t)start_clipped; a->end_clip = (uint8_t)end_clipped;
        a->rc = (uint8_t)rc;
        a->secondary = (uint8_t)(n > 1 && i != 0);
    }
    vec_free(&ids);
    return n;
}

uint32_t oracle_align_read(const oracle_index *idx, const uint8_t *read, uint32_t len, int rc, uint32_t window,
                           oracle_aln *out, uint32_t cap)
{
    oracle_run *r = calloc(1, sizeof *r);
    r->idx = idx;
    vec_init(&r->seeds, sizeof(oracle_seed));
    vec_init(&r->alns, sizeof(oracle_aln));
    vec_init(&r->sketches, sizeof(uint64_t));
    vec_init(&r->trav_nodes, sizeof(uint32_t));
    vec_init(&r->trav_off, sizeof(uint32_t));
    uint32_t n = align_read(r, 0, read, (int)len, rc, window);
    for (uint32_t i = 0; i < n && i < cap; i++) out[i] = ((oracle_aln *)r->alns.p)[i];
    vec_free(&r->alns); vec_free(&r->trav_nodes); vec_free(&r->trav_off);
    free(r->path_buf);
    free(r);
    return n;
}

/* graph.go:401-451 IncrementSubPath (ContainedNodes iterated in ascending SegmentID) */
static void increment_sub_path(const oracle_index *ix, uint32_t w, double num_kmers, double *kf, uint64_t *kt)
{
    uint32_t c0 = ix->win_cn_off[w], c1 = ix->win_cn_off[w + 1];
    if (c1 - c0 == 1) {                                                 /* :409-422 */
        kf[ix->cn_node[c0]] += num_kmers;
        return;
    }
    double total = 0.0;
    for (uint32_t c = c0; c < c1; c++) {                                /* :427-434 */
        uint32_t nd = ix->cn_node[c];
        total += (double)(ix->node_seq_off[nd + 1] - ix->node_seq_off[nd]);
    }
    for (uint32_t c = c0; c < c1; c++) {                                /* :437-446 */
        uint32_t nd = ix->cn_node[c];
        double seg_len = (double)(ix->node_seq_off[nd + 1] - ix->node_seq_off[nd]);
        double share = ((seg_len / total) * num_kmers) * (double)ix->cn_count[c];
        kf[nd] += share;
    }
    kt[ix->win_graph[w]] += (uint64_t)num_kmers;                        /* :449 */
}

oracle_run *oracle_run_new(const oracle_index *idx, double containment_threshold, int no_exact_align)
{
    oracle_run *r = calloc(1, sizeof *r);
    r->idx = idx; r->threshold = containment_threshold; r->no_align = no_exact_align;
    r->lshe = oracle_lshe_build(idx->win_sketch, idx->n_windows, idx->sketch_size, idx->num_part, idx->max_k,
                                idx->num_window_kmers);
    vec_init(&r->seeds, sizeof(oracle_seed));
    vec_init(&r->alns, sizeof(oracle_aln));
    vec_init(&r->sketches, sizeof(uint64_t));
    vec_init(&r->trav_nodes, sizeof(uint32_t));
    vec_init(&r->trav_off, sizeof(uint32_t));
    r->kf_direct = calloc(idx->n_nodes ? idx->n_nodes : 1, sizeof(double));
    r->kt_direct = calloc(idx->n_graphs ? idx->n_graphs : 1, sizeof(uint64_t));
    return r;
}

void oracle_run_free(oracle_run *r)
{
    if (!r) return;
    oracle_lshe_free(r->lshe);
    vec_free(&r->seeds); vec_free(&r->alns); vec_free(&r->sketches);
    vec_free(&r->trav_nodes); vec_free(&r->trav_off);
    free(r->attempts); free(r->kf_direct); free(r->kt_direct); free(r->path_buf);
    free(r);
}

static void attempts_grow(oracle_run *r, uint32_t q)
{
    if (r->have_attempts && q <= r->max_q) return;
    uint32_t new_max = r->have_attempts ? r->max_q : 0;
    if (q > new_max) new_max = q;
    size_t nw = r->idx->n_windows;
    uint32_t *a = calloc((size_t)(new_max + 1) * (nw ? nw : 1), sizeof(uint32_t));
    if (r->have_attempts) memcpy(a, r->attempts, sizeof(uint32_t) * (size_t)(r->max_q + 1) * nw);
    free(r->attempts);
    r->attempts = a; r->max_q = new_max; r->have_attempts = 1;
}

typedef struct { uint32_t seg, off, w; } seed_key;
static int cmp_seed(const void *a, const void *b)
{
    const seed_key *x = a, *y = b;
    if (x->seg != y->seg) return x->seg < y->seg ? -1 : 1;   /* lshe.go:33 Keys.Less: Node only      */
    if (x->off != y->off) return x->off < y->off ? -1 : 1;   /* canonical tie-break      */
    return x->w < y->w ? -1 : x->w > y->w;
}

/* boss.go:145-202 (one sketching minion) + graphminion.go:46-102 (per-graph minion, inline) */
int oracle_run_batch(oracle_run *r, const uint8_t *seq, const uint64_t *seq_off, uint32_t n_reads, uint32_t first_read_id)
{
    const oracle_index *ix = r->idx;
    uint32_t s = ix->sketch_size, k = ix->kmer_size;
    uint64_t *sk = malloc(sizeof(uint64_t) * s);
    uint32_t hit_cap = 1024;
    uint32_t *hits = malloc(sizeof(uint32_t) * hit_cap);
    for (uint32_t i = 0; i < n_reads; i++) {
        uint32_t read_id = first_read_id + i;
        const uint8_t *rd = seq + seq_off[i];
        uint32_t len = (uint32_t)(seq_off[i + 1] - seq_off[i]);
        if (oracle_khf_sketch(rd, len, k, s, sk)) { free(sk); free(hits); return -1; }   /* boss.go:163-166 */
        for (uint32_t j = 0; j < s; j++) *(uint64_t *)vec_push(&r->sketches) = sk[j];
        int kmer_count = (int)len - (int)k + 1;                                            /* boss.go:169 */
        uint32_t nh = oracle_lshe_query(r->lshe, sk, kmer_count, r->threshold, hits, hit_cap);
        if (nh > hit_cap) {
            hit_cap = nh; hits = realloc(hits, sizeof(uint32_t) * hit_cap);
            nh = oracle_lshe_query(r->lshe, sk, kmer_count, r->threshold, hits, hit_cap);
        }
        r->counts.received++;
        r->counts.seeds += nh;
        for (uint32_t h = 0; h < nh; h++) {
            oracle_seed *sd = vec_push(&r->seeds);
            sd->read_id = read_id; sd->window_id = hits[h];
        }
        if (!nh) continue;
        /* group by graph (results map[uint32]Keys, boss.go:184); graphs handled in ascending id */
        uint32_t n_graph_groups = 0;
        seed_key *keys = malloc(sizeof(seed_key) * nh);
        uint8_t *fwd = malloc(len), *work = malloc(len);
        memcpy(fwd, rd, len);
        /* distinct graphs, ascending */
        uint32_t *graphs = malloc(sizeof(uint32_t) * nh);
        for (uint32_t h = 0; h < nh; h++) graphs[h] = ix->win_graph[hits[h]];
        qsort(graphs, nh, sizeof(uint32_t), cmp_u32);
        for (uint32_t h = 0; h < nh; h++) if (h == 0 || graphs[h] != graphs[h - 1]) graphs[n_graph_groups++] = graphs[h];
        r->counts.mapped++;                                                                /* boss.go:195-200 */
        if (n_graph_groups > 1) r->counts.multimapped++;
        double kmer_count_f = (double)((int)len - (int)k) + 1.0;                            /* graphminion.go:60 */
        attempts_grow(r, (uint32_t)kmer_count);
        for (uint32_t g = 0; g < n_graph_groups; g++) {
            uint32_t nk = 0;
            for (uint32_t h = 0; h < nh; h++)
                if (ix->win_graph[hits[h]] == graphs[g]) {
                    keys[nk].seg = ix->node_seg_id[ix->win_node[hits[h]]];
                    keys[nk].off = ix->win_offset[hits[h]];
                    keys[nk].w = hits[h];
                    nk++;
                }
            qsort(keys, nk, sizeof(seed_key), cmp_seed);                                    /* graphminion.go:57 */
            memcpy(work, fwd, len);                       /* each minion gets its own copy of the read */
            int rc = 0, found = 0;
            for (uint32_t m = 0; m < nk && !found; m++) {
                uint32_t w = keys[m].w;
                increment_sub_path(ix, w, kmer_count_f, r->kf_direct, r->kt_direct);        /* :67 */
                r->attempts[(size_t)kmer_count * ix->n_windows + w]++;
                if (r->no_align) continue;                                                  /* :70-72 */
                for (int t = 0; t < 2; t++) {                                               /* :76-95 */
                    uint32_t na = align_read(r, read_id, work, (int)len, rc, w);
                    if (na) { r->counts.alignments += na; found = 1; break; }
                    r->counts.revcomp_panics += oracle_revcomp(work, NULL, len) ? 1 : 0;
                    rc = !rc;
                }
            }
        }
        free(keys); free(fwd); free(work); free(graphs);
    }
    free(sk); free(hits);
    return 0;
}

void oracle_run_counts(const oracle_run *r, oracle_counts *c) { *c = r->counts; }
/* streaming use (the timed CPU baseline): forget the per-read outputs collected so far, keep the counters, the
 * call counts and the buffers' capacity -- the reference writes its records to the BAM and drops them too */
void oracle_run_drop_records(oracle_run *r) { r->seeds.n = 0; r->alns.n = 0; r->sketches.n = 0; }
uint64_t oracle_run_seeds(const oracle_run *r, const oracle_seed **out) { *out = r->seeds.p; return r->seeds.n; }
uint64_t oracle_run_alns(const oracle_run *r, const oracle_aln **out) { *out = r->alns.p; return r->alns.n; }
uint64_t oracle_run_sketches(const oracle_run *r, const uint64_t **out) { *out = r->sketches.p; return r->sketches.n; }
uint32_t oracle_run_attempts(const oracle_run *r, const uint32_t **out)
{
    *out = r->attempts;
    return r->have_attempts ? r->max_q + 1 : 0;
}

void oracle_run_weights(const oracle_run *r, int order, double *node_kmer_freq, uint64_t *graph_kmer_total)
{
    const oracle_index *ix = r->idx;
    if (order == 0) {
        memcpy(node_kmer_freq, r->kf_direct, sizeof(double) * ix->n_nodes);
        memcpy(graph_kmer_total, r->kt_direct, sizeof(uint64_t) * ix->n_graphs);
        return;
    }
    memset(node_kmer_freq, 0, sizeof(double) * ix->n_nodes);
    memset(graph_kmer_total, 0, sizeof(uint64_t) * ix->n_graphs);
    if (!r->have_attempts) return;
    for (uint32_t w = 0; w < ix->n_windows; w++)
        for (uint32_t q = 0; q <= r->max_q; q++) {
            uint32_t c = r->attempts[(size_t)q * ix->n_windows + w];
            for (uint32_t i = 0; i < c; i++) increment_sub_path(ix, w, (double)q, node_kmer_freq, graph_kmer_total);
        }
}

/* graph.go:455-525 Prune */
void oracle_prune(const oracle_index *ix, const double *kf, double min_cov, uint8_t *graph_kept, uint8_t *path_kept,
                  uint8_t *node_removed)
{
    memset(node_removed, 0, ix->n_nodes);
    for (uint32_t g = 0; g < ix->n_graphs; g++) {
        uint32_t p0 = ix->graph_path_off[g], p1 = ix->graph_path_off[g + 1];
        uint32_t n_removed_paths = 0, n_removed_nodes = 0;
        for (uint32_t p = p0; p < p1; p++) path_kept[p] = 1;
        for (uint32_t n = ix->graph_node_off[g]; n < ix->graph_node_off[g + 1]; n++) {
            double seg_len = (double)(ix->node_seq_off[n + 1] - ix->node_seq_off[n]);
            double cov = kf[n] / seg_len;                                /* :463 */
            if (cov < min_cov) {                                         /* :466-471 */
                for (uint32_t j = ix->node_np_off[n]; j < ix->node_np_off[n + 1]; j++) {
                    uint32_t p = p0 + ix->np_path[j];
                    if (path_kept[p]) { path_kept[p] = 0; n_removed_paths++; }
                    if (!node_removed[n]) { node_removed[n] = 1; n_removed_nodes++; }
                }
            }
        }
        if (n_removed_paths == p1 - p0) {                                /* :475-477 */
            graph_kept[g] = 0;
            continue;
        }
        graph_kept[g] = 1;
        (void)n_removed_nodes;
    }
}
