/*
 * groot_index.h -- flat, read-only view of a GROOT index, shared by the host library
 * (include/groot_host.h) and the device library (include/groot_hip.h).
 *
 * The reference keeps the index as two Go gob files: groot.gg = pipeline.Info incl. graph.Store
 * (src/pipeline/runtime.go:15-33, src/graph/graph.go:18-34, src/graph/node.go:13-22) and
 * groot.lshe = lshe.ContainmentIndex (src/lshe/lshe.go:38-44, Key at lshe.go:17-28).  This view
 * carries exactly those exported fields as little-endian POD arrays so that it can be handed
 * across a C ABI (cgo / ctypes) without any Go/C++ types, and uploaded to HBM verbatim.
 *
 * Conventions
 *   - graph g owns nodes [graph_node_off[g], graph_node_off[g+1]) in GrootGraph.SortedNodes order;
 *     a "global node index" is a position in that concatenation (it replaces NodeLookup[SegmentID]).
 *   - graph g owns paths [graph_path_off[g], graph_path_off[g+1]); local path id = GrootGraph.Paths key.
 *   - windows (lshe.Key) are numbered in the canonical seed order of SURVEY 8c:
 *     (GraphID, Key.Node SegmentID, Key.OffSet, position in the "g%dn%do%d" list).
 */
#ifndef GROOT_INDEX_H
#define GROOT_INDEX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct groot_index_view {
    /* pipeline.Info / lshe.ContainmentIndex scalars */
    uint32_t kmer_size;        /* Info.KmerSize                                   */
    uint32_t sketch_size;      /* Info.SketchSize                                 */
    uint32_t window_size;      /* Info.WindowSize (= Key.WindowSize)              */
    uint32_t num_part;         /* Info.NumPart                                    */
    uint32_t max_k;            /* Info.MaxK                                       */
    uint32_t num_window_kmers; /* ContainmentIndex.NumWindowKmers = w-k+1         */
    uint32_t path_words;       /* u64 words per node path bitset                  */
    uint32_t reserved0;
    uint32_t n_graphs, n_nodes, n_edges, n_paths, n_windows, reserved1;
    uint64_t n_bases, n_np, n_cn, n_wref, n_name_bytes;
    /* graph.Store */
    const uint32_t *graph_node_off;  /* [n_graphs+1]                                            */
    const uint32_t *graph_path_off;  /* [n_graphs+1]                                            */
    const uint8_t  *graph_masked;    /* [n_graphs]   GrootGraph.Masked                          */
    const uint32_t *node_seg_id;     /* [n_nodes]    GrootGraphNode.SegmentID                   */
    const uint32_t *node_seq_off;    /* [n_nodes+1]  into bases (SegmentLength = difference)    */
    const uint32_t *node_edge_off;   /* [n_nodes+1]  into edges, GrootGraphNode.OutEdges order  */
    const uint32_t *node_np_off;     /* [n_nodes+1]  into np_path/np_pos, PathIDs order         */
    const uint64_t *node_mask;       /* [n_nodes*path_words] bitset over local path ids         */
    const uint8_t  *bases;           /* [n_bases]    GrootGraphNode.Sequence, upper-case ACGTN  */
    const uint32_t *edges;           /* [n_edges]    global node index of the neighbour         */
    const uint32_t *np_path;         /* [n_np]       local path id                              */
    const uint32_t *np_pos;          /* [n_np]       GrootGraphNode.Position[pathID]            */
    const uint32_t *path_len;        /* [n_paths]    GrootGraph.Lengths                         */
    const uint32_t *path_name_off;   /* [n_paths+1]  into path_names                            */
    const char     *path_names;      /* GrootGraph.Paths values, concatenated, no terminators   */
    /* lshe.ContainmentIndex.WindowLookup */
    const uint32_t *win_graph;       /* [n_windows]  Key.GraphID                                */
    const uint32_t *win_node;        /* [n_windows]  global node index of Key.Node              */
    const uint32_t *win_offset;      /* [n_windows]  Key.OffSet                                 */
    const uint32_t *win_merge_span;  /* [n_windows]  Key.MergeSpan                              */
    const uint32_t *win_cn_off;      /* [n_windows+1] into cn_node/cn_count                     */
    const uint32_t *cn_node;         /* [n_cn]  Key.ContainedNodes keys, ascending SegmentID    */
    const uint32_t *cn_count;        /* [n_cn]  Key.ContainedNodes values (integral float64)    */
    const uint32_t *win_ref_off;     /* [n_windows+1] into win_ref                              */
    const uint32_t *win_ref;         /* [n_wref] Key.Ref (local path ids)                       */
    const uint64_t *win_sketch;      /* [n_windows*sketch_size] Key.Sketch                      */
} groot_index_view;

#ifdef __cplusplus
}
#endif
#endif
