// Package groothip binds libgroot_hip.so / libgroot_host.so (include/groot_hip.h, include/groot_host.h): the MI355X
// device path of `groot align` behind a Go API shaped for theBoss.mapReads (src/pipeline/boss.go:108-242).
//
// Source only in this repository: the image it was written in has no Go toolchain, so this package has never been
// compiled.  It uses nothing but the C ABI and the standard library; cgo/patch/boss_hip.go is the file that goes into
// the reference's src/pipeline package and uses it.
//
// cgo pointer rules: every groot_hip_submit* call copies the caller's buffers into the ctx's pinned staging BEFORE it
// returns and keeps no caller pointer afterwards, so Go slices are passed directly (&s[0]); results come back as
// pointers into C-owned pinned memory that stay valid until Release.
//
// Build (on a machine with Go, ROCm and this repository's build/ directory):
//
//	CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/build -lgroot_hip -lgroot_host -Wl,-rpath,<repo>/build" go build -tags hip ./...
package groothip

/*
#include <stdlib.h>
#include "groot_hip.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// Trav mirrors groot_trav: one successful traversal of performAlignment with the path ids processTraversal assigned
type Trav struct {
	ReadID, GraphID, Node, Offset uint32
	Ord                           uint16
	Flags, Reserved               uint8
}

// flag bits of Trav.Flags
const (
	TravRC        = 1 // read.RC: the reverse complement aligned (sam.Reverse)
	TravStartClip = 2 // 1H before the M op (alignment.go:72-85)
	TravEndClip   = 4 // 1H after the M op (alignment.go:87-103)
	TravFirst     = 8 // first traversal of its AlignRead call
)

// Counts mirrors groot_counts (the boss counters, boss.go:24-27)
type Counts struct {
	Received, Mapped, Multimapped, Alignments, Seeds, Travs, RevcompPanics, ShortReads, FullSketchReads, WalkedReads, LeanReads uint64
}

// Index owns a flat index (groot_index) loaded from the files `groot index` wrote
type Index struct {
	h    *C.groot_index
	view C.groot_index_view
}

// LoadGob reads <dir>/groot.gg + <dir>/groot.lshe (cmd/align.go:93-107) through the C++ gob reader: the Go process
// never has to flatten its own graph.Store -- both sides read the same two files
func LoadGob(dir string) (*Index, error) {
	gg, lshe := C.CString(dir+"/groot.gg"), C.CString(dir+"/groot.lshe")
	defer C.free(unsafe.Pointer(gg))
	defer C.free(unsafe.Pointer(lshe))
	idx := &Index{}
	if rc := C.groot_index_load_gob(gg, lshe, &idx.h); rc != 0 {
		return nil, fmt.Errorf("groot_index_load_gob: %s", C.GoString(C.groot_host_last_error()))
	}
	C.groot_index_get_view(idx.h, &idx.view)
	return idx, nil
}

// Close frees the index
func (idx *Index) Close() {
	if idx.h != nil {
		C.groot_index_free(idx.h)
		idx.h = nil
	}
}

// NumGraphs, NumNodes, NumWindows of the flat index
func (idx *Index) NumGraphs() int  { return int(idx.view.n_graphs) }
func (idx *Index) NumNodes() int   { return int(idx.view.n_nodes) }
func (idx *Index) NumWindows() int { return int(idx.view.n_windows) }
func (idx *Index) PathWords() int  { return int(idx.view.path_words) }

func u32s(p *C.uint32_t, n int) []uint32 {
	if n == 0 {
		return nil
	}
	return (*[1 << 30]uint32)(unsafe.Pointer(p))[:n:n]
}

// GraphNodeOff[g] .. GraphNodeOff[g+1] = global node indices of graph g, in SortedNodes order
func (idx *Index) GraphNodeOff() []uint32 { return u32s(idx.view.graph_node_off, idx.NumGraphs()+1) }

// GraphPathOff[g] = global path index of local path id 0 of graph g
func (idx *Index) GraphPathOff() []uint32 { return u32s(idx.view.graph_path_off, idx.NumGraphs()+1) }

// NodeSegID[n] = GrootGraphNode.SegmentID of global node n
func (idx *Index) NodeSegID() []uint32 { return u32s(idx.view.node_seg_id, idx.NumNodes()) }

// Ctx is one GPU context (groot_ctx): the replicated index in HBM plus a ring of batches in flight
type Ctx struct {
	h      *C.groot_ctx
	idx    *Index
	device int
	params Params
}

// Params mirrors the fields of groot_params a host sets
type Params struct {
	ContainmentThreshold float64
	NoExactAlign         bool
	MaxReadLen           uint32
	MaxBatchReads        uint32
	PipelineDepth        uint32
	// MemoBudgetMB: 0 = the library's default budget for the memo of groot_hip_open, MemoOff = no memo (a run over a few
	// million reads: the memo costs more at open than it saves), else MiB
	MemoBudgetMB uint32
	// Background: GROOT_OPEN_BACKGROUND -- Open returns once the ctx can take batches and builds the prefix tables and the signature
	// index on a thread of its own (the Index must stay alive, which the Ctx sees to)
	Background bool
}

// MemoOff is GROOT_MEMO_OFF
const MemoOff = uint32(C.GROOT_MEMO_OFF)

// DeviceCount returns the number of visible GPUs
func DeviceCount() int {
	var n C.int
	if C.groot_hip_device_count(&n) != 0 {
		return 0
	}
	return int(n)
}

// Open uploads the index to GPU `device`
func Open(device int, idx *Index, p Params) (*Ctx, error) {
	var prm C.groot_params
	C.groot_params_default(&prm)
	prm.containment_threshold = C.double(p.ContainmentThreshold)
	if p.NoExactAlign {
		prm.no_exact_align = 1
	}
	if p.MaxReadLen != 0 {
		prm.max_read_len = C.uint32_t(p.MaxReadLen)
	}
	if p.MaxBatchReads != 0 {
		prm.max_batch_reads = C.uint32_t(p.MaxBatchReads)
	}
	if p.PipelineDepth != 0 {
		prm.pipeline_depth = C.uint32_t(p.PipelineDepth)
	}
	prm.memo_budget_mb = C.uint32_t(p.MemoBudgetMB)
	c := &Ctx{idx: idx, device: device, params: p}
	if c.params.MaxReadLen == 0 {
		c.params.MaxReadLen = uint32(prm.max_read_len)
	}
	var flags C.uint32_t
	if p.Background {
		flags = C.GROOT_OPEN_BACKGROUND
	}
	if rc := C.groot_hip_open_flags(&c.h, C.int(device), &idx.view, &prm, flags); rc != 0 {
		return nil, fmt.Errorf("groot_hip_open: %s", C.GoString(C.groot_hip_last_error(nil)))
	}
	return c, nil
}

// OpenAbandon tells a background open (Params.Background) to stop at its next checkpoint: for an input that has ended before the
// tables were there.  The ctx keeps working, through the full-width kernels.
func (c *Ctx) OpenAbandon() { C.groot_hip_open_abandon(c.h) }

// MaxReadLen is the longest read the ctx accepts (a longer one fails its batch with GROOT_E_NOSPACE)
func (c *Ctx) MaxReadLen() int { return int(c.params.MaxReadLen) }

// Reopen replaces the ctx by one that accepts reads of up to maxReadLen bases and carries the IncrementSubPath call counts
// over (groot_hip_attempts_export -> close -> open -> groot_hip_attempts_import, in that order): the reference has no read length limit
// (boss.go:145-203), the device sizes its LDS staging and DFS stacks for one.  Nothing may be in flight.
func (c *Ctx) Reopen(maxReadLen int) error {
	var nRows, nWin C.uint32_t
	if rc := C.groot_hip_attempts_export(c.h, nil, nil, 0, &nRows, &nWin); rc != 0 {
		return c.err("groot_hip_attempts_export")
	}
	q := make([]uint32, int(nRows)+1)
	counts := make([]uint32, int(nRows)*int(nWin)+1)
	if rc := C.groot_hip_attempts_export(c.h, (*C.uint32_t)(unsafe.Pointer(&q[0])), (*C.uint32_t)(unsafe.Pointer(&counts[0])), nRows, &nRows, &nWin); rc != 0 {
		return c.err("groot_hip_attempts_export")
	}
	// close first, then open: the index and the tables of groot_hip_open would otherwise sit in HBM twice
	old := c.params
	p := c.params
	p.MaxReadLen = uint32(maxReadLen)
	C.groot_hip_close(c.h)
	c.h = nil
	reopen := func(p Params) (*Ctx, error) {
		n, err := Open(c.device, c.idx, p)
		if err != nil {
			return nil, err
		}
		if rc := C.groot_hip_attempts_import(n.h, (*C.uint32_t)(unsafe.Pointer(&q[0])), (*C.uint32_t)(unsafe.Pointer(&counts[0])), nRows); rc != 0 {
			err := n.err("groot_hip_attempts_import")
			n.Close()
			return nil, err
		}
		return n, nil
	}
	bigger, err := reopen(p)
	if err != nil {
		// the larger ctx could not be had (HBM, a kmerCount outside its range): back to a ctx with the old limit and the counts exported
		// above, so that the caller still holds a working Ctx and loses nothing; only if that fails too is the Ctx dead (c.h == nil)
		if back, err2 := reopen(old); err2 == nil {
			c.h, c.params = back.h, back.params
			return fmt.Errorf("reopen for reads of %d bases: %v (the ctx keeps its limit of %d)", maxReadLen, err, old.MaxReadLen)
		}
		return fmt.Errorf("reopen for reads of %d bases: %v; the ctx could not be restored either and is closed", maxReadLen, err)
	}
	c.h, c.params = bigger.h, bigger.params
	return nil
}

// Close releases the GPU context
func (c *Ctx) Close() {
	if c.h != nil {
		C.groot_hip_close(c.h)
		c.h = nil
	}
}

func (c *Ctx) err(what string) error {
	return fmt.Errorf("%s: %s", what, C.GoString(C.groot_hip_last_error(c.h)))
}

// Batch is the wire format of one batch of reads: 2 bits per base, one uint16 length per read, and the list of bytes
// that are not A, C, G or T (their position in the concatenation of all reads of the batch)
type Batch struct {
	Packed  []byte
	Lens    []uint16
	ExcPos  []uint64
	ExcByte []byte
	nBases  uint64
	maxLen  int
}

// Add appends read.Seq to the batch (FASTQread.Seq as it came: no upper-casing, sketch.go:258-282 removed the QC)
func (b *Batch) Add(seq []byte) error {
	if len(seq) > 65535 {
		return fmt.Errorf("read longer than 65535 bases")
	}
	for _, ch := range seq {
		if b.nBases&3 == 0 {
			b.Packed = append(b.Packed, 0)
		}
		b.Packed[len(b.Packed)-1] |= ((ch >> 1) & 3) << (2 * (b.nBases & 3))
		if ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' {
			b.ExcPos = append(b.ExcPos, b.nBases)
			b.ExcByte = append(b.ExcByte, ch)
		}
		b.nBases++
	}
	b.Lens = append(b.Lens, uint16(len(seq)))
	if len(seq) > b.maxLen {
		b.maxLen = len(seq)
	}
	return nil
}

// MaxLen is the length of the longest read in the batch
func (b *Batch) MaxLen() int { return b.maxLen }

// Len is the number of reads in the batch
func (b *Batch) Len() int { return len(b.Lens) }

// Reset empties the batch, keeping its capacity
func (b *Batch) Reset() {
	b.Packed, b.Lens, b.ExcPos, b.ExcByte, b.nBases, b.maxLen = b.Packed[:0], b.Lens[:0], b.ExcPos[:0], b.ExcByte[:0], 0, 0
}

// Submit enqueues the batch (copy to pinned staging -> H2D -> kernels -> D2H, asynchronous).  ErrFull means
// PipelineDepth batches are submitted and not released: Collect + Release first.
func (c *Ctx) Submit(b *Batch) error {
	if b.Len() == 0 {
		return nil
	}
	var packed *C.uint8_t
	if len(b.Packed) > 0 {
		packed = (*C.uint8_t)(unsafe.Pointer(&b.Packed[0]))
	}
	var excPos *C.uint64_t
	var excByte *C.uint8_t
	if len(b.ExcPos) > 0 {
		excPos = (*C.uint64_t)(unsafe.Pointer(&b.ExcPos[0]))
		excByte = (*C.uint8_t)(unsafe.Pointer(&b.ExcByte[0]))
	}
	rc := C.groot_hip_submit_packed16(c.h, packed, (*C.uint16_t)(unsafe.Pointer(&b.Lens[0])), C.uint32_t(len(b.Lens)), 0,
		excPos, excByte, C.uint64_t(len(b.ExcPos)))
	if rc == C.GROOT_E_STATE {
		return ErrFull
	}
	if rc != 0 {
		return c.err("groot_hip_submit_packed16")
	}
	return nil
}

// ErrFull is returned by Submit when every pipeline slot is taken
var ErrFull = fmt.Errorf("groot-hip: pipeline full")

// Result describes one finished batch; Travs / Masks point into pinned memory owned by the ctx and stay valid until Release
type Result struct {
	Ticket    uint64
	NumReads  int
	Counts    Counts
	Travs     []Trav
	Masks     []uint64 // PathWords words per traversal: bit p = local path id p (widened from the compact wire form)
	PathWords int
}

// InFlight is the number of batches submitted and not yet collected
func (c *Ctx) InFlight() int {
	var n, free C.uint32_t
	C.groot_hip_in_flight(c.h, &n, &free)
	return int(n)
}

// Collect blocks until the oldest submitted batch is finished.  The reference's panics come back as errors: a read
// shorter than k (boss.go:164-166), a byte > 'T' reaching RevComplement (seqio.go:126).
func (c *Ctx) Collect() (*Result, error) {
	var r C.groot_batch_result
	rc := C.groot_hip_collect(c.h, &r)
	if rc != 0 {
		return nil, c.err("groot_hip_collect")
	}
	n := int(r.n_travs)
	res := &Result{Ticket: uint64(r.ticket), NumReads: int(r.n_reads), PathWords: int(r.path_words)}
	res.Counts = Counts{uint64(r.counts.received), uint64(r.counts.mapped), uint64(r.counts.multimapped), uint64(r.counts.alignments),
		uint64(r.counts.seeds), uint64(r.counts.travs), uint64(r.counts.revcomp_panics), uint64(r.counts.short_reads), uint64(r.counts.full_sketch_reads), uint64(r.counts.walked_reads), uint64(r.counts.lean_reads)}
	if n > 0 {
		res.Travs = (*[1 << 28]Trav)(unsafe.Pointer(r.travs))[:n:n]
		// the path sets travel compact (as many bytes as the traversal's graph has paths / 8); widen them once per batch
		res.Masks = make([]uint64, n*res.PathWords)
		if rc := C.groot_host_unpack_masks(&c.idx.view, r.travs, C.uint64_t(n), r.masks, (*C.uint64_t)(unsafe.Pointer(&res.Masks[0]))); rc != 0 {
			return nil, fmt.Errorf("groot_host_unpack_masks: %s", C.GoString(C.groot_host_last_error()))
		}
	}
	return res, nil
}

// Release hands the batch's slot back to the ctx; the slices of the Result must not be used afterwards
func (c *Ctx) Release(r *Result) {
	C.groot_hip_release(c.h, C.uint64_t(r.Ticket))
	r.Travs, r.Masks = nil, nil
}

// Weights sums the IncrementSubPath call counts over the ctxs (RCCL all-reduce across GPUs) and replays
// GrootGraph.IncrementSubPath (graph.go:401-451) in the canonical order: KmerFreq per global node, KmerTotal per graph
func Weights(ctxs []*Ctx) (kmerFreq []float64, kmerTotal []uint64, err error) {
	if len(ctxs) == 0 {
		return nil, nil, fmt.Errorf("no ctx")
	}
	hs := make([]*C.groot_ctx, len(ctxs))
	for i, c := range ctxs {
		hs[i] = c.h
	}
	// the array of ctx handles lives in C memory for the call (a Go slice of C pointers is fine too; this keeps vet quiet)
	arr := (**C.groot_ctx)(C.malloc(C.size_t(len(hs)) * C.size_t(unsafe.Sizeof(hs[0]))))
	defer C.free(unsafe.Pointer(arr))
	copy((*[1 << 20]*C.groot_ctx)(unsafe.Pointer(arr))[:len(hs):len(hs)], hs)
	if rc := C.groot_hip_attempts_allreduce(arr, C.int(len(hs))); rc != 0 {
		return nil, nil, ctxs[0].err("groot_hip_attempts_allreduce")
	}
	c := ctxs[0]
	var nRows, nWin C.uint32_t
	if rc := C.groot_hip_attempts_export(c.h, nil, nil, 0, &nRows, &nWin); rc != 0 {
		return nil, nil, c.err("groot_hip_attempts_export")
	}
	q := make([]uint32, int(nRows)+1)
	counts := make([]uint32, int(nRows)*int(nWin)+1)
	if rc := C.groot_hip_attempts_export(c.h, (*C.uint32_t)(unsafe.Pointer(&q[0])), (*C.uint32_t)(unsafe.Pointer(&counts[0])), nRows, &nRows, &nWin); rc != 0 {
		return nil, nil, c.err("groot_hip_attempts_export")
	}
	kmerFreq = make([]float64, c.idx.NumNodes()+1)
	kmerTotal = make([]uint64, c.idx.NumGraphs()+1)
	if rc := C.groot_host_weights_rows(&c.idx.view, (*C.uint32_t)(unsafe.Pointer(&q[0])), nRows, (*C.uint32_t)(unsafe.Pointer(&counts[0])),
		(*C.double)(unsafe.Pointer(&kmerFreq[0])), (*C.uint64_t)(unsafe.Pointer(&kmerTotal[0]))); rc != 0 {
		return nil, nil, fmt.Errorf("groot_host_weights_rows: %s", C.GoString(C.groot_host_last_error()))
	}
	return kmerFreq[:c.idx.NumNodes()], kmerTotal[:c.idx.NumGraphs()], nil
}
