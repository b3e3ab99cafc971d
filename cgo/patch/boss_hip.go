// +build hip

// boss_hip.go -- drop into src/pipeline of will-rowe/groot v1.1.2 (package pipeline) and build with `-tags hip`:
// theBoss.mapReadsHIP replaces the body of theBoss.mapReads (boss.go:108-242) with the MI355X path.  The one-line hook in
// boss.go is shown in INTEGRATION.md:
//
//	func (theBoss *theBoss) mapReads() error {
//		if hipEnabled { return theBoss.mapReadsHIP() }
//		...
//
// Source only: the image this was written in has no Go toolchain, so this file has never been compiled.
//
// What stays in Go: the read channel, FASTQread (ID / Seq / Qual), sam.Record construction, the BAM writer, the graph
// Store that Prune / SaveGraphAsGFA work on afterwards.  What moves to the GPU: RunMinHash, ContainmentIndex.Query, the
// graphMinion loop, IncrementSubPath (as exact call counts) and AlignRead.
package pipeline

import (
	"fmt"
	"math/bits"
	"os"
	"strconv"

	"github.com/biogo/hts/sam"
	"github.com/will-rowe/groot/src/seqio"

	"github.com/will-rowe/groot-hip/cgo/groothip"
)

const hipEnabled = true

// batchReads is the number of reads handed to the GPU at once (GROOT_HIP_BATCH overrides)
func batchReads() int {
	if v, err := strconv.Atoi(os.Getenv("GROOT_HIP_BATCH")); err == nil && v > 0 {
		return v
	}
	return 1 << 20
}

// hipBatch keeps what Go still needs of the reads of one batch in flight
type hipBatch struct {
	reads []*seqio.FASTQread
	wire  groothip.Batch
}

func (theBoss *theBoss) mapReadsHIP() error {
	if !theBoss.info.Sketch.NoExactAlign {
		if err := theBoss.setupBAM(); err != nil {
			return err
		}
	}
	// the flat index comes from the same two files `groot align` has just loaded (cmd/align.go:93-107)
	idx, err := groothip.LoadGob(theBoss.info.IndexDir)
	if err != nil {
		return err
	}
	defer idx.Close()
	nGPU := groothip.DeviceCount()
	if nGPU == 0 {
		return fmt.Errorf("no HIP device available (the hip build of groot has no CPU fallback)")
	}
	if v, err := strconv.Atoi(os.Getenv("GROOT_HIP_GPUS")); err == nil && v > 0 && v < nGPU {
		nGPU = v
	}
	const depth = 3
	ctxs := make([]*groothip.Ctx, nGPU)
	for d := range ctxs {
		c, err := groothip.Open(d, idx, groothip.Params{ContainmentThreshold: theBoss.info.ContainmentThreshold,
			NoExactAlign: theBoss.info.Sketch.NoExactAlign, MaxReadLen: 256, MaxBatchReads: uint32(batchReads()), PipelineDepth: depth})
		if err != nil {
			return err
		}
		defer c.Close()
		ctxs[d] = c
	}
	graphNodeOff, graphPathOff := idx.GraphNodeOff(), idx.GraphPathOff()
	_ = graphPathOff

	// batches in flight per GPU, oldest first
	pending := make([][]*hipBatch, nGPU)
	collect := func(d int) error {
		res, err := ctxs[d].Collect()
		if err != nil {
			return err
		}
		b := pending[d][0]
		pending[d] = pending[d][1:]
		theBoss.receivedReadCount += int(res.Counts.Received)
		theBoss.mappedCount += int(res.Counts.Mapped)
		theBoss.multimappedCount += int(res.Counts.Multimapped)
		if !theBoss.info.Sketch.NoExactAlign {
			for t := range res.Travs {
				if err := theBoss.writeTraversal(&res.Travs[t], res.Masks[t*res.PathWords:(t+1)*res.PathWords], b, graphNodeOff); err != nil {
					return err
				}
			}
		}
		ctxs[d].Release(res)
		return nil
	}
	submit := func(d int, b *hipBatch) error {
		// The reference accepts reads of any length (boss.go:145-203); a ctx is sized for one.  A batch holding a longer read makes
		// EVERY ctx grow -- drain what is in flight, reopen with room to spare, call counts carried over -- so that all ctxs keep
		// the same kmerCount range and groot_hip_attempts_allreduce can sum their tables at the end (groot_hip_main.cpp does the same).
		if b.wire.MaxLen() > ctxs[d].MaxReadLen() {
			want := 2 * ctxs[d].MaxReadLen()
			for want < b.wire.MaxLen() {
				want *= 2
			}
			if want > 65535 {
				want = 65535
			}
			for e := range ctxs {
				for len(pending[e]) > 0 {
					if err := collect(e); err != nil {
						return err
					}
				}
				if err := ctxs[e].Reopen(want); err != nil {
					return err
				}
			}
		}
		for len(pending[d]) >= depth {
			if err := collect(d); err != nil {
				return err
			}
		}
		if err := ctxs[d].Submit(&b.wire); err != nil {
			return err
		}
		pending[d] = append(pending[d], b)
		return nil
	}

	// drain the read channel into batches, round-robin over the GPUs (reads are independent: boss.go:145-203)
	next := 0
	cur := &hipBatch{}
	for read := range theBoss.reads {
		if len(read.Seq) < theBoss.info.KmerSize {
			panic(fmt.Errorf("k size is greater than sequence length (%d vs %d)", theBoss.info.KmerSize, len(read.Seq))) // boss.go:164-166
		}
		if err := cur.wire.Add(read.Seq); err != nil {
			return err
		}
		cur.reads = append(cur.reads, read)
		if cur.wire.Len() == batchReads() {
			if err := submit(next, cur); err != nil {
				return err
			}
			next = (next + 1) % nGPU
			cur = &hipBatch{}
		}
	}
	if cur.wire.Len() > 0 {
		if err := submit(next, cur); err != nil {
			return err
		}
	}
	for d := range ctxs {
		for len(pending[d]) > 0 {
			if err := collect(d); err != nil {
				return err
			}
		}
	}

	// graph weights: what IncrementSubPath would have accumulated (graph.go:401-451), summed over the GPUs
	kmerFreq, kmerTotal, err := groothip.Weights(ctxs)
	if err != nil {
		return err
	}
	for graphID, g := range theBoss.info.Store {
		n0 := int(graphNodeOff[graphID])
		for i, node := range g.SortedNodes { // the flat index keeps SortedNodes order
			node.KmerFreq = kmerFreq[n0+i]
		}
		g.KmerTotal = kmerTotal[graphID]
	}
	if !theBoss.info.Sketch.NoExactAlign {
		return theBoss.bamwriter.Close()
	}
	return nil
}

// writeTraversal turns one traversal record into the sam.Records AlignRead builds (alignment.go:113-156): one record per
// path id in the traversal's path set, ascending, Secondary on all but the first record of the AlignRead call
func (theBoss *theBoss) writeTraversal(t *groothip.Trav, mask []uint64, b *hipBatch, graphNodeOff []uint32) error {
	read := b.reads[t.ReadID]
	g := theBoss.info.Store[t.GraphID]
	node := g.SortedNodes[int(t.Node)-int(graphNodeOff[t.GraphID])]
	references := theBoss.refSAMheaders[int(t.GraphID)]
	seq, qual := read.Seq, read.Qual
	if t.Flags&groothip.TravRC != 0 { // the record carries the reverse complement (seqio.go:120-133)
		rc := &seqio.FASTQread{Sequence: seqio.Sequence{ID: read.ID, Seq: append([]byte(nil), read.Seq...)}, Qual: append([]byte(nil), read.Qual...)}
		rc.RevComplement()
		seq, qual = rc.Seq, rc.Qual
	}
	startClip, endClip := 0, 0
	if t.Flags&groothip.TravStartClip != 0 {
		startClip = 1
	}
	if t.Flags&groothip.TravEndClip != 0 {
		endClip = 1
	}
	seqLength := len(seq) - startClip - endClip
	first := t.Flags&groothip.TravFirst != 0
	for w, word := range mask {
		for word != 0 {
			id := w*64 + bits.TrailingZeros64(word)
			word &= word - 1
			record := &sam.Record{Name: string(read.ID[1:]), Seq: sam.NewSeq(seq[0:seqLength]), Qual: qual[0:seqLength]}
			record.Ref = references[id]
			record.Pos = node.Position[id] + int(t.Offset) // alignment.go:296
			cigar := sam.Cigar{}
			if startClip != 0 {
				cigar = append(cigar, sam.NewCigarOp(sam.CigarHardClipped, startClip))
			}
			cigar = append(cigar, sam.NewCigarOp(sam.CigarMatch, seqLength))
			if endClip != 0 {
				cigar = append(cigar, sam.NewCigarOp(sam.CigarHardClipped, endClip))
			}
			record.Cigar = cigar
			record.MapQ = 30
			if !first {
				record.Flags |= sam.Secondary
			}
			first = false
			if t.Flags&groothip.TravRC != 0 {
				record.Flags |= sam.Reverse
			}
			theBoss.alignmentCount++
			if err := theBoss.bamwriter.Write(record); err != nil {
				return err
			}
		}
	}
	return nil
}
