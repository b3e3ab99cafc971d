// Command golden asks the REFERENCE itself -- will-rowe/groot v1.1.2 with exactly the third-party modules its go.mod
// pins (will-rowe/nthash v0.2.0, ekzhu/lshensemble v1.1.0, biogo/hts v1.1.0, will-rowe/gfa) -- for the values the
// MI355X build restates from memory, and writes them where tests/test_reference_golden.py picks them up:
//
//	<out>/golden.json       ntHash / MultiHash values of fixed sequences        (nthash: NewHasher, MultiHash)
//	                        KHF sketch of every read                           (src/minhash/khf.go:35-55)
//	                        ContainmentIndex.Query hits of every read          (src/lshe/lshe.go:153-175)
//	                        lshensemble.Containment of every hit, the read counters of the align run (boss.go:194-200)
//	<out>/groot.gg          pipeline.Info incl. graph.Store, as `groot index` writes it (runtime.go:64-72)
//	<out>/groot.lshe        lshe.ContainmentIndex                                (lshe.go:71-92)
//	<out>/out.bam           the BAM of `groot align` on the reads                (boss.go:45-105,225-240)
//	<out>/graphs/*.gfa      the weighted graphs                                  (graphio.go:19-112)
//
// This program cannot be built in the image the MI355X build was written in (no Go toolchain, no module cache); it is
// shipped as source.  On any machine with Go >= 1.14 and network access (or a populated module cache):
//
//	cd tools/golden && go mod tidy && go run . -msa <dir with cluster*.msa> -fastq reads.fq[.gz] -out ../../tests/golden/reference/<case> \
//	        -k 31 -s 21 -w 100 -t 0.99
//
// (or copy main.go into a checkout of will-rowe/groot at v1.1.2 as tools/golden/main.go and `go run ./tools/golden ...`).
// tests/golden/reference/README.md lists the cases the intake test expects.
package main

import (
	"bufio"
	"compress/gzip"
	"encoding/json"
	"flag"
	"fmt"
	"io"
	"io/ioutil"
	"log"
	"os"
	"path/filepath"
	"sort"
	"strings"

	"github.com/ekzhu/lshensemble"
	"github.com/will-rowe/groot/src/lshe"
	"github.com/will-rowe/groot/src/minhash"
	"github.com/will-rowe/groot/src/pipeline"
	"github.com/will-rowe/groot/src/version"
	"github.com/will-rowe/nthash"
)

// hashVector: MultiHash(canonical=true, n) of every k-mer of one sequence -- h[0] is the canonical ntHash itself
type hashVector struct {
	Seq    string     `json:"seq"`
	K      uint       `json:"k"`
	N      uint       `json:"n"`
	Hashes [][]uint64 `json:"hashes"`
}

// one lshe.Key returned by Query, in the fields that identify the window
type hit struct {
	GraphID     uint32   `json:"graph"`
	Node        uint64   `json:"node"`
	OffSet      uint32   `json:"offset"`
	MergeSpan   uint32   `json:"merge_span"`
	Ref         []uint32 `json:"ref"`
	Containment float64  `json:"containment"`
}

type readOut struct {
	Name      string   `json:"name"`
	Len       int      `json:"len"`
	KmerCount int      `json:"kmer_count"`
	Sketch    []uint64 `json:"sketch"`
	Hits      []hit    `json:"hits"`
}

type golden struct {
	GrootVersion string       `json:"groot_version"`
	Fastq        string       `json:"fastq"`
	K            int          `json:"k"`
	S            int          `json:"s"`
	W            int          `json:"w"`
	NumPart      int          `json:"num_part"`
	MaxK         int          `json:"max_k"`
	Threshold    float64      `json:"threshold"`
	MinKmerCov   float64      `json:"min_kmer_cov"`
	NtHash       []hashVector `json:"nthash"`
	Reads        []readOut    `json:"reads"`
	// [received, mapped, multimapped, total k-mers projected] of the align run (sketch.go:303-305)
	ReadStats  [4]int   `json:"read_stats"`
	KeptPaths  []string `json:"kept_paths"`
	NumWindows int      `json:"num_windows"`
	NumGraphs  int      `json:"num_graphs"`
}

func check(err error) {
	if err != nil {
		log.Fatal(err)
	}
}

// every four lines form one read (sketch.go:213-236); .gz by extension (sketch.go:60-68)
func readFastq(path string) (names []string, seqs [][]byte) {
	fh, err := os.Open(path)
	check(err)
	defer fh.Close()
	var r io.Reader = fh
	if strings.HasSuffix(path, ".gz") {
		gz, err := gzip.NewReader(fh)
		check(err)
		defer gz.Close()
		r = gz
	}
	sc := bufio.NewScanner(r)
	sc.Buffer(make([]byte, 1<<20), 1<<26)
	var lines [][]byte
	for sc.Scan() {
		lines = append(lines, append([]byte(nil), sc.Bytes()...))
		if len(lines) == 4 {
			if len(lines[0]) == 0 || lines[0][0] != '@' {
				log.Fatalf("read ID in fastq file does not begin with @: %s", lines[0])
			}
			names = append(names, string(lines[0][1:]))
			seqs = append(seqs, lines[1])
			lines = lines[:0]
		}
	}
	check(sc.Err())
	return names, seqs
}

func multiHashes(seq string, k, n uint) hashVector {
	b := []byte(seq)
	hasher, err := nthash.NewHasher(&b, k)
	check(err)
	out := hashVector{Seq: seq, K: k, N: n}
	for hv := range hasher.MultiHash(true, n) {
		out.Hashes = append(out.Hashes, append([]uint64(nil), hv...))
	}
	return out
}

func main() {
	msaDir := flag.String("msa", "", "directory with cluster*.msa files (what `groot index -m` takes)")
	fastq := flag.String("fastq", "", "FASTQ file (optionally .gz)")
	outDir := flag.String("out", "", "output directory")
	k := flag.Int("k", 31, "k-mer size")
	s := flag.Int("s", 21, "sketch size")
	w := flag.Int("w", 100, "window size")
	x := flag.Int("x", 8, "LSH Ensemble partitions")
	y := flag.Int("y", 4, "LSH Ensemble max K")
	t := flag.Float64("t", 0.99, "containment threshold")
	c := flag.Float64("c", 1.0, "minimum k-mer coverage for pruning")
	p := flag.Int("p", 1, "processors")
	flag.Parse()
	if *msaDir == "" || *fastq == "" || *outDir == "" {
		flag.Usage()
		os.Exit(1)
	}
	check(os.MkdirAll(filepath.Join(*outDir, "graphs"), 0755))
	g := golden{GrootVersion: version.GetVersion(), Fastq: filepath.Base(*fastq), K: *k, S: *s, W: *w, NumPart: *x, MaxK: *y, Threshold: *t, MinKmerCov: *c}

	// ---- 1. third-party hash arithmetic on fixed inputs (seqA of src/minhash/minhash_test.go:13 among them) ----
	fixed := []string{
		"ACTGCGTGCGTGAAACGTGCACGTGACGTG",
		"CACGTCACGTGCACGTTTCACGCACGCAGT",
		"ATGAAAGGATTAAAAGGGCTATTGGTTCTGGCTTTAGGCTTTACAGGACTAC",
		"NNNNACGTacgtNNNNACGTACGTTTTTGGGGCCCCAAAATGCATGCATGCA",
	}
	for _, sq := range fixed {
		g.NtHash = append(g.NtHash, multiHashes(sq, 7, 10))
		g.NtHash = append(g.NtHash, multiHashes(sq, uint(*k), uint(*s)))
	}

	// ---- 2. `groot index` (cmd/index.go:96-131) ----
	msaList, err := filepath.Glob(*msaDir + "/cluster*.msa")
	check(err)
	if len(msaList) == 0 {
		log.Fatalf("no cluster*.msa files in %s", *msaDir)
	}
	info := &pipeline.Info{
		Version: version.GetVersion(), NumProc: *p, KmerSize: *k, SketchSize: *s, WindowSize: *w, NumPart: *x, MaxK: *y,
		MaxSketchSpan: 30, IndexDir: *outDir,
	}
	indexing := pipeline.NewPipeline()
	msaConverter := pipeline.NewMSAconverter(info)
	graphSketcher := pipeline.NewGraphSketcher(info)
	sketchIndexer := pipeline.NewSketchIndexer(info)
	msaConverter.Connect(msaList)
	graphSketcher.Connect(msaConverter)
	sketchIndexer.Connect(graphSketcher)
	indexing.AddProcesses(msaConverter, graphSketcher, sketchIndexer)
	indexing.Run()
	check(info.SaveDB(filepath.Join(*outDir, "groot.lshe")))
	check(info.Dump(filepath.Join(*outDir, "groot.gg")))

	// ---- 3. load it back as `groot align` does (cmd/align.go:93-107) ----
	loaded := new(pipeline.Info)
	check(loaded.Load(filepath.Join(*outDir, "groot.gg")))
	index := &lshe.ContainmentIndex{}
	check(index.Load(filepath.Join(*outDir, "groot.lshe")))
	loaded.AttachDB(index)
	g.NumWindows = len(index.WindowLookup)
	g.NumGraphs = len(loaded.Store)

	// ---- 4. per read: KHF sketch and Query hits, exactly the two calls of boss.go:163-172 ----
	names, seqs := readFastq(*fastq)
	for i, sq := range seqs {
		mh := minhash.NewKHFsketch(uint(*k), uint(*s))
		check(mh.AddSequence(sq))
		sketch := append([]uint64(nil), mh.GetSketch()...)
		kmerCount := (len(sq) - *k) + 1
		results, err := index.Query(sketch, kmerCount, *t)
		check(err)
		ro := readOut{Name: names[i], Len: len(sq), KmerCount: kmerCount, Sketch: sketch}
		for _, keys := range results {
			for _, key := range keys {
				ro.Hits = append(ro.Hits, hit{GraphID: key.GraphID, Node: key.Node, OffSet: key.OffSet, MergeSpan: key.MergeSpan,
					Ref: append([]uint32(nil), key.Ref...),
					Containment: lshensemble.Containment(sketch, key.Sketch, kmerCount, index.NumWindowKmers)})
			}
		}
		// map iteration order is random: fix an order for the dump
		sort.Slice(ro.Hits, func(a, b int) bool {
			ha, hb := ro.Hits[a], ro.Hits[b]
			if ha.GraphID != hb.GraphID {
				return ha.GraphID < hb.GraphID
			}
			if ha.Node != hb.Node {
				return ha.Node < hb.Node
			}
			if ha.OffSet != hb.OffSet {
				return ha.OffSet < hb.OffSet
			}
			return ha.MergeSpan < hb.MergeSpan
		})
		g.Reads = append(g.Reads, ro)
	}

	// ---- 5. `groot align` (cmd/align.go:110-161): BAM + weighted graphs ----
	loaded.NumProc = *p
	loaded.ContainmentThreshold = *t
	loaded.Sketch = pipeline.AlignCmd{MinKmerCoverage: *c, BAMout: filepath.Join(*outDir, "out.bam")}
	aligning := pipeline.NewPipeline()
	dataStream := pipeline.NewDataStreamer(loaded)
	fastqHandler := pipeline.NewFastqHandler(loaded)
	fastqChecker := pipeline.NewFastqChecker(loaded)
	readMapper := pipeline.NewReadMapper(loaded)
	graphPruner := pipeline.NewGraphPruner(loaded, false)
	dataStream.Connect([]string{*fastq})
	fastqHandler.Connect(dataStream)
	fastqChecker.Connect(fastqHandler)
	readMapper.Connect(fastqChecker)
	graphPruner.Connect(readMapper)
	aligning.AddProcesses(dataStream, fastqHandler, fastqChecker, readMapper, graphPruner)
	aligning.Run()
	g.ReadStats = readMapper.CollectReadStats()
	g.KeptPaths = append([]string(nil), graphPruner.CollectOutput()...)
	sort.Strings(g.KeptPaths)
	for graphID, gr := range loaded.Store {
		_, err := gr.SaveGraphAsGFA(fmt.Sprintf("%s/graphs/groot-graph-%d.gfa", *outDir, graphID), g.ReadStats[3])
		check(err)
	}

	blob, err := json.Marshal(&g)
	check(err)
	check(ioutil.WriteFile(filepath.Join(*outDir, "golden.json"), blob, 0644))
	log.Printf("wrote %s: %d reads, %d windows, %d graphs", filepath.Join(*outDir, "golden.json"), len(g.Reads), g.NumWindows, g.NumGraphs)
}
