"""One worker of bench.py's cpu_baseline leg: the oracle (test infrastructure -- the CPU restatement of the reference's
align path) timed on its own slice of the synthetic read stream, as one of N independent processes (the stand-in for the
reference's `groot align -p N` goroutines: reads are independent, the index is read-only).

    python oracle/cpu_worker.py <index.gidx> <first_read> <n_reads> <read_len> <chunk>

Protocol: prints "ready" once the index is loaded and the reads are generated, waits for a line on stdin, runs, prints one
JSON line {reads, t_start, t_end, seconds, mapped}.  Records are dropped chunk by chunk (the reference writes them to the
BAM and forgets them); only the counters and call counts accumulate."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    path, first, n, read_len, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    from groot_amd import host, synth
    from oracle import oracle_py as O

    index = host.Index.load(path)
    cat, off, lens = synth.reference_sequences(index)
    chunks = []
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        seq, seq_off, _ = synth.reads_np(cat, off, lens, m, read_len, first=first + c0)
        chunks.append((seq, seq_off, first + c0))
    run = O.Run(index, 0.99)
    print("ready", flush=True)
    sys.stdin.readline()               # all workers start together: the aggregate is measured under full load
    t0 = time.time()
    for seq, seq_off, f in chunks:
        run.batch(seq, seq_off, first_read_id=f)
        run.drop_records()
    t1 = time.time()
    c = run.counts()
    print(json.dumps({"reads": n, "t_start": t0, "t_end": t1, "seconds": t1 - t0, "mapped": int(c["mapped"])}), flush=True)


if __name__ == "__main__":
    main()
