"""The FASTQ ingest alone, on the host: groot_reads_open / next / free over a plain (or .gz) FASTQ file.   python tools/parse_bench.py FILE [threads] [repeat]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groot_amd import host

path = sys.argv[1]
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = host.lib()
L.groot_reads_batch_free.argtypes = [C.c_void_p]
L.groot_reads_batch_free.restype = None
L.groot_reads_batch_view.argtypes = [C.c_void_p, C.POINTER(host.ReadsView)]
L.groot_reads_batch_view.restype = None
L.groot_reads_close.argtypes = [C.c_void_p]
L.groot_reads_close.restype = None
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    h = C.c_void_p()
    arr = (C.c_char_p * 1)(path.encode())
    t0 = time.perf_counter()
    host._check(L.groot_reads_open(arr, C.c_uint32(1), C.c_uint32(threads), C.c_uint64(0), C.c_uint32(262144), C.c_uint64(0), C.byref(h)))
    n = bases = 0
    while True:
        b = C.c_void_p()
        host._check(L.groot_reads_next(h, C.byref(b)))
        if not b.value:
            break
        v = host.ReadsView()
        L.groot_reads_batch_view(b, C.byref(v))
        n += v.n_reads
        bases += v.n_bases
        L.groot_reads_batch_free(b)
    dt = time.perf_counter() - t0
    L.groot_reads_close(h)
    print("%d reads, %d bases, %d threads: %.3f s = %.1f Mreads/s, %.0f ns per read and thread" % (n, bases, threads, dt, n / dt / 1e6, dt * threads / n * 1e9))
