// differential check of groot_amd/csrc/host/gz_inflate.hpp against zlib's gzread (tests/test_gz_inflate.py compiles and runs it):
//   gz_check FILE [chunk]   -> "same N" | "differ ..." | "error: MESSAGE" (zlib's verdict beside it);  gz_check --time FILE -> seconds of both
#include <fcntl.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../groot_amd/csrc/host/gz_inflate.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool with_zlib(const char *path, std::vector<uint8_t> &out)
{
    gzFile f = gzopen(path, "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    std::vector<uint8_t> buf(1 << 20);
    bool ok = true;
    for (;;) {
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) { ok = false; break; }
        if (n == 0) break;
        out.insert(out.end(), buf.begin(), buf.begin() + n);
    }
    int errnum = 0;
    gzerror(f, &errnum);
    if (errnum != Z_OK && errnum != Z_STREAM_END) ok = false;
    gzclose(f);
    return ok;
}
static int with_ours(const char *path, std::vector<uint8_t> &out, size_t chunk, std::string &err)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { err = "open"; return -1; }
    groot::GzInflater *g = new groot::GzInflater(fd);
    std::vector<uint8_t> buf(chunk);
    int rc = 0;
    for (;;) {
        const ssize_t n = g->read(buf.data(), buf.size());
        if (n < 0) { err = g->error(); rc = -1; break; }
        if (n == 0) break;
        out.insert(out.end(), buf.begin(), buf.begin() + n);
    }
    delete g;
    close(fd);
    return rc;
}
int main(int argc, char **argv)
{
    if (argc >= 3 && !strcmp(argv[1], "--raw")) {           // inflate only, nothing kept
        std::vector<uint8_t> buf(1 << 20);
        double t0 = now();
        size_t na = 0, nb = 0;
        { gzFile f = gzopen(argv[2], "rb"); gzbuffer(f, 1 << 20); for (int n; (n = gzread(f, buf.data(), (unsigned)buf.size())) > 0;) na += (size_t)n; gzclose(f); }
        double t1 = now();
        { const int fd = open(argv[2], O_RDONLY); groot::GzInflater *g = new groot::GzInflater(fd); for (ssize_t n; (n = g->read(buf.data(), buf.size())) > 0;) nb += (size_t)n; delete g; close(fd); }
        double t2 = now();
        printf("zlib %.3f s (%.0f MB/s), ours %.3f s (%.0f MB/s), %zu / %zu bytes\n", t1 - t0, na / (t1 - t0) / 1e6, t2 - t1, nb / (t2 - t1) / 1e6, na, nb);
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "--time")) {
        std::vector<uint8_t> a, b;
        std::string err;
        double t0 = now();
        with_zlib(argv[2], a);
        double t1 = now();
        with_ours(argv[2], b, 1 << 20, err);
        double t2 = now();
        printf("zlib %.3f s, ours %.3f s, %zu bytes, %s\n", t1 - t0, t2 - t1, b.size(), a == b ? "same" : "DIFFER");
        return 0;
    }
    const size_t chunk = argc > 2 ? (size_t)atol(argv[2]) : (1 << 20);
    std::vector<uint8_t> a, b;
    std::string err;
    const bool zok = with_zlib(argv[1], a);
    const int rc = with_ours(argv[1], b, chunk, err);
    if (rc) { printf("error: %s (zlib %s)\n", err.c_str(), zok ? "ok" : "error"); return 0; }
    if (a == b) printf("same %zu (zlib %s)\n", b.size(), zok ? "ok" : "error");
    else printf("differ: ours %zu bytes, zlib %zu bytes (zlib %s)\n", b.size(), a.size(), zok ? "ok" : "error");
    return 0;
}
