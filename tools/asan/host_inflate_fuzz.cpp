// AddressSanitizer / UBSan fuzz of the host gzip decoder (sicelore-2.1_amd/csrc/smi_inflate_host.hip), CPU build only:
//   tools/asan/run.sh
// (smi_internal.h next to this file is a stub without the HIP headers.)  6,000 streams: zlib output of four kinds of
// data at random level / strategy / memLevel, decoded as they are (must equal the source), with 1-4 flipped bits, cut at a random place, or
// into a buffer that is too small; input and output live in exact-size heap blocks so that any overrun is seen.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>
#include "smi_internal.h"
namespace smi { static std::string g_err; void set_error(const std::string &m) { g_err = m; } }
#include "smi_inflate_host.hip"
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }
int main() {
    size_t n_ok = 0, n_bad = 0, n_same = 0;
    for (int iter = 0; iter < 6000; iter++) {
        // a source text of mixed character
        size_t n = rnd() % 70000;
        std::vector<uint8_t> src(n);
        int kind = rnd() % 4;
        for (size_t i = 0; i < n; i++) src[i] = kind == 0 ? (uint8_t)rnd() : kind == 1 ? "ACGT"[rnd() & 3] : kind == 2 ? (uint8_t)(33 + rnd() % 40) : (uint8_t)((i / 7) & 0xFF);
        if (kind == 1 && n > 100) for (size_t i = 50; i < n; i += 97) std::memcpy(&src[i], &src[i - 50], std::min<size_t>(40, n - i));
        z_stream zs; std::memset(&zs, 0, sizeof zs);
        int level = rnd() % 10, strat = rnd() % 5, mem = 1 + rnd() % 9;
        deflateInit2(&zs, level, Z_DEFLATED, 31, mem, strat);
        std::vector<uint8_t> gz(n + n / 4 + 4096);
        zs.next_in = src.data(); zs.avail_in = n; zs.next_out = gz.data(); zs.avail_out = gz.size();
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { printf("harness: deflate buffer too small\n"); return 2; } gz.resize(zs.total_out); deflateEnd(&zs);
        // exact-size heap buffers so that ASan sees any overrun
        std::vector<uint8_t> in(gz), out(n);
        int mode = rnd() % 4;
        if (mode == 1 && in.size() > 20) for (int k = 0; k < 1 + (int)(rnd() % 4); k++) in[10 + rnd() % (in.size() - 18)] ^= (uint8_t)(1u << (rnd() & 7));
        if (mode == 2 && in.size() > 1) in.resize(rnd() % in.size());
        size_t cap = mode == 3 && n ? rnd() % n : n;
        out.resize(cap);
        uint8_t *ob = (uint8_t *)malloc(cap ? cap : 1);
        uint8_t *ib = (uint8_t *)malloc(in.size() ? in.size() : 1);
        std::memcpy(ib, in.data(), in.size());
        size_t ip = 0, op = 0;
        int rc = smi::host_gunzip(ib, in.size(), &ip, ob, cap, &op);
        if (rc == 0) { n_ok++; if (op == n && std::memcmp(ob, src.data(), n) == 0) n_same++; else if (mode == 0) { printf("MISMATCH iter %d\n", iter); return 1; } }
        else n_bad++;
        if (mode == 0 && rc != 0) { printf("FAILED on a valid stream iter %d: %s  n=%zu kind=%d level=%d strat=%d mem=%d gz=%zu\n", iter, smi::g_err.c_str(), n, kind, level, strat, mem, gz.size()); return 1; }
        free(ob); free(ib);
    }
    printf("ok %zu (identical %zu) refused %zu\n", n_ok, n_same, n_bad);
    return 0;
}
