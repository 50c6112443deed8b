// Per-64-byte-block cost of the host packer's primitives on this CPU, cache-resident (L2) and from DRAM: which forms are slow on the
// GPU box's EPYC (Zen 5) and which on the build container.  g++ -O3 -std=c++17 host_simd_probe.cpp -o probe && ./probe
#include <immintrin.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#define T512 __attribute__((target("avx512f,avx512bw,avx512vbmi")))
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
T512 static uint64_t enc_unmasked(const uint8_t *p, size_t n, uint64_t *out) {
    uint64_t acc = 0; size_t w = 0;
    for (size_t b = 0; b + 64 <= n; b += 64, w += 4) {
        const __m512i v = _mm512_and_si512(_mm512_loadu_si512(p + b), _mm512_set1_epi8((char)0xDF));
        const __mmask64 a = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('A')), g = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('G'));
        const __mmask64 c = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('C')), t = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('T'));
        const __mmask64 o = ~(a | g | c | t);
        out[w & 1023] = a | o; out[(w + 1) & 1023] = g | o; out[(w + 2) & 1023] = c | o; out[(w + 3) & 1023] = t | o;
    }
    return acc;
}
T512 static uint64_t enc_masked(const uint8_t *p, size_t n, uint64_t *out) {
    uint64_t acc = 0; size_t w = 0;
    for (size_t b = 0; b + 64 <= n; b += 64, w += 4) {
        const __mmask64 keep = ~0ull >> (b & 1);  // a mask the compiler cannot fold away
        const __m512i v = _mm512_and_si512(_mm512_maskz_loadu_epi8(keep, p + b), _mm512_set1_epi8((char)0xDF));
        const __mmask64 a = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('A')), g = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('G'));
        const __mmask64 c = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('C')), t = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('T'));
        const __mmask64 o = ~(a | g | c | t);
        out[w & 1023] = a | o; out[(w + 1) & 1023] = g | o; out[(w + 2) & 1023] = c | o; out[(w + 3) & 1023] = t | o;
    }
    return acc;
}
// the same planes through vpshufb-free arithmetic on 2 x 32 bytes with AVX2 movemask (no k registers)
__attribute__((target("avx2"))) static uint64_t enc_avx2(const uint8_t *p, size_t n, uint64_t *out) {
    size_t w = 0;
    const __m256i up = _mm256_set1_epi8((char)0xDF);
    for (size_t b = 0; b + 64 <= n; b += 64, w += 4) {
        uint64_t m[4] = {0, 0, 0, 0};
        for (int h = 0; h < 2; h++) {
            const __m256i v = _mm256_and_si256(_mm256_loadu_si256((const __m256i *)(p + b + 32 * h)), up);
            const uint32_t a = _mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('A'))), g = _mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('G')));
            const uint32_t c = _mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('C'))), t = _mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('T')));
            const uint32_t o = ~(a | g | c | t);
            m[0] |= (uint64_t)(a | o) << (32 * h); m[1] |= (uint64_t)(g | o) << (32 * h); m[2] |= (uint64_t)(c | o) << (32 * h); m[3] |= (uint64_t)(t | o) << (32 * h);
        }
        out[w & 1023] = m[0]; out[(w + 1) & 1023] = m[1]; out[(w + 2) & 1023] = m[2]; out[(w + 3) & 1023] = m[3];
    }
    return 0;
}
T512 static void copy_nt(const uint8_t *s, uint8_t *d, size_t n) { for (size_t k = 0; k + 64 <= n; k += 64) _mm512_stream_si512((__m512i *)(d + k), _mm512_loadu_si512(s + k)); }
T512 static void copy_masked(const uint8_t *s, uint8_t *d, size_t n) { for (size_t k = 0; k + 64 <= n; k += 64) _mm512_mask_storeu_epi8(d + k, ~0ull >> (k & 1), _mm512_loadu_si512(s + k)); }
T512 static void rev_rc(const uint8_t *s, uint8_t *d, size_t n, const uint8_t *tab) {
    alignas(64) uint8_t idx[64]; for (int i = 0; i < 64; i++) idx[i] = 63 - i;
    const __m512i rev = _mm512_load_si512(idx), lo = _mm512_loadu_si512(tab), hi = _mm512_loadu_si512(tab + 64);
    for (size_t k = 0; k + 64 <= n; k += 64) {
        __m512i v = _mm512_permutexvar_epi8(rev, _mm512_loadu_si512(s + n - 64 - k));
        v = _mm512_maskz_permutex2var_epi8(~_mm512_movepi8_mask(v), lo, v, hi);
        _mm512_storeu_si512(d + k, v);
    }
}
template <class F> static void run(const char *name, size_t bytes, int reps, F f) {
    f(); double best = 1e9;
    for (int r = 0; r < 3; r++) { double t0 = now(); for (int k = 0; k < reps; k++) f(); double dt = (now() - t0) / reps; if (dt < best) best = dt; }
    printf("%-34s %8.2f ns per 64-byte block  %7.2f GB/s\n", name, best / (bytes / 64.0) * 1e9, bytes / best / 1e9);
}
int main() {
    if (!__builtin_cpu_supports("avx512vbmi")) { puts("no avx512vbmi"); return 0; }
    std::vector<uint64_t> out(1024);
    uint8_t tab[128]; for (int i = 0; i < 128; i++) tab[i] = i == 'A' ? 'T' : i == 'C' ? 'G' : i == 'G' ? 'C' : i == 'T' ? 'A' : i == 'N' ? 'N' : 0;
    for (size_t bytes : {(size_t)256 << 10, (size_t)512 << 20}) {
        std::vector<uint8_t> src(bytes + 64), dst(bytes + 128);
        for (size_t i = 0; i < bytes; i++) src[i] = "ACGT"[(i * 2654435761u >> 7) & 3];
        uint8_t *d = dst.data() + (64 - ((uintptr_t)dst.data() & 63));
        const int reps = bytes < (1 << 24) ? 2000 : 2;
        printf("---- %zu KiB ----\n", bytes >> 10);
        run("encode, plain loads", bytes, reps, [&] { enc_unmasked(src.data(), bytes, out.data()); });
        run("encode, masked byte loads", bytes, reps, [&] { enc_masked(src.data(), bytes, out.data()); });
        run("encode, AVX2 movemask", bytes, reps, [&] { enc_avx2(src.data(), bytes, out.data()); });
        run("memchr (no hit)", bytes, reps, [&] { if (memchr(src.data(), '\n', bytes)) abort(); });
        run("memcpy", bytes, reps, [&] { memcpy(d, src.data(), bytes); });
        run("copy, streaming stores", bytes, reps, [&] { copy_nt(src.data(), d, bytes); });
        run("copy, masked byte stores", bytes, reps, [&] { copy_masked(src.data(), d, bytes); });
        run("reverse complement (vpermb x2)", bytes, reps, [&] { rev_rc(src.data(), d, bytes, tab); });
    }
    return 0;
}
