// smi_hostio.hip -- the host side of the packed boundary of `scanfastq` (no device code in this file).
//
// The text workers of smi_worker.hip ship 2.4 KB of FASTQ text per read up the link and 2.5 KB of finished text down: the link
// (57.5 GB/s per direction) then caps a host at ~14 M reads/s whatever the kernels do.  Here the host keeps the text and does the three
// byte-moving steps itself, on n_threads threads:
//   smi_fastq_index_host   record index of a chunk (htsjdk FastqReader rules; K-FQ's result)                 reads 2.4 KB / read
//   smi_pack_reads_host    bases -> four IUPAC bit-planes per read (K-PACKR's layout), 0.5 byte per base       reads 1.2, writes 0.6 KB
//   smi_pack_quals_host    pass 1: quality sum and the 224-quality tail (K-PACK's k_pack_quals)
//   smi_fastq_write_host   `passed` / `failed` records from the text + the decisions the device sent back       reads 2.4, writes 2.5 KB
// Reference units: FastqFileReader$OneFastqFileWorker (FJ!nanoporereadscanner/readerwriter/FastqFileReader.java:L138-167, htsjdk
// FastqReader), FastqRecordExt.getRecordForWriting (FastqRecordExt.java:L209-311), FastqWriterThreadPool$FastQoneFileThread.run
// (FastqWriterThreadPool.java:L300-306).  The formatter is the one the device writer uses (smi_name.h); the device kernels these functions
// stand in for (K-FQ, K-PACKR, K-PACK's quality half, K-WLEN / K-WNAME / K-WRITE) remain the specification: tests compare byte for byte.
//
// SIMD: AVX-512 (BW, + VBMI for the reverse complement) where the CPU has it, else AVX2, else plain C++ (runtime dispatch; the three
// forms are tested against each other through SMI_HOST_SIMD=0|1|2).
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "smi_internal.h"
#include "smi_name.h"

using namespace smi;

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// a team of threads for one call: spawned once, phases separated by a barrier (spin, then yield: the box may give this process
// fewer CPUs than threads)
// ---------------------------------------------------------------------------------------------------------------------------
class Team {
    int n_;
    std::atomic<int> arrived_{0};
    std::atomic<int> sense_{0};

public:
    explicit Team(int n) : n_(n) {}
    int size() const { return n_; }
    void barrier() {
        if (n_ == 1) return;
        const int s = sense_.load(std::memory_order_acquire);
        if (arrived_.fetch_add(1, std::memory_order_acq_rel) == n_ - 1) {
            arrived_.store(0, std::memory_order_relaxed);
            sense_.store(s + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (sense_.load(std::memory_order_acquire) == s) {
                if (++spins < 200)
                    _mm_pause();
                else
                    std::this_thread::yield();
            }
        }
    }
    template <class F>
    static void run(int n, F &&f) {
        Team team(n);
        std::vector<std::thread> th;
        th.reserve((size_t)n);
        for (int t = 1; t < n; t++) th.emplace_back([&team, &f, t] { f(t, team); });
        f(0, team);
        for (auto &x : th) x.join();
    }
};

inline void prefetch_range(const uint8_t *p, size_t n) {
    for (size_t o = 0; o < n; o += 64) _mm_prefetch(reinterpret_cast<const char *>(p + o), _MM_HINT_T0);
}

int clamp_threads(int n_threads, size_t work_items, size_t min_per_thread) {
    int t = n_threads > 0 ? n_threads : 1;
    if (t > 256) t = 256;
    const size_t by_work = work_items / (min_per_thread ? min_per_thread : 1);
    if ((size_t)t > by_work) t = by_work ? (int)by_work : 1;
    return t;
}

// 0 plain C++, 1 AVX2, 2 AVX-512 (BW; the reverse complement also needs VBMI)
int simd_level() {
    static const int cpu = [] {
        int lv = 0;
        if (__builtin_cpu_supports("avx2")) lv = 1;
        if (__builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("avx512vl")) lv = 2;
        return lv;
    }();
    int lv = cpu;
    if (const char *e = std::getenv("SMI_HOST_SIMD")) {  // testing: force a lower form (read per call, so one process can compare them)
        const int want = std::atoi(e);
        if (want >= 0 && want < lv) lv = want;
    }
    return lv;
}

// ---------------------------------------------------------------------------------------------------------------------------
// FASTQ index
// ---------------------------------------------------------------------------------------------------------------------------
inline const uint8_t *find_nl(const uint8_t *p, const uint8_t *end) {
    return p < end ? static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(end - p))) : nullptr;
}

// One record starting at text[p] (a line start).  Returns the position behind it (the next line start) or 0 when fewer than four lines
// remain (the caller decides what that means); *bad collects SMI_FQ_* bits exactly as k_fq_records does.
inline size_t parse_record(const uint8_t *text, size_t n, size_t p, smi_fastq_record *r, uint32_t *bad) {
    const uint8_t *end = text + n;
    size_t ls[4], le[4];  // line starts / ends (one past the last character, CR stripped)
    size_t cur = p;
    for (int k = 0; k < 4; k++) {
        if (cur >= n) return 0;  // a line that would start at the end of the text does not exist
        ls[k] = cur;
        const uint8_t *nl = find_nl(text + cur, end);
        size_t e = nl ? (size_t)(nl - text) : n;  // the last line may lack its newline
        cur = nl ? e + 1 : n;
        if (e > ls[k] && text[e - 1] == '\r') e--;
        le[k] = e;
    }
    if (le[0] == ls[0] || text[ls[0]] != '@') *bad |= SMI_FQ_BAD_SEQ_HEADER;
    if (le[2] == ls[2] || text[ls[2]] != '+') *bad |= SMI_FQ_BAD_QUAL_HEADER;
    if (le[1] - ls[1] != le[3] - ls[3]) *bad |= SMI_FQ_LENGTH_MISMATCH;
    r->name_start = ls[0] + 1;
    r->name_len = (uint32_t)(le[0] > ls[0] ? le[0] - ls[0] - 1 : 0);
    r->seq_start = ls[1];
    r->seq_len = (uint32_t)(le[1] - ls[1]);
    r->plus_start = ls[2] + 1;
    r->plus_len = (uint32_t)(le[2] > ls[2] ? le[2] - ls[2] - 1 : 0);
    r->qual_start = ls[3];
    r->reserved = 0;
    return cur;
}

// number of complete lines left from p on (a last line without newline counts): 0..3 are the only values the callers need
inline int lines_left_upto4(const uint8_t *text, size_t n, size_t p) {
    int k = 0;
    while (p < n && k < 4) {
        const uint8_t *nl = find_nl(text + p, text + n);
        k++;
        if (!nl) break;
        p = (size_t)(nl - text) + 1;
    }
    return k;
}

// A record start at or behind `from`, found without knowing the line parity: the first line L with text[L] == '@' whose line + 2 starts
// with '+' and whose line + 1 and line + 3 have equal lengths.  In well-formed FASTQ a quality line may start with '@' but a sequence
// line never starts with '+', so the guess is right; it is VERIFIED anyway (the thread in front must arrive exactly here), and any
// disagreement sends the whole chunk through the sequential parse.
size_t guess_record_start(const uint8_t *text, size_t n, size_t from) {
    const uint8_t *end = text + n;
    size_t p = from;
    if (p > 0) {  // move to a line start
        const uint8_t *nl = find_nl(text + p - 1, end);
        if (!nl) return n;
        p = (size_t)(nl - text) + 1;
    }
    for (int tries = 0; tries < 12 && p < n; tries++) {
        size_t ls[5];
        size_t cur = p;
        int k = 0;
        for (; k < 4 && cur < n; k++) {
            ls[k] = cur;
            const uint8_t *nl = find_nl(text + cur, end);
            cur = nl ? (size_t)(nl - text) + 1 : n + 1;
        }
        if (k < 4) return n;
        ls[4] = cur > n ? n + 1 : cur;
        auto len_of = [&](int j) {
            size_t e = ls[j + 1] - 1;
            if (e > ls[j] && e <= n && text[e - 1] == '\r') e--;
            return e - ls[j];
        };
        if (text[ls[0]] == '@' && text[ls[2]] == '+' && len_of(1) == len_of(3)) return p;
        p = ls[1];
    }
    return n;  // no candidate: this thread takes nothing, verification decides
}

}  // namespace

extern "C" int smi_fastq_index_host(const uint8_t *text, size_t n_bytes, smi_fastq_record *recs, uint64_t *offsets, size_t cap_records,
                                    size_t *n_records, uint32_t *errors, int n_threads) {
    if (!n_records || !errors || (n_bytes && (!text || !recs || !offsets))) {
        set_error("smi_fastq_index_host: null argument");
        return SMI_ERR_INVALID;
    }
    *n_records = 0;
    *errors = 0;
    if (offsets && cap_records + 1 > 0) offsets[0] = 0;
    if (!n_bytes) return SMI_OK;
    int nt = clamp_threads(n_threads, n_bytes, 1 << 20);
    bool overflow = false;
    // sequential parse from byte `from` with `have` records already stored; used alone (one thread) and as the fallback
    auto sequential = [&](size_t from, size_t have) -> int {
        size_t p = from, k = have;
        uint32_t bad = 0;
        while (p < n_bytes) {
            if (k >= cap_records) {
                overflow = true;
                break;
            }
            const size_t nx = parse_record(text, n_bytes, p, &recs[k], &bad);
            if (!nx) {
                if (lines_left_upto4(text, n_bytes, p) % 4) bad |= SMI_FQ_TRUNCATED;
                break;
            }
            k++;
            p = nx;
        }
        *errors |= bad;
        *n_records = k;
        return SMI_OK;
    };
    bool parallel_ok = nt > 1;
    if (parallel_ok) {
        // speculative split: thread t parses from a guessed record start behind t * n / T up to the next thread's guess, into its own
        // vector; accepted only when every thread arrives exactly at its successor's start
        std::vector<size_t> start((size_t)nt + 1, n_bytes);
        // (one cache line pair per thread: the vector headers are written with every record, neighbours must not share a line)
        struct alignas(128) Part : std::vector<smi_fastq_record> {};
        std::vector<Part> part((size_t)nt);
        std::vector<uint32_t> bad((size_t)nt, 0);
        std::vector<size_t> reached((size_t)nt, 0);
        std::vector<uint8_t> truncated((size_t)nt, 0);
        Team::run(nt, [&](int t, Team &team) {
            start[t] = t == 0 ? 0 : guess_record_start(text, n_bytes, n_bytes / (size_t)nt * (size_t)t);
            team.barrier();
            size_t p = start[t];
            size_t stop = n_bytes;
            for (int u = t + 1; u <= nt; u++)
                if (start[u] > p || u == nt) {
                    stop = start[u];
                    break;
                }
            auto &v = part[t];
            v.reserve((stop > p ? stop - p : 0) / 1500 + 16);
            uint32_t b = 0;
            while (p < stop) {
                smi_fastq_record r;
                const size_t nx = parse_record(text, n_bytes, p, &r, &b);
                if (!nx) {
                    truncated[t] = 1;
                    break;
                }
                v.push_back(r);
                p = nx;
            }
            reached[t] = p;
            bad[t] = b;
        });
        // verification: monotone starts, every thread stops exactly where the next non-empty one begins, no error anywhere
        size_t expect = 0, total = 0;
        for (int t = 0; t < nt && parallel_ok; t++) {
            if (bad[t] || truncated[t]) parallel_ok = false;
            if (part[t].empty()) continue;
            if (start[t] != expect) parallel_ok = false;
            expect = reached[t];
            total += part[t].size();
        }
        if (expect != n_bytes) parallel_ok = false;
        if (parallel_ok) {
            if (total > cap_records)
                overflow = true;
            else {
                std::vector<size_t> base((size_t)nt + 1, 0);
                for (int t = 0; t < nt; t++) base[t + 1] = base[t] + part[t].size();
                Team::run(nt, [&](int t, Team &) {
                    if (!part[t].empty()) std::memcpy(recs + base[t], part[t].data(), part[t].size() * sizeof(smi_fastq_record));
                });
                *n_records = total;
            }
        }
    }
    if (!parallel_ok && !overflow) sequential(0, 0);
    if (overflow) {
        set_error("smi_fastq_index_host: record buffers too small");
        return SMI_ERR_INVALID;
    }
    // prefix sums of the read lengths (two levels when threaded)
    const size_t n = *n_records;
    nt = clamp_threads(n_threads, n, 1 << 16);
    if (nt == 1) {
        uint64_t acc = 0;
        for (size_t r = 0; r < n; r++) {
            offsets[r] = acc;
            acc += recs[r].seq_len;
        }
        offsets[n] = acc;
    } else {
        std::vector<uint64_t> tot((size_t)nt + 1, 0);
        Team::run(nt, [&](int t, Team &team) {
            const size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
            uint64_t acc = 0;
            for (size_t r = lo; r < hi; r++) acc += recs[r].seq_len;
            tot[t + 1] = acc;
            team.barrier();
            uint64_t base = 0;
            for (int u = 1; u <= t; u++) base += tot[u];
            for (size_t r = lo; r < hi; r++) {
                offsets[r] = base;
                base += recs[r].seq_len;
            }
            if (t == nt - 1) offsets[n] = base;
        });
    }
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// planes
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

// 4-bit IUPAC code of a character as enc4x4 (smi_internal.h) defines it: A 1, G 2, C 4, T 8 (either case), anything else 15
inline uint32_t code4_of(uint8_t c) {
    switch (c & 0xDF) {
    case 'A': return 1;
    case 'G': return 2;
    case 'C': return 4;
    case 'T': return 8;
    default: return 15;
    }
}

// plane words of `nb` (1..64) bases at p: four 64-bit masks, bit b = base b
inline void planes64_scalar(const uint8_t *p, int nb, uint64_t (&pl)[4]) {
    pl[0] = pl[1] = pl[2] = pl[3] = 0;
    for (int b = 0; b < nb; b++) {
        const uint32_t c = code4_of(p[b]);
        for (int k = 0; k < 4; k++) pl[k] |= (uint64_t)((c >> k) & 1u) << b;
    }
}

__attribute__((target("avx2"))) inline void planes64_avx2(const uint8_t *p, int nb, uint64_t (&pl)[4]) {
    // two 32-byte halves; a partial block goes through a zero-padded copy (zero bytes encode as N: masked off below)
    alignas(32) uint8_t tmp[64];
    const uint8_t *q = p;
    if (nb < 64) {
        std::memset(tmp, 0, 64);
        std::memcpy(tmp, p, (size_t)nb);
        q = tmp;
    }
    const __m256i up = _mm256_set1_epi8((char)0xDF);
    uint64_t m[4] = {0, 0, 0, 0};
    for (int h = 0; h < 2; h++) {
        const __m256i v = _mm256_and_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(q + 32 * h)), up);
        const uint32_t a = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('A')));
        const uint32_t g = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('G')));
        const uint32_t c = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('C')));
        const uint32_t t = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('T')));
        const uint32_t other = ~(a | g | c | t);
        m[0] |= (uint64_t)(a | other) << (32 * h);
        m[1] |= (uint64_t)(g | other) << (32 * h);
        m[2] |= (uint64_t)(c | other) << (32 * h);
        m[3] |= (uint64_t)(t | other) << (32 * h);
    }
    const uint64_t keep = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
    for (int k = 0; k < 4; k++) pl[k] = m[k] & keep;
}

__attribute__((target("avx512f,avx512bw"))) inline void planes64_avx512(const uint8_t *p, int nb, uint64_t (&pl)[4]) {
    const __mmask64 keep = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
    // (a masked byte load costs several times a plain one on Zen 4 / 5: only the last, partial block of a read takes it)
    const __m512i v = _mm512_and_si512(nb >= 64 ? _mm512_loadu_si512(p) : _mm512_maskz_loadu_epi8(keep, p), _mm512_set1_epi8((char)0xDF));
    const __mmask64 a = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('A'));
    const __mmask64 g = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('G'));
    const __mmask64 c = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('C'));
    const __mmask64 t = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('T'));
    const __mmask64 other = ~(a | g | c | t) & keep;
    pl[0] = a | other;
    pl[1] = g | other;
    pl[2] = c | other;
    pl[3] = t | other;
}

// one read: ceil(len / 32) data words + the pad words, all written (K-PACKR writes the pad words as zeros too); one function per SIMD
// form so that the 64-base encoder is inlined into code compiled for the same target
#define SMI_DEFINE_PACK_RANGE(NAME, TARGET_ATTR, PLANES64)                                                                              \
    TARGET_ATTR void NAME(const uint8_t *text, const smi_fastq_record *recs, const uint64_t *offsets, size_t lo, size_t hi,             \
                          uint32_t *planes, size_t stride) {                                                                            \
        for (size_t r = lo; r < hi; r++) {                                                                                              \
            const uint8_t *seq = text + recs[r].seq_start;                                                                              \
            const int64_t len = (int64_t)recs[r].seq_len;                                                                               \
            uint32_t *p0 = planes + plane_start(offsets[r], r);                                                                         \
            const int64_t n_words = (len + 31) / 32 + kReadPadWords - 1;                                                                \
            int64_t w = 0;                                                                                                              \
            for (int64_t b = 0; b < len; b += 64, w += 2) {                                                                             \
                const int nb = (int)(len - b < 64 ? len - b : 64);                                                                      \
                uint64_t pl[4];                                                                                                         \
                PLANES64(seq + b, nb, pl);                                                                                              \
                for (int c = 0; c < 4; c++) std::memcpy(p0 + (size_t)c * stride + w, &pl[c], 8); /* >= 4 pad words follow the data */   \
            }                                                                                                                           \
            for (int64_t z = w; z < n_words; z++)                                                                                       \
                for (int c = 0; c < 4; c++) p0[(size_t)c * stride + z] = 0;                                                             \
        }                                                                                                                               \
    }
SMI_DEFINE_PACK_RANGE(pack_range_scalar, , planes64_scalar)
SMI_DEFINE_PACK_RANGE(pack_range_avx2, __attribute__((target("avx2"))), planes64_avx2)
SMI_DEFINE_PACK_RANGE(pack_range_avx512, __attribute__((target("avx512f,avx512bw"))), planes64_avx512)

}  // namespace

extern "C" int smi_pack_reads_host(const uint8_t *text, const smi_fastq_record *recs, const uint64_t *offsets, size_t n, uint32_t *planes,
                                   int n_threads) {
    if (n && (!text || !recs || !offsets || !planes)) {
        set_error("smi_pack_reads_host: null argument");
        return SMI_ERR_INVALID;
    }
    if (!n) return SMI_OK;
    const size_t stride = read_planes_stride(offsets[n], n);
    const int nt = clamp_threads(n_threads, n, 256);
    const int level = simd_level();
    Team::run(nt, [&](int t, Team &) {
        const size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
        if (level == 2)
            pack_range_avx512(text, recs, offsets, lo, hi, planes, stride);
        else if (level == 1)
            pack_range_avx2(text, recs, offsets, lo, hi, planes, stride);
        else
            pack_range_scalar(text, recs, offsets, lo, hi, planes, stride);
        // the words no read owns (gaps of the formula's rounding, the slack behind the last read): zero, so that the buffer is
        // defined everywhere the device may look (K-PACKR leaves them as the caller's memset left them)
        for (size_t r = lo; r < hi; r++) {
            const size_t used_end = plane_start(offsets[r], r) + (size_t)((recs[r].seq_len + 31) / 32 + kReadPadWords - 1);
            const size_t next = r + 1 < n ? plane_start(offsets[r + 1], r + 1) : stride;
            for (size_t z = used_end; z < next; z++)
                for (int c = 0; c < 4; c++) planes[(size_t)c * stride + z] = 0;
        }
    });
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// index + planes in ONE pass over the text (smi_fastq_index_pack_host)
//
// Indexing first and packing afterwards reads every sequence twice, the second time in 1 KB bursts that the hardware prefetcher
// cannot follow (measured on the GPU box: 2.3 GB/s per thread against 21 GB/s for the index pass).  Here every thread walks its
// share of the text once, record by record: name line (memchr), sequence line in 64-byte blocks that are encoded while the line end is
// looked for, '+' line (memchr), and the quality line is STEPPED OVER -- its length is known, only the character behind it is checked;
// that no line end hides inside it is checked by whoever reads the qualities later (recs[r].reserved = 1 asks for it:
// smi_fastq_write_host, smi_pack_quals_host).  A thread cannot know where its planes belong in K-PACKR's layout before the threads in
// front of it are done, so the planes are written in segments, one per thread, and every read carries its word offset (pstart).
// Anything unexpected -- a malformed record, a segment that overflows, threads that do not meet -- sends the chunk through
// smi_fastq_index_host + smi_pack_reads_host, which is exact by construction.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

struct alignas(128) FusedThread {  // written with every record: no two threads' state in one cache line
    std::vector<smi_fastq_record> recs;
    std::vector<uint32_t> wstart;  // word offset of the read inside the thread's segment
    size_t words = 0, reached = 0, bases = 0;
    bool ok = true;
};

// one 64-byte block at p (avail >= 1 bytes exist): plane masks of all 64 positions and the mask of '\n' bytes
inline void block64_scalar(const uint8_t *p, size_t avail, uint64_t (&pl)[4], uint64_t *nl) {
    const int nb = (int)(avail < 64 ? avail : 64);
    planes64_scalar(p, nb, pl);
    uint64_t m = 0;
    for (int b = 0; b < nb; b++) m |= (uint64_t)(p[b] == '\n') << b;
    *nl = m;
}
__attribute__((target("avx2"))) inline void block64_avx2(const uint8_t *p, size_t avail, uint64_t (&pl)[4], uint64_t *nl) {
    const int nb = (int)(avail < 64 ? avail : 64);
    planes64_avx2(p, nb, pl);
    alignas(32) uint8_t tmp[64];
    const uint8_t *q = p;
    if (nb < 64) {
        std::memset(tmp, 0, 64);
        std::memcpy(tmp, p, (size_t)nb);
        q = tmp;
    }
    const __m256i lf = _mm256_set1_epi8('\n');
    const uint32_t lo = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(q)), lf));
    const uint32_t hi = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(q + 32)), lf));
    *nl = (uint64_t)lo | ((uint64_t)hi << 32);
}
__attribute__((target("avx512f,avx512bw"))) inline void block64_avx512(const uint8_t *p, size_t avail, uint64_t (&pl)[4], uint64_t *nl) {
    const __mmask64 keep = avail >= 64 ? ~0ull : ((1ull << avail) - 1ull);
    const __m512i raw = avail >= 64 ? _mm512_loadu_si512(p) : _mm512_maskz_loadu_epi8(keep, p);  // masked byte loads are slow on Zen 4 / 5
    *nl = _mm512_cmpeq_epi8_mask(raw, _mm512_set1_epi8('\n'));
    const __m512i v = _mm512_and_si512(raw, _mm512_set1_epi8((char)0xDF));
    const __mmask64 a = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('A'));
    const __mmask64 g = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('G'));
    const __mmask64 c = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('C'));
    const __mmask64 t = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('T'));
    const __mmask64 other = ~(a | g | c | t) & keep;
    pl[0] = a | other;
    pl[1] = g | other;
    pl[2] = c | other;
    pl[3] = t | other;
}

// records from byte p up to (not across) `stop`; planes into the thread's segment seg[c * stride .. ) of seg_cap words
#define SMI_DEFINE_FUSED_RANGE(NAME, TARGET_ATTR, BLOCK64)                                                                                \
    TARGET_ATTR void NAME(const uint8_t *text, size_t n, size_t p, size_t stop, uint32_t *seg, size_t stride, size_t seg_cap, FusedThread &T) { \
        const uint8_t *end = text + n;                                                                                                    \
        size_t lw = 0;                                                                                                                    \
        while (p < stop) {                                                                                                                \
            smi_fastq_record r;                                                                                                           \
            /* line 0: '@' name */                                                                                                        \
            const uint8_t *nl0 = find_nl(text + p, end);                                                                                  \
            if (!nl0) { T.ok = false; break; }                                                                                            \
            size_t e0 = (size_t)(nl0 - text);                                                                                             \
            const size_t l1 = e0 + 1;                                                                                                     \
            if (e0 > p && text[e0 - 1] == '\r') e0--;                                                                                     \
            if (e0 == p || text[p] != '@' || l1 >= n) { T.ok = false; break; }                                                            \
            r.name_start = p + 1;                                                                                                         \
            r.name_len = (uint32_t)(e0 - p - 1);                                                                                          \
            /* line 1: bases, encoded while the line end is looked for */                                                                 \
            size_t b = 0;                                                                                                                 \
            bool found = false;                                                                                                           \
            for (;;) {                                                                                                                    \
                const size_t at = l1 + b;                                                                                                 \
                if (at >= n) break;                                                                                                       \
                if (lw + (b >> 5) + 2 + kReadPadWords > seg_cap) { T.ok = false; break; }                                                 \
                uint64_t pl[4], nlm;                                                                                                      \
                BLOCK64(text + at, n - at, pl, &nlm);                                                                                     \
                if (__builtin_expect(nlm == 0 && n - at >= 64, 1)) {                                                                      \
                    /* the usual block: 64 bases, no line end.  A BRANCH, not arithmetic on the mask: the address of the next block must  \
                       not wait for this block's compare, or every cache miss of the line is taken one after the other */                 \
                    for (int c = 0; c < 4; c++) std::memcpy(seg + (size_t)c * stride + lw + (b >> 5), &pl[c], 8);                         \
                    b += 64;                                                                                                              \
                    continue;                                                                                                             \
                }                                                                                                                         \
                const size_t avail = n - at < 64 ? n - at : 64;                                                                           \
                const int nb = nlm ? (int)__builtin_ctzll(nlm) : (int)avail;                                                              \
                const uint64_t keep = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);                                                           \
                for (int c = 0; c < 4; c++) {                                                                                             \
                    const uint64_t v = pl[c] & keep;                                                                                      \
                    std::memcpy(seg + (size_t)c * stride + lw + (b >> 5), &v, 8);                                                         \
                }                                                                                                                         \
                b += (size_t)nb;                                                                                                          \
                if (nlm) { found = true; break; }                                                                                         \
                if (avail < 64) break; /* the text ends inside the sequence line */                                                       \
            }                                                                                                                             \
            if (!T.ok || !found) { T.ok = false; break; }                                                                                 \
            const size_t l2 = l1 + b + 1;                                                                                                 \
            size_t len = b;                                                                                                               \
            if (len > 0 && text[l1 + len - 1] == '\r') { /* CR LF: the CR was encoded as a base (N): take it out again */                  \
                len--;                                                                                                                    \
                for (int c = 0; c < 4; c++) seg[(size_t)c * stride + lw + (len >> 5)] &= ~(1u << (len & 31));                             \
            }                                                                                                                             \
            const size_t n_words = (len + 31) / 32 + kReadPadWords - 1;                                                                   \
            for (size_t z = (b + 63) / 64 * 2; z < n_words; z++)                                                                          \
                for (int c = 0; c < 4; c++) seg[(size_t)c * stride + lw + z] = 0;                                                         \
            r.seq_start = l1;                                                                                                             \
            r.seq_len = (uint32_t)len;                                                                                                    \
            /* line 2: '+' */                                                                                                             \
            if (l2 >= n || text[l2] != '+') { T.ok = false; break; }                                                                      \
            const uint8_t *nl2 = find_nl(text + l2, end);                                                                                 \
            if (!nl2) { T.ok = false; break; }                                                                                            \
            size_t e2 = (size_t)(nl2 - text);                                                                                             \
            const size_t l3 = e2 + 1;                                                                                                     \
            if (e2 > l2 && text[e2 - 1] == '\r') e2--;                                                                                    \
            r.plus_start = l2 + 1;                                                                                                        \
            r.plus_len = (uint32_t)(e2 - l2 - 1);                                                                                         \
            /* line 3: qualities, stepped over; what follows must be a line end (or the end of the text) */                               \
            const size_t e3 = l3 + len;                                                                                                   \
            size_t next;                                                                                                                  \
            if (e3 == n) {                                                                                                                \
                if (len > 0 && text[n - 1] == '\n') { T.ok = false; break; } /* that line end belongs to a shorter quality line */          \
                next = n;                                                                                                                 \
            }                                                                                                                             \
            else if (e3 < n && text[e3] == '\n')                                                                                          \
                next = e3 + 1;                                                                                                            \
            else if (e3 + 1 < n && text[e3] == '\r' && text[e3 + 1] == '\n')                                                              \
                next = e3 + 2;                                                                                                            \
            else if (e3 + 1 == n && text[e3] == '\r')                                                                                     \
                next = n;                                                                                                                 \
            else { T.ok = false; break; }                                                                                                 \
            if (l3 >= n) { T.ok = false; break; } /* a line that starts at the end of the text does not exist */                          \
            /* the next record's first lines: its name and the head of its bases (the hardware prefetcher lost the stream at the jump) */  \
            { const size_t ahead = len + 192 < 4096 ? len + 192 : 4096; /* its bases are about as long as this record's */                \
              for (size_t o = 0; o < ahead; o += 64) _mm_prefetch(reinterpret_cast<const char *>(text + (next + o < n ? next + o : n - 1)), _MM_HINT_T0); } \
            r.qual_start = l3;                                                                                                            \
            r.reserved = 1; /* the quality line has not been looked at: whoever reads it checks it for line ends */                       \
            T.recs.push_back(r);                                                                                                          \
            T.wstart.push_back((uint32_t)lw);                                                                                             \
            lw += n_words;                                                                                                                \
            T.bases += len;                                                                                                               \
            p = next;                                                                                                                     \
        }                                                                                                                                 \
        T.words = lw;                                                                                                                     \
        T.reached = p;                                                                                                                    \
    }
SMI_DEFINE_FUSED_RANGE(fused_range_scalar, , block64_scalar)
SMI_DEFINE_FUSED_RANGE(fused_range_avx2, __attribute__((target("avx2"))), block64_avx2)
SMI_DEFINE_FUSED_RANGE(fused_range_avx512, __attribute__((target("avx512f,avx512bw"))), block64_avx512)

// words per plane of one thread's segment: sequence <= half of the bytes, a pad of five words per 512 bytes of text, and room for one very
// long read that straddles the thread's nominal share
size_t fused_seg_cap(size_t bytes) { return bytes / 64 + kReadPadWords * (bytes / 512) + 65536; }

}  // namespace

extern "C" size_t smi_packed_planes_words(size_t n_bytes, int n_threads) {
    const int nt = clamp_threads(n_threads, n_bytes, 1 << 20);
    // segments of the fused packer, and never less than the two-step packer needs for records of 64 bytes or more
    const size_t fused = (size_t)nt * fused_seg_cap(n_bytes / (size_t)nt + 1);
    const size_t two_step = read_planes_stride(n_bytes / 2, n_bytes / 64 + 1);
    return 4 * std::max(fused, two_step);
}

extern "C" int smi_fastq_index_pack_host(const uint8_t *text, size_t n_bytes, smi_fastq_record *recs, uint64_t *offsets, uint32_t *pstart,
                                         size_t cap_records, uint32_t *planes, size_t planes_words, smi_packed_reads *packed, size_t *n_records,
                                         uint32_t *errors, int n_threads) {
    if (!n_records || !errors || !packed || (n_bytes && (!text || !recs || !offsets || !pstart || !planes))) {
        set_error("smi_fastq_index_pack_host: null argument");
        return SMI_ERR_INVALID;
    }
    *n_records = 0;
    *errors = 0;
    std::memset(packed, 0, sizeof *packed);
    if (offsets) offsets[0] = 0;
    if (!n_bytes) return SMI_OK;
    const int nt = clamp_threads(n_threads, n_bytes, 1 << 20);
    const size_t stride = planes_words / 4;
    const size_t seg_cap = fused_seg_cap(n_bytes / (size_t)nt + 1);
    bool fused_ok = stride >= (size_t)nt * seg_cap && nt <= SMI_PACKED_MAX_SEGMENTS;
    std::vector<FusedThread> TT((size_t)nt);
    std::vector<size_t> start((size_t)nt + 1, n_bytes);
    if (fused_ok) {
        const int level = simd_level();
        Team::run(nt, [&](int t, Team &team) {
            start[t] = t == 0 ? 0 : guess_record_start(text, n_bytes, n_bytes / (size_t)nt * (size_t)t);
            team.barrier();
            size_t p = start[t], stop = n_bytes;
            for (int u = t + 1; u <= nt; u++)
                if (start[u] > p || u == nt) {
                    stop = start[u];
                    break;
                }
            FusedThread &T = TT[t];
            T.recs.reserve((stop > p ? stop - p : 0) / 1500 + 16);
            T.wstart.reserve(T.recs.capacity());
            uint32_t *seg = planes + (size_t)t * seg_cap;
            if (level == 2)
                fused_range_avx512(text, n_bytes, p, stop, seg, stride, seg_cap, T);
            else if (level == 1)
                fused_range_avx2(text, n_bytes, p, stop, seg, stride, seg_cap, T);
            else
                fused_range_scalar(text, n_bytes, p, stop, seg, stride, seg_cap, T);
        });
        size_t expect = 0, total = 0;
        for (int t = 0; t < nt && fused_ok; t++) {
            if (!TT[t].ok) fused_ok = false;
            if (TT[t].recs.empty()) continue;
            if (start[t] != expect) fused_ok = false;
            expect = TT[t].reached;
            total += TT[t].recs.size();
        }
        if (expect != n_bytes || total > cap_records) fused_ok = false;
        if (fused_ok) {
            // compact (device) layout: the segments back to back
            std::vector<size_t> rbase((size_t)nt + 1, 0);
            std::vector<uint64_t> obase((size_t)nt + 1, 0), wbase((size_t)nt + 1, 0);
            int n_seg = 0;
            for (int t = 0; t < nt; t++) {
                rbase[t + 1] = rbase[t] + TT[t].recs.size();
                obase[t + 1] = obase[t] + TT[t].bases;
                wbase[t + 1] = wbase[t] + TT[t].words;
                if (TT[t].words) {
                    packed->seg_host_word[n_seg] = (uint64_t)t * seg_cap;
                    packed->seg_dev_word[n_seg] = wbase[t];
                    packed->seg_words[n_seg] = TT[t].words;
                    n_seg++;
                }
            }
            if (wbase[nt] + 8 > 0xFFFFFFFFull) fused_ok = false;  // pstart is 32 bits
            if (fused_ok) {
                Team::run(nt, [&](int t, Team &) {
                    const FusedThread &T = TT[t];
                    if (T.recs.empty()) return;
                    std::memcpy(recs + rbase[t], T.recs.data(), T.recs.size() * sizeof(smi_fastq_record));
                    uint64_t acc = obase[t];
                    for (size_t j = 0; j < T.recs.size(); j++) {
                        offsets[rbase[t] + j] = acc;
                        acc += T.recs[j].seq_len;
                        pstart[rbase[t] + j] = (uint32_t)(wbase[t] + T.wstart[j]);
                    }
                });
                offsets[total] = obase[nt];
                packed->planes = planes;
                packed->stride = stride;
                packed->pstart = pstart;
                packed->n_seg = n_seg;
                packed->total_words = (size_t)wbase[nt] + 8;  // a few words of slack behind the last read, as K-PACKR's layout has
                *n_records = total;
                return SMI_OK;
            }
        }
    }
    // the exact two-step path
    if (int rc = smi_fastq_index_host(text, n_bytes, recs, offsets, cap_records, n_records, errors, n_threads)) return rc;
    if (*errors || *n_records == 0) return SMI_OK;
    const size_t n = *n_records;
    const size_t need = smi_read_planes_words(offsets[n], n);
    if (need > planes_words) {
        set_error("smi_fastq_index_pack_host: plane buffer too small for this chunk (records shorter than 64 bytes): size it with smi_read_planes_words");
        return SMI_ERR_INVALID;
    }
    if (int rc = smi_pack_reads_host(text, recs, offsets, n, planes, n_threads)) return rc;
    packed->planes = planes;
    packed->stride = need / 4;
    packed->pstart = nullptr;
    packed->n_seg = 1;
    packed->seg_host_word[0] = 0;
    packed->seg_dev_word[0] = 0;
    packed->seg_words[0] = need / 4;
    packed->total_words = need / 4;
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// qualities of pass 1
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
inline uint64_t byte_sum_scalar(const uint8_t *p, size_t n) {
    uint64_t s = 0;
    for (size_t i = 0; i < n; i++) s += p[i];
    return s;
}
__attribute__((target("avx2"))) inline uint64_t byte_sum_avx2(const uint8_t *p, size_t n) {
    __m256i acc = _mm256_setzero_si256();
    size_t i = 0;
    for (; i + 32 <= n; i += 32) acc = _mm256_add_epi64(acc, _mm256_sad_epu8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i)), _mm256_setzero_si256()));
    alignas(32) uint64_t l[4];
    _mm256_store_si256(reinterpret_cast<__m256i *>(l), acc);
    return l[0] + l[1] + l[2] + l[3] + byte_sum_scalar(p + i, n - i);
}
__attribute__((target("avx512f,avx512bw"))) inline uint64_t byte_sum_avx512(const uint8_t *p, size_t n) {
    __m512i acc = _mm512_setzero_si512();
    size_t i = 0;
    for (; i + 64 <= n; i += 64) acc = _mm512_add_epi64(acc, _mm512_sad_epu8(_mm512_loadu_si512(p + i), _mm512_setzero_si512()));
    if (i < n) acc = _mm512_add_epi64(acc, _mm512_sad_epu8(_mm512_maskz_loadu_epi8((1ull << (n - i)) - 1ull, p + i), _mm512_setzero_si512()));
    return (uint64_t)_mm512_reduce_add_epi64(acc);
}
}  // namespace

extern "C" int smi_pack_quals_host(const uint8_t *text, const smi_fastq_record *recs, size_t n, int five_prime, uint8_t *qtail, uint32_t *qsum,
                                   int n_threads) {
    if (n && (!text || !recs || !qtail || !qsum)) {
        set_error("smi_pack_quals_host: null argument");
        return SMI_ERR_INVALID;
    }
    if (!n) return SMI_OK;
    const int nt = clamp_threads(n_threads, n, 256);
    const int level = simd_level();
    std::atomic<bool> bad_nl{false};
    Team::run(nt, [&](int t, Team &) {
        const size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
        for (size_t r = lo; r < hi; r++) {
            if (r + 2 < hi) prefetch_range(text + recs[r + 2].qual_start, recs[r + 2].seq_len);
            const uint8_t *q = text + recs[r].qual_start;
            const size_t len = recs[r].seq_len;
            if (recs[r].reserved && len && std::memchr(q, '\n', len)) bad_nl.store(true);  // stepped over by the one-pass index: checked here
            const uint64_t s = level == 2 ? byte_sum_avx512(q, len) : (level == 1 ? byte_sum_avx2(q, len) : byte_sum_scalar(q, len));
            qsum[r] = (uint32_t)s - 33u * (uint32_t)len;  // u32 arithmetic as on the device
            uint8_t *o = qtail + r * (size_t)SMI_END_BASES;
            if (len >= (size_t)SMI_END_BASES)
                std::memcpy(o, five_prime ? q : q + (len - SMI_END_BASES), SMI_END_BASES);
            else if (five_prime) {
                std::memcpy(o, q, len);
                std::memset(o + len, 33, SMI_END_BASES - len);
            } else {
                std::memset(o, 33, SMI_END_BASES - len);
                std::memcpy(o + (SMI_END_BASES - len), q, len);
            }
        }
    });
    if (bad_nl.load()) {
        set_error("smi_pack_quals_host: malformed FASTQ, a line end inside a quality string (htsjdk's FastqReader throws here)");
        return SMI_ERR_INVALID;
    }
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// record writer
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kSuffixCapHost = 1024;  // as kSuffixCap of the device writer

struct HostPlan {           // RecPlan of smi_write.hip
    uint64_t name_beg, qh_beg, rd, ql;
    uint32_t name_tok_len, qh_len;
    int32_t len, cut_beg, cut_len;
    uint32_t src;
    int32_t frag;
    bool passed, rev, forced_failed, had_blank;
};

inline const char *split_tag_host(int reason) {
    switch (reason) {
    case SMI_SPLIT_FWD_ADAPTER: return "FA";
    case SMI_SPLIT_RA_FA: return "RA_FA";
    case SMI_SPLIT_RA_FT: return "RA_FT";
    case SMI_SPLIT_RT_FA: return "RT_FA";
    case SMI_SPLIT_RT_FT: return "RT_FT";
    default: return "RA";
    }
}

struct WriteJob {
    const uint8_t *text;
    const smi_fastq_record *recs;
    const uint64_t *offsets;
    const smi_pass2_decisions *dec;
    uint32_t first_read_id;
    int five_prime, trim;
};

inline bool record_passed(const WriteJob &J, size_t i) {
    const smi_pass2_decisions &D = *J.dec;
    const uint32_t src = D.frag_src ? (D.frag_src[i] >> 2) : (uint32_t)i;
    const bool forced = D.chim && (D.chim[src].flags & SMI_CHIM_MULTI);
    return !forced && (D.scan[i].flags & (SMI_F_PASSED_FWD | SMI_F_PASSED_REV));
}

inline HostPlan plan_record_host(const WriteJob &J, size_t i) {  // plan_record of smi_write.hip
    const smi_pass2_decisions &D = *J.dec;
    HostPlan R;
    const uint32_t fs = D.frag_src ? D.frag_src[i] : (uint32_t)(i << 2);
    R.src = D.frag_src ? (fs >> 2) : (uint32_t)i;
    const smi_chimera_result *ch = D.chim ? D.chim + R.src : nullptr;
    R.frag = (ch && ch->n_split) ? (int)(fs & 3u) : -1;
    R.forced_failed = ch && (ch->flags & SMI_CHIM_MULTI);
    const smi_fastq_record &rec = J.recs[R.src];
    R.name_beg = rec.name_start;
    const uint8_t *nm = J.text + rec.name_start;
    const void *blank = rec.name_len ? std::memchr(nm, ' ', rec.name_len) : nullptr;
    R.name_tok_len = blank ? (uint32_t)(static_cast<const uint8_t *>(blank) - nm) : rec.name_len;
    R.had_blank = blank != nullptr;
    R.qh_beg = rec.plus_start;
    R.qh_len = rec.plus_len;
    const uint64_t base = D.frag_offsets[i];
    const uint64_t in_read = base - J.offsets[R.src];
    R.rd = rec.seq_start + in_read;
    R.ql = rec.qual_start + in_read;
    R.len = (int32_t)(D.frag_offsets[i + 1] - base);
    const smi_scan_result &sc = D.scan[i];
    R.passed = !R.forced_failed && (sc.flags & (SMI_F_PASSED_FWD | SMI_F_PASSED_REV));
    R.rev = R.passed && (sc.flags & SMI_F_PASSED_REV);
    R.cut_beg = 0;
    R.cut_len = R.len;
    const smi_bc_result &b = D.bc[i];
    if (R.passed && J.trim && b.found == 1) {
        const int bc_start = sc.adapter_end + 1 + b.offset;
        const int beg = J.five_prime ? bc_start + 30 : (sc.tso_end != 0 ? sc.tso_end : 1);
        const int end = sc.polya_end != 0 ? sc.polya_start : R.len;
        if (beg < end) {
            R.cut_beg = std::min(std::max(beg - 1, 0), R.len);
            R.cut_len = std::max(std::min(end, R.len) - R.cut_beg, 0);
        }
    }
    return R;
}

// The writer's own sink: the scratch behind it always has room for the longest suffix there is (< 300 characters: every field is a
// bounded integer, 16 + 16 barcode letters, 43 window letters), so nothing is checked per character; integers two digits per step, barcodes
// four letters per step.  Same interface as NameSink, so it runs through the same append_name_suffix as the device writer.
struct DigitPairs {
    char t[200];
    DigitPairs() {
        for (int i = 0; i < 100; i++) t[2 * i] = (char)('0' + i / 10), t[2 * i + 1] = (char)('0' + i % 10);
    }
};
const DigitPairs g_digit_pairs;
struct KmerQuads {  // TWOBIT_TO_BASE_ARRAY (A G C T), four bases per byte of the key, first base in the top bits
    uint32_t t[256];
    KmerQuads() {
        const char L[4] = {'A', 'G', 'C', 'T'};
        for (int b = 0; b < 256; b++) {
            char c[4] = {L[(b >> 6) & 3], L[(b >> 4) & 3], L[(b >> 2) & 3], L[b & 3]};
            std::memcpy(&t[b], c, 4);
        }
    }
};
const KmerQuads g_kmer_quads;

struct HostSink {
    static constexpr bool kFastKmer = true;
    char *p;
    int n;
    inline void put(char c) { p[n++] = c; }
    inline void puts(const char *s) {
        const size_t k = __builtin_strlen(s);  // a constant for the literals this is called with
        std::memcpy(p + n, s, k);
        n += (int)k;
    }
    inline void put_u32(uint32_t v) {
        if (v < 10u) {
            p[n++] = (char)('0' + v);
            return;
        }
        char t[10];
        int k = 10;
        while (v >= 100u) {
            const uint32_t q = v / 100u, r = v - q * 100u;
            k -= 2;
            std::memcpy(t + k, g_digit_pairs.t + 2 * r, 2);
            v = q;
        }
        if (v >= 10u) {
            k -= 2;
            std::memcpy(t + k, g_digit_pairs.t + 2 * v, 2);
        } else
            t[--k] = (char)('0' + v);
        std::memcpy(p + n, t + k, 10 - k);
        n += 10 - k;
    }
    inline void put_i32(int v) {
        if (v < 0) {
            p[n++] = '-';
            put_u32(0u - (uint32_t)v);
        } else
            put_u32((uint32_t)v);
    }
    inline void put_u64(unsigned long long v) {
        char t[20];
        int k = 20;
        do {
            t[--k] = (char)('0' + (int)(v % 10));
            v /= 10;
        } while (v);
        std::memcpy(p + n, t + k, 20 - k);
        n += 20 - k;
    }
    inline void put_kmer16(uint32_t key) {
        for (int b = 3; b >= 0; b--) {
            std::memcpy(p + n, &g_kmer_quads.t[(key >> (8 * b)) & 0xFFu], 4);
            n += 4;
        }
    }
};

// the X= / Q= window of a record, read where the text has it: X= is the window without its first stranded character, Q= sums all of it
struct HostSeqWindow {
    static constexpr bool kBulk = true;
    const uint8_t *raw;  // the fragment's bases
    inline char operator()(int) const { return 0; }  // (never used: bulk_x below)
    template <class Sink>
    inline void bulk_x(Sink &s, const NameWindow &nw) const {
        const int m = nw.n_chars - 1;
        char *o = s.p + s.n;
        if (!nw.rev)
            std::memcpy(o, raw + nw.lo + 1, (size_t)m);
        else  // stranded character k = complement of raw[lo + n_chars - 1 - k], k = 1 .. n_chars - 1
            for (int j = 0; j < m; j++) o[j] = rc_char(raw[nw.lo + m - 1 - j]);
        s.n += m;
    }
};
struct HostQualWindow {
    static constexpr bool kBulk = true;
    const uint8_t *raw;
    inline char operator()(int) const { return 0; }
    inline int bulk_sum(const NameWindow &nw) const {
        int sum = 0;
        for (int k = 0; k < nw.n_chars; k++) sum += raw[nw.lo + k];
        return sum - 33 * nw.n_chars;
    }
};

// fragment tag + suffix (format_record_suffix of smi_write.hip)
inline int format_suffix_host(const WriteJob &J, const HostPlan &R, size_t i, uint32_t read_id, HostSink &s, bool *quals_set) {
    const smi_pass2_decisions &D = *J.dec;
    if (R.frag >= 0 && R.had_blank) {
        const smi_chimera_result &ch = D.chim[R.src];
        const int cut = R.frag < ch.n_split ? R.frag : ch.n_split - 1;
        s.put('_');
        s.puts(split_tag_host(cut == 0 ? ch.reason[0] : ch.reason[1]));
        s.puts("sp");
        s.put_i32(R.frag + 1);
    }
    if (R.forced_failed) {
        s.puts("_FAILED ");
        *quals_set = true;
        return NAME_OK;
    }
    const smi_scan_result &sc = D.scan[i];
    const HostSeqWindow seq_w{J.text + R.rd};
    const HostQualWindow qual_w{J.text + R.ql};
    return append_name_suffix(s, sc, &D.bc[i], D.rank ? D.rank[i] : 0, read_id, J.five_prime != 0, R.len, seq_w, qual_w, quals_set);
}

// ---- reverse complement / reversal of a run ------------------------------------------------------------------------------------
struct RcTable {
    uint8_t t[256];
    RcTable() {
        for (int c = 0; c < 256; c++) t[c] = (uint8_t)rc_char((unsigned char)c);
    }
};
const RcTable g_rc;

// out[k] = rc(src[n - 1 - k]) (complement = true) or src[n - 1 - k]
inline void reverse_scalar(const uint8_t *src, size_t n, uint8_t *out, bool complement) {
    if (complement)
        for (size_t k = 0; k < n; k++) out[k] = g_rc.t[src[n - 1 - k]];
    else
        for (size_t k = 0; k < n; k++) out[k] = src[n - 1 - k];
}

__attribute__((target("avx2"))) inline void reverse_avx2(const uint8_t *src, size_t n, uint8_t *out, bool complement) {
    const __m256i rev16 = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    size_t k = 0;
    // AVX2 has no cheap 256-entry table: reverse 32 bytes at a time, complement A C G T N with compares, anything else through the table
    for (; k + 32 <= n; k += 32) {
        __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + n - 32 - k));
        v = _mm256_shuffle_epi8(v, rev16);
        v = _mm256_permute2x128_si256(v, v, 0x01);
        if (complement) {
            const __m256i a = _mm256_cmpeq_epi8(v, _mm256_set1_epi8('A')), c = _mm256_cmpeq_epi8(v, _mm256_set1_epi8('C'));
            const __m256i g = _mm256_cmpeq_epi8(v, _mm256_set1_epi8('G')), t = _mm256_cmpeq_epi8(v, _mm256_set1_epi8('T'));
            const __m256i nn = _mm256_cmpeq_epi8(v, _mm256_set1_epi8('N'));
            const __m256i known = _mm256_or_si256(_mm256_or_si256(a, c), _mm256_or_si256(_mm256_or_si256(g, t), nn));
            if (_mm256_movemask_epi8(known) != -1) {
                reverse_scalar(src + n - 32 - k, 32, out + k, true);
                continue;
            }
            __m256i r = _mm256_and_si256(a, _mm256_set1_epi8('T'));
            r = _mm256_or_si256(r, _mm256_and_si256(c, _mm256_set1_epi8('G')));
            r = _mm256_or_si256(r, _mm256_and_si256(g, _mm256_set1_epi8('C')));
            r = _mm256_or_si256(r, _mm256_and_si256(t, _mm256_set1_epi8('A')));
            r = _mm256_or_si256(r, _mm256_and_si256(nn, _mm256_set1_epi8('N')));
            v = r;
        }
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(out + k), v);
    }
    if (k < n) reverse_scalar(src, n - k, out + k, complement);
}

alignas(64) const uint8_t g_iota64[64] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21,
                                          22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43,
                                          44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63};
// out[k .. k + m) = mirror of src[n - k - m .. n - k), m <= 64 (masked load, one permute, masked store); -> line feeds seen
__attribute__((target("avx512f,avx512bw,avx512vbmi"))) inline __mmask64 reverse_piece_avx512(const uint8_t *src, size_t n, uint8_t *out, size_t k, size_t m,
                                                                                            bool complement) {
    const __mmask64 keep = m >= 64 ? ~0ull : ((1ull << m) - 1ull);
    const __m512i raw = _mm512_maskz_loadu_epi8(keep, src + n - k - m);
    const __mmask64 nl = _mm512_cmpeq_epi8_mask(raw, _mm512_set1_epi8('\n'));
    const __m512i idx = _mm512_sub_epi8(_mm512_set1_epi8((char)(m - 1)), _mm512_load_si512(g_iota64));  // m - 1 - i (low six bits)
    __m512i v = _mm512_permutexvar_epi8(idx, raw);
    if (complement) v = _mm512_maskz_permutex2var_epi8(~_mm512_movepi8_mask(v), _mm512_loadu_si512(g_rc.t), v, _mm512_loadu_si512(g_rc.t + 64));
    _mm512_mask_storeu_epi8(out + k, keep, v);
    return nl;
}

// The AVX-512 forms write the long runs of a record with NON-TEMPORAL stores: the output is 2.5 KB per read that nobody reads again
// before it leaves for a file or a compressor, and a cached store first reads the line it overwrites -- a third of the writer's memory
// traffic.  Head and tail of a run (up to 63 bytes each) take masked ordinary stores so that the streamed part is whole, aligned lines.
// Both return whether the run holds a line feed (the check the one-pass index leaves to whoever reads the qualities).
__attribute__((target("avx512f,avx512bw,avx512vbmi"))) inline bool reverse_avx512(const uint8_t *src, size_t n, uint8_t *out, bool complement) {
    alignas(64) static const uint8_t idx_rev[64] = {63, 62, 61, 60, 59, 58, 57, 56, 55, 54, 53, 52, 51, 50, 49, 48, 47, 46, 45, 44, 43, 42,
                                                    41, 40, 39, 38, 37, 36, 35, 34, 33, 32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20,
                                                    19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9,  8,  7,  6,  5,  4,  3,  2,  1,  0};
    const __m512i rev = _mm512_load_si512(idx_rev), lf = _mm512_set1_epi8('\n');
    // FastqRecordExt.REVERSE_COMPLEMENT for characters 0..127 as two 64-byte tables (vpermi2b indexes 128 entries); >= 128 maps to 0
    const __m512i t_lo = _mm512_loadu_si512(g_rc.t), t_hi = _mm512_loadu_si512(g_rc.t + 64);
    __mmask64 nl = 0;
    size_t k = 0;
    const size_t head = (size_t)(-(uintptr_t)out) & 63;
    if (n >= 256 && head) {
        nl |= reverse_piece_avx512(src, n, out, 0, head, complement);
        k = head;
    }
    const bool stream = n >= 256;
    for (; k + 64 <= n; k += 64) {
        const __m512i raw = _mm512_loadu_si512(src + n - 64 - k);
        nl |= _mm512_cmpeq_epi8_mask(raw, lf);
        __m512i v = _mm512_permutexvar_epi8(rev, raw);
        if (complement) v = _mm512_maskz_permutex2var_epi8(~_mm512_movepi8_mask(v), t_lo, v, t_hi);
        if (stream)
            _mm512_stream_si512(reinterpret_cast<__m512i *>(out + k), v);
        else
            _mm512_storeu_si512(out + k, v);
    }
    if (k < n) nl |= reverse_piece_avx512(src, n, out, k, n - k, complement);
    return nl != 0;
}

__attribute__((target("avx512f,avx512bw"))) inline bool copy_avx512(const uint8_t *src, size_t n, uint8_t *out) {
    const __m512i lf = _mm512_set1_epi8('\n');
    __mmask64 nl = 0;
    size_t k = 0;
    const bool stream = n >= 256;
    const size_t head = (size_t)(-(uintptr_t)out) & 63;
    if (stream && head) {
        const __mmask64 keep = (1ull << head) - 1ull;
        const __m512i v = _mm512_maskz_loadu_epi8(keep, src);
        nl |= _mm512_cmpeq_epi8_mask(v, lf) & keep;
        _mm512_mask_storeu_epi8(out, keep, v);
        k = head;
    }
    for (; k + 64 <= n; k += 64) {
        const __m512i v = _mm512_loadu_si512(src + k);
        nl |= _mm512_cmpeq_epi8_mask(v, lf);
        if (stream)
            _mm512_stream_si512(reinterpret_cast<__m512i *>(out + k), v);
        else
            _mm512_storeu_si512(out + k, v);
    }
    if (k < n) {
        const __mmask64 keep = (1ull << (n - k)) - 1ull;
        const __m512i v = _mm512_maskz_loadu_epi8(keep, src + k);
        nl |= _mm512_cmpeq_epi8_mask(v, lf) & keep;
        _mm512_mask_storeu_epi8(out + k, keep, v);
    }
    return nl != 0;
}

// mirrored (and complemented) copy of a run; -> the run holds a line feed (only looked at for quality runs)
inline bool reverse_run(const uint8_t *src, size_t n, uint8_t *out, bool complement, int level) {
    if (!n) return false;
    if (level == 2) return reverse_avx512(src, n, out, complement);
    if (level == 1)
        reverse_avx2(src, n, out, complement);
    else
        reverse_scalar(src, n, out, complement);
    return !complement && std::memchr(src, '\n', n) != nullptr;
}
inline bool copy_run(const uint8_t *src, size_t n, uint8_t *out, int level, bool check) {
    if (!n) return false;
    if (level == 2) return copy_avx512(src, n, out);
    std::memcpy(out, src, n);
    return check && std::memchr(src, '\n', n) != nullptr;
}

struct alignas(128) ThreadOut {  // (aligned: neighbouring threads' state must not share a cache line)
    std::vector<char> sfx;          // suffixes of this thread's records, back to back
    std::vector<uint32_t> sfx_off;  // per record (+1)
    std::vector<uint64_t> bytes;    // record length
    std::vector<uint8_t> flags;     // 1 passed, 2 quals_set
    uint64_t tot[2] = {0, 0};       // bytes passed / failed
    uint32_t err = 0;
    // grow-only: a chunk worker calls the writer once per chunk with about the same record count, and a fresh 20 MB vector per thread and
    // call is 20 MB of zero-fill and page faults per thread and call
    void reset(size_t n_rec) {
        if (sfx.size() < n_rec * 192 + kSuffixCapHost + 64) sfx.resize(n_rec * 192 + kSuffixCapHost + 64);
        if (sfx_off.size() < n_rec + 1) sfx_off.resize(n_rec + 1);
        if (bytes.size() < n_rec) bytes.resize(n_rec);
        if (flags.size() < n_rec) flags.resize(n_rec);
        tot[0] = tot[1] = 0;
        err = 0;
    }
};

// per-thread scratch kept between calls (several lanes may write at once: each call takes its own set)
struct ScratchPool {
    std::mutex mu;
    std::vector<std::unique_ptr<ThreadOut>> idle;
    std::unique_ptr<ThreadOut> take() {
        std::lock_guard<std::mutex> g(mu);
        if (idle.empty()) return std::unique_ptr<ThreadOut>(new ThreadOut());
        std::unique_ptr<ThreadOut> t = std::move(idle.back());
        idle.pop_back();
        return t;
    }
    void give(std::unique_ptr<ThreadOut> t) {
        std::lock_guard<std::mutex> g(mu);
        if (idle.size() < 256) idle.push_back(std::move(t));
    }
};
ScratchPool g_scratch;

}  // namespace

extern "C" int smi_fastq_write_host(const uint8_t *text, const smi_fastq_record *recs, const uint64_t *offsets, const smi_pass2_decisions *dec,
                                    uint32_t first_read_id, const smi_write_config *cfg, uint8_t *passed, size_t cap_passed, uint8_t *failed,
                                    size_t cap_failed, uint64_t *totals, uint32_t *errors, int n_threads) {
    if (!dec || !cfg || !totals || !errors) {
        set_error("smi_fastq_write_host: null argument");
        return SMI_ERR_INVALID;
    }
    totals[0] = totals[1] = totals[2] = 0;
    *errors = 0;
    const size_t m = dec->n_records_out;
    if (!m) return SMI_OK;
    if (!text || !recs || !offsets || !dec->frag_offsets || !dec->scan || !dec->bc || !passed || !failed ||
        ((dec->frag_src == nullptr) != (dec->chim == nullptr))) {
        set_error("smi_fastq_write_host: null argument");
        return SMI_ERR_INVALID;
    }
    const WriteJob J{text, recs, offsets, dec, first_read_id, cfg->five_prime, cfg->trim_fastq};
    const int nt = clamp_threads(n_threads, m, 128);
    const int level = simd_level();
    std::vector<std::unique_ptr<ThreadOut>> TO((size_t)nt);
    for (auto &t : TO) t = g_scratch.take();
    struct GiveBack {
        std::vector<std::unique_ptr<ThreadOut>> &v;
        ~GiveBack() {
            for (auto &t : v) g_scratch.give(std::move(t));
        }
    } give_back{TO};
    std::vector<uint64_t> n_passed((size_t)nt + 1, 0), base_p((size_t)nt + 1, 0), base_f((size_t)nt + 1, 0);
    std::atomic<uint32_t> err{0};
    const bool timing = std::getenv("SMI_PK_TIMING") != nullptr;
    double t_phase[4] = {0, 0, 0, 0};
    auto stamp = [&](int t, int k) {
        if (timing && t == 0) t_phase[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    Team::run(nt, [&](int t, Team &team) {
        const size_t lo = m * (size_t)t / (size_t)nt, hi = m * (size_t)(t + 1) / (size_t)nt;
        stamp(t, 0);
        // 1. passed records of my range -> read ids (GET_NEXT_READID per passed record, in record order)
        uint64_t np = 0;
        for (size_t i = lo; i < hi; i++) np += record_passed(J, i) ? 1 : 0;
        n_passed[t + 1] = np;
        team.barrier();
        uint64_t ord = 0;
        for (int u = 1; u <= t; u++) ord += n_passed[u];
        stamp(t, 1);
        // 2. suffixes and record lengths
        ThreadOut &O = *TO[t];
        O.reset(hi - lo);
        size_t at = 0;
        constexpr size_t kAheadFmt = 6;  // the formatter reads ~3 cache lines per record that nothing else has touched: ask for them early
        for (size_t i = lo; i < hi; i++) {
            if (i + kAheadFmt < hi) {
                const size_t j = i + kAheadFmt;
                const uint32_t src = dec->frag_src ? (dec->frag_src[j] >> 2) : (uint32_t)j;
                const smi_fastq_record &rj = recs[src];
                _mm_prefetch(reinterpret_cast<const char *>(text + rj.name_start), _MM_HINT_T0);
                const smi_scan_result &sj = dec->scan[j];
                if (sj.found) {
                    const int len_j = (int)(dec->frag_offsets[j + 1] - dec->frag_offsets[j]);
                    const NameWindow nwj = name_window(sj, J.five_prime != 0, len_j);
                    if (nwj.has) {
                        const uint64_t in_read = dec->frag_offsets[j] - offsets[src];
                        prefetch_range(text + rj.seq_start + in_read + nwj.lo, 64);
                        prefetch_range(text + rj.qual_start + in_read + nwj.lo, 64);
                    }
                }
            }
            const HostPlan R = plan_record_host(J, i);
            if (at + kSuffixCapHost + 64 > O.sfx.size()) O.sfx.resize(O.sfx.size() * 2 + kSuffixCapHost);
            HostSink s{O.sfx.data() + at, 0};
            bool quals_set = true;
            const int st = format_suffix_host(J, R, i, first_read_id + (uint32_t)ord, s, &quals_set);
            if (st == NAME_RANGE) O.err |= SMI_WR_NAME_RANGE;
            // the device measures the suffix with id 0 for this check (one digit); the limit is on that length
            int id_extra = 0;
            if (R.passed) {
                uint32_t v = first_read_id + (uint32_t)ord;
                const smi_scan_result &sc = dec->scan[i];
                const int begin = J.five_prime ? sc.adapter_end - 3 : sc.adapter_end - 41;
                if (sc.found && begin >= 0)
                    while (v >= 36u) {
                        v /= 36u;
                        id_extra++;
                    }
            }
            if (s.n - id_extra + 8 > kSuffixCapHost) O.err |= SMI_WR_NAME_TOO_LONG;
            const int n_sfx = s.n > kSuffixCapHost ? kSuffixCapHost : s.n;
            O.sfx_off[i - lo] = (uint32_t)at;
            at += (size_t)n_sfx;
            const uint64_t bytes = 1ull + R.name_tok_len + (uint64_t)s.n + 1 + (uint64_t)R.cut_len + 1 + 1 + R.qh_len + 1 +
                                   (quals_set ? (uint64_t)R.cut_len : 4ull) + 1;
            O.bytes[i - lo] = bytes;
            O.flags[i - lo] = (uint8_t)((R.passed ? 1 : 0) | (quals_set ? 2 : 0));
            O.tot[R.passed ? 0 : 1] += bytes;
            if (R.passed) ord++;
        }
        O.sfx_off[hi - lo] = (uint32_t)at;
        base_p[t + 1] = O.tot[0];
        base_f[t + 1] = O.tot[1];
        if (O.err) err.fetch_or(O.err);
        team.barrier();
        stamp(t, 2);
        if (err.load()) return;
        uint64_t off_p = 0, off_f = 0, all_p = 0, all_f = 0;
        for (int u = 1; u <= nt; u++) {
            if (u <= t) {
                off_p += base_p[u];
                off_f += base_f[u];
            }
            all_p += base_p[u];
            all_f += base_f[u];
        }
        if (all_p > cap_passed || all_f > cap_failed) {
            if (t == 0) err.fetch_or(SMI_WR_OVERFLOW);
            return;
        }
        // 3. the records
        bool qual_nl = false;
        for (size_t i = lo; i < hi; i++) {
            if (i + 2 < hi) {  // the next but one record's bases and qualities: 2.4 KB in two bursts the hardware prefetcher is too slow for
                const size_t j = i + 2;
                const uint32_t src = dec->frag_src ? (dec->frag_src[j] >> 2) : (uint32_t)j;
                const uint64_t in_read = dec->frag_offsets[j] - offsets[src], len_j = dec->frag_offsets[j + 1] - dec->frag_offsets[j];
                prefetch_range(text + recs[src].seq_start + in_read, (size_t)len_j);
                prefetch_range(text + recs[src].qual_start + in_read, (size_t)len_j);
                _mm_prefetch(reinterpret_cast<const char *>(text + recs[src].name_start), _MM_HINT_T0);
            }
            const HostPlan R = plan_record_host(J, i);
            const bool quals_set = O.flags[i - lo] & 2;
            const bool verify = recs[R.src].reserved != 0;  // the index stepped over this quality line
            uint8_t *o = R.passed ? passed + off_p : failed + off_f;
            uint8_t *const o0 = o;
            *o++ = '@';
            std::memcpy(o, text + R.name_beg, R.name_tok_len);
            o += R.name_tok_len;
            const uint32_t sl = O.sfx_off[i - lo + 1] - O.sfx_off[i - lo];
            std::memcpy(o, O.sfx.data() + O.sfx_off[i - lo], sl);
            o += sl;
            *o++ = '\n';
            const uint8_t *rd = text + R.rd, *ql = text + R.ql;
            if (R.rev)
                reverse_run(rd + (R.len - R.cut_beg - R.cut_len), (size_t)R.cut_len, o, true, level);
            else
                copy_run(rd + R.cut_beg, (size_t)R.cut_len, o, level, false);
            o += R.cut_len;
            *o++ = '\n';
            *o++ = '+';
            std::memcpy(o, text + R.qh_beg, R.qh_len);
            o += R.qh_len;
            *o++ = '\n';
            if (!quals_set) {
                std::memcpy(o, "null", 4);
                o += 4;
                if (verify && R.len && std::memchr(ql, '\n', (size_t)R.len)) qual_nl = true;
            } else {
                bool nl;
                if (R.rev)
                    nl = reverse_run(ql + (R.len - R.cut_beg - R.cut_len), (size_t)R.cut_len, o, false, level);
                else
                    nl = copy_run(ql + R.cut_beg, (size_t)R.cut_len, o, level, verify);
                o += R.cut_len;
                if (verify) {
                    qual_nl |= nl;
                    if (R.cut_len != R.len && std::memchr(ql, '\n', (size_t)R.len)) qual_nl = true;  // -u: the part that is not written
                }
            }
            *o++ = '\n';
            const uint64_t wrote = (uint64_t)(o - o0);
            if (wrote != O.bytes[i - lo]) err.fetch_or(0x80000000u);  // internal consistency (never expected)
            (R.passed ? off_p : off_f) += wrote;
        }
        if (qual_nl) err.fetch_or(SMI_WR_QUAL_NEWLINE);
        if (level == 2) _mm_sfence();  // the streamed stores are visible before the team is joined
        stamp(t, 3);
    });
    if (timing)
        fprintf(stderr, "smi_fastq_write_host (thread 0 of %d): passed count %.2f  suffixes %.2f  records %.2f ms\n", nt, t_phase[1] - t_phase[0],
                t_phase[2] - t_phase[1], t_phase[3] - t_phase[2]);
    uint64_t all_p = 0, all_f = 0, all_n = 0;
    for (int u = 1; u <= nt; u++) {
        all_p += base_p[u];
        all_f += base_f[u];
        all_n += n_passed[u];
    }
    *errors = err.load();
    if (*errors) {
        set_error("smi_fastq_write_host: see *errors (SMI_WR_*)");
        return SMI_ERR_INVALID;
    }
    totals[0] = all_p;
    totals[1] = all_f;
    totals[2] = all_n;
    return SMI_OK;
}
