// smi_deflate.hip -- K-DEFLATE: gzip members made on the device, for the `--compress` output of scanfastq
// (FastqWriterThreadPool.java:L242-257: one GZIPOutputStream per `<base>_passed.fastq.gz` / `<base>_failed.fastq.gz`; quickrun-2.1.sh:35 runs
// the reference with --compress).  The records K-WRITE leaves in HBM are deflated where they lie and only the compressed bytes cross the
// link; any inflater (java.util.zip, zlib, gzip) reads the result, and the parity test is exactly that round trip plus the CRC-32 / ISIZE
// trailer every gzip reader checks.
//
// Format (RFC 1951 / 1952).  The input is cut into blocks of 64 KiB; each becomes ONE dynamic-Huffman deflate block (BTYPE = 10) of
// literals only -- FASTQ text is two thirds bases and qualities whose order-0 entropy is what zlib reaches on them too; the repeated
// key words of the name line, which LZ77 would catch, are a few per cent of the bytes -- followed by an empty stored block
// (00 00 FF FF), which ends on a byte boundary (the "sync flush" of zlib, what pigz puts between its independently compressed chunks), so the
// blocks of a member are deflated independently and then just concatenated.  A member ends with the empty final block 03 00.
//
// One workgroup (256 threads) per block:
//   pass A  the block's bytes -> symbol histogram (8 privatised copies in LDS) and the CRC-32 of each 64-byte run (a thread owns runs
//           tid, tid + 256, ...: a wave reads 4 KiB in one piece, and each run is one piece of the bit stream)
//   tree    used symbols compacted and sorted by frequency (bitonic, LDS), code lengths by the in-place minimum-redundancy algorithm of
//           Moffat & Katajainen (one lane; 60 - 90 used symbols for FASTQ), frequencies halved and the tree rebuilt while a length exceeds
//           15; canonical codes, bit-reversed for the LSB-first stream
//   header  HLIT = 257 codes, HDIST = 1 (length 0: no distances), code lengths run-length coded with a FIXED complete code-length code
//           (sixteen-plus-two symbols, 4 or 5 bits), so no second tree is built
//   pass B  bits per run, exclusive scan -> every run's bit offset in the block's slot
//   pass C  codes packed into a 64-bit accumulator, whole words stored, the two boundary words of a run OR-ed in atomically
// then a scan of the block sizes and a copy kernel make the member contiguous behind its 10-byte header, and one lane writes the trailer.
// CRC-32 of a concatenation is linear in the CRCs of its parts: crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] / P, so every thread
// multiplies its segment's CRC by x^(8 * bytes behind it) and the workgroups XOR their sums into one word.
// Byte work, HBM / LDS bound: no MFMA.
#include <hipcub/hipcub.hpp>

#include "smi_internal.h"

namespace smi {

namespace {

constexpr int kDeflateBlock = 65536;                                  // input bytes per deflate block
constexpr int kDefThreads = 256;
constexpr int kRun = 64;                                              // bytes per run: a thread owns runs tid, tid + 256, tid + 512, tid + 768
constexpr int kRunsPerThread = kDeflateBlock / kRun / kDefThreads;    // 4
constexpr int kSlotBytes = kDeflateBlock + kDeflateBlock / 4 + 1024;  // worst case is < 9.1 bits per byte + header; checked
constexpr int kSyms = 257;                                            // literals + end-of-block (no length codes are used)
constexpr uint32_t kCrcPoly = 0xEDB88320u;

// ---- GF(2) arithmetic of the CRC (reflected representation: bit 31 is x^0) ---------------------------------------------------
__host__ __device__ constexpr uint32_t gf_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
struct PowTables {
    uint32_t x2n[40];    // x^(8 * 2^k) mod P
    uint32_t run[1024];  // x^(8 * 64 * k) mod P
};
constexpr PowTables make_pow_tables() {
    PowTables t{};
    uint32_t p = 1u << 30;                        // x^1
    for (int k = 0; k < 3; k++) p = gf_mul(p, p);  // x^8
    for (int k = 0; k < 40; k++) {
        t.x2n[k] = p;
        p = gf_mul(p, p);
    }
    uint32_t s = 1u << 31;  // x^0
    for (int k = 0; k < 1024; k++) {
        t.run[k] = s;
        s = gf_mul(s, t.x2n[6]);  // * x^(8 * 64)
    }
    return t;
}
__constant__ PowTables c_pow = make_pow_tables();

__device__ __forceinline__ uint32_t x_pow_bytes(uint64_t n) {  // x^(8 n) mod P
    uint32_t p = 1u << 31;
    for (int k = 0; n; n >>= 1, k++)
        if (n & 1u) p = gf_mul(c_pow.x2n[k], p);
    return p;
}

// the fixed code-length code: symbols 0..15 = a code length, 17 = 3-10 zeros (3 extra bits), 18 = 11-138 zeros (7 extra bits); 16 unused.
// Lengths 4 for {0, 2..12, 17, 18}, 5 for {1, 13, 14, 15}: fourteen 4-bit and four 5-bit codes fill the code space exactly.
__host__ __device__ constexpr int cl_len(int s) { return s == 16 ? 0 : (s == 1 || (s >= 13 && s <= 15)) ? 5 : 4; }
struct ClCode {
    uint8_t len[19];
    uint8_t code[19];  // bit-reversed, ready for the LSB-first stream
};
constexpr ClCode make_cl_code() {
    ClCode c{};
    int next[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < 19; s++) cnt[cl_len(s)]++;
    cnt[0] = 0;
    int code = 0;
    for (int b = 1; b < 8; b++) {
        code = (code + cnt[b - 1]) << 1;
        next[b] = code;
    }
    for (int s = 0; s < 19; s++) {
        const int l = cl_len(s);
        c.len[s] = (uint8_t)l;
        if (!l) continue;
        const int v = next[l]++;
        int r = 0;
        for (int b = 0; b < l; b++) r |= ((v >> b) & 1) << (l - 1 - b);
        c.code[s] = (uint8_t)r;
    }
    return c;
}
__constant__ ClCode c_cl = make_cl_code();

struct DeflateLds {
    union {
        uint32_t hist[8][260];  // pass A: privatised histograms (260 = 8 * 32 + 4: a symbol sits in eight different banks in the eight copies)
        uint32_t tail[1024];    // pass C: the bits a run leaves in its last, partial word (the next run's owner stores that word)
    };
    uint32_t ufreq[260];   // used symbols, unsorted
    uint16_t usym[260];
    uint32_t sfreq[260];   // sorted ascending by (frequency, symbol)
    uint16_t ssym[260];
    uint32_t tree[260];    // Moffat-Katajainen work array -> code lengths of the sorted symbols
    uint32_t code[260];    // per symbol: reversed code | length << 16
    uint32_t crc_tab[256];
    uint32_t hdr[64];      // header bits: 74 + at most 258 tokens of 5 bits = 1364 bits
    uint32_t nz[10];       // bit s: position s of the code-length sequence (257 literal / length codes, 1 distance code) is non-zero
    uint32_t wave_sum[4][4];
    uint32_t n_used, hdr_bits, blk_shift, bl_count[16], next_code[16];
};

// ---- one deflate block per workgroup ---------------------------------------------------------------------------------------------
// block_in: input bytes per block (<= kDeflateBlock, a multiple of 64).  block_crc != nullptr (BGZF): every block's own CRC-32 goes there
// instead of into the member's accumulator.
__global__ __launch_bounds__(kDefThreads) void k_deflate_blocks(const uint8_t *__restrict__ in, uint64_t n_bytes, uint32_t block_in, uint8_t *__restrict__ slots,
                                                                uint32_t *__restrict__ block_bytes, uint32_t *__restrict__ crc_acc,
                                                                uint32_t *__restrict__ err, uint32_t *__restrict__ block_crc) {
    __shared__ DeflateLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t b0 = (uint64_t)blockIdx.x * block_in;
    const uint32_t blen = (uint32_t)min((uint64_t)block_in, n_bytes - b0);
    const uint8_t *src = in + b0;
    uint32_t *slot = reinterpret_cast<uint32_t *>(slots + (uint64_t)blockIdx.x * kSlotBytes);
    // ---- tables, zeroes -------------------------------------------------------------------------------------------------------
    for (int i = tid; i < 8 * 260; i += kDefThreads) (&L.hist[0][0])[i] = 0;
    {
        uint32_t c = (uint32_t)tid;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        L.crc_tab[tid] = c;
    }
    if (tid < 64) L.hdr[tid] = 0;
    if (tid < 16) L.bl_count[tid] = 0;
    if (tid == 0) {
        L.n_used = 0;
        L.blk_shift = block_crc ? (1u << 31) : x_pow_bytes(n_bytes - (b0 + blen));  // x^(8 * bytes behind this block); BGZF: the block's own CRC
    }
    __syncthreads();
    // ---- pass A: histogram + CRC of my runs ---------------------------------------------------------------------------------------
    // Run q = j * 256 + tid covers bytes [64 q, 64 q + 64) of the block: the 64 lanes of a wave read one contiguous 4 KiB per j (four
    // 16-byte loads per lane whose lines the wave shares), and every run is a contiguous piece of the bit stream with one offset.
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    auto run_begin = [&](int j) { return ((uint32_t)j * kDefThreads + (uint32_t)tid) * kRun; };
    auto for_run_bytes = [&](int j, auto &&f) {  // f(byte) over run j of this thread, in order
        const uint32_t r0 = run_begin(j);
        if (r0 >= blen) return;
        if (r0 + kRun <= blen && aligned) {
            const uint4 *p = reinterpret_cast<const uint4 *>(src + r0);
            const uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
            const uint32_t w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
            for (int k = 0; k < 16; k++)
#pragma unroll
                for (int q = 0; q < 4; q++) f((w[k] >> (8 * q)) & 0xFFu);
        } else {
            const uint32_t r1 = min(r0 + (uint32_t)kRun, blen);
            for (uint32_t i = r0; i < r1; i++) f((uint32_t)src[i]);
        }
    };
    {
        uint32_t *h = L.hist[tid & 7];
        uint32_t part = 0;
        auto fold = [&](int j, uint32_t crc) {  // CRC of run j times x^(8 * bytes of this block behind it)
            const uint32_t r1 = min(run_begin(j) + (uint32_t)kRun, blen);
            const uint32_t behind = blen - r1;
            const uint32_t f = (behind % (uint32_t)kRun == 0) ? c_pow.run[behind / kRun] : x_pow_bytes(behind);
            part ^= gf_mul(f, ~crc);
        };
#pragma unroll 1
        for (int j = 0; j < kRunsPerThread; j += 2) {
            if (run_begin(j) >= blen) break;
            if (aligned && run_begin(j + 1) + kRun <= blen) {
                // two full runs at once: their CRC chains (a table look-up per byte, each waiting for the one before) overlap
                const uint4 *p = reinterpret_cast<const uint4 *>(src + run_begin(j)), *q4 = reinterpret_cast<const uint4 *>(src + run_begin(j + 1));
                uint32_t ca = 0xFFFFFFFFu, cb = 0xFFFFFFFFu;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint4 va = p[k], vb = q4[k];
                    const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t a = (wa[i] >> (8 * q)) & 0xFFu, b = (wb[i] >> (8 * q)) & 0xFFu;
                            atomicAdd(&h[a], 1u);
                            atomicAdd(&h[b], 1u);
                            ca = L.crc_tab[(ca ^ a) & 0xFFu] ^ (ca >> 8);
                            cb = L.crc_tab[(cb ^ b) & 0xFFu] ^ (cb >> 8);
                        }
                }
                fold(j, ca);
                fold(j + 1, cb);
            } else {
                for (int jj = j; jj < j + 2; jj++) {
                    if (run_begin(jj) >= blen) break;
                    uint32_t crc = 0xFFFFFFFFu;
                    for_run_bytes(jj, [&](uint32_t c) {
                        atomicAdd(&h[c], 1u);
                        crc = L.crc_tab[(crc ^ c) & 0xFFu] ^ (crc >> 8);
                    });
                    fold(jj, crc);
                }
            }
        }
        part = gf_mul(L.blk_shift, part);
#pragma unroll
        for (int o = 32; o; o >>= 1) part ^= __shfl_xor(part, o);
        if (lane == 0) L.wave_sum[0][wave] = part;
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t c = L.wave_sum[0][0] ^ L.wave_sum[0][1] ^ L.wave_sum[0][2] ^ L.wave_sum[0][3];
        if (block_crc)
            block_crc[blockIdx.x] = c;
        else
            atomicXor(crc_acc, c);
    }
    // ---- symbol frequencies -> list of the used ones, sorted by rank counting ------------------------------------------------------
    {
        uint32_t f_mine = 0;
        for (int k = 0; k < 8; k++) f_mine += L.hist[k][tid];
        if (f_mine) {
            const uint32_t at = atomicAdd(&L.n_used, 1u);
            L.ufreq[at] = f_mine;
            L.usym[at] = (uint16_t)tid;
        }
        L.code[tid] = 0;
        if (tid == 0) L.code[256] = 0;
    }
    __syncthreads();
    if (tid == 0) {  // end-of-block, once
        const uint32_t at = L.n_used++;
        L.ufreq[at] = 1u;
        L.usym[at] = 256;
    }
    __syncthreads();
    const uint32_t n_used = L.n_used;  // >= 2: a block holds at least one byte
    for (uint32_t i = (uint32_t)tid; i < n_used; i += kDefThreads) {
        const uint32_t fi = L.ufreq[i], si = L.usym[i];
        uint32_t rank = 0;
        for (uint32_t k = 0; k < n_used; k++) {
            const uint32_t fk = L.ufreq[k], sk = L.usym[k];
            rank += (fk < fi || (fk == fi && sk < si)) ? 1u : 0u;
        }
        L.sfreq[rank] = fi;
        L.ssym[rank] = (uint16_t)si;
    }
    __syncthreads();
    // ---- code lengths (one lane): Moffat & Katajainen, "In-place calculation of minimum-redundancy codes" ----------------------
    if (tid == 0) {
        const int n = (int)n_used;
        uint32_t *A = L.tree;
        for (;;) {
            for (int i = 0; i < n; i++) A[i] = L.sfreq[i];
            // phase 1: parent pointers
            A[0] += A[1];
            int root = 0, leaf = 2;
            for (int next = 1; next < n - 1; next++) {
                if (leaf >= n || A[root] < A[leaf]) {
                    A[next] = A[root];
                    A[root++] = (uint32_t)next;
                } else
                    A[next] = A[leaf++];
                if (leaf >= n || (root < next && A[root] < A[leaf])) {
                    A[next] += A[root];
                    A[root++] = (uint32_t)next;
                } else
                    A[next] += A[leaf++];
            }
            // phase 2: internal node depths
            A[n - 2] = 0;
            for (int next = n - 3; next >= 0; next--) A[next] = A[A[next]] + 1;
            // phase 3: leaf depths
            int avbl = 1, used = 0, dpth = 0;
            root = n - 2;
            int next = n - 1;
            while (avbl > 0) {
                while (root >= 0 && (int)A[root] == dpth) {
                    used++;
                    root--;
                }
                while (avbl > used) {
                    A[next--] = (uint32_t)dpth;
                    avbl--;
                }
                avbl = 2 * used;
                dpth++;
                used = 0;
            }
            if (A[0] <= 15u) break;
            for (int i = 0; i < n; i++) L.sfreq[i] = (L.sfreq[i] + 1u) >> 1;  // flatter frequencies, same order: shallower tree
        }
    }
    __syncthreads();
    // ---- canonical codes -----------------------------------------------------------------------------------------------------------
    for (uint32_t i = (uint32_t)tid; i < n_used; i += kDefThreads) {
        const uint32_t len = L.tree[i];
        L.code[L.ssym[i]] = len << 16;
        atomicAdd(&L.bl_count[len], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t code = 0;
        L.next_code[0] = 0;
        for (int b = 1; b <= 15; b++) {
            code = (code + (b > 1 ? L.bl_count[b - 1] : 0u)) << 1;
            L.next_code[b] = code;
        }
    }
    __syncthreads();
    for (int s = tid; s < kSyms; s += kDefThreads) {
        const uint32_t len = L.code[s] >> 16;
        if (len) {
            uint32_t rank = 0;  // symbols before s with the same length
            for (int q = 0; q < s; q++) rank += ((L.code[q] >> 16) == len) ? 1u : 0u;
            const uint32_t c = L.next_code[len] + rank;
            L.code[s] = (L.code[s] & 0xFFFF0000u) | (__brev(c) >> (32u - len));
        }
    }
    __syncthreads();
    // ---- header: every position of the code-length sequence finds its token(s) and their place by itself ---------------------------
    // positions 0..255 = the literals (thread = position), 256 = end-of-block (always used), 257 = the one distance code (length 0)
    {
        const uint32_t my_len = L.code[tid] >> 16;
        const uint64_t m = __ballot(my_len != 0);
        if (lane == 0) {
            L.nz[2 * wave] = (uint32_t)m;
            L.nz[2 * wave + 1] = (uint32_t)(m >> 32);
        }
        if (tid == 0) {
            L.nz[8] = 1u;  // 256: used; 257: zero
            L.nz[9] = 0u;
        }
        __syncthreads();
        // zero runs: a run starts where the position before is used; as the serial greedy coder would, it is cut into pieces of 138
        // (symbol 18), then one piece of 11..137 (18) or 3..10 (17), or one or two single zeros
        uint32_t bits = 0, run = 0;
        if (my_len)
            bits = c_cl.len[my_len];
        else if (tid == 0 || ((L.nz[(tid - 1) >> 5] >> ((tid - 1) & 31)) & 1u)) {
            // next used position behind tid (256 is always one)
            uint32_t p = (uint32_t)tid + 1;
            for (;;) {
                const uint32_t w = L.nz[p >> 5] >> (p & 31u);
                if (w) {
                    p += (uint32_t)__builtin_ctz(w);
                    break;
                }
                p = (p | 31u) + 1u;
            }
            run = p - (uint32_t)tid;
            const uint32_t k = run / 138u, r = run % 138u;
            bits = k * (c_cl.len[18] + 7u) + (r >= 11u ? c_cl.len[18] + 7u : r >= 3u ? c_cl.len[17] + 3u : r * c_cl.len[0]);
        }
        uint32_t inc = bits;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(inc, o);
            if (lane >= o) inc += y;
        }
        if (lane == 63) L.wave_sum[1][wave] = inc;
        __syncthreads();
        uint32_t at = 74u + inc - bits;  // 3 + 5 + 5 + 4 + 19 * 3 bits come first
        for (int w = 0; w < wave; w++) at += L.wave_sum[1][w];
        auto put = [&](uint32_t v, uint32_t nb) {
            const uint32_t w = at >> 5, sh = at & 31u;
            atomicOr(&L.hdr[w], v << sh);
            if (sh + nb > 32u) atomicOr(&L.hdr[w + 1], v >> (32u - sh));
            at += nb;
        };
        if (my_len)
            put(c_cl.code[my_len], c_cl.len[my_len]);
        else if (run) {
            uint32_t r = run;
            for (; r >= 138u; r -= 138u) {
                put(c_cl.code[18], c_cl.len[18]);
                put(127u, 7);
            }
            if (r >= 11u) {
                put(c_cl.code[18], c_cl.len[18]);
                put(r - 11u, 7);
            } else if (r >= 3u) {
                put(c_cl.code[17], c_cl.len[17]);
                put(r - 3u, 3);
            } else
                for (; r; r--) put(c_cl.code[0], c_cl.len[0]);
        }
        if (tid == 0) {
            at = 0;
            put(0u, 1);   // BFINAL = 0
            put(2u, 2);   // BTYPE = 10, dynamic Huffman
            put(0u, 5);   // HLIT: 257 literal / length codes
            put(0u, 5);   // HDIST: 1 distance code
            put(15u, 4);  // HCLEN: all 19 code-length codes
            const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            for (int k = 0; k < 19; k++) put((uint32_t)c_cl.len[order[k]], 3);
            // behind the 256 literal positions: end-of-block (used) and the distance code (a zero run of one)
            at = 74u + L.wave_sum[1][0] + L.wave_sum[1][1] + L.wave_sum[1][2] + L.wave_sum[1][3];
            const uint32_t le = L.code[256] >> 16;
            put(c_cl.code[le], c_cl.len[le]);
            put(c_cl.code[0], c_cl.len[0]);
            L.hdr_bits = at;
        }
    }
    __syncthreads();
    // ---- pass B: bits per run, exclusive scan over the runs in stream order (run q = j * 256 + tid) ---------------------------------
    uint32_t run_bits[kRunsPerThread], run_off[kRunsPerThread];
#pragma unroll
    for (int j = 0; j < kRunsPerThread; j++) {
        uint32_t b = 0;
        for_run_bytes(j, [&](uint32_t c) { b += L.code[c] >> 16; });
        run_bits[j] = b;
        uint32_t inc = b;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(inc, o);
            if (lane >= o) inc += y;
        }
        run_off[j] = inc - b;  // exclusive within the wave
        if (lane == 63) L.wave_sum[j][wave] = inc;
    }
    __syncthreads();
    uint32_t data_end = L.hdr_bits;
#pragma unroll
    for (int j = 0; j < kRunsPerThread; j++) {
        uint32_t before = data_end;
        for (int w = 0; w < 4; w++) {
            if (w < wave) before += L.wave_sum[j][w];
            data_end += L.wave_sum[j][w];
        }
        run_off[j] += before;
    }
    const uint32_t eob = L.code[256];
    // trailer: end-of-block, then BFINAL = 0 / BTYPE = 00, padding to the byte, LEN = 0, NLEN = 0xFFFF
    const uint32_t after_eob = data_end + (eob >> 16) + 3u;
    const uint32_t total_bytes = ((after_eob + 7u) >> 3) + 4u;
    if (total_bytes + 8u > (uint32_t)kSlotBytes) {
        if (tid == 0) {
            atomicOr(err, 1u);
            block_bytes[blockIdx.x] = 0;
        }
        return;  // (uniform)
    }
    // header: its whole words; the last, partial one goes out with the first run's first word
    {
        const uint32_t hw = L.hdr_bits >> 5;
        for (uint32_t w = (uint32_t)tid; w < hw; w += kDefThreads) slot[w] = L.hdr[w];
    }
    // ---- pass C: the codes -------------------------------------------------------------------------------------------------------------
    // A run is at least 64 bits, so it begins in one word, fills some, and ends in another.  Whole words are stored as they fill up; the
    // word a run begins in is kept back and stored after the barrier together with the bits the run before it left there (L.tail), so no
    // word has two writers and nothing has to be zeroed or OR-ed in.  Two codes (<= 30 bits) are joined before they enter the 64-bit
    // accumulator: one long shift and one overflow test per pair.
    uint32_t head[kRunsPerThread];
#pragma unroll
    for (int j = 0; j < kRunsPerThread; j++) {
        head[j] = 0;
        if (!run_bits[j]) continue;
        uint32_t w = run_off[j] >> 5;
        uint32_t nb = run_off[j] & 31u;
        uint64_t acc = 0;
        bool first = true;
        uint32_t pend = 0, pend_len = 0xFFFFFFFFu;  // first code of a pair
        auto push = [&](uint32_t bits, uint32_t len) {
            acc |= (uint64_t)bits << nb;
            nb += len;
            if (nb >= 32u) {
                if (first) {
                    head[j] = (uint32_t)acc;
                    first = false;
                } else
                    slot[w] = (uint32_t)acc;
                w++;
                acc >>= 32;
                nb -= 32u;
            }
        };
        for_run_bytes(j, [&](uint32_t c) {
            const uint32_t e = L.code[c];
            if (pend_len == 0xFFFFFFFFu) {
                pend = e & 0xFFFFu;
                pend_len = e >> 16;
            } else {
                push(pend | ((e & 0xFFFFu) << pend_len), pend_len + (e >> 16));
                pend_len = 0xFFFFFFFFu;
            }
        });
        if (pend_len != 0xFFFFFFFFu) push(pend, pend_len);  // (a run of odd length: the end of the input)
        L.tail[j * kDefThreads + tid] = (uint32_t)acc;        // nb < 32 bits, the rest is zero
    }
    __syncthreads();
    const uint32_t hdr_tail = (L.hdr_bits & 31u) ? L.hdr[L.hdr_bits >> 5] : 0u;
#pragma unroll
    for (int j = 0; j < kRunsPerThread; j++)
        if (run_bits[j]) {
            const uint32_t q = (uint32_t)j * kDefThreads + (uint32_t)tid;
            const uint32_t before = q ? L.tail[q - 1] : hdr_tail;
            if ((run_off[j] & 31u) + run_bits[j] >= 32u)
                slot[run_off[j] >> 5] = head[j] | before;
            else
                L.tail[q] |= before;  // a run that stays inside its first word (only the short last run of the input can): hand everything on
        }
    __syncthreads();
    if (tid == 0) {
        // the last run's leftover bits, end-of-block, the empty stored block
        const uint32_t n_runs = (blen + kRun - 1) / kRun;
        uint64_t acc = L.tail[n_runs - 1];
        uint32_t nb = data_end & 31u, w = data_end >> 5;
        auto push = [&](uint32_t bits, uint32_t len) {
            acc |= (uint64_t)bits << nb;
            nb += len;
            if (nb >= 32u) {
                slot[w++] = (uint32_t)acc;
                acc >>= 32;
                nb -= 32u;
            }
        };
        push(eob & 0xFFFFu, eob >> 16);
        push(0u, 3);
        if (nb & 7u) push(0u, 8u - (nb & 7u));
        push(0u, 16);       // LEN = 0x0000
        push(0xFFFFu, 16);  // NLEN = 0xFFFF
        if (nb) slot[w] = (uint32_t)acc;
        block_bytes[blockIdx.x] = total_bytes;
    }
}

// member = [10-byte gzip header][block 0][block 1]...[03 00][CRC-32][ISIZE]; offs = exclusive scan of block_bytes
__global__ __launch_bounds__(256) void k_deflate_gather(const uint8_t *__restrict__ slots, const uint32_t *__restrict__ block_bytes,
                                                        const uint64_t *__restrict__ offs, uint8_t *__restrict__ out, uint32_t head) {
    const uint8_t *s = slots + (uint64_t)blockIdx.x * kSlotBytes;
    uint8_t *d = out + head + offs[blockIdx.x];
    const uint32_t n = block_bytes[blockIdx.x];
    // destination-aligned 4-byte stores, the source read bytewise (it sits in L2: the block was written a moment ago)
    const uint32_t lead = min(n, (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(d) & 3u)) & 3u));
    if (threadIdx.x < lead) d[threadIdx.x] = s[threadIdx.x];
    const uint32_t words = (n - lead) >> 2;
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) {
        const uint8_t *p = s + lead + 4u * w;
        reinterpret_cast<uint32_t *>(d + lead)[w] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    }
    const uint32_t done = lead + 4u * words;
    if (threadIdx.x < n - done) d[done + threadIdx.x] = s[done + threadIdx.x];
}

// BGZF: block b = [18-byte header with BSIZE][its deflate data][03 00][CRC-32][ISIZE]; offs = exclusive scan of (block_bytes + 28); the
// 28-byte end-of-file block behind the last one
__global__ __launch_bounds__(256) void k_bgzf_gather(const uint8_t *__restrict__ slots, const uint32_t *__restrict__ block_bytes, const uint64_t *__restrict__ offs,
                                                     const uint32_t *__restrict__ block_crc, uint64_t n_bytes, uint32_t block_in, uint32_t n_blocks,
                                                     uint8_t *__restrict__ out, uint64_t *__restrict__ total) {
    const uint32_t b = blockIdx.x;
    const uint8_t *s = slots + (uint64_t)b * kSlotBytes;
    const uint32_t n = block_bytes[b];
    uint8_t *d = out + offs[b] + 28ull * b;
    const uint32_t bsize = 18u + n + 2u + 8u;
    if (threadIdx.x == 0) {
        const uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bsize - 1u) & 0xFFu), (uint8_t)((bsize - 1u) >> 8)};
        for (int i = 0; i < 18; i++) d[i] = head[i];
        uint8_t *t = d + 18 + n;
        t[0] = 0x03;  // the empty final block of the member
        t[1] = 0x00;
        const uint32_t crc = block_crc[b];
        const uint32_t isize = (uint32_t)min((uint64_t)block_in, n_bytes - (uint64_t)b * block_in);
        for (int i = 0; i < 4; i++) t[2 + i] = (uint8_t)(crc >> (8 * i));
        for (int i = 0; i < 4; i++) t[6 + i] = (uint8_t)(isize >> (8 * i));
        if (b + 1 == n_blocks) {
            const uint8_t eof[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 28; i++) t[10 + i] = eof[i];
            total[0] = (uint64_t)(t + 38 - out);
        }
    }
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) d[18 + k] = s[k];
}

__global__ void k_deflate_finish(uint8_t *__restrict__ out, const uint64_t *__restrict__ offs, const uint32_t *__restrict__ block_bytes,
                                 uint32_t n_blocks, const uint32_t *__restrict__ state, uint64_t n_bytes, int gzip, uint64_t *__restrict__ total) {
    if (threadIdx.x || blockIdx.x) return;
    const uint32_t head = gzip ? 10u : 0u;
    if (gzip) {
        const uint8_t h[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255};  // deflate, no flags, no mtime, XFL 0, OS unknown
        for (int i = 0; i < 10; i++) out[i] = h[i];
    }
    uint64_t at = head + (n_blocks ? offs[n_blocks - 1] + block_bytes[n_blocks - 1] : 0ull);
    out[at++] = 0x03;  // BFINAL = 1, BTYPE = 01, end-of-block of the fixed code
    out[at++] = 0x00;
    if (gzip) {
        const uint32_t crc = state[0], isize = (uint32_t)n_bytes;
        for (int i = 0; i < 4; i++) out[at++] = (uint8_t)(crc >> (8 * i));
        for (int i = 0; i < 4; i++) out[at++] = (uint8_t)(isize >> (8 * i));
    }
    total[0] = at;
    total[1] = state[1];  // 1: a block outgrew its slot, 2: header overflow (neither is expected)
}

}  // namespace

size_t deflate_bound(size_t n_bytes) {
    const size_t n_blocks = (n_bytes + kDeflateBlock - 1) / kDeflateBlock;
    return n_bytes + n_bytes / 8 + n_blocks * 600 + 64;
}

size_t deflate_scratch_bytes(size_t n_bytes) {
    const size_t n_blocks = (n_bytes + kDeflateBlock - 1) / kDeflateBlock;
    size_t scan_tmp = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const uint32_t *)nullptr, (uint64_t *)nullptr, (int)std::max<size_t>(n_blocks, 1));
    return n_blocks * (size_t)kSlotBytes + n_blocks * 4 + (n_blocks + 1) * 8 + scan_tmp + 4096;  // (sections are 256-byte aligned)
}

// d_in[0 .. n_bytes) -> d_out: one gzip member (gzip != 0) or one raw deflate stream; *d_total (device, 8 bytes) = its size.
// d_scratch: deflate_scratch_bytes(n_bytes); d_out: deflate_bound(n_bytes) at least; d_total: two 8-byte words (size, error flags).
int launch_deflate(smi_ctx *ctx, const uint8_t *d_in, size_t n_bytes, uint8_t *d_out, size_t out_cap, uint8_t *d_scratch, uint64_t *d_total, int gzip,
                   hipStream_t s) {
    if (out_cap < deflate_bound(n_bytes)) {
        set_error("deflate: output buffer below deflate_bound()");
        return SMI_ERR_INVALID;
    }
    const size_t n_blocks = (n_bytes + kDeflateBlock - 1) / kDeflateBlock;
    if (n_blocks > 0x7FFFFFFFull) {
        set_error("deflate: input too large for one call");
        return SMI_ERR_INVALID;
    }
    uint8_t *slots = d_scratch;
    size_t at = (n_blocks * (size_t)kSlotBytes + 255) & ~(size_t)255;
    uint32_t *d_bb = reinterpret_cast<uint32_t *>(d_scratch + at);
    at = (at + n_blocks * 4 + 255) & ~(size_t)255;
    uint64_t *d_offs = reinterpret_cast<uint64_t *>(d_scratch + at);
    at = (at + (n_blocks + 1) * 8 + 255) & ~(size_t)255;
    uint32_t *d_state = reinterpret_cast<uint32_t *>(d_scratch + at);  // CRC accumulator, error flags
    at += 256;
    void *d_tmp = d_scratch + at;
    size_t scan_tmp = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const uint32_t *)nullptr, (uint64_t *)nullptr, (int)std::max<size_t>(n_blocks, 1));
    SMI_HIP(hipMemsetAsync(d_state, 0, 8, s));
    if (n_blocks) {
        hipLaunchKernelGGL(k_deflate_blocks, dim3((unsigned)n_blocks), dim3(kDefThreads), 0, s, d_in, (uint64_t)n_bytes, (uint32_t)kDeflateBlock, slots, d_bb, d_state,
                           d_state + 1, (uint32_t *)nullptr);
        SMI_HIP(hipGetLastError());
        SMI_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, scan_tmp, d_bb, d_offs, (int)n_blocks, s));
        hipLaunchKernelGGL(k_deflate_gather, dim3((unsigned)n_blocks), dim3(256), 0, s, slots, d_bb, d_offs, d_out, gzip ? 10u : 0u);
        SMI_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_deflate_finish, dim3(1), dim3(1), 0, s, d_out, d_offs, d_bb, (uint32_t)n_blocks, d_state, (uint64_t)n_bytes, gzip, d_total);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int ensure_deflate_scratch(smi_ctx *ctx, size_t n_bytes, size_t extra_bytes) {
    const size_t want = deflate_scratch_bytes(n_bytes) + extra_bytes;
    if (ctx->deflate_scratch_bytes >= want) return SMI_OK;
    if (ctx->deflate_scratch) SMI_HIP(hipFree(ctx->deflate_scratch));
    ctx->deflate_scratch = nullptr;
    ctx->deflate_scratch_bytes = 0;
    const size_t grown = want + want / 4;
    SMI_HIP(hipMalloc(&ctx->deflate_scratch, grown));
    ctx->deflate_scratch_bytes = grown;
    return SMI_OK;
}

int deflate_pair(smi_ctx *ctx, const uint8_t *d_a, size_t na, const uint8_t *d_b, size_t nb, uint8_t **d_za, uint8_t **d_zb, uint64_t **d_totals,
                 hipStream_t s) {
    const size_t ca = (deflate_bound(na) + 255) & ~(size_t)255, cb = (deflate_bound(nb) + 255) & ~(size_t)255;
    const size_t work = (deflate_scratch_bytes(std::max(na, nb)) + 255) & ~(size_t)255;
    if (int rc = ensure_deflate_scratch(ctx, std::max(na, nb), ca + cb + 1024)) return rc;
    uint8_t *base = static_cast<uint8_t *>(ctx->deflate_scratch);
    *d_za = base + work;
    *d_zb = base + work + ca;
    *d_totals = reinterpret_cast<uint64_t *>(base + work + ca + cb);
    // the two members share the block slots: same stream, one after the other
    if (int rc = launch_deflate(ctx, d_a, na, *d_za, ca, base, *d_totals, 1, s)) return rc;
    return launch_deflate(ctx, d_b, nb, *d_zb, cb, base, *d_totals + 2, 1, s);
}

}  // namespace smi

extern "C" size_t smi_deflate_bound(size_t n_bytes) { return smi::deflate_bound(n_bytes); }

// BGZF on the device: see sicelore_mi.h
extern "C" int smi_bgzf_deflate_device(smi_ctx *ctx, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *n_out) {
    using namespace smi;
    constexpr uint32_t kBgzfIn = 0xF000;  // 61,440 input bytes per block: a Huffman-only block of any content stays below the 64 KiB a BGZF block may have
    if ((!in && n_in) || !n_out || !ctx) {
        set_error("smi_bgzf_deflate_device: bad argument");
        return SMI_ERR_INVALID;
    }
    const size_t n_blocks = (n_in + kBgzfIn - 1) / kBgzfIn;
    const size_t bound = n_in + n_in / 8 + n_blocks * (600 + 28) + 28 + 64;
    if (!out) {
        *n_out = bound;
        return SMI_OK;
    }
    static const uint8_t kEof[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (n_blocks == 0) {
        if (cap_out < 28) {
            set_error("smi_bgzf_deflate_device: output buffer too small");
            return SMI_ERR_INVALID;
        }
        std::memcpy(out, kEof, 28);
        *n_out = 28;
        return SMI_OK;
    }
    if (cap_out < bound) {
        set_error("smi_bgzf_deflate_device: output buffer below the bound (call with out == NULL for it)");
        return SMI_ERR_INVALID;
    }
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // scratch: block slots | sizes | offsets | scan storage | state | CRCs | input | output
    size_t scan_tmp = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const uint32_t *)nullptr, (uint64_t *)nullptr, (int)n_blocks);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_bb = al(n_blocks * (size_t)kSlotBytes), o_offs = o_bb + al(n_blocks * 4), o_tmp = o_offs + al((n_blocks + 1) * 8), o_state = o_tmp + al(scan_tmp),
                 o_crc = o_state + 256, o_in = o_crc + al(n_blocks * 4), o_out = o_in + al(n_in + 64), end = o_out + al(bound);
    if (ctx->deflate_scratch_bytes < end) {
        if (ctx->deflate_scratch) SMI_HIP(hipFree(ctx->deflate_scratch));
        ctx->deflate_scratch = nullptr;
        ctx->deflate_scratch_bytes = 0;
        SMI_HIP(hipMalloc(&ctx->deflate_scratch, end + end / 4));
        ctx->deflate_scratch_bytes = end + end / 4;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->deflate_scratch);
    uint32_t *d_bb = reinterpret_cast<uint32_t *>(base + o_bb), *d_state = reinterpret_cast<uint32_t *>(base + o_state), *d_crc = reinterpret_cast<uint32_t *>(base + o_crc);
    uint64_t *d_offs = reinterpret_cast<uint64_t *>(base + o_offs);
    uint8_t *d_in = base + o_in, *d_out = base + o_out;
    SMI_HIP(hipMemcpyAsync(d_in, in, n_in, hipMemcpyDefault, s));
    SMI_HIP(hipMemsetAsync(d_state, 0, 64, s));
    hipLaunchKernelGGL(k_deflate_blocks, dim3((unsigned)n_blocks), dim3(kDefThreads), 0, s, d_in, (uint64_t)n_in, kBgzfIn, base, d_bb, d_state, d_state + 1, d_crc);
    size_t tmp = scan_tmp;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(base + o_tmp, tmp, d_bb, d_offs, (int)n_blocks, s));
    hipLaunchKernelGGL(k_bgzf_gather, dim3((unsigned)n_blocks), dim3(256), 0, s, base, d_bb, d_offs, d_crc, (uint64_t)n_in, kBgzfIn, (uint32_t)n_blocks, d_out,
                       reinterpret_cast<uint64_t *>(d_state + 4));
    SMI_HIP(hipGetLastError());
    uint64_t h_state[4] = {0, 0, 0, 0};
    SMI_HIP(hipMemcpyAsync(h_state, d_state, 32, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    const uint32_t flags = (uint32_t)(h_state[0] >> 32);  // d_state[1]
    const uint64_t total = h_state[2];                     // d_state[4..5]
    if (flags || total > cap_out) {
        set_error("smi_bgzf_deflate_device: a block outgrew its slot, or the output buffer is too small");
        return SMI_ERR_INVALID;
    }
    SMI_HIP(hipMemcpyAsync(out, d_out, total, hipMemcpyDefault, s));
    SMI_HIP(hipStreamSynchronize(s));
    *n_out = total;
    return SMI_OK;
}

extern "C" int smi_gzip_device(smi_ctx *ctx, const uint8_t *d_in, size_t n_bytes, uint8_t *d_out, size_t out_cap, uint64_t *d_total, int raw_deflate,
                               void *stream) {
    using namespace smi;
    if (!ctx || !d_out || !d_total || (n_bytes && !d_in)) {
        set_error("smi_gzip_device: null argument");
        return SMI_ERR_INVALID;
    }
    SMI_HIP(hipSetDevice(ctx->device));
    if (int rc = ensure_deflate_scratch(ctx, n_bytes, 0)) return rc;
    return launch_deflate(ctx, d_in, n_bytes, d_out, out_cap, static_cast<uint8_t *>(ctx->deflate_scratch), d_total, raw_deflate ? 0 : 1,
                          (hipStream_t)stream);
}
