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
//   pass A  the block's bytes -> symbol histogram (16 privatised copies in LDS) and the CRC-32 of each thread's 256-byte segment
//   tree    used symbols compacted and sorted by frequency (bitonic, LDS), code lengths by the in-place minimum-redundancy algorithm of
//           Moffat & Katajainen (one lane; 60 - 90 used symbols for FASTQ), frequencies halved and the tree rebuilt while a length exceeds
//           15; canonical codes, bit-reversed for the LSB-first stream
//   header  HLIT = 257 codes, HDIST = 1 (length 0: no distances), code lengths run-length coded with a FIXED complete code-length code
//           (sixteen-plus-two symbols, 4 or 5 bits), so no second tree is built
//   pass B  bits per thread, exclusive scan -> every thread's bit offset in the block's slot
//   pass C  codes packed into a 64-bit accumulator, whole words stored, the two boundary words of a thread OR-ed in atomically
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
constexpr int kSeg = kDeflateBlock / kDefThreads;                     // 256 bytes per thread
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
    uint32_t seg[256];   // x^(8 * 256 * k) mod P
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
    for (int k = 0; k < 256; k++) {
        t.seg[k] = s;
        s = gf_mul(s, t.x2n[8]);  // * x^(8 * 256)
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

struct BitBuf {  // LSB-first bit writer into 32-bit words of LDS (header and trailer: a few hundred bits, one lane)
    uint32_t *w;
    uint32_t n;
    __device__ __forceinline__ void put(uint32_t v, int bits) {
        const uint32_t at = n >> 5, sh = n & 31u;
        w[at] |= v << sh;
        if (sh + (uint32_t)bits > 32u) w[at + 1] |= v >> (32u - sh);
        n += (uint32_t)bits;
    }
};

struct DeflateLds {
    uint32_t hist[16][260];
    uint32_t sfreq[512];   // sorted: frequency
    uint16_t ssym[512];    //         symbol
    uint32_t tree[260];    // Moffat-Katajainen work array -> code lengths of the sorted symbols
    uint32_t code[260];    // per symbol: reversed code | length << 16
    uint32_t crc_tab[256];
    uint32_t hdr[96];      // header bits (<= 57 + 258 * 12 bits in the worst case = 395 bytes; 96 words = 384 bytes is the practical bound, checked)
    uint32_t wave_sum[4];
    uint32_t n_used, hdr_bits, blk_shift, bl_count[16], next_code[16];
};

// ---- one deflate block per workgroup ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kDefThreads) void k_deflate_blocks(const uint8_t *__restrict__ in, uint64_t n_bytes, uint8_t *__restrict__ slots,
                                                                uint32_t *__restrict__ block_bytes, uint32_t *__restrict__ crc_acc,
                                                                uint32_t *__restrict__ err) {
    __shared__ DeflateLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t b0 = (uint64_t)blockIdx.x * kDeflateBlock;
    const uint32_t blen = (uint32_t)min((uint64_t)kDeflateBlock, n_bytes - b0);
    const uint8_t *src = in + b0;
    uint32_t *slot = reinterpret_cast<uint32_t *>(slots + (uint64_t)blockIdx.x * kSlotBytes);
    // ---- tables, zeroes -------------------------------------------------------------------------------------------------------
    for (int i = tid; i < 16 * 260; i += kDefThreads) (&L.hist[0][0])[i] = 0;
    {
        uint32_t c = (uint32_t)tid;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        L.crc_tab[tid] = c;
    }
    for (int i = tid; i < 96; i += kDefThreads) L.hdr[i] = 0;
    for (int i = tid; i < 512; i += kDefThreads) {
        L.sfreq[i] = 0xFFFFFFFFu;
        L.ssym[i] = 0xFFFFu;
    }
    if (tid < 16) L.bl_count[tid] = 0;
    if (tid == 0) {
        L.n_used = 0;
        L.blk_shift = x_pow_bytes(n_bytes - (b0 + blen));  // x^(8 * bytes behind this block)
    }
    __syncthreads();
    // ---- pass A: histogram + CRC of my segment ---------------------------------------------------------------------------------
    const uint32_t s0 = (uint32_t)tid * kSeg, s1 = min(s0 + (uint32_t)kSeg, blen);
    uint32_t crc = 0xFFFFFFFFu;
    {
        uint32_t *h = L.hist[tid & 15];
        const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
        if (s0 + kSeg <= blen && aligned) {
            const uint4 *p = reinterpret_cast<const uint4 *>(src + s0);
            for (int k = 0; k < kSeg / 16; k++) {
                const uint4 v = p[k];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t c = (w[j] >> (8 * q)) & 0xFFu;
                        atomicAdd(&h[c], 1u);
                        crc = L.crc_tab[(crc ^ c) & 0xFFu] ^ (crc >> 8);
                    }
            }
        } else {
            for (uint32_t i = s0; i < s1; i++) {
                const uint32_t c = src[i];
                atomicAdd(&h[c], 1u);
                crc = L.crc_tab[(crc ^ c) & 0xFFu] ^ (crc >> 8);
            }
        }
    }
    crc = s1 > s0 ? ~crc : 0u;  // CRC-32 of the segment (0 for an empty one)
    {
        // bytes of this block behind my segment: whole segments for a full block (table), anything for the last one
        const uint32_t behind = blen - s1;
        const uint32_t f = (behind % (uint32_t)(kSeg) == 0 && s1 > s0) ? c_pow.seg[behind / kSeg] : x_pow_bytes(behind);
        uint32_t part = s1 > s0 ? gf_mul(gf_mul(f, L.blk_shift), crc) : 0u;
#pragma unroll
        for (int o = 32; o; o >>= 1) part ^= __shfl_xor(part, o);
        if (lane == 0) L.wave_sum[wave] = part;
    }
    __syncthreads();
    if (tid == 0) atomicXor(crc_acc, L.wave_sum[0] ^ L.wave_sum[1] ^ L.wave_sum[2] ^ L.wave_sum[3]);
    // ---- symbol frequencies -> sorted list of the used ones --------------------------------------------------------------------
    uint32_t f_mine = 0;
    for (int k = 0; k < 16; k++) f_mine += L.hist[k][tid];
    {
        // compact (order does not matter: the sort follows)
        const bool used = f_mine != 0;
        if (used) {
            const uint32_t at = atomicAdd(&L.n_used, 1u);
            L.sfreq[at] = f_mine;
            L.ssym[at] = (uint16_t)tid;
        }
        L.code[tid] = 0;
        if (tid == 0) L.code[256] = 0;
    }
    __syncthreads();
    if (tid == 0) {  // end-of-block, once
        const uint32_t at = L.n_used++;
        L.sfreq[at] = 1u;
        L.ssym[at] = 256;
    }
    __syncthreads();
    const uint32_t n_used = L.n_used;  // >= 2: a block holds at least one byte
    // bitonic sort of 512 (freq, sym) pairs, ascending by frequency then symbol (the unused slots hold the maximum)
    for (uint32_t k = 2; k <= 512; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = (uint32_t)tid; i < 512; i += kDefThreads) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const uint64_t a = ((uint64_t)L.sfreq[i] << 16) | L.ssym[i], b = ((uint64_t)L.sfreq[x] << 16) | L.ssym[x];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        L.sfreq[i] = (uint32_t)(b >> 16);
                        L.ssym[i] = (uint16_t)b;
                        L.sfreq[x] = (uint32_t)(a >> 16);
                        L.ssym[x] = (uint16_t)a;
                    }
                }
            }
            __syncthreads();
        }
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
    // ---- header (one lane) -----------------------------------------------------------------------------------------------------------
    if (tid == 0) {
        BitBuf B{L.hdr, 0};
        B.put(0u, 1);   // BFINAL = 0
        B.put(2u, 2);   // BTYPE = 10, dynamic Huffman
        B.put(0u, 5);   // HLIT: 257 literal / length codes
        B.put(0u, 5);   // HDIST: 1 distance code
        B.put(15u, 4);  // HCLEN: all 19 code-length codes
        const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int k = 0; k < 19; k++) B.put((uint32_t)c_cl.len[order[k]], 3);
        // 257 literal / length code lengths + the one distance code (length 0: "no distance codes", RFC 1951 3.2.7)
        int s = 0;
        while (s < kSyms + 1) {
            const uint32_t len = s < kSyms ? (L.code[s] >> 16) : 0u;
            if (len == 0) {
                int run = 1;
                while (s + run < kSyms + 1 && (s + run < kSyms ? (L.code[s + run] >> 16) : 0u) == 0u && run < 138) run++;
                if (run >= 11) {
                    B.put(c_cl.code[18], c_cl.len[18]);
                    B.put((uint32_t)(run - 11), 7);
                } else if (run >= 3) {
                    B.put(c_cl.code[17], c_cl.len[17]);
                    B.put((uint32_t)(run - 3), 3);
                } else {
                    run = 1;
                    B.put(c_cl.code[0], c_cl.len[0]);
                }
                s += run;
            } else {
                B.put(c_cl.code[len], c_cl.len[len]);
                s++;
            }
            if (B.n > 96 * 32 - 64) {  // (cannot happen: 257 lengths cost <= 5 bits each + 57 = 1342 bits)
                atomicOr(err, 2u);
                break;
            }
        }
        L.hdr_bits = B.n;
    }
    __syncthreads();
    // ---- pass B: bits per thread, exclusive scan ---------------------------------------------------------------------------------
    uint32_t my_bits = 0;
    for (uint32_t i = s0; i < s1; i++) my_bits += L.code[src[i]] >> 16;
    uint32_t inc = my_bits;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(inc, o);
        if (lane >= o) inc += y;
    }
    if (lane == 63) L.wave_sum[wave] = inc;
    __syncthreads();
    uint32_t before = L.hdr_bits;
    for (int w = 0; w < wave; w++) before += L.wave_sum[w];
    const uint32_t my_off = before + inc - my_bits;
    const uint32_t data_end = L.hdr_bits + L.wave_sum[0] + L.wave_sum[1] + L.wave_sum[2] + L.wave_sum[3];
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
    // ---- zero the words two writers share: the first word of every thread's run, and the trailer's words ---------------------------
    // header words are stored whole by their owner below, except the last (partial) one, which thread 0's first bits share
    if (s1 > s0) slot[my_off >> 5] = 0u;
    if (tid == 0) {
        const uint32_t w0 = data_end >> 5, w1 = (total_bytes + 3u) >> 2;
        for (uint32_t w = w0; w <= w1; w++) slot[w] = 0u;
        slot[L.hdr_bits >> 5] = 0u;
    }
    __syncthreads();
    // header: whole words plain, the last partial word OR-ed (it is thread 0's first word, or the trailer's if the block is tiny)
    {
        const uint32_t hw = L.hdr_bits >> 5;
        for (uint32_t w = (uint32_t)tid; w < hw; w += kDefThreads) slot[w] = L.hdr[w];
        if (tid == 0 && (L.hdr_bits & 31u)) atomicOr(&slot[hw], L.hdr[hw]);
    }
    // ---- pass C: the codes -------------------------------------------------------------------------------------------------------------
    if (s1 > s0) {
        uint32_t w = my_off >> 5;
        uint32_t nb = my_off & 31u;
        uint64_t acc = 0;
        bool first = true;
        for (uint32_t i = s0; i < s1; i++) {
            const uint32_t e = L.code[src[i]];
            acc |= (uint64_t)(e & 0xFFFFu) << nb;
            nb += e >> 16;
            if (nb >= 32u) {
                if (first) {
                    atomicOr(&slot[w], (uint32_t)acc);
                    first = false;
                } else
                    slot[w] = (uint32_t)acc;
                w++;
                acc >>= 32;
                nb -= 32u;
            }
        }
        if (nb) atomicOr(&slot[w], (uint32_t)acc);  // the next thread's first word (zeroed above), or the trailer's
    }
    if (tid == 0) {
        uint32_t at = data_end;
        auto put = [&](uint32_t v, uint32_t bits) {
            const uint32_t w = at >> 5, sh = at & 31u;
            atomicOr(&slot[w], v << sh);
            if (sh + bits > 32u) atomicOr(&slot[w + 1], v >> (32u - sh));
            at += bits;
        };
        put(eob & 0xFFFFu, eob >> 16);
        put(0u, 3);
        at = (at + 7u) & ~7u;
        put(0xFFFF0000u, 32);  // LEN = 0x0000, NLEN = 0xFFFF (little endian)
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
        hipLaunchKernelGGL(k_deflate_blocks, dim3((unsigned)n_blocks), dim3(kDefThreads), 0, s, d_in, (uint64_t)n_bytes, slots, d_bb, d_state, d_state + 1);
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
