// smi_bc.hip -- cell-barcode matching kernels for gfx950 (CDNA4), hand-written HIP.
//
// What the reference does per read (FJ!nanoporereadscanner/analyzers/Parser.java:L195-315,
// BarcodeMatchTester.java:L198-374): for each of 5 window offsets, probe the 16-mer and every sequence
// reachable by one "mutation cycle" (48 substitutions, 60 insertions, 15 deletions, in a fixed order) in a
// hash set of barcodes; the first hit per (offset, level) wins; then a best/second rule over the merged set.
//
// MI355X mapping: the hash set becomes a 3-level bit pyramid over the 2^32 key universe (smi_internal.h);
// one 64-lane wavefront owns one read at a time and its lanes ARE the enumeration order (lane e of round A
// is mutant e, lane e of round B is mutant 64+e, lane 63 of round B is the un-mutated window), so the
// reference's "first hit wins" is a ballot + count-trailing-zeros.  All 10 probe rounds of a read
// (5 offsets x 2) are issued level by level, giving 10 independent gathers in flight per wave and level.
// Integer/bitwise work only: no MFMA, no LDS (the pyramid's top level lives in L2, the window batch in
// registers).
#include <chrono>

#include <hipcub/hipcub.hpp>

#include "smi_internal.h"

namespace smi {

// ---------------------------------------------------------------------------------------------------------
// pyramid build
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lowmask_h(int nbits) { return nbits >= 32 ? 0xFFFFFFFFu : ((1u << nbits) - 1u); }  // nbits in [0, 32]
__device__ __forceinline__ uint32_t suffix_index(uint32_t k) { return ((k & 0x3FFFu) << (18 - kG0)) | (k >> (14 + kG0)); }

__global__ void k_set_bits(const uint32_t *__restrict__ keys, size_t n, uint32_t *__restrict__ l0,
                           uint32_t *__restrict__ l0s, uint32_t *__restrict__ l1, uint32_t *__restrict__ fine,
                           uint32_t *__restrict__ t2) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t k = keys[i];
        atomicOr(&fine[k >> 5], 1u << (k & 31));
        atomicOr(&l1[l1_word(k)], 1u << l1_bit(k));
        uint32_t i0 = k >> kG0;
        atomicOr(&l0[i0 >> 5], 1u << (i0 & 31));
        uint32_t is = suffix_index(k);
        atomicOr(&l0s[is >> 5], 1u << (is & 31));
        atomicOr(&t2[2 * (i0 >> 5)], 1u << (i0 & 31));
        atomicOr(&t2[2 * (i0 >> 5) + 1], 1u << t2_prefix_bit(k));
        atomicOr(&t2[2 * (kL0Words + (is >> 5))], 1u << (is & 31));
        atomicOr(&t2[2 * (kL0Words + (is >> 5)) + 1], 1u << t2_twin_bit(k));
    }
}

// K-BC2's item filter for short used lists.  A level-1 item X can only produce a level-2 hit if one mutation step of the reference's
// enumeration leads from X to a barcode, i.e. if X lies in the INVERSE one-step neighbourhood of the list.  Per barcode w that is (a superset
// of) 169 sequences: w itself; its 48 substitutions (symmetric); for "insert b behind position q, drop the last base" (q = 0..14) every X
// = w without position q + 1, left-shifted, with ANY last base (60); for "delete position q, append the next read base" (q = 0..14) every
// X = w with ANY base inserted at q, w's last base dropped (60; that w's last base equals the appended read base is not checked -- a
// superset is all the filter needs).  Stored like l1: word = X >> 10, bit = 5-bit hash of the low 10 bits (16 MiB); 5 k barcodes set
// 0.6 % of the bits, so 99 of 100 items are dismissed with one load instead of the 123 probes of their children.
// A second table of the same layout holds the cells that TWO OR MORE different barcodes reach: when the window itself is a barcode
// (the usual case at the true offset), every child of it has that barcode as a neighbour, but a level-2 mutant equal to the root is
// never probed (it is in the dedup set), so such an item only matters if ANOTHER barcode is one step away from it.
constexpr int kN1Slots = 169;
constexpr size_t kN1MaxKeys = 65536;  // beyond this the table is too dense to dismiss anything (169 x keys of 2^27 bits)
__device__ __forceinline__ uint32_t n1_member(uint32_t k, int slot) {
    if (slot < 48) return k ^ ((uint32_t)(slot % 3 + 1) << (30 - 2 * (slot / 3)));
    if (slot < 108) {
        const int j = slot - 48, p1 = 1 + j / 4, sh = 30 - 2 * p1;  // position that the insertion filled
        return (k & ~lowmask_h(sh + 2)) | ((k & lowmask_h(sh)) << 2) | (uint32_t)(j & 3);
    }
    if (slot < 168) {
        const int j = slot - 108, q = j / 4, top = 32 - 2 * q;       // position that the deletion removed
        return (k & ~lowmask_h(top)) | ((uint32_t)(j & 3) << (30 - 2 * q)) | ((k & lowmask_h(top)) >> 2);
    }
    return k;
}
__device__ __forceinline__ uint32_t n1_cell(uint32_t x) { return (l1_word(x) << 5) | l1_bit(x); }
// The same filter a second time, indexed by the LAST nine bases first (round 6).  The 123 children of a window are probed together, and in the layout
// above -- a 64-byte line per nine-base prefix -- the 52 children mutated at positions 9 .. 15 share the window's own line while the other 71 each
// have a line of their own: 72 misses per window, and the kernel's L2 misses ran at the memory system's random-access rate.  A child mutated at
// a position <= 6 keeps nine bases at its END: the substitutions end in the window's bases 7 .. 15, the insertions in 6 .. 14, the deletions in
// 8 .. 15 + the appended base -- three lines of a suffix-major table for all 56 of them.  Only the 15 children of positions 7 and 8 keep neither end.
// Both tables hold every member; a child is looked up in the one where its neighbours are.
__device__ __forceinline__ uint32_t n1_cell_s(uint32_t x) {
    const uint32_t rest = x >> 18;  // the first seven bases, folded into the nine bits inside the line
    return ((x & 0x3FFFFu) << 9) | ((rest ^ (rest >> 9)) & 511u);
}
// pass 1: owner[cell] = smallest index of a barcode that reaches the cell, n1 bit set
template <bool kSuffix>
__global__ void k_set_n1(const uint32_t *__restrict__ keys, size_t n, uint32_t *__restrict__ n1, uint32_t *__restrict__ owner) {
    const size_t total = n * kN1Slots;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(i / kN1Slots);
        const uint32_t x = n1_member(keys[b], (int)(i % kN1Slots));
        const uint32_t cell = kSuffix ? n1_cell_s(x) : n1_cell(x);
        atomicOr(&n1[cell >> 5], 1u << (cell & 31));
        atomicMin(&owner[cell], b);
    }
}
// pass 2: a cell that a barcode other than its owner reaches is reached by two
template <bool kSuffix>
__global__ void k_set_n2(const uint32_t *__restrict__ keys, size_t n, const uint32_t *__restrict__ owner, uint32_t *__restrict__ n2) {
    const size_t total = n * kN1Slots;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(i / kN1Slots);
        const uint32_t x = n1_member(keys[b], (int)(i % kN1Slots));
        const uint32_t cell = kSuffix ? n1_cell_s(x) : n1_cell(x);
        if (owner[cell] != b) atomicOr(&n2[cell >> 5], 1u << (cell & 31));
    }
}

// K-BC1's offset filter: the same inverse one-step neighbourhood, one exact bit per key (512 MiB).  An offset whose window K has its bit
// clear has no barcode among K and its 123 mutants, so none of its 124 probes is made.
//
// nb5: the same bits where the five probes of a read are neighbours.  The windows of the five offsets (-2 .. 2) are one 20-base stretch cut at five
// places, so they share their 12 middle bases: in the key of offset d (d = the offset for 3' barcoding, where the key is the reverse complement, and
// minus the offset for 5') that core sits at base positions 2 + d .. 13 + d, with 2 + d flank bases in front of it and 2 - d behind.  nb5 is indexed by
// the core first: 40 words per core, eight words (256 bits, one per value of the four flank bases) for each d.  A read's five filter bits then lie in
// 160 consecutive bytes -- three 64-byte sectors -- instead of five random sectors of the 512 MiB bitmap.  A window with an N has an unrelated key (the
// emulation in make_key) and simply lands elsewhere: every (d, key) pair has its own bit, so the test is nb's for any key.  2.5 GiB.
constexpr size_t kNb5Words = ((size_t)1 << 24) * 40;
__device__ __forceinline__ uint32_t nb5_bit_index(uint32_t key, int d, uint32_t &word) {
    const int sh = 2 * (2 - d);                                             // bits of the flank behind the core: 8, 6, 4, 2, 0
    const uint32_t core = (key >> sh) & 0xFFFFFFu;
    const uint32_t low = (1u << sh) - 1u;
    const uint32_t flank = ((key >> 24) & ~low & 0xFFu) | (key & low);      // the 2 + d leading bases above the 2 - d trailing ones
    word = core * 40u + (uint32_t)(d + 2) * 8u + (flank >> 5);
    return flank & 31u;
}

__global__ void k_set_nb(const uint32_t *__restrict__ keys, size_t n, uint32_t *__restrict__ nb, uint32_t *__restrict__ nb5) {
    const size_t total = n * kN1Slots;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t x = n1_member(keys[i / kN1Slots], (int)(i % kN1Slots));
        atomicOr(&nb[x >> 5], 1u << (x & 31));
        if (nb5) {  // (round 5's form: five more scattered atomics per member; kept behind SMI_BC1_NB5_ATOMIC as the cross-check of k_nb5_from_nb)
#pragma unroll
            for (int d = -2; d <= 2; d++) {
                uint32_t w;
                const uint32_t b = nb5_bit_index(x, d, w);
                atomicOr(&nb5[w], 1u << b);
            }
        }
    }
}

// nb5 DERIVED from nb (round 6): the same bits in another order, so the 2.5 GiB are a streaming transposition of the 512 MiB bitmap instead of five
// scattered atomics per neighbourhood member (k_set_nb took 133 ms for the 3.6 M list, 100 of them for nb5).  With key = [lead: 2 + d bases][core: 12][trail: 2 - d]
// and flank = lead << 2 (2 - d) | trail, bit (flank & 31) of nb5 word core * 40 + (d + 2) * 8 + (flank >> 5) is bit (key & 31) of nb word key >> 5:
//   d = -2  no lead: the eight words of a core are the eight consecutive nb words core * 8 ..                       (copy)
//   d = -1  one lead base: word j = nb word (j >> 1) << 25 | core << 1 | (j & 1)                                     (copy)
//   d =  0  two lead bases: word j = the half-word of core in nb word lead << 23 | core >> 1 for lead = 2 j, 2 j + 1  (2 x 2 half-words)
//   d =  1  three: nibble m of word j = nibble (core & 7) of nb word (8 j + m) << 21 | core >> 3                     (8 x 8 nibbles per 8 cores)
//   d =  2  four: bit b of word j = bit (core & 31) of nb word (32 j + b) << 19 | core >> 5                          (32 x 32 bits per 32 cores)
// A workgroup takes 512 consecutive cores: its inputs are 5 x 16 KiB in pieces of at least one 64-byte line (d = 2: 256 lead rows x 64 bytes), its output
// is 80 KiB contiguous; an item = (32 cores, d, j) gathers its <= 32 input words and leaves 32 output words in LDS, which then goes out in 16-byte pieces.
constexpr int kNb5TileCores = 512;
constexpr int kNb5TileWords = kNb5TileCores * 40;  // 20,480 words = 80 KiB of LDS
__global__ __launch_bounds__(256) void k_nb5_from_nb(const uint32_t *__restrict__ nb, uint32_t *__restrict__ nb5) {
    extern __shared__ uint32_t tile[];  // [core in tile][40]
    const uint32_t c0 = blockIdx.x * (uint32_t)kNb5TileCores;
    for (int item = threadIdx.x; item < (kNb5TileCores / 32) * 40; item += blockDim.x) {
        const int dj = item % 40, g = item / 40;  // consecutive lanes: consecutive words of a core's forty (LDS banks), the same 32 cores
        const int d = dj >> 3, j = dj & 7;         // d here = offset + 2
        const uint32_t c5 = (c0 >> 5) + (uint32_t)g;  // cores c5 * 32 .. + 31
        uint32_t *out = tile + (size_t)g * 32 * 40 + dj;
        if (d == 0) {
#pragma unroll 8
            for (int i = 0; i < 32; i++) out[i * 40] = nb[((size_t)c5 * 32 + i) * 8 + j];
        } else if (d == 1) {
            const size_t base = ((size_t)(j >> 1) << 25) + (size_t)c5 * 64 + (j & 1);
#pragma unroll 8
            for (int i = 0; i < 32; i++) out[i * 40] = nb[base + 2 * i];
        } else if (d == 2) {
            const size_t b0 = ((size_t)(2 * j) << 23) + (size_t)c5 * 16, b1 = ((size_t)(2 * j + 1) << 23) + (size_t)c5 * 16;
#pragma unroll 4
            for (int p = 0; p < 16; p++) {
                const uint32_t lo = nb[b0 + p], hi = nb[b1 + p];
                out[(2 * p) * 40] = (lo & 0xFFFFu) | (hi << 16);
                out[(2 * p + 1) * 40] = (lo >> 16) | (hi & 0xFFFF0000u);
            }
        } else if (d == 3) {
#pragma unroll 1
            for (int q = 0; q < 4; q++) {  // eight cores per input word
                uint32_t in[8];
#pragma unroll
                for (int m = 0; m < 8; m++) in[m] = nb[((size_t)(8 * j + m) << 21) + (size_t)c5 * 4 + q];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    uint32_t w = 0;
#pragma unroll
                    for (int m = 0; m < 8; m++) w |= ((in[m] >> (4 * i)) & 15u) << (4 * m);
                    out[(8 * q + i) * 40] = w;
                }
            }
        } else {
            uint32_t in[32];
#pragma unroll
            for (int b = 0; b < 32; b++) in[b] = nb[((size_t)(32 * j + b) << 19) + c5];
#pragma unroll 4
            for (int i = 0; i < 32; i++) {
                uint32_t w = 0;
#pragma unroll
                for (int b = 0; b < 32; b++) w |= ((in[b] >> i) & 1u) << b;
                out[i * 40] = w;
            }
        }
    }
    __syncthreads();
    uint4 *dst = reinterpret_cast<uint4 *>(nb5 + (size_t)c0 * 40);
    const uint4 *src = reinterpret_cast<const uint4 *>(tile);
    for (int k = threadIdx.x; k < kNb5TileWords / 4; k += blockDim.x) dst[k] = src[k];
}

// K-BC2's offset filter for short used lists: one exact bit per key for the inverse TWO-step neighbourhood (the inverse one-step
// neighbourhood of every member of the inverse one-step neighbourhood: 169 x 169 sequences per barcode, a superset of everything from
// which two mutations of the reference's enumeration reach the barcode, and of everything from which one or none does).  A window whose
// bit is clear has no match at level 0, 1 or 2 and the whole per-offset machinery of K-BC2 -- the 123 children, their dedup table and
// creation order -- is skipped for it.  The bitmap lives in the build scratch of the n2 table, which is idle once that is built.
constexpr size_t kNb2MaxKeys = 32768;  // 169^2 x keys <= 2^30: at most a fifth of all 16-mers set
__global__ void k_set_nb2(const uint32_t *__restrict__ keys, size_t n, uint32_t *__restrict__ nb2) {
    const size_t total = n * kN1Slots * kN1Slots;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / ((size_t)kN1Slots * kN1Slots);
        const int s12 = (int)(i - b * kN1Slots * kN1Slots);
        const uint32_t x = n1_member(n1_member(keys[b], s12 / kN1Slots), s12 % kN1Slots);
        atomicOr(&nb2[x >> 5], 1u << (x & 31));
    }
}

// K-BC1's neighbourhood table: for every (barcode w, step) pair the sequence X = n1_member(w, step) together with the ONE mutation of the
// reference's enumeration that leads from X back to w -- kind (0 X is w, 1 substitution, 2 insertion, 3 deletion), position, and the base
// that the step writes (substitution: w's base there; insertion: the inserted base; deletion: the base it appends, which must be the
// read's next base).  A window that the offset filter lets through is looked up here and its level-0 / level-1 matches are read off the
// entries: no mutant is generated and no membership is probed.  Open addressing over 8-byte slots, linear probing, 1.6 slots per pair.
__device__ __forceinline__ uint32_t nt_hash(uint32_t x) { return x * 0x9E3779B1u; }
// first slot of the 8-slot bucket (one 64-byte line) of x; cap is a multiple of 8
__device__ __forceinline__ uint32_t nt_slot(uint32_t x, uint32_t cap) { return (uint32_t)(((uint64_t)nt_hash(x) * (cap >> 3)) >> 32) << 3; }
__global__ void k_set_nt(const uint32_t *__restrict__ keys, size_t n, unsigned long long *__restrict__ nt, uint32_t cap) {
    const size_t total = n * kN1Slots;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t w = keys[i / kN1Slots];
        const int slot = (int)(i % kN1Slots);
        const uint32_t x = n1_member(w, slot);
        uint32_t kind, pos, base;
        if (slot < 48) {  // X = w with position pos changed: the step back substitutes w's base
            kind = 1u, pos = (uint32_t)(slot / 3), base = (w >> (30 - 2 * pos)) & 3u;
        } else if (slot < 108) {  // X = w without position p1: the step back inserts w[p1] behind position p1 - 1
            const int p1 = 1 + (slot - 48) / 4;
            kind = 2u, pos = (uint32_t)(p1 - 1), base = (w >> (30 - 2 * p1)) & 3u;
        } else if (slot < 168) {  // X = w with a base inserted at q: the step back deletes position q and appends w's last base
            kind = 3u, pos = (uint32_t)((slot - 108) / 4), base = w & 3u;
        } else {
            kind = 0u, pos = 0u, base = 0u;
        }
        const unsigned long long entry = (1ull << 40) | ((unsigned long long)x << 8) | (kind | (pos << 2) | (base << 6));
        // first free slot of the bucket, else of the next one: a bucket that still has a free slot has never overflowed
        uint32_t idx = nt_slot(x, cap);
        for (;;) {
            const unsigned long long old = atomicCAS(&nt[idx], 0ull, entry);
            if (old == 0ull || old == entry) break;
            idx = idx + 1 == cap ? 0u : idx + 1;
        }
    }
}

// The same table with the slot of an entry handed out by a COUNTER per bucket (round 6): one 32-bit atomicAdd into a 4-byte-per-bucket array (0.9 GB for
// the 3.6 M list) and a plain 8-byte store, instead of a 64-bit compare-and-swap on the 14.6 GB table itself (k_set_nt: 62.7 ms for 608 M entries, 9.7 G/s;
// the same number of 32-bit atomics on the 512 MiB bitmap run at 28 G/s).  A bucket's slots fill in counter order and an entry whose bucket is full goes to
// the next bucket, so "a bucket with a free slot has never overflowed" holds as before; which slot of a bucket an entry sits in was never defined (it
// depended on the order the threads arrived).  Needs DISTINCT barcodes (the compare-and-swap build drops an entry it meets again): the caller checks.
__global__ void k_set_nt_counted(const uint32_t *__restrict__ keys, size_t n, unsigned long long *__restrict__ nt, uint32_t *__restrict__ cnt, uint32_t cap) {
    const size_t total = n * kN1Slots;
    const uint32_t n_buckets = cap >> 3;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t w = keys[i / kN1Slots];
        const int slot = (int)(i % kN1Slots);
        const uint32_t x = n1_member(w, slot);
        uint32_t kind, pos, base;
        if (slot < 48) {
            kind = 1u, pos = (uint32_t)(slot / 3), base = (w >> (30 - 2 * pos)) & 3u;
        } else if (slot < 108) {
            const int p1 = 1 + (slot - 48) / 4;
            kind = 2u, pos = (uint32_t)(p1 - 1), base = (w >> (30 - 2 * p1)) & 3u;
        } else if (slot < 168) {
            kind = 3u, pos = (uint32_t)((slot - 108) / 4), base = w & 3u;
        } else {
            kind = 0u, pos = 0u, base = 0u;
        }
        const unsigned long long entry = (1ull << 40) | ((unsigned long long)x << 8) | (kind | (pos << 2) | (base << 6));
        uint32_t b = nt_slot(x, cap) >> 3;
        for (;;) {
            const uint32_t k = atomicAdd(&cnt[b], 1u);
            if (k < 8u) {
                nt[(size_t)b * 8 + k] = entry;
                break;
            }
            b = b + 1 == n_buckets ? 0u : b + 1;
        }
    }
}

// popcount of every 256-key block of the fine bitmap (8 words, read as two 16-B vectors)
__global__ void k_block_counts(const uint4 *__restrict__ fine, uint32_t *__restrict__ counts) {
    size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (b >= kRankEntries) return;
    uint4 a = fine[2 * b], c = fine[2 * b + 1];
    counts[b] = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(c.x) + __popc(c.y) + __popc(c.z) +
                __popc(c.w);
}

// bits set / an order-sensitive digest of a bitmap, entries of the table: what smi_set_stats reports (cross-check of the build kernels)
__global__ void k_bits_digest(const uint32_t *__restrict__ w, size_t n_words, unsigned long long *__restrict__ out) {
    unsigned long long bits = 0, dig = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long p = (unsigned long long)__popc(w[i]);
        bits += p;
        dig += p * ((i & 0xFFFFFu) + 1u);
    }
    for (int o = 32; o > 0; o >>= 1) {
        bits += __shfl_xor(bits, o);
        dig += __shfl_xor(dig, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], bits);
        atomicAdd(&out[1], dig);
    }
}
__global__ void k_nt_entries(const uint64_t *__restrict__ nt, size_t cap, unsigned long long *__restrict__ out) {
    unsigned long long cnt = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < cap; i += (size_t)gridDim.x * blockDim.x) cnt += nt[i] != 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, cnt);
}
int launch_set_digests(smi_ctx *ctx, uint64_t *out5, hipStream_t s) {
    unsigned long long *d = nullptr;
    SMI_HIP(hipMalloc((void **)&d, 5 * 8));
    SMI_HIP(hipMemsetAsync(d, 0, 5 * 8, s));
    if (ctx->nb_valid) hipLaunchKernelGGL(k_bits_digest, dim3(4096), dim3(256), 0, s, ctx->nb, kFineWords, d);
    if (ctx->nb5_valid) hipLaunchKernelGGL(k_bits_digest, dim3(4096), dim3(256), 0, s, ctx->nb5, kNb5Words, d + 2);
    if (ctx->nt_cap) hipLaunchKernelGGL(k_nt_entries, dim3(4096), dim3(256), 0, s, ctx->nt, (size_t)ctx->nt_cap, d + 4);
    unsigned long long h[5];
    hipError_t e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    SMI_HIP(e);
    out5[0] = h[0];  // bits of nb
    out5[1] = h[2];  // bits of nb5
    out5[2] = h[3];  // digest of nb5
    out5[3] = ctx->nt_cap;
    out5[4] = h[4];
    return SMI_OK;
}

// Zero fill of the big structures by a kernel of this library: hipMemsetAsync wrote the 14.6 GB neighbourhood table at 0.78 TB/s (18.8 ms of a 67 ms
// build, on its critical path); 16-byte stores from a grid-stride loop run at the rate a device copy writes.  Sizes are multiples of 16 (bitmaps and tables).
__global__ __launch_bounds__(256) void k_fill_zero16(uint4 *__restrict__ p, size_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = z;
}
static hipError_t fill_zero(void *p, size_t bytes, hipStream_t s) {
    if (bytes < ((size_t)64 << 20) || (bytes & 15) || ((uintptr_t)p & 15) || std::getenv("SMI_SET_MEMSET")) return hipMemsetAsync(p, 0, bytes, s);
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(k_fill_zero16, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 256 * 32)), dim3(256), 0, s, static_cast<uint4 *>(p), n16);
    return hipGetLastError();
}

int launch_build_pyramid(smi_ctx *ctx, const uint32_t *d_keys, size_t n, hipStream_t s, bool membership_only) {
    const auto t_build0 = std::chrono::steady_clock::now();
    const bool set_timing = std::getenv("SMI_SET_TIMING") != nullptr;  // host clock per phase on stderr (each mark waits for the stream: measurement only)
    auto mark = [&](const char *what) {
        if (!set_timing) return;
        (void)hipStreamSynchronize(s);
        if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
        std::fprintf(stderr, "set build: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count());
    };
    SMI_HIP(hipMemsetAsync(ctx->l0, 0, kL0Words * 4, s));
    SMI_HIP(hipMemsetAsync(ctx->l0s, 0, kL0Words * 4, s));
    SMI_HIP(hipMemsetAsync(ctx->l1, 0, kL1Words * 4, s));
    SMI_HIP(hipMemsetAsync(ctx->t2, 0, 4 * kL0Words * 4, s));
    SMI_HIP(fill_zero(ctx->fine, kFineWords * 4, s));
    if (n) {
        unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_set_bits, dim3(grid), dim3(256), 0, s, d_keys, n, ctx->l0, ctx->l0s, ctx->l1, ctx->fine, ctx->t2);
        SMI_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_block_counts, dim3((unsigned)(kRankEntries / 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint4 *>(ctx->fine), ctx->block_counts);
    SMI_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, ctx->block_counts, ctx->rank, (int)kRankEntries, s));
    struct Tmp {  // freed on every path out of this function
        void *p = nullptr;
        ~Tmp() {
            if (p) (void)hipFree(p);
        }
    } tmp;
    mark("pyramid bits + block counts");
    SMI_HIP(hipMalloc(&tmp.p, tmp_bytes));
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, ctx->block_counts, ctx->rank, (int)kRankEntries, s));
    // distinct keys = rank[last] + counts[last]
    uint32_t last[2] = {0, 0};
    SMI_HIP(hipMemcpyAsync(&last[0], ctx->rank + (kRankEntries - 1), 4, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(&last[1], ctx->block_counts + (kRankEntries - 1), 4, hipMemcpyDeviceToHost, s));
    ctx->nb_valid = false;
    ctx->nb5_valid = false;
    ctx->nt_cap = 0;
    mark("rank scan");
    if (n > 0 && !membership_only && !std::getenv("SMI_BC1_NO_FILTER")) {  // (the switch: tests run K-BC1 with and without the filter)
        if (!ctx->nb) SMI_HIP(hipMalloc((void **)&ctx->nb, kFineWords * 4));
        const unsigned gb = (unsigned)std::min<size_t>((n * kN1Slots + 255) / 256, 256 * 256);
        // nb5 goes with the table path only (its kernel is the one that reads it); a device that cannot spare 2.5 GiB keeps the plain bitmap
        ctx->nb5_valid = false;
        const bool want_nb5 = !std::getenv("SMI_BC1_NO_TABLE") && !std::getenv("SMI_BC1_NO_NB5");  // (the second switch: cross-checks of the two layouts)
        if (want_nb5 && !ctx->nb5 && hipMalloc((void **)&ctx->nb5, kNb5Words * 4) != hipSuccess) {
            ctx->nb5 = nullptr;
            (void)hipGetLastError();
        }
        // Round 6: the bitmap chain (nb, then nb5 from it) and the table are independent and bound by different things (atomics on 512 MiB against
        // atomics / stores over gigabytes): the chain runs on the context's side stream beside the table's kernel.  SMI_SET_ONE_STREAM: one after the other.
        hipStream_t sb = s;
        const bool two_streams = !std::getenv("SMI_SET_ONE_STREAM") && !std::getenv("SMI_BC1_NO_TABLE");
        if (two_streams) {
            if (!ctx->side_stream) {
                SMI_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
                SMI_HIP(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
                SMI_HIP(hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming));
            }
            SMI_HIP(hipEventRecord(ctx->side_fork, s));
            SMI_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
            sb = ctx->side_stream;
        }
        // from here to the join an error must not leave the side stream running on buffers the caller may free: every exit waits for it
        auto join = [&]() -> int {
            if (!two_streams) return SMI_OK;
            if (hipEventRecord(ctx->side_join, ctx->side_stream) != hipSuccess || hipStreamWaitEvent(s, ctx->side_join, 0) != hipSuccess) {
                (void)hipStreamSynchronize(ctx->side_stream);
                return hip_fail(hipGetLastError(), "smi_set_barcode_set (side stream)");
            }
            return SMI_OK;
        };
#define SMI_SET_HIP(call)                                \
    do {                                                 \
        const hipError_t e_ = (call);                    \
        if (e_ != hipSuccess) {                          \
            (void)hipStreamSynchronize(ctx->side_stream ? ctx->side_stream : s); \
            return hip_fail(e_, #call);                  \
        }                                                \
    } while (0)
        SMI_SET_HIP(fill_zero(ctx->nb, kFineWords * 4, sb));
        // nb5 by scattered atomics (round 5's way) instead of the transposition: the cross-check switch, and the way of SHORT lists -- the transposition streams
        // the whole 512 MiB bitmap into 2.5 GiB whatever the list holds (3.7 ms), five atomics per neighbour of a used list are a few microseconds behind a
        // 0.6 ms zero fill.  SMI_BC1_NB5_TRANSPOSE keeps the transposition for every list.
        const bool nb5_atomic = std::getenv("SMI_BC1_NB5_ATOMIC") != nullptr || (n <= kN1MaxKeys && !std::getenv("SMI_BC1_NB5_TRANSPOSE"));
        if (want_nb5 && ctx->nb5 && nb5_atomic) SMI_SET_HIP(fill_zero(ctx->nb5, kNb5Words * 4, sb));
        hipLaunchKernelGGL(k_set_nb, dim3(gb), dim3(256), 0, sb, d_keys, n, ctx->nb, want_nb5 && nb5_atomic ? ctx->nb5 : nullptr);
        SMI_SET_HIP(hipGetLastError());
        if (want_nb5 && ctx->nb5 && !nb5_atomic) {
            static const bool lds_ok = [] { return hipFuncSetAttribute(reinterpret_cast<const void *>(k_nb5_from_nb), hipFuncAttributeMaxDynamicSharedMemorySize, kNb5TileWords * 4) == hipSuccess; }();
            if (!lds_ok) {
                (void)hipStreamSynchronize(sb);
                set_error("smi_set_barcode_set: 80 KiB of LDS per workgroup refused (k_nb5_from_nb)");
                return SMI_ERR_HIP;
            }
            hipLaunchKernelGGL(k_nb5_from_nb, dim3((unsigned)(((size_t)1 << 24) / kNb5TileCores)), dim3(256), kNb5TileWords * 4, sb, ctx->nb, ctx->nb5);  // (every word of nb5 is written: no memset)
            SMI_SET_HIP(hipGetLastError());
        }
        ctx->nb_valid = true;
        ctx->nb5_valid = want_nb5 && ctx->nb5 != nullptr;
        ctx->nt_cap = 0;
        if (!std::getenv("SMI_BC1_NO_TABLE")) {  // (the switch: the filtered kernel that enumerates the mutants of the flagged offsets)
            const size_t pairs = n * (size_t)kN1Slots;
            // Slots per pair.  A look-up walks on to the next bucket when its own is full, and a WAVE walks on when one of its ~ 20 look-ups does: at
            // 1.6 slots per pair (five entries per eight-slot bucket on average) 13 % of the buckets are full and nearly every wave paid a second bucket
            // (a dependent miss and ~ 200 instructions: 458 VALU instructions per wave where one bucket takes ~ 320); at 3 slots per pair 0.6 % are and one
            // wave in nine does.  14.6 GB for the 3.6 M list (7.8 GB at 1.6, which a device short of memory still gets; without either: the enumerating
            // kernel behind the offset filter).
            auto cap_at = [&](size_t tenths) { return (std::max<size_t>(4096, pairs * tenths / 10) + 7) & ~(size_t)7; };
            size_t cap = cap_at(30);
            if (cap >= 0xFFFFFFFFull) cap = cap_at(16);
            if (cap < 0xFFFFFFFFull) {
                // (advisor, round 5) a table that was allocated at the 1.6-slot size because the 3-slot one did not fit is KEPT for lists of that size:
                // no free + failing malloc + malloc on every later load
                if (ctx->nt_alloc < cap && ctx->nt_alloc >= cap_at(16) && ctx->nt_small_only) cap = cap_at(16);
                if (ctx->nt_alloc < cap) {
                    (void)hipStreamSynchronize(sb);  // (hipFree synchronises the device anyway)
                    if (ctx->nt) SMI_SET_HIP(hipFree(ctx->nt));
                    ctx->nt = nullptr;
                    ctx->nt_alloc = 0;
                    ctx->nt_small_only = false;
                    // (+ cap / 8 counters of four bytes behind the table: the scratch of k_set_nt_counted, 6 % of the table)
                    if (hipMalloc((void **)&ctx->nt, cap * sizeof(uint64_t) + (cap >> 3) * sizeof(uint32_t)) == hipSuccess)
                        ctx->nt_alloc = cap;
                    else {
                        ctx->nt = nullptr;
                        (void)hipGetLastError();
                        cap = cap_at(16);
                        if (hipMalloc((void **)&ctx->nt, cap * sizeof(uint64_t) + (cap >> 3) * sizeof(uint32_t)) == hipSuccess) {
                            ctx->nt_alloc = cap;
                            ctx->nt_small_only = true;
                        } else {
                            ctx->nt = nullptr;
                            (void)hipGetLastError();
                        }
                    }
                }
                if (ctx->nt_alloc >= cap) {
                    SMI_SET_HIP(fill_zero(ctx->nt, cap * sizeof(uint64_t), s));
                    // distinct barcodes (the usual case): slots by per-bucket counters; else the compare-and-swap build, which drops repeated entries
                    SMI_SET_HIP(hipStreamSynchronize(s));  // (`last` has arrived: the number of distinct keys)
                    const bool distinct = (size_t)last[0] + last[1] == n;
                    // (the counters live behind the table's nt_alloc slots: allocated with it, idle between builds)
                    uint32_t *cnt = distinct && !std::getenv("SMI_SET_NT_CAS") ? reinterpret_cast<uint32_t *>(ctx->nt + ctx->nt_alloc) : nullptr;
                    if (cnt) {
                        SMI_SET_HIP(fill_zero(cnt, (cap >> 3) * sizeof(uint32_t), s));
                        hipLaunchKernelGGL(k_set_nt_counted, dim3(gb), dim3(256), 0, s, d_keys, n, reinterpret_cast<unsigned long long *>(ctx->nt), cnt, (uint32_t)cap);
                    } else
                        hipLaunchKernelGGL(k_set_nt, dim3(gb), dim3(256), 0, s, d_keys, n, reinterpret_cast<unsigned long long *>(ctx->nt), (uint32_t)cap);
                    SMI_SET_HIP(hipGetLastError());
                    ctx->nt_cap = (uint32_t)cap;
                }
            }
        }
        if (int rc = join()) return rc;
#undef SMI_SET_HIP
    }
    mark("nb / nb5 / nt");
    ctx->n1_valid = false;
    ctx->nb2_valid = false;
    if (n > 0 && !membership_only && n <= kN1MaxKeys && !std::getenv("SMI_BC2_NO_FILTER")) {  // (the switch: tests run K-BC2 with and without the filter)
        // four tables: n1, n2 (prefix-major), n1s, n2s (suffix-major: n1_cell_s); the owner scratch serves one layout after the other
        if (!ctx->n1) SMI_HIP(hipMalloc((void **)&ctx->n1, 4 * kL1Words * 4));
        SMI_HIP(hipMemsetAsync(ctx->n1, 0, 4 * kL1Words * 4, s));
        if (!ctx->n1_owner) SMI_HIP(hipMalloc((void **)&ctx->n1_owner, kL1Words * 32 * sizeof(uint32_t)));  // one u32 per cell
        const unsigned g1 = (unsigned)std::min<size_t>((n * kN1Slots + 255) / 256, 256 * 64);
        SMI_HIP(hipMemsetAsync(ctx->n1_owner, 0xFF, kL1Words * 32 * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_set_n1<false>, dim3(g1), dim3(256), 0, s, d_keys, n, ctx->n1, ctx->n1_owner);
        hipLaunchKernelGGL(k_set_n2<false>, dim3(g1), dim3(256), 0, s, d_keys, n, ctx->n1_owner, ctx->n1 + kL1Words);
        ctx->n1s_valid = !std::getenv("SMI_BC2_ONE_FILTER");  // (cross-check switch: every child through the prefix-major tables)
        if (ctx->n1s_valid) {
            SMI_HIP(hipMemsetAsync(ctx->n1_owner, 0xFF, kL1Words * 32 * sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_set_n1<true>, dim3(g1), dim3(256), 0, s, d_keys, n, ctx->n1 + 2 * kL1Words, ctx->n1_owner);
            hipLaunchKernelGGL(k_set_n2<true>, dim3(g1), dim3(256), 0, s, d_keys, n, ctx->n1_owner, ctx->n1 + 3 * kL1Words);
        }
        SMI_HIP(hipGetLastError());
        ctx->n1_valid = true;
        ctx->nb2_valid = false;
        mark("n1 / n2 (two layouts)");
        if (n <= kNb2MaxKeys && !std::getenv("SMI_BC2_NO_OFFSET_FILTER")) {
            // stream order: k_set_n2 has read the scratch as the owner array before it is cleared and filled as the two-step bitmap
            SMI_HIP(fill_zero(ctx->n1_owner, kFineWords * 4, s));
            const unsigned g2 = (unsigned)std::min<size_t>((n * kN1Slots * kN1Slots + 255) / 256, 256 * 256);
            hipLaunchKernelGGL(k_set_nb2, dim3(g2), dim3(256), 0, s, d_keys, n, ctx->n1_owner);
            SMI_HIP(hipGetLastError());
            ctx->nb2_valid = true;
        }
    }
    mark("nb2");
    SMI_HIP(hipStreamSynchronize(s));
    ctx->n_keys = (size_t)last[0] + last[1];
    ctx->set_build_us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_build0).count();
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K-WIN: window extraction (Parser.java:L205-221: substring + 2-bit packing, without the reverse complement,
// which the matcher applies).  One thread per read; 24-25 scattered bytes per read.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t base_code(uint8_t c, bool &is_n) {
    // BASE_TO_TWOBIT_ARRAY, NucleicAcidTwoBitPerBase.java:L78-87
    switch (c) {
    case 'A': case 'a': return 0;
    case 'G': case 'g': return 1;
    case 'C': case 'c': return 2;
    case 'T': case 't': return 3;
    default: is_n = true; return 0;
    }
}

__global__ void k_extract_windows(const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets,
                                  const int32_t *__restrict__ adapter_end, size_t n, int five_prime,
                                  smi_bc_window *__restrict__ out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t beg = offsets[i];
    const int64_t len = (int64_t)(offsets[i + 1] - beg);
    const int64_t ae = adapter_end[i];
    const int W = five_prime ? SMI_WIN_BASES_5P : SMI_WIN_BASES_3P;
    // first window base, 1-based stranded coordinate
    const int64_t first = five_prime ? ae - 1 : ae - 22;
    smi_bc_window w;
    w.bases = 0;
    w.nmask = 0;
    w.flags = 0;
    // every substring() of the five offsets must lie inside the read, else the reference throws
    bool ok = ae > 0 && first >= 1 && first + W - 1 <= len;
    if (ok) {
        const uint8_t *p = reads + beg + (first - 1);
        uint64_t b = 0;
        uint32_t nm = 0;
        for (int j = 0; j < W; j++) {
            bool is_n = false;
            uint32_t c = base_code(p[j], is_n);
            b = (b << 2) | c;
            nm |= (uint32_t)is_n << j;
        }
        w.bases = b;
        w.nmask = nm;
        w.flags = SMI_WIN_VALID;
    }
    out[i] = w;
}

int launch_extract_windows(smi_ctx *, const uint8_t *d_reads, const uint64_t *d_offsets, const int32_t *d_ae, size_t n,
                           int five_prime, smi_bc_window *d_win, hipStream_t s) {
    if (!n) return SMI_OK;
    hipLaunchKernelGGL(k_extract_windows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_reads, d_offsets, d_ae,
                       n, five_prime, d_win);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K-BC1: ed <= 1 matcher
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t revcomp16(uint32_t w) {
    // reverse the 16 2-bit groups and complement (REVERSE_COMP_ARRAY {3,2,1,0}: complement = 3 - code)
    uint32_t r = __brev(w);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    return ~r;
}

__device__ __forceinline__ uint32_t lowmask(int nbits) {  // nbits in [0, 32]
    return nbits >= 32 ? 0xFFFFFFFFu : ((1u << nbits) - 1u);
}

// Per-offset key material derived from the window record (wave-uniform).
struct OffsetKey {
    uint32_t key;      // un-mutated 16-mer in barcode orientation
    uint32_t del_base; // 2-bit code appended by a level-1 "deletion" (post[1], BarcodeMatchTester.java:L329)
    bool usable;       // false: N-poisoned 5' window -> no probe can hit
};

__device__ __forceinline__ OffsetKey make_key(uint64_t bases, uint32_t nmask, int o, bool five_prime) {
    OffsetKey k;
    if (!five_prime) {
        // window = base indices 6+o .. 21+o of the 24-base record (stranded orientation)
        const int j0 = 6 + o;
        uint32_t w = (uint32_t)(bases >> (2 * (24 - 16 - j0)));
        uint32_t nm = (nmask >> j0) & 0xFFFFu;
        if (nm) {
            // getLongHashForSeq ORs (long)-2 for a non-ACGT char (L185) and reverseComplement keeps only the low
            // 32 bits (L477-484): every base before the LAST N reads as T, the N itself as C.
            int i = 31 - __clz(nm);  // window index of the last N
            uint32_t keep = lowmask(30 - 2 * i);
            w = (w & keep) | (0xFFFFFFFFu << (31 - 2 * i));
        }
        k.key = revcomp16(w);
        k.usable = true;
        // post = revcomp(substring(bcStart-5, bcStart)) -> post[1] = complement of stranded[bcStart] = window base 0
        // taken from the 4-bit string: N stays N and BYTE_TO_2BITLONG_ARRAY[0][15] = 0 (L92-98)
        uint32_t b0 = (uint32_t)(bases >> (2 * (24 - 1 - j0))) & 3u;
        k.del_base = ((nmask >> j0) & 1u) ? 0u : (3u - b0);
    } else {
        // window = base indices 2+o .. 17+o of the 25-base record; post[1] = index 18+o
        const int j0 = 2 + o;
        k.key = (uint32_t)(bases >> (2 * (25 - 16 - j0)));
        k.usable = ((nmask >> j0) & 0xFFFFu) == 0;
        const int jp = 18 + o;
        uint32_t bp = (uint32_t)(bases >> (2 * (25 - 1 - jp))) & 3u;
        k.del_base = ((nmask >> jp) & 1u) ? 0u : bp;
    }
    return k;
}

// Lane-constant description of the mutant a lane generates in one round.
//   e = 8*p + r : p = position 0..15; r = 0..2 substitutions (ascending base, current base skipped,
//   BarcodeMatchTester.java:L259-260), r = 3..6 insertions of A,G,C,T after p (L286, SET_BITS order),
//   r = 7 deletion of p (L330).  Position 15 has substitutions only (L234).  e == 127: the window itself.
struct LaneMut {
    int s;           // 30 - 2p : bit offset of base p
    uint32_t lm_s;   // lowmask(s)
    uint32_t lm_s2;  // lowmask(s + 2)
    int r;
    bool valid;
    bool exact;
    bool far;  // mutant changes the leading 7 bases: probe the suffix-major top level instead of the prefix one
};

__device__ __forceinline__ LaneMut make_lane(int e) {
    LaneMut m;
    const int p = (e >> 3) & 15;
    m.r = e & 7;
    m.s = 30 - 2 * p;
    m.lm_s = lowmask(m.s);
    m.lm_s2 = lowmask(m.s + 2);
    m.exact = (e == 127);
    m.valid = (e < 123) || m.exact;
    // substitution at p changes base p, insertion after p bases p+1.., deletion of p bases p..
    // which mutants go to the suffix-major twin: measured (ms per 10 M reads) sub/ins/del <= 6/5/6: 8.4, 7/6/7: 7.5,
    // 8/7/8: 7.3, 8/6/8: 7.4, 8/8/8: 8.5, 9/8/9: 9.0 -- positions 6..8 lie in the bits the twin drops, so their
    // substitutions test the window's own bit there
    m.far = !m.exact && (m.r < 3 ? p <= 8 : (m.r < 7 ? p <= 7 : p <= 8));
    return m;
}

// Returns the mutant; ok=false where the reference's 64-bit value cannot equal a 32-bit barcode.
__device__ __forceinline__ uint32_t mutate(const LaneMut &m, uint32_t K, uint32_t del_base, bool &ok) {
    ok = m.valid;
    if (m.exact) return K;
    const uint32_t cur = (K >> m.s) & 3u;
    // substitution: getLongHashReplaceByteDeg L228-233
    const uint32_t j = (uint32_t)m.r;
    const uint32_t b = j + (j >= cur ? 1u : 0u);
    const uint32_t sub = K ^ ((cur ^ b) << m.s);
    // insertion: getLongHashInsertByteDeg L300-309; at p == 14 the Java shift count 64 wraps to 0 and the dropped
    // last base survives in bits 62..63, so the value is a barcode only when that base is A
    const uint32_t x = (uint32_t)(m.r - 3) & 3u;
    const int xs = m.s >= 2 ? m.s - 2 : 0;
    const uint32_t ins = (K & ~m.lm_s) | ((K & m.lm_s) >> 2) | (x << xs);
    // deletion: getLongHashdeleteByte L321-327
    const uint32_t del = (K & ~m.lm_s2) | ((K & m.lm_s) << 2) | del_base;
    if (m.r >= 3 && m.r <= 6 && m.s == 2 && (K & 3u) != 0u) ok = false;
    return m.r < 3 ? sub : (m.r < 7 ? ins : del);
}

// which kind of mutant is enumeration index e: ins - del contribution (OneMatch.getOffsetForReadEnd, L533;
// insertions() bumps nDeletions, deletions() bumps nInsertions: BarcodeMatchTester.java:L289,L346)
__device__ __forceinline__ int ins_minus_del_of(int e) {
    const int r = e & 7;
    return r < 3 ? 0 : (r < 7 ? -1 : 1);
}

// java.util.HashSet<OneMatch> iteration order + Stream.sorted() + distinctByKey (Parser.java:L244-252).
// Candidate slot i = 2*q + level for the q-th tested offset (order 0,-1,+1,-2,+2), i.e. slots are already in
// HashSet insertion order; `present` marks the filled ones.  HashMap: index = spread(hash) & (cap-1), cap 16,
// doubled when size > 0.75 cap or (below 64 buckets) when a node is appended to a bin already holding >= 8
// nodes; bins keep insertion order.  OneMatch.hashCode = (int)(readSeq ^ readSeq >>> 32) = the 32-bit window key
// (BarcodeMatchTester.java:L443); OneMatch.compareTo = (ed, offset == 0 first) (L449-461).
__device__ __forceinline__ void pick_best(const uint32_t (&bc)[10], const uint32_t (&rs)[10], const int (&imd)[10],
                                          uint32_t present, int max_ed, smi_bc_result &res) {
    constexpr int OFFS[5] = {0, -1, 1, -2, 2};
    const int n = __popc(present);
    res.bc = 0;
    res.ed_sec = 2147483647;
    res.found = 0;
    res.ed = 0;
    res.offset = 0;
    res.ins_minus_del = 0;
    res.n_matches = (uint32_t)n;
    if (n == 0) return;
    uint32_t h[10];
#pragma unroll
    for (int i = 0; i < 10; i++) h[i] = rs[i] ^ (rs[i] >> 16);
    int cap = 16;
    if (n >= 9) {  // the only way the table can grow with <= 10 elements: >= 8 nodes already in the target bin
        int size = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            if (!((present >> i) & 1u)) continue;
            int in_bin = 0;
#pragma unroll
            for (int k = 0; k < i; k++)
                in_bin += (((present >> k) & 1u) && ((h[k] ^ h[i]) & (uint32_t)(cap - 1)) == 0) ? 1 : 0;
            size++;
            if (in_bin >= 8 && cap < 64) cap <<= 1;
            if (size > (cap * 3) / 4) cap <<= 1;
        }
    }
    // total order: (ed, offset != 0, bin, insertion index)
    uint32_t key[10];
    uint32_t best_key = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        key[i] = ((uint32_t)(i & 1) << 20) | ((i >> 1) != 0 ? (1u << 16) : 0u) | ((h[i] & (uint32_t)(cap - 1)) << 8) |
                 (uint32_t)i;
        if (!((present >> i) & 1u)) key[i] = 0xFFFFFFFFu;
        best_key = min(best_key, key[i]);
    }
    uint32_t best_bc = 0;
    int best_imd = 0;
#pragma unroll
    for (int i = 0; i < 10; i++)
        if (key[i] == best_key) {
            best_bc = bc[i];
            best_imd = imd[i];
        }
    uint32_t second_key = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 10; i++)
        if (key[i] != 0xFFFFFFFFu && bc[i] != best_bc) second_key = min(second_key, key[i]);
    const int best_ed = (int)(best_key >> 20);
    const int second_ed = (int)(second_key >> 20);
    const bool has_second = second_key != 0xFFFFFFFFu;
    if (best_ed > max_ed) return;                      // L251
    if (has_second && best_ed >= second_ed) return;    // L252
    res.found = 1;
    res.bc = best_bc;
    res.ed = (int8_t)best_ed;
    res.ed_sec = has_second ? second_ed : 2147483647;  // L288
    res.offset = (int8_t)OFFS[(best_key & 0xFF) >> 1];
    res.ins_minus_del = (int8_t)best_imd;
}

__device__ __forceinline__ uint32_t bit_of(const uint32_t *__restrict__ words, uint32_t idx) {
    return (words[idx >> 5] >> (idx & 31)) & 1u;
}

// Lane-constant masks for the mutant a lane generates (same enumeration as LaneMut / mutate, 14 VALU ops instead
// of ~30): mutant = (K & keep) | ((K & low) >> 2 & m_ins) | ((K & low) << 2 & m_del) | ins_const | sub base | del base
struct LaneMasks {
    uint32_t keep, low, m_ins, m_del, ins_const, m_sub, m_db;
    int s;            // bit offset of base p
    uint32_t r;       // kind index (substitution rank for r < 3)
    uint32_t p14_ins; // all-ones for "insert behind position 14" (the mutant is a barcode only when the last base is A)
    uint32_t valid;   // all-ones when the lane generates a mutant in this round
    uint32_t exact;   // all-ones for the lane that probes the window itself
    bool far;
};

__device__ __forceinline__ LaneMasks make_masks(int e) {
    LaneMasks m;
    const LaneMut b = make_lane(e);
    m.s = b.s;
    m.r = (uint32_t)b.r;
    m.far = b.far;
    m.valid = b.valid ? 0xFFFFFFFFu : 0u;
    m.exact = b.exact ? 0xFFFFFFFFu : 0u;
    const bool is_sub = !b.exact && b.r < 3, is_ins = !b.exact && b.r >= 3 && b.r < 7, is_del = !b.exact && b.r == 7;
    m.keep = b.exact ? 0xFFFFFFFFu : is_sub ? ~(3u << b.s) : is_ins ? ~b.lm_s : ~b.lm_s2;
    m.low = (is_ins || is_del) ? b.lm_s : 0u;
    m.m_ins = is_ins ? 0xFFFFFFFFu : 0u;
    m.m_del = is_del ? 0xFFFFFFFFu : 0u;
    const uint32_t x = (uint32_t)(b.r - 3) & 3u;
    m.ins_const = is_ins ? (x << (b.s >= 2 ? b.s - 2 : 0)) : 0u;
    m.m_sub = is_sub ? 0xFFFFFFFFu : 0u;
    m.m_db = is_del ? 3u : 0u;
    m.p14_ins = (is_ins && b.s == 2) ? 0xFFFFFFFFu : 0u;
    return m;
}

// live = all-ones when the mutant has to be probed
__device__ __forceinline__ uint32_t mutate2(const LaneMasks &m, uint32_t K, uint32_t del_base, uint32_t &live) {
    const uint32_t cur = (K >> m.s) & 3u;
    const uint32_t bsub = m.r + (m.r >= cur ? 1u : 0u);
    const uint32_t lo = K & m.low;
    uint32_t v = K & m.keep;
    v |= (lo >> 2) & m.m_ins;
    v |= (lo << 2) & m.m_del;
    v |= m.ins_const;
    v |= (bsub << m.s) & m.m_sub;
    v |= del_base & m.m_db;
    live = m.valid & ~(m.p14_ins & (0u - (uint32_t)((K & 3u) != 0u)));
    return v;
}

int launch_bc_match2(smi_ctx *ctx, const smi_bc_window *d_win, size_t n, int five_prime, smi_bc_result *d_out,
                     hipStream_t s);

// Lane-constant description of the probe a lane makes in the two rounds of one offset.  The 124 sequences of an offset
// are dealt to the lanes by KIND, so that a round needs two or three VALU operations per mutant, with the shifted
// copies of the window coming from the scalar unit:
//   round A: lanes 0..47 substitutions (position lane / 3, the base XORed with 1 + lane % 3), 48..62 deletions of
//            position lane - 48, lane 63 the window itself:        mutant = bfi(keep, K, K << 2 | del_base) ^ flip
//   round B: lanes 0..59 insertions (after position lane / 4, base lane % 4), 60..63 idle:
//                                                                mutant = (bfi(keep, K, K >> 2) & clear) | put
// The reference's enumeration index of a hit (what the HashSet rule needs) is recovered afterwards, for hits only.
struct ProbeLanes {
    uint32_t keepA, flipA, rotA, rot2A, tabA;  // tab: byte offset of the lane's table inside t2; rot2: where the stage-2 bits start
    uint32_t keepB, clearB, putB, rotB, rot2B, tabB;
    uint32_t enumA;  // enumeration index e for deletion lanes; 8 * position for substitution lanes (rank added later)
    uint32_t enumB;
    int sA;          // bit offset of the substituted base (substitution lanes)
    uint32_t subA;   // 1 + lane % 3 for substitution lanes, else 0
};

__device__ __forceinline__ ProbeLanes make_probe_lanes(int lane) {
    ProbeLanes L;
    {  // round A
        const bool is_sub = lane < 48, is_del = lane >= 48 && lane < 63;
        const int p = is_sub ? lane / 3 : (is_del ? lane - 48 : 0);
        const int s = 30 - 2 * p;
        const uint32_t d = (uint32_t)(lane % 3) + 1u;
        L.keepA = is_del ? ~lowmask(s + 2) : 0xFFFFFFFFu;
        L.flipA = is_sub ? d << s : 0u;
        L.sA = s;
        L.subA = is_sub ? d : 0u;
        L.enumA = is_sub ? 8u * p : (is_del ? 8u * p + 7u : 127u);
        const bool far = make_lane(is_sub ? 8 * p : (is_del ? 8 * p + 7 : 127)).far;
        L.rotA = far ? 14u + kG0 : (uint32_t)kG0;
        L.rot2A = far ? 14u : 0u;
        L.tabA = far ? (uint32_t)(kL0Words * 8) : 0u;
    }
    {  // round B
        const bool is_ins = lane < 60;
        const int p = lane >> 2;
        const int s = 30 - 2 * p;  // >= 2 for p <= 14
        const uint32_t x = (uint32_t)lane & 3u;
        L.keepB = is_ins ? ~lowmask(s) : 0xFFFFFFFFu;
        L.clearB = is_ins ? ~(3u << (s - 2)) : 0xFFFFFFFFu;
        L.putB = is_ins ? x << (s - 2) : 0u;
        L.enumB = is_ins ? 8u * p + 3u + x : 255u;
        const bool far = is_ins && make_lane(8 * p + 3).far;
        L.rotB = far ? 14u + kG0 : (uint32_t)kG0;
        L.rot2B = far ? 14u : 0u;
        L.tabB = far ? (uint32_t)(kL0Words * 8) : 0u;
    }
    return L;
}

__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

// smallest enumeration index among the lanes of `hits` (wave-uniform; only reached for windows that have a level-1 hit)
__device__ __forceinline__ uint32_t min_over(unsigned long long hits, uint32_t e, uint32_t m) {
    while (hits) {
        const int l = __builtin_ctzll(hits);
        m = min(m, (uint32_t)__builtin_amdgcn_readlane(e, l));
        hits &= hits - 1;
    }
    return m;
}

__device__ __forceinline__ uint32_t load_at(const uint32_t *base, uint32_t byte_off) {
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(base) + byte_off);
}

// Work split inside a wavefront (64 reads per batch):
//   phase 1 (lane = read)   : every lane derives the five window keys of ITS read (N handling, reverse complement)
//   phase 2 (lane = mutant) : for r = 0..63 the keys of read r are broadcast (v_readlane) and the 64 lanes probe the
//                             124 sequences of each offset; per offset the hit masks are folded into one byte
//                             {exact hit, enumeration index + 1 of the first level-1 hit} which goes back to lane r
//   phase 3 (lane = read)   : every lane runs the HashSet-order / best-second rule for ITS read and stores 16 B
// Phase 2 is written for instruction count: a probe is a 32-bit byte offset against a uniform table base, a lane that a
// level has ruled out reads word 0 of the next (offset & mask, mask = the sign-extended bit it just extracted).
// Levels: the two-stage top level (t2: one 8-byte load gives the stage-1 word of l0 | l0s and its stage-2 word), then the
// exact level -- two dependent round trips and 20 gathers per read instead of three and 30.
template <int MAX_ED>
__global__ __launch_bounds__(256) void k_bc_match_ed1(const smi_bc_window *__restrict__ win, size_t n, int five_prime,
                                                      Pyramid P, smi_bc_result *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const ProbeLanes L = make_probe_lanes(lane);
    constexpr uint32_t kTopOffMask2 = (uint32_t)(kL0Words * 8 - 8);  // byte offset of an 8-byte entry of one table of t2
    const unsigned long long exact_lane = 1ull << 63;
    const unsigned long long lanesA = MAX_ED == 0 ? exact_lane : ~0ull;
    const unsigned long long lanesB = MAX_ED == 0 ? 0ull : (1ull << 60) - 1ull;
    const unsigned long long p14B = 0xFull << 56;  // "insert behind position 14": a barcode only when the last base is A
    const unsigned long long subsA = (1ull << 48) - 1ull;
    const bool fp = five_prime != 0;
    constexpr int OFFS[5] = {0, -1, 1, -2, 2};  // the reference's order (Parser.java:L203)

    for (size_t base = wave * 64; base < n; base += n_waves * 64) {
        smi_bc_window my;
        my.bases = 0;
        my.nmask = 0;
        my.flags = 0;
        if (base + lane < n) my = win[base + lane];
        uint32_t key[5];
        uint32_t packed = (my.flags & SMI_WIN_VALID) ? (1u << 15) : 0u;  // [1:0]..[9:8] del bases, [14:10] usable
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const OffsetKey k = make_key(my.bases, my.nmask, OFFS[q], fp);
            key[q] = k.key;
            packed |= k.del_base << (2 * q);
            packed |= (k.usable ? 1u : 0u) << (10 + q);
        }
        uint32_t sum_lo = 0, sum_hi = 0;
        const int cnt = (int)min((size_t)64, n - base);
        for (int r = 0; r < cnt; r++) {
            const uint32_t pk = __builtin_amdgcn_readlane(packed, r);
            if (!(pk & (1u << 15))) continue;  // wave-uniform
            uint32_t K[5];
            uint32_t mut[10], live[10], w[10];
#pragma unroll
            for (int q = 0; q < 5; q++) {
                K[q] = __builtin_amdgcn_readlane(key[q], r);
                const uint32_t db = (pk >> (2 * q)) & 3u;
                mut[2 * q] = bfi(L.keepA, K[q], (K[q] << 2) | db) ^ L.flipA;
                mut[2 * q + 1] = (bfi(L.keepB, K[q], K[q] >> 2) & L.clearB) | L.putB;
            }
            uint32_t w2[10];
#pragma unroll
            for (int t = 0; t < 10; t++) {
                const uint32_t rot = __builtin_amdgcn_alignbit(mut[t], mut[t], (t & 1) ? L.rotB : L.rotA);  // bit index: low 5 bits
                const uint2 e = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(P.t2) +
                                                                  (((rot >> 2) & kTopOffMask2) | ((t & 1) ? L.tabB : L.tabA)));
                w[t] = e.x;
                w2[t] = e.y;
                live[t] = rot;
            }
#pragma unroll
            for (int t = 0; t < 10; t++) {
                const uint32_t r2 = __builtin_amdgcn_alignbit(mut[t], mut[t], (t & 1) ? L.rot2B : L.rot2A);  // stage-2 bit: low 5 bits
                live[t] = (uint32_t)__builtin_amdgcn_sbfe(w[t], live[t], 1) & (uint32_t)__builtin_amdgcn_sbfe(w2[t], r2, 1);
            }
#pragma unroll
            for (int t = 0; t < 10; t++) w[t] = load_at(P.fine, (mut[t] >> 3) & (live[t] & ~3u));

            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const bool usable = (pk >> (10 + q)) & 1u;
                const unsigned long long liveA = usable ? lanesA : 0ull;
                const unsigned long long liveB = usable ? ((K[q] & 3u) ? lanesB & ~p14B : lanesB) : 0ull;
                const uint32_t fa = live[2 * q] & (uint32_t)__builtin_amdgcn_sbfe(w[2 * q], mut[2 * q], 1);
                const uint32_t fb = live[2 * q + 1] & (uint32_t)__builtin_amdgcn_sbfe(w[2 * q + 1], mut[2 * q + 1], 1);
                const unsigned long long ha = __ballot(fa != 0u) & liveA;
                const unsigned long long hb = __ballot(fb != 0u) & liveB;
                uint32_t code = (uint32_t)(ha >> 63) << 7;  // exact match (BarcodeMatchTester.java:L204-206)
                const unsigned long long ha1 = ha & ~exact_lane;
                if (ha1 | hb) {
                    // first hit in the reference's enumeration order = the only level-1 OneMatch the HashSet keeps:
                    // position-major; at one position the substitutions by ascending base (the current base skipped,
                    // L259-260), then the insertions, then the deletion
                    const uint32_t cur = (K[q] >> L.sA) & 3u;
                    const uint32_t b = cur ^ L.subA;
                    const uint32_t eA = L.enumA + (((ha1 & subsA) >> lane) & 1ull ? b - (b > cur ? 1u : 0u) : 0u);
                    code |= min_over(hb, L.enumB, min_over(ha1, eA, 255u)) + 1u;
                }
                if (q < 4)
                    lo |= code << (8 * q);
                else
                    hi = code;
            }
            const bool mine = lane == r;
            sum_lo = mine ? lo : sum_lo;
            sum_hi = mine ? hi : sum_hi;
        }
        smi_bc_result res;
        uint32_t c_bc[10], c_rs[10];
        int c_imd[10];
        uint32_t present = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const uint32_t code = q < 4 ? (sum_lo >> (8 * q)) & 0xFFu : sum_hi & 0xFFu;
            const uint32_t db = (packed >> (2 * q)) & 3u;
            c_rs[2 * q] = c_rs[2 * q + 1] = key[q];
            c_bc[2 * q] = key[q];
            c_imd[2 * q] = 0;
            present |= (code >> 7) << (2 * q);
            const int e = (int)(code & 0x7Fu) - 1;
            bool dummy;
            c_bc[2 * q + 1] = mutate(make_lane(e < 0 ? 0 : e), key[q], db, dummy);
            c_imd[2 * q + 1] = ins_minus_del_of(e < 0 ? 0 : e);
            present |= (e >= 0 ? 1u : 0u) << (2 * q + 1);
        }
        pick_best(c_bc, c_rs, c_imd, present, MAX_ED, res);
        if (!(packed & (1u << 15))) {
            res.found = -1;
            res.n_matches = 0;
        }
        if (base + lane < n) out[base + lane] = res;
    }
}

// K-BC1 behind the offset filter (P.nb).  Phase 1 also reads, per lane, the neighbourhood bit of its five windows; phase 2 then runs
// per OFFSET: the reads of the batch whose window at that offset has a barcode in reach are taken four at a time (eight probe rounds in
// flight), all others cost nothing.  Against the 3.6 M whitelist the neighbourhood holds 14 % of all 16-mers, so a read keeps
// 1.6 of its 5 offsets on average (its true one and a chance one now and then); against a used list only the true one.
__global__ __launch_bounds__(256) void k_bc_match_ed1f(const smi_bc_window *__restrict__ win, size_t n, int five_prime,
                                                       Pyramid P, smi_bc_result *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const ProbeLanes L = make_probe_lanes(lane);
    constexpr uint32_t kTopOffMask2 = (uint32_t)(kL0Words * 8 - 8);
    const unsigned long long exact_lane = 1ull << 63;
    const unsigned long long lanesA = ~0ull;
    const unsigned long long lanesB = (1ull << 60) - 1ull;
    const unsigned long long p14B = 0xFull << 56;
    const unsigned long long subsA = (1ull << 48) - 1ull;
    const bool fp = five_prime != 0;
    constexpr int OFFS[5] = {0, -1, 1, -2, 2};
    constexpr int kSlots = 4;  // (read, offset) pairs probed together: 3 / 4 / 5 / 8 measured 2.71 / 2.68 / 2.71 / 2.81 ms per 10 M reads

    for (size_t base = wave * 64; base < n; base += n_waves * 64) {
        smi_bc_window my;
        my.bases = 0;
        my.nmask = 0;
        my.flags = 0;
        if (base + lane < n) my = win[base + lane];
        uint32_t key[5];
        uint32_t packed = (my.flags & SMI_WIN_VALID) ? (1u << 15) : 0u;  // [1:0]..[9:8] del bases, [14:10] usable
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const OffsetKey k = make_key(my.bases, my.nmask, OFFS[q], fp);
            key[q] = k.key;
            packed |= k.del_base << (2 * q);
            packed |= (k.usable ? 1u : 0u) << (10 + q);
        }
        // the neighbourhood bit of every usable window of this lane's read: five independent loads
        uint32_t nbw[5];
#pragma unroll
        for (int q = 0; q < 5; q++) nbw[q] = P.nb[key[q] >> 5];
        uint32_t near = 0;
#pragma unroll
        for (int q = 0; q < 5; q++)
            near |= ((packed >> 15) & (packed >> (10 + q)) & (nbw[q] >> (key[q] & 31u)) & 1u) << q;

        uint32_t sum_lo = 0, sum_hi = 0;
#pragma unroll 1
        for (int q = 0; q < 5; q++) {
            unsigned long long todo = __ballot((near >> q) & 1u);
            if (!todo) continue;
            const uint32_t keyq = q == 0 ? key[0] : q == 1 ? key[1] : q == 2 ? key[2] : q == 3 ? key[3] : key[4];
            const uint32_t dbq = (packed >> (2 * q)) & 3u;
            while (todo) {
                int rr[kSlots];
                bool on[kSlots];
#pragma unroll
                for (int t = 0; t < kSlots; t++) {
                    on[t] = todo != 0ull;
                    rr[t] = on[t] ? __builtin_ctzll(todo) : 0;
                    if (on[t]) todo &= todo - 1ull;
                }
                uint32_t K[kSlots], mut[2 * kSlots], live[2 * kSlots], w[2 * kSlots], w2[2 * kSlots];
#pragma unroll
                for (int t = 0; t < kSlots; t++) {
                    K[t] = __builtin_amdgcn_readlane(keyq, rr[t]);
                    const uint32_t db = __builtin_amdgcn_readlane(dbq, rr[t]);
                    mut[2 * t] = bfi(L.keepA, K[t], (K[t] << 2) | db) ^ L.flipA;
                    mut[2 * t + 1] = (bfi(L.keepB, K[t], K[t] >> 2) & L.clearB) | L.putB;
                }
#pragma unroll
                for (int t = 0; t < 2 * kSlots; t++) {
                    const uint32_t rot = __builtin_amdgcn_alignbit(mut[t], mut[t], (t & 1) ? L.rotB : L.rotA);
                    const uint2 e = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(P.t2) +
                                                                      (((rot >> 2) & kTopOffMask2) | ((t & 1) ? L.tabB : L.tabA)));
                    w[t] = e.x;
                    w2[t] = e.y;
                    live[t] = rot;
                }
#pragma unroll
                for (int t = 0; t < 2 * kSlots; t++) {
                    const uint32_t r2 = __builtin_amdgcn_alignbit(mut[t], mut[t], (t & 1) ? L.rot2B : L.rot2A);
                    live[t] = (uint32_t)__builtin_amdgcn_sbfe(w[t], live[t], 1) & (uint32_t)__builtin_amdgcn_sbfe(w2[t], r2, 1);
                }
#pragma unroll
                for (int t = 0; t < 2 * kSlots; t++) w[t] = load_at(P.fine, (mut[t] >> 3) & (live[t] & ~3u));
#pragma unroll
                for (int t = 0; t < kSlots; t++) {
                    const unsigned long long liveA = on[t] ? lanesA : 0ull;
                    const unsigned long long liveB = on[t] ? ((K[t] & 3u) ? lanesB & ~p14B : lanesB) : 0ull;
                    const uint32_t fa = live[2 * t] & (uint32_t)__builtin_amdgcn_sbfe(w[2 * t], mut[2 * t], 1);
                    const uint32_t fb = live[2 * t + 1] & (uint32_t)__builtin_amdgcn_sbfe(w[2 * t + 1], mut[2 * t + 1], 1);
                    const unsigned long long ha = __ballot(fa != 0u) & liveA;
                    const unsigned long long hb = __ballot(fb != 0u) & liveB;
                    uint32_t code = (uint32_t)(ha >> 63) << 7;  // exact match (BarcodeMatchTester.java:L204-206)
                    const unsigned long long ha1 = ha & ~exact_lane;
                    if (ha1 | hb) {  // first hit in the reference's enumeration order (see k_bc_match_ed1)
                        const uint32_t cur = (K[t] >> L.sA) & 3u;
                        const uint32_t b = cur ^ L.subA;
                        const uint32_t eA = L.enumA + (((ha1 & subsA) >> lane) & 1ull ? b - (b > cur ? 1u : 0u) : 0u);
                        code |= min_over(hb, L.enumB, min_over(ha1, eA, 255u)) + 1u;
                    }
                    const bool mine = on[t] && lane == rr[t];
                    if (q < 4)
                        sum_lo |= mine ? code << (8 * q) : 0u;
                    else
                        sum_hi = mine ? code : sum_hi;
                }
            }
        }
        smi_bc_result res;
        uint32_t c_bc[10], c_rs[10];
        int c_imd[10];
        uint32_t present = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const uint32_t code = q < 4 ? (sum_lo >> (8 * q)) & 0xFFu : sum_hi & 0xFFu;
            const uint32_t db = (packed >> (2 * q)) & 3u;
            c_rs[2 * q] = c_rs[2 * q + 1] = key[q];
            c_bc[2 * q] = key[q];
            c_imd[2 * q] = 0;
            present |= (code >> 7) << (2 * q);
            const int e = (int)(code & 0x7Fu) - 1;
            bool dummy;
            c_bc[2 * q + 1] = mutate(make_lane(e < 0 ? 0 : e), key[q], db, dummy);
            c_imd[2 * q + 1] = ins_minus_del_of(e < 0 ? 0 : e);
            present |= (e >= 0 ? 1u : 0u) << (2 * q + 1);
        }
        pick_best(c_bc, c_rs, c_imd, present, 1, res);
        if (!(packed & (1u << 15))) {
            res.found = -1;
            res.n_matches = 0;
        }
        if (base + lane < n) out[base + lane] = res;
    }
}

// K-BC1 from the neighbourhood table (P.nt), two kernels.
// k_bc_codes_ed1t: one lane = one (read, offset) pair, so that every look-up of the batch is in flight at once (with a lane per read the
// five windows of a read were looked up one behind the other and a wave waited for its slowest lane: 2.5 ms, 1.5 of them waiting).  The
// window's filter bit (P.nb) first; a window whose bit is set reads the 64-byte bucket of its sequence and every entry with that sequence
// names one mutation of the reference's enumeration that turns the window into a barcode: the exact match (kind 0), or the enumeration
// index 8 * position + kind slot of a level-1 match -- valid unless it is a deletion whose appended base is not the read's next base, or
// the insertion behind position 14 of a window that does not end in A (the Java shift wrap, see `mutate`).  The smallest valid index is
// the match the reference's HashSet keeps.  -> one byte per pair: exact << 7 | (index + 1).  No mutant is generated, no membership probed.
// k_bc_pick_ed1t: one lane = one read: the HashSet-order / best-second rule over its five bytes.
__global__ __launch_bounds__(256) void k_bc_codes_ed1t(const smi_bc_window *__restrict__ win, size_t n, int five_prime, Pyramid P,
                                                       uint8_t *__restrict__ codes) {
    const bool fp = five_prime != 0;
    const size_t total = n * 5;
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = j / 5;
        const int q = (int)(j - 5 * i);
        const int off = q == 0 ? 0 : (q == 1 ? -1 : (q == 2 ? 1 : (q == 3 ? -2 : 2)));  // the reference's order (Parser.java:L203)
        // the window as ONE 16-byte load (as `win[i]` the flags word was fetched and tested first, the bases behind that wait: a fourth
        // dependent round trip in front of the filter bit, the bucket and the store)
        smi_bc_window my;
        {
            const uint4 raw = *reinterpret_cast<const uint4 *>(win + i);
            my.bases = ((uint64_t)raw.y << 32) | raw.x;
            my.nmask = raw.z;
            my.flags = raw.w;
        }
        uint32_t code = 0;
        {
            // key and filter word for every lane, valid window or not (any 32-bit key lies inside the 512 MiB bitmap): nothing but the window's
            // own load stands in front of the probe
            const OffsetKey k = make_key(my.bases, my.nmask, off, fp);
            const uint32_t K = k.key;
            uint32_t word, wbit = K & 31u;
            if (P.nb5) {  // (wave-uniform) the five lanes of a read ask for neighbouring words: three sectors per read instead of five
                uint32_t wi;
                wbit = nb5_bit_index(K, fp ? -off : off, wi);
                word = P.nb5[wi];
            } else
                word = P.nb[K >> 5];
            if ((my.flags & SMI_WIN_VALID) && k.usable && ((word >> wbit) & 1u)) {
                uint32_t best = 255u;
                uint32_t idx = nt_slot(K, P.nt_cap);
                for (;;) {
                    uint64_t e[8];
                    __builtin_memcpy(e, P.nt + idx, 64);  // one bucket = one line
                    bool open = false;
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        open = open || e[t] == 0ull;
                        if ((uint32_t)(e[t] >> 8) == K && (e[t] >> 40)) {
                            const uint32_t kind = (uint32_t)e[t] & 3u, pos = ((uint32_t)e[t] >> 2) & 15u, base = ((uint32_t)e[t] >> 6) & 3u;
                            if (kind == 0u) {
                                code |= 0x80u;  // exact match (BarcodeMatchTester.java:L204-206)
                            } else if (kind == 1u) {
                                const uint32_t cur = (K >> (30 - 2 * pos)) & 3u;
                                best = min(best, 8u * pos + base - (base > cur ? 1u : 0u));  // substitutions by ascending base, the current one skipped
                            } else if (kind == 2u) {
                                if (!(pos == 14u && (K & 3u) != 0u)) best = min(best, 8u * pos + 3u + base);
                            } else {
                                if (base == k.del_base) best = min(best, 8u * pos + 7u);
                            }
                        }
                    }
                    if (open) break;  // a bucket with a free slot has never overflowed into the next one
                    idx = idx + 8 == P.nt_cap ? 0u : idx + 8;
                }
                if (best != 255u) code |= best + 1u;
            }
        }
        codes[j] = (uint8_t)code;
    }
}

__global__ __launch_bounds__(256) void k_bc_pick_ed1t(const smi_bc_window *__restrict__ win, size_t n, int five_prime,
                                                      const uint8_t *__restrict__ codes, smi_bc_result *__restrict__ out) {
    const bool fp = five_prime != 0;
    constexpr int OFFS[5] = {0, -1, 1, -2, 2};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const smi_bc_window my = win[i];
        const bool valid = my.flags & SMI_WIN_VALID;
        smi_bc_result res;
        uint32_t c_bc[10], c_rs[10];
        int c_imd[10];
        uint32_t present = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const OffsetKey k = make_key(my.bases, my.nmask, OFFS[q], fp);
            const uint32_t code = codes[5 * i + q];
            c_rs[2 * q] = c_rs[2 * q + 1] = k.key;
            c_bc[2 * q] = k.key;
            c_imd[2 * q] = 0;
            present |= (code >> 7) << (2 * q);
            const int e = (int)(code & 0x7Fu) - 1;
            bool dummy;
            c_bc[2 * q + 1] = mutate(make_lane(e < 0 ? 0 : e), k.key, k.del_base, dummy);
            c_imd[2 * q + 1] = ins_minus_del_of(e < 0 ? 0 : e);
            present |= (e >= 0 ? 1u : 0u) << (2 * q + 1);
        }
        pick_best(c_bc, c_rs, c_imd, present, 1, res);
        if (!valid) {
            res.found = -1;
            res.n_matches = 0;
        }
        out[i] = res;
    }
}

int launch_bc_match(smi_ctx *ctx, const smi_bc_window *d_win, size_t n, int max_ed, int five_prime,
                    smi_bc_result *d_out, hipStream_t s) {
    if (!n) return SMI_OK;
    if (max_ed == 2) return launch_bc_match2(ctx, d_win, n, five_prime, d_out, s);
    Pyramid P = pyramid_of(ctx);
    const size_t n_waves = (n + 63) / 64;
    const unsigned grid = (unsigned)((n_waves + 3) / 4);  // one batch of 64 reads per wave: measured 3 % faster than a capped grid
    if (max_ed != 0 && P.nt && ctx->bc_codes_bytes < 5 * n) {  // one byte per (read, offset) between the two kernels of the table path
        SMI_HIP(hipStreamSynchronize(s));
        if (ctx->bc_codes) SMI_HIP(hipFree(ctx->bc_codes));
        ctx->bc_codes = nullptr;
        ctx->bc_codes_bytes = 0;
        SMI_HIP(hipMalloc((void **)&ctx->bc_codes, 5 * n + 5 * n / 4));
        ctx->bc_codes_bytes = 5 * n + 5 * n / 4;
    }
    if (int rc = time_begin(ctx, SMI_K_BC_MATCH, s)) return rc;
    if (max_ed == 0)
        hipLaunchKernelGGL(k_bc_match_ed1<0>, dim3(grid), dim3(256), 0, s, d_win, n, five_prime, P, d_out);
    else if (P.nt) {
        hipLaunchKernelGGL(k_bc_codes_ed1t, dim3((unsigned)std::min<size_t>((5 * n + 255) / 256, 256 * 256)), dim3(256), 0, s, d_win, n, five_prime, P,
                           ctx->bc_codes);
        hipLaunchKernelGGL(k_bc_pick_ed1t, dim3((unsigned)std::min<size_t>((n + 255) / 256, 256 * 64)), dim3(256), 0, s, d_win, n, five_prime,
                           ctx->bc_codes, d_out);
    }
    else if (P.nb)
        hipLaunchKernelGGL(k_bc_match_ed1f, dim3(grid), dim3(256), 0, s, d_win, n, five_prime, P, d_out);
    else
        hipLaunchKernelGGL(k_bc_match_ed1<1>, dim3(grid), dim3(256), 0, s, d_win, n, five_prime, P, d_out);
    SMI_HIP(hipGetLastError());
    if (int rc = time_end(ctx, SMI_K_BC_MATCH, s)) return rc;
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K-HIST: pass-1 exact membership + histogram (UsedCellBCListGenerator.java:L207-229).  One thread per read:
// coalesced key/flag loads, one pyramid walk, one atomic per accepted read.  hist is indexed by the key's rank
// in ascending key order (rank[] + popcount inside the 256-key block), so the vector summed across GPUs is dense.
// ---------------------------------------------------------------------------------------------------------
__global__ void k_hist(const uint32_t *__restrict__ keys, const uint8_t *__restrict__ pass, size_t n, Pyramid P,
                       uint32_t *__restrict__ hist) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (pass && !pass[i]) continue;
        const uint32_t k = keys[i];
        if (!bit_of(P.l0, k >> kG0)) continue;
        if (!((P.l1[l1_word(k)] >> l1_bit(k)) & 1u)) continue;
        const uint32_t blk = k >> 8;
        const uint32_t *w = P.fine + (size_t)blk * 8;
        const uint32_t wi = (k >> 5) & 7u;
        const uint32_t word = w[wi];
        if (!((word >> (k & 31)) & 1u)) continue;
        uint32_t ord = P.rank[blk] + __popc(word & ((1u << (k & 31)) - 1u));
        for (uint32_t j = 0; j < wi; j++) ord += __popc(w[j]);
        atomicAdd(&hist[ord], 1u);
    }
}

// K-CNT: assignedBarcodes2ndPass[bc].addCountForEd(ed) (Parser.java:L305-311, Parser$BarcodeCounts L339-354) for a batch of
// pass-2 results: counts[3 * ordinal(bc) + ed] += 1 for every assigned read.  The assigned barcode is a member of the loaded
// set, so its ordinal comes straight from rank[] + the popcount inside its 256-key block; the vector is dense and sums across
// GPUs like the pass-1 histogram.
__global__ void k_bc_counts(const smi_bc_result *__restrict__ res, size_t n, Pyramid P, uint32_t *__restrict__ counts) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const smi_bc_result r = res[i];
        if (r.found != 1 || r.ed < 0 || r.ed > 2) continue;
        const uint32_t k = r.bc;
        const uint32_t blk = k >> 8;
        const uint32_t *w = P.fine + (size_t)blk * 8;
        const uint32_t wi = (k >> 5) & 7u;
        const uint32_t word = w[wi];
        if (!((word >> (k & 31)) & 1u)) continue;  // cannot happen for a result of this context's matcher
        uint32_t ord = P.rank[blk] + __popc(word & ((1u << (k & 31)) - 1u));
        for (uint32_t j = 0; j < wi; j++) ord += __popc(w[j]);
        atomicAdd(&counts[3 * (size_t)ord + (uint32_t)r.ed], 1u);
    }
}

int launch_bc_counts(smi_ctx *ctx, const smi_bc_result *d_res, size_t n, uint32_t *d_counts, hipStream_t s) {
    if (!n) return SMI_OK;
    Pyramid P = pyramid_of(ctx);
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_bc_counts, dim3(grid), dim3(256), 0, s, d_res, n, P, d_counts);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int launch_hist(smi_ctx *ctx, const uint32_t *d_keys, const uint8_t *d_pass, size_t n, uint32_t *d_hist,
                hipStream_t s) {
    if (!n) return SMI_OK;
    Pyramid P = pyramid_of(ctx);
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
    if (int rc = time_begin(ctx, SMI_K_HIST, s)) return rc;
    hipLaunchKernelGGL(k_hist, dim3(grid), dim3(256), 0, s, d_keys, d_pass, n, P, d_hist);
    SMI_HIP(hipGetLastError());
    if (int rc = time_end(ctx, SMI_K_HIST, s)) return rc;
    return SMI_OK;
}

// pass-1 histogram from scan output: offset-0 key of the window (same make_key as the matcher: N emulation included)
__global__ void k_hist_windows(const smi_bc_window *__restrict__ win, const smi_scan_result *__restrict__ scan, size_t n,
                               Pyramid P, uint32_t *__restrict__ hist) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (!scan[i].pass1_ok) continue;
        const smi_bc_window w = win[i];
        if (!(w.flags & SMI_WIN_VALID)) continue;
        // 3': reverse complement of stranded[AE-16 .. AE-1]; 5': stranded[AE+1 .. AE+16] (UsedCellBCListGenerator.java:L210-221)
        const OffsetKey ok = make_key(w.bases, w.nmask, 0, (w.flags & SMI_WIN_5P) != 0);
        if (!ok.usable) continue;  // an N poisons the 5' key: never a whitelist member
        const uint32_t k = ok.key;
        if (!bit_of(P.l0, k >> kG0)) continue;
        if (!((P.l1[l1_word(k)] >> l1_bit(k)) & 1u)) continue;
        const uint32_t blk = k >> 8;
        const uint32_t *wd = P.fine + (size_t)blk * 8;
        const uint32_t wi = (k >> 5) & 7u;
        const uint32_t word = wd[wi];
        if (!((word >> (k & 31)) & 1u)) continue;
        uint32_t ord = P.rank[blk] + __popc(word & ((1u << (k & 31)) - 1u));
        for (uint32_t j = 0; j < wi; j++) ord += __popc(wd[j]);
        atomicAdd(&hist[ord], 1u);
    }
}

int launch_hist_windows(smi_ctx *ctx, const smi_bc_window *d_win, const smi_scan_result *d_scan, size_t n,
                        uint32_t *d_hist, hipStream_t s) {
    if (!n) return SMI_OK;
    Pyramid P = pyramid_of(ctx);
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_hist_windows, dim3(grid), dim3(256), 0, s, d_win, d_scan, n, P, d_hist);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

// pass 1 without a list of possible barcodes (`-a none`, generateUsedBarcodesWithoutWhitelist: the membership predicate is "true",
// UsedCellBCListGenerator.java:L255-256): the barcode of every read that passes the quality filter is counted, so the chunk leaves its KEYS
// (appended to a list; sorted and counted at the end of the pass, launch_count_keys) instead of increments of a dense histogram.
// The key is the reference's long: a 3' barcode goes through reverseComplement, which keeps the low 32 bits (an N comes out as "T .. T C",
// make_key); a 5' barcode does not, so an N leaves its -2 in the long: bits 63 .. 32 all set, below them the same "T .. T C" pattern
// (tests/golden/ref_exec_pass1_nowl_5p.json has five of them).
__global__ void k_keys_windows(const smi_bc_window *__restrict__ win, const smi_scan_result *__restrict__ scan, size_t n,
                               uint64_t *__restrict__ keys, unsigned long long cap, unsigned long long *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // (every lane of a wave takes part in every turn: the append is one atomic per wave, not one per read -- atomics on a single address serialise)
    for (size_t i0 = blockIdx.x * (size_t)blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride) {
        const size_t i = i0 + lane;
        bool have = false;
        uint64_t k = 0;
        if (i < n && scan[i].pass1_ok) {
            const smi_bc_window w = win[i];
            if (w.flags & SMI_WIN_VALID) {
                const bool fp = (w.flags & SMI_WIN_5P) != 0;
                const OffsetKey ok = make_key(w.bases, w.nmask, 0, fp);
                k = ok.key;
                if (!ok.usable) {  // 5' window with an N (make_key: base indices 2 .. 17 of the 25-base record)
                    const uint32_t nm = (w.nmask >> 2) & 0xFFFFu;
                    const int last = 31 - __clz(nm);  // window index of the last N
                    const uint32_t low = (ok.key & lowmask(30 - 2 * last)) | (0xFFFFFFFFu << (31 - 2 * last));
                    k = 0xFFFFFFFF00000000ull | low;
                }
                have = true;
            }
        }
        const unsigned long long m = __ballot(have);
        if (!m) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(count, (unsigned long long)__popcll(m));  // (the order of the list is of no consequence: it is sorted before it is counted)
        base = __shfl(base, 0);
        const unsigned long long at = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
        if (have && at < cap) keys[at] = k;
    }
}

int launch_keys_windows(smi_ctx *, const smi_bc_window *d_win, const smi_scan_result *d_scan, size_t n, uint64_t *d_keys, size_t cap,
                        unsigned long long *d_count, hipStream_t s) {
    if (!n) return SMI_OK;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_keys_windows, dim3(grid), dim3(256), 0, s, d_win, d_scan, n, d_keys, (unsigned long long)cap, d_count);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

// the key list of a whole pass -> its distinct keys (ascending) and their counts: radix sort + run-length encoding (hipcub), once per run
int launch_count_keys(smi_ctx *, const uint64_t *d_keys, size_t n, uint64_t *d_unique, uint32_t *d_counts, uint64_t *d_n_unique, hipStream_t s) {
    if (!n) {
        SMI_HIP(hipMemsetAsync(d_n_unique, 0, 8, s));
        return SMI_OK;
    }
    if (n > 0x7FFFFFFFull) {
        set_error("smi_count_keys_device: more than 2^31 - 1 keys in one call");
        return SMI_ERR_INVALID;
    }
    size_t t_sort = 0, t_rle = 0;
    uint64_t *d_sorted = nullptr;
    int *d_runs = nullptr;
    void *d_tmp = nullptr;
    SMI_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, t_sort, d_keys, d_sorted, (int)n, 0, 64, s));
    SMI_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, t_rle, d_sorted, d_unique, d_counts, d_runs, (int)n, s));
    const size_t tmp_bytes = std::max(t_sort, t_rle);
    SMI_HIP(hipMalloc((void **)&d_sorted, n * 8));
    hipError_t e = hipMalloc(&d_tmp, tmp_bytes + 16);
    if (e == hipSuccess) e = hipMalloc((void **)&d_runs, 16);
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortKeys(d_tmp, t_sort, d_keys, d_sorted, (int)n, 0, 64, s);
    if (e == hipSuccess) e = hipcub::DeviceRunLengthEncode::Encode(d_tmp, t_rle, d_sorted, d_unique, d_counts, d_runs, (int)n, s);
    int h_runs = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_runs, d_runs, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    const uint64_t runs64 = (uint64_t)h_runs;
    if (e == hipSuccess) e = hipMemcpyAsync(d_n_unique, &runs64, 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_sorted);
    (void)hipFree(d_tmp);
    (void)hipFree(d_runs);
    if (e != hipSuccess) return hip_fail(e, "smi_count_keys_device");
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K-BC2: ed <= 2 matcher.  Same contract as K-BC1, but the second mutation level follows the reference's
// depth-first order and its dedup set (NucTwoBitPerBaseEDtesterBase.java:L82-95,L105-120, BarcodeMatchTester.java:
// L218-241):
//   * level-1 children c_e (the 123 mutants of K-BC1's enumeration) are CREATED unless (int)c_e is already in
//     `tested`, i.e. equals the root (for positions >= 1) or an earlier-position child: created(e) <=>
//     p_e == min position among children with the same low 32 bits (and not the root unless p_e == 0);
//   * created children are expanded LIFO: by root position ascending, inside a position in reverse creation order;
//     an item visits positions 0..15 except the one it was created at;
//   * a level-2 mutant m is skipped when (int)m is the root, an item expanded earlier, or the item's own sequence
//     after its first visited position; otherwise it is probed, and the first hit in this order is THE level-2 match.
// A sequence is (low 32 bits, g) where g = bits 62..63 left behind by "insert after position 14" (Java shift wrap);
// g != 0 can never equal a barcode but still takes part in the (int) dedup.
// One wavefront per read; lanes are the 128 (position, kind) child slots of the item being expanded, so the DFS
// order inside an item is again ballot + ctz, and items are walked in order with early exit on the first hit.
// The dedup set is a 256-slot open-addressing table in LDS (per wave), value = expansion order of the sequence.
// ---------------------------------------------------------------------------------------------------------
struct Seq {
    uint32_t low, g;
};

// child of s at position q, kind r (0..2 substitutions, 3..6 insertions A,G,C,T, 7 deletion appending del_base)
__device__ __forceinline__ Seq child_of(Seq s, int q, int r, uint32_t del_base) {
    const int sh = 30 - 2 * q;
    const uint32_t lm = lowmask(sh), lm2 = lowmask(sh + 2);
    Seq o = s;
    if (r < 3) {
        const uint32_t cur = (s.low >> sh) & 3u;
        const uint32_t j = (uint32_t)r, b = j + (j >= cur ? 1u : 0u);
        o.low = s.low ^ ((cur ^ b) << sh);
    } else if (r < 7) {
        const uint32_t x = (uint32_t)(r - 3);
        const int xs = sh >= 2 ? sh - 2 : 0;
        o.low = (s.low & ~lm) | ((s.low & lm) >> 2) | (x << xs);
        if (q == 14) o.g = s.g | (s.low & 3u);  // getLongHashInsertByteDeg L303-305: shift count 64 wraps to 0
    } else {
        o.low = (s.low & ~lm2) | ((s.low & lm) << 2) | del_base;
    }
    return o;
}

__device__ __forceinline__ bool slot_valid(int q, int r) { return q < 15 || r < 3; }

// post[k] (1-based, k = 1..5) as the 2-bit code appended by a deletion (BarcodeMatchTester.java:L329; N -> A)
__device__ __forceinline__ uint32_t post_base(uint64_t bases, uint32_t nmask, int o, bool five_prime, int k) {
    if (!five_prime) {
        const int j = 6 + o - (k - 1);  // revcomp(substring(bcStart-5, bcStart)): Parser.java:L218
        const uint32_t b = (uint32_t)(bases >> (2 * (24 - 1 - j))) & 3u;
        return ((nmask >> j) & 1u) ? 0u : (3u - b);
    }
    const int j = 17 + o + k;  // substring(bcEnd, bcEnd + 5): Parser.java:L219
    const uint32_t b = (uint32_t)(bases >> (2 * (25 - 1 - j))) & 3u;
    return ((nmask >> j) & 1u) ? 0u : b;
}

__device__ __forceinline__ bool member(const Pyramid &P, uint32_t k) {
    if (!bit_of(P.l0, k >> kG0)) return false;
    if (!((P.l1[l1_word(k)] >> l1_bit(k)) & 1u)) return false;
    return bit_of(P.fine, k);
}

constexpr int kTabSlots = 256;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;

// 15-candidate version of pick_best: slot = 3*q + level (insertion order of offsets; order inside an offset is
// irrelevant because compareTo separates different levels)
__device__ __forceinline__ void pick_best15(const uint32_t (&bc)[15], const uint32_t (&rs)[5], const int (&imd)[15],
                                            uint32_t present, int max_ed, smi_bc_result &res) {
    constexpr int OFFS[5] = {0, -1, 1, -2, 2};
    const int n = __popc(present);
    res.bc = 0;
    res.ed_sec = 2147483647;
    res.found = 0;
    res.ed = 0;
    res.offset = 0;
    res.ins_minus_del = 0;
    res.n_matches = (uint32_t)n;
    if (n == 0) return;
    uint32_t h[5];
#pragma unroll
    for (int q = 0; q < 5; q++) h[q] = rs[q] ^ (rs[q] >> 16);
    int cap = 16;
    if (n >= 9) {  // HashMap growth: size > 0.75 cap, or a 9th node in one bin while cap < 64 (treeifyBin -> resize)
        int size = 0;
        for (int i = 0; i < 15; i++) {
            if (!((present >> i) & 1u)) continue;
            int in_bin = 0;
            for (int k = 0; k < i; k++)
                in_bin += (((present >> k) & 1u) && ((h[k / 3] ^ h[i / 3]) & (uint32_t)(cap - 1)) == 0) ? 1 : 0;
            size++;
            if (in_bin >= 8) {
                if (cap < 64) {
                    cap <<= 1;
                } else {
                    // a real tree bin: its iteration order depends on System.identityHashCode (HashMap.tieBreakOrder),
                    // i.e. the reference itself is not reproducible here; flagged instead of guessed
                    res.found = -2;
                    return;
                }
            }
            if (size > (cap * 3) / 4) cap <<= 1;
        }
    }
    uint32_t best_key = 0xFFFFFFFFu, second_key = 0xFFFFFFFFu;
    uint32_t key[15];
#pragma unroll
    for (int i = 0; i < 15; i++) {
        const int q = i / 3, lvl = i % 3;
        key[i] = ((uint32_t)lvl << 20) | (q != 0 ? (1u << 16) : 0u) | ((h[q] & (uint32_t)(cap - 1)) << 8) | (uint32_t)i;
        if (!((present >> i) & 1u)) key[i] = 0xFFFFFFFFu;
        best_key = min(best_key, key[i]);
    }
    uint32_t best_bc = 0;
    int best_imd = 0;
#pragma unroll
    for (int i = 0; i < 15; i++)
        if (key[i] == best_key) {
            best_bc = bc[i];
            best_imd = imd[i];
        }
#pragma unroll
    for (int i = 0; i < 15; i++)
        if (key[i] != 0xFFFFFFFFu && bc[i] != best_bc) second_key = min(second_key, key[i]);
    const int best_ed = (int)(best_key >> 20), second_ed = (int)(second_key >> 20);
    const bool has_second = second_key != 0xFFFFFFFFu;
    if (best_ed > max_ed) return;
    if (has_second && best_ed >= second_ed) return;
    res.found = 1;
    res.bc = best_bc;
    res.ed = (int8_t)best_ed;
    res.ed_sec = has_second ? second_ed : 2147483647;
    res.offset = (int8_t)OFFS[(best_key & 0xFF) / 3];
    res.ins_minus_del = (int8_t)best_imd;
}

// pick_best15 with its fifteen candidates read where they lie (the wave's LDS) instead of held in 35 registers by every lane: three short passes
// (least key; the barcode and ins - del under it; least key among the other barcodes).  Same rules, same order.
__device__ __forceinline__ void pick_best15_mem(const uint32_t *bc, const uint32_t *rs, const int *imd, uint32_t present, int max_ed, smi_bc_result &res) {
    constexpr int OFFS[5] = {0, -1, 1, -2, 2};
    const int n = __popc(present);
    res.bc = 0;
    res.ed_sec = 2147483647;
    res.found = 0;
    res.ed = 0;
    res.offset = 0;
    res.ins_minus_del = 0;
    res.n_matches = (uint32_t)n;
    if (n == 0) return;
    auto hq = [&](int q) -> uint32_t {
        const uint32_t r = rs[q];
        return r ^ (r >> 16);
    };
    int cap = 16;
    if (n >= 9) {  // HashMap growth: size > 0.75 cap, or a 9th node in one bin while cap < 64 (treeifyBin -> resize)
        int size = 0;
        for (int i = 0; i < 15; i++) {
            if (!((present >> i) & 1u)) continue;
            int in_bin = 0;
            for (int k = 0; k < i; k++)
                in_bin += (((present >> k) & 1u) && ((hq(k / 3) ^ hq(i / 3)) & (uint32_t)(cap - 1)) == 0) ? 1 : 0;
            size++;
            if (in_bin >= 8) {
                if (cap < 64) {
                    cap <<= 1;
                } else {
                    res.found = -2;  // a real tree bin: the reference itself is not reproducible here (pick_best15)
                    return;
                }
            }
            if (size > (cap * 3) / 4) cap <<= 1;
        }
    }
    auto key_of = [&](int i) -> uint32_t {  // (i: a set bit of `present`)
        const int q = i / 3, lvl = i % 3;
        return ((uint32_t)lvl << 20) | (q != 0 ? (1u << 16) : 0u) | ((hq(q) & (uint32_t)(cap - 1)) << 8) | (uint32_t)i;
    };
    // the matches are few (one to three as a rule): both passes walk the set bits of `present`, not the fifteen slots
    uint32_t best_key = 0xFFFFFFFFu;
#pragma unroll 1
    for (uint32_t m = present; m; m &= m - 1) best_key = min(best_key, key_of(__builtin_ctz(m)));
    const int bi = (int)(best_key & 0xFFu);  // (a key carries its index; n > 0, so there is one)
    const uint32_t best_bc = bc[bi];
    const int best_imd = imd[bi];
    uint32_t second_key = 0xFFFFFFFFu;
#pragma unroll 1
    for (uint32_t m = present; m; m &= m - 1) {
        const int i = __builtin_ctz(m);
        if (bc[i] != best_bc) second_key = min(second_key, key_of(i));
    }
    const int best_ed = (int)(best_key >> 20), second_ed = (int)(second_key >> 20);
    const bool has_second = second_key != 0xFFFFFFFFu;
    if (best_ed > max_ed) return;
    if (has_second && best_ed >= second_ed) return;
    res.found = 1;
    res.bc = best_bc;
    res.ed = (int8_t)best_ed;
    res.ed_sec = has_second ? second_ed : 2147483647;
    res.offset = (int8_t)OFFS[(best_key & 0xFF) / 3];
    res.ins_minus_del = (int8_t)best_imd;
}

// kTwoStage: dense barcode sets (the whole whitelist): both stages of the top level from t2.  kFilter: short used lists: only the items
// that P.n1 lets through are expanded (k_set_n1).
// kTable: level 2 comes from the neighbourhood table (P.nt) -- the kernel then carries neither the enumerating loop nor its lane masks, the final pick
// reads the fifteen answers from LDS (pick_best15_mem) instead of holding them in 35 registers per lane, and the kernel fits SEVEN waves per SIMD
// (72 registers, nothing spilled).  A wavefront works one window at a time through a chain of dependent steps, and the number of waves a SIMD can
// switch between is what the kernel's time followed -- per 2 M windows against a 5 k list: four waves (126 registers, what the compiler picks for the
// kernel with everything in it) 9.37 ms; the same code held to five / six / seven waves by the launch bounds, with 5 / 21 / 29 spilled values, 8.06 /
// 7.58 / 7.41 (not shipped: a value spilled inside a divergent region is stored for the active lanes only, NOTES R4.5); this form 6.80, and at eight
// waves (6 spills) 7.76.  At seven waves VALU + SALU issue fills the SIMD cycles: what is left is instruction count.
// (the dense-list form of the table path -- ed <= 2 against the whole 3.6 M list: 79 registers, six waves, nothing spilled: 23.4 ms per 2 M windows;
// with the answers in registers and spills: four waves 29.9 ms, five 26.3, six 24.6, seven 26.9)
#ifndef SMI_BC2_DENSE_WAVES
#define SMI_BC2_DENSE_WAVES 6
#endif
template <bool kTwoStage, bool kFilter, bool kTable = false>
__global__ __launch_bounds__(256, kTable ? (kTwoStage ? SMI_BC2_DENSE_WAVES : 7) : 1) void k_bc_match_ed2(const smi_bc_window *__restrict__ win, size_t n, int five_prime,
                                                      Pyramid P, smi_bc_result *__restrict__ out) {
    __shared__ uint32_t s_keys[4][kTabSlots];
    __shared__ uint32_t s_vals[4][kTabSlots];
    __shared__ uint32_t s_low[4][128];
    __shared__ uint32_t s_ord2e[4][128];
    __shared__ uint32_t s_pass[4][128];
    __shared__ uint32_t s_vpos[4][kTabSlots];
    __shared__ uint32_t s_res[4][36];  // bc[15] | rs[5] | imd[15]
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    uint32_t *keys = s_keys[wv], *vals = s_vals[wv], *lows = s_low[wv], *ord2e = s_ord2e[wv], *passf = s_pass[wv], *vpos = s_vpos[wv];
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const bool fp = five_prime != 0;
    constexpr int OFFS[5] = {0, -1, 1, -2, 2};
    // level-2 child slots of a lane: slot = 8 * position + kind, the same enumeration as level 1 (slots >= 123 unused)
    LaneMasks mk[2];
    if (!kTable) {
        mk[0] = make_masks(lane);
        mk[1] = make_masks(64 + lane);
        if (64 + lane >= 123) mk[1].valid = 0u;
    }

    for (size_t rd = wave; rd < n; rd += n_waves) {
        const smi_bc_window w = win[rd];  // wave-uniform (scalar loads)
        smi_bc_result res;
        res.bc = 0;
        res.ed_sec = 2147483647;
        res.found = -1;
        res.ed = 0;
        res.offset = 0;
        res.ins_minus_del = 0;
        res.n_matches = 0;
        if (w.flags & SMI_WIN_VALID) {
            // the per-offset answers (wave-uniform) wait for pick_best15 in the wave's LDS, not in 35 registers across the offset loop: the
            // kernel's register count decides how many waves a SIMD holds, and the waves' dependent chains are what the kernel waits for
            uint32_t *r_bc = s_res[wv], *r_rs = s_res[wv] + 15;
            int *r_imd = reinterpret_cast<int *>(s_res[wv] + 20);
            uint32_t present = 0;
            // the two-step filter bits of the five windows: five independent loads in front of the per-offset loop (inside it they
            // would be five dependent round trips per read)
            uint32_t near2 = 31u;
            if (kFilter && P.nb2 != nullptr) {
                uint32_t wds[5], kys[5];
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    kys[q] = make_key(w.bases, w.nmask, OFFS[q], fp).key;
                    wds[q] = P.nb2[kys[q] >> 5];
                }
                near2 = 0u;
#pragma unroll
                for (int q = 0; q < 5; q++) near2 |= ((wds[q] >> (kys[q] & 31u)) & 1u) << q;
            }
#pragma unroll 1
            for (int q = 0; q < 5; q++) {
                const OffsetKey ok = make_key(w.bases, w.nmask, OFFS[q], fp);
                const uint32_t post1 = post_base(w.bases, w.nmask, OFFS[q], fp, 1);
                const uint32_t post2 = post_base(w.bases, w.nmask, OFFS[q], fp, 2);
                const uint32_t K = ok.key;
                uint32_t bc0 = K, bc1 = 0, bc2 = 0;
                int imd1 = 0, imd2 = 0;
                bool hit0 = false, hit1 = false, hit2 = false;
                if (ok.usable && ((near2 >> q) & 1u)) {  // (a clear bit of the two-step filter: no match at any level)
                    const Seq root = {K, 0u};
                    // ---- level 1: the 123 children, two per lane (e = lane, 64 + lane) ------------------------
                    Seq c[2];
                    bool val[2];
                    int pe[2], re[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int e = 64 * h + lane;
                        pe[h] = e >> 3;
                        re[h] = e & 7;
                        val[h] = e < 123;
                        c[h] = child_of(root, pe[h] & 15, re[h], post1);
                        lows[e] = val[h] ? c[h].low : K;  // invalid slots mirror the root (never created)
                    }
                    // Filter bit and membership of EVERY valid child, before any bookkeeping (kFilter: short used lists).  A child that is never
                    // created is a copy of one that is, so "no valid child passes the filter and none is a barcode" says the same of the created ones:
                    // then levels 1 and 2 have nothing to find, and the dedup table, the creation order and the expansion order -- most of this
                    // kernel's work per window -- are not needed.  (A window that is itself a barcode asks its children for a SECOND barcode in
                    // reach: the exact hits of a read, a fifth of the windows that get here, leave through this door.)
                    hit0 = member(P, K);
                    bool pany[2] = {false, false}, mem1[2] = {false, false};
                    if (kFilter) {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const bool k_counts = hit0 && !(re[h] == 7 && pe[h] == 0);
                            // (children of positions <= 6 in the suffix-major tables, where they share three lines: n1_cell_s)
                            const bool sfx = P.n1s != nullptr && pe[h] <= 6;
                            const uint32_t *tab = sfx ? (k_counts ? P.n2s : P.n1s) : (k_counts ? P.n2 : P.n1);
                            const uint32_t cell = sfx ? n1_cell_s(c[h].low) : n1_cell(c[h].low);
                            const bool live = val[h] && c[h].g == 0u;
                            pany[h] = live && ((tab[cell >> 5] >> (cell & 31u)) & 1u);
                            mem1[h] = live && member(P, c[h].low);
                        }
                    }
                    if (!kFilter || __ballot(pany[0] || pany[1] || mem1[0] || mem1[1])) {  // (wave-uniform)
                    // clear the tables
#pragma unroll
                    for (int k = 0; k < kTabSlots / 64; k++) {
                        keys[64 * k + lane] = kEmpty;
                        vals[64 * k + lane] = 0xFFFFFFFFu;
                        vpos[64 * k + lane] = 0xFFFFFFFFu;
                    }
                    wave_sync();
                    // created(e): first position among equal sequences, and not the root except at position 0.  The first position of
                    // a sequence comes out of the dedup table (slot of the sequence -> smallest position), not out of a comparison of
                    // every child with every other one (123 LDS reads per child: two thirds of this kernel's instructions once the
                    // item filter had removed the level-2 work).  The all-T key equals the empty-slot sentinel: wave reductions.
                    uint32_t slot_of[2] = {0, 0};
                    uint32_t t_pmin = 99u;
                    {
                        uint32_t mine = 99u;
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (!val[h]) continue;
                            if (c[h].low == kEmpty) {
                                mine = min(mine, (uint32_t)pe[h]);
                                continue;
                            }
                            uint32_t slot = (c[h].low * 2654435761u) >> 24;
                            for (;;) {
                                const uint32_t prev = atomicCAS(&keys[slot], kEmpty, c[h].low);
                                if (prev == kEmpty || prev == c[h].low) break;
                                slot = (slot + 1) & (kTabSlots - 1);
                            }
                            atomicMin(&vpos[slot], (uint32_t)pe[h]);
                            slot_of[h] = slot;
                        }
                        if (__ballot(mine != 99u)) {
#pragma unroll
                            for (int o = 32; o > 0; o >>= 1) mine = min(mine, (uint32_t)__shfl_xor((int)mine, o));
                            t_pmin = mine;
                        }
                    }
                    wave_sync();
                    bool created[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t pmin = !val[h] ? 99u : (c[h].low == kEmpty ? t_pmin : vpos[slot_of[h]]);
                        created[h] = val[h] && (uint32_t)pe[h] == pmin && (c[h].low != K || pe[h] == 0);
                    }
                    const unsigned long long ca = __ballot(created[0]), cb = __ballot(created[1]);
                    const int n_items = __popcll(ca) + __popcll(cb);
                    // expansion order: positions ascending, reverse creation order inside a position
                    uint32_t ord[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const unsigned long long mine = h == 0 ? ca : cb;
                        const int base_bit = 8 * (pe[h] & 7);  // first lane of this position inside its ballot
                        const int before = (h == 1 ? __popcll(ca) : 0) + __popcll(mine & ((1ull << base_bit) - 1ull));
                        const unsigned long long sib = (mine >> base_bit) & 0xFFull;
                        const int later = __popcll(sib >> (re[h] + 1));  // created siblings generated after me
                        ord[h] = (uint32_t)(before + later);
                        if (created[h]) ord2e[ord[h]] = (uint32_t)(64 * h + lane);
                    }
                    // dedup table: low 32 bits -> smallest expansion order; the all-T key is kept in a register instead
                    uint32_t t_ord = 0xFFFFFFFFu;
                    {
                        uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
                        for (int h = 0; h < 2; h++)
                            if (created[h] && c[h].low == kEmpty) mine = min(mine, ord[h]);
                        if (__ballot(mine != 0xFFFFFFFFu)) {
#pragma unroll
                            for (int o = 32; o > 0; o >>= 1) mine = min(mine, (uint32_t)__shfl_xor((int)mine, o));
                            t_ord = mine;
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; h++)
                        if (created[h] && c[h].low != kEmpty) atomicMin(&vals[slot_of[h]], ord[h]);
                    // items worth expanding, as a bit mask over the expansion order
                    unsigned long long pm0 = ~0ull, pm1 = ~0ull;
                    bool pass_h[2] = {created[0] && c[0].g == 0u, created[1] && c[1].g == 0u};  // this lane's items that are worth expanding
                    if (!kFilter && P.nt != nullptr && P.nb != nullptr) {
                        // lists too long for the hashed item filter: the exact bitmap of K-BC1's offset filter answers the same question
                        // (is any barcode one step away from this item?) before the item's bucket is read
#pragma unroll
                        for (int h = 0; h < 2; h++)
                            pass_h[h] = pass_h[h] && ((P.nb[c[h].low >> 5] >> (c[h].low & 31u)) & 1u);
                    }
                    if (kFilter) {
                        // K a barcode: its children have K as a neighbour by construction (substitutions; insertion children through the
                        // "base inserted, last dropped" members of K's neighbourhood or, behind position 14, as substitutions of the last
                        // base; deletion children through "position removed, any last base" for positions >= 1) -- all but the child
                        // that lost position 0 -- and K itself is never a level-2 hit, so they need a SECOND barcode in reach
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (!created[h]) continue;
                            pass_h[h] = pany[h];  // (read above, for every valid child)
                            passf[ord[h]] = pass_h[h] ? 1u : 0u;
                        }
                    }
                    if (!kFilter && P.nt != nullptr) {
#pragma unroll
                        for (int h = 0; h < 2; h++)
                            if (created[h]) passf[ord[h]] = pass_h[h] ? 1u : 0u;
                    }
                    wave_sync();
                    if (kFilter || P.nt != nullptr) {  // the passing items as a mask over the expansion order
                        pm0 = __ballot(lane < n_items && passf[lane] != 0u);
                        pm1 = __ballot(64 + lane < n_items && passf[64 + lane] != 0u);
                    } else {
                        pm0 = n_items >= 64 ? ~0ull : ((1ull << n_items) - 1ull);
                        pm1 = n_items <= 64 ? 0ull : (n_items >= 128 ? ~0ull : ((1ull << (n_items - 64)) - 1ull));
                    }
                    // ---- probes of level 0 and 1 (only created children are ever probed) ----------------------
                    const bool h1a = created[0] && (kFilter ? mem1[0] : (c[0].g == 0u && member(P, c[0].low)));
                    const bool h1b = created[1] && (kFilter ? mem1[1] : (c[1].g == 0u && member(P, c[1].low)));
                    const unsigned long long ha = __ballot(h1a), hb = __ballot(h1b);
                    if (ha | hb) {
                        const int e = ha ? __builtin_ctzll(ha) : 64 + __builtin_ctzll(hb);
                        hit1 = true;
                        bc1 = child_of(root, e >> 3, e & 7, post1).low;
                        imd1 = ins_minus_del_of(e);
                    }
                    // ---- level 2: expand the created children in order, early exit on the first hit ------------
                    // kGroup items (2 * kGroup half-rounds of 64 mutants) are generated, looked up in the dedup table and
                    // tested against the top level TOGETHER: the loads of a group are independent, so their latencies
                    // overlap; the (rare) survivors of the top level are then walked in order, which keeps "first hit".
                    if (kTable || P.nt != nullptr) {
                        // Level 2 from the neighbourhood table (k_set_nt): an item X is not expanded into its 123 children; the bucket of
                        // X lists every mutation that turns X into a barcode, and the rules of the expansion are applied to those few --
                        // the position the item was created at is not visited, an insertion behind position 14 needs X to end in A, a
                        // deletion appends the read's next base, the root / X itself behind its first position / a sequence that was
                        // expanded earlier are in the dedup set.  Every lane does this for its own two items at once; the item that comes
                        // first in the expansion order among those with a match is the one the serial expansion would have stopped at.
                        auto ord_seen_of = [&](uint32_t ml) -> uint32_t {
                            if (ml == kEmpty) return t_ord;
                            uint32_t sl = (ml * 2654435761u) >> 24;
                            for (;;) {
                                const uint32_t kk = keys[sl];
                                if (kk == ml) return vals[sl];
                                if (kk == kEmpty) return 0xFFFFFFFFu;
                                sl = (sl + 1) & (kTabSlots - 1);
                            }
                        };
                        // The passing items are few (a handful of the 123) and sit in arbitrary lanes: they are taken in expansion order, eight
                        // per round, and a group of eight lanes reads one item's bucket -- one 8-byte entry per lane, 64 contiguous bytes per
                        // group -- instead of every owning lane walking its eight entries while the rest of the wave idles.
                        uint32_t cand = 0xFFFFFFFFu;  // ord << 8 | s2 of the best item this lane has seen
                        unsigned long long r0 = pm0, r1 = pm1;  // (wave-uniform: ballots)
                        const int grp = lane >> 3, sub = lane & 7;
                        while (r0 | r1) {
                            int my_t = -1;
#pragma unroll
                            for (int g8 = 0; g8 < 8; g8++) {
                                int t = -1;
                                if (r0) {
                                    t = __builtin_ctzll(r0);
                                    r0 &= r0 - 1;
                                } else if (r1) {
                                    t = 64 + __builtin_ctzll(r1);
                                    r1 &= r1 - 1;
                                }
                                if (grp == g8) my_t = t;
                            }
                            bool live = my_t >= 0;
                            const int e = (int)ord2e[live ? my_t : 0];
                            const uint32_t X = lows[e];
                            const int pX = (e >> 3) & 15, rX = e & 7;
                            const uint32_t delb = (rX >= 3 && rX <= 6) ? post2 : post1;  // post[nDeletions + 1]
                            const int q0 = pX == 0 ? 1 : 0;
                            uint32_t best = 255u;
                            uint32_t bucket = nt_slot(X, P.nt_cap);
                            for (;;) {
                                const uint64_t en = live ? P.nt[bucket + (uint32_t)sub] : 1ull;
                                if (live && (uint32_t)(en >> 8) == X && (en >> 40)) {
                                    const uint32_t kind = (uint32_t)en & 3u, pos = ((uint32_t)en >> 2) & 15u, base = ((uint32_t)en >> 6) & 3u;
                                    // X itself is no mutant of X; the item skips its own position
                                    bool use = kind != 0u && (int)pos != pX;
                                    uint32_t r2 = 7u;
                                    if (kind == 1u) {
                                        const uint32_t cur = (X >> (30 - 2 * pos)) & 3u;
                                        r2 = base - (base > cur ? 1u : 0u);
                                    } else if (kind == 2u) {
                                        use = use && !(pos == 14u && (X & 3u) != 0u);
                                        r2 = 3u + base;
                                    } else {
                                        use = use && base == delb;
                                    }
                                    if (use) {
                                        const Seq Xs = {X, 0u};
                                        const uint32_t ml = child_of(Xs, (int)pos, (int)r2, delb).low;
                                        // the root, X itself after its first position, a sequence that was expanded earlier
                                        if (!(ml == K || (ml == X && (int)pos > q0)) && !(ord_seen_of(ml) < (uint32_t)my_t)) best = min(best, 8u * pos + r2);
                                    }
                                }
                                // a bucket with a free entry ends the item's chain
                                const unsigned long long z = __ballot(live && en == 0ull);
                                if ((z >> (lane & 56)) & 0xFFull) live = false;
                                if (!__ballot(live)) break;
                                bucket = bucket + 8 == P.nt_cap ? 0u : bucket + 8;
                            }
                            best = min(best, (uint32_t)__shfl_xor((int)best, 1));
                            best = min(best, (uint32_t)__shfl_xor((int)best, 2));
                            best = min(best, (uint32_t)__shfl_xor((int)best, 4));
                            if (my_t >= 0 && best != 255u) cand = min(cand, ((uint32_t)my_t << 8) | best);
                        }
                        uint32_t first_c = cand;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) first_c = min(first_c, (uint32_t)__shfl_xor((int)first_c, o));
                        if (first_c != 0xFFFFFFFFu) {
                            const int t = (int)(first_c >> 8), s2 = (int)(first_c & 0xFFu);
                            const int e = (int)ord2e[t];
                            const int pX = e >> 3, rX = e & 7;
                            const Seq X = child_of(root, pX, rX, post1);
                            const uint32_t delb = (rX >= 3 && rX <= 6) ? post2 : post1;
                            hit2 = true;
                            bc2 = child_of(X, s2 >> 3, s2 & 7, delb).low;
                            imd2 = ins_minus_del_of(e) + ins_minus_del_of(s2);
                        }
                        pm0 = pm1 = 0ull;  // nothing is left for the enumerating loop
                    }
                    constexpr int kGroup = 4;
                    if constexpr (!kTable)
                    while ((pm0 | pm1) && !hit2) {
                        uint32_t m_low[2 * kGroup], m_ok[2 * kGroup], w0[2 * kGroup], w1[2 * kGroup];
                        int t_sel[kGroup];  // the next kGroup items in expansion order (wave-uniform)
#pragma unroll
                        for (int g = 0; g < kGroup; g++) {
                            if (pm0) {
                                t_sel[g] = __builtin_ctzll(pm0);
                                pm0 &= pm0 - 1;
                            } else if (pm1) {
                                t_sel[g] = 64 + __builtin_ctzll(pm1);
                                pm1 &= pm1 - 1;
                            } else
                                t_sel[g] = -1;
                        }
#pragma unroll
                        for (int g = 0; g < kGroup; g++) {
                            const int t = t_sel[g];
                            const bool on = t >= 0;  // wave-uniform
                            // the item being expanded is wave-uniform: keep it on the scalar unit
                            const int e = __builtin_amdgcn_readfirstlane((int)ord2e[on ? t : 0]);
                            const int pX = e >> 3, rX = e & 7;
                            // the item is one of the level-1 children, whose low words are in LDS already; its g bits
                            // are non-zero only for "insert behind position 14" (child_of)
                            Seq X;
                            X.low = (uint32_t)__builtin_amdgcn_readfirstlane((int)lows[e]);
                            X.g = (rX >= 3 && rX <= 6 && pX == 14) ? (K & 3u) : 0u;
                            const uint32_t delb = (rX >= 3 && rX <= 6) ? post2 : post1;  // post[nDeletions + 1]
                            const int q0 = pX == 0 ? 1 : 0;                              // first visited position
                            const bool x_ok = on && X.g == 0u;  // a child of a g != 0 item can never equal a barcode
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                const int qq = (64 * h + lane) >> 3;
                                uint32_t lv;
                                Seq m;
                                m.low = mutate2(mk[h], X.low, delb, lv);  // lv: valid slot and g == 0
                                // dedup set: root, anything expanded before X, X itself after its first position
                                bool live = x_ok && lv != 0u && qq != pX && m.low != K && !(m.low == X.low && qq > q0);
                                m_low[2 * g + h] = m.low;
                                m_ok[2 * g + h] = live ? 1u : 0u;
                                if (kTwoStage) {
                                    // both stages of the top level from one 8-byte load (t2, prefix-major part): one probe in
                                    // a hundred instead of one in ten goes on to the dedup table and the lower levels
                                    const uint2 e2 = reinterpret_cast<const uint2 *>(P.t2)[live ? (m.low >> (kG0 + 5)) : 0u];
                                    w0[2 * g + h] = e2.x;
                                    w1[2 * g + h] = e2.y;
                                } else {  // a used list of a few thousand barcodes: the 4 MiB l0 stays in L2 and lets almost nothing through
                                    w0[2 * g + h] = P.l0[live ? (m.low >> (kG0 + 5)) : 0u];
                                    w1[2 * g + h] = 0xFFFFFFFFu;
                                }
                            }
                        }
                        // top-level test of the whole group first: with a short used list nothing survives it
                        uint32_t any_pass = 0;
#pragma unroll
                        for (int i = 0; i < 2 * kGroup; i++) {
                            m_ok[i] &= (w0[i] >> ((m_low[i] >> kG0) & 31u)) & (w1[i] >> t2_prefix_bit(m_low[i])) & 1u;
                            any_pass |= m_ok[i];
                        }
                        if (!__ballot(any_pass != 0u)) continue;
#pragma unroll
                        for (int i = 0; i < 2 * kGroup; i++) {
                            if (hit2) break;
                            bool pass = m_ok[i] != 0u;
                            if (!__ballot(pass)) continue;
                            // Only a mutant that can be a barcode needs the dedup set ("expanded earlier" cannot change
                            // the outcome of a probe that misses), so the LDS look-up runs on the few top-level survivors
                            const int t = t_sel[i >> 1];
                            if (pass) {
                                const uint32_t ml = m_low[i];
                                uint32_t ord_seen = 0xFFFFFFFFu;  // expansion order of an equal sequence, if any
                                if (ml == kEmpty)
                                    ord_seen = t_ord;
                                else {
                                    uint32_t sl = (ml * 2654435761u) >> 24;
                                    for (;;) {
                                        const uint32_t kk = keys[sl];
                                        if (kk == ml) {
                                            ord_seen = vals[sl];
                                            break;
                                        }
                                        if (kk == kEmpty) break;
                                        sl = (sl + 1) & (kTabSlots - 1);
                                    }
                                }
                                pass = !(ord_seen < (uint32_t)t);
                            }
                            const bool hit = pass && ((P.l1[l1_word(m_low[i])] >> l1_bit(m_low[i])) & 1u) && bit_of(P.fine, m_low[i]);
                            const unsigned long long hm = __ballot(hit);
                            if (hm) {
                                const int e = (int)ord2e[t];
                                const int pX = e >> 3, rX = e & 7;
                                const Seq X = child_of(root, pX, rX, post1);
                                const uint32_t delb = (rX >= 3 && rX <= 6) ? post2 : post1;
                                const int s2 = 64 * (i & 1) + __builtin_ctzll(hm);
                                hit2 = true;
                                bc2 = child_of(X, s2 >> 3, s2 & 7, delb).low;
                                imd2 = ins_minus_del_of(e) + ins_minus_del_of(s2);
                            }
                        }
                    }
                    }  // levels 1 and 2
                }
                if (lane == 0) {
                    r_rs[q] = K;
                    r_bc[3 * q] = bc0;
                    r_bc[3 * q + 1] = bc1;
                    r_bc[3 * q + 2] = bc2;
                    r_imd[3 * q] = 0;
                    r_imd[3 * q + 1] = imd1;
                    r_imd[3 * q + 2] = imd2;
                }
                present |= (hit0 ? 1u : 0u) << (3 * q);
                present |= (hit1 ? 1u : 0u) << (3 * q + 1);
                present |= (hit2 ? 1u : 0u) << (3 * q + 2);
            }
            wave_sync();
            pick_best15_mem(r_bc, r_rs, r_imd, present, 2, res);
            wave_sync();  // (the next read writes s_res again)
        }
        if (lane == 0) out[rd] = res;
    }
}

int launch_bc_match2(smi_ctx *ctx, const smi_bc_window *d_win, size_t n, int five_prime, smi_bc_result *d_out,
                     hipStream_t s) {
    if (!n) return SMI_OK;
    Pyramid P = pyramid_of(ctx);
    const unsigned grid = (unsigned)std::min<size_t>((n + 3) / 4, 256 * 32);
    if (int rc = time_begin(ctx, SMI_K_BC_MATCH, s)) return rc;
    // density of the top level: 2^25 cells; from ~1 % occupied cells on the second stage pays for its wider loads
    // level 2 from the table pays when the table is small enough to stay in cache (a used list: 11 MB); against the whole whitelist the 550
    // bucket reads of a read miss to HBM (measured 29 instead of 32 M reads/s), so dense sets keep the enumeration through the two-stage top level
    if (std::getenv("SMI_BC2_NO_TABLE") || (ctx->n_keys > 300000 && std::getenv("SMI_BC2_DENSE_ENUM"))) P.nt = nullptr;
    if (ctx->n_keys > 300000 && P.nt)
        hipLaunchKernelGGL((k_bc_match_ed2<true, false, true>), dim3(grid), dim3(256), 0, s, d_win, n, five_prime, P, d_out);
    else if (ctx->n_keys > 300000)
        hipLaunchKernelGGL((k_bc_match_ed2<true, false>), dim3(grid), dim3(256), 0, s, d_win, n, five_prime, P, d_out);
    else if (P.n1 && P.nt)
        hipLaunchKernelGGL((k_bc_match_ed2<false, true, true>), dim3(grid), dim3(256), 0, s, d_win, n, five_prime, P, d_out);
    else if (P.n1)
        hipLaunchKernelGGL((k_bc_match_ed2<false, true>), dim3(grid), dim3(256), 0, s, d_win, n, five_prime, P, d_out);
    else
        hipLaunchKernelGGL((k_bc_match_ed2<false, false>), dim3(grid), dim3(256), 0, s, d_win, n, five_prime, P, d_out);
    SMI_HIP(hipGetLastError());
    if (int rc = time_end(ctx, SMI_K_BC_MATCH, s)) return rc;
    return SMI_OK;
}

}  // namespace smi
