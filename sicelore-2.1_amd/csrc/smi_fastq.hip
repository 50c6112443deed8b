// smi_fastq.hip -- K-FQ: FASTQ text resident in HBM -> record index and contiguous read / quality buffers.
//
// SURVEY section 8f.1 (the caller side of the path): replaces the htsjdk FastqReader loop of
// FastqFileReader$OneFastqFileWorker (FJ!nanoporereadscanner/readerwriter/FastqFileReader.java:L138-167 submits one per
// file; 10,000-record chunks, WorkerReadscanner.java:L186) for uncompressed text: gz inflate stays on the host.
// htsjdk 4.1.3 FastqReader semantics kept: four lines per record, '@' and '+' headers, sequence and quality lines of
// equal length, line ends "\n" or "\r\n"; a violation is reported through the error word, never repaired.
//
// MI355X mapping: newline positions are found by all lanes at once (64 bytes per lane, block-level counts, one
// exclusive scan over the blocks, then a second sweep writes the start of every line); records are then four consecutive
// line starts.  Byte work at HBM speed, no MFMA.
#include <hipcub/hipcub.hpp>

#include "smi_internal.h"

namespace smi {

constexpr int kFqBlock = 256;
constexpr int kFqBytesPerThread = 64;
constexpr int kFqTile = kFqBlock * kFqBytesPerThread;  // 16 KiB of text per block

// newline flags of 16 bytes: bit k set when p[k] == '\n' (SWAR zero-byte test per dword)
__device__ __forceinline__ uint32_t nl_mask16w(const uint32_t (&w)[4]) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t t = w[k] ^ 0x0A0A0A0Au;
        const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);  // 0x80 in every zero byte, exact
        m |= (((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xFu) << (4 * k);
    }
    return m;
}
// newline flags of the 64 bytes at i0: four 16-byte loads in flight together
__device__ __forceinline__ uint64_t nl_mask64(const uint8_t *p, size_t i0, size_t n) {
    if (i0 + kFqBytesPerThread <= n) {
        uint32_t w[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++) __builtin_memcpy(w[q], p + i0 + 16 * q, 16);
        uint64_t m = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) m |= (uint64_t)nl_mask16w(w[q]) << (16 * q);
        return m;
    }
    uint64_t m = 0;
    for (int k = 0; k < kFqBytesPerThread; k++) m |= (i0 + k < n && p[i0 + k] == '\n') ? (1ull << k) : 0ull;
    return m;
}

// sum over the block (every thread gets it): wave sums through shuffles, the four of them through LDS
__device__ __forceinline__ uint32_t block_sum_and_prefix(uint32_t mine, uint32_t *wave_tot, uint32_t &before) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(inc, o);
        if (lane >= o) inc += y;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kFqBlock / 64; k++) {
        const uint32_t t = wave_tot[k];
        base += k < wv ? t : 0u;
        total += t;
    }
    before = base + inc - mine;
    return total;
}

__global__ __launch_bounds__(kFqBlock) void k_fq_count(const uint8_t *__restrict__ text, size_t n, uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t wave_tot[kFqBlock / 64];
    const size_t i0 = (size_t)blockIdx.x * kFqTile + (size_t)threadIdx.x * kFqBytesPerThread;
    uint32_t before;
    const uint32_t total = block_sum_and_prefix(i0 < n ? (uint32_t)__popcll(nl_mask64(text, i0, n)) : 0u, wave_tot, before);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

// line_start[L] = first byte of line L (line 0 starts at 0); block_base[b] = newlines before block b
__global__ __launch_bounds__(kFqBlock) void k_fq_lines(const uint8_t *__restrict__ text, size_t n,
                                                       const uint64_t *__restrict__ block_base, uint64_t *__restrict__ line_start,
                                                       size_t cap_lines) {
    __shared__ uint32_t wave_tot[kFqBlock / 64];
    const size_t i0 = (size_t)blockIdx.x * kFqTile + (size_t)threadIdx.x * kFqBytesPerThread;
    const uint64_t mask = i0 < n ? nl_mask64(text, i0, n) : 0ull;
    uint32_t before;
    block_sum_and_prefix((uint32_t)__popcll(mask), wave_tot, before);
    uint64_t line = block_base[blockIdx.x] + before;  // newlines before my first byte
    if (blockIdx.x == 0 && threadIdx.x == 0 && cap_lines > 0) line_start[0] = 0;
    for (uint64_t m = mask; m; m &= m - 1) {
        line++;
        if (line < cap_lines) line_start[line] = i0 + (size_t)__builtin_ctzll(m) + 1;
    }
}

// ---- the line index with ONE sweep over the text (round 4): the count sweep also leaves every thread's 64 newline flags (one bit per byte of
// text, 1/8 of its size), and the sweep that writes the line starts reads those instead of the text again: 1.2 GB + 2 x 0.15 GB through HBM
// instead of 2 x 1.2 GB.  (Tried first: a true single sweep, tiles publishing their counts and looking back over their predecessors' --
// with 16 KiB tiles arriving every 3 ns and a look-back step of ~1 us every tile walks the whole resident window, 150 M device-scope loads
// per chunk: 7 ms instead of 0.4.)
__global__ __launch_bounds__(kFqBlock) void k_fq_count_masks(const uint8_t *__restrict__ text, size_t n, uint32_t *__restrict__ block_counts,
                                                             uint64_t *__restrict__ masks) {
    __shared__ uint32_t wave_tot[kFqBlock / 64];
    const size_t t = (size_t)blockIdx.x * kFqBlock + threadIdx.x, i0 = t * kFqBytesPerThread;
    const uint64_t mask = i0 < n ? nl_mask64(text, i0, n) : 0ull;
    masks[t] = mask;
    uint32_t before;
    const uint32_t total = block_sum_and_prefix((uint32_t)__popcll(mask), wave_tot, before);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}
__global__ __launch_bounds__(kFqBlock) void k_fq_lines_masks(const uint64_t *__restrict__ masks, const uint64_t *__restrict__ block_base,
                                                             uint64_t *__restrict__ line_start, size_t cap_lines) {
    __shared__ uint32_t wave_tot[kFqBlock / 64];
    const size_t t = (size_t)blockIdx.x * kFqBlock + threadIdx.x, i0 = t * kFqBytesPerThread;
    const uint64_t mask = masks[t];
    uint32_t before;
    block_sum_and_prefix((uint32_t)__popcll(mask), wave_tot, before);
    uint64_t line = block_base[blockIdx.x] + before;  // newlines before my first byte
    if (blockIdx.x == 0 && threadIdx.x == 0 && cap_lines > 0) line_start[0] = 0;
    for (uint64_t m = mask; m; m &= m - 1) {
        line++;
        if (line < cap_lines) line_start[line] = i0 + (size_t)__builtin_ctzll(m) + 1;
    }
}

// what the host used to compute between the two halves of the index, on the device: the record kernel and the length scan are queued
// right behind the line sweep, and the host reads these words once, at the end of the call
struct FqTotals {
    uint64_t n_newlines, n_lines, n_rec;  // n_rec = 0 when the buffers are too small (`overflow` says which)
    uint32_t err, overflow;               // err: SMI_FQ_* of the record kernel; overflow: 1 line buffer, 2 record buffers
};
__global__ void k_fq_totals(const uint8_t *__restrict__ text, size_t n_bytes, size_t n_blocks, const uint64_t *__restrict__ block_base,
                            const uint32_t *__restrict__ block_counts, size_t cap_lines, size_t cap_records, FqTotals *__restrict__ t) {
    if (blockIdx.x || threadIdx.x) return;
    const uint64_t nl = block_base[n_blocks - 1] + block_counts[n_blocks - 1];
    const uint64_t lines = nl + (text[n_bytes - 1] == '\n' ? 0 : 1);  // a last line without newline still counts
    uint64_t rec = lines / 4;
    uint32_t over = 0;
    if (lines + 1 > cap_lines) over |= 1u;
    if (rec >= cap_records) over |= 2u;  // the length scan reads d_seq_len[n_rec]: one spare entry is part of the contract
    if (over) rec = 0;
    t->n_newlines = nl;
    t->n_lines = lines;
    t->n_rec = rec;
    t->overflow = over;
}

// record r = lines 4r .. 4r+3
__global__ void k_fq_records(const uint8_t *__restrict__ text, size_t n_bytes, const uint64_t *__restrict__ line_start,
                             FqTotals *__restrict__ tot, size_t cap_records, uint64_t *__restrict__ name_start,
                             uint32_t *__restrict__ name_len, uint64_t *__restrict__ seq_start, uint32_t *__restrict__ seq_len,
                             uint64_t *__restrict__ qual_start) {
    const size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const uint64_t n_lines_complete = tot->n_newlines;
    const size_t n_rec = (size_t)tot->n_rec;
    uint32_t *err = &tot->err;
    if (r >= n_rec) {
        if (r < cap_records) seq_len[r] = 0;  // (the length scan runs over the whole capacity: offsets[n_rec ..] = total bases)
        return;
    }
    auto line_end = [&](uint64_t L) -> uint64_t {  // one past the last character of line L (CR / LF stripped)
        uint64_t e = L + 1 <= n_lines_complete ? line_start[L + 1] - 1 : n_bytes;  // the last line may lack its newline
        if (e > line_start[L] && text[e - 1] == '\r') e--;
        return e;
    };
    const uint64_t l0 = line_start[4 * r], l1 = line_start[4 * r + 1], l2 = line_start[4 * r + 2], l3 = line_start[4 * r + 3];
    const uint64_t e0 = line_end(4 * r), e1 = line_end(4 * r + 1), e2 = line_end(4 * r + 2), e3 = line_end(4 * r + 3);
    uint32_t bad = 0;
    if (e0 == l0 || text[l0] != '@') bad |= SMI_FQ_BAD_SEQ_HEADER;
    if (e2 == l2 || text[l2] != '+') bad |= SMI_FQ_BAD_QUAL_HEADER;
    if (e1 - l1 != e3 - l3) bad |= SMI_FQ_LENGTH_MISMATCH;
    if (bad) atomicOr(err, bad);
    name_start[r] = l0 + 1;
    name_len[r] = (uint32_t)(e0 > l0 ? e0 - l0 - 1 : 0);
    seq_start[r] = l1;
    seq_len[r] = (uint32_t)(e1 - l1);
    qual_start[r] = l3;
}

// one wave per record: out[offsets[r] .. offsets[r+1]) = text[start[r] ..]
__global__ __launch_bounds__(256) void k_fq_gather(const uint8_t *__restrict__ text, const uint64_t *__restrict__ start,
                                                   const uint64_t *__restrict__ offsets, size_t n, uint8_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < n; r += n_waves) {
        const uint64_t s = start[r], o = offsets[r], len = offsets[r + 1] - o;
        for (uint64_t i = lane; i < len; i += 64) out[o + i] = text[s + i];
    }
}

static int ensure_scan_tmp(smi_ctx *ctx, size_t bytes) {
    if (bytes > ctx->scan_tmp_bytes) {
        if (ctx->scan_tmp) SMI_HIP(hipFree(ctx->scan_tmp));
        ctx->scan_tmp = nullptr;
        ctx->scan_tmp_bytes = 0;
        SMI_HIP(hipMalloc(&ctx->scan_tmp, bytes));
        ctx->scan_tmp_bytes = bytes;
    }
    return SMI_OK;
}

// number of lines of a text in device memory (a last line without newline counts): what the chunk workers size their buffers by --
// a host-side memchr pass over a 1.2 GB chunk costs 50-130 ms, this 0.25 ms
__global__ __launch_bounds__(kFqBlock) void k_fq_count_total(const uint8_t *__restrict__ text, size_t n, unsigned long long *__restrict__ total) {
    uint32_t c = 0;
    for (size_t i0 = ((size_t)blockIdx.x * kFqBlock + threadIdx.x) * kFqBytesPerThread; i0 < n;
         i0 += (size_t)gridDim.x * kFqBlock * kFqBytesPerThread)
        c += (uint32_t)__popcll(nl_mask64(text, i0, n));
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, (unsigned long long)c);
}

int launch_count_lines(smi_ctx *ctx, const uint8_t *d_text, size_t n_bytes, size_t *n_lines, hipStream_t s) {
    *n_lines = 0;
    if (!n_bytes) return SMI_OK;
    if (int rc = ensure_scan_tmp(ctx, 256)) return rc;
    unsigned long long *d_total = (unsigned long long *)ctx->scan_tmp;
    SMI_HIP(hipMemsetAsync(d_total, 0, 8, s));
    const unsigned grid = (unsigned)std::min<size_t>((n_bytes + kFqTile - 1) / kFqTile, 256 * 16);
    hipLaunchKernelGGL(k_fq_count_total, dim3(grid), dim3(kFqBlock), 0, s, d_text, n_bytes, d_total);
    SMI_HIP(hipGetLastError());
    unsigned long long h = 0;
    uint8_t last = 0;
    SMI_HIP(hipMemcpyAsync(&h, d_total, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(&last, d_text + (n_bytes - 1), 1, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    *n_lines = (size_t)h + (last == '\n' ? 0 : 1);
    return SMI_OK;
}

// scratch of the index: block counts (u32), block bases (u64), scalars, the newline flags (a u64 per thread of the sweep), hipcub temp
struct FqScratch {
    size_t n_blocks, off_base, off_scal, off_masks, off_cub, total;
};
static FqScratch fq_scratch(size_t n_bytes, bool two_sweeps, size_t cub_bytes) {
    FqScratch L;
    L.n_blocks = (n_bytes + kFqTile - 1) / kFqTile;
    L.off_base = (L.n_blocks * 4 + 255) & ~(size_t)255;
    L.off_scal = L.off_base + ((L.n_blocks * 8 + 255) & ~(size_t)255);
    L.off_masks = L.off_scal + 256;
    L.off_cub = L.off_masks + (two_sweeps ? 0 : L.n_blocks * kFqBlock * 8);
    L.total = L.off_cub + cub_bytes;
    return L;
}
__global__ void k_fq_sweep_lines(const uint8_t *__restrict__ text, size_t n_bytes, size_t n_blocks, const uint64_t *__restrict__ block_base,
                                 const uint32_t *__restrict__ block_counts, uint64_t *__restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    *out = block_base[n_blocks - 1] + block_counts[n_blocks - 1] + (text[n_bytes - 1] == '\n' ? 0 : 1);
}

// The first half of the index on its own: the ONE sweep over the text (newline flags, block counts, block bases, all left in the context's
// scratch) and the number of lines for the host, which sizes the record buffers by it.  launch_fastq_index on the same text right afterwards
// continues from the flags instead of reading the text again (the workers used to count the lines with a sweep of their own: a second
// full read of the chunk).
int launch_fastq_sweep(smi_ctx *ctx, const uint8_t *d_text, size_t n_bytes, size_t *n_lines, hipStream_t s) {
    *n_lines = 0;
    ctx->fq_swept_text = nullptr;
    if (!n_bytes) return SMI_OK;
    if (getenv("SMI_FQ_TWO_SWEEPS") != nullptr) return launch_count_lines(ctx, d_text, n_bytes, n_lines, s);
    const size_t n_blocks = (n_bytes + kFqTile - 1) / kFqTile;
    // the record count is not known yet: hipcub's scratch for as many records as a text of this size can hold at most (8 bytes each)
    const size_t max_rec = std::min<size_t>(n_bytes / 8 + 2, ((size_t)1 << 31) - 4);
    size_t cub_a = 0, cub_b = 0;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_a, (uint32_t *)nullptr, (uint64_t *)nullptr, (int)n_blocks, s));
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_b, (uint32_t *)nullptr, (uint64_t *)nullptr, (int)max_rec + 1, s));
    const FqScratch L = fq_scratch(n_bytes, false, std::max(cub_a, cub_b));
    if (int rc = ensure_scan_tmp(ctx, L.total)) return rc;
    uint8_t *tmp = (uint8_t *)ctx->scan_tmp;
    uint32_t *counts = (uint32_t *)tmp;
    uint64_t *bases = (uint64_t *)(tmp + L.off_base);
    uint64_t *d_lines = (uint64_t *)(tmp + L.off_scal + 128);
    uint64_t *masks = (uint64_t *)(tmp + L.off_masks);
    hipLaunchKernelGGL(k_fq_count_masks, dim3((unsigned)n_blocks), dim3(kFqBlock), 0, s, d_text, n_bytes, counts, masks);
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(tmp + L.off_cub, cub_a, counts, bases, (int)n_blocks, s));
    hipLaunchKernelGGL(k_fq_sweep_lines, dim3(1), dim3(1), 0, s, d_text, n_bytes, n_blocks, (const uint64_t *)bases, (const uint32_t *)counts, d_lines);
    SMI_HIP(hipGetLastError());
    uint64_t h = 0;
    uint64_t *pw = static_cast<uint64_t *>(pin_words(ctx));
    SMI_HIP(hipMemcpyAsync(pw ? pw : &h, d_lines, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    if (pw) h = *pw;
    *n_lines = (size_t)h;
    ctx->fq_swept_text = d_text;
    ctx->fq_swept_bytes = n_bytes;
    return SMI_OK;
}

int launch_fastq_index(smi_ctx *ctx, const uint8_t *d_text, size_t n_bytes, uint64_t *d_line_start, size_t cap_lines,
                       uint64_t *d_name_start, uint32_t *d_name_len, uint64_t *d_seq_start, uint32_t *d_seq_len,
                       uint64_t *d_qual_start, uint64_t *d_offsets, size_t cap_records, size_t *n_records, uint32_t *errors,
                       hipStream_t s, uint64_t *total_bases) {
    *n_records = 0;
    *errors = 0;
    if (total_bases) *total_bases = 0;
    if (!n_bytes) return SMI_OK;
    if (cap_records == 0) {
        set_error("smi_fastq_index_device: record buffers too small (need n_records + 1 entries)");
        return SMI_ERR_INVALID;
    }
    size_t cub_a = 0, cub_b = 0;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_a, (uint32_t *)nullptr, (uint64_t *)nullptr, (int)((n_bytes + kFqTile - 1) / kFqTile), s));
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_b, (uint32_t *)nullptr, (uint64_t *)nullptr, (int)cap_records + 1, s));
    const bool two_sweeps = getenv("SMI_FQ_TWO_SWEEPS") != nullptr;  // cross-check switch (read per call): the line starts from a second sweep over the text
    const FqScratch L = fq_scratch(n_bytes, two_sweeps, std::max(cub_a, cub_b));
    const size_t n_blocks = L.n_blocks, off_base = L.off_base, off_scal = L.off_scal, off_masks = L.off_masks, off_cub = L.off_cub;
    // launch_fastq_sweep ran over this very text just before: its flags, counts and bases are in the scratch (unless the scratch has to grow now)
    const bool swept = !two_sweeps && ctx->fq_swept_text == d_text && ctx->fq_swept_bytes == n_bytes && L.total <= ctx->scan_tmp_bytes;
    ctx->fq_swept_text = nullptr;
    if (int rc = ensure_scan_tmp(ctx, L.total)) return rc;
    uint8_t *tmp = (uint8_t *)ctx->scan_tmp;
    uint32_t *counts = (uint32_t *)tmp;
    uint64_t *bases = (uint64_t *)(tmp + off_base);
    FqTotals *d_tot = (FqTotals *)(tmp + off_scal);
    uint64_t *masks = (uint64_t *)(tmp + off_masks);
    SMI_HIP(hipMemsetAsync(d_tot, 0, sizeof(FqTotals), s));
    if (!swept) {
        if (two_sweeps)
            hipLaunchKernelGGL(k_fq_count, dim3((unsigned)n_blocks), dim3(kFqBlock), 0, s, d_text, n_bytes, counts);
        else
            hipLaunchKernelGGL(k_fq_count_masks, dim3((unsigned)n_blocks), dim3(kFqBlock), 0, s, d_text, n_bytes, counts, masks);
        SMI_HIP(hipcub::DeviceScan::ExclusiveSum(tmp + off_cub, cub_a, counts, bases, (int)n_blocks, s));
    }
    if (two_sweeps)
        hipLaunchKernelGGL(k_fq_lines, dim3((unsigned)n_blocks), dim3(kFqBlock), 0, s, d_text, n_bytes, bases, d_line_start, cap_lines);
    else
        hipLaunchKernelGGL(k_fq_lines_masks, dim3((unsigned)n_blocks), dim3(kFqBlock), 0, s, masks, bases, d_line_start, cap_lines);
    hipLaunchKernelGGL(k_fq_totals, dim3(1), dim3(1), 0, s, d_text, n_bytes, n_blocks, (const uint64_t *)bases, (const uint32_t *)counts, cap_lines, cap_records,
                       d_tot);
    // records and lengths for as many records as the text turns out to have (the kernel reads the totals on the device; entries behind the
    // last record get length 0), offsets over the whole capacity: nothing here waits for the host
    hipLaunchKernelGGL(k_fq_records, dim3((unsigned)((cap_records + 255) / 256)), dim3(256), 0, s, d_text, n_bytes, d_line_start, d_tot, cap_records,
                       d_name_start, d_name_len, d_seq_start, d_seq_len, d_qual_start);
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(tmp + off_cub, cub_b, d_seq_len, d_offsets, (int)cap_records, s));
    FqTotals h;
    {
        // the sum of all sequence lengths rides on the same wait: lengths behind the last record are 0, so the last offset of the buffer is
        // the total whatever the record count turns out to be (a text that fills the buffer to its last entry is refused below)
        static_assert(sizeof(FqTotals) % 8 == 0 && sizeof(FqTotals) + 8 <= 256, "the total sits behind the scalars in the pinned words");
        uint8_t *pw = static_cast<uint8_t *>(pin_words(ctx));
        uint64_t tot = 0;
        SMI_HIP(hipMemcpyAsync(pw ? (void *)pw : (void *)&h, d_tot, sizeof h, hipMemcpyDeviceToHost, s));
        if (total_bases)
            SMI_HIP(hipMemcpyAsync(pw ? (void *)(pw + sizeof h) : (void *)&tot, d_offsets + (cap_records - 1), 8, hipMemcpyDeviceToHost, s));
        SMI_HIP(hipStreamSynchronize(s));
        if (pw) {
            std::memcpy(&h, pw, sizeof h);
            std::memcpy(&tot, pw + sizeof h, 8);
        }
        if (total_bases) *total_bases = tot;
    }
    if (h.overflow & 1u) {
        set_error("smi_fastq_index_device: line buffer too small");
        return SMI_ERR_INVALID;
    }
    if (h.overflow & 2u) {
        set_error("smi_fastq_index_device: record buffers too small (need n_records + 1 entries)");
        return SMI_ERR_INVALID;
    }
    uint32_t host_err = 0;
    if (h.n_lines % 4 != 0) host_err |= SMI_FQ_TRUNCATED;  // htsjdk: "missing ... line" at end of file
    *n_records = (size_t)h.n_rec;
    *errors = host_err | h.err;
    return SMI_OK;
}

int launch_fastq_gather(smi_ctx *, const uint8_t *d_text, const uint64_t *d_start, const uint64_t *d_offsets, size_t n,
                        uint8_t *d_out, hipStream_t s) {
    if (!n) return SMI_OK;
    const unsigned grid = (unsigned)std::min<size_t>((n + 3) / 4, 256 * 64);
    hipLaunchKernelGGL(k_fq_gather, dim3(grid), dim3(256), 0, s, d_text, d_start, d_offsets, n, d_out);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

}  // namespace smi
