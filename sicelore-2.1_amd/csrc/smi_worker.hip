// smi_worker.hip -- the per-chunk workers of `scanfastq` as ONE native call each: host FASTQ text in, results out.
//
// Reference units: WorkerReadscanner.scan -> Parser.call over a FastqFileReader$ReadChunk
// (FJ!nanoporereadscanner/WorkerReadscanner.java:L186-273, FJ!nanoporereadscanner/analyzers/Parser.java:L132-185), pass 1 =
// UsedCellBCListGenerator.call (FJ!nanoporereadscanner/analyzers/UsedCellBCListGenerator.java:L198-229), and the writer loop
// of FastqWriterThreadPool$FastQoneFileThread.run (L300-306).  This is the call a JNI shim makes per chunk: everything
// between the upload of the text and the download of the finished `passed` / `failed` text runs on the device, through the
// same entry points the parity tests exercise one by one (include/sicelore_mi.h).  Device memory comes from a grow-only
// arena owned by the context, so steady-state chunks allocate nothing.
#include <algorithm>
#include <cstring>
#include <vector>

#include "smi_internal.h"

using namespace smi;

namespace {

// rank of the assigned barcode in the used list of pass 1 (sorted keys): the rk= field / BH tag
__global__ void k_rank_lookup(const smi_bc_result *__restrict__ bc, size_t n, const uint64_t *__restrict__ keys,
                              const int32_t *__restrict__ values, size_t n_keys, int32_t *__restrict__ rank) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t r = 0;
    if (bc[i].found == 1 && n_keys) {
        const uint64_t k = bc[i].bc;
        size_t lo = 0, hi = n_keys;
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if (keys[mid] < k)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < n_keys && keys[lo] == k) r = values[lo];
    }
    rank[i] = r;
}

struct Arena {
    smi_ctx *ctx;
    size_t used = 0;
    explicit Arena(smi_ctx *c) : ctx(c) {}
    template <class T>
    T *take(size_t count) {
        const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        T *p = reinterpret_cast<T *>(static_cast<uint8_t *>(ctx->arena) + used);
        used += bytes;
        return p;
    }
};

inline size_t pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

int ensure_arena(smi_ctx *ctx, size_t bytes) {
    if (ctx->arena_bytes >= bytes) return SMI_OK;
    if (ctx->arena) SMI_HIP(hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_bytes = 0;
    const size_t want = bytes + bytes / 4;  // head-room: chunks of one run are of similar size
    SMI_HIP(hipMalloc(&ctx->arena, want));
    ctx->arena_bytes = want;
    return SMI_OK;
}

size_t count_lines(const uint8_t *text, size_t n) {
    size_t lines = 0;
    const uint8_t *p = text, *end = text + n;
    while ((p = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(end - p)))) != nullptr) {
        lines++;
        p++;
    }
    return lines + ((n && text[n - 1] != '\n') ? 1 : 0);
}

#define SMI_RC(call)                 \
    do {                             \
        const int rc__ = (call);     \
        if (rc__ != SMI_OK) return rc__; \
    } while (0)

}  // namespace

extern "C" int smi_pass2_default_config(smi_pass2_config *cfg) {
    if (!cfg) {
        set_error("smi_pass2_default_config: null argument");
        return SMI_ERR_INVALID;
    }
    std::memset(cfg, 0, sizeof *cfg);
    cfg->max_ed = 1;          // --bcEditDistance of quickrun-2.1.sh
    cfg->split_chimeras = 1;  // Parser.java:L176
    cfg->first_read_id = 1;   // FastqRecordExt.READCOUNTER starts at 0, incrementAndGet (L40-43)
    return SMI_OK;
}

extern "C" int smi_scanfastq_pass2_chunk(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, const smi_pass2_config *cfg,
                                         smi_pass2_output *out) {
    if (!ctx || !cfg || !out || (!text && n_bytes)) {
        set_error("smi_scanfastq_pass2_chunk: null argument");
        return SMI_ERR_INVALID;
    }
    std::memset(out, 0, sizeof *out);
    if (n_bytes == 0) return SMI_OK;
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const bool five = cfg->five_prime != 0;
    const bool split = cfg->split_chimeras && !(five && cfg->dont_search_polya);  // Parser.java:L176
    const size_t cap = count_lines(text, n_bytes) / 4 + 2;  // records
    // worst-case sizes before anything is known about the chunk: bases + qualities <= text, fragments <= 3 per record
    const size_t m_cap = split ? 3 * cap : cap;
    const size_t bases_cap = n_bytes;
    const size_t planes_words = split ? smi_read_planes_words(bases_cap, cap) : 0;
    const size_t out_cap = 2 * bases_cap + n_bytes + 320 * m_cap + 64;
    size_t need = pad(n_bytes) + pad((4 * cap + 8) * 8) + 4 * pad(cap * 8) + pad((cap + 1) * 8) + 2 * pad(cap * 4) + 2 * pad(bases_cap) +
                  pad(planes_words * 4) + pad(cap * sizeof(smi_chimera_result)) + pad(((cap + 1023) / 1024 + 1) * 4) + pad(8) +
                  pad((3 * cap + 1) * 8) + pad(3 * cap * 4) + pad((size_t)SMI_ENDS_ROWS * 2 * m_cap * 4) + 2 * pad(m_cap * 4) +
                  pad(m_cap * (size_t)SMI_END_BASES) + pad(m_cap * sizeof(smi_scan_result)) + pad(m_cap * sizeof(smi_bc_window)) +
                  pad(m_cap * sizeof(smi_bc_result)) + pad(m_cap * 4) + pad(cfg->n_ranks * 8) + pad(cfg->n_ranks * 4) +
                  2 * pad(out_cap) + pad((m_cap + 1) * 8) + pad(m_cap) + 4096;
    SMI_RC(ensure_arena(ctx, need));
    Arena A(ctx);
    uint8_t *d_text = A.take<uint8_t>(n_bytes);
    uint64_t *d_line = A.take<uint64_t>(4 * cap + 8);
    uint64_t *d_ns = A.take<uint64_t>(cap), *d_ss = A.take<uint64_t>(cap), *d_qs = A.take<uint64_t>(cap);
    uint64_t *d_offs = A.take<uint64_t>(cap + 1);
    uint32_t *d_nl = A.take<uint32_t>(cap), *d_sl = A.take<uint32_t>(cap);
    SMI_HIP(hipMemcpyAsync(d_text, text, n_bytes, hipMemcpyHostToDevice, s));
    size_t n = 0;
    uint32_t fq_err = 0;
    SMI_RC(smi_fastq_index_device(ctx, d_text, n_bytes, d_line, 4 * cap + 8, d_ns, d_nl, d_ss, d_sl, d_qs, d_offs, cap, &n, &fq_err, s));
    out->n_records_in = n;
    out->fastq_errors = fq_err;
    if (fq_err) {
        set_error("smi_scanfastq_pass2_chunk: malformed FASTQ (see fastq_errors, SMI_FQ_*): htsjdk's FastqReader throws here");
        return SMI_ERR_INVALID;
    }
    if (n == 0) return SMI_OK;
    uint64_t total = 0;
    SMI_HIP(hipMemcpyAsync(&total, d_offs + n, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    uint8_t *d_reads = A.take<uint8_t>(bases_cap), *d_quals = A.take<uint8_t>(bases_cap);
    SMI_RC(smi_fastq_gather_device(ctx, d_text, d_ss, d_offs, n, d_reads, s));
    SMI_RC(smi_fastq_gather_device(ctx, d_text, d_qs, d_offs, n, d_quals, s));
    // ---- chimera splitter ----------------------------------------------------------------------------------------------------
    size_t m = n;
    const uint64_t *d_rec_offs = d_offs;
    smi_chimera_result *d_chim = nullptr;
    uint32_t *d_fsrc = nullptr;
    if (split) {
        uint32_t *d_planes = A.take<uint32_t>(planes_words);
        d_chim = A.take<smi_chimera_result>(cap);
        uint32_t *d_scr = A.take<uint32_t>((cap + 1023) / 1024 + 1);
        uint64_t *d_nfrag = A.take<uint64_t>(1);
        uint64_t *d_foffs = A.take<uint64_t>(3 * cap + 1);
        d_fsrc = A.take<uint32_t>(3 * cap);
        smi_chimera_config cc;
        SMI_RC(five ? smi_chimera_default_config_5p(&cc) : smi_chimera_default_config(&cc));
        SMI_RC(smi_pack_reads_device(ctx, d_reads, d_offs, n, total, d_planes, s));
        SMI_RC(smi_chimera_device(ctx, d_planes, d_offs, n, total, &cc, d_chim, s));
        SMI_RC(smi_split_offsets_device(ctx, d_chim, d_offs, n, d_scr, d_nfrag, d_foffs, d_fsrc, s));
        uint64_t nf = 0;
        std::vector<smi_chimera_result> h_chim(n);
        SMI_HIP(hipMemcpyAsync(&nf, d_nfrag, 8, hipMemcpyDeviceToHost, s));
        SMI_HIP(hipMemcpyAsync(h_chim.data(), d_chim, n * sizeof(smi_chimera_result), hipMemcpyDeviceToHost, s));
        SMI_HIP(hipStreamSynchronize(s));
        for (const auto &c : h_chim)
            if (c.flags & (SMI_CHIM_RANGE | SMI_CHIM_OVERFLOW)) {
                set_error("smi_scanfastq_pass2_chunk: a read outside what the splitter supports (SMI_CHIM_RANGE / SMI_CHIM_OVERFLOW)");
                return SMI_ERR_INVALID;
            }
        m = (size_t)nf;
        d_rec_offs = d_foffs;
    }
    out->n_records_out = m;
    // ---- scan + barcode -------------------------------------------------------------------------------------------------------
    uint32_t *d_ends = A.take<uint32_t>((size_t)SMI_ENDS_ROWS * 2 * m_cap);
    int32_t *d_len = A.take<int32_t>(m_cap);
    uint32_t *d_qsum = A.take<uint32_t>(m_cap);
    uint8_t *d_qtail = A.take<uint8_t>(m_cap * (size_t)SMI_END_BASES);
    smi_scan_result *d_scan = A.take<smi_scan_result>(m_cap);
    smi_bc_window *d_win = A.take<smi_bc_window>(m_cap);
    smi_bc_result *d_bc = A.take<smi_bc_result>(m_cap);
    smi_scan_config sc;
    SMI_RC(five ? smi_scan_default_config_5p(2, cfg->dont_search_polya, &sc) : smi_scan_default_config(2, &sc));
    SMI_RC(smi_pack_ends_device(ctx, d_reads, d_quals, d_rec_offs, m, five, d_ends, d_len, d_qtail, d_qsum, s));
    SMI_RC(smi_scan_device(ctx, d_ends, d_len, d_qtail, d_qsum, m, &sc, d_scan, d_win, s));
    SMI_RC(smi_bc_match_device(ctx, d_win, m, cfg->max_ed, five, d_bc, s));
    int32_t *d_rank = nullptr;
    if (cfg->rank_keys && cfg->n_ranks) {
        d_rank = A.take<int32_t>(m_cap);
        uint64_t *d_keys = A.take<uint64_t>(cfg->n_ranks);
        int32_t *d_vals = A.take<int32_t>(cfg->n_ranks);
        SMI_HIP(hipMemcpyAsync(d_keys, cfg->rank_keys, cfg->n_ranks * 8, hipMemcpyHostToDevice, s));
        SMI_HIP(hipMemcpyAsync(d_vals, cfg->rank_values, cfg->n_ranks * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_rank_lookup, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_bc, m, d_keys, d_vals, cfg->n_ranks,
                           d_rank);
        SMI_HIP(hipGetLastError());
    }
    // ---- records --------------------------------------------------------------------------------------------------------------
    uint8_t *d_passed = A.take<uint8_t>(out_cap), *d_failed = A.take<uint8_t>(out_cap);
    uint64_t *d_roff = A.take<uint64_t>(m_cap + 1);
    uint8_t *d_isp = A.take<uint8_t>(m_cap);
    smi_write_config wc{cfg->five_prime, cfg->trim_fastq};
    uint64_t totals[3] = {0, 0, 0};
    uint32_t werr = 0;
    SMI_RC(smi_fastq_write_device(ctx, d_text, d_line, d_reads, d_quals, d_rec_offs, split ? d_fsrc : nullptr, split ? d_chim : nullptr,
                                  d_scan, d_bc, d_rank, m, cfg->first_read_id, &wc, d_passed, out_cap, d_failed, out_cap, d_roff, d_isp,
                                  totals, &werr, s));
    for (int k = 0; k < 2; k++)  // pinned, grow-only: the download runs at link speed and nothing is zero-filled
        if (ctx->host_out_bytes[k] < totals[k]) {
            if (ctx->host_out[k]) SMI_HIP(hipHostFree(ctx->host_out[k]));
            ctx->host_out[k] = nullptr;
            ctx->host_out_bytes[k] = 0;
            const size_t want = totals[k] + totals[k] / 4 + 4096;
            SMI_HIP(hipHostMalloc((void **)&ctx->host_out[k], want, hipHostMallocDefault));
            ctx->host_out_bytes[k] = want;
        }
    if (totals[0]) SMI_HIP(hipMemcpyAsync(ctx->host_out[0], d_passed, totals[0], hipMemcpyDeviceToHost, s));
    if (totals[1]) SMI_HIP(hipMemcpyAsync(ctx->host_out[1], d_failed, totals[1], hipMemcpyDeviceToHost, s));
    if (cfg->want_results) {
        ctx->host_scan.resize(m);
        ctx->host_bc.resize(m);
        SMI_HIP(hipMemcpyAsync(ctx->host_scan.data(), d_scan, m * sizeof(smi_scan_result), hipMemcpyDeviceToHost, s));
        SMI_HIP(hipMemcpyAsync(ctx->host_bc.data(), d_bc, m * sizeof(smi_bc_result), hipMemcpyDeviceToHost, s));
        out->scan = ctx->host_scan.data();
        out->bc = ctx->host_bc.data();
    }
    SMI_HIP(hipStreamSynchronize(s));
    out->passed = ctx->host_out[0];
    out->failed = ctx->host_out[1];
    out->passed_bytes = totals[0];
    out->failed_bytes = totals[1];
    out->n_passed = totals[2];
    return SMI_OK;
}

extern "C" int smi_scanfastq_pass1_chunk(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, int five_prime, int dont_search_polya,
                                         uint32_t *d_hist, size_t *n_records, uint32_t *fastq_errors) {
    if (!ctx || !d_hist || !n_records || (!text && n_bytes)) {
        set_error("smi_scanfastq_pass1_chunk: null argument");
        return SMI_ERR_INVALID;
    }
    *n_records = 0;
    if (fastq_errors) *fastq_errors = 0;
    if (n_bytes == 0) return SMI_OK;
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t cap = count_lines(text, n_bytes) / 4 + 2;
    const size_t need = pad(n_bytes) + pad((4 * cap + 8) * 8) + 4 * pad(cap * 8) + pad((cap + 1) * 8) + 2 * pad(cap * 4) +
                        2 * pad(n_bytes) + pad((size_t)SMI_ENDS_ROWS * 2 * cap * 4) + 2 * pad(cap * 4) + pad(cap * (size_t)SMI_END_BASES) +
                        pad(cap * sizeof(smi_scan_result)) + pad(cap * sizeof(smi_bc_window)) + 4096;
    SMI_RC(ensure_arena(ctx, need));
    Arena A(ctx);
    uint8_t *d_text = A.take<uint8_t>(n_bytes);
    uint64_t *d_line = A.take<uint64_t>(4 * cap + 8);
    uint64_t *d_ns = A.take<uint64_t>(cap), *d_ss = A.take<uint64_t>(cap), *d_qs = A.take<uint64_t>(cap);
    uint64_t *d_offs = A.take<uint64_t>(cap + 1);
    uint32_t *d_nl = A.take<uint32_t>(cap), *d_sl = A.take<uint32_t>(cap);
    SMI_HIP(hipMemcpyAsync(d_text, text, n_bytes, hipMemcpyHostToDevice, s));
    size_t n = 0;
    uint32_t fq_err = 0;
    SMI_RC(smi_fastq_index_device(ctx, d_text, n_bytes, d_line, 4 * cap + 8, d_ns, d_nl, d_ss, d_sl, d_qs, d_offs, cap, &n, &fq_err, s));
    *n_records = n;
    if (fastq_errors) *fastq_errors = fq_err;
    if (fq_err) {
        set_error("smi_scanfastq_pass1_chunk: malformed FASTQ (SMI_FQ_*): htsjdk's FastqReader throws here");
        return SMI_ERR_INVALID;
    }
    if (n == 0) return SMI_OK;
    uint8_t *d_reads = A.take<uint8_t>(n_bytes), *d_quals = A.take<uint8_t>(n_bytes);
    SMI_RC(smi_fastq_gather_device(ctx, d_text, d_ss, d_offs, n, d_reads, s));
    SMI_RC(smi_fastq_gather_device(ctx, d_text, d_qs, d_offs, n, d_quals, s));
    uint32_t *d_ends = A.take<uint32_t>((size_t)SMI_ENDS_ROWS * 2 * cap);
    int32_t *d_len = A.take<int32_t>(cap);
    uint32_t *d_qsum = A.take<uint32_t>(cap);
    uint8_t *d_qtail = A.take<uint8_t>(cap * (size_t)SMI_END_BASES);
    smi_scan_result *d_scan = A.take<smi_scan_result>(cap);
    smi_bc_window *d_win = A.take<smi_bc_window>(cap);
    smi_scan_config sc;
    SMI_RC(five_prime ? smi_scan_default_config_5p(1, dont_search_polya, &sc) : smi_scan_default_config(1, &sc));
    SMI_RC(smi_pack_ends_device(ctx, d_reads, d_quals, d_offs, n, five_prime, d_ends, d_len, d_qtail, d_qsum, s));
    SMI_RC(smi_scan_device(ctx, d_ends, d_len, d_qtail, d_qsum, n, &sc, d_scan, d_win, s));
    SMI_RC(smi_hist_windows_device(ctx, d_win, d_scan, n, d_hist, s));
    SMI_HIP(hipStreamSynchronize(s));
    return SMI_OK;
}

// Page-locked host memory for the text a caller hands to the chunk workers (a JNI shim reads the file straight into it, e.g.
// through a direct ByteBuffer): the upload then runs at link speed instead of through the runtime's pageable staging.
extern "C" int smi_host_alloc(size_t bytes, void **out) {
    if (!out || !bytes) {
        set_error("smi_host_alloc: bad argument");
        return SMI_ERR_INVALID;
    }
    SMI_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return SMI_OK;
}

extern "C" int smi_host_free(void *p) {
    if (p) SMI_HIP(hipHostFree(p));
    return SMI_OK;
}
