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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <thread>
#include <unordered_map>
#include <vector>

#include "smi_internal.h"
#include "smi_umi_stage.h"

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

// a read the splitter discarded whole (MULTI_CHIMERIC_READS_DISCARDED) is never scanned by the reference (Parser.java:L92): its record goes
// to `failed` as it is and must not count as an assigned barcode anywhere (per-barcode counters, statistics, rank)
// scanfastq -e (smi_ctx_set_random_barcodes): the window bases of record i become a random sequence drawn from (seed, id of the record); N marks cleared.
// splitmix64 of seed + id: every record its own value, the same in every run.
__global__ void k_random_windows(smi_bc_window *__restrict__ win, size_t m, uint64_t seed, uint64_t first_id) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    smi_bc_window w = win[i];
    if (!(w.flags & SMI_WIN_VALID)) return;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (first_id + i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const int nb = (w.flags & SMI_WIN_5P) ? 25 : 24;
    w.bases = z & ((1ull << (2 * nb)) - 1ull);
    w.nmask = 0u;
    win[i] = w;
}

__global__ void k_drop_discarded(smi_bc_result *__restrict__ bc, const uint32_t *__restrict__ frag_src,
                                 const smi_chimera_result *__restrict__ chim, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (chim[frag_src[i] >> 2].flags & SMI_CHIM_MULTI) bc[i].found = 0;
}

// any read the splitter refused (SMI_CHIM_RANGE / SMI_CHIM_OVERFLOW)?  one word for the host instead of the chunk's results
__global__ void k_chim_refused(const smi_chimera_result *__restrict__ chim, size_t n, uint64_t *__restrict__ flag) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const bool bad = i < n && (chim[i].flags & (SMI_CHIM_RANGE | SMI_CHIM_OVERFLOW));
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr((unsigned long long *)flag, 1ull);
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

// The chunk's text goes to the front of the arena first and its lines are counted THERE: the buffers behind it are sized by the record
// count, and counting on the host (memchr over ~1.2 GB) took longer than the upload.  *d_text stays valid when the arena grows afterwards.
// text that lies in device memory already (K-INFLATE's output, a caller's device buffer) is read where it is
bool is_device_text(const uint8_t *text) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, text) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary host pointer: not an error of this call
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}
int upload_and_count(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, hipStream_t s, size_t *n_lines) {
    // the count is the first half of the record index (its one sweep over the text): launch_fastq_index continues from what it left
    if (is_device_text(text)) return launch_fastq_sweep(ctx, text, n_bytes, n_lines, s);
    if (int rc = ensure_arena(ctx, pad(n_bytes) + 4096)) return rc;
    SMI_HIP(hipMemcpyAsync(ctx->arena, text, n_bytes, hipMemcpyDefault, s));
    return launch_fastq_sweep(ctx, static_cast<const uint8_t *>(ctx->arena), n_bytes, n_lines, s);
}
// grow the arena to `bytes`, keeping its first keep_bytes (the uploaded text)
int grow_arena_keep(smi_ctx *ctx, size_t bytes, size_t keep_bytes, hipStream_t s) {
    if (ctx->arena_bytes >= bytes) return SMI_OK;
    const size_t want = bytes + bytes / 4;
    void *fresh = nullptr;
    SMI_HIP(hipMalloc(&fresh, want));
    if (keep_bytes) SMI_HIP(hipMemcpyAsync(fresh, ctx->arena, keep_bytes, hipMemcpyDeviceToDevice, s));
    SMI_HIP(hipStreamSynchronize(s));
    if (keep_bytes && ctx->fq_swept_text == ctx->arena) ctx->fq_swept_text = static_cast<const uint8_t *>(fresh);  // the swept text moved with the arena
    SMI_HIP(hipFree(ctx->arena));
    ctx->arena = fresh;
    ctx->arena_bytes = want;
    return SMI_OK;
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
    const bool timing = std::getenv("SMI_WK_TIMING") != nullptr;  // the phases of this call (host clock, one line on stderr)
    auto t_prev = std::chrono::steady_clock::now();
    double t_ph[6] = {0, 0, 0, 0, 0, 0};
    auto phase = [&](int k) {
        const auto now = std::chrono::steady_clock::now();
        t_ph[k] += std::chrono::duration<double, std::milli>(now - t_prev).count();
        t_prev = now;
    };
    size_t n_lines = 0;
    if (int rc = upload_and_count(ctx, text, n_bytes, s, &n_lines)) return rc;
    phase(0);
    const size_t cap = n_lines / 4 + 2;  // records
    // worst-case sizes before anything is known about the chunk: bases + qualities <= text, fragments <= 3 per record
    const size_t m_cap = split ? 3 * cap : cap;
    const size_t bases_cap = n_bytes;
    const size_t planes_words = split ? smi_read_planes_words(bases_cap, cap) : 0;
    const size_t out_cap = 2 * bases_cap + n_bytes + 320 * m_cap + 64;
    size_t need = pad(n_bytes) + pad((4 * cap + 8) * 8) + 4 * pad(cap * 8) + pad((cap + 1) * 8) + 2 * pad(cap * 4) + 2 * pad(m_cap * 8) +
                  pad(planes_words * 4) + pad(cap * sizeof(smi_chimera_result)) + pad(((cap + 1023) / 1024 + 1) * 4) + pad(8) +
                  pad((3 * cap + 1) * 8) + pad(3 * cap * 4) + pad((size_t)SMI_ENDS_ROWS * 2 * m_cap * 4) + 2 * pad(m_cap * 4) +
                  pad(m_cap * (size_t)SMI_END_BASES) + pad(m_cap * sizeof(smi_scan_result)) + pad(m_cap * sizeof(smi_bc_window)) +
                  pad(m_cap * sizeof(smi_bc_result)) + pad(m_cap * 4) + pad(cfg->n_ranks * 8) + pad(cfg->n_ranks * 4) +
                  2 * pad(out_cap) + pad((m_cap + 1) * 8) + pad(m_cap) + 4096;
    const bool in_place = is_device_text(text);
    SMI_RC(grow_arena_keep(ctx, need, in_place ? 0 : n_bytes, s));
    Arena A(ctx);
    const uint8_t *d_text = in_place ? text : A.take<uint8_t>(n_bytes);  // already there
    uint64_t *d_line = A.take<uint64_t>(4 * cap + 8);
    uint64_t *d_ns = A.take<uint64_t>(cap), *d_ss = A.take<uint64_t>(cap), *d_qs = A.take<uint64_t>(cap);
    uint64_t *d_offs = A.take<uint64_t>(cap + 1);
    uint32_t *d_nl = A.take<uint32_t>(cap), *d_sl = A.take<uint32_t>(cap);
    size_t n = 0;
    uint32_t fq_err = 0;
    uint64_t total = 0;  // all bases of the chunk: comes back on the index's own wait
    SMI_RC(launch_fastq_index(ctx, d_text, n_bytes, d_line, 4 * cap + 8, d_ns, d_nl, d_ss, d_sl, d_qs, d_offs, cap, &n, &fq_err, s, &total));
    phase(1);
    out->n_records_in = n;
    out->fastq_errors = fq_err;
    if (fq_err) {
        set_error("smi_scanfastq_pass2_chunk: malformed FASTQ (see fastq_errors, SMI_FQ_*): htsjdk's FastqReader throws here");
        return SMI_ERR_INVALID;
    }
    if (n == 0) return SMI_OK;
    // bases and qualities stay where the text has them: every consumer below takes per-record text positions (no gathered copies)
    uint64_t *d_bstart = A.take<uint64_t>(m_cap), *d_qstart = A.take<uint64_t>(m_cap);
    // ---- chimera splitter ----------------------------------------------------------------------------------------------------
    size_t m = n;
    const uint64_t *d_rec_offs = d_offs;
    smi_chimera_result *d_chim = nullptr;
    uint32_t *d_fsrc = nullptr;
    std::vector<smi_chimera_result> h_chim;
    if (split) {
        uint32_t *d_planes = A.take<uint32_t>(planes_words);
        d_chim = A.take<smi_chimera_result>(cap);
        uint32_t *d_scr = A.take<uint32_t>((cap + 1023) / 1024 + 1);
        uint64_t *d_nfrag = A.take<uint64_t>(2);  // [0] fragments, [1] != 0: a read the splitter refused
        uint64_t *d_foffs = A.take<uint64_t>(3 * cap + 1);
        d_fsrc = A.take<uint32_t>(3 * cap);
        smi_chimera_config cc;
        SMI_RC(worker_chimera_config(ctx, five, &cc));
        SMI_RC(smi_pack_reads_text_device(ctx, d_text, d_ss, d_offs, n, total, d_planes, s));
        SMI_RC(smi_chimera_device(ctx, d_planes, d_offs, n, total, &cc, d_chim, s));
        SMI_RC(smi_split_offsets_device(ctx, d_chim, d_offs, n, d_scr, d_nfrag, d_foffs, d_fsrc, s));
        SMI_HIP(hipMemsetAsync(d_nfrag + 1, 0, 8, s));
        hipLaunchKernelGGL(k_chim_refused, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_chim, n, d_nfrag + 1);
        SMI_HIP(hipGetLastError());
        uint64_t nf[2] = {0, 0};
        uint64_t *pw = static_cast<uint64_t *>(pin_words(ctx));
        SMI_HIP(hipMemcpyAsync(pw ? pw : nf, d_nfrag, 16, hipMemcpyDeviceToHost, s));
        if (cfg->want_results) {  // the statistics count the split decisions; nothing else on the host reads them
            h_chim.resize(n);
            SMI_HIP(hipMemcpyAsync(h_chim.data(), d_chim, n * sizeof(smi_chimera_result), hipMemcpyDeviceToHost, s));
        }
        SMI_HIP(hipStreamSynchronize(s));
        if (pw) std::memcpy(nf, pw, 16);
        if (nf[1]) {
            set_error("smi_scanfastq_pass2_chunk: a read outside what the splitter supports (SMI_CHIM_RANGE: longer than the plane offsets can address)");
            return SMI_ERR_INVALID;
        }
        m = (size_t)nf[0];
        d_rec_offs = d_foffs;
    }
    phase(2);
    out->n_records_out = m;
    // ---- scan + barcode -------------------------------------------------------------------------------------------------------
    uint32_t *d_ends = A.take<uint32_t>((size_t)SMI_ENDS_ROWS * 2 * m_cap);
    int32_t *d_len = A.take<int32_t>(m_cap);
    smi_scan_result *d_scan = A.take<smi_scan_result>(m_cap);
    smi_bc_window *d_win = A.take<smi_bc_window>(m_cap);
    smi_bc_result *d_bc = A.take<smi_bc_result>(m_cap);
    smi_scan_config sc;
    SMI_RC(worker_scan_config(ctx, 2, five, cfg->dont_search_polya, &sc));
    // no qualities here: the quality filter (pass1_ok) belongs to pass 1 (UsedCellBCListGenerator.java:L198-202)
    SMI_RC(smi_frag_text_starts_device(ctx, d_ss, d_qs, d_offs, d_rec_offs, split ? d_fsrc : nullptr, m, d_bstart, d_qstart, s));
    SMI_RC(smi_pack_ends_text_device(ctx, d_text, d_bstart, d_rec_offs, m, d_ends, d_len, s));
    SMI_RC(smi_scan_device(ctx, d_ends, d_len, nullptr, nullptr, m, &sc, d_scan, d_win, s));
    if (ctx->random_bc_seed && m) {  // scanfastq -e: the matcher sees random windows (ids: the chunk's first read id onwards)
        hipLaunchKernelGGL(k_random_windows, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_win, m, ctx->random_bc_seed, (uint64_t)cfg->first_read_id);
        SMI_HIP(hipGetLastError());
    }
    SMI_RC(smi_bc_match_device(ctx, d_win, m, cfg->max_ed, five, d_bc, s));
    if (split && m) {
        hipLaunchKernelGGL(k_drop_discarded, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_bc, d_fsrc, d_chim, m);
        SMI_HIP(hipGetLastError());
    }
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
    SMI_RC(smi_fastq_write_text_device(ctx, d_text, d_line, d_bstart, d_qstart, d_rec_offs, split ? d_fsrc : nullptr,
                                       split ? d_chim : nullptr, d_scan, d_bc, d_rank, m, cfg->first_read_id, &wc, d_passed, out_cap, d_failed,
                                       out_cap, d_roff, d_isp, totals, &werr, s));
    phase(3);
    out->passed_text_bytes = totals[0];
    out->failed_text_bytes = totals[1];
    const uint8_t *d_src[2] = {d_passed, d_failed};
    uint64_t down[2] = {totals[0], totals[1]};
    if (cfg->compress) {
        // --compress: the two texts become one gzip member each where they lie; only the members cross the link
        uint8_t *d_z[2] = {nullptr, nullptr};
        uint64_t *d_zt = nullptr;
        SMI_RC(deflate_pair(ctx, d_passed, totals[0], d_failed, totals[1], &d_z[0], &d_z[1], &d_zt, s));
        uint64_t zt[4] = {0, 0, 0, 0};
        SMI_HIP(hipMemcpyAsync(zt, d_zt, sizeof zt, hipMemcpyDeviceToHost, s));
        SMI_HIP(hipStreamSynchronize(s));
        if (zt[1] | zt[3]) {
            set_error("smi_scanfastq_pass2_chunk: K-DEFLATE reported an error (a block outgrew its slot)");
            return SMI_ERR_INVALID;
        }
        d_src[0] = d_z[0];
        d_src[1] = d_z[1];
        down[0] = zt[0];
        down[1] = zt[2];
    }
    if (cfg->device_output) {
        // the caller reads the two texts (or gzip members) where K-WRITE / K-DEFLATE left them; the totals' wait above was the call's last
        if (cfg->want_results) {
            set_error("smi_scanfastq_pass2_chunk: device_output and want_results are two different callers (the per-record results are a host structure)");
            return SMI_ERR_INVALID;
        }
        out->passed = d_src[0];
        out->failed = d_src[1];
        out->passed_bytes = down[0];
        out->failed_bytes = down[1];
        out->n_passed = totals[2];
        if (timing)
            std::fprintf(stderr, "pass2_chunk: sweep+count %.3f  index %.3f  split %.3f  ends..write %.3f ms (%zu records)\n", t_ph[0], t_ph[1], t_ph[2], t_ph[3], n);
        return SMI_OK;
    }
    for (int k = 0; k < 2; k++)  // pinned, grow-only: the download runs at link speed and nothing is zero-filled
        if (ctx->host_out_bytes[k] < down[k]) {
            if (ctx->host_out[k]) SMI_HIP(hipHostFree(ctx->host_out[k]));
            ctx->host_out[k] = nullptr;
            ctx->host_out_bytes[k] = 0;
            const size_t want = down[k] + down[k] / 4 + 4096;
            SMI_HIP(hipHostMalloc((void **)&ctx->host_out[k], want, hipHostMallocDefault));
            ctx->host_out_bytes[k] = want;
        }
    if (down[0]) SMI_HIP(hipMemcpyAsync(ctx->host_out[0], d_src[0], down[0], hipMemcpyDeviceToHost, s));
    if (down[1]) SMI_HIP(hipMemcpyAsync(ctx->host_out[1], d_src[1], down[1], hipMemcpyDeviceToHost, s));
    if (cfg->want_results) {
        ctx->host_scan.resize(m);
        ctx->host_bc.resize(m);
        SMI_HIP(hipMemcpyAsync(ctx->host_scan.data(), d_scan, m * sizeof(smi_scan_result), hipMemcpyDeviceToHost, s));
        SMI_HIP(hipMemcpyAsync(ctx->host_bc.data(), d_bc, m * sizeof(smi_bc_result), hipMemcpyDeviceToHost, s));
        out->scan = ctx->host_scan.data();
        out->bc = ctx->host_bc.data();
    }
    std::vector<uint64_t> h_foffs;
    std::vector<uint32_t> h_fsrc;
    if (cfg->want_results) {  // what the statistics need beside scan / bc: where each output record came from
        h_foffs.resize(m + 1);
        SMI_HIP(hipMemcpyAsync(h_foffs.data(), d_rec_offs, (m + 1) * 8, hipMemcpyDeviceToHost, s));
        if (split) {
            h_fsrc.resize(m);
            SMI_HIP(hipMemcpyAsync(h_fsrc.data(), d_fsrc, m * 4, hipMemcpyDeviceToHost, s));
        }
    }
    SMI_HIP(hipStreamSynchronize(s));
    if (cfg->want_results) {
        smi_pass2_decisions dec;
        std::memset(&dec, 0, sizeof dec);
        dec.n_records_in = n;
        dec.n_records_out = m;
        dec.chim = split ? h_chim.data() : nullptr;
        dec.frag_offsets = h_foffs.data();
        dec.frag_src = split ? h_fsrc.data() : nullptr;
        dec.scan = ctx->host_scan.data();
        dec.bc = ctx->host_bc.data();
        std::memset(&ctx->host_stats, 0, sizeof ctx->host_stats);  // ReadFlags.addForCounting over the chunk
        SMI_RC(smi_scan_stats_add(&ctx->host_stats, &dec));
        out->stats = &ctx->host_stats;
    }
    out->passed = ctx->host_out[0];
    out->failed = ctx->host_out[1];
    out->passed_bytes = down[0];
    out->failed_bytes = down[1];
    out->n_passed = totals[2];
    return SMI_OK;
}

namespace {
// pass 1 of one chunk; the barcodes of the reads that pass the filter go into the dense histogram of the loaded list (d_hist) or, without a
// list of possible barcodes, as keys onto a list (d_keys / cap / d_count)
int pass1_chunk_core(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, int five_prime, int dont_search_polya, uint32_t *d_hist, uint64_t *d_keys,
                     size_t cap_keys, uint64_t *d_count, size_t *n_records, uint32_t *fastq_errors);
}  // namespace

extern "C" int smi_scanfastq_pass1_chunk(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, int five_prime, int dont_search_polya,
                                         uint32_t *d_hist, size_t *n_records, uint32_t *fastq_errors) {
    if (!ctx || !d_hist || !n_records || (!text && n_bytes)) {
        set_error("smi_scanfastq_pass1_chunk: null argument");
        return SMI_ERR_INVALID;
    }
    return pass1_chunk_core(ctx, text, n_bytes, five_prime, dont_search_polya, d_hist, nullptr, 0, nullptr, n_records, fastq_errors);
}

extern "C" int smi_scanfastq_pass1_chunk_keys(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, int five_prime, int dont_search_polya,
                                              uint64_t *d_keys, size_t cap_keys, uint64_t *d_count, size_t *n_records, uint32_t *fastq_errors) {
    if (!ctx || !d_keys || !d_count || !n_records || (!text && n_bytes)) {
        set_error("smi_scanfastq_pass1_chunk_keys: null argument");
        return SMI_ERR_INVALID;
    }
    return pass1_chunk_core(ctx, text, n_bytes, five_prime, dont_search_polya, nullptr, d_keys, cap_keys, d_count, n_records, fastq_errors);
}

namespace {
int pass1_chunk_core(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, int five_prime, int dont_search_polya, uint32_t *d_hist, uint64_t *d_keys,
                     size_t cap_keys, uint64_t *d_count, size_t *n_records, uint32_t *fastq_errors) {
    *n_records = 0;
    if (fastq_errors) *fastq_errors = 0;
    if (n_bytes == 0) return SMI_OK;
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    size_t n_lines = 0;
    if (int rc = upload_and_count(ctx, text, n_bytes, s, &n_lines)) return rc;
    const size_t cap = n_lines / 4 + 2;
    const size_t need = pad(n_bytes) + pad((4 * cap + 8) * 8) + 4 * pad(cap * 8) + pad((cap + 1) * 8) + 2 * pad(cap * 4) +
                        2 * pad(n_bytes) + pad((size_t)SMI_ENDS_ROWS * 2 * cap * 4) + 2 * pad(cap * 4) + pad(cap * (size_t)SMI_END_BASES) +
                        pad(cap * sizeof(smi_scan_result)) + pad(cap * sizeof(smi_bc_window)) + 4096;
    const bool in_place = is_device_text(text);
    SMI_RC(grow_arena_keep(ctx, need, in_place ? 0 : n_bytes, s));
    Arena A(ctx);
    const uint8_t *d_text = in_place ? text : A.take<uint8_t>(n_bytes);  // already there
    uint64_t *d_line = A.take<uint64_t>(4 * cap + 8);
    uint64_t *d_ns = A.take<uint64_t>(cap), *d_ss = A.take<uint64_t>(cap), *d_qs = A.take<uint64_t>(cap);
    uint64_t *d_offs = A.take<uint64_t>(cap + 1);
    uint32_t *d_nl = A.take<uint32_t>(cap), *d_sl = A.take<uint32_t>(cap);
    size_t n = 0;
    uint32_t fq_err = 0;
    SMI_RC(launch_fastq_index(ctx, d_text, n_bytes, d_line, 4 * cap + 8, d_ns, d_nl, d_ss, d_sl, d_qs, d_offs, cap, &n, &fq_err, s));
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
    SMI_RC(worker_scan_config(ctx, 1, five_prime, dont_search_polya, &sc));
    SMI_RC(smi_pack_ends_device(ctx, d_reads, d_quals, d_offs, n, five_prime, d_ends, d_len, d_qtail, d_qsum, s));
    SMI_RC(smi_scan_device(ctx, d_ends, d_len, d_qtail, d_qsum, n, &sc, d_scan, d_win, s));
    if (d_hist)
        SMI_RC(smi_hist_windows_device(ctx, d_win, d_scan, n, d_hist, s));
    else
        SMI_RC(smi_pass1_keys_device(ctx, d_win, d_scan, n, d_keys, cap_keys, d_count, s));
    SMI_HIP(hipStreamSynchronize(s));
    return SMI_OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// The packed boundary (include/sicelore_mi.h, "The packed boundary of scanfastq"): planes + offsets up, decisions down.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
int pass2_packed_core(smi_ctx *ctx, const smi_packed_reads *pk, const uint64_t *offsets, size_t n, const smi_pass2_config *cfg,
                      smi_pass2_decisions *out);
}
extern "C" int smi_scanfastq_pass2_packed(smi_ctx *ctx, const uint32_t *planes, const uint64_t *offsets, size_t n, const smi_pass2_config *cfg,
                                          smi_pass2_decisions *out) {
    if (!ctx || !cfg || !out || (n && (!planes || !offsets))) {
        set_error("smi_scanfastq_pass2_packed: null argument");
        return SMI_ERR_INVALID;
    }
    smi_packed_reads pk;
    std::memset(&pk, 0, offsetof(smi_packed_reads, seg_host_word));
    if (n) {
        pk.planes = planes;
        pk.stride = smi_read_planes_words(offsets[n], n) / 4;
        pk.n_seg = 1;
        pk.total_words = pk.stride;
        pk.seg_host_word[0] = pk.seg_dev_word[0] = 0;
        pk.seg_words[0] = pk.stride;
    }
    return pass2_packed_core(ctx, &pk, offsets, n, cfg, out);
}
extern "C" int smi_scanfastq_pass2_packed_seg(smi_ctx *ctx, const smi_packed_reads *packed, const uint64_t *offsets, size_t n, const smi_pass2_config *cfg,
                                              smi_pass2_decisions *out) {
    if (!ctx || !cfg || !out || !packed || (n && (!packed->planes || !offsets || packed->n_seg < 1 || packed->n_seg > SMI_PACKED_MAX_SEGMENTS))) {
        set_error("smi_scanfastq_pass2_packed_seg: bad argument");
        return SMI_ERR_INVALID;
    }
    return pass2_packed_core(ctx, packed, offsets, n, cfg, out);
}

namespace {
// planes of a chunk into the arena, compact: the segments back to back in each of the four planes (+ pstart when the packer gave one)
int upload_planes(smi_ctx *ctx, const smi_packed_reads *pk, size_t n, uint32_t *d_planes, uint32_t *d_pstart, hipStream_t s) {
    for (int c = 0; c < 4; c++)
        for (int k = 0; k < pk->n_seg; k++)
            SMI_HIP(hipMemcpyAsync(d_planes + (size_t)c * pk->total_words + pk->seg_dev_word[k], pk->planes + (size_t)c * pk->stride + pk->seg_host_word[k],
                                   (size_t)pk->seg_words[k] * 4, hipMemcpyHostToDevice, s));
    if (pk->pstart) SMI_HIP(hipMemcpyAsync(d_pstart, pk->pstart, n * 4, hipMemcpyHostToDevice, s));
    return SMI_OK;
}

int pass2_packed_core(smi_ctx *ctx, const smi_packed_reads *pk, const uint64_t *offsets, size_t n, const smi_pass2_config *cfg,
                      smi_pass2_decisions *out) {
    std::memset(out, 0, sizeof *out);
    if (n == 0) return SMI_OK;
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const bool five = cfg->five_prime != 0;
    const bool split = cfg->split_chimeras && !(five && cfg->dont_search_polya);  // Parser.java:L176
    const uint64_t total = offsets[n];
    const size_t pstride = pk->total_words;  // words per plane on the device
    const size_t planes_words = 4 * pstride;
    const size_t m_cap = split ? 3 * n : n;
    const size_t need = pad(planes_words * 4) + pad(n * 4) + pad((n + 1) * 8) + pad(n * sizeof(smi_chimera_result)) + pad(((n + 1023) / 1024 + 1) * 4) + pad(8) +
                        pad((3 * n + 1) * 8) + pad(3 * n * 4) + pad((size_t)SMI_ENDS_ROWS * 2 * m_cap * 4) + pad(m_cap * 4) +
                        pad(m_cap * sizeof(smi_scan_result)) + pad(m_cap * sizeof(smi_bc_window)) + pad(m_cap * sizeof(smi_bc_result)) +
                        pad(m_cap * 4) + pad(cfg->n_ranks * 8) + pad(cfg->n_ranks * 4) + 4096;
    SMI_RC(ensure_arena(ctx, need));
    Arena A(ctx);
    uint32_t *d_planes = A.take<uint32_t>(planes_words);
    uint32_t *d_pstart = A.take<uint32_t>(n);
    uint64_t *d_offs = A.take<uint64_t>(n + 1);
    SMI_RC(upload_planes(ctx, pk, n, d_planes, d_pstart, s));
    if (!pk->pstart) d_pstart = nullptr;
    SMI_HIP(hipMemcpyAsync(d_offs, offsets, (n + 1) * 8, hipMemcpyHostToDevice, s));
    size_t m = n;
    const uint64_t *d_rec_offs = d_offs;
    smi_chimera_result *d_chim = nullptr;
    uint32_t *d_fsrc = nullptr;
    uint64_t *d_foffs = nullptr;
    if (split) {
        d_chim = A.take<smi_chimera_result>(n);
        uint32_t *d_scr = A.take<uint32_t>((n + 1023) / 1024 + 1);
        uint64_t *d_nfrag = A.take<uint64_t>(1);
        d_foffs = A.take<uint64_t>(3 * n + 1);
        d_fsrc = A.take<uint32_t>(3 * n);
        smi_chimera_config cc;
        SMI_RC(worker_chimera_config(ctx, five, &cc));
        SMI_RC(launch_chimera(ctx, d_planes, d_offs, n, total, &cc, d_chim, s, d_pstart, pstride));
        SMI_RC(smi_split_offsets_device(ctx, d_chim, d_offs, n, d_scr, d_nfrag, d_foffs, d_fsrc, s));
        SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_CHIM, n * sizeof(smi_chimera_result)));
        smi_chimera_result *h_chim = static_cast<smi_chimera_result *>(ctx->host_buf[smi_ctx::HB_CHIM]);
        uint64_t nf = 0;
        SMI_HIP(hipMemcpyAsync(&nf, d_nfrag, 8, hipMemcpyDeviceToHost, s));
        SMI_HIP(hipMemcpyAsync(h_chim, d_chim, n * sizeof(smi_chimera_result), hipMemcpyDeviceToHost, s));
        SMI_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < n; i++)
            if (h_chim[i].flags & (SMI_CHIM_RANGE | SMI_CHIM_OVERFLOW)) {
                set_error("smi_scanfastq_pass2_packed: a read outside what the splitter supports (SMI_CHIM_RANGE: longer than the plane offsets can address)");
                return SMI_ERR_INVALID;
            }
        m = (size_t)nf;
        d_rec_offs = d_foffs;
        out->chim = h_chim;
    }
    out->n_records_in = n;
    out->n_records_out = m;
    uint32_t *d_ends = A.take<uint32_t>((size_t)SMI_ENDS_ROWS * 2 * m_cap);
    int32_t *d_len = A.take<int32_t>(m_cap);
    smi_scan_result *d_scan = A.take<smi_scan_result>(m_cap);
    smi_bc_window *d_win = A.take<smi_bc_window>(m_cap);
    smi_bc_result *d_bc = A.take<smi_bc_result>(m_cap);
    smi_scan_config sc;
    SMI_RC(worker_scan_config(ctx, 2, five, cfg->dont_search_polya, &sc));
    SMI_RC(launch_ends_from_planes(ctx, d_planes, pstride, d_offs, d_rec_offs, split ? d_fsrc : nullptr, m, d_ends, d_len, s, d_pstart));
    SMI_RC(smi_scan_device(ctx, d_ends, d_len, nullptr, nullptr, m, &sc, d_scan, d_win, s));
    if (ctx->random_bc_seed && m) {  // scanfastq -e: the matcher sees random windows (ids: the chunk's first read id onwards)
        hipLaunchKernelGGL(k_random_windows, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_win, m, ctx->random_bc_seed, (uint64_t)cfg->first_read_id);
        SMI_HIP(hipGetLastError());
    }
    SMI_RC(smi_bc_match_device(ctx, d_win, m, cfg->max_ed, five, d_bc, s));
    if (split && m) {
        hipLaunchKernelGGL(k_drop_discarded, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_bc, d_fsrc, d_chim, m);
        SMI_HIP(hipGetLastError());
    }
    int32_t *d_rank = nullptr;
    if (cfg->rank_keys && cfg->n_ranks) {
        d_rank = A.take<int32_t>(m_cap);
        uint64_t *d_keys = A.take<uint64_t>(cfg->n_ranks);
        int32_t *d_vals = A.take<int32_t>(cfg->n_ranks);
        SMI_HIP(hipMemcpyAsync(d_keys, cfg->rank_keys, cfg->n_ranks * 8, hipMemcpyHostToDevice, s));
        SMI_HIP(hipMemcpyAsync(d_vals, cfg->rank_values, cfg->n_ranks * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_rank_lookup, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_bc, m, d_keys, d_vals, cfg->n_ranks, d_rank);
        SMI_HIP(hipGetLastError());
    }
    // decisions down: ~80 bytes per record
    SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_SCAN, m * sizeof(smi_scan_result)));
    SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_BC, m * sizeof(smi_bc_result)));
    SMI_HIP(hipMemcpyAsync(ctx->host_buf[smi_ctx::HB_SCAN], d_scan, m * sizeof(smi_scan_result), hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(ctx->host_buf[smi_ctx::HB_BC], d_bc, m * sizeof(smi_bc_result), hipMemcpyDeviceToHost, s));
    if (d_rank) {
        SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_RANK, m * 4));
        SMI_HIP(hipMemcpyAsync(ctx->host_buf[smi_ctx::HB_RANK], d_rank, m * 4, hipMemcpyDeviceToHost, s));
        out->rank = static_cast<const int32_t *>(ctx->host_buf[smi_ctx::HB_RANK]);
    }
    if (split) {
        SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_FOFFS, (m + 1) * 8));
        SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_FSRC, m * 4));
        SMI_HIP(hipMemcpyAsync(ctx->host_buf[smi_ctx::HB_FOFFS], d_foffs, (m + 1) * 8, hipMemcpyDeviceToHost, s));
        SMI_HIP(hipMemcpyAsync(ctx->host_buf[smi_ctx::HB_FSRC], d_fsrc, m * 4, hipMemcpyDeviceToHost, s));
        out->frag_offsets = static_cast<const uint64_t *>(ctx->host_buf[smi_ctx::HB_FOFFS]);
        out->frag_src = static_cast<const uint32_t *>(ctx->host_buf[smi_ctx::HB_FSRC]);
    } else
        out->frag_offsets = offsets;  // the caller's array: output records = input records
    SMI_HIP(hipStreamSynchronize(s));
    out->scan = static_cast<const smi_scan_result *>(ctx->host_buf[smi_ctx::HB_SCAN]);
    out->bc = static_cast<const smi_bc_result *>(ctx->host_buf[smi_ctx::HB_BC]);
    return SMI_OK;
}
}  // namespace

namespace {
// host index + planes of a chunk (one pass over the text, smi_fastq_index_pack_host) into the context's page-locked buffers
int index_and_pack(smi_ctx *ctx, const char *who, const uint8_t *text, size_t n_bytes, int n_threads, size_t *n_out, uint32_t *fq_err,
                   smi_packed_reads *pk, double *ms) {
    const auto t0 = std::chrono::steady_clock::now();
    // a record has at least four line ends + '@' + '+' = 6 bytes (an empty read); the buffers are first sized for records of 64 bytes
    // or more and only a chunk of tinier ones pays for the worst case
    const size_t cap_worst = n_bytes / 6 + 2;
    size_t cap_try = std::min(cap_worst, n_bytes / 64 + 1024);
    size_t planes_words = smi_packed_planes_words(n_bytes, n_threads);
    for (;;) {
        SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_RECS, cap_try * sizeof(smi_fastq_record)));
        SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_OFFS, (cap_try + 1) * 8));
        SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_PSTART, cap_try * 4));
        SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_PLANES, planes_words * 4));
        const int rc = smi_fastq_index_pack_host(text, n_bytes, static_cast<smi_fastq_record *>(ctx->host_buf[smi_ctx::HB_RECS]),
                                                 static_cast<uint64_t *>(ctx->host_buf[smi_ctx::HB_OFFS]),
                                                 static_cast<uint32_t *>(ctx->host_buf[smi_ctx::HB_PSTART]), cap_try,
                                                 static_cast<uint32_t *>(ctx->host_buf[smi_ctx::HB_PLANES]), planes_words, pk, n_out, fq_err, n_threads);
        if (rc == SMI_OK) break;
        if (cap_try >= cap_worst) return rc;
        cap_try = cap_worst;  // tiny records: once more with the worst-case capacities
        planes_words = std::max(planes_words, 4 * read_planes_stride(n_bytes / 2, cap_worst));
    }
    if (*fq_err) {
        set_error(std::string(who) + ": malformed FASTQ (see fastq_errors, SMI_FQ_*): htsjdk's FastqReader throws here");
        return SMI_ERR_INVALID;
    }
    *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SMI_OK;
}
}  // namespace

extern "C" int smi_scanfastq_pass2_chunk_packed(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, const smi_pass2_config *cfg, int n_threads,
                                                smi_pass2_output *out) {
    if (!ctx || !cfg || !out || (!text && n_bytes)) {
        set_error("smi_scanfastq_pass2_chunk_packed: null argument");
        return SMI_ERR_INVALID;
    }
    std::memset(out, 0, sizeof *out);
    if (cfg->compress || cfg->device_output) {
        set_error("smi_scanfastq_pass2_chunk_packed: compress and device_output are options of the text worker (the packed worker's records are written by the host)");
        return SMI_ERR_INVALID;
    }
    if (n_bytes == 0) return SMI_OK;
    const bool timing = std::getenv("SMI_PK_TIMING") != nullptr;  // the stages of this call on stderr
    double ms_index = 0;
    size_t n = 0;
    uint32_t fq_err = 0;
    smi_packed_reads pk;
    const int rc_ix = index_and_pack(ctx, "smi_scanfastq_pass2_chunk_packed", text, n_bytes, n_threads, &n, &fq_err, &pk, &ms_index);
    out->n_records_in = n;
    out->fastq_errors = fq_err;
    if (rc_ix != SMI_OK) return rc_ix;
    if (n == 0) return SMI_OK;
    const smi_fastq_record *recs = static_cast<const smi_fastq_record *>(ctx->host_buf[smi_ctx::HB_RECS]);
    const uint64_t *offs = static_cast<const uint64_t *>(ctx->host_buf[smi_ctx::HB_OFFS]);
    const auto t0 = std::chrono::steady_clock::now();
    smi_pass2_decisions dec;
    SMI_RC(pass2_packed_core(ctx, &pk, offs, n, cfg, &dec));
    const auto t1 = std::chrono::steady_clock::now();
    out->n_records_out = dec.n_records_out;
    // the records: never longer than the text they are cut from + a name suffix each
    // (bases and qualities once; the name token and the '+' line once per fragment, at most three fragments per record)
    const size_t out_cap = n_bytes + 320 * dec.n_records_out + (dec.frag_src ? 2 * (n_bytes - std::min<size_t>(n_bytes, 2 * (size_t)offs[n])) : 0) + 64;
    for (int k = 0; k < 2; k++)
        if (ctx->host_out_bytes[k] < out_cap) {
            if (ctx->host_out[k]) SMI_HIP(hipHostFree(ctx->host_out[k]));
            ctx->host_out[k] = nullptr;
            ctx->host_out_bytes[k] = 0;
            const size_t want = out_cap + out_cap / 4 + 4096;
            SMI_HIP(hipHostMalloc((void **)&ctx->host_out[k], want, hipHostMallocDefault));
            ctx->host_out_bytes[k] = want;
        }
    smi_write_config wc{cfg->five_prime, cfg->trim_fastq};
    uint64_t totals[3] = {0, 0, 0};
    uint32_t werr = 0;
    const int rc_w = smi_fastq_write_host(text, recs, offs, &dec, cfg->first_read_id, &wc, ctx->host_out[0], ctx->host_out_bytes[0], ctx->host_out[1],
                                          ctx->host_out_bytes[1], totals, &werr, n_threads);
    if (rc_w != SMI_OK) {
        if (werr & SMI_WR_QUAL_NEWLINE) {  // a line end inside a quality line that the one-pass index stepped over: what the exact index reports
            out->fastq_errors = SMI_FQ_LENGTH_MISMATCH;
            set_error("smi_scanfastq_pass2_chunk_packed: malformed FASTQ (a line end inside a quality string): htsjdk's FastqReader throws here");
        }
        return rc_w;
    }
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "smi_scanfastq_pass2_chunk_packed: %zu reads  index+pack %.2f (%d segments)  device %.2f  write %.2f ms (%d threads)\n", n, ms_index,
                pk.n_seg, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(), n_threads);
    }
    if (cfg->want_results) {
        ctx->host_scan.assign(dec.scan, dec.scan + dec.n_records_out);
        ctx->host_bc.assign(dec.bc, dec.bc + dec.n_records_out);
        out->scan = ctx->host_scan.data();
        out->bc = ctx->host_bc.data();
        std::memset(&ctx->host_stats, 0, sizeof ctx->host_stats);  // ReadFlags.addForCounting over the chunk (the counters behind ReadScanner.html)
        SMI_RC(smi_scan_stats_add(&ctx->host_stats, &dec));
        out->stats = &ctx->host_stats;
    }
    out->passed = ctx->host_out[0];
    out->failed = ctx->host_out[1];
    out->passed_bytes = out->passed_text_bytes = totals[0];
    out->failed_bytes = out->failed_text_bytes = totals[1];
    out->n_passed = totals[2];
    return SMI_OK;
}

extern "C" int smi_scanfastq_pass1_chunk_packed(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, int five_prime, int dont_search_polya,
                                                uint32_t *d_hist, int n_threads, size_t *n_records, uint32_t *fastq_errors) {
    if (!ctx || !d_hist || !n_records || (!text && n_bytes)) {
        set_error("smi_scanfastq_pass1_chunk_packed: null argument");
        return SMI_ERR_INVALID;
    }
    *n_records = 0;
    if (fastq_errors) *fastq_errors = 0;
    if (n_bytes == 0) return SMI_OK;
    size_t n = 0;
    uint32_t fq_err = 0;
    double ms_index = 0;
    smi_packed_reads pk;
    const int rc_ix = index_and_pack(ctx, "smi_scanfastq_pass1_chunk_packed", text, n_bytes, n_threads, &n, &fq_err, &pk, &ms_index);
    *n_records = n;
    if (fastq_errors) *fastq_errors = fq_err;
    if (rc_ix != SMI_OK) return rc_ix;
    if (n == 0) return SMI_OK;
    const smi_fastq_record *recs = static_cast<const smi_fastq_record *>(ctx->host_buf[smi_ctx::HB_RECS]);
    const uint64_t *offs = static_cast<const uint64_t *>(ctx->host_buf[smi_ctx::HB_OFFS]);
    SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_QTAIL, n * (size_t)SMI_END_BASES));
    SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_QSUM, n * 4));
    uint8_t *h_qtail = static_cast<uint8_t *>(ctx->host_buf[smi_ctx::HB_QTAIL]);
    uint32_t *h_qsum = static_cast<uint32_t *>(ctx->host_buf[smi_ctx::HB_QSUM]);
    if (int rc = smi_pack_quals_host(text, recs, n, five_prime, h_qtail, h_qsum, n_threads)) {
        if (fastq_errors && std::strstr(smi_last_error(), "line end inside")) *fastq_errors = SMI_FQ_LENGTH_MISMATCH;
        return rc;
    }
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t pstride = pk.total_words, planes_words = 4 * pstride;
    const size_t need = pad(planes_words * 4) + pad(n * 4) + pad((n + 1) * 8) + pad((size_t)SMI_ENDS_ROWS * 2 * n * 4) + 2 * pad(n * 4) + pad(n * (size_t)SMI_END_BASES) +
                        pad(n * sizeof(smi_scan_result)) + pad(n * sizeof(smi_bc_window)) + 4096;
    SMI_RC(ensure_arena(ctx, need));
    Arena A(ctx);
    uint32_t *d_planes = A.take<uint32_t>(planes_words);
    uint32_t *d_pstart = A.take<uint32_t>(n);
    uint64_t *d_offs = A.take<uint64_t>(n + 1);
    uint32_t *d_ends = A.take<uint32_t>((size_t)SMI_ENDS_ROWS * 2 * n);
    int32_t *d_len = A.take<int32_t>(n);
    uint32_t *d_qsum = A.take<uint32_t>(n);
    uint8_t *d_qtail = A.take<uint8_t>(n * (size_t)SMI_END_BASES);
    smi_scan_result *d_scan = A.take<smi_scan_result>(n);
    smi_bc_window *d_win = A.take<smi_bc_window>(n);
    SMI_RC(upload_planes(ctx, &pk, n, d_planes, d_pstart, s));
    if (!pk.pstart) d_pstart = nullptr;
    SMI_HIP(hipMemcpyAsync(d_offs, offs, (n + 1) * 8, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_qtail, h_qtail, n * (size_t)SMI_END_BASES, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_qsum, h_qsum, n * 4, hipMemcpyHostToDevice, s));
    smi_scan_config sc;
    SMI_RC(worker_scan_config(ctx, 1, five_prime, dont_search_polya, &sc));
    SMI_RC(launch_ends_from_planes(ctx, d_planes, pstride, d_offs, d_offs, nullptr, n, d_ends, d_len, s, d_pstart));
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

// ---------------------------------------------------------------------------------------------------------------------------
// `assignumis`, one chunk of BamReader (records of one chromosome stretch): what OneBatchExecutor.call + UmiClustering.cluster do
// with it (FJ!umifinder/OneBatchExecutor.java:L61-90, FJ!umifinder/analyzers/clustering/UmiClustering.java:L97-161) after
// ReadGrouper.groupSams (FJ!umifinder/bamreaders/ReadGrouper.java:L82-230): parse the read names, clustering position, region
// grouping, (cell barcode, region) groups in first-member order with members in input order, 14-base windows, K-UMI on the
// device, clustering on host threads, the values behind U8 / U7 / U1 / U2 per record.  3' barcoding.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

// FastqRecordExt.lambda$getScanDatFromReadName$4 (L397-408): text behind the first `tag` up to the next '_'
bool extract(const char *sub, size_t n, const char *tag, const char **val, size_t *len) {
    const size_t tl = std::strlen(tag);
    if (n < tl) return false;
    for (size_t i = 0; i + tl <= n; i++)
        if (std::memcmp(sub + i, tag, tl) == 0) {
            size_t a = i + tl, b = a;
            while (b < n && sub[b] != '_') b++;
            *val = sub + a;
            *len = b - a;
            return true;
        }
    return false;
}

bool to_int(const char *v, size_t n, long *out) {  // Integer.parseInt
    if (n == 0 || n > 11) return false;
    char buf[16];
    std::memcpy(buf, v, n);
    buf[n] = 0;
    char *end = nullptr;
    const long x = std::strtol(buf, &end, 10);
    if (end != buf + n) return false;
    *out = x;
    return true;
}

struct NameData {
    bool present = false, has_bc = false;
    long ae = 0, ps = 0, bc_end = 0;
    bool has_ps = false, has_bc_end = false;
    const char *bc = nullptr, *x = nullptr;
    size_t bc_len = 0, x_len = 0;
    float q = 0.0f;
    bool has_q = false;
};

// FastqRecordExt.getScanDatFromReadName (L395-496), the fields this step needs; returns SMI_ERR_INVALID where the reference
// throws AdapterInfoNotFoundInReadException / NumberFormatException
int parse_name(const char *name, size_t n, int bc_edit_limit, NameData &d) {
    d = NameData();
    const char *mark = nullptr;
    for (size_t i = 0; i + 5 <= n && !mark; i++)
        if (std::memcmp(name + i, "_REV_", 5) == 0) mark = name + i;
    if (!mark)
        for (size_t i = 0; i + 5 <= n && !mark; i++)
            if (std::memcmp(name + i, "_FWD_", 5) == 0) mark = name + i;
    if (!mark) return SMI_OK;  // Optional.absent()
    const char *sub = mark + 4;
    const size_t sn = (size_t)(name + n - sub);
    const char *v;
    size_t l;
    if (!extract(sub, sn, "AE=", &v, &l) || !to_int(v, l, &d.ae)) {
        set_error("smi_assignumis_chunk: adapter position (AE=) not found in a read name (AdapterInfoNotFoundInReadException)");
        return SMI_ERR_INVALID;
    }
    d.present = true;
    if (extract(sub, sn, "PS=", &v, &l)) d.has_ps = to_int(v, l, &d.ps);
    long ed = 0;
    if (extract(sub, sn, "ed=", &v, &l) && to_int(v, l, &ed) && (bc_edit_limit < 0 || ed <= bc_edit_limit)) {
        if (extract(sub, sn, "bc=", &v, &l)) {
            d.bc = v;
            d.bc_len = l;
            d.has_bc = true;
        }
        if (extract(sub, sn, "bcEnd=", &v, &l)) d.has_bc_end = to_int(v, l, &d.bc_end);
    }
    if (extract(sub, sn, "X=", &v, &l)) {
        d.x = v;
        d.x_len = l;
    }
    if (extract(sub, sn, "Q=", &v, &l) && l < 30) {
        char buf[32];
        size_t k = 0;
        while (k < l && v[k] != ' ') {
            buf[k] = v[k];
            k++;
        }
        buf[k] = 0;
        d.q = std::strtof(buf, nullptr);  // Float.parseFloat
        d.has_q = k > 0;
    }
    return SMI_OK;
}

inline uint32_t code4(char c) {
    switch (c) {
    case 'A': return 1;
    case 'G': return 2;
    case 'C': return 4;
    case 'T': return 8;
    default: return 15;
    }
}
inline uint32_t comp4(uint32_t c) { return c == 1 ? 8 : c == 8 ? 1 : c == 2 ? 4 : c == 4 ? 2 : 15; }
inline char dec4(uint32_t c) { return c == 1 ? 'A' : c == 2 ? 'G' : c == 4 ? 'C' : c == 8 ? 'T' : 'N'; }

// the three umi_length-mers (12 as shipped) at offsets -1, 0, +1 behind the barcode on the reverse complement of X= (ClusteringEditDistanceBase L297-350,
// getStrandedShortSeqPosFromReadPos FastqRecordExt.java:L378): umi_length + 2 bases, 4-bit codes, base k in bits [4k+3:4k]
// 5' barcoding: X= is read forwards (getSeq(), L312-313) and the barcode end on it is bcEnd - AE + 3 (L378 with is5pBarcoding)
bool umi_window(const NameData &d, bool five_prime, int umi_len, uint64_t *packed) {
    const long pos = five_prime ? d.bc_end - d.ae + 3 : d.ae + 3 - d.bc_end;
    if (!d.x || pos < 1 || pos + umi_len + 1 > (long)d.x_len) return false;
    uint64_t w = 0;
    for (int k = 0; k < umi_len + 2; k++) {
        const uint32_t c = five_prime ? code4(d.x[(size_t)(pos - 1 + k)]) : comp4(code4(d.x[d.x_len - (size_t)(pos + k)]));
        w |= (uint64_t)c << (4 * k);
    }
    *packed = w;
    return true;
}

}  // namespace

namespace {
// assignumis -f: the UMI window of record i as a random sequence of umi_len + 2 bases (4-bit codes A G C T), from (seed, i): the same function on the
// host path and in K-UPARSE (smi_umi_stage.hip)
uint64_t random_umi_window(uint64_t seed, uint64_t i, int umi_len) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    uint64_t w = 0;
    for (int k = 0; k < umi_len + 2; k++) w |= (uint64_t)(1u << ((z >> (2 * k)) & 3u)) << (4 * k);
    return w;
}

// umis/umi_length of a chunk: the configuration's, else the context's knob, else 12
int chunk_umi_length(const smi_ctx *ctx, const smi_assignumis_config *cfg) { return cfg->umi_length > 0 ? cfg->umi_length : ctx_umi_length(ctx); }

// The chunk on host threads, K-UMI alone on the device (rounds 1 and 2): the path of record for anything the device parser does not
// evaluate itself (UP_NONSTD: barcodes that are not 16 letters of ACGT, exotic number formats) and for SMI_AU_HOST=1, which the tests
// use to hold the device stage to it.
int assignumis_chunk_host(smi_ctx *ctx, const char *names, const uint32_t *name_off, const uint16_t *flags,
                          const int32_t *pos0, const uint32_t *cigars, const uint32_t *cigar_off, int32_t n,
                          const smi_assignumis_config *cfg, smi_umi_tag *out, int32_t *n_done) {
    *n_done = 0;
    if (n == 0) return SMI_OK;
    const int UL = chunk_umi_length(ctx, cfg);
    // SMI_AU_TIMING=1: the host stages of this call on stderr (where the time of the second worker goes)
    const bool timing = std::getenv("SMI_AU_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "smi_assignumis_chunk %-12s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::vector<NameData> nd((size_t)n);
    std::vector<int32_t> cpos((size_t)n, 0);
    std::vector<uint8_t> has_pos((size_t)n, 0), rev((size_t)n, 0);
    // names and clustering positions: independent per record, on cfg->n_threads host threads (the reference parses them inside its
    // parallel stream, UmiFinderWorker)
    {
        const int nt = std::max(1, std::min(cfg->n_threads > 0 ? cfg->n_threads : 1, 64));
        std::vector<int> rcs((size_t)nt, SMI_OK);
        std::vector<std::string> msgs((size_t)nt);  // the error text is thread-local: carried back to the calling thread
        auto work = [&](int t) {
            const int32_t lo = (int32_t)((int64_t)n * t / nt), hi = (int32_t)((int64_t)n * (t + 1) / nt);
            for (int32_t i = lo; i < hi; i++) {
                int rc = parse_name(names + name_off[i], name_off[i + 1] - name_off[i], cfg->bc_edit_limit, nd[i]);
                if (rc != SMI_OK) {
                    rcs[t] = rc;
                    msgs[t] = smi_last_error();
                    return;
                }
                rev[i] = (flags[i] & 16) ? 1 : 0;
                // NanoporeRead$ReadScanData.generateReadScanData / getGenomePosition (L86-116): reference position under read position
                // polyA start - distanceFromReadEndForGrouping (3'), adapter end + cell_bc_length + umi_length + that distance (5')
                const bool five = cfg->five_prime != 0;
                if (nd[i].present && (five || nd[i].has_ps) && !(flags[i] & 4)) {
                    int32_t p = 0;
                    const int32_t read_pos = five ? (int32_t)nd[i].ae + 16 + UL + cfg->grouping_distance : (int32_t)nd[i].ps - cfg->grouping_distance;
                    rc = smi_ref_position_at_read_position(cigars + cigar_off[i], (int32_t)(cigar_off[i + 1] - cigar_off[i]), pos0[i] + 1, read_pos, &p);
                    if (rc < 0) {
                        rcs[t] = rc;
                        msgs[t] = smi_last_error();
                        return;
                    }
                    if (rc == 1) {
                        has_pos[i] = 1;
                        cpos[i] = p;
                    }
                }
            }
        };
        if (nt == 1 || n < 4096)
            for (int t = 0; t < nt; t++) work(t);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < nt; t++)
            if (rcs[t] != SMI_OK) {
                set_error(msgs[t]);
                return rcs[t];
            }
    }
    lap("names");
    std::vector<int32_t> region((size_t)n, -1);
    SMI_RC(smi_region_group(cpos.data(), has_pos.data(), rev.data(), n, cfg->max_dist, cfg->keep_data_end, region.data(), n_done));
    lap("regions");
    const int32_t nd_ = *n_done;
    for (int32_t i = 0; i < n; i++) {
        smi_umi_tag &t = out[i];
        std::memset(&t, 0, sizeof t);
        t.region = i < nd_ ? region[i] : -1;
        t.center = -1;
        t.u1 = t.u2 = -1;
    }
    // the reads' own UMI windows (U7): independent per record, on the host threads
    std::vector<uint64_t> win((size_t)n, 0);
    std::vector<uint8_t> has_w((size_t)n, 0);
    {
        const int nt = std::max(1, std::min(cfg->n_threads > 0 ? cfg->n_threads : 1, 64));
        auto work = [&](int t) {
            const int32_t lo = (int32_t)((int64_t)nd_ * t / nt), hi = (int32_t)((int64_t)nd_ * (t + 1) / nt);
            for (int32_t i = lo; i < hi; i++) {
                const NameData &d = nd[i];
                const bool bc_ok = d.present && d.has_bc && d.has_bc_end && d.x && d.has_q;
                uint64_t w = 0;
                bool ok = bc_ok && umi_window(d, cfg->five_prime != 0, UL, &w);
                if (ok && cfg->random_umi_seed) w = random_umi_window(cfg->random_umi_seed, (uint64_t)i, UL);
                if (d.present && d.has_bc) out[i].flags |= SMI_UMI_HAS_BC;
                if (ok) {
                    win[i] = w;
                    has_w[i] = 1;
                    out[i].flags |= SMI_UMI_HAS_U7;
                    for (int k = 0; k < UL; k++) out[i].u7[k] = dec4((uint32_t)(w >> (4 * (k + 1))) & 15u);
                }
            }
        };
        if (nt == 1 || nd_ < 4096)
            for (int t = 0; t < nt; t++) work(t);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
    }
    // (cell barcode, region) groups in the order their first member appears, members in input order.  Key = the barcode's bytes (up to
    // 16, the usual case, held in two words; longer ones through a string map) + region: one pass gives every record its group and every
    // group its size, a second one fills the member lists of the groups that have two or more
    struct GKey {
        uint64_t a, b;
        uint32_t region, len;
        bool operator==(const GKey &o) const { return a == o.a && b == o.b && region == o.region && len == o.len; }
    };
    struct GHash {
        size_t operator()(const GKey &k) const {
            uint64_t h = k.a * 0x9E3779B97F4A7C15ull ^ (k.b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full ^ ((uint64_t)k.region << 8 | k.len) * 0x165667B19E3779F9ull;
            return (size_t)(h ^ (h >> 29));
        }
    };
    std::unordered_map<GKey, uint32_t, GHash> index;
    index.reserve((size_t)nd_);
    std::unordered_map<std::string, uint32_t> long_index;
    std::vector<int32_t> gid((size_t)nd_, -1);
    std::vector<uint32_t> gsize;
    for (int32_t i = 0; i < nd_; i++) {
        if (!has_w[i] || region[i] < 0) continue;
        const NameData &d = nd[i];
        uint32_t g;
        if (d.bc_len <= 16) {
            GKey k{0, 0, (uint32_t)region[i], (uint32_t)d.bc_len};
            std::memcpy(&k.a, d.bc, std::min<size_t>(d.bc_len, 8));
            if (d.bc_len > 8) std::memcpy(&k.b, d.bc + 8, d.bc_len - 8);
            auto it = index.find(k);
            if (it == index.end()) {
                g = (uint32_t)gsize.size();
                index.emplace(k, g);
                gsize.push_back(0);
            } else
                g = it->second;
        } else {
            std::string key(d.bc, d.bc_len);
            key += '#';
            key += std::to_string(region[i]);
            auto it = long_index.find(key);
            if (it == long_index.end()) {
                g = (uint32_t)gsize.size();
                long_index.emplace(std::move(key), g);
                gsize.push_back(0);
            } else
                g = it->second;
        }
        gid[i] = (int32_t)g;
        gsize[g]++;
    }
    std::vector<uint32_t> goff(1, 0);
    std::vector<uint64_t> poff(1, 0), moff(1, 0);
    std::vector<int32_t> slot(gsize.size(), -1);  // group -> its number among the groups that are clustered
    for (size_t g = 0; g < gsize.size(); g++) {
        if (gsize[g] < 2) continue;  // UmiClustering.lambda$cluster$6
        slot[g] = (int32_t)goff.size() - 1;
        const uint64_t k = gsize[g];
        goff.push_back(goff.back() + (uint32_t)k);
        poff.push_back(poff.back() + k * (k + 1) / 2);
        moff.push_back(moff.back() + k * k);
    }
    std::vector<int32_t> order((size_t)goff.back());
    {
        std::vector<uint32_t> cursor(goff.begin(), goff.end() - 1);
        for (int32_t i = 0; i < nd_; i++)
            if (gid[i] >= 0 && slot[(size_t)gid[i]] >= 0) order[cursor[(size_t)slot[(size_t)gid[i]]]++] = i;
    }
    const uint32_t n_groups = (uint32_t)goff.size() - 1;
    if (n_groups == 0) return SMI_OK;
    const size_t m = order.size();
    std::vector<uint64_t> wpk(m);
    std::vector<float> qv(m);
    for (size_t j = 0; j < m; j++) {
        wpk[j] = win[order[j]];
        qv[j] = nd[order[j]].q;
    }
    lap("groups");
    // ---- K-UMI ------------------------------------------------------------------------------------------------------------
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    SMI_RC(ensure_arena(ctx, pad(m * 8) + pad(goff.size() * 4) + 2 * pad(goff.size() * 8) + pad(moff.back()) + 1024));
    Arena A(ctx);
    uint64_t *d_w = A.take<uint64_t>(m);
    uint32_t *d_go = A.take<uint32_t>(goff.size());
    uint64_t *d_po = A.take<uint64_t>(goff.size()), *d_mo = A.take<uint64_t>(goff.size());
    uint8_t *d_dist = A.take<uint8_t>(moff.back());
    SMI_HIP(hipMemcpyAsync(d_w, wpk.data(), m * 8, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_go, goff.data(), goff.size() * 4, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_po, poff.data(), goff.size() * 8, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_mo, moff.data(), goff.size() * 8, hipMemcpyHostToDevice, s));
    SMI_RC(launch_umi_dist(ctx, d_w, d_go, d_po, d_mo, n_groups, poff.back(), d_dist, s, UL));
    std::vector<uint8_t> dist(moff.back());
    SMI_HIP(hipMemcpyAsync(dist.data(), d_dist, moff.back(), hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    lap("k-umi");
    // ---- clustering -------------------------------------------------------------------------------------------------------
    std::vector<smi_umi_assignment> asg(m);
    std::vector<uint8_t> skipped(m, 0);
    smi_umi_cluster_config cc;
    SMI_RC(smi_umi_cluster_default_config(&cc));
    if (cfg->cluster) cc = *cfg->cluster;
    SMI_RC(smi_umi_cluster_groups(dist.data(), moff.data(), goff.data(), n_groups, qv.data(), &cc, asg.data(), skipped.data(),
                                  cfg->n_threads > 0 ? cfg->n_threads : 1));
    lap("clustering");
    for (uint32_t g = 0; g < n_groups; g++)
        for (uint32_t j = goff[g]; j < goff[g + 1]; j++) {
            smi_umi_tag &t = out[order[j]];
            if (asg[j].center < 0) {
                if (skipped[j]) t.flags |= SMI_UMI_SKIPPED;  // UMI_CLUSTERING_SKIPPED_HIGHCOMPLEXITY | DONT_ASSIGN_UMI
                continue;
            }
            const int32_t c = order[goff[g] + (uint32_t)asg[j].center];
            t.flags |= SMI_UMI_CLUSTERED;
            t.center = c;
            t.u1 = asg[j].ed;
            t.u2 = asg[j].ed_second;
            const uint64_t cw = win[c];
            for (int k = 0; k < UL; k++) t.u8[k] = dec4((uint32_t)(cw >> (4 * (k + 1 + asg[j].offset))) & 15u);
        }
    return SMI_OK;
}

}  // namespace

extern "C" int smi_umi_cluster_groups_device(smi_ctx *ctx, const uint8_t *d_dist, const uint64_t *d_mat_off, const uint32_t *d_group_off, uint32_t n_groups,
                                             const float *d_mean_qv, const smi_umi_cluster_config *cfg, smi_umi_assignment *d_out, uint8_t *d_skipped, void *stream) {
    if (!ctx || !cfg || (n_groups && (!d_dist || !d_mat_off || !d_group_off || !d_mean_qv || !d_out || !d_skipped)) || cfg->complete_link_ed < 0 ||
        cfg->fold_depth_below_max <= 0) {
        set_error("smi_umi_cluster_groups_device: bad argument");
        return SMI_ERR_INVALID;
    }
    SMI_HIP(hipSetDevice(ctx->device));
    int dev_max = std::min(cfg->own_clusterer_above, kUmiClusterDeviceMax);
    if (cfg->single_link_switch < dev_max) dev_max = cfg->single_link_switch;
    return launch_umi_cluster(ctx, d_dist, d_mat_off, d_group_off, n_groups, d_mean_qv, *cfg, dev_max, d_out, d_skipped, static_cast<hipStream_t>(stream));
}

// `assignumis` for one chunk, the UMI stage on the device (smi_umi_stage.hip): K-UPARSE -> [host: region grouping] -> key sort ->
// K-UMI -> K-UCLUST (groups of up to 100 reads; larger groups: their matrix comes back and ClusterOne_MyClustering runs on the host) ->
// K-UTAG.  Same results as the host path, record by record (tests/test_umi_stage_gpu.py).
extern "C" int smi_assignumis_chunk(smi_ctx *ctx, const char *names, const uint32_t *name_off, const uint16_t *flags,
                                    const int32_t *pos0, const uint32_t *cigars, const uint32_t *cigar_off, int32_t n,
                                    const smi_assignumis_config *cfg, smi_umi_tag *out, int32_t *n_done) {
    if (!ctx || !names || !name_off || !flags || !pos0 || !cigar_off || !cfg || !out || !n_done || n < 0) {
        set_error("smi_assignumis_chunk: null argument");
        return SMI_ERR_INVALID;
    }
    *n_done = 0;
    if (n == 0) return SMI_OK;
    const int UL = chunk_umi_length(ctx, cfg);
    if (UL < 8 || UL > 12) {
        set_error("smi_assignumis_chunk: umis/umi_length must be 8 .. 12 in this build");
        return SMI_ERR_INVALID;
    }
    smi_umi_cluster_config cc;
    SMI_RC(smi_umi_cluster_default_config(&cc));
    if (cfg->cluster) cc = *cfg->cluster;
    if (std::getenv("SMI_AU_HOST")) return assignumis_chunk_host(ctx, names, name_off, flags, pos0, cigars, cigar_off, n, cfg, out, n_done);
    const bool timing = std::getenv("SMI_AU_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "smi_assignumis_chunk(device) %-12s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t N = (size_t)n, name_bytes = name_off[n], n_cig = cigar_off[n];
    const size_t tmp_bytes = umi_group_scratch_bytes(n + 2);
    const size_t G = N / 2 + 2;
    const size_t fixed = pad(name_bytes + 16) + 2 * pad((N + 1) * 4) + pad(N * 2) + pad(N * 4) + pad((n_cig + 1) * 4) + pad(N * sizeof(UmiParsed)) + pad(N * 4) +
                         3 * pad((N + 2) * 8) + 4 * pad((N + 2) * 4) + pad(16) + pad(16) + pad((N / 64 + 2) * 8) + 4 * pad((N + 2) * 4) + 4 * pad((N + 2) * 8) + pad(G * 4) + pad(N * 4) + 2 * pad(G * 8) +
                         pad(N * 8) + pad(N * 4) + pad(tmp_bytes) + pad(N * sizeof(smi_umi_assignment)) + pad(N) + pad(N * sizeof(smi_umi_tag)) + 8192;
    SMI_RC(ensure_arena(ctx, fixed));
    Arena A(ctx);
    char *d_names = A.take<char>(name_bytes + 16);
    uint32_t *d_noff = A.take<uint32_t>(N + 1), *d_coff = A.take<uint32_t>(N + 1);
    uint16_t *d_flags = A.take<uint16_t>(N);
    int32_t *d_pos0 = A.take<int32_t>(N);
    uint32_t *d_cig = A.take<uint32_t>(n_cig + 1);
    UmiParsed *d_parsed = A.take<UmiParsed>(N);
    int32_t *d_region = A.take<int32_t>(N);
    uint32_t *d_rcount = A.take<uint32_t>(4);
    uint64_t *d_rbits = A.take<uint64_t>((N + 63) / 64 + 1);
    UmiGroupBuffers B;
    B.keys = A.take<uint64_t>(N + 2);
    B.keys_sorted = A.take<uint64_t>(N + 2);
    B.run_keys = A.take<uint64_t>(N + 2);
    B.idx = A.take<uint32_t>(N + 2);
    B.idx_sorted = A.take<uint32_t>(N + 2);
    B.run_len = A.take<uint32_t>(N + 2);
    B.run_start = A.take<uint32_t>(N + 2);
    B.n_runs = A.take<uint32_t>(4);
    B.gsize = A.take<uint32_t>(N + 2);
    B.gkept = A.take<uint32_t>(N + 2);
    B.gslot = A.take<uint32_t>(N + 2);
    B.goff_run = A.take<uint32_t>(N + 2);
    B.gpairs = A.take<uint64_t>(N + 2);
    B.gmat = A.take<uint64_t>(N + 2);
    B.poff_run = A.take<uint64_t>(N + 2);
    B.moff_run = A.take<uint64_t>(N + 2);
    B.group_off = A.take<uint32_t>(G);
    B.order = A.take<uint32_t>(N);
    B.pair_off = A.take<uint64_t>(G);
    B.mat_off = A.take<uint64_t>(G);
    B.wpk = A.take<uint64_t>(N);
    B.qv = A.take<float>(N);
    B.tmp = A.take<uint8_t>(tmp_bytes);
    B.tmp_bytes = tmp_bytes;
    smi_umi_assignment *d_asg = A.take<smi_umi_assignment>(N);
    uint8_t *d_skipped = A.take<uint8_t>(N);
    smi_umi_tag *d_tags = A.take<smi_umi_tag>(N);
    SMI_HIP(hipMemcpyAsync(d_names, names, name_bytes, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_noff, name_off, (N + 1) * 4, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_coff, cigar_off, (N + 1) * 4, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_flags, flags, N * 2, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_pos0, pos0, N * 4, hipMemcpyHostToDevice, s));
    if (n_cig) SMI_HIP(hipMemcpyAsync(d_cig, cigars, n_cig * 4, hipMemcpyHostToDevice, s));
    SMI_RC(launch_umi_parse(ctx, d_names, d_noff, d_flags, d_pos0, d_cig, d_coff, n, cfg->five_prime != 0, cfg->grouping_distance, cfg->bc_edit_limit, UL, cfg->random_umi_seed, d_parsed, s));
    // region grouping: the sort by clustering position runs on the device; the sorted keys (8 bytes per read with a position) come down, the
    // chains and their refinement -- a sequential sweep with the reference's own quirks -- run on the host, the region numbers go up
    const size_t n_words = (N + 63) / 64;
    SMI_RC(launch_umi_region_keys(ctx, d_parsed, n, B, d_rcount, d_rbits, s));
    SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_RKEYS, N * 8 + 16));
    SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_RBITS, n_words * 8 + 16));
    SMI_RC(ensure_host_buf(ctx, smi_ctx::HB_REGION, N * 4));
    uint64_t *h_keys = static_cast<uint64_t *>(ctx->host_buf[smi_ctx::HB_RKEYS]);
    uint64_t *h_bits = static_cast<uint64_t *>(ctx->host_buf[smi_ctx::HB_RBITS]);
    int32_t *h_region = static_cast<int32_t *>(ctx->host_buf[smi_ctx::HB_REGION]);
    uint32_t *h_rcount = reinterpret_cast<uint32_t *>(h_bits + n_words);
    SMI_HIP(hipMemcpyAsync(h_rcount, d_rcount, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(h_keys, B.keys_sorted, N * 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(h_bits, d_rbits, n_words * 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    lap("parse+sort");
    if (h_rcount[1])  // a name the device parser does not evaluate: the host path reads it itself (and reports AE= missing as the reference does)
        return assignumis_chunk_host(ctx, names, name_off, flags, pos0, cigars, cigar_off, n, cfg, out, n_done);
    SMI_RC(region_group_from_sorted(&ctx->region_work, h_keys, h_rcount[0], n, h_bits, cfg->max_dist, cfg->keep_data_end, h_region, n_done));
    lap("regions");
    SMI_HIP(hipMemcpyAsync(d_region, h_region, N * 4, hipMemcpyHostToDevice, s));
    uint64_t totals[4];
    SMI_RC(launch_umi_groups(ctx, d_parsed, d_region, n, *n_done, B, totals, s));
    const uint32_t n_groups = (uint32_t)totals[0], m = (uint32_t)totals[1];
    lap("groups");
    if (n_groups) {
        // the matrices: a grow-only buffer of the context beside the arena
        if (ctx->umi_dist_bytes < totals[3]) {
            SMI_HIP(hipStreamSynchronize(s));
            if (ctx->umi_dist) SMI_HIP(hipFree(ctx->umi_dist));
            ctx->umi_dist = nullptr;
            ctx->umi_dist_bytes = 0;
            const size_t want = (size_t)totals[3] + (size_t)totals[3] / 4 + 4096;
            SMI_HIP(hipMalloc(&ctx->umi_dist, want));
            ctx->umi_dist_bytes = want;
        }
        uint8_t *d_dist = static_cast<uint8_t *>(ctx->umi_dist);
        SMI_RC(launch_umi_dist(ctx, B.wpk, B.group_off, B.pair_off, B.mat_off, n_groups, totals[2], d_dist, s, UL, true));  // rows padded to whole lines (smi_umi_stage.h)
        int dev_max = std::min(cc.own_clusterer_above, kUmiClusterDeviceMax);
        if (cc.single_link_switch < dev_max) dev_max = cc.single_link_switch;  // (never with the shipped values: the switch sits at 3000)
        SMI_RC(launch_umi_cluster(ctx, d_dist, B.mat_off, B.group_off, n_groups, B.qv, cc, dev_max, d_asg, d_skipped, s, true));
        // groups the kernel left alone: their matrix comes back, the host clusters them, the assignments go up again
        std::vector<uint32_t> goff((size_t)n_groups + 1);
        SMI_HIP(hipMemcpyAsync(goff.data(), B.group_off, ((size_t)n_groups + 1) * 4, hipMemcpyDeviceToHost, s));
        SMI_HIP(hipStreamSynchronize(s));
        lap("k-umi+clust");
        std::vector<uint32_t> big;
        for (uint32_t g = 0; g < n_groups; g++)
            if ((int)(goff[g + 1] - goff[g]) > dev_max) big.push_back(g);
        if (!big.empty()) {
            std::vector<uint64_t> moff((size_t)n_groups + 1);
            SMI_HIP(hipMemcpyAsync(moff.data(), B.mat_off, ((size_t)n_groups + 1) * 8, hipMemcpyDeviceToHost, s));
            SMI_HIP(hipStreamSynchronize(s));
            const bool own_on_host = getenv("SMI_AU_OWN_HOST") != nullptr;  // cross-check switch: the host clusterer for every large group
            for (uint32_t g : big) {
                const uint32_t k = goff[g + 1] - goff[g];
                if ((int)k > cc.own_clusterer_above && !own_on_host) {
                    // ClusterOne_MyClustering: its n^2 loops on the device over the matrix where it lies (smi_cluster.hip)
                    SMI_RC(umi_cluster_own_device(ctx, d_dist + moff[g], (int)k, B.qv + goff[g], cc, d_asg + goff[g], d_skipped + goff[g], s, (int)umi_ld(k, true)));
                    continue;
                }
                std::vector<uint8_t> mat((size_t)k * k);
                std::vector<float> qv(k);
                std::vector<smi_umi_assignment> asg(k);
                std::vector<uint8_t> sk(k, 0);
                SMI_HIP(hipMemcpy2DAsync(mat.data(), k, d_dist + moff[g], (size_t)umi_ld(k, true), k, k, hipMemcpyDeviceToHost, s));  // (rows of the padded matrix -> dense)
                SMI_HIP(hipMemcpyAsync(qv.data(), B.qv + goff[g], (size_t)k * 4, hipMemcpyDeviceToHost, s));
                SMI_HIP(hipStreamSynchronize(s));
                const uint64_t zero64 = 0;
                const uint32_t one_group[2] = {0, k};
                SMI_RC(smi_umi_cluster_groups(mat.data(), &zero64, one_group, 1, qv.data(), &cc, asg.data(), sk.data(), cfg->n_threads > 0 ? cfg->n_threads : 1));
                SMI_HIP(hipMemcpyAsync(d_asg + goff[g], asg.data(), (size_t)k * sizeof(smi_umi_assignment), hipMemcpyHostToDevice, s));
                SMI_HIP(hipMemcpyAsync(d_skipped + goff[g], sk.data(), k, hipMemcpyHostToDevice, s));
                SMI_HIP(hipStreamSynchronize(s));
            }
            lap("big groups");
        }
    }
    SMI_RC(launch_umi_tags(ctx, d_parsed, d_region, n, *n_done, B, n_groups, m, d_asg, d_skipped, UL, d_tags, s));
    SMI_HIP(hipMemcpyAsync(out, d_tags, N * sizeof(smi_umi_tag), hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    lap("tags");
    return SMI_OK;
}

extern "C" int smi_assignumis_default_config(smi_assignumis_config *cfg) {
    if (!cfg) {
        set_error("smi_assignumis_default_config: null argument");
        return SMI_ERR_INVALID;
    }
    std::memset(cfg, 0, sizeof *cfg);
    cfg->max_dist = 500;           // max_GenomeDistance_forGrouping
    cfg->grouping_distance = 100;  // distanceFromReadEndForGrouping
    cfg->bc_edit_limit = -1;       // -b not given
    cfg->n_threads = 4;
    return SMI_OK;
}
