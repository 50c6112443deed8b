// smi_ctx.hip -- context, error reporting and the extern "C" boundary of libsicelore_mi (see include/sicelore_mi.h).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "smi_internal.h"
#include "smi_umi_stage.h"

namespace smi {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char *what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return SMI_ERR_HIP;
}

int time_begin(smi_ctx *ctx, int kid, hipStream_t s) {
    if (!ctx->timing) return SMI_OK;
    SMI_HIP(hipEventRecord(ctx->kev[kid][0], s));
    return SMI_OK;
}

int time_end(smi_ctx *ctx, int kid, hipStream_t s) {
    if (!ctx->timing) return SMI_OK;
    SMI_HIP(hipEventRecord(ctx->kev[kid][1], s));
    ctx->kev_valid[kid] = true;
    ctx->ev0 = ctx->kev[kid][0];
    ctx->ev1 = ctx->kev[kid][1];
    ctx->ev_valid = true;
    return SMI_OK;
}

Pyramid pyramid_of(const smi_ctx *ctx) {
    Pyramid p;
    p.l0 = ctx->l0;
    p.l0s = ctx->l0s;
    p.l1 = ctx->l1;
    p.fine = ctx->fine;
    p.rank = ctx->rank;
    p.t2 = ctx->t2;
    p.n1 = ctx->n1_valid ? ctx->n1 : nullptr;
    p.n2 = ctx->n1_valid ? ctx->n1 + kL1Words : nullptr;
    p.n1s = ctx->n1_valid && ctx->n1s_valid ? ctx->n1 + 2 * kL1Words : nullptr;
    p.n2s = ctx->n1_valid && ctx->n1s_valid ? ctx->n1 + 3 * kL1Words : nullptr;
    p.nb2 = ctx->n1_valid && ctx->nb2_valid ? ctx->n1_owner : nullptr;
    p.nb = ctx->nb_valid ? ctx->nb : nullptr;
    p.nb5 = ctx->nb_valid && ctx->nb5_valid ? ctx->nb5 : nullptr;
    p.nt = ctx->nb_valid && ctx->nt_cap ? ctx->nt : nullptr;
    p.nt_cap = ctx->nt_cap;
    return p;
}

void *pin_words(smi_ctx *ctx) {
    if (!ctx->pin_words && hipHostMalloc(&ctx->pin_words, 4096, hipHostMallocDefault) != hipSuccess) ctx->pin_words = nullptr;
    return ctx->pin_words;
}

int ensure_host_buf(smi_ctx *ctx, int which, size_t bytes) {
    if (ctx->host_buf_bytes[which] >= bytes) return SMI_OK;
    if (ctx->host_buf[which]) SMI_HIP(hipHostFree(ctx->host_buf[which]));
    ctx->host_buf[which] = nullptr;
    ctx->host_buf_bytes[which] = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    SMI_HIP(hipHostMalloc(&ctx->host_buf[which], want, hipHostMallocDefault));
    ctx->host_buf_bytes[which] = want;
    return SMI_OK;
}

static int ensure_stage(smi_ctx *ctx, size_t in_bytes, size_t out_bytes) {
    if (in_bytes > ctx->stage_in_bytes) {
        if (ctx->stage_in) SMI_HIP(hipFree(ctx->stage_in));
        ctx->stage_in = nullptr;
        ctx->stage_in_bytes = 0;
        SMI_HIP(hipMalloc(&ctx->stage_in, in_bytes));
        ctx->stage_in_bytes = in_bytes;
    }
    if (out_bytes > ctx->stage_out_bytes) {
        if (ctx->stage_out) SMI_HIP(hipFree(ctx->stage_out));
        ctx->stage_out = nullptr;
        ctx->stage_out_bytes = 0;
        SMI_HIP(hipMalloc(&ctx->stage_out, out_bytes));
        ctx->stage_out_bytes = out_bytes;
    }
    return SMI_OK;
}

static int bind(const smi_ctx *ctx) {
    if (!ctx) {
        set_error("null context");
        return SMI_ERR_INVALID;
    }
    SMI_HIP(hipSetDevice(ctx->device));
    return SMI_OK;
}

}  // namespace smi

using namespace smi;

extern "C" {

const char *smi_last_error(void) { return g_last_error.c_str(); }

const char *smi_version(void) { return "sicelore-mi 0.1 (gfx950)"; }

int smi_ctx_create(int device, smi_ctx **out) {
    if (!out) {
        set_error("smi_ctx_create: out is null");
        return SMI_ERR_INVALID;
    }
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        set_error("smi_ctx_create: no HIP device visible (this library has no CPU fallback)");
        return SMI_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n_dev) {
        set_error("smi_ctx_create: device ordinal out of range");
        return SMI_ERR_INVALID;
    }
    SMI_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SMI_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string("smi_ctx_create: device is ") + prop.gcnArchName + ", this build targets gfx950 only");
        return SMI_ERR_NO_DEVICE;
    }
    smi_ctx *ctx = new smi_ctx();
    ctx->device = device;
    auto fail = [&](int rc) {
        smi_ctx_destroy(ctx);
        return rc;
    };
#define SMI_TRY(call)                                     \
    do {                                                  \
        hipError_t e2 = (call);                           \
        if (e2 != hipSuccess) return fail(hip_fail(e2, #call)); \
    } while (0)
    SMI_TRY(hipMalloc((void **)&ctx->l0, 2 * kL0Words * 4));  // l0 | l0s: one allocation, K-BC1 addresses both from l0
    ctx->l0s = ctx->l0 + kL0Words;
    SMI_TRY(hipMalloc((void **)&ctx->l1, kL1Words * 4));
    SMI_TRY(hipMalloc((void **)&ctx->t2, 4 * kL0Words * 4));
    SMI_TRY(hipMalloc((void **)&ctx->fine, kFineWords * 4));
    SMI_TRY(hipMalloc((void **)&ctx->rank, kRankEntries * 4));
    SMI_TRY(hipMalloc((void **)&ctx->block_counts, kRankEntries * 4));
    SMI_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (int k = 0; k < SMI_K_COUNT; k++) {
        SMI_TRY(hipEventCreate(&ctx->kev[k][0]));
        SMI_TRY(hipEventCreate(&ctx->kev[k][1]));
    }
#undef SMI_TRY
    *out = ctx;
    return SMI_OK;
}

// A worker lane: a context of its own (stream, arena, pinned output, timing) that READS the barcode set of `owner` instead of
// holding the 616 MiB pyramid a second time.  The reference runs nCPU Parser workers over one hashMapForBCfinding
// (FJ!nanoporereadscanner/WorkerReadscanner.java:L188-204); this is the same shape: one set, several workers whose transfers and
// kernels overlap.  After the owner has loaded another set (pass 1 -> pass 2) the lanes call smi_ctx_lane_refresh.
int smi_ctx_create_lane(smi_ctx *owner, smi_ctx **out) {
    if (!owner || !out) {
        set_error("smi_ctx_create_lane: null argument");
        return SMI_ERR_INVALID;
    }
    if (owner->set_owner) {
        set_error("smi_ctx_create_lane: the owner must be a full context, not a lane");
        return SMI_ERR_INVALID;
    }
    *out = nullptr;
    SMI_HIP(hipSetDevice(owner->device));
    smi_ctx *ctx = new smi_ctx();
    ctx->device = owner->device;
    ctx->set_owner = owner;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    for (int k = 0; k < SMI_K_COUNT && e == hipSuccess; k++) {
        e = hipEventCreate(&ctx->kev[k][0]);
        if (e == hipSuccess) e = hipEventCreate(&ctx->kev[k][1]);
    }
    if (e != hipSuccess) {
        smi_ctx_destroy(ctx);
        return hip_fail(e, "smi_ctx_create_lane");
    }
    *out = ctx;
    return smi_ctx_lane_refresh(ctx);
}

int smi_ctx_lane_refresh(smi_ctx *lane) {
    if (!lane || !lane->set_owner) {
        set_error("smi_ctx_lane_refresh: not a lane");
        return SMI_ERR_INVALID;
    }
    const smi_ctx *o = lane->set_owner;
    lane->l0 = o->l0;
    lane->l0s = o->l0s;
    lane->l1 = o->l1;
    lane->t2 = o->t2;
    lane->n1 = o->n1;
    lane->n1_valid = o->n1_valid;
    lane->n1s_valid = o->n1s_valid;
    lane->n1_owner = o->n1_owner;
    lane->nb2_valid = o->nb2_valid;
    lane->nb = o->nb;
    lane->nb_valid = o->nb_valid;
    lane->nb5 = o->nb5;
    lane->nb5_valid = o->nb5_valid;
    lane->nt = o->nt;
    lane->nt_cap = o->nt_cap;
    lane->fine = o->fine;
    lane->rank = o->rank;
    lane->block_counts = o->block_counts;
    lane->n_keys = o->n_keys;
    lane->set_mode = o->set_mode;
    lane->polya_len = o->polya_len;
    lane->polya_frac = o->polya_frac;
    lane->polya_window = o->polya_window;
    lane->knobs = o->knobs;
    lane->knobs_set = o->knobs_set;
    lane->random_bc_seed = o->random_bc_seed;
    return SMI_OK;
}

int smi_ctx_destroy(smi_ctx *ctx) {
    if (!ctx) return SMI_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (!ctx->set_owner) {  // a lane borrows these
        (void)hipFree(ctx->l0);
        (void)hipFree(ctx->l1);
        (void)hipFree(ctx->t2);
        (void)hipFree(ctx->n1);
        (void)hipFree(ctx->nb);
        (void)hipFree(ctx->nb5);
        (void)hipFree(ctx->nt);
        (void)hipFree(ctx->n1_owner);
        (void)hipFree(ctx->fine);
        (void)hipFree(ctx->rank);
        (void)hipFree(ctx->block_counts);
    }
    (void)hipFree(ctx->bc_codes);
    (void)hipFree(ctx->stage_in);
    (void)hipFree(ctx->scan_tmp);
    (void)hipFree(ctx->arena);
    (void)hipFree(ctx->umi_dist);
    (void)hipFree(ctx->deflate_scratch);
    region_work_free(ctx->region_work);
    (void)hipFree(ctx->chim_list);
    (void)hipFree(ctx->chim_slots);
    (void)hipFree(ctx->chim_work);
    (void)hipFree(ctx->chim_flat);
    (void)hipFree(ctx->umi_own);
    (void)hipFree(ctx->umi_plan);
    if (ctx->pin_words) (void)hipHostFree(ctx->pin_words);
    (void)hipHostFree(ctx->host_out[0]);
    (void)hipHostFree(ctx->host_out[1]);
    for (void *hb : ctx->host_buf) (void)hipHostFree(hb);
    (void)hipFree(ctx->stage_out);
    for (int k = 0; k < SMI_K_COUNT; k++) {
        if (ctx->kev[k][0]) (void)hipEventDestroy(ctx->kev[k][0]);
        if (ctx->kev[k][1]) (void)hipEventDestroy(ctx->kev[k][1]);
    }
    if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
    if (ctx->side_join) (void)hipEventDestroy(ctx->side_join);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SMI_OK;
}

int smi_ctx_device(const smi_ctx *ctx) { return ctx ? ctx->device : -1; }

int smi_ctx_set_polya(smi_ctx *ctx, int polya_len, float polya_frac, int window_polya) {
    if (!ctx) {
        set_error("null context");
        return SMI_ERR_INVALID;
    }
    const int len = polya_len > 0 ? polya_len : 15, win = window_polya > 0 ? window_polya : 150;
    if (len < 5 || len > 30 || win + len + 10 > 175 || (polya_frac != 0.0f && !(polya_frac > 0.0f && polya_frac <= 1.0f))) {
        set_error("smi_ctx_set_polya: this build scans 175 bases of each read end: 5 <= polyA length <= 30, window + length + 10 <= 175, 0 < fraction <= 1");
        return SMI_ERR_INVALID;
    }
    ctx->polya_len = polya_len > 0 ? polya_len : 0;
    ctx->polya_frac = polya_frac;
    ctx->polya_window = window_polya > 0 ? window_polya : 0;
    return SMI_OK;
}

int smi_set_timing(smi_ctx *ctx, int enabled) {
    if (!ctx) {
        set_error("null context");
        return SMI_ERR_INVALID;
    }
    ctx->timing = enabled != 0;
    ctx->ev_valid = false;
    for (int k = 0; k < SMI_K_COUNT; k++) ctx->kev_valid[k] = false;
    return SMI_OK;
}

int smi_last_kernel_ms(smi_ctx *ctx, float *ms) {
    if (!ctx || !ms) {
        set_error("smi_last_kernel_ms: null argument");
        return SMI_ERR_INVALID;
    }
    *ms = -1.0f;
    if (!ctx->ev_valid) return SMI_OK;
    if (int rc = bind(ctx)) return rc;
    SMI_HIP(hipEventSynchronize(ctx->ev1));
    SMI_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return SMI_OK;
}

int smi_kernel_ms(smi_ctx *ctx, int kernel_id, float *ms) {
    if (!ctx || !ms || kernel_id < 0 || kernel_id >= SMI_K_COUNT) {
        set_error("smi_kernel_ms: bad argument");
        return SMI_ERR_INVALID;
    }
    *ms = -1.0f;
    if (!ctx->kev_valid[kernel_id]) return SMI_OK;
    if (int rc = bind(ctx)) return rc;
    SMI_HIP(hipEventSynchronize(ctx->kev[kernel_id][1]));
    SMI_HIP(hipEventElapsedTime(ms, ctx->kev[kernel_id][0], ctx->kev[kernel_id][1]));
    return SMI_OK;
}

int smi_set_barcode_set_device(smi_ctx *ctx, const uint32_t *d_keys, size_t n, int mode, void *stream) {
    if (ctx && ctx->set_owner) {
        set_error("smi_set_barcode_set: a lane reads its owner's barcode set; load it on the owner and call smi_ctx_lane_refresh");
        return SMI_ERR_STATE;
    }
    if (int rc = bind(ctx)) return rc;
    if ((!d_keys && n) || (mode != SMI_SET_USED_LIST && mode != SMI_SET_WHITELIST && mode != SMI_SET_MEMBERSHIP)) {
        set_error("smi_set_barcode_set_device: bad argument");
        return SMI_ERR_INVALID;
    }
    ctx->set_mode = -1;
    if (int rc = launch_build_pyramid(ctx, d_keys, n, (hipStream_t)stream, mode == SMI_SET_MEMBERSHIP)) return rc;
    ctx->set_mode = mode;
    return SMI_OK;
}

int smi_set_stats(smi_ctx *ctx, uint64_t out[8], int digests) {
    if (int rc = bind(ctx)) return rc;
    if (!out) {
        set_error("smi_set_stats: null argument");
        return SMI_ERR_INVALID;
    }
    const smi_ctx *o = ctx->set_owner ? ctx->set_owner : ctx;
    if (o->set_mode < 0) {
        set_error("smi_set_stats: no barcode set loaded");
        return SMI_ERR_STATE;
    }
    for (int i = 0; i < 8; i++) out[i] = 0;
    out[0] = o->n_keys;
    // the membership pyramid: l0 + l0s, l1, t2, fine, rank + block counts; then what is valid for the loaded set
    uint64_t bytes = (2 * kL0Words + kL1Words + 4 * kL0Words + kFineWords) * 4ull + 2ull * kRankEntries * 4;
    if (o->nb_valid) bytes += kFineWords * 4ull;
    if (o->nb5_valid) bytes += ((uint64_t)1 << 24) * 40 * 4;
    if (o->nt_cap) bytes += (uint64_t)o->nt_alloc * 8 + (o->nt_alloc >> 3) * 4;  // (the table as allocated, with the build's bucket counters behind it)
    if (o->n1_valid) bytes += 4ull * kL1Words * 4;
    if (o->nb2_valid) bytes += kFineWords * 4ull;
    out[1] = bytes;
    out[2] = o->set_build_us;
    if (digests) {
        uint64_t d5[5];
        if (int rc = launch_set_digests(const_cast<smi_ctx *>(o), d5, ctx->stream)) return rc;
        for (int i = 0; i < 5; i++) out[3 + i] = d5[i];
    }
    return SMI_OK;
}

int smi_set_barcode_set(smi_ctx *ctx, const uint64_t *keys, size_t n, int mode) {
    if (int rc = bind(ctx)) return rc;
    if (!keys && n) {
        set_error("smi_set_barcode_set: keys is null");
        return SMI_ERR_INVALID;
    }
    std::vector<uint32_t> k32(n);
    for (size_t i = 0; i < n; i++) {
        if (keys[i] >> 32) {
            set_error("smi_set_barcode_set: key does not fit 16 nt (N in a barcode, or cell_bc_length != 16)");
            return SMI_ERR_INVALID;
        }
        k32[i] = (uint32_t)keys[i];
    }
    if (int rc = ensure_stage(ctx, std::max<size_t>(n * 4, 16), 0)) return rc;
    if (n) SMI_HIP(hipMemcpyAsync(ctx->stage_in, k32.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    SMI_HIP(hipStreamSynchronize(ctx->stream));
    return smi_set_barcode_set_device(ctx, (const uint32_t *)ctx->stage_in, n, mode, ctx->stream);
}

static int check_match_args(smi_ctx *ctx, const void *in, const void *out, size_t n, int max_ed) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!in || !out)) {
        set_error("smi_bc_match: null buffer");
        return SMI_ERR_INVALID;
    }
    if (max_ed < 0 || max_ed > 2) {
        set_error("smi_bc_match: bcEditDistance must be 0, 1 or 2");
        return SMI_ERR_INVALID;
    }
    if (ctx->set_mode < 0) {
        set_error("smi_bc_match: no barcode set loaded (call smi_set_barcode_set first)");
        return SMI_ERR_STATE;
    }
    return SMI_OK;
}

int smi_bc_match_device(smi_ctx *ctx, const smi_bc_window *d_windows, size_t n, int max_ed, int five_prime,
                        smi_bc_result *d_out, void *stream) {
    if (int rc = check_match_args(ctx, d_windows, d_out, n, max_ed)) return rc;
    return launch_bc_match(ctx, d_windows, n, max_ed, five_prime, d_out, (hipStream_t)stream);
}

int smi_bc_match_batch(smi_ctx *ctx, const smi_bc_window *windows, size_t n, int max_ed, int five_prime,
                       smi_bc_result *out) {
    if (int rc = check_match_args(ctx, windows, out, n, max_ed)) return rc;
    if (!n) return SMI_OK;
    if (int rc = ensure_stage(ctx, n * sizeof(smi_bc_window), n * sizeof(smi_bc_result))) return rc;
    SMI_HIP(hipMemcpyAsync(ctx->stage_in, windows, n * sizeof(smi_bc_window), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = launch_bc_match(ctx, (const smi_bc_window *)ctx->stage_in, n, max_ed, five_prime,
                                 (smi_bc_result *)ctx->stage_out, ctx->stream))
        return rc;
    SMI_HIP(hipMemcpyAsync(out, ctx->stage_out, n * sizeof(smi_bc_result), hipMemcpyDeviceToHost, ctx->stream));
    SMI_HIP(hipStreamSynchronize(ctx->stream));
    return SMI_OK;
}

int smi_extract_windows_device(smi_ctx *ctx, const uint8_t *d_reads, const uint64_t *d_offsets,
                               const int32_t *d_adapter_end, size_t n, int five_prime, smi_bc_window *d_windows,
                               void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!d_reads || !d_offsets || !d_adapter_end || !d_windows)) {
        set_error("smi_extract_windows_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_extract_windows(ctx, d_reads, d_offsets, d_adapter_end, n, five_prime, d_windows, (hipStream_t)stream);
}

int smi_hist_device(smi_ctx *ctx, const uint32_t *d_keys, const uint8_t *d_pass, size_t n, uint32_t *d_hist,
                    void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!d_keys || !d_hist)) {
        set_error("smi_hist_device: null buffer");
        return SMI_ERR_INVALID;
    }
    if (ctx->set_mode < 0) {
        set_error("smi_hist_device: no barcode set loaded");
        return SMI_ERR_STATE;
    }
    return launch_hist(ctx, d_keys, d_pass, n, d_hist, (hipStream_t)stream);
}


int smi_bc_counts_device(smi_ctx *ctx, const smi_bc_result *d_results, size_t n, uint32_t *d_counts, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!d_results || !d_counts)) {
        set_error("smi_bc_counts_device: null buffer");
        return SMI_ERR_INVALID;
    }
    if (ctx->set_mode < 0) {
        set_error("smi_bc_counts_device: no barcode set loaded");
        return SMI_ERR_STATE;
    }
    return launch_bc_counts(ctx, d_results, n, d_counts, (hipStream_t)stream);
}

// ---- config.xml's knobs (smi_run_knobs) ---------------------------------------------------------------------------------------------------
int smi_run_knobs_default(smi_run_knobs *k) {
    if (!k) {
        set_error("smi_run_knobs_default: null argument");
        return SMI_ERR_INVALID;
    }
    std::memset(k, 0, sizeof(*k));
    k->min_read_length = 200;  // Jar/config.xml:21
    k->min_mean_bc_qv = 8;     // :55
    k->min_mean_read_qv = 8;   // :57
    k->min_adapter_3p_matches = 8;  // :59
    k->polya_len = 15;         // :95
    k->polya_frac = 0.75f;     // :97
    k->window_polya = 150;     // :105
    k->internal_pat_len = 15;  // :99
    k->internal_pat_frac = 0.70f;  // :101
    std::strcpy(k->adapter3p, "CTTCCGATCT");                       // :111
    std::strcpy(k->adapter3p_complete, "CTACACGACGCTCTTCCGATCT");  // :113
    k->adapter3p_max_mm = 3;                                       // :115
    k->adapter3p_complete_max_mm = 5;                              // :118
    std::strcpy(k->adapter5p, "CTTCCGATCT");                       // :124
    std::strcpy(k->adapter5p_complete, "CTACACGACGCTCTTCCGATCT");  // :126
    k->adapter5p_max_mm = 3;                                       // :129
    k->adapter5p_complete_max_mm = 5;                              // :132
    k->adapter5p_window = 110;                                     // :134
    std::strcpy(k->adapter3p5_complete, "AAGCAGTGGTATCAACGCAGAGTAC");  // :141
    k->adapter3p5_complete_max_mm = 5;                             // :146
    std::strcpy(k->tso_complete, "AAGCAGTGGTATCAACGCAGAGTACAT");   // :170
    k->tso_complete_max_mm = 6;                                    // :172
    k->umi_length = 12;                                            // :264
    std::strcpy(k->tso_scan, "AACGCAGAGTACATGG");                  // :155
    k->tso_scan_max_mm = 5;                                        // :157
    k->tso_scan_min_consec = 8;                                    // :161
    k->tso_scan_min_two_best = 12;                                 // :164
    k->tso_scan_window = 90;                                       // :166
    return SMI_OK;
}

namespace {
const smi_run_knobs *shipped_knobs() {  // (a pointer: this file's definitions have C linkage)
    static const smi_run_knobs k = [] {
        smi_run_knobs v;
        smi_run_knobs_default(&v);
        return v;
    }();
    return &k;
}

// the limits of this build, knob by knob (the message names config.xml's element)
int check_knobs(const smi_run_knobs &k) {
    auto seq_ok = [](const char *s, size_t cap, int want) {
        size_t n = 0;
        while (n < cap && s[n]) n++;
        if (n != (size_t)want) return false;
        for (size_t i = 0; i < n; i++)
            if (s[i] != 'A' && s[i] != 'C' && s[i] != 'G' && s[i] != 'T') return false;
        return true;
    };
    struct {
        const char *name;
        const char *s;
        int want;
    } seqs[] = {{"adapter_for3pBarcoding/sequence", k.adapter3p, 10},
                {"adapter_for3pBarcoding/sequence_complete", k.adapter3p_complete, 22},
                {"fiveprimeadapter_for5pBarcoding/sequence", k.adapter5p, 10},
                {"fiveprimeadapter_for5pBarcoding/sequence_complete", k.adapter5p_complete, 22},
                {"threeprimeadapter_for5pBarcoding/sequence_complete", k.adapter3p5_complete, 25},
                {"tso_for3pBarcoding/sequence_complete", k.tso_complete, 27},
                {"tso_for3pBarcoding/sequence", k.tso_scan, 16}};
    for (const auto &q : seqs)
        if (!seq_ok(q.s, q.want == 16 ? 20 : 32, q.want)) {
            char msg[256];
            std::snprintf(msg, sizeof msg, "smi_ctx_set_knobs: %s: this build has kernels for %d bases of A / C / G / T here (the length of the shipped sequence)", q.name, q.want);
            set_error(msg);
            return SMI_ERR_INVALID;
        }
    struct {
        const char *name;
        int v, lo, hi;
    } ints[] = {{"readscanner/minReadLength", k.min_read_length, 0, 1 << 30},
                {"readscanner/minMeanBCqv", k.min_mean_bc_qv, 0, 93},
                {"readscanner/minMeanReadqv", k.min_mean_read_qv, 0, 93},
                {"readscanner/minAdapter3pMatches", k.min_adapter_3p_matches, 0, 22},
                {"polyAT/polyATlength", k.polya_len, 5, 30},
                {"polyAT/internalpATlength", k.internal_pat_len, 2, 15},
                {"adapter_for3pBarcoding/maxNeedlemanMismatches", k.adapter3p_max_mm, 0, 30},
                {"adapter_for3pBarcoding/maxCompleteSeqNeedlemanMismatches", k.adapter3p_complete_max_mm, 0, 30},
                {"fiveprimeadapter_for5pBarcoding/maxNeedlemanMismatches", k.adapter5p_max_mm, 0, 29},
                {"fiveprimeadapter_for5pBarcoding/maxCompleteSeqNeedlemanMismatches", k.adapter5p_complete_max_mm, 0, 30},
                {"threeprimeadapter_for5pBarcoding/maxCompleteSeqNeedlemanMismatches", k.adapter3p5_complete_max_mm, 0, 30},
                {"tso_for3pBarcoding/maxCompleteSeqNeedlemanMismatches", k.tso_complete_max_mm, 0, 30},
                {"umis/umi_length", k.umi_length, 8, 12},
                {"tso_for3pBarcoding/maxNeedlemanMismatches", k.tso_scan_max_mm, 0, 30},
                {"tso_for3pBarcoding/minTSO_NeedlemanConsecutiveMatches", k.tso_scan_min_consec, 0, 32},
                {"tso_for3pBarcoding/minTSO_TwoBestConsecutiveMatches", k.tso_scan_min_two_best, 0, 32},
                {"tso_for3pBarcoding/windowForTSOsearch", k.tso_scan_window, 16, 112}};
    for (const auto &q : ints)
        if (q.v < q.lo || q.v > q.hi) {
            char msg[256];
            std::snprintf(msg, sizeof msg, "smi_ctx_set_knobs: %s = %d: this build takes %d .. %d", q.name, q.v, q.lo, q.hi);
            set_error(msg);
            return SMI_ERR_INVALID;
        }
    if (k.window_polya < 1 || k.window_polya + k.polya_len + 10 > 175) {
        set_error("smi_ctx_set_knobs: polyAT/windowSearchForPolyA + polyATlength + 10 must fit the 175 scanned bases of a read end");
        return SMI_ERR_INVALID;
    }
    if (!(k.polya_frac > 0.0f && k.polya_frac <= 1.0f) || !(k.internal_pat_frac > 0.0f && k.internal_pat_frac <= 1.0f)) {
        set_error("smi_ctx_set_knobs: polyAT/fractionATInPolyAT and internalFractionATInPolyAT must lie in (0, 1]");
        return SMI_ERR_INVALID;
    }
    // 5' barcoding: the adapter is searched in the first AdapterSearchWindow + adapter + mismatches + 5 bases (PolyATadapterAnalyzer_5pBCUMI.java:L49-61); pass 1 scans the 22-mer
    if (k.adapter5p_window < 1 || k.adapter5p_window + 22 + k.adapter5p_max_mm + 1 + 5 > 192) {
        set_error("smi_ctx_set_knobs: fiveprimeadapter_for5pBarcoding/AdapterSearchWindow + 22 + maxNeedlemanMismatches + 6 must fit 192 bases");
        return SMI_ERR_INVALID;
    }
    return SMI_OK;
}
}  // namespace

int smi_ctx_set_knobs(smi_ctx *ctx, const smi_run_knobs *knobs) {
    if (!ctx) {
        set_error("null context");
        return SMI_ERR_INVALID;
    }
    if (!knobs) {
        ctx->knobs_set = false;
        return SMI_OK;
    }
    if (int rc = check_knobs(*knobs)) return rc;
    ctx->knobs = *knobs;
    ctx->knobs_set = true;
    return SMI_OK;
}

int smi_ctx_set_random_barcodes(smi_ctx *ctx, uint64_t seed) {
    if (!ctx) {
        set_error("null context");
        return SMI_ERR_INVALID;
    }
    ctx->random_bc_seed = seed;
    return SMI_OK;
}

int smi_ctx_get_knobs(const smi_ctx *ctx, smi_run_knobs *knobs) {
    if (!ctx || !knobs) {
        set_error("smi_ctx_get_knobs: null argument");
        return SMI_ERR_INVALID;
    }
    *knobs = ctx->knobs_set ? ctx->knobs : *shipped_knobs();
    return SMI_OK;
}

int smi_scan_config_from_knobs(const smi_run_knobs *knobs, int pass, int five_prime, int dont_search_polya, smi_scan_config *cfg) {
    if (!cfg || (pass != 1 && pass != 2)) {
        set_error("smi_scan_config_from_knobs: bad argument");
        return SMI_ERR_INVALID;
    }
    if (knobs)
        if (int rc = check_knobs(*knobs)) return rc;
    const smi_run_knobs &k = knobs ? *knobs : *shipped_knobs();
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->min_read_length = k.min_read_length;
    cfg->polya_len = k.polya_len;
    cfg->polya_frac = k.polya_frac;
    cfg->window_polya = k.window_polya;
    // Parser.java:L134-136: the adapter of the protocol; pass 2 scans `sequence`, pass 1 (no barcode map yet) `sequence_complete`, both with
    // maxNeedlemanMismatches; 5' barcoding allows one more (L99)
    cfg->max_mismatches = five_prime ? k.adapter5p_max_mm + 1 : k.adapter3p_max_mm;
    cfg->min_adapter_3p_matches = k.min_adapter_3p_matches;
    cfg->min_mean_bc_qv = k.min_mean_bc_qv;
    cfg->min_mean_read_qv = k.min_mean_read_qv;
    const char *ad = five_prime ? (pass == 1 ? k.adapter5p_complete : k.adapter5p) : (pass == 1 ? k.adapter3p_complete : k.adapter3p);
    cfg->adapter_len = (int32_t)std::strlen(ad);
    for (int i = 0; i < cfg->adapter_len; i++)
        cfg->adapter4[i] = ad[i] == 'A' ? 1u : ad[i] == 'G' ? 2u : ad[i] == 'C' ? 4u : 8u;
    if (five_prime) {
        cfg->five_prime = 1;
        cfg->dont_search_polya = dont_search_polya ? 1 : 0;
        cfg->adapter_search_window = k.adapter5p_window;
    }
    for (int i = 0; i < 16; i++) cfg->tso4[i] = k.tso_scan[i] == 'A' ? 1u : k.tso_scan[i] == 'G' ? 2u : k.tso_scan[i] == 'C' ? 4u : 8u;
    cfg->tso_window = k.tso_scan_window;
    cfg->tso_max_mismatches = k.tso_scan_max_mm;
    cfg->tso_min_consec = k.tso_scan_min_consec;
    cfg->tso_min_two_best = k.tso_scan_min_two_best;
    return SMI_OK;
}

int smi_chimera_config_from_knobs(const smi_run_knobs *knobs, int five_prime, smi_chimera_config *cfg) {
    if (!cfg) {
        set_error("smi_chimera_config_from_knobs: null argument");
        return SMI_ERR_INVALID;
    }
    if (knobs)
        if (int rc = check_knobs(*knobs)) return rc;
    const smi_run_knobs &k = knobs ? *knobs : *shipped_knobs();
    cfg->internal_pat_len = k.internal_pat_len;
    cfg->internal_pat_frac = k.internal_pat_frac;
    cfg->window_polya = k.window_polya;
    if (!five_prime) {
        cfg->tso_complete = k.tso_complete;
        cfg->adapter_complete = k.adapter3p_complete;
        cfg->tso_max_errors = k.tso_complete_max_mm;
        cfg->adapter_max_errors = k.adapter3p_complete_max_mm;
        cfg->bc_umi_len = 16 + k.umi_length;  // ChimeraFindernew.java:L74: umi_length + cell_bc_length
    } else {
        // ChimeraFindernew.<init> L75-78 for scantype != THREEP_BARCODE: the 5' adapter plays the TSO's part, the 3' adapter
        // of the 5' protocol is searched next to internal polyA / polyT, and no barcode + UMI lies between them (hasBCUMI = false)
        cfg->tso_complete = k.adapter5p_complete;
        cfg->adapter_complete = k.adapter3p5_complete;
        cfg->tso_max_errors = k.adapter5p_complete_max_mm;
        cfg->adapter_max_errors = k.adapter3p5_complete_max_mm;
        cfg->bc_umi_len = 0;
    }
    return SMI_OK;
}

int smi_scan_default_config(int pass, smi_scan_config *cfg) { return smi_scan_config_from_knobs(nullptr, pass, 0, 0, cfg); }

int smi_fastq_index_device(smi_ctx *ctx, const uint8_t *d_text, size_t n_bytes, uint64_t *d_line_start, size_t cap_lines,
                           uint64_t *d_name_start, uint32_t *d_name_len, uint64_t *d_seq_start, uint32_t *d_seq_len,
                           uint64_t *d_qual_start, uint64_t *d_offsets, size_t cap_records, size_t *n_records,
                           uint32_t *errors, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (!n_records || !errors || (n_bytes && (!d_text || !d_line_start || !d_name_start || !d_name_len || !d_seq_start ||
                                              !d_seq_len || !d_qual_start || !d_offsets))) {
        set_error("smi_fastq_index_device: null argument");
        return SMI_ERR_INVALID;
    }
    if (n_bytes >= ((size_t)1 << 42) || cap_records >= ((size_t)1 << 31) - 2) {
        set_error("smi_fastq_index_device: buffer too large for one call");
        return SMI_ERR_INVALID;
    }
    ctx->fq_swept_text = nullptr;  // (a caller of this entry point has not swept: nothing a worker left behind is taken for this text's flags)
    return launch_fastq_index(ctx, d_text, n_bytes, d_line_start, cap_lines, d_name_start, d_name_len, d_seq_start, d_seq_len,
                              d_qual_start, d_offsets, cap_records, n_records, errors, (hipStream_t)stream);
}

int smi_fastq_gather_device(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_start, const uint64_t *d_offsets, size_t n,
                            uint8_t *d_out, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!d_text || !d_start || !d_offsets || !d_out)) {
        set_error("smi_fastq_gather_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_fastq_gather(ctx, d_text, d_start, d_offsets, n, d_out, (hipStream_t)stream);
}

int smi_chimera_default_config(smi_chimera_config *cfg) { return smi_chimera_config_from_knobs(nullptr, 0, cfg); }

int smi_chimera_default_config_5p(smi_chimera_config *cfg) { return smi_chimera_config_from_knobs(nullptr, 1, cfg); }

size_t smi_read_planes_words(uint64_t total_bases, size_t n) { return 4 * read_planes_stride(total_bases, n); }

int smi_pack_reads_device(smi_ctx *ctx, const uint8_t *d_reads, const uint64_t *d_offsets, size_t n, uint64_t total_bases,
                          uint32_t *d_planes, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!d_reads || !d_offsets || !d_planes)) {
        set_error("smi_pack_reads_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_pack_reads(ctx, d_reads, d_offsets, nullptr, n, total_bases, d_planes, (hipStream_t)stream);
}

int smi_chimera_device(smi_ctx *ctx, const uint32_t *d_planes, const uint64_t *d_offsets, size_t n, uint64_t total_bases,
                       const smi_chimera_config *cfg, smi_chimera_result *d_out, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (!cfg || !cfg->tso_complete || !cfg->adapter_complete || (n && (!d_planes || !d_offsets || !d_out))) {
        set_error("smi_chimera_device: null argument");
        return SMI_ERR_INVALID;
    }
    return launch_chimera(ctx, d_planes, d_offsets, n, total_bases, cfg, d_out, (hipStream_t)stream);
}

int smi_split_offsets_device(smi_ctx *ctx, const smi_chimera_result *d_chim, const uint64_t *d_offsets, size_t n,
                             uint32_t *d_scratch, uint64_t *d_n_frag, uint64_t *d_frag_offsets, uint32_t *d_frag_src,
                             void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!d_chim || !d_offsets || !d_scratch || !d_n_frag || !d_frag_offsets)) {
        set_error("smi_split_offsets_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_split_offsets(ctx, d_chim, d_offsets, n, d_scratch, d_n_frag, d_frag_offsets, d_frag_src, (hipStream_t)stream);
}

int smi_scan_default_config_5p(int pass, int dont_search_polya, smi_scan_config *cfg) {
    return smi_scan_config_from_knobs(nullptr, pass, 1, dont_search_polya, cfg);
}

int smi_pack_ends_device(smi_ctx *ctx, const uint8_t *d_reads, const uint8_t *d_quals, const uint64_t *d_offsets,
                         size_t n, int five_prime, uint32_t *d_ends, int32_t *d_read_len, uint8_t *d_qtail,
                         uint32_t *d_qsum, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!d_reads || !d_offsets || !d_ends || !d_read_len || (d_quals && (!d_qtail || !d_qsum)))) {
        set_error("smi_pack_ends_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_pack_ends(ctx, d_reads, d_quals, d_offsets, nullptr, n, five_prime, d_ends, d_read_len, d_qtail, d_qsum,
                            (hipStream_t)stream);
}

// ---- the same packers reading the bases where the FASTQ text has them (no gathered copy of the chunk) ----------------------------------
int smi_pack_reads_text_device(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_seq_start, const uint64_t *d_offsets, size_t n,
                               uint64_t total_bases, uint32_t *d_planes, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!d_text || !d_seq_start || !d_offsets || !d_planes)) {
        set_error("smi_pack_reads_text_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_pack_reads(ctx, d_text, d_offsets, d_seq_start, n, total_bases, d_planes, (hipStream_t)stream);
}

int smi_pack_ends_text_device(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_base_start, const uint64_t *d_offsets, size_t n,
                              uint32_t *d_ends, int32_t *d_read_len, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!d_text || !d_base_start || !d_offsets || !d_ends || !d_read_len)) {
        set_error("smi_pack_ends_text_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_pack_ends(ctx, d_text, nullptr, d_offsets, d_base_start, n, 0, d_ends, d_read_len, nullptr, nullptr, (hipStream_t)stream);
}

int smi_ends_from_planes_device(smi_ctx *ctx, const uint32_t *d_planes, const uint64_t *d_read_offsets, size_t n_reads,
                                uint64_t total_bases, const uint64_t *d_rec_offsets, const uint32_t *d_frag_src, size_t n_records,
                                uint32_t *d_ends, int32_t *d_read_len, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n_records && (!d_planes || !d_read_offsets || !d_rec_offsets || !d_ends || !d_read_len)) {
        set_error("smi_ends_from_planes_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_ends_from_planes(ctx, d_planes, read_planes_stride(total_bases, n_reads), d_read_offsets, d_rec_offsets, d_frag_src,
                                   n_records, d_ends, d_read_len, (hipStream_t)stream);
}

int smi_frag_text_starts_device(smi_ctx *ctx, const uint64_t *d_seq_start, const uint64_t *d_qual_start, const uint64_t *d_offsets,
                                const uint64_t *d_frag_offsets, const uint32_t *d_frag_src, size_t n_out, uint64_t *d_base_start,
                                uint64_t *d_qual_out, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n_out && (!d_seq_start || !d_base_start || (d_qual_out && !d_qual_start) || (d_frag_src && (!d_offsets || !d_frag_offsets)))) {
        set_error("smi_frag_text_starts_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_frag_text_starts(ctx, d_seq_start, d_qual_start, d_offsets, d_frag_offsets, d_frag_src, n_out, d_base_start, d_qual_out,
                                   (hipStream_t)stream);
}

int smi_scan_device(smi_ctx *ctx, const uint32_t *d_ends, const int32_t *d_read_len, const uint8_t *d_qtail,
                    const uint32_t *d_qsum, size_t n, const smi_scan_config *cfg, smi_scan_result *d_out,
                    smi_bc_window *d_windows, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (!cfg || (n && (!d_ends || !d_read_len || !d_out)) || ((d_qtail == nullptr) != (d_qsum == nullptr))) {
        set_error("smi_scan_device: null buffer");
        return SMI_ERR_INVALID;
    }
    if (cfg->adapter_len != 10 && cfg->adapter_len != 22) {
        set_error("smi_scan_device: adapter length must be 10 or 22 in this build (config.xml sequence / sequence_complete)");
        return SMI_ERR_INVALID;
    }
    if (cfg->polya_len < 5 || cfg->polya_len > 30 || cfg->window_polya + cfg->polya_len + 10 > 175 ||
        cfg->window_polya < 1) {
        set_error("smi_scan_device: polyA window does not fit the 175-base scan region");
        return SMI_ERR_INVALID;
    }
    if (cfg->five_prime && (cfg->adapter_search_window < 1 ||
                            cfg->adapter_search_window + cfg->adapter_len + cfg->max_mismatches + 5 > 192)) {
        set_error("smi_scan_device: 5' search window + adapter + mismatches + 5 must fit 192 bases");
        return SMI_ERR_INVALID;
    }
    if (cfg->tso_window != 0) {
        bool ok = cfg->tso_window >= 16 && cfg->tso_window <= 112 && cfg->tso_max_mismatches >= 0 && cfg->tso_max_mismatches <= 30 && cfg->tso_min_consec >= 0 &&
                  cfg->tso_min_two_best >= 0;
        for (int i = 0; ok && i < 16; i++) ok = cfg->tso4[i] == 1u || cfg->tso4[i] == 2u || cfg->tso4[i] == 4u || cfg->tso4[i] == 8u;
        if (!ok) {
            set_error("smi_scan_device: the TSO of the read scan must be 16 bases of A / C / G / T, windowForTSOsearch 16 .. 112, the limits >= 0 (tso_window == 0: the shipped parameters)");
            return SMI_ERR_INVALID;
        }
    }
    {
        // The reference cuts window + length + 10 bases off each read end before it looks for polyA / polyT (PolyATSearcher.java:L178-181) and
        // AdapterSearchWindow + adapter + mismatches + 5 in 5' barcoding (PolyATadapterAnalyzer_5pBCUMI.java:L49-61): a read that passes
        // minReadLength and is shorter than that ends its run with a StringIndexOutOfBoundsException.  With the shipped 200 no read can; a
        // smaller minReadLength is refused here instead of guessing what such reads should become.
        int need = 0;
        if (!cfg->five_prime || !cfg->dont_search_polya) need = cfg->window_polya + cfg->polya_len + 10;
        if (cfg->five_prime) need = std::max(need, cfg->adapter_search_window + cfg->adapter_len + cfg->max_mismatches + 5);
        // 3' barcoding: windowForTSOsearch + 16 + 10 bases of each end for the TSO scan (PolyATadapterAnalyzer_3pBCUMI.scanReadForTSOs L128-131)
        if (!cfg->five_prime && cfg->tso_window) need = std::max(need, cfg->tso_window + 26);
        if (cfg->min_read_length < need) {
            char msg[320];
            std::snprintf(msg, sizeof msg,
                          "smi_scan_device: readscanner/minReadLength = %d, but %d bases are cut off each read end (windowSearchForPolyA + polyATlength + 10, "
                          "5': AdapterSearchWindow + adapter + mismatches + 5): the reference ends with an exception on the first read of %d .. %d bases",
                          cfg->min_read_length, need, cfg->min_read_length, need - 1);
            set_error(msg);
            return SMI_ERR_INVALID;
        }
    }
    return launch_scan(ctx, d_ends, d_read_len, d_qtail, d_qsum, n, cfg, d_out, d_windows, (hipStream_t)stream);
}

int smi_hist_windows_device(smi_ctx *ctx, const smi_bc_window *d_windows, const smi_scan_result *d_scan, size_t n,
                            uint32_t *d_hist, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!d_windows || !d_scan || !d_hist)) {
        set_error("smi_hist_windows_device: null buffer");
        return SMI_ERR_INVALID;
    }
    if (ctx->set_mode < 0) {
        set_error("smi_hist_windows_device: no barcode set loaded");
        return SMI_ERR_STATE;
    }
    return launch_hist_windows(ctx, d_windows, d_scan, n, d_hist, (hipStream_t)stream);
}

int smi_pass1_keys_device(smi_ctx *ctx, const smi_bc_window *d_windows, const smi_scan_result *d_scan, size_t n, uint64_t *d_keys, size_t cap,
                          uint64_t *d_count, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (!d_count || (n && (!d_windows || !d_scan || !d_keys))) {
        set_error("smi_pass1_keys_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_keys_windows(ctx, d_windows, d_scan, n, d_keys, cap, reinterpret_cast<unsigned long long *>(d_count), (hipStream_t)stream);
}

int smi_count_keys_device(smi_ctx *ctx, const uint64_t *d_keys, size_t n, uint64_t *d_unique, uint32_t *d_counts, uint64_t *d_n_unique, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (!d_n_unique || (n && (!d_keys || !d_unique || !d_counts))) {
        set_error("smi_count_keys_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_count_keys(ctx, d_keys, n, d_unique, d_counts, d_n_unique, (hipStream_t)stream);
}

uint64_t smi_umi_padded_row(uint32_t n) { return umi_ld(n, true); }
uint64_t smi_umi_padded_bytes(uint32_t n) { return umi_mat_bytes(n, true); }

int smi_umi_dist_device_padded(smi_ctx *ctx, const uint64_t *d_windows, const uint32_t *d_group_off, const uint64_t *d_pair_off, const uint64_t *d_mat_off,
                               uint32_t n_groups, uint64_t total_pairs, uint8_t *d_out, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n_groups && (!d_windows || !d_group_off || !d_pair_off || !d_mat_off || !d_out)) {
        set_error("smi_umi_dist_device_padded: null buffer");
        return SMI_ERR_INVALID;
    }
    if ((reinterpret_cast<uintptr_t>(d_out) & 63u) != 0) {
        set_error("smi_umi_dist_device_padded: the matrix buffer must start on a 64-byte boundary");
        return SMI_ERR_INVALID;
    }
    return launch_umi_dist(ctx, d_windows, d_group_off, d_pair_off, d_mat_off, n_groups, total_pairs, d_out, (hipStream_t)stream, ctx_umi_length(ctx), true);
}

int smi_umi_dist_device(smi_ctx *ctx, const uint64_t *d_windows, const uint32_t *d_group_off,
                        const uint64_t *d_pair_off, const uint64_t *d_mat_off, uint32_t n_groups,
                        uint64_t total_pairs, uint8_t *d_out, void *stream) {
    if (int rc = bind(ctx)) return rc;
    if (n_groups && (!d_windows || !d_group_off || !d_pair_off || !d_mat_off || !d_out)) {
        set_error("smi_umi_dist_device: null buffer");
        return SMI_ERR_INVALID;
    }
    return launch_umi_dist(ctx, d_windows, d_group_off, d_pair_off, d_mat_off, n_groups, total_pairs, d_out,
                           (hipStream_t)stream, ctx_umi_length(ctx));
}

}  // extern "C"

namespace smi {
int worker_scan_config(const smi_ctx *ctx, int pass, int five_prime, int dont_search_polya, smi_scan_config *sc) {
    if (int rc = smi_scan_config_from_knobs(ctx->knobs_set ? &ctx->knobs : nullptr, pass, five_prime, dont_search_polya, sc)) return rc;
    // -p / -f / -w of the command line win over config.xml (NanoporeReadScannerMain.java:L227-234)
    if (ctx->polya_len) sc->polya_len = ctx->polya_len;
    if (ctx->polya_frac != 0.0f) sc->polya_frac = ctx->polya_frac;
    if (ctx->polya_window) sc->window_polya = ctx->polya_window;
    return SMI_OK;
}
int worker_chimera_config(const smi_ctx *ctx, int five_prime, smi_chimera_config *cc) {
    if (int rc = smi_chimera_config_from_knobs(ctx->knobs_set ? &ctx->knobs : nullptr, five_prime, cc)) return rc;
    if (ctx->polya_window) cc->window_polya = ctx->polya_window;  // (the splitter keeps away from the read ends by windowSearchForPolyA + 70)
    return SMI_OK;
}
}  // namespace smi
