// smi_collective.hip -- the one exchange of the path behind the C ABI, and the host-buffer forms of the scan / UMI entry points.
//
// smi_hist_allreduce: sum of the pass-1 used-barcode histograms of several contexts (one per GPU of this process) with RCCL over
// xGMI -- what a single-process host (the reference is one JVM) calls between pass 1 and the used-list finalize.
// The reference has no counterpart: its threads add into one ConcurrentHashMap
// (FJ!nanoporereadscanner/analyzers/UsedCellBCListGenerator.java:L224-229, L254).  RCCL is opened with dlopen at the first call, so
// the library itself carries no link-time dependency on it (hosts that never use more than one GPU never load it).
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include "smi_internal.h"

using namespace smi;

namespace {
// the few RCCL symbols used, with the ABI of rccl.h (ncclResult_t = int, ncclComm_t = opaque pointer, ncclUint32 = 3, ncclSum = 0)
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return SMI_OK;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        set_error(std::string("smi_hist_allreduce: cannot open librccl.so: ") + dlerror());
        return SMI_ERR_STATE;
    }
    Rccl r;
    r.h = h;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd) {
        set_error("smi_hist_allreduce: librccl.so lacks an expected symbol");
        dlclose(h);
        return SMI_ERR_STATE;
    }
    g_rccl = r;
    return SMI_OK;
}

int rccl_fail(int rc, const char *what) {
    set_error(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
    return SMI_ERR_HIP;
}
}  // namespace

namespace {
// communicators are cached per device list: ncclCommInitAll costs hundreds of ms on 8 GPUs and a host calls the exchange once per
// run at least (twice with the BarcodesAssigned counters); smi_hist_allreduce_release destroys them
struct CommSet {
    std::vector<int> devs;
    std::vector<void *> comms;
};
std::vector<CommSet> g_comm_sets;
std::mutex g_comm_mu;  // also serialises the collective itself: one RCCL group at a time per process

struct DeviceRestore {  // the calling thread's current HIP device is left as it was found
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

int comms_for(const std::vector<int> &devs, std::vector<void *> **out) {
    for (auto &cs : g_comm_sets)
        if (cs.devs == devs) {
            *out = &cs.comms;
            return SMI_OK;
        }
    CommSet cs;
    cs.devs = devs;
    cs.comms.assign(devs.size(), nullptr);
    if (int rc = g_rccl.CommInitAll(cs.comms.data(), (int)devs.size(), devs.data())) {
        for (void *c : cs.comms)  // a partial initialisation leaks nothing
            if (c) (void)g_rccl.CommDestroy(c);
        return rccl_fail(rc, "ncclCommInitAll");
    }
    g_comm_sets.push_back(std::move(cs));
    *out = &g_comm_sets.back().comms;
    return SMI_OK;
}
}  // namespace

extern "C" int smi_hist_allreduce_after(smi_ctx **ctxs, int n_ctx, uint32_t **d_hist, size_t n_counters, void *const *producer_streams) {
    if (!ctxs || !d_hist || n_ctx <= 0) {
        set_error("smi_hist_allreduce: null argument");
        return SMI_ERR_INVALID;
    }
    std::vector<int> devs((size_t)n_ctx);
    for (int i = 0; i < n_ctx; i++) {
        if (!ctxs[i] || !d_hist[i]) {
            set_error("smi_hist_allreduce: null context or histogram");
            return SMI_ERR_INVALID;
        }
        devs[i] = ctxs[i]->device;
        for (int k = 0; k < i; k++)
            if (devs[k] == devs[i]) {
                set_error("smi_hist_allreduce: one context per GPU (two contexts on one device share a histogram instead)");
                return SMI_ERR_INVALID;
            }
    }
    if (n_counters == 0) return SMI_OK;
    if (int rc = load_rccl()) return rc;
    DeviceRestore restore;
    std::lock_guard<std::mutex> lk(g_comm_mu);
    std::vector<void *> *comms = nullptr;
    if (int rc = comms_for(devs, &comms)) return rc;
    // order every context's stream behind the stream that produced its histogram (an event on that stream; the default stream when the
    // caller names it as 0 is a legal producer too)
    if (producer_streams)
        for (int i = 0; i < n_ctx; i++) {
            SMI_HIP(hipSetDevice(devs[i]));
            hipEvent_t ev = nullptr;
            SMI_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            hipError_t e = hipEventRecord(ev, static_cast<hipStream_t>(producer_streams[i]));
            if (e == hipSuccess) e = hipStreamWaitEvent(ctxs[i]->stream, ev, 0);
            (void)hipEventDestroy(ev);  // released once the wait has been satisfied
            if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent (smi_hist_allreduce)");
        }
    int rc = g_rccl.GroupStart();
    for (int i = 0; i < n_ctx && rc == 0; i++) {
        hipError_t e = hipSetDevice(devs[i]);
        if (e != hipSuccess) {
            rc = -1;
            break;
        }
        rc = g_rccl.AllReduce(d_hist[i], d_hist[i], n_counters, /*ncclUint32*/ 3, /*ncclSum*/ 0, (*comms)[i], ctxs[i]->stream);
    }
    const int rc_end = g_rccl.GroupEnd();
    int ret = SMI_OK;
    if (rc != 0 || rc_end != 0) ret = rccl_fail(rc ? rc : rc_end, "ncclAllReduce");
    for (int i = 0; i < n_ctx; i++) {
        (void)hipSetDevice(devs[i]);
        hipError_t e = hipStreamSynchronize(ctxs[i]->stream);
        if (e != hipSuccess && ret == SMI_OK) ret = hip_fail(e, "hipStreamSynchronize (smi_hist_allreduce)");
    }
    return ret;
}

extern "C" int smi_hist_allreduce(smi_ctx **ctxs, int n_ctx, uint32_t **d_hist, size_t n_counters) {
    return smi_hist_allreduce_after(ctxs, n_ctx, d_hist, n_counters, nullptr);
}

extern "C" int smi_hist_allreduce_release(void) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    DeviceRestore restore;
    for (auto &cs : g_comm_sets)
        for (void *c : cs.comms)
            if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    g_comm_sets.clear();
    return SMI_OK;
}

// ---- host-buffer forms: upload, the device entry points, download ------------------------------------------------
extern "C" int smi_scan_batch(smi_ctx *ctx, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, size_t n_reads,
                              const smi_scan_config *cfg, smi_scan_result *out, smi_bc_window *out_windows) {
    if (!ctx || !cfg || (n_reads && (!bases || !offsets || !out))) {
        set_error("smi_scan_batch: null argument");
        return SMI_ERR_INVALID;
    }
    if (!n_reads) return SMI_OK;
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t total = (size_t)offsets[n_reads];
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t need = pad(total) * (quals ? 2 : 1) + pad((n_reads + 1) * 8) + pad((size_t)SMI_ENDS_ROWS * 2 * n_reads * 4) + pad(n_reads * 4) * 2 +
                        pad(n_reads * (size_t)SMI_END_BASES) + pad(n_reads * sizeof(smi_scan_result)) + pad(n_reads * sizeof(smi_bc_window)) + 4096;
    uint8_t *base = nullptr;
    SMI_HIP(hipMalloc(&base, need));
    struct Free {
        void *p;
        ~Free() { (void)hipFree(p); }
    } guard{base};
    size_t at = 0;
    auto take = [&](size_t bytes) {
        uint8_t *p = base + at;
        at += pad(bytes);
        return p;
    };
    uint8_t *d_bases = take(total), *d_quals = quals ? take(total) : nullptr;
    uint64_t *d_offs = reinterpret_cast<uint64_t *>(take((n_reads + 1) * 8));
    uint32_t *d_ends = reinterpret_cast<uint32_t *>(take((size_t)SMI_ENDS_ROWS * 2 * n_reads * 4));
    int32_t *d_len = reinterpret_cast<int32_t *>(take(n_reads * 4));
    uint32_t *d_qsum = reinterpret_cast<uint32_t *>(take(n_reads * 4));
    uint8_t *d_qtail = take(n_reads * (size_t)SMI_END_BASES);
    smi_scan_result *d_scan = reinterpret_cast<smi_scan_result *>(take(n_reads * sizeof(smi_scan_result)));
    smi_bc_window *d_win = reinterpret_cast<smi_bc_window *>(take(n_reads * sizeof(smi_bc_window)));
    SMI_HIP(hipMemcpyAsync(d_bases, bases, total, hipMemcpyHostToDevice, s));
    if (quals) SMI_HIP(hipMemcpyAsync(d_quals, quals, total, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_offs, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, s));
    if (int rc = smi_pack_ends_device(ctx, d_bases, d_quals, d_offs, n_reads, cfg->five_prime, d_ends, d_len, quals ? d_qtail : nullptr,
                                      quals ? d_qsum : nullptr, s))
        return rc;
    if (int rc = smi_scan_device(ctx, d_ends, d_len, quals ? d_qtail : nullptr, quals ? d_qsum : nullptr, n_reads, cfg, d_scan, d_win, s)) return rc;
    SMI_HIP(hipMemcpyAsync(out, d_scan, n_reads * sizeof(smi_scan_result), hipMemcpyDeviceToHost, s));
    if (out_windows) SMI_HIP(hipMemcpyAsync(out_windows, d_win, n_reads * sizeof(smi_bc_window), hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    return SMI_OK;
}

extern "C" int smi_umi_dist_batch(smi_ctx *ctx, const uint64_t *windows, const uint32_t *group_off, uint32_t n_groups, uint8_t *out) {
    if (!ctx || (n_groups && (!windows || !group_off || !out))) {
        set_error("smi_umi_dist_batch: null argument");
        return SMI_ERR_INVALID;
    }
    if (!n_groups) return SMI_OK;
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<uint64_t> pair_off((size_t)n_groups + 1, 0), mat_off((size_t)n_groups + 1, 0);
    for (uint32_t g = 0; g < n_groups; g++) {
        if (group_off[g + 1] < group_off[g]) {
            set_error("smi_umi_dist_batch: group offsets must ascend");
            return SMI_ERR_INVALID;
        }
        const uint64_t m = group_off[g + 1] - group_off[g];
        pair_off[g + 1] = pair_off[g] + m * (m + 1) / 2;
        mat_off[g + 1] = mat_off[g] + m * m;
    }
    const size_t n_reads = group_off[n_groups], n_mat = (size_t)mat_off[n_groups];
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t need = pad(n_reads * 8) + pad(((size_t)n_groups + 1) * 4) + 2 * pad(((size_t)n_groups + 1) * 8) + pad(n_mat) + 4096;
    uint8_t *base = nullptr;
    SMI_HIP(hipMalloc(&base, need));
    struct Free {
        void *p;
        ~Free() { (void)hipFree(p); }
    } guard{base};
    size_t at = 0;
    auto take = [&](size_t bytes) {
        uint8_t *p = base + at;
        at += pad(bytes);
        return p;
    };
    uint64_t *d_win = reinterpret_cast<uint64_t *>(take(n_reads * 8));
    uint32_t *d_go = reinterpret_cast<uint32_t *>(take(((size_t)n_groups + 1) * 4));
    uint64_t *d_po = reinterpret_cast<uint64_t *>(take(((size_t)n_groups + 1) * 8));
    uint64_t *d_mo = reinterpret_cast<uint64_t *>(take(((size_t)n_groups + 1) * 8));
    uint8_t *d_out = take(n_mat);
    SMI_HIP(hipMemcpyAsync(d_win, windows, n_reads * 8, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_go, group_off, ((size_t)n_groups + 1) * 4, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_po, pair_off.data(), ((size_t)n_groups + 1) * 8, hipMemcpyHostToDevice, s));
    SMI_HIP(hipMemcpyAsync(d_mo, mat_off.data(), ((size_t)n_groups + 1) * 8, hipMemcpyHostToDevice, s));
    if (int rc = smi_umi_dist_device(ctx, d_win, d_go, d_po, d_mo, n_groups, pair_off[n_groups], d_out, s)) return rc;
    SMI_HIP(hipMemcpyAsync(out, d_out, n_mat, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    return SMI_OK;
}
