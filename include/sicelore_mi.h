/*
 * sicelore_mi.h -- C ABI of the MI355X-native barcode/UMI assignment path (libsicelore_mi.so).
 *
 * The reference (ucagenomix/sicelore-2.1) exposes no FFI for this path: the hot loops live inside
 * Jar/NanoporeBC_UMI_finder-2.1.jar (bytecode only, package com.rw.*) and are reached through the jar
 * CLI (quickrun-2.1.sh:35,42).  Each entry point below names the reference unit it stands in for, so a
 * Java host can call it where that unit is called today (INTEGRATION.md shows the JNI stub).
 * Citation form: FJ!pkg/Class.java:Lnn = original source line from the class's LineNumberTable inside
 * Jar/NanoporeBC_UMI_finder-2.1.jar!/com/rw/ ; TB! = Jar/lib/TwoFourBitNucAcidLibraryMaven-1.0.jar!/com/rw/.
 *
 * Conventions: plain pointers and sizes only; all functions return 0 on success and a negative
 * smi_status otherwise (message via smi_last_error(), thread-local); no exceptions cross the boundary.
 * One context per GPU; calls on one context must be serialised by the caller; different contexts may be
 * used from different threads (the reference runs these code paths from nCPU worker threads,
 * FJ!nanoporereadscanner/WorkerReadscanner.java:L188-204).
 * "_device" entry points take device pointers (hipMalloc'ed or a torch tensor's data_ptr()) and a
 * hipStream_t passed as void*; they enqueue work and return without synchronising.
 * "_batch" entry points take host buffers, copy, run, synchronise and copy back.
 * There is no CPU fallback anywhere in this library.
 */
#ifndef SICELORE_MI_H
#define SICELORE_MI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SMI_OK = 0,
    SMI_ERR_INVALID = -1,  /* bad argument (null pointer, size, unsupported knob) */
    SMI_ERR_HIP = -2,      /* a HIP runtime call failed */
    SMI_ERR_NO_DEVICE = -3,/* no gfx950 device visible */
    SMI_ERR_STATE = -4     /* call order (e.g. match before a barcode set was loaded) */
} smi_status;

typedef struct smi_ctx smi_ctx;

/* 2-bit base code of the reference: A=0 G=1 C=2 T=3 (TB!nuc/encoding/TwoBit/NucleicAcidTwoBitPerBase.java:L78-87). */

/* ---- barcode window handed to the matcher --------------------------------------------------------
 * Everything Parser.assignBarcode (FJ!nanoporereadscanner/analyzers/Parser.java:L195-242) reads from the
 * stranded read for the five tested offsets (testPlusMinusPos = 2, Jar/config.xml:35):
 *   3' protocol: W = 24 bases stranded[AE-22 .. AE+1]   (16-mer windows AE-16+o..AE-1+o, post bases
 *                substring(bcStart-5, bcStart), Parser.java:L206-207,L214,L218)
 *   5' protocol: W = 25 bases stranded[AE-1 .. AE+23]   (windows AE+1+o..AE+16+o, post bases
 *                substring(bcEnd, bcEnd+5), Parser.java:L209-210,L219)
 * AE = adapter_result.end, 1-based.  Base j (0-based inside the window) sits in bits
 * [2*(W-1-j)+1 : 2*(W-1-j)] of `bases`; bit j of `nmask` is set when that base is not A/C/G/T
 * (its 2-bit code is then ignored).  flags bit 0 = window is valid (all W bases inside the read);
 * where it is not the reference throws StringIndexOutOfBoundsException and the result is flagged. */
typedef struct {
    uint64_t bases;
    uint32_t nmask;
    uint32_t flags;
} smi_bc_window;
#define SMI_WIN_VALID 1u
#define SMI_WIN_BASES_3P 24
#define SMI_WIN_BASES_5P 25

/* ---- result of Parser.assignBarcode (Parser.java:L244-311) ----------------------------------------
 * found: 1 = barcode accepted (BC_FOUND), 0 = none/ambiguous, -1 = window invalid (reference would throw)
 * bc: matching barcode, 2-bit, first base most significant (16 nt -> 32 bits)
 * ed / ed_sec: editDistance and editDistanceSecondBest (INT32_MAX when there is no second barcode, L288)
 * offset: offsetFromPredicted of the best match; ins_minus_del: OneMatch.getOffsetForReadEnd (L533), so
 *   3': bcStart = AE-1+offset, bcEnd = bcStart-15-ins_minus_del ; 5': bcStart = AE+1+offset, bcEnd = bcStart+15+ins_minus_del
 * n_matches: size of the merged Matches set (diagnostic). */
typedef struct {
    uint32_t bc;
    int32_t ed_sec;
    int8_t found;
    int8_t ed;
    int8_t offset;
    int8_t ins_minus_del;
    uint32_t n_matches;
} smi_bc_result;

typedef enum {
    SMI_SET_USED_LIST = 0, /* pass-2 search set = used-barcode list (WorkerReadscanner$BarcodesMapForBCfinding) */
    SMI_SET_WHITELIST = 1  /* -g/--cellRangerBCs: search set = the whole list (NanoporeReadScannerMain.java:L300-302) */
} smi_set_mode;

const char *smi_last_error(void);
const char *smi_version(void);

/* one context per GPU (device ordinal as seen by HIP) */
int smi_ctx_create(int device, smi_ctx **out);
int smi_ctx_destroy(smi_ctx *ctx);
int smi_ctx_device(const smi_ctx *ctx);

/* Replaces the Set<Long> handed to BarcodeMatchTester (hashMapForBCfinding.keySet(), Parser.java:L234) and the
 * LongOpenHashSet of all possible barcodes used by pass 1 (NanoporeReadScannerMain.readBarcodesFile L480-503).
 * keys: n 16-nt barcodes, 2-bit packed in the low 32 bits.  Builds the HBM-resident membership pyramid. */
int smi_set_barcode_set(smi_ctx *ctx, const uint64_t *keys, size_t n, int mode);
int smi_set_barcode_set_device(smi_ctx *ctx, const uint32_t *d_keys, size_t n, int mode, void *stream);

/* Replaces BarcodeMatchTester.call for the 5 offsets + the best/second rule of Parser.assignBarcode
 * (BarcodeMatchTester.java:L198-374, Parser.java:L203-311).  max_ed in {0,1,2}; five_prime = 1 for -h/--fivePbc. */
int smi_bc_match_batch(smi_ctx *ctx, const smi_bc_window *windows, size_t n, int max_ed, int five_prime,
                       smi_bc_result *out);
int smi_bc_match_device(smi_ctx *ctx, const smi_bc_window *d_windows, size_t n, int max_ed, int five_prime,
                        smi_bc_result *d_out, void *stream);

/* Window extraction = the substring()/2-bit packing half of Parser.lambda$assignBarcode$4 (Parser.java:L205-221).
 * reads: stranded reads, 1 byte per base (ASCII), concatenated; read i occupies [offsets[i], offsets[i+1]);
 * adapter_end[i] = AE (1-based) or <= 0 when no adapter was found (window flagged invalid). */
int smi_extract_windows_device(smi_ctx *ctx, const uint8_t *d_reads, const uint64_t *d_offsets,
                               const int32_t *d_adapter_end, size_t n, int five_prime, smi_bc_window *d_windows,
                               void *stream);

/* Replaces UsedCellBCListGenerator$Worker's membership test + histogram increment
 * (FJ!nanoporereadscanner/analyzers/UsedCellBCListGenerator.java:L207-229): for every read with pass[i] != 0,
 * if keys[i] is in the loaded set, ++hist[ordinal(keys[i])].  Ordinals are positions in the key array given to
 * smi_set_barcode_set (first occurrence).  hist must hold n_set uint32 counters. */
int smi_hist_device(smi_ctx *ctx, const uint32_t *d_keys, const uint8_t *d_pass, size_t n, uint32_t *d_hist,
                    void *stream);

/* device-time of the dominant kernel of the last *_device call on this context, measured with HIP events on the
 * stream the kernel was launched on; valid after the stream has been synchronised.  ms <= 0: not available. */
int smi_last_kernel_ms(smi_ctx *ctx, float *ms);
int smi_set_timing(smi_ctx *ctx, int enabled);

#ifdef __cplusplus
}
#endif
#endif
