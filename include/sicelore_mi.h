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
#define SMI_WIN_5P 2u /* the record is a 25-base 5' window (set by smi_scan_device in 5' mode; informational) */
#define SMI_WIN_BASES_3P 24
#define SMI_WIN_BASES_5P 25

/* ---- result of Parser.assignBarcode (Parser.java:L244-311) ----------------------------------------
 * found: 1 = barcode accepted (BC_FOUND), 0 = none/ambiguous, -1 = window invalid (reference would throw),
 *        -2 = (ed 2 only) >= 11 matches share one HashSet bucket: the reference's tree bin orders them by
 *        System.identityHashCode, so its own answer is not reproducible (needs >= 4 identical windows, i.e. homopolymers)
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
    SMI_SET_WHITELIST = 1, /* -g/--cellRangerBCs: search set = the whole list (NanoporeReadScannerMain.java:L300-302) */
    SMI_SET_MEMBERSHIP = 2 /* the list of all possible barcodes as pass 1 uses it (UsedCellBCListGenerator.java:L207-229: exact membership + histogram): the
                            * pyramid only, none of the matchers' neighbourhood structures (3.6 M barcodes: ~ 10 ms instead of ~ 300).  A matcher call on
                            * such a set still answers, through the pyramid kernels */
} smi_set_mode;

const char *smi_last_error(void);
const char *smi_version(void);

/* one context per GPU (device ordinal as seen by HIP) */
int smi_ctx_create(int device, smi_ctx **out);
int smi_ctx_destroy(smi_ctx *ctx);
int smi_ctx_device(const smi_ctx *ctx);
/* `scanfastq -p <polyAlength> -f <fractionAT> -w <windowAT>` (NanoporeReadScannerMain.java:L227-234): the polyA / polyT finder of the per-chunk
 * workers of this context (smi_scanfastq_pass{1,2}_chunk[_packed], _keys) runs with these values instead of config.xml's 15 / 0.75 / 150 (0 keeps
 * the shipped one); the chimera splitter's distance from the read ends follows the window.  Lanes take the values of their owner when they are
 * created or refreshed.  Limits of this build (175 scanned bases per end): 5 <= length <= 30, window + length + 10 <= 175. */
int smi_ctx_set_polya(smi_ctx *ctx, int polya_len, float polya_frac, int window_polya);

/* The knobs of Jar/config.xml that shape the per-read algorithms, as RUN-TIME parameters of a context (round 6): what
 * ParametersReadScannerApp / ParametersBarcodeUMiFinderAppParams hold after JAXB has read the file (FJ!nanoporereadscanner/parameters/
 * ParametersReadScannerApp.java:L84-130, FJ!parameters/{ReadScannerParameters,PolyATparameters,AdapterParameters,Adapter5p_5pBC_Parameters,
 * Adapter3p_5pBC_Parameters,TSOparameters_3pBarcoding,UMIparameters}.java).  Every chunk worker of the context (smi_scanfastq_pass{1,2}_chunk[_packed],
 * _keys) builds its scan / splitter configuration from these instead of the shipped values; lanes take their owner's when they are created
 * or refreshed.  Where each knob acts (reference unit -> kernel):
 *   min_read_length               PolyATadapterAnalyzerBase.java:L109-118 (READ_TOO_SHORT)                              K-SCAN
 *   min_mean_bc_qv / _read_qv / min_adapter_3p_matches   UsedCellBCListGenerator$Worker.java:L198-202 (pass 1 only)       K-SCAN
 *   polya_len / polya_frac / window_polya                PolyATSearcher.java:L56-252 (also -p -f -w: smi_ctx_set_polya wins) K-SCAN, K-CHIM
 *   internal_pat_len / internal_pat_frac                 PolyATadapterInternalSearcherBase.java:L78-270                   K-CHIM
 *   adapter3p / adapter3p_complete / adapter3p_max_mm    Parser.java:L134-136 (pass 2: sequence, pass 1: sequence_complete; both with
 *                                                        maxNeedlemanMismatches)                                           K-SCAN
 *   adapter3p_complete_max_mm, tso_complete, tso_complete_max_mm   ChimeraFindernew.java:L74-81 (3' barcoding)              K-CHIM
 *   adapter5p*, adapter5p_window                         PolyATadapterAnalyzer_5pBCUMI.java:L43-76; max mismatches + 1 (Parser.java:L99) K-SCAN
 *   adapter5p_complete + _max_mm, adapter3p5_complete + _max_mm    ChimeraFindernew.java:L75-78 (5' barcoding)              K-CHIM
 *   umi_length                                           ChimeraFindernew.java:L74 (cell_bc_length + umi_length between polyA and adapter) K-CHIM;
 *                                                        the UMI stage takes it from smi_assignumis_config
 * Limits of this build (checked by smi_ctx_set_knobs, message names the knob): sequence 10 bases, sequence_complete 22, the complete TSO 27,
 * the complete 3' adapter of 5' barcoding 25, A / C / G / T only; the polyA limits of smi_ctx_set_polya; 8 <= umi_length <= 12;
 * counts and mismatch limits 0 .. 30; the TSO of the read scan 16 bases, its window 16 .. 112.  A TSO, an adapter or a polyA window other than
 * the shipped ones run K-SCAN's generic kernels (every candidate aligned over the full matrix, the finder as a loop), the shipped ones the
 * kernels specialised for them; the results are the same function of the parameters (tests run both).  Compiled in (no knob):
 * testPlusMinusPos = 2, cell_bc_length = 16, the read-name prefixes, nbasesOfAdapterSeqInReadname = 3. */
typedef struct {
    int32_t min_read_length;          /* readscanner/minReadLength                              config.xml:21   200 */
    int32_t min_mean_bc_qv;           /* readscanner/minMeanBCqv                                :55   8 */
    int32_t min_mean_read_qv;         /* readscanner/minMeanReadqv                              :57   8 */
    int32_t min_adapter_3p_matches;   /* readscanner/minAdapter3pMatches                        :59   8 */
    int32_t polya_len;                /* polyAT/polyATlength                                    :95   15 */
    float polya_frac;                 /* polyAT/fractionATInPolyAT                              :97   0.75 */
    int32_t window_polya;             /* polyAT/windowSearchForPolyA                            :105  150 */
    int32_t internal_pat_len;         /* polyAT/internalpATlength                               :99   15 */
    float internal_pat_frac;          /* polyAT/internalFractionATInPolyAT                      :101  0.70 */
    char adapter3p[32];               /* adapter_for3pBarcoding/sequence                        :111  CTTCCGATCT */
    char adapter3p_complete[32];      /* adapter_for3pBarcoding/sequence_complete               :113  CTACACGACGCTCTTCCGATCT */
    int32_t adapter3p_max_mm;         /* adapter_for3pBarcoding/maxNeedlemanMismatches          :115  3 */
    int32_t adapter3p_complete_max_mm;/* adapter_for3pBarcoding/maxCompleteSeqNeedlemanMismatches :118 5 */
    char adapter5p[32];               /* fiveprimeadapter_for5pBarcoding/sequence               :124  CTTCCGATCT */
    char adapter5p_complete[32];      /* fiveprimeadapter_for5pBarcoding/sequence_complete      :126  CTACACGACGCTCTTCCGATCT */
    int32_t adapter5p_max_mm;         /* fiveprimeadapter_for5pBarcoding/maxNeedlemanMismatches :129  3 */
    int32_t adapter5p_complete_max_mm;/* .../maxCompleteSeqNeedlemanMismatches                  :132  5 */
    int32_t adapter5p_window;         /* .../AdapterSearchWindow                                :134  110 */
    char adapter3p5_complete[32];     /* threeprimeadapter_for5pBarcoding/sequence_complete     :141  AAGCAGTGGTATCAACGCAGAGTAC */
    int32_t adapter3p5_complete_max_mm;/* .../maxCompleteSeqNeedlemanMismatches                 :146  5 */
    char tso_complete[32];            /* tso_for3pBarcoding/sequence_complete                   :170  AAGCAGTGGTATCAACGCAGAGTACAT */
    int32_t tso_complete_max_mm;      /* tso_for3pBarcoding/maxCompleteSeqNeedlemanMismatches   :172  6 */
    int32_t umi_length;               /* umis/umi_length                                        :264  12 */
    char tso_scan[20];                /* tso_for3pBarcoding/sequence                            :155  AACGCAGAGTACATGG (16 bases) */
    int32_t tso_scan_max_mm;          /* tso_for3pBarcoding/maxNeedlemanMismatches              :157  5 */
    int32_t tso_scan_min_consec;      /* tso_for3pBarcoding/minTSO_NeedlemanConsecutiveMatches  :161  8 */
    int32_t tso_scan_min_two_best;    /* tso_for3pBarcoding/minTSO_TwoBestConsecutiveMatches    :164  12 */
    int32_t tso_scan_window;          /* tso_for3pBarcoding/windowForTSOsearch                  :166  90 */
    int32_t reserved[6];
} smi_run_knobs;
/* `scanfastq -e / --randomBarcode` (NanoporeReadScannerMain.java:L212-215; Parser.java:L212-215: "the read bc sequence gets replaced by a random sequence",
 * /root/reference/README.md:176): the specificity experiment -- pass 2 of this context's chunk workers matches RANDOM sequences where the read's barcode
 * windows stood, so every barcode it still assigns is a chance assignment.  seed != 0 switches it on, 0 off; lanes take their owner's when created or
 * refreshed.  The reference draws five independent random 16-mers per read from an unseeded java.util.Random (not reproducible); here the 24 / 25 window
 * bases of a read become one sequence drawn from (seed, read id), the five windows are cut from it as they are from a read -- each window is uniformly
 * random, so the expected number of chance assignments is the reference's, and a run can be repeated. */
int smi_ctx_set_random_barcodes(smi_ctx *ctx, uint64_t seed);
/* the shipped config.xml */
int smi_run_knobs_default(smi_run_knobs *knobs);
/* knobs == NULL: back to the shipped values.  SMI_ERR_INVALID (smi_last_error names the knob) for a value this build has no kernel for. */
int smi_ctx_set_knobs(smi_ctx *ctx, const smi_run_knobs *knobs);
int smi_ctx_get_knobs(const smi_ctx *ctx, smi_run_knobs *knobs);
/* A worker lane of `owner`: a context with its own stream, device arena, pinned output buffers and timing that READS the owner's barcode
 * set instead of holding the 616 MiB membership pyramid and the neighbourhood bitmaps and table (up to 18 GB for the whole whitelist) again -- several host threads, one lane each, overlap their uploads, kernels and
 * downloads on one GPU over ONE set, as the reference's nCPU Parser workers share one hashMapForBCfinding
 * (FJ!nanoporereadscanner/WorkerReadscanner.java:L188-204).  Every entry point takes a lane except smi_set_barcode_set*: load the set on the
 * owner (no lane busy meanwhile), then smi_ctx_lane_refresh each lane.  Destroy the lanes before the owner. */
int smi_ctx_create_lane(smi_ctx *owner, smi_ctx **out);
int smi_ctx_lane_refresh(smi_ctx *lane);

/* Replaces the Set<Long> handed to BarcodeMatchTester (hashMapForBCfinding.keySet(), Parser.java:L234) and the
 * LongOpenHashSet of all possible barcodes used by pass 1 (NanoporeReadScannerMain.readBarcodesFile L480-503).
 * keys: n 16-nt barcodes, 2-bit packed in the low 32 bits.  Builds the HBM-resident membership pyramid (616 MiB) and, from the set's
 * inverse one-step neighbourhood (169 sequences per barcode), what lets the matchers skip the reference's mutant enumeration without
 * changing a result: an exact bitmap of that neighbourhood (512 MiB) and a table of it with the mutation step back to the barcode in every
 * entry (three eight-byte slots per neighbour: 14.6 GB for 3.6 M barcodes, 20 MB for 5 k; 1.6 slots or none at all where the device cannot spare them); for lists of up to
 * 65,536 / 32,768 barcodes also the item filter and the two-step bitmap of the ed <= 2 matcher (32 MiB + a 512 MiB scratch).  Round 5: the neighbourhood bitmap
 * a second time in the layout that makes a read's five probes neighbours (2.5 GiB) and the table at three slots per entry (14.6 GB for 3.6 M barcodes).  3.6 M barcodes
 * load in ~300 ms (SMI_SET_MEMBERSHIP: ~10 ms), a used list in a few ms.  Results never depend on these structures ("Switches"). */
int smi_set_barcode_set(smi_ctx *ctx, const uint64_t *keys, size_t n, int mode);
int smi_set_barcode_set_device(smi_ctx *ctx, const uint32_t *d_keys, size_t n, int mode, void *stream);
/* What the last smi_set_barcode_set[_device] of this context built (round 6: the one-time cost behind the matchers, for `bench.py`'s set_build_ms /
 * set_hbm_bytes and for cross-checks of the build kernels): out[0] distinct keys, out[1] bytes of device memory the structures of the loaded set
 * occupy (pyramid + nb + nb5 + nt + n1 / n2 / nb2 as far as they are valid for this set), out[2] wall time of the build in microseconds (launch to
 * drained stream), out[3] bits set in nb, out[4] bits set in nb5, out[5] an order-sensitive digest of nb5 (sum of popcount(word) * (index mod 2^20 + 1)),
 * out[6] slots of nt, out[7] entries in nt.  With digests != 0 the counting kernels run (a few ms for the whole whitelist); 0 leaves out[3..7] zero. */
int smi_set_stats(smi_ctx *ctx, uint64_t out[8], int digests);

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
 * if keys[i] is in the loaded set, ++hist[ordinal(keys[i])].  ordinal(key) = rank of the key in ascending key order among the
 * distinct keys given to smi_set_barcode_set.  hist must hold n_distinct uint32 counters. */
int smi_hist_device(smi_ctx *ctx, const uint32_t *d_keys, const uint8_t *d_pass, size_t n, uint32_t *d_hist,
                    void *stream);


/* ================================================================================================================
 * Read scan (3' protocol): polyA/T finder + k-mer gated Needleman-Wunsch adapter scan + strand decision.
 * Replaces PolyATadapterAnalyzer_3pBCUMI.search incl. the TSO scan
 * (FJ!nanoporereadscanner/analyzers/PolyATadapterAnalyzer_3pBCUMI.java:L45-190 ->
 *  FJ!nanopore/analyzers/PolyATadapterAnalyzerBase.java:L109-319, PolyATSearcher.java:L56-252,
 *  AdapterTSOanalyzer.java:L84-110) and, with qualities, the pass-1 filter of
 * UsedCellBCListGenerator$Worker (UsedCellBCListGenerator.java:L198-202).
 *
 * Device read batch ("ends"): for read i the head (first SMI_END_BASES bases) is end 2i, the reverse complement of
 * its last SMI_END_BASES bases is end 2i+1 -- the two orientations PolyATSearcher scans (L178-181).  An end is
 * stored as four bit-planes (bit c of the reference's 4-bit IUPAC code A=1 G=2 C=4 T=8 N=15,
 * TB!nuc/encoding/NucleicAcidByteCodeBase.java:L45-78) of SMI_PLANE_WORDS 32-bit words; word w of plane c of end e
 * lives at ends[(c * SMI_PLANE_WORDS + w) * (2 n) + e], bit p%32 of word p/32 = base p.  Bases beyond the read are 0.
 * ================================================================================================================ */
#define SMI_END_BASES 224
#define SMI_PLANE_WORDS 7
#define SMI_ENDS_ROWS (4 * SMI_PLANE_WORDS)

/* flag bits = ReadFlags$Flags.getValue() of FJ!nanoporereadscanner/stats/ReadFlags$Flags (ReadFlags.java:L72-109): FAILED = 0x20,
 * PASSED_FWD = 0x100, ... (NOT the enum ordinals: ALL_READS_AFTER_SPLIT = -1 sits in between; values pinned by
 * tests/golden/ref_exec_pass2_*.json `flag_values`, read from the reference's own enum) */
#define SMI_F_FAILED (1u << 5)
#define SMI_F_PASSED_FWD (1u << 8)
#define SMI_F_PASSED_REV (1u << 9)
#define SMI_F_POLY_T_5P (1u << 11)
#define SMI_F_POLY_A_3P (1u << 12)
#define SMI_F_POLY_A_NOT_FOUND (1u << 13)
#define SMI_F_POLY_T_5P_POLY_A_3P (1u << 14)
#define SMI_F_ADAPTER_5P (1u << 15)
#define SMI_F_ADAPTER_3P (1u << 16)
#define SMI_F_ADAPTER_SELECTED_DESP_BOTH (1u << 19)
#define SMI_F_READ_TOO_SHORT (1u << 20)
#define SMI_F_ADAPTER_5P_AND_3P (1u << 21)
#define SMI_F_TSO_5P (1u << 17)
#define SMI_F_TSO_3P (1u << 18)
#define SMI_F_TSO_5P_AND_3P (1u << 22)

typedef struct {
    int32_t min_read_length;        /* Jar/config.xml:21   200 */
    int32_t polya_len;              /* :95  15 */
    float polya_frac;               /* :97  0.75 */
    int32_t window_polya;           /* :105 150 */
    int32_t max_mismatches;         /* :115 maxNeedlemanMismatches 3 */
    int32_t min_adapter_3p_matches; /* :59  8 */
    int32_t min_mean_bc_qv;         /* :55  8 */
    int32_t min_mean_read_qv;       /* :57  8 */
    int32_t adapter_len;            /* 10 (pass 2, "CTTCCGATCT") or 22 (pass 1, complete adapter), Parser.java:L135 */
    uint32_t adapter4[22];          /* 4-bit codes of the adapter */
    int32_t five_prime;             /* 1: 5' barcoding, PolyATadapterAnalyzer_5pBCUMI.search (PolyATadapterAnalyzer_5pBCUMI.java:L43-76) */
    int32_t dont_search_polya;      /* 5' only: --noPolyARequired (dontSearchPolyAFor5pBarcoding) */
    int32_t adapter_search_window;  /* 5' only: AdapterSearchWindow, config.xml:134 (110) */
    /* the TSO scan of 3' barcoding (PolyATadapterAnalyzer_3pBCUMI.scanReadForTSOs L122-190, PolyATadapterAnalyzerBase.scanForTSO L324-369);
     * tso_window == 0: the shipped parameters (a configuration built by hand before round 6 leaves these zero) */
    uint32_t tso4[16];              /* 4-bit codes of tso_for3pBarcoding/sequence (16 bases), :155 */
    int32_t tso_window;             /* windowForTSOsearch :166 (90); 16 .. 112 */
    int32_t tso_max_mismatches;     /* maxNeedlemanMismatches :157 (5) */
    int32_t tso_min_consec;         /* minTSO_NeedlemanConsecutiveMatches :161 (8) */
    int32_t tso_min_two_best;       /* minTSO_TwoBestConsecutiveMatches :164 (12) */
} smi_scan_config;

typedef struct {
    uint32_t flags;
    int32_t adapter_start, adapter_end; /* AS / AE: stranded, 1-based (ReadScanResult.java:L445-447); 0 = none */
    int32_t polya_start, polya_end;     /* PS / PE (ReadScanResult.java:L346,L356) */
    int16_t scan_end;                   /* adapter end in scan orientation */
    int16_t tso_start;                  /* TSOresult.start (0 = null) */
    int16_t adapter_nmis;               /* NeedlemanMatch.getNerrorsNeedleman of the accepted alignment */
    int8_t found;                       /* adapterFound() */
    int8_t reverse;                     /* 1: stranded read = reverse complement of the raw read (PASSED_REV) */
    int8_t pass1_ok;                    /* pass-1 quality filter passed (only when qualities were given) */
    int8_t reserved;
    int16_t tso_end;                    /* TSOresult.end (0 = null): the read name's T= field */
} smi_scan_result;

/* fills cfg with the shipped config.xml values; pass = 1 (complete adapter) or 2 (short adapter) */
int smi_scan_default_config(int pass, smi_scan_config *cfg);
/* the same for 5' barcoding: fiveprimeadapter_for5pBarcoding (config.xml:122-135), max_mismatches =
 * maxNeedlemanMismatches + 1 (Parser.java:L99), window 110 */
int smi_scan_default_config_5p(int pass, int dont_search_polya, smi_scan_config *cfg);

/* ASCII reads (concatenated, read i = [offsets[i], offsets[i+1])) -> ends / lengths (+ right-aligned tail qualities
 * [n][SMI_END_BASES] and sum of (q-33) per read when quals != NULL; with five_prime the FIRST SMI_END_BASES qualities,
 * left-aligned, because the pass-1 filter then reads positions near the read start).  The packing half of FastqRecordExt /
 * NucleicAcidOneBytePerBase construction (PolyATSearcher.java:L178-181). */
int smi_pack_ends_device(smi_ctx *ctx, const uint8_t *d_reads, const uint8_t *d_quals, const uint64_t *d_offsets,
                         size_t n, int five_prime, uint32_t *d_ends, int32_t *d_read_len, uint8_t *d_qtail,
                         uint32_t *d_qsum, void *stream);

/* smi_scan_batch: the same from host buffers (ASCII bases, optional ASCII qualities, offsets[n_reads + 1]): upload, K-PACK, K-SCAN, download;
 * out_windows may be NULL */
int smi_scan_batch(smi_ctx *ctx, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, size_t n_reads,
                   const smi_scan_config *cfg, smi_scan_result *out, smi_bc_window *out_windows);
/* d_qtail / d_qsum may be NULL (pass 2); d_windows may be NULL.  windows[i] is the smi_bc_window of read i (valid flag
 * clear when no adapter was found), ready for smi_bc_match_device. */
int smi_scan_device(smi_ctx *ctx, const uint32_t *d_ends, const int32_t *d_read_len, const uint8_t *d_qtail,
                    const uint32_t *d_qsum, size_t n, const smi_scan_config *cfg, smi_scan_result *d_out,
                    smi_bc_window *d_windows, void *stream);

/* The one exchange of the path for a single-process host that drives several GPUs -- the shape of the reference, one JVM:
 * element-wise sum of the pass-1 histograms of n_ctx contexts, one per GPU, in place, over RCCL (xGMI).  d_hist[i]: n_counters u32 on the
 * device of ctxs[i] (n_counters = number of loaded keys; every context holds the same barcode set).  Runs on the contexts' own streams and
 * returns when all of them have drained; RCCL is dlopen-ed on first use.  The reference's threads add into one ConcurrentHashMap instead
 * (UsedCellBCListGenerator.java:L224-229); multi-process hosts use their own collective (sicelore-2.1_amd/distributed.py).
 * Stream order: a context's stream is NOT ordered against the stream that filled d_hist[i].  smi_hist_allreduce expects every producer
 * stream to have been synchronised by the caller (smi_scanfastq_pass1_chunk returns drained, so its histograms qualify);
 * smi_hist_allreduce_after takes the producer streams (hipStream_t as void*, one per context, NULL entry = the device's default stream) and
 * orders each context's stream behind its producer with an event.  The calling thread's current HIP device is restored on return.  The
 * communicators are created on the first call for a device list and kept (ncclCommInitAll is hundreds of ms on 8 GPUs);
 * smi_hist_allreduce_release destroys them (call it before the contexts go when the process outlives them). */
int smi_hist_allreduce(smi_ctx **ctxs, int n_ctx, uint32_t **d_hist, size_t n_counters);
int smi_hist_allreduce_after(smi_ctx **ctxs, int n_ctx, uint32_t **d_hist, size_t n_counters, void *const *producer_streams);
int smi_hist_allreduce_release(void);

/* pass-1 histogram straight from scan output: for reads with pass1_ok, key = offset-0 barcode of the window
 * (UsedCellBCListGenerator.java:L207-229); ++hist[ordinal(key)] when the key is in the loaded set */
int smi_hist_windows_device(smi_ctx *ctx, const smi_bc_window *d_windows, const smi_scan_result *d_scan, size_t n,
                            uint32_t *d_hist, void *stream);
/* pass 1 WITHOUT a list of possible barcodes (`-a none`: ReadScannerParameters.generateUsedBarcodesWithoutWhitelist,
 * NanoporeReadScannerMain.java:L132-133; the membership predicate of UsedCellBCListGenerator$Worker.call is "true", L255-256): the barcode of
 * every read with pass1_ok is APPENDED to d_keys as the reference's long (a 5' barcode with an N: bits 63..32 set, ref_exec_pass1_nowl_5p.json);
 * *d_count (device, u64, zeroed by the caller) counts every append, entries beyond cap are dropped (the caller checks count <= cap).
 * smi_count_keys_device turns the list of a pass into distinct keys (ascending) and counts: the input of smi_finalize_used_list. */
int smi_pass1_keys_device(smi_ctx *ctx, const smi_bc_window *d_windows, const smi_scan_result *d_scan, size_t n, uint64_t *d_keys, size_t cap,
                          uint64_t *d_count, void *stream);
int smi_count_keys_device(smi_ctx *ctx, const uint64_t *d_keys, size_t n, uint64_t *d_unique, uint32_t *d_counts, uint64_t *d_n_unique, void *stream);

/* Pass-2 counters per barcode and edit distance = assignedBarcodes2ndPass[bc].addCountForEd(ed) (Parser.java:L305-311,
 * Parser$BarcodeCounts L339-354): d_counts[3 * ordinal(bc) + ed] += 1 for every result with found == 1; ordinal = index of the
 * barcode in the ascending key list of the loaded set (the index smi_hist_device uses), so the vector is dense and is summed
 * across GPUs with one all-reduce.  d_counts: 3 * n_keys u32, zeroed by the caller before the first batch. */
int smi_bc_counts_device(smi_ctx *ctx, const smi_bc_result *d_results, size_t n, uint32_t *d_counts, void *stream);
/* BarcodeList.tsv (ParseStatsHtmlPrinter.writesedBarcodesListTSV, FJ!nanoporereadscanner/stats/ParseStatsHtmlPrinter.java:L235-285) from the
 * same inputs as smi_finalize_used_list: the used barcodes in rank order with their pass-1 counts and, per edit distance that occurs, the
 * barcode each collides with -- `BC(count x)` when that one is in the used list too, `BC(count m)` when it was merged away.  no_whitelist:
 * the run had no list of possible barcodes (rows with AAAAA / TTTTT are left out, UsedCellBCListGenerator.java:L419).  out == NULL: size only.
 * Host only. */
int smi_barcode_list_tsv(const uint64_t *keys, const uint32_t *counts, size_t n, uint32_t record_count, int merge_ed, int min_count_fold,
                         int cells_fold_below_max, int no_whitelist, char *out, size_t cap, size_t *n_out);
/* BarcodesAssigned.tsv from those counters (ParseStatsHtmlPrinter.writeAssignedTSV, ParseStatsHtmlPrinter.java:L294-327);
 * keys ascending as loaded; rows with equal counts by ascending key (the reference: HashMap order).  out == NULL: size only. */
int smi_assigned_tsv(const uint64_t *keys, const uint32_t *counts, size_t n_keys, int max_ed, char *out, size_t cap,
                     size_t *n_out);

/* End of pass 1 on the host (no device work): low-count filter, collision merge, low-depth cut and rank of the
 * used-barcode list = UsedCellBCListGenerator$UsedBarcodesListData.finalizeData
 * (FJ!nanoporereadscanner/analyzers/UsedCellBCListGenerator.java:L379-425), BarcodeDatasetColissionTester
 * (BarcodeDatasetColissionTester.java:L68-229) and the rank assignment of WorkerReadscanner.scan
 * (FJ!nanoporereadscanner/WorkerReadscanner.java:L265-273).
 * keys/counts: the non-zero entries of the (all-reduced) pass-1 histogram; record_count = number of 10,000-read
 * chunks seen in pass 1 (the reference's cutoff uses the chunk count, L254,L391); merge_ed = mergeBCsED
 * (defaults to --bcEditDistance); min_count_fold = 10, cells_fold_below_max = 500 (Jar/config.xml:61,27).
 * Outputs (capacity n each): barcodes sorted by count descending (equal counts: ascending key -- canonical order,
 * see DESIGN.md), their counts and 1-based ranks (the rk= field). */
int smi_finalize_used_list(const uint64_t *keys, const uint32_t *counts, size_t n, uint32_t record_count,
                           int merge_ed, int min_count_fold, int cells_fold_below_max, uint64_t *out_keys,
                           uint32_t *out_counts, uint32_t *out_rank, size_t *n_out);

/* ================================================================================================================
 * UMI pair distances of `assignumis`: replaces ClusteringEditDistanceBase.generateDistanceMatrix
 * (FJ!clustering/ClusteringEditDistanceBase.java:L168-259; per pair calcEditDistances L297-350 + calcBestEditDistance
 * L67-80 + apachemod/LevenshteinDistance.limitedCompare, threshold 4).
 * windows[r]: 14 bases of read r as 4-bit codes (A=1 G=2 C=4 T=8 N=15), base k in bits [4k+3:4k]: the bases
 * bcEnd .. bcEnd+13 (1-based bcEnd = barcode end on the tested read-name sequence, FastqRecordExt.java:L378), i.e. the
 * three 12-mers getSubSequence(bcEnd+1+i, 12), i = -1,0,+1.  With another umis/umi_length L on the context (smi_ctx_set_knobs; 8 .. 12): L + 2
 * bases, the three L-mers (ClusteringEditDistanceBase.java:L316-329 cut params.umis.umi_length bases).
 * Groups = (cell barcode, genomic region) sets of reads (UmiClustering.java:L105): group g owns reads
 * [group_off[g], group_off[g+1]).  pair_off[g] = sum over earlier groups of n(n+1)/2, mat_off[g] = sum of n^2.
 * out + mat_off[g] is the n x n byte matrix of group g: ed | pos1 << 4 | pos2 << 6, ed in 0..5 (5 = above the
 * threshold), pos = PlusMinusOneEnum.getValue(): 0 MINUSONE, 1 ZERO, 2 PLUSONE; [v][i] holds the transposed copy. */
int smi_umi_dist_device(smi_ctx *ctx, const uint64_t *d_windows, const uint32_t *d_group_off,
                        const uint64_t *d_pair_off, const uint64_t *d_mat_off, uint32_t n_groups,
                        uint64_t total_pairs, uint8_t *d_out, void *stream);
/* The same with the layout the chunk worker (smi_assignumis_chunk) gives its own matrices (round 6): a group of more than 64 reads has rows of
 * smi_umi_padded_row(n) = n rounded up to 64 bytes, and every group starts on a 64-byte boundary -- mat_off[g] = sum over earlier groups of
 * smi_umi_padded_bytes(n); cell [i][v] of group g = out[mat_off[g] + i * smi_umi_padded_row(n) + v].  Every 64-byte row piece a tile of the kernel writes
 * is then one whole line; with dense rows (a row starts at any byte) the pieces' first and last lines were shared with the neighbouring tile and went out
 * twice: 1.55 x the matrix bytes written (profiles/r05/umi_pmc.json), now about 1.0 x. */
uint64_t smi_umi_padded_row(uint32_t n);
uint64_t smi_umi_padded_bytes(uint32_t n);
int smi_umi_dist_device_padded(smi_ctx *ctx, const uint64_t *d_windows, const uint32_t *d_group_off, const uint64_t *d_pair_off, const uint64_t *d_mat_off,
                               uint32_t n_groups, uint64_t total_pairs, uint8_t *d_out, void *stream);
/* host buffers in and out: windows of all groups back to back, group g = reads [group_off[g], group_off[g+1]); out: the n x n matrices of
 * the groups back to back (sum of n^2 bytes), layout as above */
int smi_umi_dist_batch(smi_ctx *ctx, const uint64_t *windows, const uint32_t *group_off, uint32_t n_groups, uint8_t *out);

/* ================================================================================================================
 * UMI clustering of `assignumis` on the K-UMI matrices (host): replaces the clusterer a (cell barcode, genomic region)
 * group is handed to in UmiClustering$Submitter (FJ!umifinder/analyzers/clustering/UmiClustering$Submitter.java:L239-261):
 * ClusterOneHierarchical (groups <= 100 reads: LingPipe complete link cut at umi_completelinkclusteringED,
 * ClusterOneHierarchical.java:L66-217, lingpipe CompleteLinkClusterer.java:L146-237) or ClusterOne_MyClustering
 * (larger groups, ClusterOne_MyClustering.java:L59-219), the fold-depth filter, centre selection
 * (FJ!clustering/OneUmiCluster.java:L49-65) and the per-read values behind the tags U8 / U1 / U2
 * (ClusterOneBase.java:L118-168).  Canonical order rules where the reference is not reproducible: DESIGN.md.
 * ================================================================================================================ */
typedef struct {
    int32_t complete_link_ed;     /* umi_completelinkclusteringED, config.xml:270 (2) */
    int32_t single_link_ed;       /* umi_singlelinkclusteringED, :272 (1) */
    int32_t single_link_switch;   /* complexity_threshold_for_switch_to_single_link_clustering, :278 (3000) */
    int32_t fold_depth_below_max; /* foldDepthBelowMaxDiscardForClustering (50) */
    int32_t own_clusterer_above;  /* NRECORDS_SWITCH_TO_OWNCLUSTERING (100) */
} smi_umi_cluster_config;

typedef struct {
    int32_t center;   /* group-local index of the cluster centre whose UMI the read takes (tag U8); -1 = not clustered */
    int8_t offset;    /* -1, 0, +1: the centre's 12-mer is cut at this offset (getPostBCUMIseqOffset) */
    int8_t ed;        /* tag U1: distance to the centre */
    int8_t ed_second; /* tag U2: least distance to a read outside the cluster; -1 = tag absent */
    int8_t pos2;      /* PlusMinusOneEnum value (0 MINUSONE, 1 ZERO, 2 PLUSONE) of the read's best position vs the centre */
} smi_umi_assignment;

int smi_umi_cluster_default_config(smi_umi_cluster_config *cfg);

/* dist / mat_off / group_off: exactly the buffers of smi_umi_dist_device (copied to the host); mean_qv[r]: the read's
 * mean quality (the Q= field of its name); out[r] / skipped[r] (may be NULL; 1 = UMI_CLUSTERING_SKIPPED_HIGHCOMPLEXITY)
 * per read in group order; groups are independent and are spread over n_threads host threads. */
int smi_umi_cluster_groups(const uint8_t *dist, const uint64_t *mat_off, const uint32_t *group_off, uint32_t n_groups,
                           const float *mean_qv, const smi_umi_cluster_config *cfg, smi_umi_assignment *out,
                           uint8_t *skipped, int n_threads);

/* The same on the device for groups of up to 100 reads (K-UCLUST, one wavefront per group: ClusterOneHierarchical over LingPipe's
 * complete-link queue order); groups of more than min(own_clusterer_above, 100) reads are left "not clustered" for the caller to hand to
 * smi_umi_cluster_groups.  Buffers as smi_umi_dist_device leaves them; d_out / d_skipped per read in group order. */
int smi_umi_cluster_groups_device(smi_ctx *ctx, const uint8_t *d_dist, const uint64_t *d_mat_off, const uint32_t *d_group_off, uint32_t n_groups,
                                  const float *d_mean_qv, const smi_umi_cluster_config *cfg, smi_umi_assignment *d_out, uint8_t *d_skipped, void *stream);

/* ================================================================================================================
 * Genomic-region grouping of `assignumis` (host): which reads of a chunk may share a UMI group.  Replaces
 * ReadGrouper.groupSams / doClusteringOneStrand / ClusterList.refineClusters
 * (FJ!umifinder/bamreaders/ReadGrouper.java:L82-260,L455-785): per strand, position-sorted reads are chained while
 * consecutive gaps < max_dist, chains of > 2 reads become regions, regions are refined around their centre
 * (Math.round((float) mean position)) and close neighbours merged.
 * pos / has_pos / reverse: per read in BAM order (reverse = SAM flag 16; has_pos = positionOnGenomeForClustering
 * present); region[i] = ordinal of the read's region in the final list or -1 (the reference's ids come from a static
 * counter, only their equality matters); *n_done = leading reads of the chunk that are final -- with keep_data_end the
 * regions within 3 * max_dist of the right-most position are held back and re-grouped with the next chunk (L171-184).
 * ================================================================================================================ */
int smi_region_group(const int32_t *pos, const uint8_t *has_pos, const uint8_t *reverse, int32_t n, int32_t max_dist,
                     int keep_data_end, int32_t *region, int32_t *n_done);

/* NanoporeRead$ReadScanData.getReferencePositionAtReadPosition (FJ!umifinder/reads/nanopore/NanoporeRead$ReadScanData.java:
 * L133-153) on a BAM-encoded CIGAR (len << 4 | op, op = MIDNSHP=X): returns 1 and *out = reference position of the
 * 1-based read position, 0 for Optional.absent, or a negative smi_status. */
int smi_ref_position_at_read_position(const uint32_t *cigar, int32_t n_cigar, int32_t alignment_start, int32_t position,
                                      int32_t *out);

/* Read-name suffix of a scanned (and possibly barcode-assigned) read = FastqRecordExt.getRecordForWriting
 * (FJ!nanoporereadscanner/readerwriter/FastqRecordExt.java:L209-311): `<name>_{REV|FWD}_[PS=_PE=_][AE=_][T=_]
 * [bc=_ed=_ed_sec=_bcStart=_bcEnd=_[rk=_]]X=<stranded[AE-40..AE+2]>_Q=<##.#>_<base-36 id>[ cellBC=<bc>]`, or
 * `<name>_FAILED ` (host-side string formatting; with five_prime X= is stranded[AE-2..AE+39] and the barcode lies behind
 * the adapter).  raw_seq / raw_qual: the read as it came from the
 * FASTQ; bc may be NULL; rank <= 0 omits rk=.  Returns the length written (>= 0) or a negative smi_status. */
int smi_format_read_name(const char *read_name, const char *raw_seq, const char *raw_qual, int32_t len,
                         const smi_scan_result *scan, const smi_bc_result *bc, int32_t rank, uint32_t read_id,
                         int five_prime, char *out, size_t cap);

/* ================================================================================================================
 * Chimera splitter of pass 2 (3' barcoding): replaces ChimeraFindernew.findSplitPositions
 * (FJ!nanoporereadscanner/analyzers/ChimeraFindernew.java:L107-332, called per record from Parser.call,
 * Parser.java:L180) with PolyATadapterInternalSearcherBase.aTadapterScanBase
 * (FJ!nanopore/analyzers/PolyATadapterInternalSearcherBase.java:L78-270).
 * ================================================================================================================ */
typedef struct {                  /* shipped values: Jar/config.xml */
    const char *tso_complete;     /* :170 AAGCAGTGGTATCAACGCAGAGTACAT; this build: lengths 27 + 22 (3') or 22 + 25 (5') */
    const char *adapter_complete; /* :113 CTACACGACGCTCTTCCGATCT */
    int32_t tso_max_errors;       /* :172 6 */
    int32_t adapter_max_errors;   /* :118 5 */
    int32_t internal_pat_len;     /* :99  15 */
    float internal_pat_frac;      /* :101 0.70 */
    int32_t window_polya;         /* :105 150 */
    int32_t bc_umi_len;           /* :189 + :264 = 16 + 12 */
} smi_chimera_config;

/* ChimeraFindernew$SplitPosition$SplitReason ordinals (ChimeraFindernew.java:L364-370); tags RA FA RA_FA RA_FT RT_FA RT_FT */
enum { SMI_SPLIT_REV_ADAPTER = 0, SMI_SPLIT_FWD_ADAPTER = 1, SMI_SPLIT_RA_FA = 2, SMI_SPLIT_RA_FT = 3, SMI_SPLIT_RT_FA = 4,
       SMI_SPLIT_RT_FT = 5 };
#define SMI_CHIM_MULTI 1u    /* > 2 split positions: MULTI_CHIMERIC_READS_DISCARDED | FAILED, the read stays whole (L284-286) */
#define SMI_CHIM_RANGE 2u    /* split positions out of order / outside the read: the reference throws from substring */
#define SMI_CHIM_OVERFLOW 4u /* (round 1: more than 64 internal matches; such reads now go through the serial kernel and this flag is never set) */

typedef struct {
    int32_t pos[2];    /* cut offsets into the read (String.substring semantics): fragments [0,pos0) [pos0,pos1) [pos1,len) */
    uint8_t n_split;   /* 0, 1 or 2 */
    uint8_t reason[2]; /* SMI_SPLIT_* of each cut */
    uint8_t flags;     /* SMI_CHIM_* */
    int32_t n_matches; /* adapter / TSO matches that entered the split rules */
} smi_chimera_result;

int smi_chimera_default_config(smi_chimera_config *cfg);
/* 5' barcoding (ChimeraFindernew.java:L75-81: the 5' adapter is searched like the TSO, the 3' adapter next to internal
 * polyA/T, no barcode + UMI between them); only run when polyA is searched (Parser.java:L176) */
int smi_chimera_default_config_5p(smi_chimera_config *cfg);
/* The configurations the chunk workers of a context derive from its knobs (smi_ctx_set_knobs; knobs == NULL: the shipped file), for callers of
 * the per-stage entry points: pass 1 scans with sequence_complete, pass 2 with sequence, 5' barcoding with maxNeedlemanMismatches + 1
 * (Parser.java:L99,L134-136); the splitter's patterns and limits by protocol (ChimeraFindernew.java:L74-81).  The strings of the splitter's
 * configuration point into *knobs (or into constants when knobs == NULL). */
int smi_scan_config_from_knobs(const smi_run_knobs *knobs, int pass, int five_prime, int dont_search_polya, smi_scan_config *cfg);
int smi_chimera_config_from_knobs(const smi_run_knobs *knobs, int five_prime, smi_chimera_config *cfg);

/* u32 words the plane buffer of smi_pack_reads_device needs for n reads holding total_bases bases in all */
size_t smi_read_planes_words(uint64_t total_bases, size_t n);

/* ASCII reads (read i = [offsets[i], offsets[i+1]), offsets[n] = total_bases) -> four IUPAC bit-planes per read:
 * the NucleicAcidOneBytePerBase construction of ChimeraFindernew.java:L177 */
int smi_pack_reads_device(smi_ctx *ctx, const uint8_t *d_reads, const uint64_t *d_offsets, size_t n, uint64_t total_bases,
                          uint32_t *d_planes, void *stream);

/* split positions of every read (reads shorter than 240 bases are never split, L169) */
int smi_chimera_device(smi_ctx *ctx, const uint32_t *d_planes, const uint64_t *d_offsets, size_t n, uint64_t total_bases,
                       const smi_chimera_config *cfg, smi_chimera_result *d_out, void *stream);

/* The records after the split (L288-326) are consecutive slices of the same byte buffer, so splitting is a new offset
 * array: d_frag_offsets [n_frag + 1] (capacity 3n + 1), d_frag_src [n_frag] (capacity 3n, may be NULL) =
 * (source read << 2 | fragment index), *d_n_frag = number of records.  d_scratch: (n + 1023) / 1024 + 1 u32 words. */
int smi_split_offsets_device(smi_ctx *ctx, const smi_chimera_result *d_chim, const uint64_t *d_offsets, size_t n,
                             uint32_t *d_scratch, uint64_t *d_n_frag, uint64_t *d_frag_offsets, uint32_t *d_frag_src,
                             void *stream);

/* ---- the same stages without gathered copies of the chunk: bases and qualities are read where the FASTQ text has them ------------------
 * (K-FQ's two gathers move 2 x the bases of a chunk through HBM and back; the chunk workers use these instead.)
 * d_seq_start / d_qual_start: per input record, from smi_fastq_index_device; d_offsets: its prefix array of read lengths, which stays the
 * coordinate system of the planes, of the fragment offsets and of every length.  The packer reads the text in 16-byte pieces: behind the
 * bases of a record of 29 or more bases up to 31 further bytes of d_text are read (and ignored) -- in a record that smi_fastq_index_device
 * accepted those are its own line end, '+' line and quality line. */
int smi_pack_reads_text_device(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_seq_start, const uint64_t *d_offsets, size_t n,
                               uint64_t total_bases, uint32_t *d_planes, void *stream);
/* text positions of the bases / qualities of the n_out output records: fragment f of input record d_frag_src[f] >> 2 begins
 * d_frag_offsets[f] - d_offsets[src] bases into it; d_frag_src == NULL: output records = input records.  d_qual_out may be NULL. */
int smi_frag_text_starts_device(smi_ctx *ctx, const uint64_t *d_seq_start, const uint64_t *d_qual_start, const uint64_t *d_offsets,
                                const uint64_t *d_frag_offsets, const uint32_t *d_frag_src, size_t n_out, uint64_t *d_base_start,
                                uint64_t *d_qual_out, void *stream);
/* smi_pack_ends_device for pass 2 (no qualities): record i's bases begin at d_text[d_base_start[i]], its length is
 * d_offsets[i + 1] - d_offsets[i] (the fragment offsets when the splitter ran) */
int smi_pack_ends_text_device(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_base_start, const uint64_t *d_offsets, size_t n,
                              uint32_t *d_ends, int32_t *d_read_len, void *stream);

/* name of fragment k (0 .. n_split) of a split read: readName.replaceFirst(" ", "_" + tag + "sp" + (k+1) + " ")
 * (L309,L323); returns the length written or a negative smi_status */
int smi_chimera_fragment_name(const char *read_name, const smi_chimera_result *res, int fragment, char *out, size_t cap);

/* ================================================================================================================
 * FASTQ ingest (the caller side of the path): uncompressed FASTQ text resident on the device ->
 * record index + the contiguous read / quality buffers the kernels above take.  Replaces the htsjdk FastqReader loop of
 * FastqFileReader$OneFastqFileWorker (FJ!nanoporereadscanner/readerwriter/FastqFileReader.java:L138-167); gz inflate
 * stays on the host.  FastqReader's checks are reported in *errors (the reference throws SAMException), never repaired.
 * ================================================================================================================ */
#define SMI_FQ_BAD_SEQ_HEADER 1u  /* a record's first line does not start with '@' */
#define SMI_FQ_BAD_QUAL_HEADER 2u /* its third line does not start with '+' */
#define SMI_FQ_LENGTH_MISMATCH 4u /* sequence and quality lines differ in length */
#define SMI_FQ_TRUNCATED 8u       /* the text does not end on a record boundary (line count not a multiple of 4) */

/* d_line_start: scratch, cap_lines >= number of lines + 1 (4 * records + 2 is always enough);
 * per record r < *n_records: name = text[name_start .. +name_len) (without '@'), read = text[seq_start .. +seq_len),
 * qualities = text[qual_start .. +seq_len); d_offsets[r] = sum of seq_len before r (cap_records + 1 entries), the
 * record arrays need cap_records >= *n_records + 1 entries (d_seq_len[n_records] is zeroed for the scan; a text
 * with cap_records or more records is rejected with SMI_ERR_INVALID); d_offsets is the `offsets` array of smi_pack_ends_device / smi_pack_reads_device once the reads are gathered.  Synchronises the stream. */
int smi_fastq_index_device(smi_ctx *ctx, const uint8_t *d_text, size_t n_bytes, uint64_t *d_line_start, size_t cap_lines,
                           uint64_t *d_name_start, uint32_t *d_name_len, uint64_t *d_seq_start, uint32_t *d_seq_len,
                           uint64_t *d_qual_start, uint64_t *d_offsets, size_t cap_records, size_t *n_records,
                           uint32_t *errors, void *stream);

/* d_out[d_offsets[r] .. d_offsets[r+1]) = d_text[d_start[r] ..): call with seq_start for the reads, qual_start for the
 * qualities */
int smi_fastq_gather_device(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_start, const uint64_t *d_offsets, size_t n,
                            uint8_t *d_out, void *stream);

/* ================================================================================================================
 * FASTQ record writer of pass 2 (the other side of the path): the records of a chunk as the
 * reference writes them, assembled on the device into a `passed` and a `failed` byte stream.  Replaces
 * FastqRecordExt.getRecordForWriting (FJ!nanoporereadscanner/readerwriter/FastqRecordExt.java:L209-311) and the record
 * loop of FastqWriterThreadPool$FastQoneFileThread.run (.../FastqWriterThreadPool.java:L300-306) with htsjdk's
 * BasicFastqWriter layout ('@' name LF bases LF '+' quality header LF qualities LF).  File handling and gz stay on the host.
 *   passed record: name = token before the first blank [+ fragment tag] + suffix (smi_format_read_name), bases = the
 *     stranded read (reverse complement through FastqRecordExt.REVERSE_COMPLEMENT for PASSED_REV), qualities reversed
 *     with it; with trim_fastq and an assigned barcode only [TSO end (5': barcode start + 30) .. polyA start];
 *     read id = first_read_id + ordinal among the passed records of the call (GET_NEXT_READID per passed record)
 *   failed record (also every fragment-less read flagged SMI_CHIM_MULTI): name token + "_FAILED ", raw bases / qualities
 * d_text / d_line_start: the chunk and the line table smi_fastq_index_device filled; d_offsets: n_out + 1 offsets of the
 * output records in d_reads / d_quals (the fragment offsets after a split); d_frag_src / d_chim: both NULL without the
 * splitter, else per output record src << 2 | fragment and per input record the split result; d_rank may be NULL.
 * Outputs: d_rec_off[i] = offset of record i in ITS stream, d_is_passed[i]; totals[0..2] = bytes passed, bytes failed,
 * records passed (host); *errors = SMI_WR_* bits (host; any bit makes the call fail).  Synchronises the stream.
 * ================================================================================================================ */
typedef struct {
    int32_t five_prime;
    int32_t trim_fastq; /* -u / --trimFastq (NanoporeReadScannerMain.java:L240), default off */
} smi_write_config;
#define SMI_WR_NAME_RANGE 1u    /* the X= / Q= range of a passed read leaves the read: the reference throws */
#define SMI_WR_NAME_TOO_LONG 2u /* a formatted name longer than 1016 bytes */
#define SMI_WR_OVERFLOW 4u      /* cap_passed / cap_failed too small (nothing is written past a cap) */
#define SMI_WR_QUAL_NEWLINE 8u  /* smi_fastq_write_host only: a line end inside a quality string the one-pass index had stepped over (reserved = 1) */
int smi_fastq_write_device(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_line_start, const uint8_t *d_reads,
                           const uint8_t *d_quals, const uint64_t *d_offsets, const uint32_t *d_frag_src,
                           const smi_chimera_result *d_chim, const smi_scan_result *d_scan, const smi_bc_result *d_bc,
                           const int32_t *d_rank, size_t n_out, uint32_t first_read_id, const smi_write_config *cfg,
                           uint8_t *d_passed, size_t cap_passed, uint8_t *d_failed, size_t cap_failed, uint64_t *d_rec_off,
                           uint8_t *d_is_passed, uint64_t *totals, uint32_t *errors, void *stream);
/* smi_fastq_write_device with d_reads / d_quals replaced by the text positions of smi_frag_text_starts_device */
int smi_fastq_write_text_device(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_line_start, const uint64_t *d_base_start,
                                const uint64_t *d_qual_start, const uint64_t *d_offsets, const uint32_t *d_frag_src,
                                const smi_chimera_result *d_chim, const smi_scan_result *d_scan, const smi_bc_result *d_bc,
                                const int32_t *d_rank, size_t n_out, uint32_t first_read_id, const smi_write_config *cfg, uint8_t *d_passed,
                                size_t cap_passed, uint8_t *d_failed, size_t cap_failed, uint64_t *d_rec_off, uint8_t *d_is_passed,
                                uint64_t *totals, uint32_t *errors, void *stream);

/* ================================================================================================================
 * The per-chunk workers of `scanfastq` as one call each (what a JNI shim calls per FastqFileReader$ReadChunk): host
 * FASTQ text in, results out; everything between the upload and the download runs on the device through the entry points
 * above.  Replaces WorkerReadscanner.scan -> Parser.call for the chunk (FJ!nanoporereadscanner/WorkerReadscanner.java:
 * L186-273, FJ!nanoporereadscanner/analyzers/Parser.java:L132-185), UsedCellBCListGenerator.call in pass 1
 * (UsedCellBCListGenerator.java:L198-229) and the record loop of the writer thread (FastqWriterThreadPool.java:L300-306).
 * One call at a time per context (private stream, grow-only device arena).
 * ================================================================================================================ */
typedef struct {
    int32_t max_ed;              /* --bcEditDistance: 0, 1 or 2 */
    int32_t five_prime;          /* -p */
    int32_t dont_search_polya;   /* --noPolyARequired (5' only); also switches the chimera splitter off (Parser.java:L176) */
    int32_t split_chimeras;      /* 1: ChimeraFindernew.findSplitPositions as in pass 2 */
    int32_t trim_fastq;          /* -u */
    int32_t want_results;        /* 1: per-record scan / barcode results are returned as well */
    uint32_t first_read_id;      /* id of the first passed record of this chunk (READCOUNTER + 1) */
    uint32_t compress;           /* --compress: 1 = `passed` / `failed` come back as ONE gzip member each (K-DEFLATE, smi_gzip_device) instead of
                                  * text; the members of a file's chunks, written one after the other, are its .fastq.gz.  Text worker only */
    const uint64_t *rank_keys;   /* used list of pass 1, sorted ascending, or NULL (-g mode): rk= field */
    const int32_t *rank_values;
    size_t n_ranks;
    uint32_t device_output;      /* 1 (text worker, not with want_results): nothing is downloaded -- out->passed / out->failed are DEVICE pointers into the
                                  * context's arena (the text K-WRITE wrote, or K-DEFLATE's members with compress), valid until the context's next call.
                                  * With text that is in device memory already (read in place, no copy) the whole chunk is one call with four waits for
                                  * the host: line count, record index, fragment count, output sizes */
    uint32_t reserved;
} smi_pass2_config;
typedef struct {
    const uint8_t *passed, *failed; /* FASTQ text of the two output files; owned by the context, valid until its next call */
    size_t passed_bytes, failed_bytes;
    size_t n_records_in, n_records_out, n_passed; /* out = after the chimera split; next first_read_id = first + n_passed */
    const smi_scan_result *scan;    /* n_records_out entries when want_results, else NULL */
    const smi_bc_result *bc;
    uint32_t fastq_errors;          /* SMI_FQ_* (the call fails when non-zero) */
    uint32_t reserved;
    const void *stats;              /* smi_scan_stats of this chunk (packed worker with want_results; NULL otherwise); owned by the context */
    size_t passed_text_bytes, failed_text_bytes; /* size of the FASTQ text (= passed_bytes / failed_bytes unless cfg->compress) */
} smi_pass2_output;
/* page-locked host memory for the text handed to the workers (uploads at link speed); freed with smi_host_free */
int smi_host_alloc(size_t bytes, void **out);
int smi_host_free(void *p);
int smi_pass2_default_config(smi_pass2_config *cfg);
/* text: host memory (page-locked: link speed) or DEVICE memory (the output of smi_gz_inflate_device: the text never visits the host) */
int smi_scanfastq_pass2_chunk(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, const smi_pass2_config *cfg,
                              smi_pass2_output *out);
/* pass 1: adds the chunk's whitelist hits to d_hist (device, one u32 counter per key of the loaded set, mode 1).
 * Stream order: the kernels run on the context's own non-blocking stream, which is NOT ordered against the stream the
 * caller allocated / zero-filled d_hist on: the caller synchronises that stream before the first call (the Python
 * binding does); contexts sharing one d_hist only ever atomicAdd into it.  Returns after the stream has drained. */
int smi_scanfastq_pass1_chunk(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, int five_prime, int dont_search_polya,
                              uint32_t *d_hist, size_t *n_records, uint32_t *fastq_errors);
/* the same chunk without a list of possible barcodes: keys appended to d_keys / *d_count as smi_pass1_keys_device does (no barcode set needed) */
int smi_scanfastq_pass1_chunk_keys(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, int five_prime, int dont_search_polya,
                                   uint64_t *d_keys, size_t cap_keys, uint64_t *d_count, size_t *n_records, uint32_t *fastq_errors);

/* ================================================================================================================
 * The packed boundary of `scanfastq` ("multi-file parallel decode and on-host trimming"): the host keeps the
 * FASTQ text, ships the bases only -- four IUPAC bit-planes per read, 0.5 byte per base -- and gets DECISIONS back (split positions,
 * scan result, barcode result, rank: ~80 bytes per record); the `passed` / `failed` text is then assembled on host threads from the text
 * the host never gave away.  5 KB per read cross the link with the text workers above, 0.7 KB with these.  Output is byte-identical to
 * smi_scanfastq_pass2_chunk (tests/test_packed_gpu.py); host code is AVX-512 (BW + VBMI) where the CPU has it, AVX2 or plain C++ otherwise.
 * Replaces the same reference units as the text workers: FastqFileReader$ReadChunk -> Parser.call -> FastqWriterThreadPool$FastQoneFileThread.run
 * (FJ!nanoporereadscanner/readerwriter/FastqWriterThreadPool.java:L300-306).
 * ================================================================================================================ */
typedef struct {            /* one FASTQ record of a chunk of text (positions into the text; line ends and a CR in front of them stripped) */
    uint64_t name_start;    /* behind '@' */
    uint64_t seq_start;
    uint64_t plus_start;    /* behind '+' */
    uint64_t qual_start;
    uint32_t name_len, seq_len, plus_len, reserved;
} smi_fastq_record;
/* The record index smi_fastq_index_device builds, on n_threads host threads (htsjdk FastqReader rules, SMI_FQ_* in *errors, nothing
 * repaired).  recs: cap_records entries, offsets: cap_records + 1 (prefix sums of seq_len = the coordinate system of planes, fragment
 * offsets and every length, as on the device).  A text with more than cap_records records is rejected. */
int smi_fastq_index_host(const uint8_t *text, size_t n_bytes, smi_fastq_record *recs, uint64_t *offsets, size_t cap_records,
                         size_t *n_records, uint32_t *errors, int n_threads);
/* K-PACKR on the host: planes in the layout of smi_pack_reads_device (smi_read_planes_words(offsets[n], n) u32 words, every word of it
 * written) */
int smi_pack_reads_host(const uint8_t *text, const smi_fastq_record *recs, const uint64_t *offsets, size_t n, uint32_t *planes,
                        int n_threads);
/* Index and planes in ONE pass over the text (what the chunk workers below use): every thread walks its share of the text once and
 * writes its reads' planes into a segment of its own, so a read's place is no longer a function of its offset -- pstart[r] says where read
 * r begins in a plane of the compact layout in which the device holds the segments back to back.  The quality lines are stepped over
 * (their length is known): recs[r].reserved = 1 asks whoever reads the qualities later (smi_fastq_write_host, smi_pack_quals_host) to
 * check them for hidden line ends, so a malformed chunk fails in the same cases as with the text workers.  Falls back to
 * smi_fastq_index_host + smi_pack_reads_host (one segment, pstart = NULL) whenever the text is not plain well-formed FASTQ.
 * planes: smi_packed_planes_words(n_bytes, n_threads) u32 words; pstart, recs: cap_records entries, offsets: cap_records + 1. */
#define SMI_PACKED_MAX_SEGMENTS 256
typedef struct {
    const uint32_t *planes;   /* host: plane c of segment k = planes[c * stride + seg_host_word[k] .. + seg_words[k]) */
    size_t stride;
    const uint32_t *pstart;   /* n entries, or NULL: K-PACKR's layout (read r at plane_start(offsets[r], r), one segment) */
    int32_t n_seg, reserved;
    size_t total_words;       /* words per plane of the compact layout */
    uint64_t seg_host_word[SMI_PACKED_MAX_SEGMENTS], seg_dev_word[SMI_PACKED_MAX_SEGMENTS], seg_words[SMI_PACKED_MAX_SEGMENTS];
} smi_packed_reads;
size_t smi_packed_planes_words(size_t n_bytes, int n_threads);
int smi_fastq_index_pack_host(const uint8_t *text, size_t n_bytes, smi_fastq_record *recs, uint64_t *offsets, uint32_t *pstart,
                              size_t cap_records, uint32_t *planes, size_t planes_words, smi_packed_reads *packed, size_t *n_records,
                              uint32_t *errors, int n_threads);
/* the quality side of pass 1 (K-PACK's k_pack_quals): qsum[r] = sum of (q - 33) over the read, qtail[r][SMI_END_BASES] = its last
 * qualities right-aligned (five_prime: its first ones, left-aligned), '!' where the read is shorter */
int smi_pack_quals_host(const uint8_t *text, const smi_fastq_record *recs, size_t n, int five_prime, uint8_t *qtail, uint32_t *qsum,
                        int n_threads);
/* what comes back from the device for a chunk; arrays are owned by the context (page-locked), valid until its next call */
typedef struct {
    size_t n_records_in, n_records_out;   /* out = after the chimera split */
    const smi_chimera_result *chim;       /* n_records_in entries, NULL when the splitter did not run */
    const uint64_t *frag_offsets;         /* n_records_out + 1: output record i = bases [frag_offsets[i], frag_offsets[i+1]) of the chunk */
    const uint32_t *frag_src;             /* n_records_out: input record << 2 | fragment; NULL when the splitter did not run */
    const smi_scan_result *scan;          /* n_records_out */
    const smi_bc_result *bc;              /* n_records_out */
    const int32_t *rank;                  /* n_records_out, NULL without a rank table */
} smi_pass2_decisions;
/* ---- scan statistics: the counters ReadScanner.html renders and stats.pojo stores, as text ------------------------------------
 * smi_record_flags: the reference's whole 64-bit flag word of a record (ReadFlags$Flags values) from the scan result (whose low bits it
 * already is), the barcode result (BC_FOUND, BC_FOUND_ED*, BC_ED_DIFF*, BC_OFFSET*: Parser.assignBarcode), the splitter (READS_AFTER_SPLIT;
 * MULTI_CHIMERIC_READS_DISCARDED | FAILED) and ReadFlags$Flags.finalizeFlag (ReadFlags.java:L194-207); equal to the `flag` of every record
 * of tests/golden/ref_exec_pass2_*.json.  smi_scan_stats_add: ReadFlags.addForCounting + the read-length sums of Parser.processOneRecord
 * (Parser.java:L114-118) over the records of one chunk; smi_scan_stats_merge: ReadFlags.mergeStats (what `mergestats` sums);
 * smi_scan_stats_tsv: the text of ReadFlags.print (description TAB count TAB percent TAB "of <reference>").  The HTML page, stats.pojo
 * (Java serialisation) and the QV histograms are not built. */
#define SMI_N_READ_FLAGS 37
typedef struct {
    uint64_t counts[SMI_N_READ_FLAGS];  /* ReadFlags.countsMap in enum order */
    uint64_t sum_len_passed, sum_len_failed, n_reads_split;
} smi_scan_stats;
uint64_t smi_record_flags(const smi_scan_result *scan, const smi_bc_result *bc, int from_split, int multi_chimeric);
int smi_scan_stats_add(smi_scan_stats *st, const smi_pass2_decisions *dec);
int smi_scan_stats_merge(smi_scan_stats *dst, const smi_scan_stats *src);
int smi_scan_stats_tsv(const smi_scan_stats *st, char *out, size_t cap, size_t *n_out);

/* device side of pass 2 from packed reads: upload of planes + offsets (host memory, page-locked for link speed), K-CHIM, fragment
 * offsets, read ends cut out of the planes (k_ends_from_planes), K-SCAN, K-BC, rank lookup, download of the decisions */
int smi_scanfastq_pass2_packed(smi_ctx *ctx, const uint32_t *planes, const uint64_t *offsets, size_t n, const smi_pass2_config *cfg,
                               smi_pass2_decisions *out);
/* the same from the segments of smi_fastq_index_pack_host */
int smi_scanfastq_pass2_packed_seg(smi_ctx *ctx, const smi_packed_reads *packed, const uint64_t *offsets, size_t n, const smi_pass2_config *cfg,
                                   smi_pass2_decisions *out);
/* K-WRITE on host threads: the records of smi_fastq_write_device from the host's text, its index and the decisions.  passed / failed:
 * caller's buffers; totals[0..2] = bytes passed, bytes failed, records passed; *errors = SMI_WR_* (any bit fails the call). */
int smi_fastq_write_host(const uint8_t *text, const smi_fastq_record *recs, const uint64_t *offsets, const smi_pass2_decisions *dec,
                         uint32_t first_read_id, const smi_write_config *cfg, uint8_t *passed, size_t cap_passed, uint8_t *failed,
                         size_t cap_failed, uint64_t *totals, uint32_t *errors, int n_threads);
/* the three steps as one call per chunk, same contract and same bytes as smi_scanfastq_pass2_chunk (out->scan / bc with want_results);
 * n_threads host threads index, pack and write */
int smi_scanfastq_pass2_chunk_packed(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, const smi_pass2_config *cfg, int n_threads,
                                     smi_pass2_output *out);
/* pass 1 likewise: index, planes, quality sums and tails on the host; ends cut from the planes, K-SCAN<22> + filter, K-HIST on the device */
int smi_scanfastq_pass1_chunk_packed(smi_ctx *ctx, const uint8_t *text, size_t n_bytes, int five_prime, int dont_search_polya,
                                     uint32_t *d_hist, int n_threads, size_t *n_records, uint32_t *fastq_errors);
/* read ends for smi_scan_device cut out of read planes: record i = bases [d_rec_offsets[i], d_rec_offsets[i+1]) of the chunk, inside input
 * read d_frag_src[i] >> 2 (d_frag_src == NULL: record i = read i); same ends / lengths as smi_pack_ends_device gives from ASCII */
int smi_ends_from_planes_device(smi_ctx *ctx, const uint32_t *d_planes, const uint64_t *d_read_offsets, size_t n_reads,
                                uint64_t total_bases, const uint64_t *d_rec_offsets, const uint32_t *d_frag_src, size_t n_records, uint32_t *d_ends,
                                int32_t *d_read_len, void *stream);

/* `assignumis` for one chunk of BamReader (the records between two cuts of BamReader.run): OneBatchExecutor.call +
 * UmiClustering.cluster after ReadGrouper.groupSams (FJ!umifinder/OneBatchExecutor.java:L61-90,
 * FJ!umifinder/analyzers/clustering/UmiClustering.java:L97-161, FJ!umifinder/bamreaders/ReadGrouper.java:L82-230), 3' or
 * 5' barcoding (cfg->five_prime): names parsed as FastqRecordExt.getScanDatFromReadName does, clustering position, region grouping, K-UMI,
 * clustering.  names: the QNAMEs back to back, name i = names[name_off[i] .. name_off[i+1]); cigars: BAM-encoded ops of
 * all records, record i = cigars[cigar_off[i] .. cigar_off[i+1]); flags / pos0 as in the BAM record.  out[i] for the first
 * *n_done records is final; the others must be handed in again at the front of the next chunk (keep_data_end). */
typedef struct {
    int32_t max_dist;          /* max_GenomeDistance_forGrouping (500) */
    int32_t grouping_distance; /* distanceFromReadEndForGrouping (100) */
    int32_t bc_edit_limit;     /* -b: barcodes with a larger ed are ignored (FastqRecordExt.java:L450-456); -1 = no limit */
    int32_t keep_data_end;     /* 1: more records of this chromosome follow (ReadGrouper.java:L171-184) */
    int32_t n_threads;         /* host threads of the clustering */
    int32_t five_prime;        /* -p: 5' barcoding (UmiFinderMain.java:L249): X= read forwards, barcode end = bcEnd - AE + 3 on it
                                  (ClusteringEditDistanceBase.java:L312-313, FastqRecordExt.java:L378); clustering position = reference
                                  position under read position AE + cell_bc_length + umi_length + grouping_distance
                                  (NanoporeRead$ReadScanData.java:L90-92) */
    int32_t umi_length;        /* umis/umi_length (config.xml:264): 8 .. 12 in this build; 0 = the context's knob (smi_ctx_set_knobs), 12 without one.
                                  The UMI windows are umi_length + 2 bases (the umi_length-mers at offsets -1 / 0 / +1 behind the barcode,
                                  ClusteringEditDistanceBase.java:L316-329), U8 / U7 umi_length characters (OneNanoporeResult.java:L111) */
    const smi_umi_cluster_config *cluster; /* NULL: shipped values */
    uint64_t random_umi_seed;  /* `assignumis -f / --randomUMI` (ClusteringEditDistanceBase.java:L308-310, L323, L330): != 0 replaces every read's UMI window by a
                                  random one drawn from (seed, position of the record in the chunk) -- what still clusters is chance; 0 = off */
} smi_assignumis_config;
#define SMI_UMI_HAS_BC 1u    /* the name carries a barcode (the record goes to the output BAM) */
#define SMI_UMI_HAS_U7 2u    /* u7 valid: the read's own 12 bases behind the barcode */
#define SMI_UMI_CLUSTERED 4u /* u8 / u1 / u2 / center valid: UMI from clustering (tags U8 U7 UC U1 [U2]) */
#define SMI_UMI_SKIPPED 8u   /* UMI_CLUSTERING_SKIPPED_HIGHCOMPLEXITY | DONT_ASSIGN_UMI: no U8 at all */
typedef struct {
    int32_t region;  /* ordinal of the genomic region within the chunk, -1 none */
    int32_t center;  /* chunk index of the read whose UMI was taken, -1 */
    int8_t u1, u2;   /* tags U1 / U2 (-1: absent) */
    uint8_t flags;   /* SMI_UMI_* */
    uint8_t reserved;
    char u8[12], u7[12]; /* umi_length characters, zero bytes behind them when umi_length < 12 */
} smi_umi_tag;
int smi_assignumis_default_config(smi_assignumis_config *cfg);
int smi_assignumis_chunk(smi_ctx *ctx, const char *names, const uint32_t *name_off, const uint16_t *flags, const int32_t *pos0,
                         const uint32_t *cigars, const uint32_t *cigar_off, int32_t n, const smi_assignumis_config *cfg,
                         smi_umi_tag *out, int32_t *n_done);

/* ================================================================================================================
 * BAM ingest of `assignumis` (host): what BamReader (FJ!umifinder/bamreaders/BamReader.java:L82-158)
 * gets from htsjdk's SamReader -- the BGZF container and the BAM record layout (SAM specification 4.1 / 4.2; htsjdk is an
 * un-vendored jar dependency, the formats are restated from the specification).  Nothing here touches the device.
 * ================================================================================================================ */
/* size of the inflated data of the complete BGZF blocks in in[0 .. n_in); *consumed = bytes those blocks take */
int smi_bgzf_uncompressed_size(const uint8_t *in, size_t n_in, size_t *n_out, size_t *n_blocks, size_t *consumed);
/* inflates every complete block (CRC32 and ISIZE checked, as BlockCompressedInputStream does) on n_threads threads */
int smi_bgzf_inflate(const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *n_out, size_t *consumed,
                     int n_threads);
/* K-INFLATE: the *.fastq.gz inputs of scanfastq (FastqFileReader.java:L138-150: a GZIPInputStream per file; README.md:155: the reference
 * parallelises over input files) inflated on the device, one wavefront per file (the lanes decode the bit positions of a step speculatively).
 * d_in: the files' bytes on the device, file i at in_off (a multiple of 4), in_len bytes; 1 KiB of readable bytes behind the last file.
 * File i's text goes to d_out + out_off, at most out_cap bytes (single-member files say their size in their last four bytes).  Multi-member
 * files, stored / fixed / dynamic blocks and the optional header fields are handled; CRC-32 and ISIZE of every member are checked.
 * results[i].status: 0 = inflated and verified; anything else (malformed or unusual input, out_cap too small, > 256 members): the caller
 * inflates that file with smi_gz_inflate.  Synchronous (the results are on the host when it returns). */
typedef struct {
    uint64_t in_off, in_len, out_off, out_cap;
} smi_inflate_stream;
typedef struct {
    uint64_t out_len;
    uint32_t status, n_members;
} smi_inflate_result;
int smi_gz_inflate_device(smi_ctx *ctx, const uint8_t *d_in, const smi_inflate_stream *streams, int n_streams, uint8_t *d_out,
                          smi_inflate_result *results, void *stream);
/* K-DEFLATE: the `--compress` writer of scanfastq (FastqWriterThreadPool.java:L242-257: a GZIPOutputStream per passed / failed file;
 * quickrun-2.1.sh:35) on the device.  d_in[0 .. n_bytes) (device) -> d_out (device, at least smi_deflate_bound(n_bytes) bytes): ONE gzip
 * member (RFC 1952; raw_deflate != 0: the bare RFC 1951 stream) that any inflater reads back to the input: dynamic-Huffman blocks of
 * 64 KiB, literals only, each closed by an empty stored block, CRC-32 and ISIZE in the trailer.  Concatenated members are a valid .gz
 * file.  d_total (device, two 8-byte words): [0] = bytes written, [1] = error flags (0 = none).  Asynchronous on `stream`. */
size_t smi_deflate_bound(size_t n_bytes);
int smi_gzip_device(smi_ctx *ctx, const uint8_t *d_in, size_t n_bytes, uint8_t *d_out, size_t out_cap, uint64_t *d_total, int raw_deflate,
                    void *stream);
/* plain (multi-member) gzip, the *.fastq.gz inputs of scanfastq (FastqFileReader.java:L138-150 via GZIPInputStream); out ==
 * NULL: only the inflated size is returned in *n_out */
int smi_gz_inflate(const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *n_out);
/* the same member by member, for a caller that does not know the inflated size (only a single-member file says it, in its last four
 * bytes): members from *in_pos on are appended at out + *out_pos while they fit in cap_out.  SMI_OK: the input is used up; 1: the member at
 * *in_pos needs more room (both positions stay in front of it; the caller grows the buffer and calls again); < 0: malformed input.  The
 * decoder is the library's own (smi_inflate_host.hip: 64-bit bit buffer, two literals per table entry, CRC-32 by carry-less multiplication);
 * CRC-32 and ISIZE of every member are checked. */
int smi_gz_inflate_into(const uint8_t *in, size_t n_in, size_t *in_pos, uint8_t *out, size_t cap_out, size_t *out_pos);
/* BGZF writer under the output BAMs of assignumis (htsjdk BlockCompressedOutputStream under UmiFinderWorker$OneBamWriter):
 * blocks of block_bytes (<= 0xFF00) input bytes + the EOF block; out == NULL: upper bound of the size in *n_out */
int smi_bgzf_deflate(const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *n_out, int level, int block_bytes,
                     int n_threads);
/* the same writer with the blocks deflated on the device (K-DEFLATE: dynamic-Huffman blocks of literals; 61,440 input bytes per block, so
 * that a block of any content stays below the 64 KiB BGZF allows): in may be host or device memory, out is host (or device) memory of at
 * least the bound the call with out == NULL returns.  Every BGZF reader (htsjdk, samtools, smi_bgzf_inflate) reads the result. */
int smi_bgzf_deflate_device(smi_ctx *ctx, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *n_out);
typedef struct {                /* one alignment record; offsets into the inflated stream */
    uint64_t rec_off;           /* of its block_size word */
    uint64_t name_off, cigar_off, seq_off, qual_off, aux_off;
    uint32_t rec_len, aux_len;  /* rec_len includes the block_size word */
    int32_t ref_id, pos;        /* pos is 0-based (SAMRecord.getAlignmentStart() = pos + 1) */
    int32_t l_seq, next_ref_id, next_pos, tlen;
    uint16_t flag, n_cigar;
    uint8_t mapq, l_read_name;  /* l_read_name counts the terminating NUL */
    uint8_t reserved[2];
} smi_bam_record;
/* magic, header text and reference dictionary; ref_* may be NULL / shorter than *n_ref (cap_ref entries are filled);
 * *records_off = offset of the first alignment record */
int smi_bam_header(const uint8_t *bam, size_t n, uint64_t *text_off, uint32_t *text_len, int32_t *n_ref, uint64_t *ref_name_off,
                   uint32_t *ref_name_len, int32_t *ref_len, size_t cap_ref, uint64_t *records_off);
/* records from offset `start` on; stops at cap records or in front of an incomplete record (*end_off) */
int smi_bam_index_records(const uint8_t *bam, size_t n, uint64_t start, smi_bam_record *recs, size_t cap, size_t *n_recs,
                          uint64_t *end_off);

/* ---- GE / GS / XF: the default gene tagger of assignumis (host only) ---------------------------------------------------------------------
 * Replaces GennameTagger (FJ!umifinder/bamreaders/GennameTagger.java:L57-382, the DefaultTagger of config.xml:86-89) over picard-2.23.9
 * RefFlatReader / Gene / LocusFunction and the htsjdk-4.1.3 OverlapDetector: called per record from OneNanoporeSeqAnalyzer.call L95-103,
 * after writeSamFlags and before the UMI tags.
 * smi_genes_load_refflat: the text of the --annotationFile (refFlat, 11 tab-separated columns) and the names of the BAM header's reference
 * sequences (rows on other sequences are skipped, RefFlatReader.java:L87-88).  Genes the reference drops (transcripts of one name on two
 * strands / chromosomes, a transcript twice, exon count that disagrees, empty or overlapping exons; a second gene with the interval and
 * strand of an earlier one) are dropped, the "earlier" one being decided in the JDK's HashMap<String> iteration order like there. */
typedef struct smi_genes smi_genes;
int smi_genes_load_refflat(const char *text, size_t n_bytes, const char *const *ref_names, int n_refs, smi_genes **out);
/* smi_genes_load_gtf: the same from a GTF annotation (README.md:727; GeneAnnotationReader.loadAnnotationsFile picks by the file name): DropseqLib's GTFReader /
 * GTFParser / GeneFromGTFBuilder -- genes by gene_name (records of the highest gene_version), a gene's extent = the extent of all its records, transcripts by
 * transcript_id from their exon / CDS features, kept by transcript_name; a line the reference's STRICT parser rejects (no gene_id / gene_name, no
 * transcript_name / transcript_id on a feature that is not `gene`, ',' in a gene name, an attribute without a value) is an error here as it ends the run
 * there; genes the LENIENT reader skips (strand / chromosome / gene_id disagreement, a `gene` feature of another extent, a transcript without exons, a
 * transcript name twice, empty or overlapping exons, no transcript) are skipped.  Multi-gene values follow GeneFromGTF.hashCode. */
int smi_genes_load_gtf(const char *text, size_t n_bytes, const char *const *ref_names, int n_refs, smi_genes **out);
int smi_genes_free(smi_genes *g);
int smi_genes_count(const smi_genes *g, size_t *n_genes, size_t *n_lines, size_t *n_skipped);
/* the loaded model as text, a line per gene in the order the reference adds them to the OverlapDetector: name, contig, start, end (1-based, inclusive), strand,
 * then its transcripts in the order Gene.iterator() walks them, `name|txStart|txEnd|cdsStart|cdsEnd|exonStart-exonEnd,...` joined by ';' (out == NULL: size only) */
int smi_genes_dump(const smi_genes *g, char *out, size_t cap, size_t *n_out);
/* n records (BAM fields: reference index, FLAG, 0-based POS, CIGAR operations `len << 4 | op` of record i at cigars[cigar_off[i] ..
 * cigar_off[i + 1])).  Output: three strings per record, GE, GS, XF, back to back in `out`; string k of record i is
 * out[out_off[3 i + k] .. out_off[3 i + k + 1]) (out_off: 3 n + 1 entries).  Empty GE = the reference sets GE and GS to null (removes them);
 * empty XF = the reference's annotateGene threw and no tag was touched.  A multi-gene value lists the genes in the JDK's HashSet<Gene>
 * iteration order.  out == NULL: sizes and offsets only. */
int smi_gene_tag_chunk(const smi_genes *g, const int32_t *ref_id, const uint16_t *flags, const int32_t *pos0, const uint32_t *cigars,
                       const uint32_t *cigar_off, int32_t n, char *out, size_t cap, uint32_t *out_off, size_t *n_out);
/* the same for n records of an inflated BAM stream (smi_bgzf_inflate) by their index entries (smi_bam_index_records) */
int smi_gene_tag_bam(const smi_genes *g, const uint8_t *bam, size_t n_bam, const smi_bam_record *recs, int32_t n, char *out, size_t cap,
                     uint32_t *out_off, size_t *n_out);

/* ---- the writer half of assignumis (host only) --------------------------------------------------------------------------------------------
 * smi_bam_write_batch: one batch of records (the records one OneBatchExecutor hands to UmiFinderWorker$BamWriters.writeSams,
 * FJ!umifinder/UmiFinderWorker.java:L408-495) -> the uncompressed BAM records of <out>.bam (out_bc: every record with a cell barcode) and
 * <out>_umifound_.bam (out_umi: those whose UMI comes from clustering), in the order the reference writes them (a stable sort of the batch by
 * htsjdk's SAMRecordCoordinateComparator, L421).  Per record: the tags of ReadScanResult.writeSamFlags / writeBCSamFlags from the scan data
 * in the read name, GennameTagger's XF / GE / GS (gene / gene_off as smi_gene_tag_bam returns them for ALL records of `recs`, or NULL), the
 * clustering's U8 U7 UC U1 U2 or the U7 -> U8 + UZ fill (tags[i] of smi_assignumis_chunk, indexed like recs), merged into the record's
 * attribute list as htsjdk keeps it (ordered by binary tag, a repeated tag keeping its last value, integers in the smallest type).
 * batch: indices into recs; order_out (n_batch entries or NULL) receives them in write order.  out_bc == NULL: sizes only.
 * gc != NULL: GeneCounts.updateGeneCounts for every written record that ends up with U8, in write order (region / nth_record indexed like
 * recs, meanings as smi_gene_counts_add).  smi_bam_chunk_inputs gathers the names / CIGARs / flags / positions of records idx[0 .. n) in
 * the layout smi_assignumis_chunk takes (names == NULL: sizes only). */
typedef struct {
    int32_t bc_edit_limit;      /* -b (FastqRecordExt.java:L450-456); -1 = no limit */
    int32_t truncate_read_name; /* -w: read name cut at its first '_' (L431-432) */
    int32_t five_prime;         /* -p: which end's clip GeneCounts looks at */
    int32_t n_threads;
    char gene_tag[4];           /* -g / config.xml gene_name_attribute (UmiFinderMain.java:L239-246): the two-letter attribute GennameTagger writes the gene name
                                 * under (TagReadBase.TAG; GS and XF stay) and GeneCounts reads it from (OneNanoporeResult.java:L522); "GE" by default */
} smi_bam_write_config;
int smi_bam_write_default_config(smi_bam_write_config *cfg);
typedef struct smi_gene_counts smi_gene_counts;
int smi_bam_write_batch(const uint8_t *bam, size_t n_bam, const smi_bam_record *recs, const int32_t *batch, int32_t n_batch,
                        const smi_umi_tag *tags, const char *gene, const uint32_t *gene_off, const smi_bam_write_config *cfg,
                        uint8_t *out_bc, size_t cap_bc, size_t *n_bc, uint8_t *out_umi, size_t cap_umi, size_t *n_umi, int32_t *order_out,
                        smi_gene_counts *gc, const int64_t *region, const uint8_t *nth_record);
int smi_bam_chunk_inputs(const uint8_t *bam, size_t n_bam, const smi_bam_record *recs, const int32_t *idx, int32_t n, char *names,
                         uint32_t *name_off, uint32_t *cigars, uint32_t *cigar_off, uint16_t *flags, int32_t *pos0, size_t *n_name_bytes,
                         size_t *n_cigar_ops);
/* nth[i] = 1 when a record of the same read name comes earlier in recs (isNthRecordForRead, OneNanoporeSeqAnalyzer.java:L74-80) */
int smi_bam_name_seen(const uint8_t *bam, size_t n_bam, const smi_bam_record *recs, int32_t n, uint8_t *nth);
/* the same for a BAM that is read in segments: the names seen so far live in the handle (as 64-bit hashes, like the reference's statsForReads
 * keys); records [from, n) of this segment are looked up and added */
typedef struct smi_name_set smi_name_set;
int smi_name_set_create(smi_name_set **out);
int smi_name_set_free(smi_name_set *set);
int smi_name_set_seen(smi_name_set *set, const uint8_t *bam, size_t n_bam, const smi_bam_record *recs, int32_t from, int32_t n, uint8_t *nth);

/* ---- <out>.genecounts.tsv / <out>.UMIdepths.tsv of assignumis (host only) -----------------------------------------------------------------
 * Replaces GeneCounts (FJ!umifinder/scanstats/GeneCounts.java:L58-652).
 * smi_gene_counts_add = updateGeneCounts (L375-491) for the n records of one written batch, called where $BamWriters.lambda$writeSams$2
 * calls it (UmiFinderWorker.java:L453-454: records that carry a BC and, after the U7 -> U8 fill, a U8 tag).  Per record: gene = the name the
 * record counts for (first entry of its GE value split at ','; the reference draws one at random when there are several, L439), NULL =
 * no GE value (the pointer array itself may be NULL); region = genomicRegionNmber of ReadGrouper (< 0: none); cell_bc / umi = 2-bit codes of
 * the BC and U8 strings (NucleicAcidTwoBitPerBase(String).getSequence()); has_bc_umi = both tags present; flag / mapq = the BAM fields;
 * first_cigar / last_cigar = first and last CIGAR operation `len << 4 | op` (0xFFFFFFFF in first_cigar: no CIGAR); nth_record = a record of
 * the same read name went through the analyzer before this one (OneNanoporeSeqAnalyzer.java:L74-80).
 * smi_gene_counts_merge = mergeGeneCounts (L540-592).  smi_gene_counts_tsv = printCountTable (L307-357): header TAB cells, one row per gene,
 * cells by their number of UMIs (descending), genes by theirs; smi_umi_depths_tsv = printUmisPerCellTable (L256-284).  Rows the reference
 * leaves in ConcurrentHashMap order (equal totals) are by ascending key.  out == NULL: size only. */
int smi_gene_counts_create(smi_gene_counts **out);
int smi_gene_counts_free(smi_gene_counts *gc);
int smi_gene_counts_add(smi_gene_counts *gc, size_t n, const char *const *gene, const int64_t *region, const uint64_t *cell_bc,
                        const uint64_t *umi, const uint8_t *has_bc_umi, const uint16_t *flag, const uint8_t *mapq,
                        const uint32_t *first_cigar, const uint32_t *last_cigar, const uint8_t *nth_record, int five_prime);
int smi_gene_counts_merge(smi_gene_counts *dst, const smi_gene_counts *src);
/* A run split over ranks / GPUs by chromosome ("no collective, only a final merge of gene counts"; BamReader.java:L130-135 cuts
 * chunks at chromosome ends, README.md:607 asks for region-complete batches): smi_gene_counts_dump / _load carry a shard's tables between
 * processes; smi_gene_counts_merge_shard folds a LATER shard into an earlier one so that the tables are those of one process over both --
 * counters of a (gene, cell, UMI) key held by both are added (plain increments commute); where the later counter carries UMIcounts'
 * "further alignment" bits the order of the increments would matter: such keys are left alone and counted in *n_order_dependent (may be
 * NULL).  Region numbers of the shards must differ (each shard numbers from its own base).  NOT mergeGeneCounts (that is
 * smi_gene_counts_merge, whose add() combines counters with ANDs, for `mergestats`). */
int smi_gene_counts_dump(const smi_gene_counts *gc, uint8_t *out, size_t cap, size_t *n_out);
int smi_gene_counts_load(const uint8_t *data, size_t n, smi_gene_counts **out);
int smi_gene_counts_merge_shard(smi_gene_counts *dst, const smi_gene_counts *later, size_t *n_order_dependent);
/* recordsWithGene, recordsWithGeneSkippedClipping, number of genes / (gene, cell, UMI) / (region, cell, UMI) entries; any pointer may be NULL */
int smi_gene_counts_info(const smi_gene_counts *gc, int64_t *records_with_gene, int64_t *records_skipped_clipping, size_t *n_genes,
                         size_t *n_gene_entries, size_t *n_region_entries);
int smi_gene_counts_tsv(const smi_gene_counts *gc, int bc_length, char *out, size_t cap, size_t *n_out);
int smi_umi_depths_tsv(const smi_gene_counts *gc, char *out, size_t cap, size_t *n_out);

/* device-time of the dominant kernel of the last *_device call on this context, measured with HIP events on the
 * stream the kernel was launched on; valid after the stream has been synchronised.  ms <= 0: not available. */
int smi_last_kernel_ms(smi_ctx *ctx, float *ms);
/* same, per kernel: the last launch of that kernel since timing was enabled */
enum { SMI_K_BC_MATCH = 0, SMI_K_SCAN = 1, SMI_K_HIST = 2, SMI_K_PACK = 3, SMI_K_UMI = 4, SMI_K_CHIMERA = 5, SMI_K_COUNT = 6 };
int smi_kernel_ms(smi_ctx *ctx, int kernel_id, float *ms);
int smi_set_timing(smi_ctx *ctx, int enabled);

#ifdef __cplusplus
}
#endif
#endif
