/*
 * sor_scan.c -- ORACLE (test infrastructure; see sor_bc.c for the rules).
 *
 * CPU restatement of the reference's 3' read scan: polyA/T finder, k-mer gated Needleman-Wunsch adapter scan,
 * strand decision / adapter acceptance, and the pass-1 quality filter.  Citations as in sor_bc.c
 * (FJ! = NanoporeBC_UMI_finder-2.1.jar, TB! = TwoFourBitNucAcidLibraryMaven-1.0.jar, Class.java:Lnn).
 * PARITY UNPINNED by the reference (no tests/fixtures, no JVM in the image; status of all oracle files: sor_bc.c).  Held by
 * executed-bytecode fixtures -- ref_exec_polyat{,_params}.json (the finder, shipped and other windows), ref_exec_nw.json,
 * ref_exec_onebyte.json (4-mer gate), ref_exec_pass1*.json (the pass-1 filter), ref_exec_pass2*_*.json (search + rules inside whole
 * records / chunks; pass2k: other config.xml knobs) -- hand-derived vectors and an independent Python model (tests/pymodel_scan.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "sor.h"

#include "sor_nw.h"

/* ---- PolyATSearcher.findpolyAT (FJ!nanopore/analyzers/PolyATSearcher.java:L56-252) ------------------------ */
/* seq: 4-bit codes of the sub-sequence (length n = window + minlen + 10).  Returns 1 and (begin,end) 1-based. */
static int count_t(const uint8_t *seq, int start, int window) { /* lambda$findpolyAT$d2855ce0$1 L61-67 */
    int r = 0;
    for (int i = start; i < start + window; i++) r += seq[i] == T4;
    return r;
}

int sor_find_polyt(const uint8_t *seq, int n, int minlen, float minfrac, int window, int *begin1, int *end1) {
    float scores[4096];
    if (window > 4096 || n < window + minlen) return -1;
    float cur = 0.0f;
    for (int i = 0; i < minlen; i++) /* L188-190 */
        if (seq[i] == T4) cur += 1.0f;
    for (int pos = 0; pos <= window - 1; pos++) { /* L194-200: pos = -1; while (pos < window-1) { pos++; ... } */
        int d = 0;
        if (seq[pos] == T4) d -= 1;          /* lambda$findpolyAT$0 L75-82 */
        if (seq[pos + minlen] == T4) d += 1;
        cur = cur + (float)d;
        scores[pos] = cur / (float)minlen; /* entry (pos, fraction of [pos+1, pos+minlen]) */
    }
    int first = -1;
    for (int pos = 0; pos < window; pos++) { /* L217-218 */
        if (scores[pos] < minfrac) continue;                                  /* lambda$4: fails when value < min */
        if (seq[pos] != T4) continue;                                         /* lambda$2 L98 */
        if (count_t(seq, pos, 5) <= 2) continue;                              /* L101 */
        first = pos;
        break;
    }
    if (first < 0) return 0;
    int start = first;
    static const int INC[8] = {20, 15, 10, 5, 4, 3, 2, 1}; /* L223-230 */
    for (int k = 0; k < 8; k++) {
        int inc = INC[k];
        /* lambda$1 L88-92: double compare against (double)minfrac - 0.1 */
        while (start + inc < window && (double)scores[start + inc] >= (double)minfrac - 0.1) start += inc;
    }
    int endpos = start + minlen - 1; /* L231 */
    /* lambda$findpolyAT$3 L122-175 */
    for (;;) {
        if (endpos <= 4) break;
        int nts[5], tot = 0;
        for (int i = 0; i < 5; i++) {
            if (seq[endpos - i] == T4) tot++;
            nts[i] = tot;
        }
        if (nts[0] != 0 && nts[1] >= 2 && nts[3] >= 3 && nts[4] >= 4) break;
        endpos--;
    }
    while (n > endpos + 6 && count_t(seq, endpos + 1, 5) > 3) endpos += 5; /* L145-147 */
    while (n > endpos + 4 && count_t(seq, endpos + 1, 3) > 1) endpos += 3; /* L156-158 */
    while (endpos < n - 1 && seq[endpos + 1] == T4) endpos++;              /* L171-172 */
    *begin1 = first + 1; /* L239 */
    *end1 = endpos + 1;
    return 1;
}

/* ---- Needleman-Wunsch (TB!nuc/alignment/needleman/{DynamicProgramming,NeedlemanWunsch,SequenceAlignment}.java) */
/* PolyATadapterAnalyzerBase.scanForTSO L324-369 */
typedef struct {
    int present; /* Match != null */
    int passed;
    int nmis;
    int end_scan; /* best position + len - 1 + ins - del, scan coordinates */
    nw_aln aln;
} tso_match;

static void scan_for_tso(const uint8_t *seq, int seq_len, const uint8_t *tso, int tso_len, int max_mm, int scantil,
                         tso_match *m) {
    scan_rslt r;
    m->present = 0;
    m->passed = 0;
    scan_adapter_max(seq, seq_len, 1, scantil, tso, tso_len, (float)max_mm, &r);
    if (r.n_all == 0) return;
    int pos = r.best_pos[0];
    nw_align(tso, tso_len, seq + pos - 1, tso_len, &SEARCH, &m->aln); /* finalAlignment scores == search scores */
    nm_counts c = needleman_counts(&m->aln);
    m->present = 1;
    m->nmis = c.nmis;
    m->passed = c.nmis <= max_mm; /* L350 */
    m->end_scan = pos + tso_len - 1 + c.ins - c.del;
}

/* PolyATadapterAnalyzer_3pBCUMI.scanReadForTSOs L122-190 */
static void scan_read_for_tsos(const char *read, int len, const sor_scan_params *par, sor_scan_result *out) {
    /* TSOparameters_3pBarcoding: sequence, windowForTSOsearch, maxNeedlemanMismatches, minTSO_NeedlemanConsecutiveMatches,
     * minTSO_TwoBestConsecutiveMatches (Jar/config.xml:155-166); the shipped values unless the caller's parameters carry others */
    const int shipped = !par || par->tso[0] == 0;
    const char *TSO = shipped ? "AACGCAGAGTACATGG" : par->tso;
    const int tso_len = 16, window = shipped ? 90 : par->tso_window, max_mm = shipped ? 5 : par->tso_max_mm,
              min_consec = shipped ? 8 : par->tso_min_consec, min_two = shipped ? 12 : par->tso_min_two;
    const int n = window + tso_len + 10;
    uint8_t tso[16], fwd[256], rev[256];
    for (int i = 0; i < tso_len; i++) tso[i] = (uint8_t)enc4((unsigned char)TSO[i]);
    for (int i = 0; i < n; i++) fwd[i] = (uint8_t)enc4((unsigned char)read[i]);
    for (int i = 0; i < n; i++) rev[i] = (uint8_t)sor_fourbit_complement(enc4((unsigned char)read[len - 1 - i]));
    tso_match f, r;
    scan_for_tso(fwd, n, tso, tso_len, max_mm, window, &f);
    scan_for_tso(rev, n, tso, tso_len, max_mm, window, &r);
#define FOUND(m) ((m).present && (m).passed)
    if (!FOUND(f) && !FOUND(r)) { /* L146-153: rescue by >= 8 consecutive matches */
        if (f.present) f.passed = n_consecutive_matches(&f.aln) >= min_consec;
        if (r.present) r.passed = n_consecutive_matches(&r.aln) >= min_consec;
        if (!FOUND(f) && !FOUND(r)) { /* L155-162: rescue by the two stretches */
            if (f.present) f.passed = sum_best_two_stretches(&f.aln) >= min_two;
            if (r.present) r.passed = sum_best_two_stretches(&r.aln) >= min_two;
        }
    }
    if (FOUND(f) && FOUND(r) && abs(f.nmis - r.nmis) > 3) { /* L167-172 */
        if (f.nmis > r.nmis)
            f.present = 0;
        else
            r.present = 0;
    }
    int ff = FOUND(f), rf = FOUND(r);
#undef FOUND
    /* L177-181: endPosRead = len - (end_scan - 1); TSO start/end = len - (endPosRead - 1) = end_scan */
    if (ff) out->tso_start = f.end_scan;
    if (rf) out->tso_end = r.end_scan;
    if (ff && !rf)
        out->flags |= SOR_F_TSO_5P;
    else if (!ff && rf)
        out->flags |= SOR_F_TSO_3P;
    else if (ff && rf)
        out->flags |= SOR_F_TSO_5P_AND_3P;
}

/* ReadFlags$Flags.finalizeFlag (FJ!nanoporereadscanner/stats/ReadFlags.java:L194-207) */
uint64_t sor_finalize_flag(uint64_t f) {
    if (!(f & SOR_F_PASSED_FWD) && !(f & SOR_F_PASSED_REV))
        f |= SOR_F_FAILED;
    else
        f |= SOR_F_PASSED_TOTAL;
    if (((f & SOR_F_PASSED_REV) && (f & SOR_F_TSO_3P)) || ((f & SOR_F_PASSED_FWD) && (f & SOR_F_TSO_5P)))
        f |= SOR_F_PASSED_TOT_TSO;
    if ((f & SOR_F_FAILED) && (f & SOR_F_TSO_5P_AND_3P)) f |= SOR_F_TSO_5P_AND_3P_FAILED;
    return f;
}

/* PolyATadapterAnalyzerBase.createNeedlemanMatch (FJ!nanopore/analyzers/PolyATadapterAnalyzerBase.java:L237-253) */
typedef struct {
    int ok;
    int start, end; /* scan coordinates, 1-based */
    nw_aln aln;
    nm_counts cnt;
} adapter_match;

static void create_needleman_match(int pos, const uint8_t *test, const uint8_t *ad, int ad_len, int max_mm,
                                   adapter_match *m) {
    nw_align(ad, ad_len, test + pos - 1, ad_len, &SEARCH, &m->aln);
    m->cnt = needleman_counts(&m->aln);
    m->ok = 1;
    if (m->cnt.nmis > max_mm) {
        /* AdapterParameters.MIN_3P_CONSEC_MATCHES_TO_OVERRIDE_PASS = 6 (FJ!parameters/AdapterParameters.java:L22) */
        if (!has_n_3p_consecutive_matches(&m->aln, 6)) m->ok = 0;
    }
    m->start = pos;
    m->end = pos + ad_len - 1 + m->cnt.ins - m->cnt.del; /* L251 */
}

/* FastqRecordExt.getMeanQV (FJ!nanoporereadscanner/readerwriter/FastqRecordExt.java:L57-59) */
static int mean_qv(const char *qual, int qlen, int start1, int stop1, float *out) {
    long long sum = 0;
    int cnt = 0;
    if (start1 - 1 < 0) return -1; /* IntStream.skip(negative) throws */
    for (int i = start1 - 1; i < qlen && cnt < stop1 - start1 + 1; i++) {
        sum += (unsigned char)qual[i] - 33;
        cnt++;
    }
    if (cnt == 0) return -1; /* OptionalDouble.getAsDouble throws */
    *out = (float)((double)sum / (double)cnt);
    return 0;
}

/* PolyATadapterAnalyzer_3pBCUMI.search (FJ!nanoporereadscanner/analyzers/PolyATadapterAnalyzer_3pBCUMI.java:L45-114)
 * + PolyATadapterAnalyzerBase.{searchpolyA L109-121, analyze L145-221, getMatchList L275-319}
 * + pass-1 filter UsedCellBCListGenerator$Worker.lambda$call$0 (UsedCellBCListGenerator.java:L198-202).
 * TSO scan (scanReadForTSOs) is not restated yet: it only sets the T= field and TSO_* flags. */
static int scan_read_3p_core(const char *read, const char *qual, int len, const char *adapter, int max_mm,
                             const sor_scan_params *par, sor_scan_result *out);

int sor_scan_read_3p(const char *read, const char *qual, int len, const char *adapter, int max_mm,
                     const sor_scan_params *par, sor_scan_result *out) {
    int rc = scan_read_3p_core(read, qual, len, adapter, max_mm, par, out);
    if (rc != 0) return rc;
    /* search() L45-46 returns before the TSO scan for too-short reads; otherwise scanReadForTSOs always runs (L111) */
    if (!(out->flags & SOR_F_READ_TOO_SHORT)) scan_read_for_tsos(read, len, par, out);
    return 0;
}

static int scan_read_3p_core(const char *read, const char *qual, int len, const char *adapter, int max_mm,
                             const sor_scan_params *par, sor_scan_result *out) {
    memset(out, 0, sizeof(*out));
    if (len < par->min_read_length) { /* testReadLength L131-137 */
        out->flags = SOR_F_READ_TOO_SHORT | SOR_F_FAILED;
        return 0;
    }
    const int sub_n = par->window_polya + par->polya_len + 10; /* PolyATSearcher.java:L178-181 */
    if (len < sub_n) return -1; /* substring / charAt would throw */
    int ad_len = (int)strlen(adapter);
    uint8_t ad[NW_MAX];
    if (ad_len > NW_MAX) return -1;
    for (int i = 0; i < ad_len; i++) ad[i] = (uint8_t)enc4((unsigned char)adapter[i]);
    uint8_t fwd[512], rev[512];
    if (sub_n > 512) return -1;
    for (int i = 0; i < sub_n; i++) fwd[i] = (uint8_t)enc4((unsigned char)read[i]);
    for (int i = 0; i < sub_n; i++) rev[i] = (uint8_t)sor_fourbit_complement(enc4((unsigned char)read[len - 1 - i]));
    int fb = 0, fe = 0, rb = 0, re = 0;
    int has_f = sor_find_polyt(fwd, sub_n, par->polya_len, par->polya_frac, par->window_polya, &fb, &fe) == 1;
    int has_r = sor_find_polyt(rev, sub_n, par->polya_len, par->polya_frac, par->window_polya, &rb, &re) == 1;
    if (!has_f && !has_r)
        out->flags |= SOR_F_POLY_A_NOT_FOUND;
    else if (has_f && !has_r)
        out->flags |= SOR_F_POLY_T_5P;
    else if (!has_f && has_r)
        out->flags |= SOR_F_POLY_A_3P;
    else
        out->flags |= SOR_F_POLY_T_5P_POLY_A_3P;
    scan_rslt sf, sr;
    /* seqTilPolyAend = sub-sequence [1, polyTend]; scan 1 .. min(len - adapter, len - 12) (L49-61) */
    if (has_f) scan_adapter(fwd, fe, 1, fe - 12, ad, ad_len, &sf);
    if (has_r) scan_adapter(rev, re, 1, re - 12, ad, ad_len, &sr);
    out->n_cand_fwd = has_f ? sf.n_all : -1;
    out->n_cand_rev = has_r ? sr.n_all : -1;
    int use_fwd = -1; /* Boolean useforward = null */
    int f_nonempty = has_f && sf.n_all > 0, r_nonempty = has_r && sr.n_all > 0;
    if (has_f || has_r) { /* analyze L145-163 */
        if (f_nonempty && r_nonempty) {
            if (fabsf(sf.best - sr.best) < 2.0f)
                out->flags |= SOR_F_ADAPTER_5P_AND_3P;
            else {
                out->flags |= SOR_F_ADAPTER_SELECTED_DESP_BOTH;
                use_fwd = sf.best < sr.best ? 1 : 0;
            }
        } else if (f_nonempty && !r_nonempty)
            use_fwd = 1;
        else if (!f_nonempty && r_nonempty)
            use_fwd = 0;
    }
    if (use_fwd < 0) { /* adapterscanResult == null -> FAILED (L178-179) */
        out->flags |= SOR_F_FAILED;
        return 0;
    }
    const int pe = use_fwd ? fe : re, pb = use_fwd ? fb : rb;
    out->polya_start = len - (pe - 1); /* setPolyAstartfromScanPosition(polyTend) ReadScanResult.java:L346 */
    out->polya_end = len - (pb - 1);   /* setPolyAendfromScanPosition(polyTbegin) L356 */
    const scan_rslt *s = use_fwd ? &sf : &sr;
    const uint8_t *test = use_fwd ? fwd : rev;
    /* getMatchList L275-319 */
    adapter_match best;
    int have = 0;
    if (s->n_best == 1) {
        create_needleman_match(s->best_pos[0], test, ad, ad_len, max_mm, &best);
        have = best.ok;
    } else {
        float best_key = 0;
        for (int i = 0; i < s->n_best; i++) {
            adapter_match m;
            create_needleman_match(s->best_pos[i], test, ad, ad_len, max_mm, &m);
            if (!m.ok) continue;
            float key = indels_mismatches_end_of_read(&m.aln, 5);
            if (!have || key < best_key) { /* smallest key; inside a key group the first (lowest position) */
                best = m;
                best_key = key;
                have = 1;
            }
        }
    }
    if (!have) {
        out->flags |= SOR_F_FAILED; /* L217 */
        return 0;
    }
    /* setAdapterMatch, 3' (ReadScanResult.java:L445-447) */
    out->adapter_found = 1;
    out->adapter_start = len - (best.start - 1);
    out->adapter_end = len - (best.end - 1);
    out->scan_end = best.end;
    out->adapter_nmis = best.cnt.nmis;
    out->flags |= use_fwd ? SOR_F_ADAPTER_5P : SOR_F_ADAPTER_3P;
    out->flags |= use_fwd ? SOR_F_PASSED_REV : SOR_F_PASSED_FWD; /* L206-213 */
    out->reverse = use_fwd ? 1 : 0;
    /* pass-1 quality filter; note the UNSTRANDED quality string is indexed with stranded coordinates (L201) */
    out->pass1_ok = 0;
    if (qual) {
        /* short-circuit && chain of lambda$call$0 (L198-202) */
        float end_err = indels_mismatches_end_of_read(&best.aln, par->min_adapter_3p_matches);
        if (end_err == 0.0f) {
            float q_bc = 0, q_read = 0;
            if (mean_qv(qual, len, out->adapter_end - 16, out->adapter_end - 1, &q_bc)) return -1;
            out->mean_qv_bc = q_bc;
            if (!(q_bc < (float)par->min_mean_bc_qv)) {
                if (mean_qv(qual, len, 1, len, &q_read)) return -1;
                out->mean_qv_read = q_read;
                out->pass1_ok = !(q_read < (float)par->min_mean_read_qv);
            }
        }
    }
    return 0;
}

/* PolyATadapterAnalyzer_5pBCUMI.search (FJ!nanoporereadscanner/analyzers/PolyATadapterAnalyzer_5pBCUMI.java:L43-76)
 * + PolyATadapterAnalyzerBase.analyze for scantype != THREEP_BARCODE (L145-221) + Adapterresult.setAdapterMatch
 * (ReadScanResult.java:L449-450: scan coordinates are kept as they are) + the pass-1 filter, which for 5' reads the
 * same quality positions AE-16 .. AE-1 (UsedCellBCListGenerator.java:L201).
 * max_mm: the caller passes maxNeedlemanMismatches + 1 as Parser.processOneRecord does (Parser.java:L99).
 * window = AdapterSearchWindow (config.xml:134, 110); dont_search_polya = --noPolyARequired. */
int sor_scan_read_5p(const char *read, const char *qual, int len, const char *adapter, int max_mm,
                     const sor_scan_params *par, int window, int dont_search_polya, sor_scan_result *out) {
    memset(out, 0, sizeof(*out));
    if (len < par->min_read_length) {
        out->flags = SOR_F_READ_TOO_SHORT | SOR_F_FAILED;
        return 0;
    }
    int ad_len = (int)strlen(adapter);
    uint8_t ad[NW_MAX];
    if (ad_len > NW_MAX) return -1;
    for (int i = 0; i < ad_len; i++) ad[i] = (uint8_t)enc4((unsigned char)adapter[i]);
    int fb = 0, fe = 0, rb = 0, re = 0, has_f = 0, has_r = 0;
    uint8_t buf[512];
    if (!dont_search_polya) { /* searchpolyA L109-121 */
        const int sub_n = par->window_polya + par->polya_len + 10;
        if (len < sub_n || sub_n > 512) return -1;
        for (int i = 0; i < sub_n; i++) buf[i] = (uint8_t)enc4((unsigned char)read[i]);
        has_f = sor_find_polyt(buf, sub_n, par->polya_len, par->polya_frac, par->window_polya, &fb, &fe) == 1;
        for (int i = 0; i < sub_n; i++) buf[i] = (uint8_t)sor_fourbit_complement(enc4((unsigned char)read[len - 1 - i]));
        has_r = sor_find_polyt(buf, sub_n, par->polya_len, par->polya_frac, par->window_polya, &rb, &re) == 1;
        if (!has_f && !has_r)
            out->flags |= SOR_F_POLY_A_NOT_FOUND;
        else if (has_f && !has_r)
            out->flags |= SOR_F_POLY_T_5P;
        else if (!has_f && has_r)
            out->flags |= SOR_F_POLY_A_3P;
        else
            out->flags |= SOR_F_POLY_T_5P_POLY_A_3P;
    }
    /* L51-68: the 3' end is scanned when a polyT was found at the 5' end (the read is the reverse strand), the 5' end when
     * a polyA was found at the 3' end */
    const int end_n = window + ad_len + max_mm + 5;
    if (len < end_n || end_n > 512) return -1; /* String.substring would throw */
    uint8_t five[512], three[512];
    for (int i = 0; i < end_n; i++) five[i] = (uint8_t)enc4((unsigned char)read[i]);
    for (int i = 0; i < end_n; i++) three[i] = (uint8_t)sor_fourbit_complement(enc4((unsigned char)read[len - 1 - i]));
    scan_rslt sf, sr;
    const int scan_rev = has_f || dont_search_polya, scan_fwd = has_r || dont_search_polya;
    if (scan_rev) scan_adapter(three, end_n, 1, window, ad, ad_len, &sr);
    if (scan_fwd) scan_adapter(five, end_n, 1, window, ad, ad_len, &sf);
    out->n_cand_fwd = scan_fwd ? sf.n_all : -1;
    out->n_cand_rev = scan_rev ? sr.n_all : -1;
    int use_fwd = -1;
    const int f_nonempty = scan_fwd && sf.n_all > 0, r_nonempty = scan_rev && sr.n_all > 0;
    if (scan_fwd || scan_rev) {
        if (f_nonempty && r_nonempty) {
            if (fabsf(sf.best - sr.best) < 2.0f)
                out->flags |= SOR_F_ADAPTER_5P_AND_3P;
            else {
                out->flags |= SOR_F_ADAPTER_SELECTED_DESP_BOTH;
                use_fwd = sf.best < sr.best ? 1 : 0;
            }
        } else if (f_nonempty)
            use_fwd = 1;
        else if (r_nonempty)
            use_fwd = 0;
    }
    if (use_fwd < 0) {
        out->flags |= SOR_F_FAILED;
        return 0;
    }
    if (!dont_search_polya) { /* L169-172: 5' protocol takes the polyA of the OTHER end */
        const int pe = use_fwd ? re : fe, pb = use_fwd ? rb : fb;
        out->polya_start = len - (pe - 1);
        out->polya_end = len - (pb - 1);
    }
    const scan_rslt *s = use_fwd ? &sf : &sr;
    const uint8_t *test = use_fwd ? five : three; /* adapterscanResult.scannedSequence L193 */
    adapter_match best;
    int have = 0;
    if (s->n_best == 1) {
        create_needleman_match(s->best_pos[0], test, ad, ad_len, max_mm, &best);
        have = best.ok;
    } else {
        float best_key = 0;
        for (int i = 0; i < s->n_best; i++) {
            adapter_match m;
            create_needleman_match(s->best_pos[i], test, ad, ad_len, max_mm, &m);
            if (!m.ok) continue;
            float key = indels_mismatches_end_of_read(&m.aln, 5);
            if (!have || key < best_key) {
                best = m;
                best_key = key;
                have = 1;
            }
        }
    }
    if (!have) {
        out->flags |= SOR_F_FAILED;
        return 0;
    }
    out->adapter_found = 1;
    out->adapter_start = best.start; /* ReadScanResult.java:L449-450 */
    out->adapter_end = best.end;
    out->scan_end = best.end;
    out->adapter_nmis = best.cnt.nmis;
    out->flags |= use_fwd ? SOR_F_ADAPTER_5P : SOR_F_ADAPTER_3P;
    out->flags |= use_fwd ? SOR_F_PASSED_FWD : SOR_F_PASSED_REV; /* L206-213 with scantype != THREEP */
    out->reverse = use_fwd ? 0 : 1;
    out->pass1_ok = 0;
    if (qual) {
        float end_err = indels_mismatches_end_of_read(&best.aln, par->min_adapter_3p_matches);
        if (end_err == 0.0f) {
            float q_bc = 0, q_read = 0;
            if (mean_qv(qual, len, out->adapter_end - 16, out->adapter_end - 1, &q_bc)) return -1;
            out->mean_qv_bc = q_bc;
            if (!(q_bc < (float)par->min_mean_bc_qv)) {
                if (mean_qv(qual, len, 1, len, &q_read)) return -1;
                out->mean_qv_read = q_read;
                out->pass1_ok = !(q_read < (float)par->min_mean_read_qv);
            }
        }
    }
    return 0;
}

/* exposed for unit tests */
int sor_nw_strings(const char *adapter, const char *read_slice, char *a1, char *dots, char *a2, float *n_errors,
                   int *ins, int *del, int *sub, float *end5) {
    static const char DEC[16] = {'-', 'A', 'G', 'R', 'C', 'M', 'S', 'V', 'T', 'W', 'K', 'D', 'Y', 'H', 'B', 'N'};
    uint8_t s1[NW_MAX], s2[NW_MAX];
    int n1 = (int)strlen(adapter), n2 = (int)strlen(read_slice);
    if (n1 > NW_MAX || n2 > NW_MAX) return -1;
    for (int i = 0; i < n1; i++) s1[i] = (uint8_t)enc4((unsigned char)adapter[i]);
    for (int i = 0; i < n2; i++) s2[i] = (uint8_t)enc4((unsigned char)read_slice[i]);
    nw_aln a;
    nw_align(s1, n1, s2, n2, &SEARCH, &a);
    for (int i = 0; i < a.len; i++) {
        a1[i] = DEC[a.a1[i]];
        a2[i] = DEC[a.a2[i]];
        dots[i] = a.dots[i];
    }
    a1[a.len] = a2[a.len] = dots[a.len] = 0;
    *n_errors = count_errors(&a);
    nm_counts c = needleman_counts(&a);
    *ins = c.ins;
    *del = c.del;
    *sub = c.sub;
    *end5 = indels_mismatches_end_of_read(&a, 5);
    return a.len;
}

int sor_scan_batch_3p(const char *reads, const char *quals, const uint64_t *offsets, size_t n, const char *adapter,
                      int max_mm, const sor_scan_params *par, sor_scan_result *out, int32_t *status, int n_threads) {
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads > 0 ? n_threads : 1)
    for (long long i = 0; i < (long long)n; i++) {
        int len = (int)(offsets[i + 1] - offsets[i]);
        status[i] = sor_scan_read_3p(reads + offsets[i], quals ? quals + offsets[i] : NULL, len, adapter, max_mm, par,
                                     &out[i]);
    }
    return 0;
}

/* ---- probes for tests/test_ref_exec.py: every alignment statistic of one (pattern, read slice) pair, and the 4-mer gate,
 * so that each can be compared with the value the reference's own bytecode produced (tests/golden/ref_exec_nw.json) ---- */
int sor_nw_stats(const char *adapter, const char *read_slice, int32_t *out9, float *out_f2, int32_t *last_row) {
    uint8_t s1[NW_MAX] = {0}, s2[NW_MAX] = {0};
    int n1 = (int)strlen(adapter), n2 = (int)strlen(read_slice);
    if (n1 > NW_MAX || n2 > NW_MAX) return -1;
    for (int i = 0; i < n1; i++) s1[i] = (uint8_t)enc4((unsigned char)adapter[i]);
    for (int i = 0; i < n2; i++) s2[i] = (uint8_t)enc4((unsigned char)read_slice[i]);
    nw_aln a;
    nw_align(s1, n1, s2, n2, &SEARCH, &a);
    nm_counts c = needleman_counts(&a);
    /* score of the alignment as SequenceAlignment.calcAlignmentScore sums it up is the bottom-right cell for these scores */
    int sc[NW_MAX + 1][NW_MAX + 1];
    for (int cc = 0; cc <= n1; cc++) sc[0][cc] = cc * SEARCH.lead2;
    for (int r = 1; r <= n2; r++) {
        sc[r][0] = r * SEARCH.lead1;
        for (int cc = 1; cc <= n1; cc++) {
            int up = sc[r - 1][cc] + SEARCH.indel, left = sc[r][cc - 1] + SEARCH.indel;
            int diag = sc[r - 1][cc - 1] + ((s2[r - 1] & s1[cc - 1]) != 0 ? SEARCH.match : SEARCH.mismatch);
            int m = up > left ? up : left;
            sc[r][cc] = diag > m ? diag : m;
        }
    }
    if (last_row)
        for (int cc = 0; cc <= n1; cc++) last_row[cc] = sc[n2][cc];
    out9[0] = a.len;
    out9[1] = has_n_3p_consecutive_matches(&a, 6);
    out9[2] = c.nmis;
    out9[3] = c.sub;
    out9[4] = c.del;
    out9[5] = c.ins;
    out9[6] = n_consecutive_matches(&a);
    out9[7] = sum_best_two_stretches(&a);
    out9[8] = c.ins - c.del;
    out_f2[0] = count_errors(&a);
    out_f2[1] = indels_mismatches_end_of_read(&a, 5);
    return 0;
}

int sor_kmers4_matching(const char *adapter, const char *read, int pos1) {
    uint8_t ad[NW_MAX], rd[4096];
    int n1 = (int)strlen(adapter), n2 = (int)strlen(read);
    if (n1 > NW_MAX || n2 > 4096) return -1;
    for (int i = 0; i < n1; i++) ad[i] = (uint8_t)enc4((unsigned char)adapter[i]);
    for (int i = 0; i < n2; i++) rd[i] = (uint8_t)enc4((unsigned char)read[i]);
    return kmers4_matching(rd, n2, ad, n1, pos1);
}
