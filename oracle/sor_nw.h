/*
 * sor_nw.h -- ORACLE internals shared by sor_scan.c and sor_chimera.c (test infrastructure; see sor_bc.c).
 *
 * 4-bit codec helpers, Needleman-Wunsch + traceback, the NeedlemanMatch statistics, the 4-mer gate and
 * AdapterTSOanalyzer.scanForAdapterOrTSOseq.  Citations: FJ! = NanoporeBC_UMI_finder-2.1.jar,
 * TB! = TwoFourBitNucAcidLibraryMaven-1.0.jar, Class.java:Lnn.  PARITY UNPINNED (see sor_bc.c; held by ref_exec_nw.json: alignments, statistics, tie-breaks).
 */
#ifndef SOR_NW_H
#define SOR_NW_H
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "sor.h"

#define SOR_UNUSED __attribute__((unused))

#define T4 8 /* NucleicAcidByteCodeBase.T */

/* ---- 4-bit helpers ----------------------------------------------------------------------------------- */
SOR_UNUSED static int enc4(unsigned char c) {
    int v = sor_fourbit_encode_char(c);
    return v < 0 ? 15 : v; /* a char outside the IUPAC table would throw in the reference; FASTQ has ACGTN only */
}

/* ---- Needleman-Wunsch (TB!nuc/alignment/needleman/NeedlemanWunsch.java, SequenceAlignment.java) ---------- */
#define NW_MAX 64
typedef struct {
    int len;                  /* number of alignment columns */
    uint8_t a1[2 * NW_MAX];   /* template (seq1 = adapter), 0 = '-' */
    uint8_t a2[2 * NW_MAX];   /* read (seq2), 0 = '-' */
    char dots[2 * NW_MAX + 1];
} nw_aln;

/* NeedlemanScores(lead1,lead2,trail1,trail2,indel,mismatch,match) = (-4,-5,-5,-5,-5,-5,+5)
 * (FJ!nanopore/analyzers/parameters/NeedlemanParameters.java:L36-38); trailing scores are never read. */
typedef struct {
    int lead1, lead2, indel, mismatch, match;
} nw_scores;
SOR_UNUSED static const nw_scores SEARCH = {-4, -5, -5, -5, 5};

SOR_UNUSED static void nw_align(const uint8_t *s1, int n1, const uint8_t *s2, int n2, const nw_scores *sc, nw_aln *out) {
    static _Thread_local int score[NW_MAX + 1][NW_MAX + 1];
    static _Thread_local uint8_t dir[NW_MAX + 1][NW_MAX + 1]; /* 0 none, 1 diag, 2 up (row-1), 3 left (col-1) */
    /* getInitialScore / getInitialPointer, NeedlemanWunsch.java:L106-122 */
    score[0][0] = 0;
    dir[0][0] = 0;
    for (int c = 1; c <= n1; c++) {
        score[0][c] = c * sc->lead2;
        dir[0][c] = 3;
    }
    for (int r = 1; r <= n2; r++) {
        score[r][0] = r * sc->lead1;
        dir[r][0] = 2;
    }
    for (int r = 1; r <= n2; r++) /* fillIn L92-98, fillInCell NeedlemanWunsch.java:L55-80 */
        for (int c = 1; c <= n1; c++) {
            int row_space = score[r - 1][c] + sc->indel;
            int col_space = score[r][c - 1] + sc->indel;
            int diag = score[r - 1][c - 1] + ((s2[r - 1] & s1[c - 1]) != 0 ? sc->match : sc->mismatch);
            if (row_space >= col_space) {
                if (diag >= row_space) {
                    score[r][c] = diag;
                    dir[r][c] = 1;
                } else {
                    score[r][c] = row_space;
                    dir[r][c] = 2;
                }
            } else {
                if (diag >= col_space) {
                    score[r][c] = diag;
                    dir[r][c] = 1;
                } else {
                    score[r][c] = col_space;
                    dir[r][c] = 3;
                }
            }
        }
    /* getTraceback, SequenceAlignment.java:L102-151 */
    uint8_t t1[2 * NW_MAX], t2[2 * NW_MAX];
    int k = 0, r = n2, c = n1;
    while (dir[r][c] != 0) {
        int d = dir[r][c];
        t2[k] = (d == 1 || d == 2) ? s2[r - 1] : 0;
        t1[k] = (d == 1 || d == 3) ? s1[c - 1] : 0;
        k++;
        if (d == 1) {
            r--;
            c--;
        } else if (d == 2)
            r--;
        else
            c--;
    }
    out->len = k;
    for (int i = 0; i < k; i++) {
        out->a1[i] = t1[k - 1 - i];
        out->a2[i] = t2[k - 1 - i];
        uint8_t b1 = out->a1[i], b2 = out->a2[i];
        out->dots[i] = (b1 == 0 || b2 == 0) ? 'x' : ((b1 & b2) == 0 ? 'x' : '.');
    }
    out->dots[k] = 0;
}

/* Match.countErrorsInNeedleman (FJ!nanopore/analyzers/Match.java:L31-34) */
SOR_UNUSED static float count_errors(const nw_aln *a) {
    int nx = 0, lead = 0;
    for (int i = 0; i < a->len; i++) nx += a->dots[i] == 'x';
    while (lead < a->len && a->a1[lead] == 0) lead++; /* count5pInsertionsInNeedleman L185-189 */
    return (float)nx - 0.9f * (float)lead;
}

/* NeedlemanMatch (FJ!nanopore/analyzers/NeedlemanMatch.java) */
typedef struct {
    int ins, del, sub, nmis;
} nm_counts;

SOR_UNUSED static nm_counts needleman_counts(const nw_aln *a) { /* countNeedlemanErrorsInRead L68-86 (byte arithmetic) */
    nm_counts c = {0, 0, 0, 0};
    for (int i = 0; i < a->len; i++)
        if (a->dots[i] == 'x') {
            if (a->a1[i] == 0)
                c.ins++;
            else if (a->a2[i] == 0)
                c.del++;
            else
                c.sub++;
        }
    int i = a->len;
    while (i > 0 && a->a2[i - 1] == 0) i--;
    c.del = (int8_t)(c.del - (a->len - i));
    c.nmis = c.ins + c.del + c.sub;
    return c;
}

SOR_UNUSED static float indels_mismatches_end_of_read(const nw_aln *a, int n) { /* countIndelsMismatchesEndOfRead L109-123 */
    float ret = 0.0f;
    int len = a->len;
    int i = len - 1, k = i;
    while (k >= len - n && i >= 0) {
        if (a->dots[i] == 'x') {
            if (k >= len - 2)
                ret = (float)((double)ret + 1.2);
            else
                ret = ret + 1.0f;
        }
        if (a->a2[i] != 0) k--;
        i--;
    }
    return ret;
}

SOR_UNUSED static int has_n_3p_consecutive_matches(const nw_aln *a, int n) { /* Match.lambda$static$2 L41-50 */
    int consec = 0;
    for (int i = a->len - 1; i >= a->len - n; i--) {
        if (i < 0) break; /* charAt would throw for an alignment shorter than n; cannot happen (len >= adapter length) */
        if (a->dots[i] != '.') break;
        consec++;
    }
    return consec == n;
}

/* NucleicAcidInmutableOneBytePerBase$Kmers.nKmersMatching_4mer
 * (TB!nuc/encoding/onebyte/NucleicAcidInmutableOneBytePerBase.java:L533-543) */
SOR_UNUSED static int kmers4_matching(const uint8_t *read, int read_len, const uint8_t *ad, int ad_len, int pos1) {
    int matches = 0;
    int n_kmers = ad_len - 3;
    for (int p = pos1 - 1, k = 0; p < read_len - 3 && k < n_kmers; p++, k++)
        if ((read[p] & ad[k] & 15) && (read[p + 1] & ad[k + 1] & 15) && (read[p + 2] & ad[k + 2] & 15) &&
            (read[p + 3] & ad[k + 3] & 15))
            matches++;
    return matches;
}

/* AdapterTSOanalyzer.scanForAdapterOrTSOseq with maxErrors = Optional.empty()
 * (FJ!nanopore/analyzers/AdapterTSOanalyzer.java:L84-110) + $AdapterScanRslt.getPosForBestScore (L279-291) */
typedef struct {
    int n_all;       /* number of (nErrors, pos) entries */
    float best;      /* least key */
    int n_best;      /* positions sharing the least key, in scan order */
    int best_pos[256];
} scan_rslt;

/* Math.round(float): floor(a + 0.5f) */
SOR_UNUSED static int jround(float a) { return (int)floorf(a + 0.5f); }

/* max_errors < 0: Optional.empty() */
SOR_UNUSED static void scan_adapter_max(const uint8_t *read, int read_len, int begin, int end, const uint8_t *ad, int ad_len,
                             float max_errors, scan_rslt *res) {
    res->n_all = 0;
    res->n_best = 0;
    res->best = 3.4028234663852886e+38f;
    int last = read_len - ad_len < end ? read_len - ad_len : end;
    int delta = 1;
    for (int pos = begin; pos <= last; pos += delta) {
        delta = 1;
        int n = kmers4_matching(read, read_len, ad, ad_len, pos);
        if (n <= 1) continue;
        nw_aln a;
        nw_align(ad, ad_len, read + pos - 1, ad_len, &SEARCH, &a);
        float ne = count_errors(&a);
        if (max_errors < 0 || !((float)jround(ne) > max_errors)) { /* L97-98 */
            res->n_all++;
            if (ne < res->best) {
                res->best = ne;
                res->n_best = 0;
            }
            if (ne == res->best && res->n_best < 256) res->best_pos[res->n_best++] = pos;
        }
        if (max_errors >= 0 && max_errors < ne) { /* L100-104 */
            delta = jround(ne - max_errors) - 1;
            if (delta < 1) delta = 1;
        }
    }
}

SOR_UNUSED static void scan_adapter(const uint8_t *read, int read_len, int begin, int end, const uint8_t *ad, int ad_len,
                         scan_rslt *res) {
    scan_adapter_max(read, read_len, begin, end, ad, ad_len, -1.0f, res);
}

/* NeedlemanMatch.getNconsecutiveMatchesNeedleman L160-173: a run only counts once a non-'.' follows it */
SOR_UNUSED static int n_consecutive_matches(const nw_aln *a) {
    int ret = 0, cur = 0;
    for (int i = 0; i < a->len; i++) {
        if (a->dots[i] == '.')
            cur++;
        else {
            if (cur > ret) ret = cur;
            cur = 0;
        }
    }
    return ret;
}

/* getSumOfBestTwoMatchStretchesNeedleman L183-196: runs > 4 (closed by a non-'.'), sorted ASCENDING, first two summed */
SOR_UNUSED static int sum_best_two_stretches(const nw_aln *a) {
    int runs[2 * NW_MAX], nr = 0, cur = 0;
    for (int i = 0; i < a->len; i++) {
        if (a->dots[i] == '.')
            cur++;
        else {
            if (cur > 4) runs[nr++] = cur;
            cur = 0;
        }
    }
    for (int i = 1; i < nr; i++) {
        int x = runs[i], j = i - 1;
        while (j >= 0 && runs[j] > x) {
            runs[j + 1] = runs[j];
            j--;
        }
        runs[j + 1] = x;
    }
    int s = 0;
    for (int i = 0; i < nr && i < 2; i++) s += runs[i];
    return s;
}


#endif
