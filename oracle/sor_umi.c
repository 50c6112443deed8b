/*
 * sor_umi.c -- ORACLE (test infrastructure; rules in sor_bc.c).
 *
 * UMI pair distances of assignumis:
 *   ClusteringEditDistanceBase.lambda$static$7 (calcEditDistances)  FJ!clustering/ClusteringEditDistanceBase.java:L297-350
 *   ClusteringEditDistanceBase.calcBestEditDistance                 L67-80   (POSITIONS order L57 = enum ordinal order
 *                                                                   ZERO, PLUSONE, MINUSONE: PlusMinusOnePosData.java:L20-22)
 *   $BestEditDistance packing                                        L425-449
 *   apachemod/LevenshteinDistance.limitedCompare                     FJ!nanopore/analyzers/apachemod/LevenshteinDistance.java:L220-283
 *   window position on the read-name sequence                        FJ!nanoporereadscanner/readerwriter/FastqRecordExt.java:L378,
 *                                                                   FJ!umifinder/reads/nanopore/OneNanoporeResult.java:L59-66
 */
#include <limits.h>
#include <string.h>

#include "sor.h"

/* limitedCompare on byte arrays (4-bit codes compared for equality, so N only equals N), threshold 4 */
static int limited_compare(const uint8_t *left, int n, const uint8_t *right, int m, int threshold) {
    int pa[64], da[64];
    int *p = pa, *d = da;
    const int boundary = threshold + 1;
    for (int i = 0; i < boundary && i <= n; i++) p[i] = i;
    for (int i = boundary; i <= n; i++) p[i] = INT_MAX;
    for (int i = 0; i <= n; i++) d[i] = INT_MAX;
    for (int j = 1; j <= m; j++) {
        const uint8_t rj = right[j - 1];
        d[0] = j;
        const int mn = j - threshold > 1 ? j - threshold : 1;
        const int mx = j > INT_MAX - threshold ? n : (n < j + threshold ? n : j + threshold);
        if (mn > 1) d[mn - 1] = INT_MAX;
        int lower = INT_MAX;
        for (int i = mn; i <= mx; i++) {
            if (left[i - 1] == rj)
                d[i] = p[i - 1];
            else {
                int a = d[i - 1] < p[i] ? d[i - 1] : p[i];
                a = a < p[i - 1] ? a : p[i - 1];
                d[i] = (int)(1u + (unsigned)a); /* Java int wrap-around */
            }
            if (d[i] < lower) lower = d[i];
        }
        if (lower > threshold) return -1;
        int *t = p;
        p = d;
        d = t;
    }
    return p[n] <= threshold ? p[n] : -1;
}

/* w1, w2: 14 4-bit codes = bases bcEnd .. bcEnd+13 of the read-name sequence (1-based bcEnd), i.e. the three
 * 12-mers getSubSequence(bcEnd+1+i, 12), i = -1,0,+1.  Returns ed | pos1.value << 4 | pos2.value << 6 with
 * value 0 = MINUSONE, 1 = ZERO, 2 = PLUSONE. */
int sor_umi_pair(const uint8_t *w1, const uint8_t *w2) { return sor_umi_pair_len(w1, w2, 12); }

/* the same for umis/umi_length = umi_len (config.xml:264; L322 / L329 cut params.umis.umi_length bases): windows of umi_len + 2 codes */
int sor_umi_pair_len(const uint8_t *w1, const uint8_t *w2, int umi_len) {
    int eds[3][3];
    for (int i = -1; i < 2; i++)
        for (int j = -1; j < 2; j++) {
            const uint8_t *s1 = w1 + (i + 1), *s2 = w2 + (j + 1);
            int ed;
            if (memcmp(s1, s2, (size_t)umi_len) == 0)
                ed = 0; /* L332-333 */
            else {
                ed = limited_compare(s1, umi_len, s2, umi_len, 4); /* L341-343 */
                if (ed == -1) ed = 5;
            }
            eds[i + 1][j + 1] = ed;
        }
    static const int ORDER[3] = {1, 2, 0}; /* values of ZERO, PLUSONE, MINUSONE in ordinal order */
    int best = 127, b1 = 0, b2 = 0;       /* L67: (127, MINUSONE, MINUSONE) */
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            int i = ORDER[a], v = ORDER[b];
            if (eds[i][v] < best) { /* L73: strict */
                best = eds[i][v];
                b1 = i;
                b2 = v;
            }
        }
    return best | (b1 << 4) | (b2 << 6);
}

/* full n x n matrix as the reference fills it: [i][v] computed for v >= i, [v][i] = transposed copy (L213-216,L255) */
void sor_umi_matrix(const uint8_t *windows, int n, uint8_t *out) { sor_umi_matrix_len(windows, n, 12, out); }

/* windows: n rows of umi_len + 2 codes */
void sor_umi_matrix_len(const uint8_t *windows, int n, int umi_len, uint8_t *out) {
    const size_t wl = (size_t)umi_len + 2;
    for (int i = 0; i < n; i++)
        for (int v = i; v < n; v++) {
            int r = sor_umi_pair_len(windows + wl * (size_t)i, windows + wl * (size_t)v, umi_len);
            out[(size_t)i * n + v] = (uint8_t)r;
            out[(size_t)v * n + i] = (uint8_t)((r & 15) | (((r >> 6) & 3) << 4) | (((r >> 4) & 3) << 6));
        }
}

/* 3': x = the 43-base X= string (stranded[AE-40 .. AE+2]); the tested sequence is its reverse complement and the
 * barcode ends at adapterend + nbasesOfAdapterSeqInReadname(3) - bcEnd on it.  Returns 0 and 14 codes, or -1 when
 * the slice does not fit (the reference would throw from System.arraycopy). */
int sor_umi_window_3p(const char *x, int xlen, int adapter_end, int bc_end, uint8_t *out14) {
    return sor_umi_window_3p_len(x, xlen, adapter_end, bc_end, 12, out14);
}

/* umi_len + 2 codes for umis/umi_length = umi_len */
int sor_umi_window_3p_len(const char *x, int xlen, int adapter_end, int bc_end, int umi_len, uint8_t *out14) {
    int pos = adapter_end + 3 - bc_end; /* getStrandedShortSeqPosFromReadPos L378 */
    if (pos < 1 || pos + umi_len + 1 > xlen) return -1;
    for (int k = 0; k < umi_len + 2; k++) {
        int p1 = pos + k;                 /* 1-based on revcomp(x) */
        char c = x[xlen - p1];            /* revcomp index */
        int code = sor_fourbit_encode_char((unsigned char)c);
        if (code < 0) return -1;
        out14[k] = (uint8_t)sor_fourbit_complement(code);
    }
    return 0;
}

/* 5' barcoding (ClusteringEditDistanceBase.java:L312-313: getSeq(), not getSeqRevComp(); bcEnd on the short sequence =
 * bcEnd - adapterend + nbasesOfAdapterSeqInReadname, FastqRecordExt.java:L378 with is5pBarcoding): x = the X= string
 * (stranded[AE-2 .. AE+39]) read forwards; the three 12-mers start at 1-based bcEnd' + 1 + {-1, 0, +1} (L322, L329). */
int sor_umi_window_5p(const char *x, int xlen, int adapter_end, int bc_end, uint8_t *out14) {
    return sor_umi_window_5p_len(x, xlen, adapter_end, bc_end, 12, out14);
}

int sor_umi_window_5p_len(const char *x, int xlen, int adapter_end, int bc_end, int umi_len, uint8_t *out14) {
    int pos = bc_end - adapter_end + 3;
    if (pos < 1 || pos + umi_len + 1 > xlen) return -1;
    for (int k = 0; k < umi_len + 2; k++) {
        int code = sor_fourbit_encode_char((unsigned char)x[pos - 1 + k]); /* 1-based pos + k on x */
        if (code < 0) return -1;
        out14[k] = (uint8_t)code;
    }
    return 0;
}

int sor_limited_compare(const uint8_t *a, int n, const uint8_t *b, int m, int threshold) {
    return limited_compare(a, n, b, m, threshold);
}
