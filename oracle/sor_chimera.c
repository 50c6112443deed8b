/*
 * sor_chimera.c -- ORACLE (test infrastructure; see sor_bc.c for the rules).
 *
 * CPU restatement of the reference's chimera splitter for 3' barcoding (pass 2, before the read scan):
 *   ChimeraFindernew.findSplitPositions            FJ!nanoporereadscanner/analyzers/ChimeraFindernew.java:L107-332
 *   PolyATadapterInternalSearcherBase.aTscan etc.  FJ!nanopore/analyzers/PolyATadapterInternalSearcherBase.java:L78-270
 *   AdapterTSOanalyzer.scanForAdapterOrTSOseqKMERsForInternal  FJ!nanopore/analyzers/AdapterTSOanalyzer.java:L130-156
 *   $AdapterScanRslt.getPosbelowMaxMismatches / getPosForBestScore  (same file L278-308)
 * Shipped parameters: Jar/config.xml:99-101 (internal polyA/T 15 bases, fraction 0.70), :105 (window 150),
 * :113,:118 (complete adapter CTACACGACGCTCTTCCGATCT, 5 errors), :170-172 (complete TSO, 6 errors),
 * :189,:264 (16-base barcode + 12-base UMI).
 * PARITY UNPINNED by the reference (no tests/fixtures, no JVM in the image; status of all oracle files: sor_bc.c).  Held by
 * executed-bytecode fixtures -- ref_exec_chimera_3p.json (findSplitPositions on whole records), ref_exec_pass2w_*.json,
 * pass2x_* (reads aimed at single branches), pass2p (other polyA windows), pass2k (other config.xml knobs) through Parser.call --
 * an independent Python model (tests/pymodel_chimera.py) and hand-built reads in tests/.
 */
#include <stdio.h>
#include <stdlib.h>

#include "sor_nw.h"

#define A4 1
#define T4 8

typedef struct {
    int pos;
    float score;
} pos_score;

/* scanForAdapterOrTSOseqKMERsForInternal: every (nErrors, pos) the scan adds to the AdapterScanRslt, in scan order */
static int scan_internal(const uint8_t *seq, int seq_len, int begin, int end, const uint8_t *ad, int ad_len,
                         float max_errors, int min_kmers, pos_score *out, int cap) {
    int n_out = 0;
    int last = seq_len - ad_len < end ? seq_len - ad_len : end;
    int delta = 1;
    for (int pos = begin; pos <= last; pos += delta) {
        delta = 1;
        if (kmers4_matching(seq, seq_len, ad, ad_len, pos) < min_kmers) continue; /* L136-138 */
        nw_aln a;
        nw_align(ad, ad_len, seq + pos - 1, ad_len, &SEARCH, &a);
        float ne = count_errors(&a);
        if (!((float)jround(ne) > max_errors) && n_out < cap) { /* L143-144 */
            out[n_out].pos = pos;
            out[n_out].score = ne;
            n_out++;
        }
        if (max_errors < ne) { /* L146-150 */
            delta = jround(ne - max_errors) - 1;
            if (delta < 1) delta = 1;
        }
    }
    return n_out;
}

static int cmp_score_then_order(const void *x, const void *y) {
    const pos_score *a = x, *b = y;
    if (a->score < b->score) return -1; /* Float.compare on the keys; no NaN, no -0.0 here */
    if (a->score > b->score) return 1;
    return a->pos - b->pos; /* same key = same map entry: list order = scan order = ascending position */
}

typedef struct {
    int is_reverse, begin, is_adapter;
} at_match; /* ChimeraFindernew$AdapterTSOmatch */

/* lambda$findSplitPositions$1 (L107-125) + lambda$5/$4/$2/$3 (L156-166) for one orientation of the complete TSO */
static int internal_tso_matches(const uint8_t *seq, int len, const uint8_t *tso, int tso_len, int max_mm, int is_reverse,
                                at_match *out, int n_out, int cap) {
    pos_score *ps = malloc(sizeof(pos_score) * (size_t)(len + 1));
    int n = scan_internal(seq, len, 70, len - 70, tso, tso_len, (float)max_mm, 2, ps, len + 1);
    /* getPosbelowMaxMismatches: keys <= max, sorted by key (stable) */
    int m = 0;
    for (int i = 0; i < n; i++)
        if (!(ps[i].score > (float)max_mm)) ps[m++] = ps[i];
    qsort(ps, (size_t)m, sizeof(pos_score), cmp_score_then_order);
    if (m > 1) { /* L115-123: drop entry i when its position is < 3 away from entry i-1 of the (score-sorted) list */
        char *drop = calloc((size_t)m, 1);
        for (int i = m - 1; i > 0; i--)
            if (abs(ps[i].pos - ps[i - 1].pos) < 3) drop[i] = 1;
        int k = 0;
        for (int i = 0; i < m; i++)
            if (!drop[i]) ps[k++] = ps[i];
        m = k;
        free(drop);
    }
    long long prev = -2147483648LL; /* AtomicInteger(Integer.MIN_VALUE) */
    for (int i = 0; i < m; i++) {
        int begin = is_reverse ? ps[i].pos + tso_len - 1 : ps[i].pos; /* L161 */
        long long lim = prev + 120;                                   /* L163: begin > prev.getAndSet(begin) + 120 */
        prev = begin;
        if ((long long)begin > lim && n_out < cap) {
            out[n_out].is_reverse = is_reverse;
            out[n_out].begin = begin;
            out[n_out].is_adapter = 0;
            n_out++;
        }
    }
    free(ps);
    return n_out;
}

/* searchATend L233-270 */
static int search_at_end(const uint8_t *seq, int len, int pos, int base, float cur, int off, int minlen, float minfrac) {
    for (int pb = pos + 1; pb < len - off - minlen - 1 && !(cur / (float)minlen < minfrac); pb++) {
        if (seq[pb] == base) cur -= 1.0f;
        if (seq[pb + minlen] == base) cur += 1.0f;
        if (!(cur / (float)minlen < minfrac)) pos = pb;
        if (seq[pb + minlen - 1] != base && seq[pb + minlen - 2] != base) break;
    }
    int end = pos + minlen - 1;
    for (;;) {
        int score = 0;
        for (int i = 0; i < 4; i++) score += seq[end - i] == base;
        if (score >= 2) break;
        end -= 4;
    }
    while (seq[end] != base) end--;
    return end;
}

/* adapterScan L159-221: start position (read coordinates) of the first accepted adapter match, or 0 */
static int adapter_scan(const uint8_t *seq, int len, int at_begin, int at_end, int base, const uint8_t *ad, int ad_len,
                        int max_errors, int bc_umi_len, int *err) {
    int start_range, end_range;
    if (base == T4) {
        start_range = at_begin - bc_umi_len - 30 - 10;
        end_range = start_range + 30 + 20;
    } else {
        end_range = at_end + bc_umi_len + 30 + 10;
        start_range = end_range - 30 - 20;
    }
    int sub_len = end_range - start_range + 1;
    if (start_range < 1 || end_range > len) { /* getSubSequence would throw */
        *err = 1;
        return 0;
    }
    uint8_t sub[64];
    for (int i = 0; i < sub_len; i++) sub[i] = seq[start_range - 1 + i];
    if (base == A4) { /* reverseComplement: swap A<->T, G<->C bits of the 4-bit code */
        uint8_t t[64];
        for (int i = 0; i < sub_len; i++) {
            uint8_t b = sub[sub_len - 1 - i];
            t[i] = (uint8_t)(((b & 1) << 3) | ((b & 8) >> 3) | ((b & 2) << 1) | ((b & 4) >> 1));
        }
        memcpy(sub, t, (size_t)sub_len);
    }
    pos_score ps[64];
    int n = scan_internal(sub, sub_len, 1, sub_len - ad_len, ad, ad_len, (float)max_errors, 3, ps, 64);
    if (n == 0) return 0;
    /* getPosForBestScore(Integer.MAX_VALUE): the entry with the least key; its positions in scan order */
    float best = ps[0].score;
    for (int i = 1; i < n; i++)
        if (ps[i].score < best) best = ps[i].score;
    int offs[64], m = 0;
    for (int i = 0; i < n; i++)
        if (ps[i].score == best) offs[m++] = ps[i].pos;
    for (int i = m - 1; i > 0; i--) /* L180-183: in-place removal, walking down */
        if (abs(offs[i] - offs[i - 1]) < 2) {
            for (int k = i; k + 1 < m; k++) offs[k] = offs[k + 1];
            m--;
        }
    for (int i = 0; i < m; i++) {
        nw_aln a;
        nw_align(ad, ad_len, sub + offs[i] - 1, ad_len, &SEARCH, &a); /* finalAlignment scores == search scores */
        nm_counts c = needleman_counts(&a);
        if (c.nmis > max_errors) continue; /* L203 (threePrimeAdapterParameters.maxCompleteSeqNeedlemanMismatches) */
        return base == T4 ? start_range + offs[i] - 1 : start_range + sub_len - offs[i]; /* L207 / L211 */
    }
    return 0;
}

static void revcomp_codes(const uint8_t *in, int n, uint8_t *out) {
    for (int i = 0; i < n; i++) {
        uint8_t b = in[n - 1 - i];
        out[i] = (uint8_t)(((b & 1) << 3) | ((b & 8) >> 3) | ((b & 2) << 1) | ((b & 4) >> 1));
    }
}

static int cmp_begin_stable(const void *x, const void *y) {
    const int *a = x, *b = y; /* {begin, original index} */
    if (a[0] != b[0]) return a[0] < b[0] ? -1 : 1;
    return a[1] - b[1];
}

int sor_chimera_split(const char *read, int len, const sor_chimera_params *par, sor_chimera_result *out) {
    memset(out, 0, sizeof(*out));
    const int off_tso = 70; /* OFFSET_FROMTSOSCAN_WINDOW L51 */
    if (len < 2 * off_tso + 100) return 0; /* L169 */
    uint8_t *seq = malloc((size_t)len);
    for (int i = 0; i < len; i++) seq[i] = (uint8_t)enc4((unsigned char)read[i]);
    const int tso_len = (int)strlen(par->tso_complete), ad_len = (int)strlen(par->adapter_complete);
    uint8_t tso[64], tso_rc[64], ad[64];
    for (int i = 0; i < tso_len; i++) tso[i] = (uint8_t)enc4((unsigned char)par->tso_complete[i]);
    for (int i = 0; i < ad_len; i++) ad[i] = (uint8_t)enc4((unsigned char)par->adapter_complete[i]);
    revcomp_codes(tso, tso_len, tso_rc);
    const int cap = len + 8;
    at_match *ms = malloc(sizeof(at_match) * (size_t)cap);
    int n = 0;
    n = internal_tso_matches(seq, len, tso, tso_len, par->tso_max_errors, 0, ms, n, cap);    /* L181,L184 */
    n = internal_tso_matches(seq, len, tso_rc, tso_len, par->tso_max_errors, 1, ms, n, cap); /* L182,L185 */
    /* aTscan L92-136 */
    const int minlen = par->internal_pat_len, off = par->window_polya + 70;
    const float minfrac = par->internal_pat_frac;
    float cur_t = 0.0f, cur_a = 0.0f;
    int end_t = 0, end_a = 0;
    long long prev_a = -2147483648LL, prev_t = -2147483648LL;
    int err = 0;
    for (int i = off - 1; i < off + minlen - 1 && i < len; i++) {
        if (seq[i] == A4) cur_a += 1.0f;
        if (seq[i] == T4) cur_t += 1.0f;
    }
    for (int pos = off - 1; pos < len - off; pos++) {
        if (seq[pos] == A4) cur_a -= 1.0f;
        if (seq[pos] == T4) cur_t -= 1.0f;
        if (seq[pos + minlen - 1] == A4) cur_a += 1.0f;
        if (seq[pos + minlen - 1] == T4) cur_t += 1.0f;
        for (int which = 0; which < 2; which++) { /* T first (L123), then A (L129) */
            const int base = which == 0 ? T4 : A4;
            const float cur = which == 0 ? cur_t : cur_a;
            int *end_cur = which == 0 ? &end_t : &end_a;
            if (cur / (float)minlen < minfrac || pos <= *end_cur || seq[pos] != base || seq[pos + 1] != base) continue;
            const int at_begin = pos + 1;
            const int at_end = search_at_end(seq, len, pos, base, cur, off, minlen, minfrac) + 1;
            *end_cur = at_end;
            /* aTadapterScanBase -> adapterScan; then lambda$6/$7/$8 (L203-210) */
            const int start = adapter_scan(seq, len, at_begin, at_end, base, ad, ad_len, par->adapter_max_errors,
                                           par->bc_umi_len, &err);
            if (start == 0) continue; /* adaptermatches == null */
            long long *prev = base == A4 ? &prev_a : &prev_t;
            const long long lim = *prev + 120;
            *prev = start;
            if ((long long)start > lim && n < cap) {
                ms[n].is_reverse = base == A4;
                ms[n].begin = start;
                ms[n].is_adapter = 1;
                n++;
            }
        }
    }
    /* split rules L229-266 */
    int sp_pos[64], sp_reason[64], n_sp = 0;
#define ISOLATED(m)                                                                        \
    do {                                                                                   \
        if (n_sp < 64) {                                                                   \
            sp_reason[n_sp] = (m).is_reverse ? SOR_SPLIT_REV_ADAPTER : SOR_SPLIT_FWD_ADAPTER; \
            sp_pos[n_sp++] = (m).is_reverse ? (m).begin + 25 : (m).begin - 25;             \
        }                                                                                  \
    } while (0)
    if (n == 1) {
        if (ms[0].is_adapter) ISOLATED(ms[0]);
    } else if (n > 1) {
        int(*key)[2] = malloc(sizeof(int[2]) * (size_t)n);
        for (int i = 0; i < n; i++) {
            key[i][0] = ms[i].begin;
            key[i][1] = i;
        }
        qsort(key, (size_t)n, sizeof(int[2]), cmp_begin_stable);
        int it = 0;
        int prev = key[it++][1]; /* index into ms, -1 = null */
        while (it < n && prev >= 0) {
            const int cur = key[it++][1];
            if (ms[cur].begin - ms[prev].begin > 160) {
                if (ms[prev].is_adapter) ISOLATED(ms[prev]);
                prev = cur;
            } else if (ms[prev].is_reverse && !ms[cur].is_reverse) {
                if (n_sp < 64) { /* lambda$11 L222-225 */
                    sp_reason[n_sp] = ms[prev].is_adapter ? (ms[cur].is_adapter ? SOR_SPLIT_RA_FA : SOR_SPLIT_RA_FT)
                                                          : (ms[cur].is_adapter ? SOR_SPLIT_RT_FA : SOR_SPLIT_RT_FT);
                    sp_pos[n_sp++] = ms[prev].begin + (ms[cur].begin - ms[prev].begin) / 2;
                }
                prev = it < n ? key[it++][1] : -1;
            } else
                prev = cur;
            if (it >= n && prev >= 0 && ms[prev].is_adapter) ISOLATED(ms[prev]); /* L263-264 */
        }
        free(key);
    }
#undef ISOLATED
    if (n_sp > 1) { /* L273-281 */
        int k = 1, keep_pos[64], keep_reason[64];
        keep_pos[0] = sp_pos[0];
        keep_reason[0] = sp_reason[0];
        for (int i = 1; i < n_sp; i++)
            if (!(sp_pos[i] - sp_pos[i - 1] < 100)) {
                keep_pos[k] = sp_pos[i];
                keep_reason[k++] = sp_reason[i];
            }
        n_sp = k;
        memcpy(sp_pos, keep_pos, sizeof(int) * (size_t)k);
        memcpy(sp_reason, keep_reason, sizeof(int) * (size_t)k);
    }
    out->n_matches = n;
    if (n_sp > 2) { /* MAX_N_SPLITPOS_FOR_CHIMERASPLIT L43,L284-286 */
        out->multi_chimeric = 1;
    } else {
        out->n_split = n_sp;
        int last = 0;
        for (int i = 0; i < n_sp; i++) {
            out->pos[i] = sp_pos[i];
            out->reason[i] = sp_reason[i];
            if (sp_pos[i] < last || sp_pos[i] > len) err = 1; /* String.substring would throw */
            last = sp_pos[i];
        }
    }
    free(ms);
    free(seq);
    return err ? -1 : 0;
}

/* fragment names L309,L323: readName.replaceFirst(" ", "_" + tag + "sp" + k + " ") */
int sor_chimera_fragment_name(const char *read_name, const sor_chimera_result *res, int fragment, char *out, size_t cap) {
    static const char *TAG[] = {"RA", "FA", "RA_FA", "RA_FT", "RT_FA", "RT_FT", ""};
    if (fragment < 0 || fragment > res->n_split || res->n_split == 0) return -1;
    const int reason = fragment < res->n_split ? res->reason[fragment] : res->reason[res->n_split - 1];
    const char *sp = strchr(read_name, ' ');
    int n;
    if (!sp)
        n = snprintf(out, cap, "%s", read_name);
    else
        n = snprintf(out, cap, "%.*s_%ssp%d %s", (int)(sp - read_name), read_name, TAG[reason], fragment + 1, sp + 1);
    return n >= (int)cap ? -1 : n;
}
