/*
 * sor_group.c -- ORACLE (test infrastructure; see sor_bc.c for the rules).
 *
 * CPU restatement of the reference's genomic-region grouping of `assignumis` (which reads may share a UMI group):
 *   ReadGrouper.groupSams / doClusteringOneStrand   FJ!umifinder/bamreaders/ReadGrouper.java:L82-260
 *   ReadGrouper$Cluster                             (same file) L455-667: centre = Math.round((float) mean), L614-615
 *   ReadGrouper$ClusterList.refineClusters          L711-785
 *   NanoporeRead$ReadScanData.getReferencePositionAtReadPosition   FJ!umifinder/reads/nanopore/NanoporeRead$ReadScanData.java:L133-153
 *   (alignment blocks as htsjdk 4.1.3 SAMUtils.getAlignmentBlocks builds them from the CIGAR)
 * Region ids: the reference numbers clusters from a static counter; only equality matters, ids here are the ordinal of
 * the cluster in the final list.  PARITY UNPINNED (no reference tests, no JVM; sor_bc.c has the status of all
 * oracle files).  Held by ref_exec_group.json (120 chunks) and ref_exec_group2.json (merge-back designs; the reference's own NullPointerException on
 * seven of them is recorded there), ref_exec_clusterpos.json (the grouping position of 320 records).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "sor.h"

typedef struct {
    int *m;     /* members: indices into the position-sorted data */
    int n, cap;
    int center, has_center;
} cl_t;

typedef struct {
    int pos, read, index, rev;
} rec_t;

static void cl_push(cl_t *c, int v) {
    if (c->n == c->cap) {
        c->cap = c->cap ? 2 * c->cap : 8;
        c->m = realloc(c->m, sizeof(int) * (size_t)c->cap);
    }
    c->m[c->n++] = v;
}

static int cl_center(cl_t *c, const rec_t *d) { /* getCenter / setCenter L491,L614-615 */
    if (!c->has_center && c->n > 0) {
        double s = 0;
        for (int i = 0; i < c->n; i++) s += d[c->m[i]].pos;
        c->center = (int)floorf((float)(s / c->n) + 0.5f);
        c->has_center = 1;
    }
    return c->center;
}

/* removeOffCenterLeft (side = -1) / Right (side = +1) = Cluster.removeOffCenter (lambda$new$5, L626-638): moved members form a new
 * cluster sorted by position.  The centre is a cached field, and the reference (a) COUNTS the off-centre members with whatever centre is
 * cached (possibly stale: a removal does not clear it), (b) only if that count is > 0 clears the cache, so that the filter pass that
 * picks the members to move recomputes the centre from the list as it is now, and (c) leaves that value cached after the members have
 * left.  Returns 1 when the reference creates a cluster (which may be empty when (a) and (b) disagree). */
static int cl_split(cl_t *c, const rec_t *d, int side, int dist, cl_t *out) {
    memset(out, 0, sizeof(*out));
    if (c->n == 0) return 0;
    int center = cl_center(c, d);
    int cnt = 0;
    for (int i = 0; i < c->n; i++) {
        const int p = d[c->m[i]].pos;
        if (side < 0 ? p < center - dist : p > center + dist) cnt++;
    }
    if (cnt == 0) return 0;
    c->has_center = 0; /* L633 */
    center = cl_center(c, d);
    int k = 0;
    for (int i = 0; i < c->n; i++) {
        const int p = d[c->m[i]].pos;
        if (side < 0 ? p < center - dist : p > center + dist)
            cl_push(out, c->m[i]);
        else
            c->m[k++] = c->m[i];
    }
    c->n = k; /* the centre stays cached at the value computed before the members left */
    if (c->n == 0) c->has_center = 0; /* setCenter on an empty list leaves null; empty clusters are dropped before any use */
    for (int i = 1; i < out->n; i++) { /* stable insertion sort by position */
        int v = out->m[i], j = i - 1;
        while (j >= 0 && d[out->m[j]].pos > d[v].pos) {
            out->m[j + 1] = out->m[j];
            j--;
        }
        out->m[j + 1] = v;
    }
    return 1;
}

typedef struct {
    cl_t *c;
    int n, cap;
} cll_t;

static void cll_push(cll_t *l, cl_t c) {
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 8;
        l->c = realloc(l->c, sizeof(cl_t) * (size_t)l->cap);
    }
    l->c[l->n++] = c;
}

static void cll_sort_nonempty(cll_t *l, const rec_t *d) { /* sortAndRemoveEmpty L703: stable by centre */
    int k = 0;
    for (int i = 0; i < l->n; i++)
        if (l->c[i].n > 0)
            l->c[k++] = l->c[i];
        else
            free(l->c[i].m);
    l->n = k;
    for (int i = 1; i < l->n; i++) {
        cl_t v = l->c[i];
        int j = i - 1;
        while (j >= 0 && cl_center(&l->c[j], d) > cl_center(&v, d)) {
            l->c[j + 1] = l->c[j];
            j--;
        }
        l->c[j + 1] = v;
    }
}

static void refine(cll_t *l, const rec_t *d, int dist) { /* refineClusters L711-785 */
    int from = 0, to = l->n;
    while (from < to) { /* off-centre passes: the clusters split off in one pass are themselves split in the next */
        for (int i = from; i < to; i++) {
            cl_t out;
            if (cl_split(&l->c[i], d, -1, dist, &out)) cll_push(l, out);
            if (cl_split(&l->c[i], d, +1, dist, &out)) cll_push(l, out);
        }
        from = to;
        to = l->n;
    }
    cll_sort_nonempty(l, d);
    for (int keep = 1; keep;) {
        keep = 0;
        for (int i = 0; i + 1 < l->n; i++) {
            if (l->c[i].n == 0) continue;
            cl_t *left = &l->c[i], *right = &l->c[i + 1];
            if (cl_center(right, d) - cl_center(left, d) >= 2 * dist) continue;
            cl_t *frm = left->n > right->n ? right : left, *dst = left->n > right->n ? left : right;
            const int tc = dst->center;
            int k = 0, moved = 0;
            for (int t = 0; t < frm->n; t++) {
                if (abs(d[frm->m[t]].pos - tc) <= dist) {
                    cl_push(dst, frm->m[t]);
                    moved++;
                } else
                    frm->m[k++] = frm->m[t];
            }
            if (moved) {
                frm->n = k;
                frm->has_center = dst->has_center = 0;
                keep = 1;
            }
        }
        int k = 0;
        for (int i = 0; i < l->n; i++)
            if (l->c[i].n > 0)
                l->c[k++] = l->c[i];
            else
                free(l->c[i].m);
        l->n = k;
    }
    int k = 0;
    for (int i = 0; i < l->n; i++)
        if (l->c[i].n > 1)
            l->c[k++] = l->c[i];
        else
            free(l->c[i].m);
    l->n = k;
}

static void one_strand(const rec_t *d, const int *idx, int n_idx, int dist, cll_t *out) { /* L234-260 */
    memset(out, 0, sizeof(*out));
    if (n_idx <= 1) return;
    cl_t cur;
    memset(&cur, 0, sizeof(cur));
    if (d[idx[1]].pos - d[idx[0]].pos < dist) cl_push(&cur, idx[0]);
    for (int i = 1; i < n_idx; i++) {
        if (d[idx[i]].pos - d[idx[i - 1]].pos < dist)
            cl_push(&cur, idx[i]);
        else if (cur.n > 2) { /* a chain of <= 2 reads is NOT closed: it keeps growing (L247) */
            cll_push(out, cur);
            memset(&cur, 0, sizeof(cur));
        }
    }
    if (cur.n > 2)
        cll_push(out, cur);
    else
        free(cur.m);
    refine(out, d, dist);
}

static int cmp_rec(const void *a, const void *b) {
    const rec_t *x = a, *y = b;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return x->index - y->index; /* Arrays.parallelSort is stable */
}

int sor_region_group(const int32_t *pos, const uint8_t *has_pos, const uint8_t *reverse, int32_t n, int32_t max_dist,
                     int keep_data_end, int32_t *region, int32_t *n_done) {
    rec_t *d = malloc(sizeof(rec_t) * (size_t)(n > 0 ? n : 1));
    int nd = 0;
    for (int i = 0; i < n; i++) {
        region[i] = -1;
        if (has_pos[i]) {
            d[nd] = (rec_t){pos[i], i, nd, reverse[i] != 0};
            nd++;
        }
    }
    qsort(d, (size_t)nd, sizeof(rec_t), cmp_rec);
    int *fi = malloc(sizeof(int) * (size_t)(nd > 0 ? nd : 1)), *ri = malloc(sizeof(int) * (size_t)(nd > 0 ? nd : 1));
    int nf = 0, nr = 0;
    for (int i = 0; i < nd; i++)
        if (d[i].rev)
            ri[nr++] = i;
        else
            fi[nf++] = i;
    cll_t all, revl;
    one_strand(d, fi, nf, max_dist, &all);
    one_strand(d, ri, nr, max_dist, &revl);
    for (int i = 0; i < revl.n; i++) cll_push(&all, revl.c[i]);
    free(revl.c);
    cll_sort_nonempty(&all, d);
    int last_index = n - 1, n_cl = all.n;
    if (keep_data_end && n_cl > 0 && nd > 0) { /* L171-184 */
        const int most_right = d[nd - 1].pos;
        while (n_cl > 0 && cl_center(&all.c[n_cl - 1], d) > most_right - 3 * max_dist) n_cl--;
        if (n_cl > 0) {
            last_index = 0;
            for (int t = 0; t < all.c[n_cl - 1].n; t++)
                if (d[all.c[n_cl - 1].m[t]].index > last_index) last_index = d[all.c[n_cl - 1].m[t]].index;
            if (last_index < n / 3) last_index = n / 3;
        }
    }
    for (int k = 0; k < n_cl; k++)
        for (int t = 0; t < all.c[k].n; t++) region[d[all.c[k].m[t]].read] = k;
    *n_done = last_index + 1;
    for (int i = 0; i < all.n; i++) free(all.c[i].m);
    free(all.c);
    free(ri);
    free(fi);
    free(d);
    return 0;
}

/* cigar: BAM encoding (len << 4 | op), op = MIDNSHP=X -> 0..8 */
int sor_ref_position_at_read_position(const uint32_t *cigar, int n_cigar, int32_t alignment_start, int32_t position,
                                      int32_t *out) {
    if (position == 0) return 0;
    int last_genomic_end = 1, last_read_end = 1, read_base = 1, ref_base = alignment_start;
    for (int i = 0; i < n_cigar; i++) {
        const int op = (int)(cigar[i] & 15u), len = (int)(cigar[i] >> 4);
        if (op == 4 || op == 1) /* S, I */
            read_base += len;
        else if (op == 3 || op == 2) /* N, D */
            ref_base += len;
        else if (op == 0 || op == 7 || op == 8) { /* M, =, X: one AlignmentBlock */
            const int rs = read_base, gs = ref_base;
            read_base += len;
            ref_base += len;
            if (rs + len - 1 < position) { /* L139-141 */
                last_genomic_end = gs + len - 1;
                last_read_end = rs + len - 1;
                continue;
            }
            *out = position < rs ? gs - abs(gs - last_genomic_end) / 2 : gs + position - rs; /* L143-146 */
            return 1;
        }
    }
    if (position - last_read_end < 300) { /* L149-151 */
        *out = last_genomic_end;
        return 1;
    }
    return 0;
}
