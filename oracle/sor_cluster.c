/*
 * sor_cluster.c -- ORACLE (test infrastructure; see sor_bc.c for the rules).
 *
 * CPU restatement of the reference's UMI clustering of one (cell barcode, genomic region) group of `assignumis`:
 *   UmiClustering$Submitter.lambda$run$2            FJ!umifinder/analyzers/clustering/UmiClustering$Submitter.java:L239-261
 *   ClusterOneHierarchical.call                     FJ!umifinder/analyzers/clustering/ClusterOneHierarchical.java:L66-217
 *   ClusterOne_MyClustering.call / clusterLocal     FJ!umifinder/analyzers/clustering/ClusterOne_MyClustering.java:L59-219
 *   ClusterOneBase.setSamflagsAndStatsForClustered  FJ!umifinder/analyzers/clustering/ClusterOneBase.java:L118-168
 *   DistanceMatrix                                  FJ!clustering/DistanceMatrix.java:L87-169
 *   OneUmiCluster.setClusterCenterNotPreGrouped     FJ!clustering/OneUmiCluster.java:L49-65
 *   CompleteLinkClusterer / SingleLinkClusterer / Dendrogram.partitionDistance / BoundedPriorityQueue
 *                                                   AL!cluster/CompleteLinkClusterer.java:L146-237 (lingpipe-4.1.2-JL1.0.jar),
 *                                                   AL!cluster/SingleLinkClusterer.java:L198-268, AL!cluster/Dendrogram.java:L205-215,
 *                                                   AL!util/BoundedPriorityQueue.java:L144-153,L342-346,L458-464
 *   it.unimi.dsi.fastutil.ints.IntOpenHashSet (fastutil 8.2.2, jar NOT in the checkout): published open-addressing
 *   layout restated from memory -- mix(k) = (h = k * 0x9E3779B9) ^ (h >>> 16), linear probing, table 32 growing by
 *   doubling at 3/4 load, iteration: key 0 first, then slots from the top down.
 *
 * The reference is NOT reproducible on this step: group members arrive in the order of a parallel
 * stream, LingPipe iterates HashSet<PairScore> by identity hash, and parallel collectors fill the fastutil maps.
 * Canonical rules used here and in the product: members in input order; PairScore sets in creation order; fastutil
 * collections filled in ascending index order.  PARITY UNPINNED (no reference tests, no JVM; sor_bc.c has the status of all
 * oracle files).  Held by ref_exec_cluster.json (70 groups under eight hash orders), ref_exec_cluster_own{,2}.json (ClusterOne_MyClustering);
 * NOT pinned: centres of 1- and 2-read clusters and owners on size ties (fastutil iteration order).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "sor.h"

/* ---- fastutil IntOpenHashSet iteration order of a set of ints (filled ascending) ------------------------------ */
static uint32_t fu_mix(uint32_t x) {
    uint32_t h = x * 0x9E3779B9u;
    return h ^ (h >> 16);
}

/* members ascending in[0..m) -> iteration order out[0..m) */
static void fastutil_order(const int *in, int m, int *out) {
    int n = 32, size = 0, has_zero = 0;
    int *key = calloc((size_t)n + 1, sizeof(int));
    for (int t = 0; t < m; t++) {
        int k = in[t];
        if (k == 0) {
            has_zero = 1;
        } else {
            int pos = (int)(fu_mix((uint32_t)k) & (uint32_t)(n - 1));
            while (key[pos] != 0) pos = (pos + 1) & (n - 1);
            key[pos] = k;
        }
        int max_fill = (int)ceil(n * 0.75);
        if (max_fill > n - 1) max_fill = n - 1;
        if (size++ >= max_fill) { /* rehash(arraySize(size + 1, f)) */
            int need = (int)ceil((size + 1) / 0.75), nn = 2;
            while (nn < need) nn <<= 1;
            int *nk = calloc((size_t)nn + 1, sizeof(int));
            int i = n, real = size - has_zero;
            for (int j = 0; j < real; j++) {
                do i--; while (key[i] == 0);
                int pos = (int)(fu_mix((uint32_t)key[i]) & (uint32_t)(nn - 1));
                while (nk[pos] != 0) pos = (pos + 1) & (nn - 1);
                nk[pos] = key[i];
            }
            free(key);
            key = nk;
            n = nn;
        }
    }
    int o = 0;
    if (has_zero) out[o++] = 0;
    for (int pos = n - 1; pos >= 0; pos--)
        if (key[pos] != 0) out[o++] = key[pos];
    free(key);
}

#define ED(i, j) ((int)(mat[(size_t)(i) * n + (j)] & 15))
#define POS1(i, j) ((int)((mat[(size_t)(i) * n + (j)] >> 4) & 3))
#define POS2(i, j) ((int)((mat[(size_t)(i) * n + (j)] >> 6) & 3))

/* OneUmiCluster.setClusterCenterNotPreGrouped: members ascending */
static int set_center(const uint8_t *mat, int n, const int *members, int m, const float *qv) {
    int *ord = malloc(sizeof(int) * (size_t)m);
    fastutil_order(members, m, ord);
    int center;
    if (m == 1)
        center = ord[0];
    else if (m == 2)
        center = qv[0] > qv[1] ? ord[0] : ord[1]; /* L53: reads 0 and 1 of the group, whatever the cluster holds */
    else {
        long best = -1;
        center = ord[0];
        for (int a = 0; a < m; a++) {
            long tot = 0;
            for (int b = 0; b < m; b++)
                if (ord[b] != ord[a]) tot += (long)(int)pow((double)ED(ord[a], ord[b]), 2.0);
            if (best < 0 || tot < best) { /* stable sorted(): first of the smallest */
                best = tot;
                center = ord[a];
            }
        }
    }
    free(ord);
    return center;
}

/* ClusterOne*.lambda (offset mean) + ClusterOneBase.setSamflagsAndStatsForClustered for the members listed in `who` */
static void assign_cluster(const uint8_t *mat, int n, const int *members, int m, int center, const int *who, int n_who,
                           const char *in_cluster, int n_clusters, const char *skipped, sor_umi_assignment *out) {
    int *ord = malloc(sizeof(int) * (size_t)m);
    fastutil_order(members, m, ord);
    long sum = 0;
    int cnt = 0;
    for (int a = 0; a < m; a++)
        if (ord[a] != center) {
            sum += POS1(center, ord[a]) - 1; /* PlusMinusOneEnum.getOffSet = value - 1 */
            cnt++;
        }
    free(ord);
    const int offset = (int)floor((double)sum / (double)cnt + 0.5); /* (int) Math.round(double) */
    for (int t = 0; t < n_who; t++) {
        const int idx = who[t];
        if (skipped[idx]) continue; /* UMI_CLUSTERING_SKIPPED_HIGHCOMPLEXITY, L122 */
        int sec = -1;
        if (n_clusters > 1) /* L161-164 */
            for (int mth = 0; mth < n; mth++)
                if (!in_cluster[mth] && (sec < 0 || ED(idx, mth) < sec)) sec = ED(idx, mth);
        out[idx].center = center;
        out[idx].offset = (int8_t)offset;
        out[idx].ed = (int8_t)ED(center, idx);
        out[idx].ed_second = (int8_t)sec;
        out[idx].pos2 = (int8_t)POS2(center, idx);
    }
}

/* ---- CompleteLinkClusterer on k elements with distance d(a,b) = ED(nb[a], nb[b]); labels[a] = partition id -------- */
typedef struct {
    int a, b; /* dendrogram node ids */
    double score;
    int active;
} pair_t;

static void complete_link(const uint8_t *mat, int n, const int *nb, int k, double max_distance, int *label) {
    const int max_nodes = 2 * k, max_pairs = k * (k - 1) / 2 + k * k + 8;
    int *parent = malloc(sizeof(int) * (size_t)max_nodes);
    double *nscore = calloc((size_t)max_nodes, sizeof(double));
    int(*kids)[2] = malloc(sizeof(int[2]) * (size_t)max_nodes);
    pair_t *ps = malloc(sizeof(pair_t) * (size_t)max_pairs);
    int n_nodes = k, n_pairs = 0;
    for (int i = 0; i < max_nodes; i++) parent[i] = -1;
    for (int i = 0; i < k; i++)
        for (int j = i + 1; j < k; j++) ps[n_pairs++] = (pair_t){i, j, (double)ED(nb[i], nb[j]), 1};
    int root = 0;
    double *buf = malloc(sizeof(double) * (size_t)max_nodes);
    char *has = malloc((size_t)max_nodes);
    for (;;) {
        int best = -1; /* TreeSet.first(): smallest score, among equals the LARGEST id (EntryComparator L458-464) */
        for (int p = 0; p < n_pairs; p++)
            if (ps[p].active && (best < 0 || ps[p].score < ps[best].score || (ps[p].score == ps[best].score && p > best)))
                best = p;
        if (best < 0) break;
        ps[best].active = 0;
        int d1 = ps[best].a, d2 = ps[best].b;
        while (parent[d1] >= 0) d1 = parent[d1];
        while (parent[d2] >= 0) d2 = parent[d2];
        const int d12 = n_nodes++;
        parent[d1] = parent[d2] = d12;
        kids[d12][0] = d1;
        kids[d12][1] = d2;
        nscore[d12] = ps[best].score;
        root = d12;
        memset(has, 0, (size_t)max_nodes);
        for (int p = 0; p < n_pairs; p++) /* pairs of dendro1 (the popped one maps dendro2 -> dist12) */
            if ((ps[p].active || p == best) && (ps[p].a == d1 || ps[p].b == d1)) {
                const int d3 = ps[p].a == d1 ? ps[p].b : ps[p].a;
                buf[d3] = ps[p].score;
                has[d3] = 1;
                ps[p].active = 0;
            }
        const int lim = n_pairs;
        for (int p = 0; p < lim; p++) /* pairs of dendro2, creation order (canonical) */
            if (ps[p].active && (ps[p].a == d2 || ps[p].b == d2)) {
                const int d3 = ps[p].a == d2 ? ps[p].b : ps[p].a;
                ps[p].active = 0;
                if (!has[d3]) continue;
                const double dd = buf[d3] > ps[p].score ? buf[d3] : ps[p].score;
                ps[n_pairs++] = (pair_t){d12, d3, dd, 1};
            }
    }
    /* Dendrogram.partitionDistance */
    int *stack = malloc(sizeof(int) * (size_t)max_nodes), sp = 0, n_label = 0;
    stack[sp++] = root;
    while (sp > 0) {
        const int cur = stack[--sp];
        if (nscore[cur] <= max_distance) {
            int *st2 = malloc(sizeof(int) * (size_t)max_nodes), s2 = 0;
            st2[s2++] = cur;
            while (s2 > 0) {
                const int x = st2[--s2];
                if (x < k)
                    label[x] = n_label;
                else {
                    st2[s2++] = kids[x][0];
                    st2[s2++] = kids[x][1];
                }
            }
            free(st2);
            n_label++;
        } else {
            stack[sp++] = kids[cur][0];
            stack[sp++] = kids[cur][1];
        }
    }
    free(stack);
    free(has);
    free(buf);
    free(ps);
    free(kids);
    free(nscore);
    free(parent);
}

static void single_link(const uint8_t *mat, int n, const int *nb, int k, double max_distance, int *label) {
    for (int i = 0; i < k; i++) label[i] = i;
    for (int changed = 1; changed;) { /* connected components of the "<= cut-off" graph */
        changed = 0;
        for (int i = 0; i < k; i++)
            for (int j = i + 1; j < k; j++)
                if ((double)ED(nb[i], nb[j]) <= max_distance && label[i] != label[j]) {
                    const int lo = label[i] < label[j] ? label[i] : label[j];
                    label[i] = label[j] = lo;
                    changed = 1;
                }
    }
}

/* clusterLocal L175-219 on `indices` (ascending); owner[c] = key of its cluster or -1 */
static void cluster_local(const uint8_t *mat, int n, const int *indices, int m, int ed, int *owner) {
    int *cnt = calloc((size_t)n, sizeof(int));
    for (int a = 0; a < m; a++)
        for (int b = 0; b < m; b++)
            if (ED(indices[a], indices[b]) <= ed) cnt[indices[a]]++;
    int *keys = malloc(sizeof(int) * (size_t)m), nk = 0;
    for (int a = 0; a < m; a++)
        if (cnt[indices[a]] > 1) keys[nk++] = indices[a];
    int *ord = malloc(sizeof(int) * (size_t)(nk > 0 ? nk : 1));
    fastutil_order(keys, nk, ord); /* Int2ObjectOpenHashMap entry order */
    for (int i = 0; i < n; i++) owner[i] = -1;
    for (int t = 0; t < nk; t++) {
        const int c = keys[t];
        int best = -1;
        for (int u = 0; u < nk; u++) { /* Stream.max keeps the first of the largest */
            const int a = ord[u];
            if (ED(a, c) <= ed && (best < 0 || cnt[a] > cnt[best])) best = a;
        }
        owner[c] = best;
    }
    free(ord);
    free(keys);
    free(cnt);
}

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

int sor_umi_cluster_group(const uint8_t *mat, int32_t n, const float *mean_qv, const sor_umi_cluster_params *par,
                          sor_umi_assignment *out, uint8_t *skipped_out) {
    char *skipped = calloc((size_t)(n > 0 ? n : 1), 1);
    for (int i = 0; i < n; i++) {
        out[i].center = -1;
        out[i].offset = 0;
        out[i].ed = -1;
        out[i].ed_second = -1;
        out[i].pos2 = 0;
    }
    if (n <= 1) goto done;
    const int ced = par->complete_link_ed;
    if (n <= par->own_clusterer_above) { /* ---- ClusterOneHierarchical ---- */
        int *nb = malloc(sizeof(int) * (size_t)n), k = 0;
        for (int i = 0; i < n; i++) { /* generateIndicesWithNeighbours L87-88 */
            int any = 0;
            for (int j = 0; j < n && !any; j++) any = i != j && ED(i, j) <= ced;
            if (any) nb[k++] = i;
        }
        if (k > 1) {
            int *label = malloc(sizeof(int) * (size_t)k);
            if (k > par->single_link_switch)
                single_link(mat, n, nb, k, (double)par->single_link_ed, label);
            else
                complete_link(mat, n, nb, k, (double)ced, label);
            /* clusters of size > 1 (L101), fold-depth filter (L123), centre, tags */
            int *size = calloc((size_t)k, sizeof(int)), mx = 0, n_kept = 0;
            for (int a = 0; a < k; a++) size[label[a]]++;
            for (int l = 0; l < k; l++)
                if (size[l] > 1 && size[l] > mx) mx = size[l];
            for (int l = 0; l < k; l++)
                if (size[l] > 1 && size[l] * par->fold_depth_below_max > mx) n_kept++;
            for (int l = 0; l < k; l++) {
                if (size[l] <= 1) continue;
                int *mem = malloc(sizeof(int) * (size_t)size[l]), m = 0;
                for (int a = 0; a < k; a++)
                    if (label[a] == l) mem[m++] = nb[a];
                if (!(size[l] * par->fold_depth_below_max > mx)) {
                    for (int t = 0; t < m; t++) skipped[mem[t]] = 1;
                } else {
                    char *in = calloc((size_t)n, 1);
                    for (int t = 0; t < m; t++) in[mem[t]] = 1;
                    const int center = set_center(mat, n, mem, m, mean_qv);
                    int *ord = malloc(sizeof(int) * (size_t)m);
                    fastutil_order(mem, m, ord);
                    assign_cluster(mat, n, mem, m, center, ord, m, in, n_kept, skipped, out);
                    free(ord);
                    free(in);
                }
                free(mem);
            }
            free(size);
            free(label);
        }
        free(nb);
    } else { /* ---- ClusterOne_MyClustering ---- */
        int *all = malloc(sizeof(int) * (size_t)n), *owner = malloc(sizeof(int) * (size_t)n);
        for (int i = 0; i < n; i++) all[i] = i;
        cluster_local(mat, n, all, n, ced, owner);
        /* clusters keyed by owner; canonical list order: ascending smallest member */
        int *cid = malloc(sizeof(int) * (size_t)n); /* cluster id of a read in the kept list, -1 none */
        int *csize = calloc((size_t)2 * n + 2, sizeof(int)), *ccenter = malloc(sizeof(int) * ((size_t)2 * n + 2));
        int *first_of_owner = malloc(sizeof(int) * (size_t)n);
        int n_cl = 0, mx = 0;
        for (int i = 0; i < n; i++) {
            cid[i] = -1;
            first_of_owner[i] = -1;
        }
        int *osize = calloc((size_t)n, sizeof(int));
        for (int i = 0; i < n; i++)
            if (owner[i] >= 0) osize[owner[i]]++;
        for (int i = 0; i < n; i++)
            if (osize[i] > mx) mx = osize[i];
        for (int i = 0; i < n; i++) { /* ascending smallest member = first time an owner is seen */
            const int o = owner[i];
            if (o < 0) continue;
            if (first_of_owner[o] == -1) first_of_owner[o] = osize[o] * par->fold_depth_below_max > mx ? n_cl++ : -2;
            if (first_of_owner[o] == -2)
                skipped[i] = 1;
            else {
                cid[i] = first_of_owner[o];
                csize[cid[i]]++;
            }
        }
        int *mem = malloc(sizeof(int) * (size_t)n);
#define MEMBERS(c, m)            \
    do {                         \
        (m) = 0;                 \
        for (int q = 0; q < n; q++) \
            if (cid[q] == (c)) mem[(m)++] = q; \
    } while (0)
        for (int c = 0; c < n_cl; c++) {
            int m;
            MEMBERS(c, m);
            ccenter[c] = set_center(mat, n, mem, m, mean_qv);
        }
        /* unclustered (L91) is taken BEFORE the off-centre removal (L102) */
        int *uncl = malloc(sizeof(int) * (size_t)n), n_un = 0, n_removed = 0;
        for (int i = 0; i < n; i++)
            if (cid[i] < 0) uncl[n_un++] = i;
        for (int c = 0; c < n_cl; c++) { /* removeOffCenter lambda$call$1 L61-64 */
            int m, any = 0;
            MEMBERS(c, m);
            int *ord = malloc(sizeof(int) * (size_t)m);
            fastutil_order(mem, m, ord);
            for (int t = 0; t < m; t++)
                if (ED(ord[t], ccenter[c]) > ced) {
                    cid[ord[t]] = -1;
                    uncl[n_un++] = ord[t];
                    n_removed++;
                    any = 1;
                }
            free(ord);
            if (any) {
                MEMBERS(c, m);
                ccenter[c] = set_center(mat, n, mem, m, mean_qv);
            }
        }
        if (n_removed > 0) { /* L106-112 */
            qsort(uncl, (size_t)n_un, sizeof(int), cmp_int);
            cluster_local(mat, n, uncl, n_un, ced, owner);
            for (int i = 0; i < n; i++) {
                first_of_owner[i] = -1;
                osize[i] = 0;
            }
            for (int t = 0; t < n_un; t++)
                if (owner[uncl[t]] >= 0) osize[owner[uncl[t]]]++;
            for (int t = 0; t < n_un; t++) {
                const int i = uncl[t], o = owner[i];
                if (o < 0 || osize[o] <= 1) continue;
                if (first_of_owner[o] == -1) first_of_owner[o] = n_cl++;
                cid[i] = first_of_owner[o];
            }
        }
        /* centres of the additional clusters (L111) */
        for (int c = 0; c < n_cl; c++) {
            int m;
            MEMBERS(c, m);
            if (csize[c] == 0 && m > 0) ccenter[c] = set_center(mat, n, mem, m, mean_qv); /* additional cluster */
        }
        for (int c = 0; c < n_cl; c++) { /* L123: size > 1 */
            int m;
            MEMBERS(c, m);
            if (m <= 1) continue;
            int *ord = malloc(sizeof(int) * (size_t)m), *filt = malloc(sizeof(int) * (size_t)m), nf = 0;
            fastutil_order(mem, m, ord);
            for (int t = 0; t < m; t++)
                if (ED(ord[t], ccenter[c]) <= ced) filt[nf++] = ord[t];
            if (nf > 1) {
                char *in = calloc((size_t)n, 1);
                for (int t = 0; t < m; t++) in[mem[t]] = 1;
                assign_cluster(mat, n, mem, m, ccenter[c], filt, nf, in, n_cl, skipped, out);
                free(in);
            }
            free(filt);
            free(ord);
        }
#undef MEMBERS
        free(uncl);
        free(mem);
        free(osize);
        free(first_of_owner);
        free(ccenter);
        free(csize);
        free(cid);
        free(owner);
        free(all);
    }
done:
    if (skipped_out) memcpy(skipped_out, skipped, (size_t)(n > 0 ? n : 0));
    free(skipped);
    return 0;
}
