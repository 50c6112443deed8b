/*
 * sor_final.c -- ORACLE (test infrastructure; rules in sor_bc.c).
 *
 * End of pass 1: low-count filter, collision merge of the used-barcode list, low-depth cut and rank.
 *   UsedCellBCListGenerator$UsedBarcodesListData.finalizeData  FJ!nanoporereadscanner/analyzers/UsedCellBCListGenerator.java:L379-425
 *     filterLowCounts L359-363
 *   BarcodeDatasetColissionTester                               FJ!nanoporereadscanner/analyzers/BarcodeDatasetColissionTester.java:L68-229
 *   WorkerReadscanner.scan (rank)                               FJ!nanoporereadscanner/WorkerReadscanner.java:L265-273
 *
 * CANONICAL ORDER.  In the reference the order of equal-count barcodes depends on thread timing (pass-1 workers
 * fill a synchronized fastutil map, collision results arrive in a ConcurrentHashMap in completion order) and on
 * fastutil-8.2.2 internals whose jar is missing from the reference checkout.  Equal-count ties are therefore broken
 * here by ascending barcode key -- everywhere a stable sort by count runs (L166-167 and WorkerReadscanner L268).
 * What IS order-dependent and deterministic given that input order is emulated exactly: the java.util.HashMap
 * iteration order of toMergeMap, which decides which removals win (L188-195).  Tree bins (>= 9 keys in one bucket)
 * are not modelled; they need 9 colliding 32-bit keys among the barcodes that have collisions.
 */
#include <stdlib.h>
#include <string.h>

#include "sor.h"

typedef struct {
    int64_t key;
    uint32_t count;
} kc;

static int cmp_key(const void *a, const void *b) {
    int64_t x = ((const kc *)a)->key, y = ((const kc *)b)->key;
    return x < y ? -1 : x > y;
}

/* stable by construction: count desc, then key asc */
static int cmp_count_desc_key(const void *a, const void *b) {
    const kc *x = (const kc *)a, *y = (const kc *)b;
    if (x->count != y->count) return x->count > y->count ? -1 : 1;
    return x->key < y->key ? -1 : x->key > y->key;
}

/* java.util.HashMap<Long, ...> iteration order for keys inserted in the given order:
 * hash = (int)(k ^ k>>>32), spread h ^ h>>>16, table 16 doubling when size > 0.75 cap, bins in insertion order */
static void jhashmap_order(const int64_t *keys, size_t n, size_t *order) {
    size_t cap = 16;
    while (n > (cap * 3) / 4) cap <<= 1; /* final capacity; splits preserve relative order, so only it matters */
    uint32_t *idx = (uint32_t *)malloc(n * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) {
        uint64_t k = (uint64_t)keys[i];
        uint32_t h = (uint32_t)(k ^ (k >> 32));
        h ^= h >> 16;
        idx[i] = h & (uint32_t)(cap - 1);
    }
    /* counting sort by bucket, stable */
    size_t *start = (size_t *)calloc(cap + 1, sizeof(size_t));
    for (size_t i = 0; i < n; i++) start[idx[i] + 1]++;
    for (size_t b = 0; b < cap; b++) start[b + 1] += start[b];
    for (size_t i = 0; i < n; i++) order[start[idx[i]]++] = i;
    free(start);
    free(idx);
}

int sor_finalize_used_list(const int64_t *keys, const uint32_t *counts, size_t n, uint32_t record_count, int merge_ed,
                           int min_count_fold, int cells_fold_below_max, int64_t *out_keys, uint32_t *out_counts,
                           uint32_t *out_rank, size_t *n_out) {
    *n_out = 0;
    if (n == 0) return 0;
    /* filterLowCounts L359-363: (float)count > cutoff && count > 1, cutoff = 2f * recordCount / 5e6f (L391) */
    const float cutoff = (2.0f * (float)record_count) / 5000000.0f;
    kc *f = (kc *)malloc(n * sizeof(kc));
    size_t nf = 0;
    for (size_t i = 0; i < n; i++)
        if ((float)counts[i] > cutoff && counts[i] > 1) {
            f[nf].key = keys[i];
            f[nf].count = counts[i];
            nf++;
        }
    qsort(f, nf, sizeof(kc), cmp_key);
    if (nf == 0) {
        free(f);
        return 0;
    }
    int64_t *fk = (int64_t *)malloc(nf * sizeof(int64_t));
    for (size_t i = 0; i < nf; i++) fk[i] = f[i].key;
    sor_set *set = sor_set_new(fk, nf);
    /* count lookup by key: binary search in the sorted array */
#define COUNT_OF(k, out)                                  \
    do {                                                  \
        size_t lo_ = 0, hi_ = nf;                         \
        while (lo_ < hi_) {                               \
            size_t mid_ = (lo_ + hi_) / 2;                \
            if (f[mid_].key < (k))                        \
                lo_ = mid_ + 1;                           \
            else                                          \
                hi_ = mid_;                               \
        }                                                 \
        (out) = lo_;                                      \
    } while (0)
    /* collisions per barcode: BarcodeMatchTester(seq, ed, skipFullMatches, allowIndels, keys, 0, 16, null, false) L213-225 */
    typedef struct {
        size_t self;   /* index into f */
        int n;
        int64_t bc[64];
        int ed[64];
    } coll;
    coll *cs = (coll *)malloc(nf * sizeof(coll));
    size_t nc = 0;
    for (size_t i = 0; i < nf; i++) {
        sor_match_t m[64];
        int nm = sor_bc_match(set, f[i].key, 16, merge_ed, 1, 1, NULL, 0, 0, 0, m, 64, NULL);
        if (nm <= 0) continue; /* FutCallBack.onSuccess: stored only when non-empty (L240-241) */
        cs[nc].self = i;
        cs[nc].n = nm;
        for (int j = 0; j < nm; j++) {
            cs[nc].bc[j] = m[j].matching_bc;
            cs[nc].ed[j] = m[j].ed;
        }
        nc++;
    }
    /* L166-167: entries sorted by count of the key, descending (canonical tie: key ascending) */
    kc *ord = (kc *)malloc((nc ? nc : 1) * sizeof(kc));
    for (size_t i = 0; i < nc; i++) {
        ord[i].key = (int64_t)i; /* index into cs */
        ord[i].count = f[cs[i].self].count;
    }
    /* sort indices by (count desc, barcode key asc) */
    for (size_t i = 1; i < nc; i++) { /* insertion sort is fine for test sizes; keep it simple and stable */
        kc x = ord[i];
        size_t j = i;
        while (j > 0) {
            const kc *p = &ord[j - 1];
            int64_t pk = f[cs[p->key].self].key, xk = f[cs[x.key].self].key;
            int before = (p->count > x.count) || (p->count == x.count && pk < xk);
            if (before) break;
            ord[j] = ord[j - 1];
            j--;
        }
        ord[j] = x;
    }
    /* toMergeMap: key -> set of colliding barcodes with ed <= merge_ed and count < count(key)/factor (L168-181) */
    int64_t *mk = (int64_t *)malloc((nc ? nc : 1) * sizeof(int64_t));
    for (size_t i = 0; i < nc; i++) mk[i] = f[cs[ord[i].key].self].key;
    size_t *it = (size_t *)malloc((nc ? nc : 1) * sizeof(size_t));
    jhashmap_order(mk, nc, it);
    uint8_t *alive = (uint8_t *)malloc(nf);
    memset(alive, 1, nf);
    for (size_t t = 0; t < nc; t++) { /* L188-195: iteration order of the HashMap; skip keys already removed */
        const coll *c = &cs[ord[it[t]].key];
        if (!alive[c->self]) continue;
        uint32_t cut = f[c->self].count / (uint32_t)min_count_fold;
        for (int j = 0; j < c->n; j++) {
            if (c->ed[j] > merge_ed) continue;
            size_t p;
            COUNT_OF(c->bc[j], p);
            if (p >= nf || f[p].key != c->bc[j]) continue;
            if (f[p].count < cut) alive[p] = 0; /* Long2ObjectMap.remove; absent keys are a no-op */
        }
    }
    /* L198-201: drop barcodes below max / cellsWithReadsnFoldBelowMaxToKeep */
    uint32_t mx = 0;
    for (size_t i = 0; i < nf; i++)
        if (alive[i] && f[i].count > mx) mx = f[i].count;
    uint32_t min_counts = mx / (uint32_t)cells_fold_below_max;
    kc *fin = (kc *)malloc(nf * sizeof(kc));
    size_t no = 0;
    for (size_t i = 0; i < nf; i++)
        if (alive[i] && f[i].count >= min_counts) fin[no++] = f[i];
    /* rank = position in count-descending order, 1-based (WorkerReadscanner.java:L266-270) */
    qsort(fin, no, sizeof(kc), cmp_count_desc_key);
    for (size_t i = 0; i < no; i++) {
        out_keys[i] = fin[i].key;
        out_counts[i] = fin[i].count;
        out_rank[i] = (uint32_t)(i + 1);
    }
    *n_out = no;
    free(fin);
    free(alive);
    free(it);
    free(mk);
    free(ord);
    free(cs);
    sor_set_free(set);
    free(fk);
    free(f);
    return 0;
#undef COUNT_OF
}
