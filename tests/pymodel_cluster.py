"""Independent plain-Python restatement of the reference's UMI clustering of one (cell, region) group, object style
(dendrogram nodes, a sorted entry list for LingPipe's BoundedPriorityQueue, dict-of-sets index, an emulated fastutil
IntOpenHashSet) -- NOT sharing code with oracle/sor_cluster.c.  Small groups only.

Cites: FJ!umifinder/analyzers/clustering/{ClusterOneHierarchical.java:L66-217,ClusterOne_MyClustering.java:L59-219,
ClusterOneBase.java:L118-168,UmiClustering$Submitter.java:L239-261}, FJ!clustering/{DistanceMatrix.java:L87-169,
OneUmiCluster.java:L38-89}, AL!cluster/{CompleteLinkClusterer.java:L146-237,SingleLinkClusterer.java:L198-268,
Dendrogram.java:L205-215}, AL!util/BoundedPriorityQueue.java:L144-153,L342-346,L458-464.

Canonical rules where the reference itself is not reproducible (DESIGN.md "UMI clustering"): group members in input
order; HashSet<PairScore> (identity hashes) iterated in creation order; fastutil collections filled in ascending
index order.
"""
import math

PHI = 0x9E3779B9


def mix(x):
    h = (x * PHI) & 0xFFFFFFFF
    return h ^ (h >> 16)


class IntOpenHashSet:
    """it.unimi.dsi.fastutil.ints.IntOpenHashSet 8.2.2 (published algorithm, restated; jar not in the checkout)"""

    def __init__(self):
        self.n = 32  # arraySize(16, 0.75)
        self.key = [0] * (self.n + 1)
        self.size = 0
        self.has_zero = False

    def _max_fill(self):
        return min(math.ceil(self.n * 0.75), self.n - 1)

    def add(self, k):
        if k == 0:
            if self.has_zero:
                return
            self.has_zero = True
        else:
            mask = self.n - 1
            pos = mix(k) & mask
            while self.key[pos] != 0:
                if self.key[pos] == k:
                    return
                pos = (pos + 1) & mask
            self.key[pos] = k
        self.size += 1
        if self.size - 1 >= self._max_fill():
            need = math.ceil((self.size + 1) / 0.75)
            new_n = 2
            while new_n < need:
                new_n *= 2
            self._rehash(new_n)

    def _rehash(self, new_n):
        mask = new_n - 1
        new_key = [0] * (new_n + 1)
        i = self.n
        real = self.size - (1 if self.has_zero else 0)
        for _ in range(real):
            i -= 1
            while self.key[i] == 0:
                i -= 1
            pos = mix(self.key[i]) & mask
            while new_key[pos] != 0:
                pos = (pos + 1) & mask
            new_key[pos] = self.key[i]
        self.n, self.key = new_n, new_key

    def __iter__(self):
        if self.has_zero:
            yield 0
        for pos in range(self.n - 1, -1, -1):
            if self.key[pos] != 0:
                yield self.key[pos]


def fastutil_order(members):
    s = IntOpenHashSet()
    for m in sorted(members):
        s.add(m)
    return list(s)


class Node:
    def __init__(self, members, score=0.0, kids=None):
        self.members, self.score, self.kids, self.parent = members, score, kids, None

    def root(self):
        x = self
        while x.parent is not None:
            x = x.parent
        return x


def complete_link(k, dist, max_distance):
    """CompleteLinkClusterer.hierarchicalCluster + Dendrogram.partitionDistance -> list of member sets"""
    if k == 1:
        return [{0}]
    leafs = [Node({i}) for i in range(k)]
    queue = []  # entries (score, -id, pair); the TreeSet's first = smallest score, then LARGEST id
    index = {id(x): [] for x in leafs}
    next_id = [0]

    def offer(ps):
        next_id[0] += 1
        queue.append((ps["score"], -next_id[0], ps))

    for i in range(k):
        for j in range(i + 1, k):
            ps = dict(a=leafs[i], b=leafs[j], score=float(dist(i, j)))
            offer(ps)
            index[id(leafs[i])].append(ps)
            index[id(leafs[j])].append(ps)
    d12 = None
    while queue:
        queue.sort(key=lambda e: (e[0], e[1]))
        _, _, nxt = queue.pop(0)
        d1, d2 = nxt["a"].root(), nxt["b"].root()
        d12 = Node(d1.members | d2.members, nxt["score"], (d1, d2))
        d1.parent = d2.parent = d12
        index[id(d12)] = []
        buf = {}
        set1 = index.pop(id(d1))
        queue = [e for e in queue if not any(e[2] is p for p in set1)]
        for ps3 in set1:
            d3 = ps3["b"] if ps3["a"] is d1 else ps3["a"]
            index[id(d3)] = [p for p in index[id(d3)] if p is not ps3]
            buf[id(d3)] = ps3["score"]
        set2 = index.pop(id(d2))
        queue = [e for e in queue if not any(e[2] is p for p in set2)]
        for ps3 in set2:
            d3 = ps3["b"] if ps3["a"] is d2 else ps3["a"]
            index[id(d3)] = [p for p in index[id(d3)] if p is not ps3]
            if id(d3) not in buf:
                continue
            ps = dict(a=d12, b=d3, score=max(buf[id(d3)], ps3["score"]))
            offer(ps)
            index[id(d12)].append(ps)
            index[id(d3)].append(ps)
    out, stack = [], [d12]
    while stack:
        cur = stack.pop(0)
        if cur.score <= max_distance:
            out.append(set(cur.members))
        else:
            stack[0:0] = [cur.kids[1], cur.kids[0]]
    return out


def single_link(k, dist, max_distance):
    parent = list(range(k))

    def find(x):
        while parent[x] != x:
            x = parent[x]
        return x

    pairs = sorted(((dist(i, j), i, j) for i in range(k) for j in range(i + 1, k)), key=lambda t: t[0])
    for d, i, j in pairs:
        if d > max_distance:
            break
        a, b = find(i), find(j)
        if a != b:
            parent[a] = b
    groups = {}
    for i in range(k):
        groups.setdefault(find(i), set()).add(i)
    return list(groups.values())


def jround_double(x):
    return int(math.floor(x + 0.5))


def set_center(members, ed, qv):
    """OneUmiCluster.setClusterCenterNotPreGrouped L49-65"""
    order = fastutil_order(members)
    if len(order) == 1:
        return order[0]
    if len(order) == 2:
        return order[0] if qv[0] > qv[1] else order[1]  # compares reads 0 and 1 of the GROUP (L53)
    best, best_sum = None, None
    for s in order:
        tot = sum(int(math.pow(ed(s, w), 2.0)) for w in order if w != s)
        if best_sum is None or tot < best_sum:
            best, best_sum = s, tot
    return best


def assign(out, cluster, center, clusters_all, n, ed, pos1, pos2, skipped, filtered=None):
    order = fastutil_order(cluster)
    offs = [pos1(center, v) - 1 for v in order if v != center]
    offset = jround_double(sum(offs) / len(offs))
    for idx in (filtered if filtered is not None else order):
        if skipped[idx]:
            continue
        sec = None
        if len(clusters_all) > 1:
            others = [ed(idx, m) for m in range(n) if m not in cluster]
            sec = min(others) if others else None
        out[idx] = dict(center=center, offset=offset, ed=ed(center, idx), ed_second=sec, pos2=pos2(center, idx))


def cluster_group(mat, n, qv, complete_ed=2, single_ed=1, single_switch=3000, fold=50, own_above=100):
    """mat: n*n packed bytes ed | pos1 << 4 | pos2 << 6.  -> (assignments list (dict or None), skipped flags)"""
    ed = lambda i, j: mat[i * n + j] & 15  # noqa: E731
    pos1 = lambda i, j: (mat[i * n + j] >> 4) & 3  # noqa: E731
    pos2 = lambda i, j: (mat[i * n + j] >> 6) & 3  # noqa: E731
    out = [None] * n
    skipped = [False] * n
    if n <= 1:
        return out, skipped
    if n <= own_above:  # ClusterOneHierarchical
        nb = [i for i in range(n) if any(i != j and ed(i, j) <= complete_ed for j in range(n))]
        k = len(nb)
        if k <= 1:
            return out, skipped
        dist = lambda a, b: ed(nb[a], nb[b])  # noqa: E731
        parts = single_link(k, dist, single_ed) if k > single_switch else complete_link(k, dist, complete_ed)
        clusters = [{nb[a] for a in p} for p in parts if len(p) > 1]
        if not clusters:
            return out, skipped
        mx = max(len(c) for c in clusters)
        kept = []
        for c in clusters:
            if len(c) * fold > mx:
                kept.append(c)
            else:
                for m in c:
                    skipped[m] = True
        for c in kept:
            assign(out, c, set_center(c, ed, qv), kept, n, ed, pos1, pos2, skipped)
        return out, skipped
    # ClusterOne_MyClustering
    def cluster_local(indices):
        nbh = {a: {v for v in indices if ed(a, v) <= complete_ed} for a in indices}
        keys = [a for a in indices if len(nbh[a]) > 1]
        order = fastutil_order_keys(keys)
        owner = {}
        for c in keys:
            best = None
            for a in order:
                if c in nbh[a] and (best is None or len(nbh[a]) > len(nbh[best])):
                    best = a
            owner.setdefault(best, set()).add(c)
        return list(owner.values())

    clusters = cluster_local(list(range(n)))
    if not clusters:
        return out, skipped
    mx = max(len(c) for c in clusters)
    kept = []
    for c in sorted(clusters, key=min):
        if len(c) * fold > mx:
            kept.append([set(c), None])
        else:
            for m in c:
                skipped[m] = True
    for kc in kept:
        kc[1] = set_center(kc[0], ed, qv)
    clustered = set().union(*[kc[0] for kc in kept]) if kept else set()
    unclustered = [i for i in range(n) if i not in clustered]
    removed = []
    for kc in kept:
        rem = [s for s in fastutil_order(kc[0]) if ed(s, kc[1]) > complete_ed]
        if rem:
            kc[0] -= set(rem)
            kc[1] = set_center(kc[0], ed, qv)
        removed += rem
    unclustered += removed
    if removed:
        for c in sorted(cluster_local(unclustered), key=min):
            if len(c) > 1:
                kept.append([set(c), set_center(c, ed, qv)])
    all_sets = [kc[0] for kc in kept]
    for members, center in kept:
        if len(members) <= 1:
            continue
        filt = [s for s in fastutil_order(members) if ed(s, center) <= complete_ed]
        if len(filt) > 1:
            assign(out, members, center, all_sets, n, ed, pos1, pos2, skipped, filtered=filt)
    return out, skipped


def fastutil_order_keys(keys):
    """Int2ObjectOpenHashMap entry order = the same open-addressing layout as IntOpenHashSet"""
    return fastutil_order(keys)
