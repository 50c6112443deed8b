// smi_cluster.hip -- UMI clustering of `assignumis` on the K-UMI distance matrices: the host clusterers, and ClusterOne_MyClustering for
// groups above 100 reads with its n^2 loops on the device (umi_cluster_own_device below).
//
// Reference units (bytecode; citation form in DESIGN.md; AL! = Jar/lib/lingpipe-4.1.2-JL1.0.jar):
//   UmiClustering$Submitter.lambda$run$2            FJ!umifinder/analyzers/clustering/UmiClustering$Submitter.java:L239-261
//   ClusterOneHierarchical.call                     FJ!umifinder/analyzers/clustering/ClusterOneHierarchical.java:L66-217
//   ClusterOne_MyClustering.call / clusterLocal     FJ!umifinder/analyzers/clustering/ClusterOne_MyClustering.java:L59-219
//   ClusterOneBase.setSamflagsAndStatsForClustered  FJ!umifinder/analyzers/clustering/ClusterOneBase.java:L118-168
//   DistanceMatrix, OneUmiCluster                   FJ!clustering/{DistanceMatrix.java:L87-169,OneUmiCluster.java:L49-65}
//   CompleteLinkClusterer, SingleLinkClusterer, Dendrogram.partitionDistance, BoundedPriorityQueue
//                                                   AL!cluster/*.java, AL!util/BoundedPriorityQueue.java:L458-464
//   fastutil 8.2.2 IntOpenHashSet / Int2ObjectOpenHashMap iteration order (jar not in the checkout; published layout)
//
// The reference is not reproducible on this step (parallel-stream arrival order, identity-hash HashSet<PairScore>);
// the canonical rules are: group members in input order, PairScore sets in creation order, fastutil collections filled
// in ascending index order (DESIGN.md "UMI clustering").
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <thread>
#include <vector>

#include "smi_internal.h"

namespace smi {
namespace {

// threads the loops INSIDE one group may use (set per worker thread of smi_umi_cluster_groups: a call with fewer groups than threads -- the one
// large group a chunk worker hands over -- gives the spare threads to the group's n^2 loops)
thread_local int g_inner_threads = 1;
template <class F>
void parallel_for(size_t n, size_t min_per_thread, F &&f) {  // f(begin, end) over [0, n) in contiguous pieces
    const int nt = (int)std::min<size_t>((size_t)g_inner_threads, std::max<size_t>(1, n / std::max<size_t>(1, min_per_thread)));
    if (nt <= 1) {
        f((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;  // (a fresh thread starts with g_inner_threads = 1: loops nested inside f stay serial)
    for (int t = 1; t < nt; t++) th.emplace_back([&, t] { f(n * (size_t)t / (size_t)nt, n * (size_t)(t + 1) / (size_t)nt); });
    const int keep = g_inner_threads;
    g_inner_threads = 1;
    f((size_t)0, n / (size_t)nt);
    g_inner_threads = keep;
    for (auto &x : th) x.join();
}

struct Dist {
    const uint8_t *m;
    int n;
    int ed(int i, int j) const { return m[(size_t)i * n + j] & 15; }
    int pos1(int i, int j) const { return (m[(size_t)i * n + j] >> 4) & 3; }
    int pos2(int i, int j) const { return (m[(size_t)i * n + j] >> 6) & 3; }
};

// iteration order of a fastutil open hash set / map that received `keys` in ascending order
std::vector<int> fastutil_order(const std::vector<int> &keys_in) {
    std::vector<int> sorted_copy;
    if (!std::is_sorted(keys_in.begin(), keys_in.end())) {
        sorted_copy = keys_in;
        std::sort(sorted_copy.begin(), sorted_copy.end());
    }
    const std::vector<int> &keys = sorted_copy.empty() ? keys_in : sorted_copy;
    auto mix = [](uint32_t x) {
        const uint32_t h = x * 0x9E3779B9u;
        return h ^ (h >> 16);
    };
    auto max_fill_of = [](size_t n) { return std::min((size_t)std::ceil(n * 0.75), n - 1); };
    // the tables of a rehash are kept by the thread: a call per cluster of a large group would otherwise allocate them every time
    static thread_local std::vector<int> tab_v, nt_v, live_v;
    // The entries of a table in descending slot order, without a branch per slot: whether a slot is taken is a coin toss to the predictor
    // (the table is 40 - 75 % full), and that misprediction per slot -- not the probing -- was most of the time of this function.
    auto entries_descending = [](const int *t, size_t n, int *dst) -> size_t {
        size_t w = 0;
        for (size_t i = n; i-- > 0;) {
            dst[w] = t[i];
            w += t[i] != 0;
        }
        return w;
    };
    size_t n = 32, max_fill = max_fill_of(n);
    tab_v.assign(n + 1, 0);
    int *tab = tab_v.data();
    bool zero = false;
    size_t size = 0;
    for (int k : keys) {
        if (k == 0)
            zero = true;
        else {
            size_t pos = mix((uint32_t)k) & (n - 1);
            while (tab[pos] != 0) pos = (pos + 1) & (n - 1);
            tab[pos] = k;
        }
        if (size++ >= max_fill) {
            const size_t need = (size_t)std::ceil((size + 1) / 0.75);
            size_t nn = 2;
            while (nn < need) nn <<= 1;
            nt_v.assign(nn + 1, 0);
            int *nt = nt_v.data();
            live_v.resize(n + 1);
            const size_t n_live = entries_descending(tab, n, live_v.data());
            for (size_t i = 0; i < n_live; i++) {  // entries move in descending slot order
                const int k2 = live_v[i];
                size_t pos = mix((uint32_t)k2) & (nn - 1);
                while (nt[pos] != 0) pos = (pos + 1) & (nn - 1);
                nt[pos] = k2;
            }
            tab_v.swap(nt_v);
            tab = tab_v.data();
            n = nn;
            max_fill = max_fill_of(n);
        }
    }
    std::vector<int> out(n + 2);
    const size_t z = zero ? 1 : 0;
    if (zero) out[0] = 0;
    const size_t n_out = entries_descending(tab, n, out.data() + z);
    out.resize(z + n_out);
    return out;
}

int choose_center(const Dist &D, const std::vector<int> &members, const float *qv) {
    const std::vector<int> ord = fastutil_order(members);
    if (ord.size() == 1) return ord[0];
    if (ord.size() == 2) return qv[0] > qv[1] ? ord[0] : ord[1];  // reads 0 and 1 of the group (OneUmiCluster.java:L53)
    // total squared distance of every member; the first minimum in iteration order wins
    std::vector<long> tot(ord.size(), 0);
    parallel_for(ord.size(), 64, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const int s = ord[i];
            long t = 0;
            for (int w : ord)
                if (w != s) t += (long)(int)std::pow((double)D.ed(s, w), 2.0);
            tot[i] = t;
        }
    });
    long best = -1;
    int center = ord[0];
    for (size_t i = 0; i < ord.size(); i++)
        if (best < 0 || tot[i] < best) {
            best = tot[i];
            center = ord[i];
        }
    return center;
}

// LingPipe complete link + cut: partition of 0..k-1 (distance d(a,b) = ED(nb[a], nb[b])) at max_distance
std::vector<std::vector<int>> complete_link(const Dist &D, const std::vector<int> &nb, double max_distance) {
    const int k = (int)nb.size();
    struct Node {
        int parent = -1, left = -1, right = -1;
        double score = 0.0;
    };
    struct Pair {
        int a, b;
        double score;
    };
    std::vector<Node> nodes(k);
    std::vector<Pair> pairs;
    // queue order: score ascending, then id DESCENDING (BoundedPriorityQueue$EntryComparator)
    auto cmp = [&](int x, int y) { return pairs[x].score != pairs[y].score ? pairs[x].score < pairs[y].score : x > y; };
    std::set<int, decltype(cmp)> queue(cmp);
    std::vector<std::vector<int>> index(k);  // node -> its pairs, creation order
    auto offer = [&](int a, int b, double s) {
        pairs.push_back({a, b, s});
        const int id = (int)pairs.size() - 1;
        queue.insert(id);
        if ((int)index.size() <= std::max(a, b)) index.resize(std::max(a, b) + 1);
        index[a].push_back(id);
        index[b].push_back(id);
    };
    for (int i = 0; i < k; i++)
        for (int j = i + 1; j < k; j++) offer(i, j, (double)D.ed(nb[i], nb[j]));
    auto root_of = [&](int x) {
        while (nodes[x].parent >= 0) x = nodes[x].parent;
        return x;
    };
    std::vector<char> dead;
    int root = 0;
    while (!queue.empty()) {
        const int nxt = *queue.begin();
        queue.erase(queue.begin());
        dead.resize(pairs.size(), 0);
        dead[nxt] = 1;
        const int d1 = root_of(pairs[nxt].a), d2 = root_of(pairs[nxt].b);
        Node link;
        link.left = d1;
        link.right = d2;
        link.score = pairs[nxt].score;
        nodes.push_back(link);
        const int d12 = (int)nodes.size() - 1;
        nodes[d1].parent = nodes[d2].parent = d12;
        root = d12;
        index.resize(nodes.size());
        std::map<int, double> buf;
        for (int p : index[d1]) {
            if (dead[p] && p != nxt) continue;
            buf[pairs[p].a == d1 ? pairs[p].b : pairs[p].a] = pairs[p].score;
            if (!dead[p]) {
                queue.erase(p);
                dead[p] = 1;
            }
        }
        const std::vector<int> of_d2 = index[d2];
        for (int p : of_d2) {
            if (dead[p]) continue;
            queue.erase(p);
            dead[p] = 1;
            const int d3 = pairs[p].a == d2 ? pairs[p].b : pairs[p].a;
            auto it = buf.find(d3);
            if (it == buf.end()) continue;
            const double s = std::max(it->second, pairs[p].score);
            offer(d12, d3, s);
            dead.resize(pairs.size(), 0);
        }
    }
    std::vector<std::vector<int>> out;
    std::vector<int> stack{root};
    while (!stack.empty()) {
        const int cur = stack.back();
        stack.pop_back();
        if (nodes[cur].score <= max_distance) {
            std::vector<int> mem, st{cur};
            while (!st.empty()) {
                const int x = st.back();
                st.pop_back();
                if (x < k)
                    mem.push_back(x);
                else {
                    st.push_back(nodes[x].left);
                    st.push_back(nodes[x].right);
                }
            }
            out.push_back(mem);
        } else {
            stack.push_back(nodes[cur].left);
            stack.push_back(nodes[cur].right);
        }
    }
    return out;
}

std::vector<std::vector<int>> single_link(const Dist &D, const std::vector<int> &nb, double max_distance) {
    const int k = (int)nb.size();
    std::vector<int> parent(k);
    for (int i = 0; i < k; i++) parent[i] = i;
    auto find = [&](int x) {
        while (parent[x] != x) x = parent[x] = parent[parent[x]];
        return x;
    };
    for (int i = 0; i < k; i++)
        for (int j = i + 1; j < k; j++)
            if ((double)D.ed(nb[i], nb[j]) <= max_distance) parent[find(i)] = find(j);
    std::map<int, std::vector<int>> g;
    for (int i = 0; i < k; i++) g[find(i)].push_back(i);
    std::vector<std::vector<int>> out;
    for (auto &kv : g) out.push_back(kv.second);
    return out;
}

struct Cluster {
    std::vector<int> members;  // ascending
    int center = -1;
};

void tag_members(const Dist &D, const Cluster &c, const std::vector<int> &who, size_t n_clusters,
                 const std::vector<char> &skipped, smi_umi_assignment *out) {
    long sum = 0;
    int cnt = 0;
    for (int v : c.members)
        if (v != c.center) {
            sum += D.pos1(c.center, v) - 1;
            cnt++;
        }
    const int offset = (int)std::floor((double)sum / (double)cnt + 0.5);  // (int) Math.round(double)
    std::vector<char> inside(D.n, 0);
    for (int v : c.members) inside[v] = 1;
    parallel_for(who.size(), 64, [&](size_t lo, size_t hi) {  // (every member writes its own record)
        for (size_t i = lo; i < hi; i++) {
            const int idx = who[i];
            if (skipped[idx]) continue;
            int sec = -1;
            if (n_clusters > 1)
                for (int m = 0; m < D.n; m++)
                    if (!inside[m] && (sec < 0 || D.ed(idx, m) < sec)) sec = D.ed(idx, m);
            out[idx].center = c.center;
            out[idx].offset = (int8_t)offset;
            out[idx].ed = (int8_t)D.ed(c.center, idx);
            out[idx].ed_second = (int8_t)sec;
            out[idx].pos2 = (int8_t)D.pos2(c.center, idx);
        }
    });
}

// clusterLocal: owner key of every index that has a neighbour, -1 otherwise
std::vector<int> cluster_local(const Dist &D, const std::vector<int> &indices, int ed) {
    std::vector<int> owner(D.n, -1), count(D.n, 0), keys;
    parallel_for(indices.size(), 64, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const int a = indices[i];
            int c = 0;
            for (int b : indices) c += D.ed(a, b) <= ed;
            count[a] = c;
        }
    });
    for (int a : indices)
        if (count[a] > 1) keys.push_back(a);
    const std::vector<int> ord = fastutil_order(keys);
    parallel_for(keys.size(), 64, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const int c = keys[i];
            int best = -1;
            for (int a : ord)
                if (D.ed(a, c) <= ed && (best < 0 || count[a] > count[best])) best = a;
            owner[c] = best;
        }
    });
    return owner;
}

void cluster_one(const uint8_t *mat, int n, const float *qv, const smi_umi_cluster_config &cfg, smi_umi_assignment *out,
                 uint8_t *skipped_out) {
    const Dist D{mat, n};
    std::vector<char> skipped(std::max(n, 1), 0);
    for (int i = 0; i < n; i++) out[i] = smi_umi_assignment{-1, 0, -1, -1, 0};
    const int ced = cfg.complete_link_ed;
    std::vector<Cluster> kept;
    if (n > 1 && n <= cfg.own_clusterer_above) {
        std::vector<int> nb;
        for (int i = 0; i < n; i++) {
            bool any = false;
            for (int j = 0; j < n && !any; j++) any = i != j && D.ed(i, j) <= ced;
            if (any) nb.push_back(i);
        }
        if (nb.size() > 1) {
            auto parts = (int)nb.size() > cfg.single_link_switch ? single_link(D, nb, (double)cfg.single_link_ed)
                                                                 : complete_link(D, nb, (double)ced);
            size_t mx = 0;
            for (auto &p : parts)
                if (p.size() > 1) mx = std::max(mx, p.size());
            for (auto &p : parts) {
                if (p.size() <= 1) continue;
                Cluster c;
                for (int a : p) c.members.push_back(nb[a]);
                std::sort(c.members.begin(), c.members.end());
                if (c.members.size() * (size_t)cfg.fold_depth_below_max > mx)
                    kept.push_back(c);
                else
                    for (int v : c.members) skipped[v] = 1;
            }
            for (auto &c : kept) c.center = choose_center(D, c.members, qv);
            for (auto &c : kept) tag_members(D, c, fastutil_order(c.members), kept.size(), skipped, out);
        }
    } else if (n > 1) {
        std::vector<int> all(n);
        for (int i = 0; i < n; i++) all[i] = i;
        auto group_by_owner = [&](const std::vector<int> &owner, const std::vector<int> &indices) {
            std::map<int, size_t> slot;  // owner -> cluster, clusters listed by their smallest member
            std::vector<Cluster> cl;
            for (int i : indices) {
                if (owner[i] < 0) continue;
                auto it = slot.find(owner[i]);
                if (it == slot.end()) {
                    slot[owner[i]] = cl.size();
                    cl.emplace_back();
                    it = slot.find(owner[i]);
                }
                cl[it->second].members.push_back(i);
            }
            return cl;
        };
        std::vector<Cluster> first = group_by_owner(cluster_local(D, all, ced), all);
        size_t mx = 0;
        for (auto &c : first) mx = std::max(mx, c.members.size());
        std::vector<char> clustered(n, 0);
        for (auto &c : first) {
            if (c.members.size() * (size_t)cfg.fold_depth_below_max > mx) {
                c.center = choose_center(D, c.members, qv);
                for (int v : c.members) clustered[v] = 1;
                kept.push_back(c);
            } else
                for (int v : c.members) skipped[v] = 1;
        }
        std::vector<int> unclustered;
        for (int i = 0; i < n; i++)
            if (!clustered[i]) unclustered.push_back(i);
        size_t n_removed = 0;
        for (auto &c : kept) {  // removeOffCenter
            std::vector<int> stay;
            for (int s : c.members)
                if (D.ed(s, c.center) > ced) {
                    unclustered.push_back(s);
                    n_removed++;
                } else
                    stay.push_back(s);
            if (stay.size() != c.members.size()) {
                c.members = stay;
                c.center = choose_center(D, c.members, qv);
            }
        }
        if (n_removed > 0) {
            std::sort(unclustered.begin(), unclustered.end());
            for (auto &c : group_by_owner(cluster_local(D, unclustered, ced), unclustered))
                if (c.members.size() > 1) {
                    c.center = choose_center(D, c.members, qv);
                    kept.push_back(c);
                }
        }
        parallel_for(kept.size(), 1, [&](size_t lo, size_t hi) {  // (clusters are disjoint: every read's record has one writer)
            for (size_t k = lo; k < hi; k++) {
                const Cluster &c = kept[k];
                if (c.members.size() <= 1) continue;
                std::vector<int> filt;
                for (int s : fastutil_order(c.members))
                    if (D.ed(s, c.center) <= ced) filt.push_back(s);
                if (filt.size() > 1) tag_members(D, c, filt, kept.size(), skipped, out);
            }
        });
    }
    if (skipped_out)
        for (int i = 0; i < n; i++) skipped_out[i] = (uint8_t)skipped[i];
}

}  // namespace
}  // namespace smi

using namespace smi;

extern "C" int smi_umi_cluster_default_config(smi_umi_cluster_config *cfg) {
    if (!cfg) {
        set_error("smi_umi_cluster_default_config: null argument");
        return SMI_ERR_INVALID;
    }
    cfg->complete_link_ed = 2;        // Jar/config.xml:270
    cfg->single_link_ed = 1;          // :272
    cfg->single_link_switch = 3000;   // :278
    cfg->fold_depth_below_max = 50;   // UMIparameters.java:L118
    cfg->own_clusterer_above = 100;   // UmiClustering.java:L52
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// ClusterOne_MyClustering for a group above 100 reads with the matrix left in HBM (round 4).
//
// What is n^2 in it runs on the device over the matrix K-UMI left there: the neighbour counts, the owner arg-max, the squared-distance
// sums of the centre choice, the least distance to a read outside the cluster (tag U2).  What is O(n) and order-bound stays on the
// host between the launches: the fastutil iteration orders (a simulation of the table's insert / rehash history, sequential by
// nature), the grouping by owner, the fold-depth rule, the ejection of members off their centre.  A few thousand bytes cross the link
// per step instead of the n^2-byte matrix (64 MB for 8,000 reads) -- and the host's sixteen threads no longer walk it.
// Every rule is the one of cluster_one() above; the two are compared on the same groups (tests/test_umi_stage_gpu.py).
// ---------------------------------------------------------------------------------------------------------------
#define SMI_OWN_RC(call)              \
    do {                              \
        const int rc__ = (call);      \
        if (rc__ != SMI_OK) return rc__; \
    } while (0)

namespace smi {
namespace {

// count[i] = how many b of idx[0 .. n_idx) have ed(idx[i], b) <= ced (the read itself included: ed = 0).  One wave per row.  `init` (first
// call of a group): the row's assignment record starts as "not clustered", its cluster number as -1, its tag flag as 0.
__global__ __launch_bounds__(256) void k_umi_own_count(const uint8_t *__restrict__ m, int n, int ld, const int *__restrict__ idx, int n_idx, int ced,
                                                       int *__restrict__ count, smi_umi_assignment *__restrict__ init, int *__restrict__ init_cid,
                                                       uint8_t *__restrict__ init_flag) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6), n_waves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
    for (int i = wave; i < n_idx; i += n_waves) {
        const uint8_t *row = m + (size_t)(idx ? idx[i] : i) * ld;
        int c = 0;
        if (!idx) {  // every read of the group: the row as it lies, eight cells per load (a byte per lane and round trip is latency, not bandwidth)
            for (int k = 8 * lane; k < n_idx; k += 512) {
                if (k + 8 <= n_idx) {
                    uint64_t v;
                    __builtin_memcpy(&v, row + k, 8);
#pragma unroll
                    for (int b = 0; b < 8; b++) c += (int)((v >> (8 * b)) & 15u) <= ced ? 1 : 0;
                } else
                    for (int b = k; b < n_idx; b++) c += (row[b] & 15) <= ced ? 1 : 0;
            }
        } else {
#pragma unroll 4
            for (int k = lane; k < n_idx; k += 64) c += (row[idx[k]] & 15) <= ced ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) {
            count[i] = c;
            if (init) {
                init[i] = smi_umi_assignment{-1, 0, -1, -1, 0};
                init_cid[i] = -1;
                init_flag[i] = 0;
            }
        }
    }
}

// owner[i] for key ord[i] (keys in fastutil iteration order): the first a in that order with ed(a, key) <= ced and the largest count
// (Stream.max keeps the first of equals).  The candidates are looked at in COLUMN order -- the key's row as it lies, eight cells per load --
// with rc[b] = {position of read b in the iteration order or -1 when b is no key, its count}: the arg-max over (count, -position) does not
// care in which order it sees them.  (Walking the positions and gathering row[ord[p]] was 0.58 ms for 8,000 keys: a byte per lane from a
// random place of the row.)  One wave per key; ed is symmetric.
__global__ __launch_bounds__(256) void k_umi_own_owner(const uint8_t *__restrict__ m, int n, int ld, const int *__restrict__ ord, const int2 *__restrict__ rc,
                                                       int n_keys, int ced, int *__restrict__ owner) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6), n_waves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
    for (int i = wave; i < n_keys; i += n_waves) {
        const uint8_t *row = m + (size_t)ord[i] * ld;
        long long best = -1;  // count << 32 | (0x7FFFFFFF - position): larger = better count, then earlier position
        auto look = [&](int b, int e) {
            if (e > ced) return;  // (a row has a handful of cells within ced: rc is read for those only)
            const int2 x = rc[b];
            if (x.x >= 0) {
                const long long v = ((long long)x.y << 32) | (long long)(0x7FFFFFFF - x.x);
                best = v > best ? v : best;
            }
        };
        for (int k = 8 * lane; k < n; k += 512) {
            if (k + 8 <= n) {
                uint64_t v;
                __builtin_memcpy(&v, row + k, 8);
#pragma unroll
                for (int b = 0; b < 8; b++) look(k + b, (int)((v >> (8 * b)) & 15u));
            } else
                for (int b = k; b < n; b++) look(b, row[b] & 15);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long y = __shfl_xor(best, o);
            best = y > best ? y : best;
        }
        if (lane == 0) owner[i] = best < 0 ? -1 : ord[0x7FFFFFFF - (int)(best & 0x7FFFFFFFLL)];
    }
}

// tot[i] = sum over the members w != s of ed(s, w)^2 for member i of its cluster (members listed cluster after cluster; c_off the offsets)
__global__ __launch_bounds__(256) void k_umi_own_sums(const uint8_t *__restrict__ m, int n, int ld, const int *__restrict__ mem, const int *__restrict__ c_of,
                                                      const int *__restrict__ c_off, int n_mem, long long *__restrict__ tot) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6), n_waves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
    for (int i = wave; i < n_mem; i += n_waves) {
        const int c = c_of[i], s = mem[i];
        const uint8_t *row = m + (size_t)s * ld;
        long long t = 0;
#pragma unroll 4
        for (int k = c_off[c] + lane; k < c_off[c + 1]; k += 64) {
            const int w = mem[k];
            const int e = row[w] & 15;
            if (w != s) t += (long long)(e * e);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) tot[i] = t;
    }
}

// setClusterCenterNotPreGrouped (OneUmiCluster.java:L49-65) for the listed clusters (members in fastutil iteration order): one member: itself;
// two: by the quality of reads 0 and 1 of the GROUP (L53); else the first minimum of tot in list order.  center[gid[c]] is written: the
// device keeps the centre of every cluster of the call under its number.  One wave per cluster.
__global__ __launch_bounds__(256) void k_umi_own_pick(const long long *__restrict__ tot, const int *__restrict__ mem, const int *__restrict__ c_off,
                                                      const int *__restrict__ gid, int n_cl, const float *__restrict__ qv, int n,
                                                      int *__restrict__ center) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6), n_waves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
    for (int c = wave; c < n_cl; c += n_waves) {
        const int a = c_off[c], b = c_off[c + 1], sz = b - a;
        if (sz <= 0) continue;
        int ctr;
        if (sz == 1)
            ctr = mem[a];
        else if (sz == 2)
            ctr = (n >= 2 && qv[0] > qv[1]) ? mem[a] : mem[a + 1];
        else {
            unsigned long long best = ~0ull;  // tot << 32 | position: smallest total, then the earliest position
            for (int k = a + lane; k < b; k += 64) {
                const unsigned long long v = ((unsigned long long)tot[k] << 32) | (unsigned)(k - a);
                best = v < best ? v : best;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long y = __shfl_xor(best, o);
                best = y < best ? y : best;
            }
            ctr = mem[a + (int)(best & 0xFFFFFFFFull)];
        }
        if (lane == 0) center[gid[c]] = ctr;
    }
}

// cell[i] = matrix byte (centre of member i's cluster, member i)
__global__ void k_umi_own_cells(const uint8_t *__restrict__ m, int n, int ld, const int *__restrict__ mem, const int *__restrict__ c_of, const int *__restrict__ gid,
                                const int *__restrict__ center, int n_mem, uint8_t *__restrict__ cell) {
    const int i = (int)(blockIdx.x * (size_t)blockDim.x + threadIdx.x);
    if (i < n_mem) cell[i] = m[(size_t)center[gid[c_of[i]]] * ld + mem[i]];
}

// What the host clusterer's last loop decides per cluster with more than one member (cluster_one / tag_members): the members within ced of
// the (final) centre are tagged if there is more than one of them; the offset is the rounded mean of pos1 - 1 over ALL members but the centre.
// A tagged cluster's members carry its number in cid (everybody else keeps -1: "outside" for every read), flag marks the reads that get tags.
// One wave per cluster.
__global__ __launch_bounds__(256) void k_umi_own_decide(const uint8_t *__restrict__ m, int n, int ld, const int *__restrict__ mem, const int *__restrict__ c_off,
                                                        const int *__restrict__ gid, int n_cl, const int *__restrict__ center, int ced,
                                                        const uint8_t *__restrict__ skipped, int *__restrict__ cid, int *__restrict__ offset,
                                                        uint8_t *__restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6), n_waves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
    for (int c = wave; c < n_cl; c += n_waves) {
        const int a = c_off[c], b = c_off[c + 1], g = gid[c], ctr = center[g];
        const uint8_t *row = m + (size_t)ctr * ld;
        int sum = 0, cnt = 0, filt = 0;
        for (int k = a + lane; k < b; k += 64) {
            const int v = mem[k];
            const int cell = row[v];
            filt += (cell & 15) <= ced ? 1 : 0;
            if (v != ctr) {
                sum += ((cell >> 4) & 3) - 1;
                cnt++;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_xor(sum, o);
            cnt += __shfl_xor(cnt, o);
            filt += __shfl_xor(filt, o);
        }
        if (filt <= 1) continue;  // (wave-uniform)
        if (lane == 0) offset[g] = (int)floor((double)sum / (double)cnt + 0.5);  // (int) Math.round(double)
        for (int k = a + lane; k < b; k += 64) {
            const int v = mem[k];
            cid[v] = g;
            flag[v] = ((row[v] & 15) <= ced && !skipped[v]) ? 1 : 0;
        }
    }
}

// tags of the flagged reads (ClusterOneBase.setSamflagsAndStatsForClustered): centre, offset, ed / pos2 to the centre, and the least distance to
// a read outside the read's cluster (cid[m] != its own; -1 when there is one cluster only).  One wave per read.
__global__ __launch_bounds__(256) void k_umi_own_tags(const uint8_t *__restrict__ m, int n, int ld, const uint8_t *__restrict__ flag, const int *__restrict__ cid,
                                                      const int *__restrict__ center, const int *__restrict__ offset, int n_clusters,
                                                      smi_umi_assignment *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6), n_waves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
    for (int idx = wave; idx < n; idx += n_waves) {
        if (!flag[idx]) continue;
        const int c = cid[idx];
        const uint8_t *row = m + (size_t)idx * ld;
        int sec = 127;
        if (n_clusters > 1)
            for (int k = 8 * lane; k < n; k += 512) {
                if (k + 8 <= n) {
                    uint64_t v;
                    __builtin_memcpy(&v, row + k, 8);
                    int id[8];
                    __builtin_memcpy(id, cid + k, 32);
#pragma unroll
                    for (int b = 0; b < 8; b++)
                        if (id[b] != c) sec = min(sec, (int)((v >> (8 * b)) & 15u));
                } else
                    for (int b = k; b < n; b++)
                        if (cid[b] != c) sec = min(sec, (int)(row[b] & 15));
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sec = min(sec, __shfl_xor(sec, o));
        if (lane == 0) {
            const uint8_t cell = m[(size_t)center[c] * ld + idx];
            smi_umi_assignment a;
            a.center = center[c];
            a.offset = (int8_t)offset[c];
            a.ed = (int8_t)(cell & 15);
            a.ed_second = (int8_t)(n_clusters > 1 && sec != 127 ? sec : -1);
            a.pos2 = (int8_t)((cell >> 6) & 3);
            out[idx] = a;
        }
    }
}

// bump allocator over one buffer (device scratch / pinned host memory): every step of a call takes fresh room, so nothing that a queued
// copy or kernel still reads is written again
struct Arena {
    char *p = nullptr;
    size_t cap = 0, at = 0;
    char *take(size_t bytes) {
        at = (at + 63) & ~(size_t)63;
        char *r = p + at;
        at += bytes;
        return at <= cap ? r : nullptr;
    }
};

// the arrays one step sends up, side by side in pinned memory and at the same offsets on the device: one copy per step
struct UpBlock {
    char *h = nullptr, *d = nullptr;
    size_t bytes = 0;
    template <class T>
    T *host(size_t off) const { return reinterpret_cast<T *>(h + off); }
    template <class T>
    T *dev(size_t off) const { return reinterpret_cast<T *>(d + off); }
};

}  // namespace

int umi_cluster_own_device(smi_ctx *ctx, const uint8_t *d_mat, int n, const float *d_qv, const smi_umi_cluster_config &cfg, smi_umi_assignment *d_out,
                           uint8_t *d_skipped, hipStream_t s, int ld) {
    if (ld <= 0) ld = n;  // row stride of the matrix (the chunk worker's matrices have rows padded to whole lines)
    const int ced = cfg.complete_link_ed;
    // Room: every array of a step has at most n + 2 entries of at most 8 bytes, and a call takes fewer than 48 of them on either side.
    const size_t room = 48 * ((size_t)n + 64) * 8;
    if (ctx->umi_own_bytes < room) {
        if (ctx->umi_own) (void)hipFree(ctx->umi_own);
        ctx->umi_own = nullptr;
        ctx->umi_own_bytes = 0;
        SMI_HIP(hipMalloc(&ctx->umi_own, room + room / 4));
        ctx->umi_own_bytes = room + room / 4;
    }
    SMI_OWN_RC(ensure_host_buf(ctx, smi_ctx::HB_OWN, room));
    Arena D{static_cast<char *>(ctx->umi_own), room, 0}, H{static_cast<char *>(ctx->host_buf[smi_ctx::HB_OWN]), room, 0};
    bool short_of_room = false;
    auto up = [](size_t x) { return (x + 63) & ~(size_t)63; };
    auto block = [&](size_t bytes) {
        UpBlock B;
        B.h = H.take(bytes);
        B.d = D.take(bytes);
        B.bytes = bytes;
        if (!B.h || !B.d) short_of_room = true;
        return B;
    };
    auto dev_only = [&](size_t bytes) {
        char *r = D.take(bytes);
        if (!r) short_of_room = true;
        return r;
    };
    auto host_only = [&](size_t bytes) {
        char *r = H.take(bytes);
        if (!r) short_of_room = true;
        return r;
    };
#define SMI_OWN_ROOM()                                                               \
    do {                                                                             \
        if (short_of_room) {                                                         \
            set_error("umi_cluster_own_device: scratch accounting (internal error)"); \
            return SMI_ERR_INVALID;                                                  \
        }                                                                            \
    } while (0)
    static const bool timing = getenv("SMI_AU_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "    own clusterer %-44s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - t_prev).count());
        t_prev = t;
    };
    const unsigned grid = (unsigned)std::min<size_t>(((size_t)n + 3) / 4, 256 * 16);
    std::vector<char> skipped((size_t)n, 0);
    // per cluster number (there are fewer clusters than reads): centre, offset; per read: cluster number, "gets tags"
    int *d_center = reinterpret_cast<int *>(dev_only(((size_t)n + 1) * 4)), *d_offset = reinterpret_cast<int *>(dev_only(((size_t)n + 1) * 4));
    int *d_cid = reinterpret_cast<int *>(dev_only(((size_t)n + 8) * 4));
    uint8_t *d_flag = reinterpret_cast<uint8_t *>(dev_only((size_t)n + 8));
    SMI_OWN_ROOM();

    // clusterLocal (L175-219) over `indices` (ascending): owner key of every index that is a key, -1 otherwise
    bool first_call = true;
    std::vector<int> keys;
    auto cluster_local_dev = [&](const std::vector<int> &indices, std::vector<int> &owner) -> int {
        owner.assign((size_t)n, -1);
        const int k = (int)indices.size();
        if (k == 0) return SMI_OK;
        const bool identity = k == n;   // (indices are ascending and distinct: all n of them = 0 .. n-1)
        const int *d_idx = nullptr;
        if (!identity) {
            const UpBlock B = block((size_t)k * 4);
            SMI_OWN_ROOM();
            std::memcpy(B.h, indices.data(), (size_t)k * 4);
            SMI_HIP(hipMemcpyAsync(B.d, B.h, B.bytes, hipMemcpyHostToDevice, s));
            d_idx = B.dev<int>(0);
        }
        int *h_count = reinterpret_cast<int *>(host_only((size_t)k * 4)), *d_count = reinterpret_cast<int *>(dev_only((size_t)k * 4));
        SMI_OWN_ROOM();
        const bool init = first_call && identity;
        hipLaunchKernelGGL(k_umi_own_count, dim3(grid), dim3(256), 0, s, d_mat, n, ld, d_idx, k, ced, d_count, init ? d_out : (smi_umi_assignment *)nullptr,
                           init ? d_cid : (int *)nullptr, init ? d_flag : (uint8_t *)nullptr);
        SMI_HIP(hipMemcpyAsync(h_count, d_count, (size_t)k * 4, hipMemcpyDeviceToHost, s));
        SMI_HIP(hipStreamSynchronize(s));
        lap("neighbour counts");
        keys.clear();
        for (int i = 0; i < k; i++)
            if (h_count[i] > 1) keys.push_back(indices[(size_t)i]);
        if (keys.empty()) return SMI_OK;
        const std::vector<int> ord = fastutil_order(keys);
        // up: ord[n_keys] | rc[n] = {position in ord or -1, count}
        const size_t o_rc = up(ord.size() * 4);
        const UpBlock B = block(o_rc + (size_t)n * 8);
        int *h_own = reinterpret_cast<int *>(host_only(ord.size() * 4)), *d_own = reinterpret_cast<int *>(dev_only(ord.size() * 4));
        SMI_OWN_ROOM();
        int *h_ord = B.host<int>(0);
        int2 *h_rc = B.host<int2>(o_rc);
        for (int i = 0; i < n; i++) h_rc[i] = int2{-1, 0};
        for (int i = 0; i < k; i++) h_rc[(size_t)indices[(size_t)i]].y = h_count[i];
        for (size_t i = 0; i < ord.size(); i++) {
            h_ord[i] = ord[i];
            h_rc[(size_t)ord[i]].x = (int)i;
        }
        lap("keys in iteration order (host)");
        SMI_HIP(hipMemcpyAsync(B.d, B.h, B.bytes, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_umi_own_owner, dim3(grid), dim3(256), 0, s, d_mat, n, ld, B.dev<int>(0), B.dev<int2>(o_rc), (int)ord.size(), ced, d_own);
        SMI_HIP(hipMemcpyAsync(h_own, d_own, ord.size() * 4, hipMemcpyDeviceToHost, s));
        SMI_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < ord.size(); i++) owner[(size_t)ord[i]] = h_own[i];
        lap("owners");
        return SMI_OK;
    };
    std::vector<int> slot((size_t)n);
    auto group_by_owner = [&](const std::vector<int> &owner, const std::vector<int> &indices) {
        std::fill(slot.begin(), slot.end(), -1);  // owner -> cluster, clusters listed by their smallest member
        std::vector<Cluster> cl;
        for (int i : indices) {
            const int o = owner[(size_t)i];
            if (o < 0) continue;
            if (slot[(size_t)o] < 0) {
                slot[(size_t)o] = (int)cl.size();
                cl.emplace_back();
            }
            cl[(size_t)slot[(size_t)o]].members.push_back(i);
        }
        return cl;
    };
    // The clusters `which` of `cl` as flat arrays on the device (one copy): mem (members cluster after cluster, in fastutil iteration order when
    // asked), c_of (a member's place in `which`), c_off (offsets into mem), gid (the cluster's number in `cl`)
    struct Listed {
        int *d_mem = nullptr, *d_c_of = nullptr, *d_c_off = nullptr, *d_gid = nullptr;
        int *h_mem = nullptr;
        int n_mem = 0, n_cl = 0;
    };
    auto list_clusters = [&](const std::vector<Cluster> &cl, const std::vector<size_t> &which, bool iteration_order, Listed &L) -> int {
        size_t total = 0;
        for (size_t w : which) total += cl[w].members.size();
        L.n_mem = (int)total;
        L.n_cl = (int)which.size();
        if (total == 0) return SMI_OK;
        const size_t o_c_of = up(total * 4), o_c_off = o_c_of + up(total * 4), o_gid = o_c_off + up((which.size() + 1) * 4);
        const UpBlock B = block(o_gid + up((which.size() + 1) * 4));
        SMI_OWN_ROOM();
        L.h_mem = B.host<int>(0);
        int *h_c_of = B.host<int>(o_c_of), *h_c_off = B.host<int>(o_c_off), *h_gid = B.host<int>(o_gid);
        size_t at = 0;
        for (size_t w = 0; w < which.size(); w++) {
            h_c_off[w] = (int)at;
            h_gid[w] = (int)which[w];
            const std::vector<int> &mm = cl[which[w]].members;
            if (iteration_order && mm.size() > 1) {
                for (int v : fastutil_order(mm)) {
                    L.h_mem[at] = v;
                    h_c_of[at++] = (int)w;
                }
            } else
                for (int v : mm) {
                    L.h_mem[at] = v;
                    h_c_of[at++] = (int)w;
                }
        }
        h_c_off[which.size()] = (int)at;
        SMI_HIP(hipMemcpyAsync(B.d, B.h, B.bytes, hipMemcpyHostToDevice, s));
        L.d_mem = B.dev<int>(0);
        L.d_c_of = B.dev<int>(o_c_of);
        L.d_c_off = B.dev<int>(o_c_off);
        L.d_gid = B.dev<int>(o_gid);
        return SMI_OK;
    };
    // setClusterCenterNotPreGrouped for the clusters `which` of `cl`: queued, the centres stay on the device (d_center[cluster number])
    auto centers_dev = [&](const std::vector<Cluster> &cl, const std::vector<size_t> &which, Listed &L) -> int {
        SMI_OWN_RC(list_clusters(cl, which, true, L));
        if (L.n_mem == 0) return SMI_OK;
        long long *d_tot = reinterpret_cast<long long *>(dev_only(((size_t)L.n_mem + 1) * 8));
        SMI_OWN_ROOM();
        hipLaunchKernelGGL(k_umi_own_sums, dim3(grid), dim3(256), 0, s, d_mat, n, ld, L.d_mem, L.d_c_of, L.d_c_off, L.n_mem, d_tot);
        hipLaunchKernelGGL(k_umi_own_pick, dim3((unsigned)std::min<size_t>(((size_t)L.n_cl + 3) / 4, 256 * 16)), dim3(256), 0, s, d_tot, L.d_mem, L.d_c_off,
                           L.d_gid, L.n_cl, d_qv, n, d_center);
        return SMI_OK;
    };

    std::vector<int> all((size_t)n), owner;
    for (int i = 0; i < n; i++) all[(size_t)i] = i;
    SMI_OWN_RC(cluster_local_dev(all, owner));
    first_call = false;
    std::vector<Cluster> first = group_by_owner(owner, all), kept;
    size_t mx = 0;
    for (auto &c : first) mx = std::max(mx, c.members.size());
    std::vector<char> clustered((size_t)n, 0);
    for (auto &c : first) {
        if (c.members.size() * (size_t)cfg.fold_depth_below_max > mx) {
            for (int v : c.members) clustered[(size_t)v] = 1;
            kept.push_back(std::move(c));
        } else
            for (int v : c.members) skipped[(size_t)v] = 1;
    }
    std::vector<size_t> every(kept.size());
    for (size_t k = 0; k < kept.size(); k++) every[k] = k;
    Listed L1;
    SMI_OWN_RC(centers_dev(kept, every, L1));
    // removeOffCenter (L90-100): members farther than ced from their centre leave; a cluster that lost some gets a new centre
    uint8_t *h_cell = nullptr;
    if (L1.n_mem) {
        h_cell = reinterpret_cast<uint8_t *>(host_only((size_t)L1.n_mem));
        uint8_t *d_cell = reinterpret_cast<uint8_t *>(dev_only((size_t)L1.n_mem));
        SMI_OWN_ROOM();
        hipLaunchKernelGGL(k_umi_own_cells, dim3((unsigned)((L1.n_mem + 255) / 256)), dim3(256), 0, s, d_mat, n, ld, L1.d_mem, L1.d_c_of, L1.d_gid, d_center, L1.n_mem,
                           d_cell);
        SMI_HIP(hipMemcpyAsync(h_cell, d_cell, (size_t)L1.n_mem, hipMemcpyDeviceToHost, s));
    }
    std::vector<int> unclustered;  // (while the device works)
    for (int i = 0; i < n; i++)
        if (!clustered[(size_t)i]) unclustered.push_back(i);
    lap("clusters, fold depth, member orders (host)");
    size_t n_removed = 0;
    std::vector<size_t> changed;
    if (L1.n_mem) {
        SMI_HIP(hipStreamSynchronize(s));
        lap("centres, distances to them");
        std::vector<uint8_t> far((size_t)n, 0);
        bool any_far = false;
        for (int i = 0; i < L1.n_mem; i++)
            if ((h_cell[i] & 15) > ced) {
                far[(size_t)L1.h_mem[i]] = 1;
                any_far = true;
            }
        if (any_far)
            for (size_t k = 0; k < kept.size(); k++) {
                std::vector<int> &mm = kept[k].members;
                size_t w = 0;
                for (size_t i = 0; i < mm.size(); i++) {
                    if (far[(size_t)mm[i]]) {
                        unclustered.push_back(mm[i]);
                        n_removed++;
                    } else
                        mm[w++] = mm[i];
                }
                if (w != mm.size()) {
                    mm.resize(w);
                    changed.push_back(k);
                }
            }
    }
    Listed L2, L3, L4;
    if (!changed.empty()) SMI_OWN_RC(centers_dev(kept, changed, L2));
    if (n_removed > 0) {  // L102-112: the ejected and the unclustered reads once more
        std::sort(unclustered.begin(), unclustered.end());
        lap("members off their centre (host)");
        SMI_OWN_RC(cluster_local_dev(unclustered, owner));
        std::vector<Cluster> more = group_by_owner(owner, unclustered);
        std::vector<size_t> fresh;
        for (auto &c : more)
            if (c.members.size() > 1) {
                fresh.push_back(kept.size());
                kept.push_back(std::move(c));
            }
        if (!fresh.empty()) SMI_OWN_RC(centers_dev(kept, fresh, L3));
    }
    // tags: the members within ced of the (final) centre, if more than one is left (L131-160) -- decided and written on the device
    std::vector<size_t> tagged;
    for (size_t k = 0; k < kept.size(); k++)
        if (kept[k].members.size() > 1) tagged.push_back(k);
    uint8_t *d_sk = d_skipped;
    {
        const UpBlock B = block((size_t)n);
        SMI_OWN_ROOM();
        for (int i = 0; i < n; i++) B.h[i] = skipped[(size_t)i];
        if (d_sk)
            SMI_HIP(hipMemcpyAsync(d_sk, B.h, (size_t)n, hipMemcpyHostToDevice, s));
        else {
            SMI_HIP(hipMemcpyAsync(B.d, B.h, (size_t)n, hipMemcpyHostToDevice, s));
            d_sk = B.dev<uint8_t>(0);
        }
    }
    if (!tagged.empty()) {
        SMI_OWN_RC(list_clusters(kept, tagged, false, L4));
        hipLaunchKernelGGL(k_umi_own_decide, dim3((unsigned)std::min<size_t>(((size_t)L4.n_cl + 3) / 4, 256 * 16)), dim3(256), 0, s, d_mat, n, ld, L4.d_mem, L4.d_c_off,
                           L4.d_gid, L4.n_cl, d_center, ced, d_sk, d_cid, d_offset, d_flag);
        hipLaunchKernelGGL(k_umi_own_tags, dim3(grid), dim3(256), 0, s, d_mat, n, ld, d_flag, d_cid, d_center, d_offset, (int)kept.size(), d_out);
    }
    lap("final lists (host)");
    SMI_HIP(hipStreamSynchronize(s));  // (the pinned arrays of this call are free again)
    SMI_HIP(hipGetLastError());
    lap("tags");
    return SMI_OK;
#undef SMI_OWN_ROOM
}

}  // namespace smi

extern "C" int smi_umi_cluster_groups(const uint8_t *dist, const uint64_t *mat_off, const uint32_t *group_off,
                                      uint32_t n_groups, const float *mean_qv, const smi_umi_cluster_config *cfg,
                                      smi_umi_assignment *out, uint8_t *skipped, int n_threads) {
    if (!cfg || (n_groups && (!dist || !mat_off || !group_off || !mean_qv || !out)) || cfg->complete_link_ed < 0 ||
        cfg->fold_depth_below_max <= 0) {
        set_error("smi_umi_cluster_groups: bad argument");
        return SMI_ERR_INVALID;
    }
    const int nt_all = std::max(1, std::min(n_threads, 256));
    const int nt = (int)std::min<uint32_t>((uint32_t)nt_all, std::max<uint32_t>(n_groups, 1u));  // one thread per group at most ...
    const int inner = std::max(1, nt_all / nt);                                                    // ... the rest work inside the groups
    auto work = [&](int t) {
        g_inner_threads = inner;
        for (uint32_t g = (uint32_t)t; g < n_groups; g += (uint32_t)nt) {
            const uint32_t a = group_off[g], n = group_off[g + 1] - a;
            cluster_one(dist + mat_off[g], (int)n, mean_qv + a, *cfg, out + a, skipped ? skipped + a : nullptr);
        }
    };
    if (nt == 1)
        work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Genomic-region grouping of `assignumis` (which reads may share a UMI group):
//   ReadGrouper.groupSams / doClusteringOneStrand   FJ!umifinder/bamreaders/ReadGrouper.java:L82-260
//   ReadGrouper$Cluster, $ClusterList.refineClusters  (same file) L455-785
//   NanoporeRead$ReadScanData.getReferencePositionAtReadPosition  FJ!umifinder/reads/nanopore/NanoporeRead$ReadScanData.java:L133-153
// ---------------------------------------------------------------------------------------------------------------
namespace smi {
namespace {

// One strand of a chunk, sorted by position (stable in BAM order): the members of a region are indices into these arrays.
struct Strand {
    std::vector<int> pos, read, index;  // clustering position, record number, rank among the chunk's reads that have a position
    std::vector<int64_t> psum;          // psum[k] = pos[0] + ... + pos[k - 1]
};

// A region's members are either a contiguous piece [lo, hi) of its strand (what a chain is when it is built, and what most regions still are
// when everything is over: nothing per member is stored, the centre comes from the prefix sums and the two "is anyone off-centre" questions
// look at the two ends) or, once a region has been split, merged or built across a gap, an explicit list in insertion order.
struct Region {
    const Strand *st = nullptr;
    int lo = 0, hi = 0;
    bool is_list = false;
    std::vector<int> list;
    bool cached = false;
    int center_value = 0;
    size_t size() const { return is_list ? list.size() : (size_t)(hi - lo); }
    bool empty() const { return size() == 0; }
    void materialize() {
        if (is_list) return;
        list.resize((size_t)(hi - lo));
        for (int k = lo; k < hi; k++) list[(size_t)(k - lo)] = k;
        is_list = true;
    }
    template <class F>
    void for_each(F &&f) const {
        if (is_list)
            for (int k : list) f(k);
        else
            for (int k = lo; k < hi; k++) f(k);
    }
    int center() {  // Math.round((float) mean) in a cached field: cleared by add / addAll / removeAll, NOT by removeOffCenter (see split_off)
        if (!cached && !empty()) {
            double s = 0;  // (the sum of the positions is an integer below 2^53: exact in a double in any order)
            if (is_list)
                for (int k : list) s += st->pos[(size_t)k];
            else
                s = (double)(st->psum[(size_t)hi] - st->psum[(size_t)lo]);
            center_value = (int)std::floor((float)(s / (double)size()) + 0.5f);
            cached = true;
        }
        return center_value;
    }
    // Cluster.removeOffCenter (ReadGrouper$Cluster.lambda$new$5, L626-638) with side = -1 (left of centre - dist) or +1 (right of centre +
    // dist).  The reference keeps the centre in a field that a removal does not clear, and works in two passes: it COUNTS the off-centre
    // members against the cached centre (possibly the one from before an earlier removal); only when that count is positive it clears the
    // field, so the pass that picks the members to move sees the centre of the list as it is now; the members then leave and the field
    // keeps that value.  Returns true when the reference creates a new cluster -- which is empty when the two passes disagree.
    bool split_off(int side, int dist, Region &out) {
        out = Region();
        out.st = st;
        if (empty()) return false;
        const int counted_with = center();
        if (!is_list) {  // sorted piece: the off-centre members, if any, include its first (side < 0) or last (side > 0) one
            const bool any = side < 0 ? st->pos[(size_t)lo] < counted_with - dist : st->pos[(size_t)hi - 1] > counted_with + dist;
            if (!any) return false;
            materialize();
        }
        const std::vector<int> &P = st->pos;
        auto off = [&](int k, int c) { return side < 0 ? P[(size_t)k] < c - dist : P[(size_t)k] > c + dist; };
        size_t n_off = 0;
        for (int k : list) n_off += off(k, counted_with);
        if (n_off == 0) return false;
        cached = false;  // L633
        const int c = center();
        std::vector<int> stay;
        out.is_list = true;
        for (int k : list) (off(k, c) ? out.list : stay).push_back(k);
        list.swap(stay);  // `cached` stays true: the field is not cleared when the members leave
        if (list.empty()) cached = false;
        std::stable_sort(out.list.begin(), out.list.end(), [&](int x, int y) { return P[(size_t)x] < P[(size_t)y]; });
        return true;
    }
};

void drop_empty(std::vector<Region> &v) {
    v.erase(std::remove_if(v.begin(), v.end(), [](const Region &r) { return r.empty(); }), v.end());
}

void sort_by_center(std::vector<Region> &v) {
    drop_empty(v);
    for (Region &r : v) r.center();
    std::stable_sort(v.begin(), v.end(), [](const Region &a, const Region &b) { return a.center_value < b.center_value; });
}

void refine_regions(std::vector<Region> &v, int dist) {
    size_t from = 0, to = v.size();
    while (from < to) {
        for (size_t i = from; i < to; i++) {
            Region out;
            if (v[i].split_off(-1, dist, out)) v.push_back(out);
            if (v[i].split_off(+1, dist, out)) v.push_back(out);
        }
        from = to;
        to = v.size();
    }
    sort_by_center(v);
    for (bool again = true; again;) {
        again = false;
        for (size_t i = 0; i + 1 < v.size(); i++) {
            if (v[i].empty()) continue;
            Region &left = v[i], &right = v[i + 1];
            if (right.center() - left.center() >= 2 * dist) continue;
            const bool left_bigger = left.size() > right.size();
            Region &from_r = left_bigger ? right : left, &to_r = left_bigger ? left : right;
            const int tc = to_r.center_value;
            const std::vector<int> &P = from_r.st->pos;
            if (!from_r.is_list) {  // sorted piece: does anyone lie within dist of the other centre at all?
                const auto b = P.begin() + from_r.lo, e = P.begin() + from_r.hi;
                const auto it = std::lower_bound(b, e, tc - dist);
                if (it == e || *it > tc + dist) continue;
                from_r.materialize();
            }
            std::vector<int> stay, move;
            for (int k : from_r.list) (std::abs(P[(size_t)k] - tc) <= dist ? move : stay).push_back(k);
            if (move.empty()) continue;
            again = true;
            to_r.materialize();
            to_r.list.insert(to_r.list.end(), move.begin(), move.end());
            from_r.list.swap(stay);
            to_r.cached = from_r.cached = false;
        }
        drop_empty(v);
    }
    v.erase(std::remove_if(v.begin(), v.end(), [](const Region &r) { return r.size() <= 1; }), v.end());
}

std::vector<Region> chain_strand(const Strand &s, int dist) {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<Region> out;
    const size_t n = s.pos.size();
    if (n <= 1) return out;
    Region cur;
    cur.st = &s;
    auto add = [&](int i) {
        if (cur.is_list)
            cur.list.push_back(i);
        else if (cur.hi == cur.lo) {
            cur.lo = i;
            cur.hi = i + 1;
        } else if (i == cur.hi)
            cur.hi++;
        else {  // a chain of <= 2 reads that went on behind a gap: the read at the gap is not a member
            cur.materialize();
            cur.list.push_back(i);
        }
    };
    if (s.pos[1] - s.pos[0] < dist) add(0);
    for (size_t i = 1; i < n; i++) {
        if (s.pos[i] - s.pos[i - 1] < dist)
            add((int)i);
        else if (cur.size() > 2) {  // a chain of <= 2 reads is not closed by a gap, it keeps growing (L247)
            out.push_back(cur);
            cur = Region();
            cur.st = &s;
        }
    }
    if (cur.size() > 2) out.push_back(cur);
    const bool timing = std::getenv("SMI_RG_TIMING") != nullptr;
    const auto t1 = std::chrono::steady_clock::now();
    const size_t n_chains = out.size();
    refine_regions(out, dist);
    if (timing)
        fprintf(stderr, "  strand of %zu reads: chains %.3f ms (%zu), refinement %.3f ms (%zu regions)\n", n,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), n_chains,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(), out.size());
    return out;
}

// everything behind the sort: chains and their refinement per strand, the regions of both strands in centre order, the cut of the chunk
template <class Lap>
int region_group_tail(Strand *strand, int32_t n, size_t n_pos, int most_right, int32_t max_dist, int keep_data_end, int32_t *region, int32_t *n_done,
                      Lap &&rg_lap) {
    // the two strands are independent (doClusteringOneStrand per strand, L113-121); a second thread only pays on very large chunks (a chunk
    // of 120 k reads is 0.2 ms of work per strand once the regions are pieces of the sorted strand: less than starting a thread costs)
    std::vector<Region> all, rv;
    if (n_pos > 2000000) {
        std::thread other([&] { rv = chain_strand(strand[1], max_dist); });
        all = chain_strand(strand[0], max_dist);
        other.join();
    } else {
        all = chain_strand(strand[0], max_dist);
        rv = chain_strand(strand[1], max_dist);
    }
    rg_lap("strands");
    all.insert(all.end(), rv.begin(), rv.end());
    sort_by_center(all);
    rg_lap("merge");
    int last_index = n - 1;
    if (keep_data_end && !all.empty() && n_pos) {  // L171-184
        while (!all.empty() && all.back().center() > most_right - 3 * max_dist) all.pop_back();
        if (!all.empty()) {
            last_index = 0;
            const Region &last = all.back();
            last.for_each([&](int k) { last_index = std::max(last_index, last.st->index[(size_t)k]); });
            last_index = std::max(last_index, n / 3);
        }
    }
    for (size_t k = 0; k < all.size(); k++) {
        const Region &r = all[k];
        r.for_each([&](int m) { region[r.st->read[(size_t)m]] = (int32_t)k; });
    }
    *n_done = last_index + 1;
    rg_lap("assign");
    return SMI_OK;
}

// pos_of(i), has_pos_of(i), rev_of(i): the chunk's records wherever the caller keeps them
template <class PosOf, class HasOf, class RevOf>
int region_group_impl(int32_t n, PosOf pos_of, HasOf has_pos_of, RevOf rev_of, int32_t max_dist, int keep_data_end, int32_t *region, int32_t *n_done) {
    const bool rg_timing = std::getenv("SMI_RG_TIMING") != nullptr;
    auto rg_t0 = std::chrono::steady_clock::now();
    auto rg_lap = [&](const char *what) {
        if (!rg_timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "smi_region_group %-10s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - rg_t0).count());
        rg_t0 = now;
    };
    // Arrays.parallelSort by position, stable (ReadGrouper.java:L128): a least-significant-digit radix sort of (biased position, rank in
    // BAM order) pairs -- three 11-bit passes, passes whose digit is the same for every record skipped -- and the two strands laid out in
    // sorted order afterwards
    std::vector<uint64_t> keys;
    keys.reserve((size_t)n);
    std::vector<int32_t> read_of;  // rank among the reads with a position -> record
    read_of.reserve((size_t)n);
    for (int i = 0; i < n; i++) {
        region[i] = -1;
        if (has_pos_of(i)) {
            keys.push_back(((uint64_t)((uint32_t)pos_of(i) ^ 0x80000000u) << 32) | (uint32_t)read_of.size());
            read_of.push_back(i);
        }
    }
    if (keys.size() > 2048) {
        std::vector<uint64_t> tmp(keys.size());
        uint64_t *src = keys.data(), *dst = tmp.data();
        for (int pass = 0; pass < 3; pass++) {
            const int shift = 32 + 11 * pass;
            uint32_t cnt[2049] = {0};
            for (size_t k = 0; k < keys.size(); k++) cnt[((src[k] >> shift) & 2047u) + 1]++;
            bool trivial = false;
            for (int b2 = 0; b2 < 2048; b2++) trivial = trivial || cnt[b2 + 1] == keys.size();
            if (trivial) continue;
            for (int b2 = 0; b2 < 2048; b2++) cnt[b2 + 1] += cnt[b2];
            for (size_t k = 0; k < keys.size(); k++) dst[cnt[(src[k] >> shift) & 2047u]++] = src[k];
            std::swap(src, dst);
        }
        if (src != keys.data()) std::memcpy(keys.data(), src, keys.size() * 8);
    } else
        std::sort(keys.begin(), keys.end());  // (position, rank) pairs are distinct: the order is the stable order by position
    Strand strand[2];  // forward, reverse
    for (Strand &t : strand) {
        t.pos.reserve(keys.size());
        t.read.reserve(keys.size());
        t.index.reserve(keys.size());
        t.psum.reserve(keys.size() + 1);
        t.psum.push_back(0);
    }
    int most_right = 0;
    for (size_t k = 0; k < keys.size(); k++) {
        const int idx = (int)(uint32_t)keys[k];
        const int i = read_of[(size_t)idx];
        const int p = (int)((uint32_t)(keys[k] >> 32) ^ 0x80000000u);
        Strand &t = strand[rev_of(i) ? 1 : 0];
        t.pos.push_back(p);
        t.read.push_back(i);
        t.index.push_back(idx);
        t.psum.push_back(t.psum.back() + p);
        most_right = p;
    }
    rg_lap("sort");
    return region_group_tail(strand, n, keys.size(), most_right, max_dist, keep_data_end, region, n_done, rg_lap);
}

}  // namespace

// the same grouping from keys the device has sorted (smi_umi_stage.hip, k_umi_region_keys): key = biased position << 32 | record << 1 | strand,
// n_pos of them; has_bits: one bit per record, set where the record has a position (ranks among those are what `n_done` counts in)
// The strands' arrays live in a workspace the context keeps between calls: a chunk of 300 k reads is ~10 MB of them, which malloc would map and
// unmap on every call -- page faults, and an unmap interrupts every thread of the process, which is what kept several lanes from scaling.
struct RegionWork {
    Strand strand[2];
    std::vector<uint32_t> rank_before;
};
void region_work_free(void *w) { delete static_cast<RegionWork *>(w); }

int region_group_from_sorted(void **work, const uint64_t *keys, size_t n_pos, int32_t n, const uint64_t *has_bits, int32_t max_dist, int keep_data_end,
                             int32_t *region, int32_t *n_done) {
    const bool rg_timing = std::getenv("SMI_RG_TIMING") != nullptr;
    auto rg_t0 = std::chrono::steady_clock::now();
    auto rg_lap = [&](const char *what) {
        if (!rg_timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "smi_region_group %-10s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - rg_t0).count());
        rg_t0 = now;
    };
    if (!*work) *work = new RegionWork();
    RegionWork &W = *static_cast<RegionWork *>(*work);
    std::fill(region, region + n, -1);
    std::vector<uint32_t> &rank_before = W.rank_before;  // records with a position in front of each 64-record word
    if (keep_data_end) {
        rank_before.resize(((size_t)n + 63) / 64 + 1);
        uint32_t acc = 0;
        for (size_t w = 0; w < rank_before.size(); w++) {
            rank_before[w] = acc;
            if (w * 64 < (size_t)n) acc += (uint32_t)__builtin_popcountll(has_bits[w]);
        }
    }
    Strand *strand = W.strand;
    for (int k = 0; k < 2; k++) {
        Strand &t = strand[k];
        t.pos.clear();
        t.read.clear();
        t.index.clear();
        t.psum.clear();
        t.pos.reserve(n_pos);
        t.read.reserve(n_pos);
        if (keep_data_end) t.index.reserve(n_pos);
        t.psum.reserve(n_pos + 1);
        t.psum.push_back(0);
    }
    int most_right = 0;
    for (size_t k = 0; k < n_pos; k++) {
        const uint64_t key = keys[k];
        const int p = (int)((uint32_t)(key >> 32) ^ 0x80000000u);
        const int i = (int)((uint32_t)key >> 1);
        Strand &t = strand[key & 1u];
        t.pos.push_back(p);
        t.read.push_back(i);
        if (keep_data_end) t.index.push_back((int)(rank_before[(size_t)i >> 6] + (uint32_t)__builtin_popcountll(has_bits[(size_t)i >> 6] & ((1ull << (i & 63)) - 1ull))));
        t.psum.push_back(t.psum.back() + p);
        most_right = p;
    }
    rg_lap("strands in");
    return region_group_tail(strand, n, n_pos, most_right, max_dist, keep_data_end, region, n_done, rg_lap);
}
}  // namespace smi

extern "C" int smi_region_group(const int32_t *pos, const uint8_t *has_pos, const uint8_t *reverse, int32_t n,
                                int32_t max_dist, int keep_data_end, int32_t *region, int32_t *n_done) {
    if (n < 0 || max_dist <= 0 || !n_done || (n && (!pos || !has_pos || !reverse || !region))) {
        set_error("smi_region_group: bad argument");
        return SMI_ERR_INVALID;
    }
    return smi::region_group_impl(
        n, [&](int i) { return pos[i]; }, [&](int i) { return has_pos[i] != 0; }, [&](int i) { return reverse[i] != 0; }, max_dist, keep_data_end, region,
        n_done);
}

extern "C" int smi_ref_position_at_read_position(const uint32_t *cigar, int32_t n_cigar, int32_t alignment_start,
                                                 int32_t position, int32_t *out) {
    if (!out || n_cigar < 0 || (n_cigar && !cigar)) {
        set_error("smi_ref_position_at_read_position: bad argument");
        return SMI_ERR_INVALID;
    }
    if (position == 0) return 0;
    int last_ref_end = 1, last_read_end = 1, read_at = 1, ref_at = alignment_start;
    for (int i = 0; i < n_cigar; i++) {
        const uint32_t op = cigar[i] & 15u;
        const int len = (int)(cigar[i] >> 4);
        switch (op) {
        case 1: case 4: read_at += len; break;  // I, S
        case 2: case 3: ref_at += len; break;   // D, N
        case 0: case 7: case 8: {               // M, =, X: an alignment block
            const int block_read = read_at, block_ref = ref_at;
            read_at += len;
            ref_at += len;
            if (block_read + len - 1 < position) {
                last_ref_end = block_ref + len - 1;
                last_read_end = block_read + len - 1;
                break;
            }
            *out = position < block_read ? block_ref - std::abs(block_ref - last_ref_end) / 2 : block_ref + position - block_read;
            return 1;
        }
        default: break;  // H, P
        }
    }
    if (position - last_read_end < 300) {
        *out = last_ref_end;
        return 1;
    }
    return 0;
}
