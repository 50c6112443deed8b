"""Independent plain-Python restatement of the reference's genomic-region grouping (objects and lists as in the Java:
Cluster with cached centre, ClusterList.refineClusters) -- NOT sharing code with oracle/sor_group.c.

Cites: FJ!umifinder/bamreaders/ReadGrouper.java:L82-260,L429-447, ReadGrouper$Cluster (L455-667),
ReadGrouper$ClusterList (L675-785), FJ!umifinder/reads/nanopore/NanoporeRead$ReadScanData.java:L109-153.
Region ids: the reference numbers clusters with a static counter (run-order dependent); only equality matters, so ids
here are the ordinal of the cluster in the final list.
"""
import math


def java_round_float(x):
    """Math.round(float): floor(x + 0.5f) in float arithmetic"""
    import numpy as np

    return int(math.floor(float(np.float32(np.float32(x) + np.float32(0.5)))))


class Cluster:
    def __init__(self, members=None):
        self.list = sorted(members, key=lambda r: r["pos"]) if members is not None else []
        self.center = None

    def get_center(self):
        if self.center is None and self.list:
            import numpy as np

            self.center = java_round_float(np.float32(sum(r["pos"] for r in self.list) / len(self.list)))
        return self.center

    def remove_off_center(self, pred):
        """lambda$new$5 L626-638, statement by statement: count with the predicate (the centre field may still hold the value from before an
        earlier removal), clear the field only when something was counted, filter again (the predicate now recomputes the centre from the
        list as it is), remove the members -- which does not touch the field"""
        n_removed = sum(1 for r in self.list if pred(r))
        if n_removed <= 0:
            return None
        self.center = None
        out = [r for r in self.list if pred(r)]
        for r in out:
            self.list.remove(r)
        return Cluster(out)


def refine(clusters, d):
    new_all = []
    cur = clusters
    while cur:
        nxt = []
        for c in cur:
            x = c.remove_off_center(lambda r: r["pos"] < c.get_center() - d)
            if x is not None:
                nxt.append(x)
            x = c.remove_off_center(lambda r: r["pos"] > c.get_center() + d)
            if x is not None:
                nxt.append(x)
        new_all += nxt
        cur = nxt
    lst = clusters + new_all
    lst = sorted([c for c in lst if c.list], key=lambda c: c.get_center())
    keep = True
    while keep:
        keep = False
        for i in range(len(lst) - 1):
            if not lst[i].list:
                continue
            left, right = lst[i], lst[i + 1]
            if right.get_center() - left.get_center() < 2 * d:
                left_bigger = len(left.list) > len(right.list)
                frm, to = (right, left) if left_bigger else (left, right)
                transfer = [r for r in frm.list if abs(r["pos"] - to.center) <= d]
                if transfer:
                    keep = True
                    to.list += transfer
                    to.center = None
                    for r in transfer:
                        frm.list.remove(r)
                    frm.center = None
        lst = [c for c in lst if c.list]
    return [c for c in lst if len(c.list) > 1]


def one_strand(data, idx, d):
    if len(idx) <= 1:
        return []
    clusters = []
    cur = Cluster()
    if data[idx[1]]["pos"] - data[idx[0]]["pos"] < d:
        cur.list.append(data[idx[0]])
    for i in range(1, len(idx)):
        if data[idx[i]]["pos"] - data[idx[i - 1]]["pos"] < d:
            cur.list.append(data[idx[i]])
            cur.center = None
        elif len(cur.list) > 2:
            clusters.append(cur)
            cur = Cluster()
    if len(cur.list) > 2:
        clusters.append(cur)
    return refine(clusters, d)


def group_sams(pos, reverse, d=500, keep_data_end=False):
    """pos[i] None = no position; -> (region id per read or -1, number of reads of the done chunk)"""
    n = len(pos)
    data = []
    for i in range(n):
        if pos[i] is not None:
            data.append(dict(pos=pos[i], rev=bool(reverse[i]), read=i, index=len(data)))
    data.sort(key=lambda r: r["pos"])
    fwd = [i for i, r in enumerate(data) if not r["rev"]]
    rev = [i for i, r in enumerate(data) if r["rev"]]
    clusters = one_strand(data, fwd, d) + one_strand(data, rev, d)
    clusters = sorted([c for c in clusters if c.list], key=lambda c: c.get_center())
    last_index = n - 1
    if keep_data_end and clusters and data:
        most_right = data[-1]["pos"]
        while clusters and clusters[-1].center > most_right - 3 * d:
            clusters.pop()
        if clusters:
            last_index = max(r["index"] for r in clusters[-1].list)
            if last_index < n // 3:
                last_index = n // 3
    region = [-1] * n
    for k, c in enumerate(clusters):
        for r in c.list:
            region[r["read"]] = k
    return region, last_index + 1


def ref_position_at_read_position(cigar, alignment_start, position):
    """cigar: [(op char, len)]; SAMRecord.getAlignmentBlocks + getReferencePositionAtReadPosition L133-153"""
    if position == 0:
        return None
    blocks = []
    read_base, ref_base = 1, alignment_start
    for op, ln in cigar:
        if op in "SI":
            read_base += ln
        elif op in "ND":
            ref_base += ln
        elif op in "M=X":
            blocks.append((read_base, ref_base, ln))
            read_base += ln
            ref_base += ln
    last_genomic_end = last_read_end = 1
    for rs, gs, ln in blocks:
        if rs + ln - 1 < position:
            last_genomic_end, last_read_end = gs + ln - 1, rs + ln - 1
            continue
        if position < rs:
            return gs - abs(gs - last_genomic_end) // 2
        return gs + position - rs
    return last_genomic_end if position - last_read_end < 300 else None
