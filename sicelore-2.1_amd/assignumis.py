"""Host-side mirror of the reference's `assignumis` worker for one chunk of aligned reads (UmiFinderWorker ->
ReadGrouper.groupSams -> UmiClustering.cluster, FJ!umifinder/UmiFinderWorker.java, FJ!umifinder/bamreaders/ReadGrouper.java:L82-260,
FJ!umifinder/analyzers/clustering/UmiClustering.java:L97-161): region grouping (host C++), UMI pair distances (K-UMI on
the device), clustering (host C++), and the values of the tags U8 / U7 / U1 / U2.  3' barcoding.  No CPU fallback.

Per read the caller supplies what the reference parses from the BAM record: the read name written by scanfastq
(FastqRecordExt.getScanDatFromReadName, FastqRecordExt.java:L395-496: X=, AE=, bcEnd=, Q=, cellBC), the strand flag
and the clustering position (NanoporeRead$ReadScanData.getGenomePosition; smi_ref_position_at_read_position).
"""
import re

import numpy as np
import torch

from . import lib as _lib

_CODE = {"A": 1, "G": 2, "C": 4, "T": 8, "N": 15}
_DEC = {1: "A", 2: "G", 4: "C", 8: "T", 15: "N"}
_COMP = {1: 8, 8: 1, 2: 4, 4: 2, 15: 15}
_FIELD = {k: re.compile(r"_" + k + r"=([^_ ]+)") for k in ("AE", "bcEnd", "X", "Q")}


def parse_name(name):
    """fields of a scanfastq read name needed here; None when the read carries no barcode"""
    m = {k: r.search(name) for k, r in _FIELD.items()}
    cell = name.split(" cellBC=")
    if len(cell) != 2 or any(v is None for v in m.values()):
        return None
    return dict(cell=cell[1].split()[0], ae=int(m["AE"].group(1)), bc_end=int(m["bcEnd"].group(1)), x=m["X"].group(1),
                q=float(m["Q"].group(1)))


def umi_window(x, adapter_end, bc_end):
    """14 bases as 4-bit codes: the three 12-mers at offsets -1, 0, +1 behind the barcode on the reverse complement of X=
    (ClusteringEditDistanceBase L297-350; getStrandedShortSeqPosFromReadPos FastqRecordExt.java:L378); None if out of range"""
    pos = adapter_end + 3 - bc_end
    if pos < 1 or pos + 13 > len(x):
        return None
    return [_COMP[_CODE.get(x[len(x) - (pos + k)], 15)] for k in range(14)]


def pack_window(w):
    v = 0
    for k, c in enumerate(w):
        v |= c << (4 * k)
    return v


def assign_umis(ctx, names, positions, reverse, max_dist=500, cluster_cfg=None, n_threads=4):
    """-> list (one per read) of None or dict(U8, U7, U1, U2, region, center) -- the tag VALUES the reference writes in
    ClusterOneBase.setSamflagsAndStatsForClustered; reads without barcode / position / neighbours get None"""
    n = len(names)
    info = [parse_name(nm) for nm in names]
    pos = [p if info[i] is not None else None for i, p in enumerate(positions)]
    region, _ = _lib.region_group(pos, reverse, max_dist=max_dist, keep_data_end=False)
    wins = [umi_window(f["x"], f["ae"], f["bc_end"]) if f is not None else None for f in info]
    groups = {}
    for i in range(n):
        if info[i] is not None and region[i] >= 0 and wins[i] is not None:
            groups.setdefault((info[i]["cell"], region[i]), []).append(i)  # canonical member order: input order
    groups = [g for g in groups.values() if len(g) > 1]  # UmiClustering.lambda$cluster$6
    out = [None] * n
    if not groups:
        return out
    order = [i for g in groups for i in g]
    sizes = [len(g) for g in groups]
    go, po, mo = ctx.umi_offsets(sizes)
    dev = torch.device("cuda", ctx.device)
    packed = np.array([pack_window(wins[i]) for i in order], dtype=np.uint64)
    d_out = torch.zeros(int(mo[-1]), dtype=torch.uint8, device=dev)
    ctx.umi_dist_device(torch.from_numpy(packed.view(np.int64)).to(dev), torch.from_numpy(go.view(np.int32)).to(dev),
                        torch.from_numpy(po.view(np.int64)).to(dev), torch.from_numpy(mo.view(np.int64)).to(dev),
                        len(sizes), int(po[-1]), d_out)
    torch.cuda.synchronize()
    qv = np.array([info[i]["q"] for i in order], dtype=np.float32)
    asg, _skipped = _lib.umi_cluster_groups(d_out.cpu().numpy(), mo, go, qv, cluster_cfg, n_threads=n_threads)
    base = np.repeat(go[:-1], sizes)
    for j, i in enumerate(order):
        a = asg[j]
        if a["center"] < 0:
            continue
        cw = wins[order[int(base[j]) + int(a["center"])]]
        off = int(a["offset"])
        out[i] = dict(U8="".join(_DEC[c] for c in cw[off + 1:off + 13]), U7="".join(_DEC[c] for c in wins[i][1:13]),
                      U1=int(a["ed"]), U2=None if a["ed_second"] < 0 else int(a["ed_second"]), region=region[i],
                      center=order[int(base[j]) + int(a["center"])])
    return out
