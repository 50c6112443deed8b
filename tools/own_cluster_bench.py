#!/usr/bin/env python3
"""ClusterOne_MyClustering (one (cell, region) group above 100 reads) through smi_assignumis_chunk: its n^2 loops on the device (default)
against the host clusterer on the downloaded matrix (SMI_AU_OWN_HOST=1).  SMI_AU_TIMING=1 makes the library print its stage laps to stderr;
the "big groups" lap is the clusterer.  Prints one JSON object; tags must be equal."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    pkg = graft.load_package()
    from sicelore_amd import lib as libmod
    from test_umi_gpu import _make_groups, _name_with_window

    rng = np.random.default_rng(n)
    n_umi = max(3, n // 12)
    base = _make_groups(n + 1, [n_umi])
    pick = np.minimum(rng.zipf(1.4, n) - 1, n_umi - 1)
    ws = base[pick].copy()
    noisy = rng.random(n) < 0.3
    ws[noisy, rng.integers(0, 14, int(noisy.sum()))] = rng.choice([1, 2, 4, 8], int(noisy.sum()))
    qs = [f"{v:.1f}".rstrip("0").rstrip(".") for v in rng.uniform(8, 25, n)]
    names = [_name_with_window(i, ws[i], qs[i]) for i in range(n)]
    flags = np.zeros(n, dtype=np.uint16)
    pos0 = np.sort((100_000 + rng.integers(0, 50, n)).astype(np.int32))
    cigars = [np.array([1000 << 4], dtype=np.uint32)] * n
    ctx = pkg.Context(0)
    ccfg = libmod.umi_cluster_config()
    res = {"reads": n}
    out = {}
    for mode in ("device", "host"):
        if mode == "host":
            os.environ["SMI_AU_OWN_HOST"] = "1"
        else:
            os.environ.pop("SMI_AU_OWN_HOST", None)
        ctx.assignumis_chunk(names, flags, pos0, cigars, n_threads=16, cluster_cfg=ccfg)     # warm-up (buffers)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            tags, n_done = ctx.assignumis_chunk(names, flags, pos0, cigars, n_threads=16, cluster_cfg=ccfg)
            ts.append(time.perf_counter() - t0)
        out[mode] = tags.copy()
        res[mode + "_chunk_ms"] = min(ts) * 1e3
    res["equal"] = bool((out["device"] == out["host"]).all())
    res["clustered"] = int(((out["device"]["flags"] & libmod.UMI_CLUSTERED) != 0).sum())
    res["clusterer_ms_saved"] = res["host_chunk_ms"] - res["device_chunk_ms"]
    print(json.dumps(res))
    sys.exit(0 if res["equal"] else 1)


if __name__ == "__main__":
    main()
