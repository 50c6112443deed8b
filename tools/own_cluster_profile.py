#!/usr/bin/env python3
"""gpurun_out/own_cluster_8000.{json,err} (tools/own_cluster_bench.py under SMI_AU_TIMING=1) -> profiles/r04/own_cluster_8000.json: the bench's
JSON line plus the clusterer's own laps (the "big groups" lap of every call, the steps of the last device call)."""
import json
import re
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/own_cluster_8000"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04/own_cluster_8000.json"
err = open(src + ".err").read().splitlines()
big = [float(ln.split()[-2]) for ln in err if "big groups" in ln]
steps = [re.sub(r"\s+", " ", ln.strip()) for ln in err if "own clusterer" in ln]
d = json.load(open(src + ".json"))
half = len(big) // 2
d["clusterer_lap_ms"] = {"device": big[:half], "host": big[half:],
                         "note": "the \"big groups\" lap of smi_assignumis_chunk (SMI_AU_TIMING=1): ClusterOne_MyClustering of the one group; "
                                 "the first call of each kind is the warm-up"}
per_call = len(steps) // max(half, 1)
d["device_steps_last_call"] = steps[-per_call:] if per_call else []
json.dump(d, open(dst, "w"), indent=1)
print(json.dumps(d["clusterer_lap_ms"]))
