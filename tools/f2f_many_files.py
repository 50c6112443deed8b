"""scanfastq file to file on MANY files of ONT's default size (4,000 reads): the host's decoder alone against the two-ended queue with
K-INFLATE (run_files inflate="host" / "auto").  -> one JSON line (profiles/r03/f2f_many_files.json).  GPU box; inputs in /dev/shm."""
import importlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    n_files = int(os.environ.get("SMI_F2F_FILES", "2048"))
    per_file = int(os.environ.get("SMI_F2F_READS_PER_FILE", "4000"))
    pkg = graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    run_files = importlib.import_module(graft.PKG_NAME + ".run_files")
    ctx = pkg.Context(0)
    dev = torch.device("cuda", 0)
    wl = synth.make_whitelist(3_600_000, seed=1, device=dev)
    used = synth.pick_used(wl, 5000, seed=2)
    keys = np.sort(wl.cpu().numpy().astype(np.uint64))
    base = tempfile.mkdtemp(prefix="smi_f2f_many_", dir="/dev/shm")
    in_dir, out_dir = os.path.join(base, "in"), os.path.join(base, "out")
    res = {"files": n_files, "reads_per_file": per_file, "runs": []}
    try:
        t0 = time.perf_counter()
        n = run_files.write_synthetic_dir(synth, in_dir, n_files, per_file, used, dev, seed=9000, chimera_frac=0.05)
        res["generate_inputs_s"] = time.perf_counter() - t0
        res["reads"] = n
        res["gz_in_bytes"] = sum(os.path.getsize(os.path.join(in_dir, f)) for f in os.listdir(in_dir))
        for mode, kw in (("host", {}), ("auto", {"device_share": 0.125}), ("auto", {"device_share": 0.25}), ("host", {})):
            shutil.rmtree(out_dir, ignore_errors=True)
            info = run_files.run(ctx, in_dir, out_dir, max_ed=1, n_workers=16, reads_per_chunk=100_000, whitelist_keys=keys, gz="device", inflate=mode, **kw)
            res["runs"].append({"inflate": mode, **kw, **{k: info[k] for k in ("reads_per_s", "wall_s", "inflate_and_pass1_s", "inflate_thread_seconds", "pass2_and_gzip_s",
                                                                                "write_files_s", "files_inflated_on_device", "device_inflate_rounds", "text_in_bytes", "text_resident_bytes", "assigned")}})
            print(json.dumps(res["runs"][-1]), file=sys.stderr, flush=True)
        res["same_assigned"] = len({r["assigned"] for r in res["runs"]}) == 1
        print(json.dumps(res))
    finally:
        shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
