"""one rank of the multi-process run_files test (started by torch.distributed.run): every rank uses the box's one GPU (gloo for the exchanges)"""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    in_dir, out_dir, keys_file = sys.argv[1:4]
    mode = sys.argv[4] if len(sys.argv) > 4 else "whitelist"      # whitelist (default two-pass) | none (-a none) | given (-g: keys_file is the list)
    dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    pkg = graft.load_package()
    run_files = importlib.import_module(graft.PKG_NAME + ".run_files")
    ctx = pkg.Context(0)
    keys = np.load(keys_file)
    kw = dict(whitelist_keys=keys) if mode == "whitelist" else dict(whitelist_keys=None) if mode == "none" else dict(whitelist_keys=None, used_keys=keys)
    info = run_files.run(ctx, in_dir, out_dir, max_ed=1, n_workers=3, reads_per_chunk=1000, gz="device", **kw)
    with open(os.path.join(out_dir, f"info_rank{dist.get_rank()}.json"), "w") as f:
        json.dump({k: v for k, v in info.items() if isinstance(v, (int, float, str)) or v is None}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
