"""one rank of the sharded assignumis test (started by torch.distributed.run): every rank uses the box's one GPU (gloo for the one exchange)"""
import importlib
import json
import os
import sys

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    in_bam, out_prefix, refflat, chunk = sys.argv[1:5]
    dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    pkg = graft.load_package()
    assignumis = importlib.import_module(graft.PKG_NAME + ".assignumis")
    ctx = pkg.Context(0)
    info = assignumis.assignumis_stream(ctx, in_bam, out_prefix, segment_bytes=9_000, chunk_size=int(chunk), n_threads=2,
                                        refflat=open(refflat).read() if refflat != "-" else None)
    with open(out_prefix + f".info_rank{dist.get_rank()}.json", "w") as f:
        json.dump({k: v for k, v in info.items() if isinstance(v, (int, float, str))}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
