"""Random-gather rate of the GPU by table size (L2 4 MiB per XCD, Infinity Cache 256 MiB, HBM beyond): torch.take of 64 M random int32 per call.
usage (GPU box): python tools/gather_probe.py > profiles/r05/gather_probe.json"""
import json

import torch

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(7)
m = 1 << 26
rows = []
for mib in (1, 4, 16, 32, 64, 128, 256, 512, 1024, 2048, 8192):
    words = mib << 18
    table = torch.ones(words, dtype=torch.int32, device=dev)
    idx = torch.randint(0, words, (m,), device=dev, generator=g, dtype=torch.int64)
    torch.take(table, idx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        torch.take(table, idx)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 3 * 1e-3
    rows.append({"table_MiB": mib, "gathers_G_per_s": m / t / 1e9})
    del table, idx
print(json.dumps({"what": "independent random 4-byte gathers (torch.take, 64 M per call, int64 indices streamed beside them) by table size", "rows": rows}))
