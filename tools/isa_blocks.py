#!/usr/bin/env python3
"""Per-basic-block VALU/SALU/VMEM/LDS instruction counts of one kernel in a hipcc -S listing.
usage: isa_blocks.py file.s kernel-name-substring [min_instrs]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = re.search(r"^(\S*" + re.escape(pat) + r"\S*):[^\n]*\n(.*?)\.Lfunc_end\d+", s, re.S | re.M)
lines = [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")
         and (not l.strip().startswith(".") or l.strip().startswith(".LBB"))]
blocks, cur = [], ["entry", []]
for l in lines:
    if l.startswith(".LBB") and l.split()[0].endswith(":"):
        blocks.append(cur)
        cur = [l.split()[0], []]
    else:
        cur[1].append(l)
blocks.append(cur)
tot = collections.Counter()
for name, ls in blocks:
    cc = collections.Counter("valu" if l.startswith("v_") else "salu" if l.startswith("s_") else "vmem" if
                             l.startswith(("global_", "buffer_", "flat_")) else "lds" if l.startswith("ds_") else "o" for l in ls)
    tot.update(cc)
    br = [l.split()[-1] for l in ls if "branch" in l]
    if sum(cc.values()) >= thr:
        print(name, dict(cc), br)
print("TOTAL", dict(tot))
