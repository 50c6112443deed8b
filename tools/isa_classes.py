#!/usr/bin/env python3
"""Issue-class table of one kernel from a `hipcc -S --cuda-device-only` listing: per basic block, how many VALU instructions are
2-cycle forms (v_and / or / xor / add / sub / subrev _u32, v_lshrrev, v_ashrrev, v_mov, v_not, v_add_f32 with VGPR / inline / literal operands
only -- tools/valu_peak.hip, profiles/r02/valu_peak.json), how many of those sit in PURE runs of eight or more (the only place the 2-cycle
rate is reached: one 4-cycle form among four brings all of them to four cycles), and how many are 4-cycle forms (every VOP3 three-operand
form, v_max / v_min, v_lshlrev, anything with an SGPR operand, compares, selects, cross-lane ops ...).  Static counts: a block's weight at
run time is not in the listing; the blocks are listed with their loop back-edges so that the inner loops can be told apart.

usage: isa_classes.py listing.s kernel-name-substring out.json"""
import collections
import json
import re
import sys

TWO_CYCLE = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_not_b32",
             "v_add_f32", "v_add_nc_u32", "v_sub_nc_u32"}


def is_two_cycle(line):
    parts = line.replace(",", " ").split()
    op = parts[0]
    base = op[:-4] if op.endswith(("_e32", "_e64")) else op
    if base not in TWO_CYCLE or op.endswith("_e64") or "dpp" in line or "sdwa" in line:
        return False
    return not any(re.fullmatch(r"s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0|ttmp\d+", a) for a in parts[1:])


def main():
    s = open(sys.argv[1]).read()
    pat, dst = sys.argv[2], sys.argv[3]
    m = re.search(r"^(\S*" + re.escape(pat) + r"\S*):[^\n]*\n(.*?)\.Lfunc_end\d+", s, re.S | re.M)
    lines = [ln.strip() for ln in m.group(2).split("\n")]
    lines = [ln.split(";")[0].strip() for ln in lines if ln and not ln.startswith(";")]
    lines = [ln for ln in lines if ln and (not ln.startswith(".") or ln.startswith(".LBB"))]
    blocks, cur = [], ["entry", []]
    for ln in lines:
        if ln.startswith(".LBB") and ln.split()[0].endswith(":"):
            blocks.append(cur)
            cur = [ln.split()[0][:-1], []]
        else:
            cur[1].append(ln)
    blocks.append(cur)
    order = {name: k for k, (name, _) in enumerate(blocks)}
    out, tot = [], collections.Counter()
    for name, ls in blocks:
        c = collections.Counter()
        run = 0
        runs = []
        for ln in ls:
            if ln.startswith("v_"):
                c["valu"] += 1
                if is_two_cycle(ln):
                    c["two_cycle_form"] += 1
                    run += 1
                    continue
                c["four_cycle_form"] += 1
            elif ln.startswith("s_"):
                c["salu"] += 1
            elif ln.startswith("ds_"):
                c["lds"] += 1
            elif ln.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
            if run:
                runs.append(run)
                run = 0
        if run:
            runs.append(run)
        c["two_cycle_in_pure_runs_of_8_or_more"] = sum(r for r in runs if r >= 8)
        # issue cycles of the block's VALU work per wave: 2 per instruction of a long pure run, 4 for everything else
        c["valu_issue_cycles_est"] = 2 * c["two_cycle_in_pure_runs_of_8_or_more"] + 4 * (c["valu"] - c["two_cycle_in_pure_runs_of_8_or_more"])
        back = [t for t in (ln.split()[-1] for ln in ls if "branch" in ln) if t in order and order[t] <= order[name]]
        tot.update(c)
        if c["valu"] >= 16:
            out.append(dict(block=name, loop_back_edge_to=back, **c))
    res = {"kernel": m.group(1), "source": "hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only smi_scan.hip", "total": dict(tot),
           "two_cycle_share_of_valu": tot["two_cycle_form"] / max(tot["valu"], 1),
           "pure_run_share_of_valu": tot["two_cycle_in_pure_runs_of_8_or_more"] / max(tot["valu"], 1),
           "largest_blocks": sorted(out, key=lambda b: -b["valu"])[:40]}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("kernel", "total", "two_cycle_share_of_valu", "pure_run_share_of_valu")}))


if __name__ == "__main__":
    main()
