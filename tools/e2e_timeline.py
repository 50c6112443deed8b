#!/usr/bin/env python3
"""Timeline of the last end-to-end repetition out of a rocprofv3 kernel trace: per dispatch its start (relative to the repetition's first
kernel), duration and the gap to the end of whatever ran before it.  usage: e2e_timeline.py <dir with *kernel_trace.csv> <out.json>"""
import csv
import glob
import json
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    files = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"]
            if "smi::" not in nm and "hipcub" not in nm and "rocprim" not in nm:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm))
    rows.sort()
    # a repetition starts at the line index's first kernel
    starts = [i for i, r in enumerate(rows) if "k_fq_count" in r[2]]
    if not starts:
        raise SystemExit("no line-index dispatch in the trace")
    a = starts[-1]
    b = max(i for i, r in enumerate(rows) if i >= a and ("k_write(" in r[2] or "k_write_name" in r[2])) + 1
    t0 = rows[a][0]
    tl, end_prev, busy = [], t0, 0
    for s, e, nm in rows[a:b]:
        short = nm.split("(")[0].replace("void ", "").replace("smi::", "")
        if "rocprim" in short or "hipcub" in short:
            short = "scan:" + short.split("::")[-1][:40]
        tl.append({"kernel": short[:60], "start_us": round((s - t0) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1), "gap_us": round((s - end_prev) / 1e3, 1)})
        busy += max(0, e - max(s, end_prev))
        end_prev = max(end_prev, e)
    span = end_prev - t0
    json.dump({"what": "last end-to-end repetition of the trace: dispatches in start order; gap = idle time in front of the dispatch (negative: overlaps the one before)",
               "span_ms": span / 1e6, "busy_ms": busy / 1e6, "idle_ms": (span - busy) / 1e6, "dispatches": len(tl), "timeline": tl}, open(out, "w"), indent=1)
    print(f"span {span/1e6:.3f} ms busy {busy/1e6:.3f} ms idle {(span-busy)/1e6:.3f} ms over {len(tl)} dispatches")
    for x in tl:
        if x["gap_us"] > 15:
            print("  gap", x["gap_us"], "us in front of", x["kernel"])


main()
