"""`assignumis` on a BAM that is read in segments (assignumis.assignumis_stream): the UMI leg's records three times over, on three chromosomes,
as a BAM file in /dev/shm -> tagged BAMs + tables; one JSON line (profiles/r03/assignumis_stream.json).  GPU box."""
import importlib
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    pkg = graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    lib = importlib.import_module(graft.PKG_NAME + ".lib")
    au = importlib.import_module(graft.PKG_NAME + ".assignumis")
    ctx = pkg.Context(0)
    wl = synth.make_whitelist(3_600_000, seed=1)
    used = synth.pick_used(wl, 5000, seed=2)
    bench.umi_stage_leg(pkg, synth, ctx, used, int(os.environ.get("SMI_ASB_MOLECULES", "50000")))
    rows = bench.umi_stage_leg.rows
    big = [(max(p, 0) + 1_000_000, f"c{r}{nm}", fl, r) for r in range(3) for p, nm, fl in rows]
    t0 = time.perf_counter()
    raw = synth.bam_from_rows(big, ref_names=("chr1", "chr2", "chr3"))
    z = lib.bgzf_deflate(raw, level=1, n_threads=16)
    td = tempfile.mkdtemp(dir="/dev/shm")
    try:
        path = os.path.join(td, "in.bam")
        z.tofile(path)
        gen_s = time.perf_counter() - t0
        del raw
        res = {"records": len(big), "bam_file_bytes": int(z.size), "generate_s": gen_s, "runs": []}
        for seg in (256 << 20, 1 << 30, 1 << 40):
            info = au.assignumis_stream(ctx, path, os.path.join(td, "out"), segment_bytes=seg, n_threads=16)
            res["runs"].append({"segment_bytes": seg if seg < (1 << 39) else "whole file", "records_per_s": info["records"] / info["wall_s"], "wall_s": info["wall_s"],
                                "batches": info["batches"], "clustered": info["clustered"], "seconds": info["seconds"],
                                "out_bam_bytes": os.path.getsize(os.path.join(td, "out.bam")), "umifound_bam_bytes": os.path.getsize(os.path.join(td, "out_umifound_.bam"))})
        same = len({(r["clustered"], r["batches"]) for r in res["runs"]}) == 1
        res["same_counts_for_every_segment_size"] = same
        print(json.dumps(res))
    finally:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
