"""File to file (sicelore-2.1_amd/run_files.py = `scanfastq -d <dir> -o <dir> --bcEditDistance 1 --compress`, both passes): the records
in `<out>/passed/*_passed.fastq.gz` / `<out>/failed/*_failed.fastq.gz` equal the chunk worker's records over the same reads after the
canonicalisation SURVEY 8c prescribes (sort by original read name, strip the base-36 read id, whose order the reference does not
reproduce either); BarcodeList.tsv and BarcodesAssigned.tsv equal the single-call results."""
import gzip
import importlib
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
_ID = re.compile(rb"(_Q=[0-9.]+)_[0-9a-z]+")


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _canon(text):
    lines = text.split(b"\n")
    assert lines[-1] == b""
    recs = []
    for k in range(0, len(lines) - 1, 4):
        name = _ID.sub(rb"\1_", lines[k], count=1)
        recs.append((name, lines[k + 1], lines[k + 2], lines[k + 3]))
    return sorted(recs)


@pytest.mark.parametrize("gz,inflate,resident,host_budget", [("device", "host", 96 << 30, 1 << 40), ("zlib", "host", 0, 1 << 40), ("device", "device", 0, 1 << 40),
                                                             ("device", "host", 12_000_000, 1 << 40), ("device", "host", 11_000_000, 13_000_000), ("zlib", "host", 0, 0), ("device", "auto", 96 << 30, 1 << 40)])
def test_directory_run_equals_chunk_worker(pkg, synth, gpu_ctx, tmp_path, gz, inflate, resident, host_budget):
    from sicelore_amd import lib as libmod

    run_files = importlib.import_module("sicelore_amd.run_files")
    dev = torch.device("cuda", gpu_ctx.device)
    wl = synth.make_whitelist(60_000, seed=801, device=dev)
    used = synth.pick_used(wl, 60, seed=802)
    keys = np.sort(wl.cpu().numpy().astype(np.uint64))
    in_dir, out_dir = str(tmp_path / "in"), str(tmp_path / "out")
    n = run_files.write_synthetic_dir(synth, in_dir, 6, 2500, used, dev, seed=810, chimera_frac=0.08)
    info = run_files.run(gpu_ctx, in_dir, out_dir, max_ed=1, n_workers=4, reads_per_chunk=1000, whitelist_keys=keys, gz=gz, inflate=inflate,
                         resident_bytes=resident, host_text_bytes=host_budget, inflate_auto_from=4, device_share=0.34)
    # "auto": host threads and the device work the queue of files from its two ends (rounds of two files here)
    assert info["files_inflated_on_device"] == (6 if inflate == "device" else 0) or (inflate == "auto" and 2 <= info["files_inflated_on_device"] <= 6)
    # files whose text found no room between the passes are inflated again in pass 2 (none / some / all of the six)
    assert info["files_inflated_twice"] == (6 if host_budget == 0 else 0) or (host_budget == 13_000_000 and 1 <= info["files_inflated_twice"] <= 4)
    # the text of host-inflated files stays in HBM between the passes while the budget lasts (all of it / none / about two files of six)
    if inflate == "auto":
        assert 0 < info["text_resident_bytes"] <= info["text_in_bytes"]
    elif inflate == "host" and gz == "device":
        # (a soft limit: the worker threads race for the budget's last bytes)
        assert (info["text_resident_bytes"] == info["text_in_bytes"]) if resident > (1 << 30) else (0 < info["text_resident_bytes"] < info["text_in_bytes"]), \
            (info["text_resident_bytes"], info["text_in_bytes"])
    else:
        assert info["text_resident_bytes"] == 0
    assert info["gz"] == gz and info["gz_out_bytes"] < 0.7 * info["text_out_bytes"]
    assert info["reads"] == n and info["files"] == 6 and info["chunks"] >= 12 and info["passed"] > 0.8 * n
    # the same reads through single calls of the chunk workers (text form), file by file
    texts = [gzip.open(os.path.join(in_dir, f)).read() for f in sorted(os.listdir(in_dir))]
    gpu_ctx.set_barcode_set(keys, mode=1)
    hist = torch.zeros(keys.size, dtype=torch.int32, device=dev)
    n_rec = [gpu_ctx.scanfastq_pass1_chunk(t, hist) for t in texts]
    h = hist.cpu().numpy()
    nz = np.nonzero(h)[0]
    record_count = sum((m + 9_999) // 10_000 for m in n_rec)      # FastqFileReader's 10,000-read chunks per file
    k, c, r = libmod.finalize_used_list(keys[nz], h[nz].astype(np.uint32), record_count, 1, 10, 500)
    assert info["used_list"] == k.size and 40 <= k.size <= 70
    assert open(os.path.join(out_dir, "BarcodeList.tsv")).read() == libmod.barcode_list_tsv(keys[nz], h[nz].astype(np.uint32), record_count, 1)
    order = np.argsort(k)
    gpu_ctx.set_barcode_set(k, mode=0)
    counts = np.zeros((k.size, 3), dtype=np.int64)
    for fi, t in enumerate(texts):
        p, f, inf = gpu_ctx.scanfastq_pass2_chunk(t, max_ed=1, rank_keys=k[order], rank_values=r[order].astype(np.int32), want_results=True)
        base = f"synth_{fi:04d}"
        got_p = gzip.open(os.path.join(out_dir, "passed", base + "_passed.fastq.gz")).read()
        got_f = gzip.open(os.path.join(out_dir, "failed", base + "_failed.fastq.gz")).read()
        assert _canon(got_p) == _canon(p) and _canon(got_f) == _canon(f)
        assert got_p.count(b"_rk=") > 1000
        bc = inf["bc"]
        ok = bc["found"] == 1
        np.add.at(counts, (np.searchsorted(k[order], bc["bc"][ok].astype(np.uint64)), bc["ed"][ok].astype(np.int64)), 1)
    assert open(os.path.join(out_dir, "BarcodesAssigned.tsv")).read() == libmod.assigned_tsv(k[order], counts.astype(np.uint32), max_ed=1)
    assert info["assigned"] == int(counts.sum())
    # the statistics file: the counters of ReadFlags.print, consistent with the records that were written; mergestats adds runs up
    rows = {ln.split("\t")[0]: ln.split("\t")[1:] for ln in open(os.path.join(out_dir, "ReadScanner.tsv")).read().split("\n")[2:] if ln}
    num = lambda key: int(rows[key][0].replace(",", ""))  # noqa: E731
    assert num("All Reads") == n and num("Reads after chimera split") == info["records_out"] and num("Passed (Adapter found)") == info["passed"]
    assert num("Barcode found") == info["assigned"] and num("Chimeric reads split") > 50
    assert num("Passed forward") + num("Passed reverse") == info["passed"]
    total = run_files.merge_stats([out_dir, out_dir], str(tmp_path / "merged"))
    merged = open(str(tmp_path / "merged" / "ReadScanner.tsv")).read()
    assert f"All Reads\t{2 * n:,}" in merged and int(total[3]) == 2 * info["records_out"]


@pytest.mark.parametrize("mode", ["whitelist", "none", "given"])
def test_two_ranks_write_what_one_process_writes(pkg, synth, gpu_ctx, tmp_path, mode):
    """run_files under torch.distributed (two ranks sharing the box's GPU, gloo): files dealt to the ranks, one histogram all-reduce, read ids
    continued across the ranks, counters and statistics summed -- every output file and TSV equals the single-process run's, byte for byte"""
    import json
    import subprocess
    import sys

    run_files = importlib.import_module("sicelore_amd.run_files")
    dev = torch.device("cuda", gpu_ctx.device)
    wl = synth.make_whitelist(60_000, seed=811, device=dev)
    used = synth.pick_used(wl, 60, seed=812)
    keys = np.sort(wl.cpu().numpy().astype(np.uint64))
    in_dir, one, two = str(tmp_path / "in"), str(tmp_path / "one"), str(tmp_path / "two")
    n = run_files.write_synthetic_dir(synth, in_dir, 5, 1800, used, dev, seed=820, chimera_frac=0.08)
    # mode: the default two-pass run | -a none (the ranks' key tables are gathered and added up) | -g (a supplied list: no pass-1 exchange at all)
    if mode == "given":
        keys = np.sort(used.cpu().numpy().astype(np.uint64))
    kw = dict(whitelist_keys=keys) if mode == "whitelist" else dict(whitelist_keys=None) if mode == "none" else dict(whitelist_keys=None, used_keys=keys)
    a = run_files.run(gpu_ctx, in_dir, one, max_ed=1, n_workers=3, reads_per_chunk=1000, gz="device", **kw)
    np.save(str(tmp_path / "keys.npy"), keys)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "tests", "_run_files_rank.py"), in_dir, two, str(tmp_path / "keys.npy"), mode]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert p.returncode == 0, p.stderr[-3000:]
    infos = [json.load(open(os.path.join(two, f"info_rank{r}.json"))) for r in (0, 1)]
    assert [i["files"] for i in infos] == [3, 2] and sum(i["reads"] for i in infos) == n == a["reads"]
    assert infos[0]["assigned"] == infos[1]["assigned"] == a["assigned"] and infos[0]["used_list"] == a["used_list"]
    for sub in ("passed", "failed"):
        names = sorted(os.listdir(os.path.join(one, sub)))
        assert names == sorted(os.listdir(os.path.join(two, sub))) and len(names) == 5
        for f in names:
            assert open(os.path.join(one, sub, f), "rb").read() == open(os.path.join(two, sub, f), "rb").read(), f
    for f in ("BarcodesAssigned.tsv", "ReadScanner.tsv", "stats.tsv") + (() if mode == "given" else ("BarcodeList.tsv",)):
        assert open(os.path.join(one, f)).read() == open(os.path.join(two, f)).read(), f


def test_supplied_barcode_list_dontwrite_trim_and_file_selection(pkg, synth, gpu_ctx, tmp_path):
    """-g (pass 1 skipped, the supplied list searched, rank 0 = no rk=), -s (no FASTQ written, same tables), -u (trimmed records), and the file
    selection options, against the default run of the same directory"""
    run_files = importlib.import_module("sicelore_amd.run_files")
    dev = torch.device("cuda", gpu_ctx.device)
    wl = synth.make_whitelist(60_000, seed=821, device=dev)
    used = synth.pick_used(wl, 50, seed=822)
    keys = np.sort(wl.cpu().numpy().astype(np.uint64))
    in_dir = str(tmp_path / "in")
    n = run_files.write_synthetic_dir(synth, in_dir, 4, 1500, used, dev, seed=830, chimera_frac=0.05)
    os.makedirs(os.path.join(in_dir, "sub"))
    os.rename(os.path.join(in_dir, "synth_0003.fastq.gz"), os.path.join(in_dir, "sub", "synth_0003.fastq.gz"))
    kw = dict(max_ed=1, n_workers=3, reads_per_chunk=700, whitelist_keys=keys, pattern=run_files.FASTQ_PATTERN)
    d0 = str(tmp_path / "default")
    a = run_files.run(gpu_ctx, in_dir, d0, **kw)
    assert a["files"] == 4 and a["reads"] == n                                  # the sub-directory's file is found (recursive by default)
    rows = [ln.split("\t") for ln in open(os.path.join(d0, "BarcodesAssigned.tsv")).read().split("\n")[1:] if ln]
    code = {"A": 0, "G": 1, "C": 2, "T": 3}
    listed = np.array(sorted(sum(code[c] << (2 * (15 - i)) for i, c in enumerate(r_[0])) for r_ in rows), dtype=np.uint64)
    assert listed.size == a["used_list"]
    # -g with the list pass 1 arrived at: the same records, without the rank field
    d1 = str(tmp_path / "given")
    b = run_files.run(gpu_ctx, in_dir, d1, used_keys=listed, **kw)
    assert b["reads"] == n and b["used_list"] == listed.size and b["assigned"] == a["assigned"] and not os.path.exists(os.path.join(d1, "BarcodeList.tsv"))
    rk = re.compile(rb"_rk=[0-9]+")
    for sub, suf in (("passed", "_passed.fastq.gz"), ("failed", "_failed.fastq.gz")):
        for fi in range(4):
            x = gzip.open(os.path.join(d0, sub, f"synth_{fi:04d}{suf}")).read()
            y = gzip.open(os.path.join(d1, sub, f"synth_{fi:04d}{suf}")).read()
            assert b"_rk=" not in y and rk.sub(b"", x) == y, (sub, fi)
    assert open(os.path.join(d1, "BarcodesAssigned.tsv")).read() == open(os.path.join(d0, "BarcodesAssigned.tsv")).read()
    assert open(os.path.join(d1, "ReadScanner.tsv")).read() == open(os.path.join(d0, "ReadScanner.tsv")).read()
    # -s: statistics and tables only
    d2 = str(tmp_path / "dontwrite")
    c = run_files.run(gpu_ctx, in_dir, d2, write_fastqs=False, **kw)
    assert c["assigned"] == a["assigned"] and sorted(os.listdir(d2)) == ["BarcodeList.tsv", "BarcodesAssigned.tsv", "ReadScanner.html", "ReadScanner.tsv", "stats.tsv"]
    for nm in ("BarcodeList.tsv", "BarcodesAssigned.tsv", "ReadScanner.tsv"):
        assert open(os.path.join(d2, nm)).read() == open(os.path.join(d0, nm)).read(), nm
    # -u: the same records by name, every trimmed read a piece of the untrimmed one
    d3 = str(tmp_path / "trim")
    run_files.run(gpu_ctx, in_dir, d3, trim_fastq=True, **kw)
    x = gzip.open(os.path.join(d0, "passed", "synth_0001_passed.fastq.gz")).read().split(b"\n")
    y = gzip.open(os.path.join(d3, "passed", "synth_0001_passed.fastq.gz")).read().split(b"\n")
    assert x[0::4] == y[0::4] and len(x) > 1000
    shorter = 0
    for full, cut, fq, cq in zip(x[1::4], y[1::4], x[3::4], y[3::4]):
        assert cut in full and len(cq) == len(cut) and cq in fq
        shorter += len(cut) < len(full)
    assert shorter > 0.5 * (len(x) // 4)
    # file selection: -n, -k / -z in file-name order, -v
    e = run_files.run(gpu_ctx, in_dir, str(tmp_path / "flat"), recursive=False, **kw)
    assert e["files"] == 3 and sorted(os.listdir(str(tmp_path / "flat" / "passed"))) == [f"synth_{k:04d}_passed.fastq.gz" for k in range(3)]
    f_ = run_files.run(gpu_ctx, in_dir, str(tmp_path / "some"), skip_files=1, only_files=2, **kw)
    assert f_["files"] == 2 and sorted(os.listdir(str(tmp_path / "some" / "passed"))) == [f"synth_{k:04d}_passed.fastq.gz" for k in (1, 2)]
    g = run_files.run(gpu_ctx, in_dir, str(tmp_path / "pat"), **dict(kw, pattern=r".*/sub/.*\.fastq\.gz"))
    assert g["files"] == 1 and os.listdir(str(tmp_path / "pat" / "passed")) == ["synth_0003_passed.fastq.gz"]


def test_run_without_a_list_of_possible_barcodes(pkg, synth, gpu_ctx, tmp_path):
    """`-a none`: pass 1 counts every barcode it cuts (key lists per worker thread, sorted and counted on the device), finalize works on that
    table, `BarcodeList.tsv` is the no-whitelist form; against the same steps called one by one (chunk keys -> numpy counts -> finalize) and
    against the run WITH the whitelist, whose used list it must contain"""
    from sicelore_amd import lib as libmod

    run_files = importlib.import_module("sicelore_amd.run_files")
    dev = torch.device("cuda", gpu_ctx.device)
    wl = synth.make_whitelist(60_000, seed=841, device=dev)
    used = synth.pick_used(wl, 40, seed=842)
    keys = np.sort(wl.cpu().numpy().astype(np.uint64))
    in_dir = str(tmp_path / "in")
    n = run_files.write_synthetic_dir(synth, in_dir, 3, 4000, used, dev, seed=850, chimera_frac=0.05)
    kw = dict(max_ed=1, n_workers=3, reads_per_chunk=900)
    a = run_files.run(gpu_ctx, in_dir, str(tmp_path / "wl"), whitelist_keys=keys, **kw)
    b = run_files.run(gpu_ctx, in_dir, str(tmp_path / "nowl"), whitelist_keys=None, **kw)
    assert b["reads"] == n == a["reads"]
    # the steps one by one
    texts = [gzip.open(os.path.join(in_dir, f)).read() for f in sorted(os.listdir(in_dir))]
    d_keys = torch.zeros(n + 8, dtype=torch.int64, device=dev)
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)
    at = 0
    for t in texts:
        assert 3000 < gpu_ctx.scanfastq_pass1_chunk_keys(t, d_keys[at:], d_count) <= 4000      # (records: 5 % of the reads were joined in pairs)
        at += int(d_count.item())
        d_count.zero_()
    hk, hc = np.unique(d_keys[:at].cpu().numpy().view(np.uint64), return_counts=True)
    record_count = 3                                                       # under 4,000 records per file: one 10,000-read chunk each
    k, c, r = libmod.finalize_used_list(hk, hc.astype(np.uint32), record_count, 1, 10, 500)
    assert b["used_list"] == k.size
    assert open(os.path.join(str(tmp_path / "nowl"), "BarcodeList.tsv")).read() == libmod.barcode_list_tsv(hk, hc.astype(np.uint32), record_count, 1, no_whitelist=True)
    # without the list more barcodes are counted (every erroneous one too), and what the list-guided pass 1 kept is among them
    wl_rows = {ln.split("\t")[0] for ln in open(os.path.join(str(tmp_path / "wl"), "BarcodesAssigned.tsv")).read().split("\n")[1:] if ln}
    nowl_rows = {ln.split("\t")[0] for ln in open(os.path.join(str(tmp_path / "nowl"), "BarcodesAssigned.tsv")).read().split("\n")[1:] if ln}
    assert hk.size > 2 * a["used_list"] and len(wl_rows & nowl_rows) >= 0.9 * len(wl_rows)
    assert b["assigned"] > 0.8 * a["assigned"] and len(os.listdir(str(tmp_path / "nowl" / "passed"))) == 3


def test_other_polya_window_through_the_chunk_workers(pkg, synth, sor, gpu_ctx, tmp_path):
    """-p / -f / -w (smi_ctx_set_polya): the text worker and the packed worker of pass 2 with another polyA window write the records the oracle
    writes with the same parameters (scan + barcode + names), and the context goes back to the shipped window afterwards"""
    from test_write_gpu import _fastq, _oracle_records

    wl = synth.make_whitelist(20_000, seed=861)
    used = synth.pick_used(wl, 100, seed=862)
    reads = synth.gen_reads(400, used, seed=863, n_rate=0.002)
    seqs, quals = zip(*(synth.materialize(reads, i) for i in range(400)))
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    text = _fastq(list(seqs), list(quals))
    base_p, base_f, _ = gpu_ctx.scanfastq_pass2_chunk(text, split_chimeras=False)
    base_p = bytes(base_p)
    differs = 0
    for polya in ((12, 0.8, 100), (20, 0.7, 140)):
        par = sor.default_scan_params()
        par["polya_len"], par["polya_frac"], par["window_polya"] = polya
        gpu_ctx.set_polya(*polya)
        try:
            got_p, got_f, _ = gpu_ctx.scanfastq_pass2_chunk(text, split_chimeras=False)
            got_p, got_f = bytes(got_p), bytes(got_f)
            pk_p, pk_f, _ = gpu_ctx.scanfastq_pass2_chunk(text, split_chimeras=False, packed=True, n_threads=2)
            assert bytes(pk_p) == got_p and bytes(pk_f) == got_f
        finally:
            gpu_ctx.set_polya()
        exp_p, exp_f, _n = _oracle_records(sor, sor.BarcodeSet(used.numpy()), list(seqs), list(quals), 1, {}, 1, split=False, scan_params=par)
        assert got_p == exp_p and got_f == exp_f
        differs += got_p != base_p
    assert differs >= 1                                        # (another window does change what is found)
    again, _f, _ = gpu_ctx.scanfastq_pass2_chunk(text, split_chimeras=False)
    assert bytes(again) == base_p

