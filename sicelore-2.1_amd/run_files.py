"""`scanfastq -d <dir> -o <dir> --bcEditDistance k --compress` file to file: what quickrun-2.1.sh:35 runs, on one GPU.

A directory of *.fastq / *.fastq.gz in, `<out>/passed/<base>_passed.fastq.gz` and `<out>/failed/<base>_failed.fastq.gz` per input file
(FastqWriterThreadPool.java:L242-257), `BarcodeList.tsv` and `BarcodesAssigned.tsv` out; both passes of the default flow:

  per file: inflate (zlib on a host thread; from 1024 files on a share of the files by K-INFLATE on the device)  ->  pass 1 of its chunks on
  a worker lane (text worker, shared histogram)  ->  finalize / rank (host)  ->  pass 2 per chunk on the lanes (text worker: the records
  are written in HBM and K-DEFLATE turns `passed` and `failed` into one gzip member each there; gz="zlib": packed boundary, records
  written by host threads, zlib on the worker threads)  ->  members appended in chunk order, TSVs, statistics.

The reference parallelises over input files (README.md:155) with one JVM; here a pool of host threads takes chunks, each thread owning
one worker lane of the GPU (smi_ctx_create_lane) -- zlib and the native workers release the GIL, so the pool scales like the
reference's thread pool does.  Read ids (GET_NEXT_READID, a process-wide counter in the reference, so their order is not reproducible
there) are given per chunk from the number of records in front of it.  Nothing here computes on the CPU what the device path computes.
"""
import os
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import lib as _lib


def _inflate(path, pinned=False):
    """-> (uint8 array of the file's text, owner).  pinned: the text lands in page-locked memory (smi_host_alloc), from which the chunk
    workers upload at link speed and side by side on several lanes (pageable memory goes through the runtime's one staging path);
    owner.close() frees it"""
    raw = np.fromfile(path, dtype=np.uint8)
    if not path.endswith(".gz"):
        if not pinned:
            return raw, None
        pb = _lib.PinnedBuffer(max(raw.size, 1))
        pb.array[:raw.size] = raw
        return pb.array[:raw.size], pb
    if not pinned:
        return _lib.gz_inflate(raw), None

    def alloc(n):
        pb = _lib.PinnedBuffer(max(n, 1))
        return pb.array, pb

    return _lib.gz_inflate(raw, alloc=alloc)


FASTQ_PATTERN = r".{1,}\.fastq(\.gz)?"     # ParametersReadScannerApp$Files.fastqFilenamePattern (ParametersReadScannerApp.java:L203)


def out_base_ext(name):
    """(base, extension) of an input file's two outputs `<base>_passed.<extension>` / `<base>_failed.<extension>` as the reference's writer thread
    names them (FastqWriterThreadPool$FastQoneFileThread.init L242-250: commons-io getBaseName / getExtension, a final `gz` taken together with the
    extension in front of it): the INPUT's extension, whatever --compress says about the content"""
    base, dot, ext = name.rpartition(".")
    if not dot:
        base, ext = name, ""
    if ext.lower() == "gz":
        b2, dot2, e2 = base.rpartition(".")
        if not dot2:
            b2, e2 = base, ""
        base, ext = b2, e2 + "." + ext
    return base, ext


def check_output_names(files):
    """two inputs whose outputs would be the same two files (the same file name in two directories of a -d list or of a recursive walk): the
    reference opens both writers on one path and the chunks of the two inputs overwrite each other; here the run stops before it starts"""
    seen = {}
    for f in files:
        key = out_base_ext(os.path.basename(f))
        if key in seen:
            raise _lib.SmiError(f"input files {seen[key]} and {f} have the same name: both would be written to passed/{key[0]}_passed.{key[1]} and "
                                f"failed/{key[0]}_failed.{key[1]} (the reference's writer threads would overwrite each other's records); rename one of them or run the directories separately")
        seen[key] = f


def find_fastqs(in_dirs, recursive=True, pattern=FASTQ_PATTERN, skip=0, limit=None):
    """the input files as the reference finds them (FileTools.getInfiles, FileTools.java:L37-59 + FoundFiles.initialize L81-82): `in_dirs` is a
    comma-separated list of directories, each walked (all levels, or only the directory itself with -n; links followed), the regular files of
    all of them ordered by FILE NAME (Path.getFileName().compareTo: the directory plays no part; equal names keep the walk's order, which is
    the file system's there and the sorted one here), kept when the WHOLE path matches `pattern` (String.matches; -v), then the first `skip`
    dropped (-k) and at most `limit` taken (-z) -> list of paths"""
    import re

    found = []
    for d in str(in_dirs).split(","):
        if not d:
            continue
        if not os.path.isdir(d):
            raise _lib.SmiError(f"Error did not find input files: {d}")
        if recursive:
            for root, dirs, names in os.walk(d, followlinks=True):
                dirs.sort()
                found += [os.path.join(root, n) for n in sorted(names)]
        else:
            found += [os.path.join(d, n) for n in sorted(os.listdir(d))]
    found = [f for f in found if os.path.isfile(f)]
    found.sort(key=os.path.basename)                      # (stable: equal names stay in walk order)
    rx = re.compile(pattern) if pattern is not None else None
    found = [f for f in found if rx is None or rx.fullmatch(f)]
    found = found[int(skip):]
    return found if limit is None else found[:int(limit)]


def write_synthetic_dir(synth, out_dir, n_files, reads_per_file, used, device, seed=9000, chimera_frac=0.05, gz_level=1, pool=None, q_lo=35, q_hi=64):
    """test / bench input: n_files `*.fastq.gz` of reads_per_file synthetic reads each (generator of synth.py), qualities drawn uniformly
    per base (constant qualities would flatter every gzip step) -> total reads"""
    os.makedirs(out_dir, exist_ok=True)
    own = pool is None
    pool = pool or ThreadPoolExecutor(16)
    futs, total = [], 0
    for fi in range(n_files):
        rd = synth.gen_reads(reads_per_file, used, seed=seed + fi, device=device, q_mean=20.0)
        text, _b, offs = synth.fastq_text_device(rd, chimera_frac=chimera_frac, seed=seed + 7 * fi)
        g = torch.Generator(device=text.device)
        g.manual_seed(seed + 13 * fi)
        is_q = text == ord("I")          # the generator's quality placeholder (no base or header character is 'I')
        text[is_q] = torch.randint(q_lo, q_hi, (int(is_q.sum()),), device=text.device, generator=g, dtype=torch.int32).to(torch.uint8)
        total += int(offs.numel()) - 1
        data = text.cpu().numpy()
        del rd, text, _b, is_q
        futs.append(pool.submit(lambda d=data, k=fi: open(os.path.join(out_dir, f"synth_{k:04d}.fastq.gz"), "wb").write(_gzip_member(memoryview(d), gz_level))))
    for f in futs:
        f.result()
    if own:
        pool.shutdown()
    return total


def stats_html(tsv_text, command_line=None, assigned_tsv=None):
    """ReadScanner.html (/root/reference/README.md:388, sicelore-nf/main.nf:24 declares it an output of the scan step): ONE static page
    over the numbers ReadScanner.tsv holds -- the rows of ReadFlags.print (description, reads, percent, of what) as a table with a bar per row --
    and, when BarcodesAssigned.tsv was written, the assigned reads by edit distance.  The reference renders the same counters through a Velocity
    template with Google charts loaded from the network (ParseStatsHtmlPrinter.java:L311-326); this page is self-contained (no script, no
    network), its layout is this build's own; the QV histograms of QVstats and stats.pojo (Java serialisation) are not built."""
    import html

    rows = [ln.split("\t") for ln in tsv_text.split("\n") if ln.strip()]
    most = max([int(r[1]) for r in rows if len(r) > 1 and r[1].isdigit()] or [1]) or 1
    out = ["<!DOCTYPE html>", "<html><head><meta charset=\"utf-8\"><title>Nanopore read scan stats</title>",
           "<style>body{font-family:sans-serif;margin:2em}table{border-collapse:collapse}td,th{padding:2px 10px;border-bottom:1px solid #ddd;text-align:left}"
           "td.n{text-align:right;font-variant-numeric:tabular-nums}div.bar{background:#4a7ab8;height:0.8em}</style></head><body>",
           "<h1>Nanopore read scan stats</h1>"]
    if command_line:
        out.append("<p><b>Command line:</b> <code>" + html.escape(command_line) + "</code></p>")
    out.append("<table><tr><th></th><th>n reads</th><th>percent</th><th></th><th></th></tr>")
    for r in rows:
        if len(r) == 1:        # a heading line of ReadFlags.print
            out.append(f"<tr><th colspan=\"5\">{html.escape(r[0].strip('= '))}</th></tr>")
            continue
        r = r + [""] * (4 - len(r))
        width = 300 * int(r[1]) // most if r[1].isdigit() else 0
        out.append(f"<tr><td>{html.escape(r[0])}</td><td class=\"n\">{html.escape(r[1])}</td><td class=\"n\">{html.escape(r[2])}</td><td>{html.escape(r[3])}</td>"
                   f"<td><div class=\"bar\" style=\"width:{width}px\"></div></td></tr>")
    out.append("</table>")
    if assigned_tsv:
        lines = [ln.split("\t") for ln in assigned_tsv.split("\n") if ln.strip()]
        if len(lines) > 1:
            head, body = lines[0], lines[1:]
            out.append(f"<h2>Barcodes assigned in pass 2: {len(body)} barcodes</h2><table><tr>" + "".join(f"<th>{html.escape(h)}</th>" for h in head[1:]) + "</tr><tr>")
            for c in range(1, len(head)):
                out.append(f"<td class=\"n\">{sum(int(b[c]) for b in body if len(b) > c and b[c].lstrip('-').isdigit())}</td>")
            out.append("</tr></table>")
    out.append("</body></html>")
    return "\n".join(out) + "\n"


def write_stats(out_dir, stats, command_line=None):
    tsv = _lib.scan_stats_tsv(stats)
    with open(os.path.join(out_dir, "ReadScanner.tsv"), "w") as f:
        f.write(tsv)
    with open(os.path.join(out_dir, "stats.tsv"), "w") as f:
        names = _lib.READ_FLAG_NAMES + ["sum_len_passed", "sum_len_failed", "n_reads_split"]
        f.write("".join(f"{nm}\t{int(v)}\n" for nm, v in zip(names, stats)))
    assigned = os.path.join(out_dir, "BarcodesAssigned.tsv")
    with open(os.path.join(out_dir, "ReadScanner.html"), "w") as f:
        f.write(stats_html(tsv, command_line, open(assigned).read() if os.path.isfile(assigned) else None))


def merge_stats(run_dirs, out_dir):
    """`mergestats` (Jar/config.xml:33, ReadFlags.mergeStats): the statistics of several scanfastq runs added up -> <out_dir>/ReadScanner.tsv, stats.tsv"""
    total = np.zeros(_lib.N_SCAN_STATS, dtype=np.uint64)
    for d in run_dirs:
        rows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(d, "stats.tsv"))]
        total += np.array([int(r[1]) for r in rows], dtype=np.uint64)
    os.makedirs(out_dir, exist_ok=True)
    write_stats(out_dir, total)
    return total


def _cut_chunks(text, reads_per_chunk):
    """byte ranges of `text` holding reads_per_chunk records each (4 lines per record)"""
    n = int(text.size)
    if n == 0:
        return []
    nl = np.flatnonzero(text == 10)
    n_lines = nl.size + (0 if text[-1] == 10 else 1)
    n_rec = n_lines // 4
    cuts, start = [], 0
    for r in range(reads_per_chunk, n_rec, reads_per_chunk):
        end = int(nl[4 * r - 1]) + 1
        cuts.append((start, end))
        start = end
    cuts.append((start, n))
    return cuts


def _gzip_member(data, level):
    c = zlib.compressobj(level, zlib.DEFLATED, 31)
    return c.compress(data) + c.flush()


def run(ctx, in_dir, out_dir, *args, polya=None, random_barcode_seed=0, **kw):
    """`_run` (below: everything about the run) with the polyA finder's parameters of `scanfastq -p <length> -f <fraction> -w <window>` set on the
    context for its duration (smi_ctx_set_polya; None / zeros: config.xml's 15 / 0.75 / 150); random_barcode_seed != 0: `scanfastq -e`, pass 2
    matches random sequences in place of the reads' barcode windows (smi_ctx_set_random_barcodes: what is still assigned is chance)"""
    if polya is not None:
        ctx.set_polya(*polya)
    if random_barcode_seed:
        ctx.set_random_barcodes(random_barcode_seed)
    try:
        return _run(ctx, in_dir, out_dir, *args, **kw)
    finally:
        if polya is not None:
            ctx.set_polya()
        if random_barcode_seed:
            ctx.set_random_barcodes(0)


def _run(ctx, in_dir, out_dir, max_ed=1, n_workers=16, reads_per_chunk=100_000, gz_level=6, whitelist_keys=None, five_prime=False,
        dont_search_polya=False, host_threads_per_call=1, compress=True, gz="device", pinned_text=False, inflate="host", device_share=0.25, group=None, resident_bytes=96 << 30,
        host_text_bytes=256 << 30, inflate_auto_from=1024, recursive=True, pattern=r".{1,}\.(fastq|fq)(\.gz)?", skip_files=0, only_files=None,
        used_keys=None, write_fastqs=True, trim_fastq=False, merge_ed=None, min_count_fold=10, cells_fold_below_max=500, command_line=None):
    """-> dict of counts and wall-clock times.  ctx: a Context (its lanes are created here); whitelist_keys: the possible barcodes
    (sorted uint64), loaded for pass 1.  gz: who deflates the output with --compress -- "device": the text worker writes the records in HBM
    and K-DEFLATE turns them into one gzip member per chunk there (dynamic Huffman, literals only: about 8 % larger files than zlib level 6);
    "zlib": the packed worker's host-written records through zlib at gz_level on the worker threads.
    resident_bytes: with gz="device" a file's text is uploaded ONCE, when it has been inflated, and stays in HBM for both passes (the chunk
    workers take device pointers) until this many bytes are held; files beyond that are uploaded per pass from the host as before.
    inflate: "host" (default) = the library's decoder on the worker threads; "device" = K-INFLATE for every *.gz file, 512 per round;
    "auto" = both, working one queue of files from its two ends, from inflate_auto_from files on (a round on the device costs the time of ONE
    file however many run in it).  Measured on 2,048 files of 4,000 reads (profiles/r03/f2f_many_files.json): the host alone 3.0 - 3.1 s for
    inflate + pass 1, host + device 4.2 - 4.3 s -- with the library's own decoder on sixteen cores the device's rounds no longer pay, they
    hold LDS that pass 1 wants; "auto" is kept for hosts with fewer cores per GPU.
    host_text_bytes: how much inflated text may wait in host memory between the passes; the text of files beyond that is dropped after pass 1
    and inflated again in pass 2 (the reference reads every file twice, NanoporeReadScannerMain.java:L306) -- a run of any size (the output side too: a chunk's members go to their file as soon as the file's earlier chunks are written).
    With torch.distributed initialised (one process per GPU) the directory's files are dealt to the ranks in contiguous runs; the only
    exchanges are the pass-1 histogram (one all-reduce, then the same finalize on every rank), the records in front of each rank (read ids),
    and the counters behind the two TSVs and the statistics, which rank 0 writes.  Every rank writes the output files of its own inputs; the
    result is the single-process run's, byte for byte.
    in_dir / recursive / pattern / skip_files / only_files: find_fastqs (-d, -n, -v, -k, -z; the command line passes the reference's own
    default pattern, a direct caller also gets *.fq).  used_keys (-g <file>, NanoporeReadScannerMain.java:L300-302): the barcodes in use are
    SUPPLIED -- pass 1 is skipped, the files are only inflated and counted, every barcode has rank 0 (no rk= in the names,
    WorkerReadscanner$BarcodesMapForBCfinding.getMapFromCellRangerData L437) and no BarcodeList.tsv is written (pass 1 writes it).
    whitelist_keys=None without used_keys (-a none, NanoporeReadScannerMain.java:L132-133): there is no list of possible barcodes -- pass 1 counts
    every barcode it cuts (lists of keys, sorted and counted on the device at the end of the pass), `BarcodeList.tsv` leaves out rows with AAAAA / TTTTT.
    write_fastqs=False (-s, L211-213): statistics and TSVs only.  trim_fastq (-u): records cut as FastqRecordExt.getRecordForWriting does
    with trimFastq."""
    import torch.distributed as dist

    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if multi else (0, 1)
    if gz not in ("device", "zlib"):
        raise ValueError("gz: 'device' or 'zlib'")
    if inflate not in ("auto", "device", "host"):
        raise ValueError("inflate: 'auto' (the device takes a share from 1024 *.gz files on: one wavefront per file is much slower than a host thread, a thousand at once are not), 'device' or 'host'")
    on_device = compress and gz == "device"
    t_all = time.perf_counter()
    files = find_fastqs(in_dir, recursive=recursive, pattern=pattern, skip=skip_files, limit=only_files)     # full paths
    if not files:
        raise _lib.SmiError(f"NO INPUT FILES FOUND in {in_dir}")
    if write_fastqs:
        check_output_names(files)          # over ALL files, before they are dealt to the ranks
    given = used_keys is not None
    nowl = not given and whitelist_keys is None      # -a none: no list of possible barcodes, every barcode seen in pass 1 is counted
    if multi:
        from . import distributed as _dist

        lo_f, hi_f = _dist.shard_range(len(files), rank, world)
        files = files[lo_f:hi_f]          # (a rank without files still takes part in the exchanges)
    os.makedirs(out_dir, exist_ok=True)
    if write_fastqs:
        os.makedirs(os.path.join(out_dir, "passed"), exist_ok=True)
        os.makedirs(os.path.join(out_dir, "failed"), exist_ok=True)
    pool = ThreadPoolExecutor(n_workers)
    lanes = [ctx] + [ctx.lane() for _ in range(n_workers)]     # one per worker thread, and one for the thread that drives K-INFLATE
    free = list(range(n_workers + 1))

    def with_lane(fn):
        def call(*a):
            k = free.pop()
            try:
                return fn(lanes[k], *a)
            finally:
                free.append(k)
        return call

    dev = torch.device("cuda", ctx.device)
    keys = (np.unique(np.ascontiguousarray(used_keys, dtype=np.uint64)) if given else
            np.zeros(0, dtype=np.uint64) if nowl else np.ascontiguousarray(whitelist_keys, dtype=np.uint64))
    # ---- inflate + pass 1: a file's chunks go to the device as soon as the file is inflated (the host inflates the next one meanwhile) -----
    t0 = time.perf_counter()
    import threading

    set_ready = threading.Event()
    hist = torch.zeros(keys.size, dtype=torch.int32, device=dev)
    key_lists, key_bufs, key_lock = threading.local(), [], threading.Lock()      # -a none: per worker thread a list of the barcodes its chunks left

    def load_set():
        """the possible barcodes on the device (the membership pyramid: ~ 10 ms for 3.6 M of them); the worker threads inflate their first files meanwhile"""
        try:
            if not given and not nowl:          # (a supplied list is loaded once, as the used list, after the files are read; -a none has none)
                ctx.set_barcode_set(keys, mode=_lib.SET_MEMBERSHIP)      # pass 1 asks for membership only: none of the matchers' structures is built
                for ln in lanes[1:]:
                    ln.refresh()
                torch.cuda.synchronize()
        finally:
            set_ready.set()                      # (on an error the workers' calls fail instead of waiting for ever)

    cpu_inflate = [0.0] * len(files)
    owners = [None] * len(files)

    def p1(lane, text, rng):
        if given:                               # -g: no pass 1, only the chunk's record count (lines / 4, as the chunk cutter counts them)
            part = text[rng[0]:rng[1]]
            if part.numel() if hasattr(part, "numel") else part.size:
                lines = int((part == 10).sum()) + (0 if int(part[-1]) == 10 else 1)
            else:
                lines = 0
            return lines // 4
        if nowl:                                # -a none: the chunk's barcodes as keys onto this thread's list (counted after the pass)
            buf = getattr(key_lists, "cur", None)
            if buf is None or buf[2] + reads_per_chunk > buf[0].numel():
                buf = [torch.empty(max(4 * reads_per_chunk, 1 << 20), dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), 0]
                key_lists.cur = buf
                with key_lock:
                    key_bufs.append(buf)
            n_ = lane.scanfastq_pass1_chunk_keys(text[rng[0]:rng[1]], buf[0][buf[2]:], buf[1], five_prime=five_prime, dont_search_polya=dont_search_polya)
            got = int(buf[1].item())
            buf[1].zero_()
            if got > buf[0].numel() - buf[2]:
                raise _lib.SmiError("pass 1 (-a none): a chunk held more records than reads_per_chunk")
            buf[2] += got
            return n_
        # the text worker: index, planes, scan and histogram all on the device, so the host's threads stay with the inflating
        set_ready.wait()
        return lane.scanfastq_pass1_chunk(text[rng[0]:rng[1]], hist, five_prime=five_prime, dont_search_polya=dont_search_polya, packed=False)

    held, held_host = [0], [0]

    staging = threading.local()          # one page-locked buffer per worker thread, reused from file to file
    staged = []

    def side_stream():
        """the calling thread's own torch stream: uploads and the line-end search must not run on the default stream, which waits for every
        other stream of the device -- a K-INFLATE round (a second) included"""
        st = getattr(staging, "stream", None)
        if st is None:
            st = staging.stream = torch.cuda.Stream(dev)
        return st

    def stage_alloc(n):
        """where a file that will live in HBM is inflated to: the thread's page-locked buffer (uploads from it run at link speed and side by side
        on all threads; pageable memory goes through the runtime's one staging path)"""
        pb = getattr(staging, "pb", None)
        if pb is None or pb.array.size < n:
            # (an outgrown buffer is freed with the others at the end of pass 1: gz_inflate still copies out of it when it grows)
            pb = _lib.PinnedBuffer(max(int(n * 1.25), 1 << 20))
            staging.pb = pb
            staged.append(pb)
        return pb.array[:n], None

    def load_and_count(fi):
        t1 = time.perf_counter()
        path = files[fi]
        resident = on_device and held[0] + 3 * os.path.getsize(path) <= resident_bytes     # (a soft limit: the threads race for its last bytes)
        if resident and path.endswith(".gz"):
            t, owner = _lib.gz_inflate(np.fromfile(path, dtype=np.uint8), alloc=stage_alloc)
        else:
            t, owner = _inflate(path, pinned=pinned_text)
        owners[fi] = owner
        cpu_inflate[fi] = time.perf_counter() - t1
        if resident:
            held[0] += t.size
            with torch.cuda.stream(side_stream()):
                td = torch.from_numpy(t).to(dev)              # one upload; both passes read the text from HBM
                cuts = device_cuts(td)
                side_stream().synchronize()
            if owner is not None:
                owner.close()
                owners[fi] = None
            return td, cuts, [with_lane(p1)(td, rng) for rng in cuts]
        cuts = _cut_chunks(t, reads_per_chunk)
        recs_ = [with_lane(p1)(t, rng) for rng in cuts]
        if held_host[0] + t.size > host_text_bytes:           # no room to keep it: pass 2 inflates the file again
            if owner is not None:
                owner.close()
                owners[fi] = None
            return None, cuts, recs_
        held_host[0] += t.size
        return t, cuts, recs_

    def device_cuts(t):
        """byte ranges of reads_per_chunk records each in a text that is on the device"""
        n = int(t.numel())
        if n == 0:
            return []
        nl = torch.nonzero(t == 10).flatten()
        n_lines = int(nl.numel()) + (0 if int(t[-1]) == 10 else 1)
        n_rec_ = n_lines // 4
        if n_rec_ <= reads_per_chunk:
            return [(0, n)]
        idx = torch.arange(reads_per_chunk, n_rec_, reads_per_chunk, device=t.device) * 4 - 1
        ends = (nl[idx] + 1).cpu().tolist() if idx.numel() else []
        return [(a, b) for a, b in zip([0] + ends, ends + [n])]

    n_on_device = 0
    dev_detail = []          # per K-INFLATE round: files, seconds waiting for the files to be read, seconds of upload + kernel + results
    use_device = on_device and (inflate == "device" or (inflate == "auto" and sum(f.endswith(".gz") for f in files) >= inflate_auto_from))  # (the packed worker wants host text)
    if use_device:
        # K-INFLATE beside the host, both working one queue of *.gz files from its two ends: the host's worker threads take files from the
        # front, one after the other; the device takes up to 512 from the back per round (one wavefront per file; a file is as fast as any
        # other however many run, so a round costs the time of its largest file), round after round until the queue is empty.  The
        # device's texts stay in HBM and the chunk workers take them from there.  A file the kernel hands back (unusual or damaged) goes
        # through the host's decoder like the others.  inflate="device": the host's threads leave the *.gz files to the device.
        import collections

        todo = collections.deque(fi for fi, f in enumerate(files) if f.endswith(".gz"))
        plain = [fi for fi, f in enumerate(files) if not f.endswith(".gz")]
        todo_lock = threading.Lock()
        results = {}
        t_dev, n_rounds = [0.0], [0]
        per_round = max(1, min(512, int(len(todo) * device_share))) if inflate == "auto" else 512

        def count_dev(fi, t):
            with torch.cuda.stream(side_stream()):
                cuts = device_cuts(t)
                side_stream().synchronize()
            return t, cuts, [with_lane(p1)(t, rng) for rng in cuts]

        def device_part():
            k_lane = free.pop()               # a lane of its own while the host's threads work with the others
            pending = []
            slots = [None, None]              # two page-locked buffers: one is read into while the device works on the other

            def take():
                # a round costs the device the time of one file (about what the host's threads need for 1.5 rounds' worth of files): near the
                # end of the queue the rest is left to the host, or its threads would idle while the device finishes
                with todo_lock:
                    if inflate == "auto" and len(todo) < int(2.5 * per_round) and inflate_auto_from > 8:
                        return []
                    return [todo.pop() for _ in range(min(per_round, len(todo)))]

            def prepare(mine, k):
                paths = [files[fi] for fi in mine]
                need = sum(((os.path.getsize(p_) + 511) & ~511) for p_ in paths) + 1024
                if slots[k] is None or slots[k].array.size < need:
                    if slots[k] is not None:
                        slots[k].close()
                    slots[k] = _lib.PinnedBuffer(int(need * 1.25))
                return _lib.pack_gz_paths(paths, buffer=slots[k].array)

            try:
                with ThreadPoolExecutor(1) as prefetch:
                    mine, k = take(), 0
                    nxt = prefetch.submit(prepare, mine, k) if mine else None
                    while mine:
                        t1 = time.perf_counter()
                        packed = nxt.result()
                        t2 = time.perf_counter()
                        following = take()            # the next round is read and laid out while this one runs on the device
                        nxt = prefetch.submit(prepare, following, 1 - k) if following else None
                        with torch.cuda.stream(side_stream()):
                            d_out, offs, lens, status, _ = lanes[k_lane].gz_inflate_device(packed=packed)
                        t_dev[0] += time.perf_counter() - t1
                        dev_detail.append((len(mine), round(t2 - t1, 3), round(time.perf_counter() - t2, 3)))
                        n_rounds[0] += 1
                        for j, fi in enumerate(mine):
                            if int(status[j]) == 0:
                                pending.append((fi, pool.submit(count_dev, fi, d_out[int(offs[j]):int(offs[j]) + int(lens[j])]), True))
                            else:
                                pending.append((fi, pool.submit(load_and_count, fi), False))
                        mine, k = following, 1 - k
            finally:
                free.append(k_lane)
                for pb in slots:
                    if pb is not None:
                        pb.close()
            return pending

        def host_part():
            """one of the host's worker threads: files from the front of the queue until it is empty"""
            while True:
                with todo_lock:
                    if not todo:
                        return
                    fi = todo.popleft()
                results[fi] = load_and_count(fi)

        dev_future = ThreadPoolExecutor(1).submit(device_part)
        host_futs = [pool.submit(host_part) for _ in range(max(n_workers - 2, 1))] if inflate == "auto" else []
        plain_futs = [(fi, pool.submit(load_and_count, fi)) for fi in plain]
        load_set()
        for f in host_futs:
            f.result()
        for fi, f in plain_futs:
            results[fi] = f.result()
        for fi, f, on_dev_ in dev_future.result():
            results[fi] = f.result()
            n_on_device += 1 if on_dev_ else 0
        loaded = [results[fi] for fi in range(len(files))]
        if files:
            cpu_inflate[0] += t_dev[0]
    else:
        futs = [pool.submit(load_and_count, fi) for fi in range(len(files))]
        load_set()
        loaded = [f.result() for f in futs]
    torch.cuda.synchronize()
    for pb in staged:
        pb.close()
    texts = [t for t, _, _ in loaded]
    chunks, n_rec = [], []  # (file index, chunk index in file, byte range); records per chunk
    for fi, (_, cuts, recs) in enumerate(loaded):
        for ci, rng in enumerate(cuts):
            chunks.append((fi, ci, rng))
        n_rec += recs
    t_pass1 = time.perf_counter() - t0
    t_inflate = float(sum(cpu_inflate))
    # ---- finalize ----------------------------------------------------------------------------------------------------------------------
    t0 = time.perf_counter()
    # the reference's recordCount: the 10,000-read chunks FastqFileReader cuts, per input file (UsedCellBCListGenerator.java:L254)
    per_file = {}
    for (fi, _, _), m in zip(chunks, n_rec):
        per_file[fi] = per_file.get(fi, 0) + m
    record_count = sum((m + 9_999) // 10_000 for m in per_file.values())
    ids_in_front = 0
    if multi:
        # one all-reduce of the dense histogram (RCCL on the device tensor), the chunk counts with it; records in front of this rank's files
        if not given and not nowl:
            hist, record_count = _dist.allreduce_histogram(hist, record_count, group)
        xdev = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")   # (gloo in the tests: small tensors on the host)
        mine = torch.tensor([int(sum(n_rec))], dtype=torch.int64, device=xdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine, group=group)
        ids_in_front = int(sum(int(t.item()) for t in every[:rank]))
    if given:
        # the supplied list IS the search set: counts and ranks 0 (CountsRank(0, 0)); the reference runs no collision merge on it
        k, rk_keys, rk_vals = keys, keys, None
    else:
        if nowl:
            # every barcode seen, sorted and counted on the device; the ranks' tables added up on the host
            parts = [b[0][:b[2]] for b in key_bufs if b[2]]
            all_keys = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
            hk, hc = ctx.count_keys_device(all_keys, int(all_keys.numel()))
            del all_keys, parts
            key_bufs.clear()
            if multi:
                every = [None] * world
                dist.all_gather_object(every, (hk, hc, record_count), group=group)
                cat_k, cat_c = np.concatenate([e[0] for e in every]), np.concatenate([e[1] for e in every]).astype(np.uint64)
                hk, inv = np.unique(cat_k, return_inverse=True)
                hc = np.bincount(inv, weights=cat_c.astype(np.float64), minlength=hk.size).astype(np.uint32)
                record_count = int(sum(e[2] for e in every))
        else:
            h = hist.cpu().numpy()
            nz = np.nonzero(h)[0]
            hk, hc = keys[nz], h[nz].astype(np.uint32)
        # config.xml: mergeBCsED (null = --bcEditDistance), minCountFold, cellsWithReadsnFoldBelowMaxToKeep (UsedCellBCListGenerator.java:L397-402,
        # BarcodeDatasetColissionTester.java:L68-229)
        m_ed = max_ed if merge_ed is None else int(merge_ed)
        k, c, r = _lib.finalize_used_list(hk, hc, record_count, m_ed, min_count_fold, cells_fold_below_max)
        if rank == 0:
            with open(os.path.join(out_dir, "BarcodeList.tsv"), "w") as f:
                f.write(_lib.barcode_list_tsv(hk, hc, record_count, m_ed, min_count_fold, cells_fold_below_max, no_whitelist=nowl))
        if nowl and k.size and int(k.max()) >> 32:
            # a 5' barcode that was cut with an N in it is a long with its upper half set (UsedCellBCListGenerator.java:L219; NOTES R5.10): it can
            # only ever equal a window with the same N, which the matcher does not probe -- such an entry stays in BarcodeList.tsv (written above), leaves the search
            # set, and can never be assigned, so BarcodesAssigned.tsv (rows with counts only) has no row for it either way
            keep = (k >> np.uint64(32)) == 0
            k, c, r = k[keep], c[keep], r[keep]
        order = np.argsort(k)
        rk_keys, rk_vals = k[order], r[order].astype(np.int32)
    ctx.set_barcode_set(k, mode=_lib.SET_WHITELIST if given else _lib.SET_USED_LIST)
    for ln in lanes[1:]:
        ln.refresh()
    t_finalize = time.perf_counter() - t0
    # ---- pass 2 + gzip -------------------------------------------------------------------------------------------------------------------
    t0 = time.perf_counter()
    first_id = np.concatenate([[0], np.cumsum(n_rec)])[:-1] + 1 + ids_in_front

    again, again_lock = {}, threading.Lock()     # file index -> [text, chunks still to come] for files whose text was dropped after pass 1
    chunks_of = {}
    for fi_, _, _ in chunks:
        chunks_of[fi_] = chunks_of.get(fi_, 0) + 1

    def text_of(fi):
        if texts[fi] is not None:
            return texts[fi]
        with again_lock:
            ent = again.get(fi)
            if ent is None:
                ent = again[fi] = [None, chunks_of[fi], threading.Lock()]
        with ent[2]:
            if ent[0] is None:
                ent[0] = _inflate(files[fi])[0]
        return ent[0]

    def text_done(fi):
        if texts[fi] is None:
            with again_lock:
                ent = again[fi]
                ent[1] -= 1
                if ent[1] == 0:
                    del again[fi]

    # The chunks of pass 2 run on the worker threads in (file, chunk) order; a chunk's members are written to its file's two outputs as soon as
    # every earlier chunk of that file has been written, and dropped -- at most a few chunks per worker wait in memory, whatever the size of
    # the run (the reference streams records to a writer thread per input file, FastqWriterThreadPool.java:L209).  The per-barcode counters
    # go into ONE table under a lock.
    counts = np.zeros((k.size, 3), dtype=np.int64)
    counts_lock = threading.Lock()
    out_state = {}                                 # file index -> [lock, next chunk to write, {chunk: (passed, failed)}, handles or None]
    for fi_ in chunks_of:
        out_state[fi_] = [threading.Lock(), 0, {}, None]

    def out_names(fi):
        base, ext = out_base_ext(os.path.basename(files[fi]))
        return os.path.join(out_dir, "passed", f"{base}_passed.{ext}"), os.path.join(out_dir, "failed", f"{base}_failed.{ext}")

    def deliver(fi, ci, zp, zf):
        if not write_fastqs:
            return
        st = out_state[fi]
        with st[0]:
            st[2][ci] = (zp, zf)
            while st[1] in st[2]:
                a, b = st[2].pop(st[1])
                if st[3] is None:
                    pn, fn = out_names(fi)
                    st[3] = (open(pn, "wb"), open(fn, "wb"))
                st[3][0].write(a)
                st[3][1].write(b)
                st[1] += 1
            if st[1] == chunks_of[fi] and st[3] is not None:
                st[3][0].close()
                st[3][1].close()
                st[3] = ()

    def p2(lane, j):
        fi, ci, rng = chunks[j]
        text_j = text_of(fi)[rng[0]:rng[1]]
        passed, failed, info = lane.scanfastq_pass2_chunk(text_j, max_ed=max_ed, five_prime=five_prime, dont_search_polya=dont_search_polya, trim_fastq=trim_fastq,
                                                          first_read_id=int(first_id[j]), rank_keys=None if given else rk_keys, rank_values=rk_vals, want_results=True, copy=False,
                                                          packed=not on_device, n_threads=host_threads_per_call, compress=on_device)
        # (advisor, round 5) read ids are dealt from the record counts of the chunks in front: with -g those come from a line count, here from K-FQ's
        # index -- a text on which the two disagree (stray blank lines) stops the run instead of shifting every later id
        if int(info["n_records_in"]) != int(n_rec[j]):
            raise _lib.SmiError(f"{files[fi]}, chunk {ci}: {int(info['n_records_in'])} FASTQ records indexed, {int(n_rec[j])} counted when the file was read")
        bc = info["bc"] if info["n_records_out"] else np.zeros(0, dtype=_lib.BC_RESULT_DTYPE)
        ok = bc["found"] == 1
        if ok.any():
            rows, eds = np.searchsorted(rk_keys, bc["bc"][ok].astype(np.uint64)), bc["ed"][ok].astype(np.int64)
            with counts_lock:
                np.add.at(counts, (rows, eds), 1)
        if compress and not on_device:
            zp, zf = _gzip_member(memoryview(passed), gz_level), _gzip_member(memoryview(failed), gz_level)
        else:
            # (gzip members already with gz="device")  the lane's buffers are reused by its next call: a copy, made by numpy, which lets
            # the other worker threads run meanwhile (bytes() would hold the interpreter lock for the whole memcpy)
            zp, zf = np.array(passed, dtype=np.uint8, copy=True), np.array(failed, dtype=np.uint8, copy=True)
        text_done(fi)
        n_z = len(zp) + len(zf)
        deliver(fi, ci, zp, zf)
        return n_z, int(info["n_records_out"]), int(info["n_passed"]), int(info["passed_text_bytes"]), int(info["failed_text_bytes"]), info.get("stats")

    results = list(pool.map(lambda j: with_lane(p2)(j), range(len(chunks))))
    t_pass2 = time.perf_counter() - t0
    # ---- files of inputs without a record (no chunk): empty outputs, as the reference's writer threads leave them -------------------------
    t0 = time.perf_counter()
    for fi in range(len(files)):
        if fi not in out_state and write_fastqs:
            for nm in out_names(fi):
                open(nm, "wb").close()
    stats = np.zeros(_lib.N_SCAN_STATS, dtype=np.uint64)
    for res in results:
        if res[5] is not None:
            stats += res[5]
    if multi:  # the counters behind BarcodesAssigned.tsv and the statistics, summed over the ranks
        xdev = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t_counts = torch.from_numpy(counts.astype(np.int64)).to(xdev)
        t_stats = torch.from_numpy(stats.astype(np.int64)).to(xdev)
        dist.all_reduce(t_counts, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(t_stats, op=dist.ReduceOp.SUM, group=group)
        counts, stats = t_counts.cpu().numpy(), t_stats.cpu().numpy().astype(np.uint64)
    if rank == 0:
        with open(os.path.join(out_dir, "BarcodesAssigned.tsv"), "w") as f:
            f.write(_lib.assigned_tsv(rk_keys, counts.astype(np.uint32), max_ed=max_ed))
        # the counters ReadScanner.html renders (ReadFlags.print) and, beside them, the raw vector `merge_stats` adds up (the reference keeps
        # them in stats.pojo for its `mergestats` sub-command)
        write_stats(out_dir, stats, command_line)
    t_write = time.perf_counter() - t0
    for ln in lanes[1:]:
        ln.close()
    pool.shutdown()
    for o in owners:
        if o is not None:
            o.close()
    n_reads = int(sum(n_rec))
    wall = time.perf_counter() - t_all
    return {"rank": rank, "ranks": world, "files": len(files), "chunks": len(chunks), "reads": n_reads, "records_out": sum(r_[1] for r_ in results), "passed": sum(r_[2] for r_ in results),
            "assigned": int(counts.sum()), "used_list": int(k.size), "text_in_bytes": int(sum(rng[1] - rng[0] for _, _, rng in chunks)), "files_inflated_twice": sum(1 for t in texts if t is None), "files_inflated_on_device": n_on_device, "device_inflate_rounds": dev_detail, "text_resident_bytes": int(held[0]),
            "text_out_bytes": sum(r_[3] + r_[4] for r_ in results), "gz_out_bytes": sum(r_[0] for r_ in results) if compress else None,
            "wall_s": wall, "reads_per_s": n_reads / wall, "inflate_and_pass1_s": t_pass1, "inflate_thread_seconds": t_inflate, "finalize_s": t_finalize, "pass2_and_gzip_s": t_pass2,
            "write_files_s": t_write, "workers": n_workers, "gz": (gz if compress else None), "gz_level": gz_level if compress and not on_device else None}
