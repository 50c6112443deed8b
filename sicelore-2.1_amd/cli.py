"""The jar's command line in front of the library.

    java -jar [-Xmx..] NanoporeBC_UMI_finder-2.1.jar scanfastq  -d <dir[,dir..]> -o <dir> --bcEditDistance k [--compress] [--ncpu N] [-h] [-y] [-a file]
                                                                [-g usedBarcodes] [-n] [-v regex] [-k skip] [-z only] [-s] [-u] [-p len] [-f frac] [-w window]   (--polyAlength --frac-f --windowAT)
    java -jar [-Xmx..] NanoporeBC_UMI_finder-2.1.jar assignumis --inFileNanopore <bam> -o <bam> [--annotationFile refFlat] [-v n] [-p] [-w] [-b ed] [-u ed] [-s]

become   python sicelore-2.1_amd scanfastq ... / assignumis ...   (the directory is runnable: __main__.py; a `java` wrapper that drops
`-jar`, `-Xmx..` and the jar's name makes /root/reference/quickrun-2.1.sh:35,42 run unchanged, tests/test_cli_gpu.py does exactly that).

Reference units: option tables NanoporeReadScannerMain.cli_otions (NanoporeReadScannerMain.java:L336-469) and UmiFinderMain (L298-447);
config discovery OneProgramMainBase.checkCfgFilePath (cwd, then the application's directory; -c names a file for assignumis);
exit code 0, or 1 after a message (WorkerReadscanner.java:L376-378).

config.xml (round 6): the file's knobs are taken at RUN TIME -- thresholds, windows, adapter / complete TSO sequences and mismatch limits, the
finalize folds, mergeBCsED, umi_length and the clustering distances go to the library as smi_run_knobs / call arguments (lib.KNOB_FIELDS,
HOST_KNOBS_* below).  Only what is genuinely compiled in (COMPILED_IN: testPlusMinusPos, cell_bc_length, the read-name
grammar) is CHECKED against the file, and a different value stops the run with the knob's name -- nothing is silently ignored.  Options the
product has no path for (Illumina-guided modes, the file watcher) are refused by name.
"""
import gzip
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# config.xml, round 6: the knobs that shape the per-read algorithms are RUN-TIME parameters of the library (smi_run_knobs,
# lib.KNOB_FIELDS) or of its host-side steps (HOST_KNOBS below); a value the library has no kernel for is refused BY THE LIBRARY with the knob's
# name (smi_ctx_set_knobs: sequence lengths, 8 <= umi_length <= 12, ...).
#
# COMPILED_IN: what is genuinely fixed in this build -- knob (path under <Parameters>) -> the shipped value the kernels are compiled for; any other
# value stops the run with the knob's name (nothing is silently ignored):
#   * testPlusMinusPos (five windows per read: the layout of smi_bc_window and of K-BC1's filter), cell_bc_length (32-bit keys)
#   * the read-name grammar (prefixes, nbasesOfAdapterSeqInReadname: smi_name.h and K-UPARSE), runningasdemon, tagGeneNameFunction
COMPILED_IN = {
    "readscanner/testPlusMinusPos": "2", "readscanner/pa_start_prefix": "PS=", "readscanner/pa_end_prefix": "PE=", "readscanner/adapter_pos_prefix": "AE=",
    "readscanner/tso_pos_prefix": "T=", "readscanner/seq_prefix": "X=", "readscanner/qv_prefix": "Q=",
    "readscanner/nbasesOfAdapterSeqInReadname": "3", "readscanner/runningasdemon": "false",
    "barcodeUMIFinder/tagGeneNameFunction": "DefaultTagger",
    "barcodes/cell_bc_length": "16",
}
# knobs of the shipped file that NO unit of the path reads (checked over the bytecode: no getfield outside print() / the Illumina-guided analyzers):
# accepted with any value, as the reference accepts them.  threeprimeadapter_for5pBarcoding's `sequence` / maxNeedlemanMismatches: only its
# sequence_complete is used (ChimeraFindernew.java:L75); tso_for5pBarcoding: the 5' analyzer has no TSO scan; offsetTSOend, internalMinPolyATlengthForReporting:
# never read; maxComplexityForUMIclustering / pregroup_for_clustering_threshold: the pre-grouping arm is dead code
NO_EFFECT = ("threeprimeadapter_for5pBarcoding/sequence", "threeprimeadapter_for5pBarcoding/maxNeedlemanMismatches", "tso_for5pBarcoding/sequence",
             "tso_for5pBarcoding/maxNeedlemanMismatches", "tso_for3pBarcoding/offsetTSOend", "polyAT/internalMinPolyATlengthForReporting",
             "umis/pregroup_for_clustering_threshold")
# host-side knobs: name -> (keyword of run_files.run / assignumis_stream, converter)
HOST_KNOBS_SCAN = {"readscanner/minCountFold": ("min_count_fold", int), "readscanner/cellsWithReadsnFoldBelowMaxToKeep": ("cells_fold_below_max", int)}
HOST_KNOBS_UMI = {"umis/umi_length": ("umi_length", int), "umis/umi_completelinkclusteringED": ("complete_link_ed", int),
                  "umis/umi_singlelinkclusteringED": ("single_link_ed", int),
                  "umis/complexity_threshold_for_switch_to_single_link_clustering": ("single_link_switch", int),
                  "barcodes/distance_from_read_end_for_grouping": ("grouping_distance", int),
                  "barcodes/max_GenomeDistance_forGrouping": ("max_dist", int), "barcodeUMIFinder/sam_records_chunk_size": ("chunk_size", int)}
RUN_TIME = ("readscanner/fileWithAllPossibleTenXbarcodes", "readscanner/mergeBCsED", "barcodeUMIFinder/gene_name_attribute")


class CliError(Exception):
    pass


def find_config(explicit=None):
    """-c file, else ./config.xml, else the application's directory (OneProgramMainBase.checkCfgFilePath)"""
    for p in ([explicit] if explicit else []) + [os.path.join(os.getcwd(), "config.xml"), os.path.join(_HERE, "config.xml")]:
        if p and os.path.isfile(p):
            return p
    if explicit:
        raise CliError(f"config file {explicit} not found")
    return None


def read_config(path):
    """-> {"section/knob": text} of the leaves two levels under <Parameters>; the knobs the library is built for are checked"""
    knobs = {}
    if path is None:
        return knobs          # no file anywhere: the built-in values are the shipped file's
    try:
        root = ET.parse(path).getroot()
    except ET.ParseError as e:
        raise CliError(f"{path}: not well-formed XML ({e})")
    for sec in root:
        for leaf in sec:
            if len(leaf) == 0 and leaf.text is not None:
                knobs[f"{sec.tag}/{leaf.tag}"] = leaf.text.strip()
    bad = [(k, knobs[k], v) for k, v in COMPILED_IN.items() if k in knobs and _norm(knobs[k]) != _norm(v)]
    if bad:
        lines = "\n".join(f"  {k} = {got!r} (this build: {want!r})" for k, got, want in bad)
        raise CliError(f"{path}: knobs that are compiled into this build of libsicelore_mi.so:\n{lines}\n"
                       "(see sicelore-2.1_amd/cli.py COMPILED_IN; every other knob of the file is taken at run time)")
    return knobs


def run_knobs_from(knobs, path="config.xml", polya=None):
    """config.xml's elements -> smi_run_knobs (lib.RunKnobs); None when the file changes none of them (the kernels compiled for the shipped values
    then run).  polya: the (length, fraction, window) the run uses -- command line over config.xml -- which then replace the file's.  A value the
    library has no kernel for is refused with the knob's name (the checks of smi_ctx_set_knobs, made here without a device)."""
    from . import lib
    over = {}
    for name, field in lib.KNOB_FIELDS.items():
        if name in knobs:
            over[name] = knobs[name]
    if not over:
        return None
    try:
        k = lib.run_knobs(**over)
        if polya is not None:
            k.polya_len, k.polya_frac, k.window_polya = polya
        lib.check_run_knobs(k)
    except (ValueError, lib.SmiError) as e:
        raise CliError(f"{path}: {e}" + (" (-p / -f / -w of the command line stand for polyAT/polyATlength, fractionATInPolyAT, windowSearchForPolyA)" if "polyAT/" in str(e) else ""))
    return None if k.as_dict() == lib.run_knobs().as_dict() else k


def host_knobs_from(knobs, table, path="config.xml"):
    out = {}
    for name, (kw, conv) in table.items():
        if name in knobs:
            try:
                out[kw] = conv(knobs[name])
            except ValueError:
                raise CliError(f"{path}: {name} = {knobs[name]!r}: not a number")
    return out


def _norm(v):
    try:
        return repr(float(v))
    except ValueError:
        return v.strip()


_CODE = {ord("A"): 0, ord("G"): 1, ord("C"): 2, ord("T"): 3}


def read_barcode_file(path, length=16):
    """one barcode per line, everything from the first '-' dropped, gz or plain (NanoporeReadScannerMain.readBarcodesFile L480-503)
    -> sorted unique uint64 keys (2 bits per base, A 0 G 1 C 2 T 3, first base most significant)"""
    op = gzip.open if path.endswith(".gz") else open
    keys = []
    with op(path, "rb") as f:
        for ln, line in enumerate(f, 1):
            s = line.split(b"-", 1)[0].strip().upper()
            if not s:
                continue
            if len(s) != length or any(c not in _CODE for c in s):
                raise CliError(f"{path}:{ln}: not a {length}-base A/C/G/T barcode: {line[:40]!r}")
            v = 0
            for c in s:
                v = (v << 2) | _CODE[c]
            keys.append(v)
    if not keys:
        raise CliError(f"{path}: no barcodes")
    return np.unique(np.array(keys, dtype=np.uint64))


def _parse(argv, spec, refused):
    """commons-cli style: spec {canonical: (short, long, takes value)}; refused {option: why}"""
    by_opt = {}
    for name, (sh, lg, has) in spec.items():
        by_opt["-" + sh] = (name, has)
        by_opt["--" + lg] = (name, has)
    out, i = {}, 0
    while i < len(argv):
        a = argv[i]
        if a in refused:
            raise CliError(f"option {a}: {refused[a]}")
        if a not in by_opt:
            raise CliError(f"unknown option {a!r}")
        name, has = by_opt[a]
        if has:
            if i + 1 >= len(argv):
                raise CliError(f"option {a} needs a value")
            out[name] = argv[i + 1]
            i += 2
        else:
            out[name] = True
            i += 1
    return out


SCAN_SPEC = {"inDir": ("d", "inDir", True), "outDir": ("o", "outDir", True), "bcEditDistance": ("b", "bcEditDistance", True),
             "compress": ("c", "compress", False), "ncpu": ("t", "ncpu", True), "fivePbc": ("h", "fivePbc", False),
             "noPolyARequired": ("y", "noPolyARequired", False), "bcWhitelist": ("a", "bcWhitelist", True), "logFile": ("l", "logFile", True),
             # NanoporeReadScannerMain.java:L138-146, L180-183 (file selection), L211-213 (-s), L239-240 (-u), L245-246 (-g)
             "cellRangerBCs": ("g", "cellRangerBCs", True), "skipNfastqs": ("k", "skipNfastqs", True), "onlyNfastqs": ("z", "onlyNfastqs", True),
             "nonrecursive": ("n", "nonrecursive", False), "pattern": ("v", "pattern", True), "dontwrite": ("s", "dontwrite", False),
             "trimfastq": ("u", "trimfastq", False),
             # -e: the specificity experiment (Parser.java:L212-215; /root/reference/README.md:176): random sequences in place of the reads' barcodes
             "randomBarcode": ("e", "randomBarcode", False),
             # the polyA finder's window (L227-234); config.xml's polyAT values where the command line has none
             "polyAlength": ("p", "polyAlength", True), "fractionAT": ("f", "frac-f", True), "windowAT": ("w", "windowAT", True)}
SCAN_REFUSED = {o: why for opts, why in (
    (("-i",), "Use either -i or -d: only -d <directories> is built"),)
    for o in opts}
UMI_SPEC = {"inFileNanopore": ("i", "inFileNanopore", True), "outfile": ("o", "outfile", True), "annotationFile": ("a", "annotationFile", True),
            "config": ("c", "config", True), "chunksize": ("v", "chunksize", True), "fivePbc": ("p", "fivePbc", False),
            "splitReadName": ("w", "splitReadNameAtUnderscore", False), "logFile": ("l", "logFile", True), "ncpu": ("t", "ncpu", True),
            # -b: barcodes whose ed in the read name is larger are ignored (UmiFinderMain.java:L181-182, FastqRecordExt.java:L450-456);
            # -u: read by the Illumina-guided UMI analyzer only (IlluminaUMIanalyzer) -- accepted, checked to be a number, without effect here
            "bcedit": ("b", "bcedit", True), "umiedit": ("u", "umiedit", True), "noclustering": ("s", "noclustering", False),
            "debug": ("d", "debug", False),
            # -e / -f: the accuracy simulations (UmiFinderMain.java:L212-215; UmiFinderWorker$BamWriters.java:L294, L412: no BAM is written under either).
            # -f: every read's UMI is replaced by a random one before the pair distances (ClusteringEditDistanceBase.java:L308-310); -e replaces barcodes in the
            # Illumina-guided analyzer only -- here it leaves the statistics-only run
            "randomBarcode": ("e", "randomBarcode", False), "randomUMI": ("f", "randomUMI", False),
            # -g: the two-letter attribute the gene name is written under and counted from (UmiFinderMain.java:L239-246; config.xml gene_name_attribute)
            "ONTgene": ("g", "ONTgene", True)}   # (-d: stepwise execution of the reference, UmiFinderMain.java:L178-179: accepted, without effect)
UMI_REFUSED = {o: why for opts, why in (
    (("-k", "--inFile10x", "-j", "-y", "-m", "-n", "-z", "--edBCbailout"), "Illumina-guided assignment is outside this build (SURVEY 2: OUT OF SCOPE)"),
    )
    for o in opts}


def _join_ranks():
    """Started once per GPU (torchrun / torch.distributed.run: WORLD_SIZE, RANK, LOCAL_RANK, MASTER_* in the environment), the command joins the
    process group: `scanfastq` then deals the files to the ranks and sums the pass-1 histogram over them, `assignumis` deals whole chromosomes
    (run_files.py, assignumis.py).  Backend: RCCL ("nccl") with a GPU per rank; SMI_DIST_BACKEND=gloo for ranks that share a device."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return False
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return False
    backend = os.environ.get("SMI_DIST_BACKEND") or ("nccl" if torch.cuda.device_count() >= int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"])) else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return True


def _context():
    import torch

    from . import lib
    return lib.Context(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))


def _ncpu(o):
    """worker threads: -t / --ncpu (scanfastq; NanoporeReadScannerMain.java:L221), else what the JVM would call availableProcessors() -- the
    `java` stand-in hands over -XX:ActiveProcessorCount=N (sicelore-nf/main.nf:83) as SMI_ACTIVE_PROCESSORS -- capped at the 16 lanes a GPU is given"""
    for v in (o.get("ncpu"), os.environ.get("SMI_ACTIVE_PROCESSORS")):
        if v:
            try:
                n = int(v)
            except ValueError:
                raise CliError(f"--ncpu / -XX:ActiveProcessorCount {v!r}: not a number")
            if n > 0:
                return n
    return min(16, len(os.sched_getaffinity(0)))


def scanfastq(argv):
    o = _parse(argv, SCAN_SPEC, SCAN_REFUSED)
    for need in ("inDir", "outDir", "bcEditDistance"):
        if need not in o:
            raise CliError(f"Missing required option: {SCAN_SPEC[need][1]}")
    try:
        ed = int(o["bcEditDistance"])
    except ValueError:
        raise CliError(f"--bcEditDistance {o['bcEditDistance']!r}: not a number")
    if ed not in (0, 1, 2):
        raise CliError("--bcEditDistance: this build matches at edit distance 0, 1 or 2")
    cfg_path = find_config()
    knobs = read_config(cfg_path)
    host_kw = host_knobs_from(knobs, HOST_KNOBS_SCAN, cfg_path)
    for kw in ("min_count_fold", "cells_fold_below_max"):
        if host_kw.get(kw, 1) <= 0:
            raise CliError(f"{cfg_path}: readscanner/{'minCountFold' if kw == 'min_count_fold' else 'cellsWithReadsnFoldBelowMaxToKeep'} must be positive")
    # mergeBCsED: null / absent = the barcode edit distance (ReadScannerParameters.validate_readScannerParameters L236-238); a smaller value than
    # --bcEditDistance is a warning in the reference (L239-240), and here
    merge_txt = knobs.get("readscanner/mergeBCsED", "null")
    merge_ed = ed
    if merge_txt not in ("null", ""):
        try:
            merge_ed = int(merge_txt)
        except ValueError:
            merge_ed = ed          # (JAXB leaves the Integer null on an unparsable text: the default applies)
        if not 0 <= merge_ed <= 2:
            raise CliError(f"{cfg_path}: readscanner/mergeBCsED = {merge_ed}: this build merges colliding barcodes at edit distance 0, 1 or 2")
        if ed > merge_ed:
            print(f"Edit distance for generating table of used barcodes <mergeBCsED> in config.xml ({merge_ed}) is smaller than edit distance for final "
                  f"barcode assignment ({ed}) - SHOULD BE AVOIDED", file=sys.stderr)
    try:
        polya = (int(o.get("polyAlength") or knobs.get("polyAT/polyATlength", 15)), float(o.get("fractionAT") or knobs.get("polyAT/fractionATInPolyAT", 0.75)),
                 int(o.get("windowAT") or knobs.get("polyAT/windowSearchForPolyA", 150)))
    except ValueError:
        raise CliError("-p / -w take whole numbers, -f a fraction (polyAT/polyATlength, fractionATInPolyAT, windowSearchForPolyA in config.xml)")
    run_knobs = run_knobs_from(knobs, cfg_path, polya)
    if run_knobs is not None:              # the window the run uses (command line over config.xml) is part of the knobs the context gets
        polya = None
    elif polya == (15, 0.75, 150):
        polya = None                       # the shipped window: the kernels compiled for it
    for d in [d for d in o["inDir"].split(",") if d]:            # -d takes a comma-separated list (FileTools.java:L52)
        if not os.path.isdir(d):
            raise CliError(f"input directory {d} does not exist")
    def number(name):
        try:
            v = int(o[name])
        except ValueError:
            raise CliError(f"--{name} {o[name]!r}: not a number")
        if v < 0:
            raise CliError(f"--{name} {v}: negative")
        return v

    skip = number("skipNfastqs") if "skipNfastqs" in o else 0
    only = number("onlyNfastqs") if "onlyNfastqs" in o else None
    from . import run_files
    used = None
    if "cellRangerBCs" in o:
        # -g: pass 1 is skipped and this list is searched (L300-302).  A file that does not exist is a warning in the reference, which then
        # runs both passes (ReadScannerParameters.java:L277-279); the whitelist is not needed with it (L246)
        if os.path.isfile(o["cellRangerBCs"]):
            used = read_barcode_file(o["cellRangerBCs"])
            print(f"Using Barcodes from: {o['cellRangerBCs']}\nCellranger list of barcodes was supplied --> skipping 1st pass (Search for used barcodes)")
        else:
            print(f"Warning: File with CellRanger barcodes {o['cellRangerBCs']} not found. Despite being provided in command line. "
                  "Searching Barcodes without CellRanger data", file=sys.stderr)
    keys = None
    if used is None:
        wl_name = o.get("bcWhitelist") or knobs.get("readscanner/fileWithAllPossibleTenXbarcodes", "3M-february-2018.txt.gz")
        wl = "" if wl_name in ("none", "None", "null", "Null") else next((p for p in (wl_name, os.path.join(os.getcwd(), wl_name), os.path.join(_HERE, wl_name)) if os.path.isfile(p)), None)
        if wl is None:
            raise CliError(f"file with all possible barcodes {wl_name!r} not found (looked in the working directory and in {_HERE}); "
                           "-a <file> names it (ReadScannerParameters.java:L247)")
        # -a none / null (L132-133): no list of possible barcodes, pass 1 counts every barcode it cuts
        keys = read_barcode_file(wl) if wl else None
    if o["outDir"] != "null":              # (-o null: statistics only, NanoporeReadScannerMain.java:L216)
        parent = os.path.dirname(os.path.abspath(o["outDir"]))
        if not os.path.isdir(parent):
            raise CliError(f"parent directory of output directory {o['outDir']} must exist")
    else:
        raise CliError("-o null (no output directory at all) is not built: -s keeps the statistics and TSVs and writes no FASTQ")
    if skip:
        print(f"skipping first {skip} fastq files as specified in command line")
    if only is not None:
        print(f"using only {only} fastq files as specified in command line")
    if "dontwrite" in o:
        print("Stats only, Won't write fastqs")
    ctx = _context()
    ncpu = _ncpu(o)
    if run_knobs is not None:
        from . import lib as _lib
        try:
            ctx.set_knobs(run_knobs)
        except _lib.SmiError as e:
            raise CliError(f"{cfg_path}: {e}")
    rnd_seed = 0
    if "randomBarcode" in o:      # the reference's java.util.Random is unseeded; here the run can be repeated (SMI_RANDOM_SEED, default 1)
        rnd_seed = int(os.environ.get("SMI_RANDOM_SEED", "1")) or 1
        print(f"Random barcode simulation: barcode sequences are replaced by random sequences during barcode assignment (seed {rnd_seed})")
    info = run_files.run(ctx, o["inDir"], o["outDir"], polya=polya, max_ed=ed, merge_ed=merge_ed, random_barcode_seed=rnd_seed, command_line="scanfastq " + " ".join(argv), **host_kw, n_workers=ncpu, whitelist_keys=keys, five_prime=bool(o.get("fivePbc")),
                         dont_search_polya=bool(o.get("noPolyARequired")), compress=bool(o.get("compress")),
                         recursive="nonrecursive" not in o, pattern=o.get("pattern", run_files.FASTQ_PATTERN), skip_files=skip, only_files=only,
                         used_keys=used, write_fastqs="dontwrite" not in o, trim_fastq="trimfastq" in o)
    print(f"{info.get('files', 0)} Files found")
    print(f"DONE -- {info.get('reads', 0)} reads, {info.get('passed', 0)} passed, {info.get('assigned', 0)} barcode-assigned, "
          f"{info.get('wall_s', 0.0):.1f} s")
    return 0


def assignumis(argv):
    o = _parse(argv, UMI_SPEC, UMI_REFUSED)
    for need in ("inFileNanopore", "outfile"):
        if need not in o:
            raise CliError(f"Missing required option: {UMI_SPEC[need][1]}")
    cfg_path = find_config(o.get("config"))
    knobs = read_config(cfg_path)
    hk = host_knobs_from(knobs, HOST_KNOBS_UMI, cfg_path)
    umi_length = hk.get("umi_length", 12)
    if not 8 <= umi_length <= 12:
        raise CliError(f"{cfg_path}: umis/umi_length = {umi_length}: this build has kernels for UMIs of 8 .. 12 bases")
    cluster_over = {k: hk[k] for k in ("complete_link_ed", "single_link_ed", "single_link_switch") if k in hk}
    for k, v in cluster_over.items():
        if v < 0 or (k != "single_link_switch" and v > 5):
            raise CliError(f"{cfg_path}: umis/{'umi_completelinkclusteringED' if k == 'complete_link_ed' else 'umi_singlelinkclusteringED' if k == 'single_link_ed' else 'complexity_threshold_for_switch_to_single_link_clustering'} = {v}: "
                           "the UMI distances are Levenshtein distances cut at 4 (0 .. 5 make sense here)")
    gene_tag = knobs.get("barcodeUMIFinder/gene_name_attribute", "GE")
    if "ONTgene" in o:
        gene_tag = o["ONTgene"]
        if len(gene_tag) != 2 or not gene_tag.isalpha() or not gene_tag.isascii():
            raise CliError("!!!!!!!!   -g option should have two letters !!!!!!!!!!!!")     # (the reference prints its help under this line and exits with 1)
    elif "annotationFile" in o and (len(gene_tag) != 2 or not gene_tag.isascii() or not gene_tag.isalpha()):
        raise CliError("Annotation file supplied but gene name attribute undefined or incorrect in config.xml (the reference goes on without genes and gene counts: "
                       "not built)")
    if not os.path.isfile(o["inFileNanopore"]):
        raise CliError(f"input BAM {o['inFileNanopore']} does not exist")
    if "annotationFile" in o:
        st = o["annotationFile"]
        base = st[:-3] if st.endswith(".gz") else st[:-4] if st.endswith(".bz2") else st       # (GeneAnnotationReader.loadAnnotationsFile L47-48 strips both)
        if not (base.endswith(".gtf") or base.lower().endswith(".refflat")):
            raise CliError(f"Annotation file name is {st} file name should end with .gtf or .refFlat")       # (UmiFinderMain.java:L255-256)
    if "annotationFile" in o and not os.path.isfile(o["annotationFile"]):
        raise CliError(f"annotation file {o['annotationFile']} does not exist")
    out = o["outfile"]
    prefix = out[:-4] if out.endswith(".bam") else out          # <out>.bam, <out>_umifound_.bam, <out>.genecounts.tsv, <out>.UMIdepths.tsv
    try:
        chunk = int(o.get("chunksize") or hk.get("chunk_size", 250000))
    except ValueError:
        raise CliError(f"--chunksize {o.get('chunksize')!r}: not a number")
    max_dist = hk.get("max_dist", 500)
    from . import assignumis as au
    from . import lib as _lib
    cluster_cfg = _lib.umi_cluster_config(**cluster_over) if cluster_over else None
    refflat = None
    if "annotationFile" in o:          # refFlat text, gz or plain (picard's RefFlatReader through IOUtil)
        import bz2
        an = o["annotationFile"]
        with (gzip.open if an.endswith(".gz") else bz2.open if an.endswith(".bz2") else open)(an, "rt") as f:
            refflat = f.read()
        if (an[:-3] if an.endswith(".gz") else an[:-4] if an.endswith(".bz2") else an).endswith(".gtf"):
            from . import lib as _l
            refflat = _l.GtfText(refflat)          # GeneAnnotationReader.loadAnnotationsFile L50-51: the GTF reader by the file's name
    bc_limit = None
    for name in ("bcedit", "umiedit"):
        if name in o:
            try:
                v = int(o[name])
            except ValueError:
                raise CliError(f"--{name} {o[name]!r}: not a number")
            if name == "bcedit":
                bc_limit = v
    simulate = "randomBarcode" in o or "randomUMI" in o
    rnd_umi = (int(os.environ.get("SMI_RANDOM_SEED", "1")) or 1) if "randomUMI" in o else 0
    ctx = _context()
    ncpu = _ncpu(o)
    info = au.assignumis_stream(ctx, o["inFileNanopore"], prefix, chunk_size=chunk, truncate_read_name=bool(o.get("splitReadName")), n_threads=ncpu,
                                refflat=refflat, max_dist=max_dist, five_prime=bool(o.get("fivePbc")), bc_edit_limit=bc_limit,
                                no_clustering="noclustering" in o, gene_tag=gene_tag, umi_length=umi_length, cluster_cfg=cluster_cfg,
                                grouping_distance=hk.get("grouping_distance"), simulate=simulate, random_umi_seed=rnd_umi)
    if simulate and info.get("rank", 0) == 0:
        print(f"SIMULATION ({'random UMIs' if rnd_umi else 'random barcodes'}): no BAM written; {info['clustered']} of {info['records']} records ended up in UMI clusters"
              + (f" by chance (seed {rnd_umi})" if rnd_umi else ""))
    if info.get("rank", 0) == 0:       # rank 0 holds the whole run's counts (assignumis_stream gathers them)
        print(f"DONE -- {info['records']} records, {info['clustered']} in UMI clusters")
        bad = int(info.get("gene_keys_order_dependent", 0))
        if bad:
            # reads with alignments on chromosomes of two ranks: their (gene, cell, UMI) counters depend on which alignment is seen first,
            # so the later rank's were left out of genecounts.tsv / UMIdepths.tsv; one process has no such keys
            print(f"WARNING: {bad} (gene, cell, UMI) keys of reads aligned on chromosomes of different ranks were counted on the first rank only: "
                  f"{prefix}.genecounts.tsv / .UMIdepths.tsv can differ from a single-process run in those keys (the BAMs do not)", file=sys.stderr)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    try:
        if not argv or argv[0] in ("-h", "--help", "help"):
            print(__doc__)
            return 0
        sub, rest = argv[0], argv[1:]
        if sub in ("scanfastq", "assignumis"):
            joined = _join_ranks()
            try:
                return scanfastq(rest) if sub == "scanfastq" else assignumis(rest)
            finally:
                if joined:
                    import torch.distributed as dist
                    dist.destroy_process_group()
        raise CliError(f"sub-command {sub!r}: this build has scanfastq and assignumis (tagbamwithread, mergestats, illuminaparser: SURVEY 2, out of scope)")
    except CliError as e:
        print(f"ERROR: {e}", file=sys.stderr)
        return 1
    except Exception as e:      # a failed batch: message, exit code 1 (WorkerReadscanner.java:L376-378); the library has no fallback to try
        from . import lib
        if isinstance(e, (lib.SmiError, OSError, ValueError)):
            print(f"ERROR: {type(e).__name__}: {e}", file=sys.stderr)
            return 1
        raise
