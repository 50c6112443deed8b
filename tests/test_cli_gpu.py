"""The jar's command line in front of the library (sicelore-2.1_amd/cli.py): lines 35 and 42 of the reference's quickrun-2.1.sh run as
they stand, with `java` bound to a wrapper that drops `-jar`, `-Xmx..` and the jar's name and starts `python sicelore-2.1_amd`.  Only
the two command strings are quoted here; the mapping step between them (minimap2 / samtools in the reference) is replaced by a BAM
built from the passed reads with synthetic positions."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# /root/reference/quickrun-2.1.sh:35 and :42, verbatim
STEP1 = "$java -jar -Xmx4G Jar/NanoporeBC_UMI_finder-2.1.jar scanfastq -d $fastqdir -o $readscandir --bcEditDistance 1 --compress "
STEP3 = "$java -jar  -Xmx4G Jar/NanoporeBC_UMI_finder-2.1.jar assignumis --inFileNanopore ${mappingdir}passed.bam -o ${umidir}passedParsed.bam --annotationFile Data/gencode.v38.chr12.refFlat"


def _wrapper(tmp_path):
    """the product's `java` stand-in (sicelore-2.1_amd/bin/java) with this interpreter"""
    os.environ["PYTHON"] = sys.executable
    return "bash " + os.path.join(ROOT, "sicelore-2.1_amd", "bin", "java")      # (`$java` is expanded unquoted in the scripts; no reliance on the mode bits of a snapshot)


def _run(cmd, env, cwd):
    return subprocess.run(["bash", "-c", cmd], env=env, cwd=cwd, capture_output=True, text=True, timeout=600)


def test_cli_refuses_what_it_does_not_implement(tmp_path):
    """no GPU needed up to the point where the options are understood: unknown / unbuilt options, missing inputs and config.xml knobs this
    build does not implement end with a message and exit code 1 (the jar: System.exit(1))"""
    env = dict(os.environ, java=_wrapper(tmp_path))
    (tmp_path / "in").mkdir()
    for cmd, needle in (("$java -jar x.jar scanfastq -d in -o out", "bcEditDistance"),
                        ("$java -jar x.jar scanfastq -d in -o out --bcEditDistance 1 -k some", "not a number"),
                        ("$java -jar x.jar scanfastq -d in,nowhere -o out --bcEditDistance 1", "nowhere does not exist"),
                        ("$java -jar x.jar assignumis -i in.bam -o out.bam --inFile10x x.obj", "Illumina-guided"),
                        ("$java -jar x.jar scanfastq -d in -o out --bcEditDistance 1 --frobnicate", "unknown option"),
                        ("$java -jar x.jar scanfastq -d nowhere -o out --bcEditDistance 1", "does not exist"),
                        ("$java -jar x.jar assignumis -o out.bam", "inFileNanopore"),
                        ("$java -jar x.jar assignumis -i in/x.bam -o out.bam -g G1", "two letters"),
                        ("$java -jar x.jar tagbamwithread --inBam a.bam", "sub-command")):
        r = _run(cmd, env, str(tmp_path))
        assert r.returncode == 1 and needle in r.stderr, (cmd, r.returncode, r.stderr[-300:])
    (tmp_path / "config.xml").write_text("<Parameters><polyAT><internalpATlength>18</internalpATlength></polyAT></Parameters>")
    r = _run("$java -jar x.jar scanfastq -d in -o out --bcEditDistance 1", env, str(tmp_path))
    assert r.returncode == 1 and "polyAT/internalpATlength" in r.stderr
    (tmp_path / "config.xml").write_text("<Parameters><polyAT><polyATlength>18</polyATlength></polyAT></Parameters>")     # a run-time knob since round 5
    r = _run("$java -jar x.jar scanfastq -d in -o out --bcEditDistance 1 -p twelve", env, str(tmp_path))
    assert r.returncode == 1 and "whole numbers" in r.stderr


def test_find_fastqs_orders_by_file_name_and_filters(pkg, tmp_path):
    """FileTools.getInfiles + FoundFiles.initialize: comma-separated directories, all levels unless -n, ordered by FILE NAME (not by path),
    the whole path against the pattern, then skip / limit"""
    import importlib

    run_files = importlib.import_module("sicelore_amd.run_files")
    a, b = tmp_path / "a", tmp_path / "b"
    (a / "deep" / "er").mkdir(parents=True)
    b.mkdir()
    for p_ in (a / "r3.fastq", a / "deep" / "r1.fastq.gz", a / "deep" / "er" / "r5.fastq", b / "r2.fastq", b / "r4.fq", a / "notes.txt", a / "r6.fastq.bak"):
        p_.write_bytes(b"")
    both = f"{a},{b}"
    base = lambda fs: [os.path.basename(f) for f in fs]  # noqa: E731
    assert base(run_files.find_fastqs(both)) == ["r1.fastq.gz", "r2.fastq", "r3.fastq", "r5.fastq"]          # r4.fq, notes.txt, *.bak: not the pattern
    assert base(run_files.find_fastqs(both, recursive=False)) == ["r2.fastq", "r3.fastq"]
    assert base(run_files.find_fastqs(both, skip=1, limit=2)) == ["r2.fastq", "r3.fastq"]
    assert base(run_files.find_fastqs(both, pattern=r".*/deep/.*")) == ["r1.fastq.gz", "r5.fastq"]             # (the whole path is matched)
    assert base(run_files.find_fastqs(both, pattern=None)) == ["notes.txt", "r1.fastq.gz", "r2.fastq", "r3.fastq", "r4.fq", "r5.fastq", "r6.fastq.bak"]
    assert run_files.find_fastqs(str(b), skip=5) == []
    from sicelore_amd import lib as libmod
    with pytest.raises(libmod.SmiError, match="did not find"):
        run_files.find_fastqs(str(tmp_path / "nowhere"))


@pytest.mark.gpu
def test_quickrun_lines_35_and_42_run_verbatim(pkg, synth, tmp_path):
    import importlib

    import torch

    import bammodel
    from test_bam import _parse_aux

    run_files = importlib.import_module("sicelore_amd.run_files")
    assignumis = importlib.import_module("sicelore_amd.assignumis")
    dev = torch.device("cuda", 0)
    wl = synth.make_whitelist(30_000, seed=4401, device=dev)
    used = synth.pick_used(wl, 40, seed=4402)
    work = tmp_path / "run"
    (work / "Data").mkdir(parents=True)
    fastqdir, readscandir, mappingdir, umidir = str(work / "fastq") + "/", str(work / "scan") + "/", str(work / "map") + "/", str(work / "umi") + "/"
    for d in (mappingdir, umidir):
        os.makedirs(d)
    n = run_files.write_synthetic_dir(synth, fastqdir, 3, 1500, used, dev, seed=4410, chimera_frac=0.05)
    # the application directory's barcode file (config.xml:37), found in the working directory
    keys = np.sort(wl.cpu().numpy().astype(np.uint64))
    with gzip.open(work / "3M-february-2018.txt.gz", "wt") as f:
        for k in keys:
            f.write("".join("AGCT"[(int(k) >> (2 * (15 - i))) & 3] for i in range(16)) + "-1\n")
    # the shipped knobs, written out: the file is found in the working directory and every knob checks out
    (work / "config.xml").write_text("<Parameters><readscanner><minReadLength>200</minReadLength><testPlusMinusPos>2</testPlusMinusPos>"
                                     "<fileWithAllPossibleTenXbarcodes>3M-february-2018.txt.gz</fileWithAllPossibleTenXbarcodes></readscanner>"
                                     "<barcodeUMIFinder><sam_records_chunk_size>250000</sam_records_chunk_size></barcodeUMIFinder>"
                                     "<polyAT><polyATlength>15</polyATlength><fractionATInPolyAT>0.75</fractionATInPolyAT></polyAT></Parameters>")
    env = dict(os.environ, java=_wrapper(tmp_path), fastqdir=fastqdir, readscandir=readscandir, mappingdir=mappingdir, umidir=umidir)
    r = _run(STEP1, env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    passed = sorted(os.listdir(readscandir + "passed"))
    assert len(passed) == 3 and all(p.endswith("_passed.fastq.gz") for p in passed)
    assert os.path.isfile(readscandir + "BarcodesAssigned.tsv") and os.path.isfile(readscandir + "BarcodeList.tsv")
    names, lens = [], []
    for p in passed:
        lines = gzip.open(readscandir + "passed/" + p).read().split(b"\n")
        names += [ln[1:].split(b" ")[0].decode() for ln in lines[0::4] if ln]
        lens += [len(ln) for ln in lines[1::4]]
    assert len(names) > 0.8 * n and sum("_bc=" in nm for nm in names) > 0.5 * n
    # "mapping": every passed read at a synthetic position of one of eight loci
    rng = np.random.default_rng(5)
    rows = sorted((int(20_000 + 4_000 * (i % 8) + rng.integers(0, 60)), nm, 16 if (i % 8) & 1 else 0, L) for i, (nm, L) in enumerate(zip(names, lens)))
    recs = [bammodel.bam_record(nm, fl, 0, p0, 30, [("M", L)], "C" * L) for p0, nm, fl, L in rows]
    header = bammodel.bam_bytes("@HD\tVN:1.6\tSO:coordinate\n", [("chr12", 10 ** 8)], [])
    with open(mappingdir + "passed.bam", "wb") as f:
        f.write(bammodel.bgzf_compress(header + b"".join(recs), block=16384))
    (work / "Data" / "gencode.v38.chr12.refFlat").write_text("".join(
        f"GENE{g}\tTX{g}\tchr12\t+\t{19_000 + 4_000 * g}\t{22_500 + 4_000 * g}\t{19_100 + 4_000 * g}\t{22_400 + 4_000 * g}\t1\t{19_000 + 4_000 * g},\t{22_500 + 4_000 * g},\n"
        for g in range(8)))
    r = _run(STEP3, env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    for suffix in ("passedParsed.bam", "passedParsed_umifound_.bam", "passedParsed.genecounts.tsv", "passedParsed.UMIdepths.tsv"):
        assert os.path.getsize(umidir + suffix) > 0, suffix
    _, _, out = bammodel.parse_bam(bammodel.bgzf_decompress(open(umidir + "passedParsed.bam", "rb").read()))
    assert len(out) > 0.5 * n
    n_u8 = n_ge = 0
    for o in out:
        tags = {t: v for t, _ty, v in _parse_aux(o["aux"])}
        d = assignumis.scan_data_from_name(o["name"])
        bc = tags["BC"].decode() if isinstance(tags["BC"], bytes) else tags["BC"]
        assert bc == d["bc"]["seq"]
        n_u8 += "U8" in tags
        n_ge += "GE" in tags
    assert n_u8 > 0.9 * len(out) and n_ge > 0.3 * len(out)
    # ---- the other options of the two sub-commands, parsed and acted on (tests/test_run_files_gpu.py holds their results to the default run's)
    listed = [ln.split("\t")[0] for ln in open(readscandir + "BarcodesAssigned.tsv").read().split("\n")[1:] if ln]
    (work / "used.txt").write_text("".join(b_ + "-1\n" for b_ in listed))
    r = _run("$java -jar Jar/x.jar scanfastq -d $fastqdir -o ${readscandir}../scan2 --bcEditDistance 1 -g used.txt -s -n -k 1 -z 2 -u -v '.*synth_.*'", env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "2 Files found" in r.stdout and "skipping 1st pass" in r.stdout and "Won't write fastqs" in r.stdout
    assert sorted(os.listdir(str(work / "scan2"))) == ["BarcodesAssigned.tsv", "ReadScanner.html", "ReadScanner.tsv", "stats.tsv"]
    r = _run("$java -jar Jar/x.jar scanfastq -d $fastqdir -o ${readscandir}../scan3 --bcEditDistance 1 -a none -s", env, str(work))   # no list of possible barcodes
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(os.listdir(str(work / "scan3"))) == ["BarcodeList.tsv", "BarcodesAssigned.tsv", "ReadScanner.html", "ReadScanner.tsv", "stats.tsv"]
    assert len(open(str(work / "scan3" / "BarcodesAssigned.tsv")).read().split("\n")) > 20
    # -p / -f / -w: another polyA window through both passes (the kernels with the finder as a loop); out of the build's range: a message, exit code 1
    r = _run("$java -jar Jar/x.jar scanfastq -d $fastqdir -o ${readscandir}../scan4 --bcEditDistance 1 -p 12 -f 0.8 -w 120 -z 1", env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(os.listdir(str(work / "scan4" / "passed"))) == 1 and os.path.getsize(str(work / "scan4" / "BarcodesAssigned.tsv")) > 100
    r = _run("$java -jar Jar/x.jar scanfastq -d $fastqdir -o ${readscandir}../scan5 --bcEditDistance 1 -p 40", env, str(work))
    assert r.returncode == 1 and ("polyA length" in r.stderr or "polyATlength = 40" in r.stderr), r.stderr[-500:]
    # ---- sicelore-nf/main.nf:32 and :83 as they stand ($params.* filled in as nextflow.config would): --ncpu, -XX:ActiveProcessorCount, the long option of -f
    nf = dict(env, PJ=env["java"], PX="-Xmx4g", PN="Jar/NanoporeBC_UMI_finder-2.1.jar")
    r = _run("$PJ -jar $PX $PN scanfastq -d $fastqdir -o ./passed_nf --ncpu 4 --bcEditDistance 1 --compress", nf, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    for p in passed:       # the run of quickrun's line, file by file (the reads' ids, ranks and names do not depend on the thread count)
        assert gzip.open(str(work / "passed_nf" / "passed" / p)).read() == gzip.open(readscandir + "passed/" + p).read(), p
    r = _run("$PJ -jar $PX -XX:ActiveProcessorCount=4 $PN assignumis --inFileNanopore ${mappingdir}passed.bam -o passedParsed_nf.bam --annotationFile Data/gencode.v38.chr12.refFlat", nf, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    _, _, out_nf = bammodel.parse_bam(bammodel.bgzf_decompress(open(str(work / "passedParsed_nf.bam"), "rb").read()))
    assert [(o_["name"], o_["aux"]) for o_ in out_nf] == [(o_["name"], o_["aux"]) for o_ in out]
    r = _run("$java -jar Jar/x.jar scanfastq -d $fastqdir -o ${readscandir}../scan6 --bcEditDistance 1 --polyAlength 12 --frac-f 0.8 --windowAT 120 -z 1 -s", env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(str(work / "scan6" / "BarcodesAssigned.tsv")).read() == open(str(work / "scan4" / "BarcodesAssigned.tsv")).read()
    # -g XG: the gene name under another attribute; the tables are the default run's
    r = _run(STEP3.replace("passedParsed.bam", "xg.bam") + " -g XG", env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    _, _, xg = bammodel.parse_bam(bammodel.bgzf_decompress(open(umidir + "xg.bam", "rb").read()))
    assert len(xg) == len(out)
    for a_, b_ in zip(xg, out):
        ta, tb = {t: v for t, _ty, v in _parse_aux(a_["aux"])}, {t: v for t, _ty, v in _parse_aux(b_["aux"])}
        assert "GE" not in ta and ta.get("XG") == tb.get("GE") and {t: v for t, v in ta.items() if t != "XG"} == {t: v for t, v in tb.items() if t != "GE"}
    assert open(umidir + "xg.genecounts.tsv").read() == open(umidir + "passedParsed.genecounts.tsv").read()
    r = _run(STEP3.replace("passedParsed.bam", "bad.bam") + " -g G1", env, str(work))
    assert r.returncode == 1 and "two letters" in r.stderr
    # --annotationFile <x.gtf> (README.md:727): the same eight genes as a GTF -> the same tags and tables as the refFlat run
    with open(str(work / "Data" / "genes.gtf"), "w") as f:
        f.write("##format: gtf\n")
        for g in range(8):
            a = f'gene_id "E{g}.1"; gene_type "protein_coding"; gene_name "GENE{g}";'
            t = a + f' transcript_id "T{g}.1"; transcript_name "TX{g}";'
            lo, hi = 19_001 + 4_000 * g, 22_500 + 4_000 * g
            f.write(f"chr12\tsrc\tgene\t{lo}\t{hi}\t.\t+\t.\t{a}\nchr12\tsrc\ttranscript\t{lo}\t{hi}\t.\t+\t.\t{t}\n"
                    f"chr12\tsrc\texon\t{lo}\t{hi}\t.\t+\t.\t{t}\nchr12\tsrc\tCDS\t{lo + 100}\t{hi - 100}\t.\t+\t0\t{t}\n")
    r = _run(STEP3.replace("passedParsed.bam", "gtf.bam").replace("Data/gencode.v38.chr12.refFlat", "Data/genes.gtf"), env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    _, _, via_gtf = bammodel.parse_bam(bammodel.bgzf_decompress(open(umidir + "gtf.bam", "rb").read()))
    assert [(o_["name"], o_["aux"]) for o_ in via_gtf] == [(o_["name"], o_["aux"]) for o_ in out]
    assert open(umidir + "gtf.genecounts.tsv").read() == open(umidir + "passedParsed.genecounts.tsv").read()
    r = _run(STEP3.replace("passedParsed.bam", "bad.bam").replace("Data/gencode.v38.chr12.refFlat", "Data/genes.txt"), env, str(work))
    assert r.returncode == 1 and "should end with .gtf or .refFlat" in r.stderr
    r = _run(STEP3.replace("passedParsed.bam", "limited.bam") + " -b 0 -u 1", env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    _, _, lim = bammodel.parse_bam(bammodel.bgzf_decompress(open(umidir + "limited.bam", "rb").read()))
    n_bc = lambda recs: sum(any(t == "BC" for t, _ty, _v in _parse_aux(o_["aux"])) for o_ in recs)  # noqa: E731
    exact = sum("_ed=0_" in o_["name"] for o_ in out)
    assert 0 < exact < len(out) and n_bc(lim) == exact and n_bc(out) == len(out)       # -b 0: only the barcodes read without an error count
    r = _run(STEP3.replace("passedParsed.bam", "unclustered.bam") + " -s", env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    _, _, unc = bammodel.parse_bam(bammodel.bgzf_decompress(open(umidir + "unclustered.bam", "rb").read()))
    tags_of = lambda o_: {t: v for t, _ty, v in _parse_aux(o_["aux"])}  # noqa: E731
    assert len(unc) == len(out) and [o_["name"] for o_ in unc] == [o_["name"] for o_ in out]
    for a_, b_ in zip(unc, out):          # -s: no UMI tag of any kind, every other tag as in the clustered run
        ta, tb = tags_of(a_), tags_of(b_)
        assert not ({"U7", "U8", "UC", "UZ", "U1", "U2"} & set(ta)) and ta == {t: v for t, v in tb.items() if t not in ("U7", "U8", "UC", "UZ", "U1", "U2")}
    assert os.path.getsize(umidir + "unclustered_umifound_.bam") < os.path.getsize(umidir + "passedParsed_umifound_.bam")



@pytest.mark.gpu
def test_config_xml_knobs_through_the_command_line(pkg, synth, tmp_path):
    """round 6: a config.xml with OTHER values -- the README's printed defaults minMeanBCqv / minMeanReadqv 10 and minCountFold 20
    (/root/reference/README.md:480-489), fewer adapter mismatches, a longer minimal read; then umi_length 10 for assignumis -- is honoured by both
    sub-commands: the run equals run_files.run / assignumis_stream called with the same knobs, and differs from the shipped file's run"""
    import importlib

    import torch

    import bammodel
    from test_bam import _parse_aux

    run_files = importlib.import_module("sicelore_amd.run_files")
    lib = importlib.import_module("sicelore_amd.lib")
    dev = torch.device("cuda", 0)
    wl = synth.make_whitelist(30_000, seed=4501, device=dev)
    used = synth.pick_used(wl, 40, seed=4502)
    work = tmp_path / "run"
    work.mkdir()
    fastqdir = str(work / "fastq") + "/"
    n = run_files.write_synthetic_dir(synth, fastqdir, 2, 1500, used, dev, seed=4510, chimera_frac=0.05, q_lo=38, q_hi=50)
    keys = np.sort(wl.cpu().numpy().astype(np.uint64))
    with gzip.open(work / "3M-february-2018.txt.gz", "wt") as f:
        for k in keys:
            f.write("".join("AGCT"[(int(k) >> (2 * (15 - i))) & 3] for i in range(16)) + "-1\n")
    env = dict(os.environ, java=_wrapper(tmp_path), fastqdir=fastqdir)
    cmd = "$java -jar Jar/x.jar scanfastq -d $fastqdir -o {out} --bcEditDistance 1 --compress"
    r = _run(cmd.format(out="shipped"), env, str(work))            # no config.xml in the working directory: the shipped values
    assert r.returncode == 0, r.stderr[-2000:]
    body = ("<readscanner><minReadLength>500</minReadLength><minMeanBCqv>10</minMeanBCqv><minMeanReadqv>10</minMeanReadqv><minCountFold>20</minCountFold>"
            "<cellsWithReadsnFoldBelowMaxToKeep>100</cellsWithReadsnFoldBelowMaxToKeep><mergeBCsED>2</mergeBCsED></readscanner>"
            "<adapter_for3pBarcoding><maxNeedlemanMismatches>2</maxNeedlemanMismatches><maxCompleteSeqNeedlemanMismatches>4</maxCompleteSeqNeedlemanMismatches></adapter_for3pBarcoding>"
            "<tso_for3pBarcoding><maxCompleteSeqNeedlemanMismatches>5</maxCompleteSeqNeedlemanMismatches></tso_for3pBarcoding>")
    (work / "config.xml").write_text("<Parameters>" + body + "</Parameters>")
    r = _run(cmd.format(out="knobs"), env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    # the same run through the library calls, knobs handed over directly
    ctx = lib.Context(0)
    ctx.set_knobs(lib.run_knobs(min_read_length=500, min_mean_bc_qv=10, min_mean_read_qv=10, adapter3p_max_mm=2, adapter3p_complete_max_mm=4, tso_complete_max_mm=5))
    run_files.run(ctx, fastqdir, str(work / "direct"), max_ed=1, n_workers=4, whitelist_keys=keys, compress=True, merge_ed=2, min_count_fold=20, cells_fold_below_max=100)
    ctx.close()
    differs = 0
    for sub in ("BarcodeList.tsv", "BarcodesAssigned.tsv", "ReadScanner.tsv"):
        a, b, c = (open(str(work / d / sub)).read() for d in ("knobs", "direct", "shipped"))
        assert a == b, sub
        differs += a != c
    assert differs == 3
    for sub in ("passed", "failed"):
        names = sorted(os.listdir(str(work / "knobs" / sub)))
        assert names == sorted(os.listdir(str(work / "direct" / sub))) and len(names) == 2
        for nm in names:     # (the reads' ids, ranks and names do not depend on the thread count)
            assert gzip.open(str(work / "knobs" / sub / nm)).read() == gzip.open(str(work / "direct" / sub / nm)).read(), (sub, nm)
    # a knob that is compiled in: refused by name, nothing run
    (work / "config.xml").write_text("<Parameters><readscanner><testPlusMinusPos>3</testPlusMinusPos></readscanner></Parameters>")
    r = _run(cmd.format(out="refused"), env, str(work))
    assert r.returncode == 1 and "readscanner/testPlusMinusPos" in r.stderr and not os.path.exists(str(work / "refused"))
    # minReadLength below the bases the reference cuts off every read end: stopped with the knob's name (the reference dies on the first read in between)
    (work / "config.xml").write_text("<Parameters><readscanner><minReadLength>120</minReadLength></readscanner></Parameters>")
    r = _run(cmd.format(out="tooshort"), env, str(work))
    assert r.returncode == 1 and "readscanner/minReadLength" in r.stderr, r.stderr[-600:]
    # ---- assignumis with <umi_length>10</umi_length>: U8 / U7 are 10 characters, the first ten of the shipped run's U7 (the same window, cut shorter)
    names, lens = [], []
    for p in sorted(os.listdir(str(work / "shipped" / "passed"))):
        lines = gzip.open(str(work / "shipped" / "passed" / p)).read().split(b"\n")
        names += [ln[1:].split(b" ")[0].decode() for ln in lines[0::4] if ln]
        lens += [len(ln) for ln in lines[1::4]]
    rng = np.random.default_rng(7)
    rows = sorted((int(20_000 + 4_000 * (i % 6) + rng.integers(0, 60)), nm, 16 if (i % 6) & 1 else 0, L) for i, (nm, L) in enumerate(zip(names, lens)))
    recs = [bammodel.bam_record(nm, fl, 0, p0, 30, [("M", L)], "C" * L) for p0, nm, fl, L in rows]
    header = bammodel.bam_bytes("@HD\tVN:1.6\tSO:coordinate\n", [("chr12", 10 ** 8)], [])
    with open(str(work / "passed.bam"), "wb") as f:
        f.write(bammodel.bgzf_compress(header + b"".join(recs), block=16384))
    (work / "config.xml").unlink()
    r = _run("$java -jar Jar/x.jar assignumis --inFileNanopore passed.bam -o u12.bam", env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    (work / "cfg10.xml").write_text("<Parameters><umis><umi_length>10</umi_length><umi_completelinkclusteringED>1</umi_completelinkclusteringED></umis></Parameters>")
    r = _run("$java -jar Jar/x.jar assignumis --inFileNanopore passed.bam -o u10.bam -c cfg10.xml", env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    _, _, o12 = bammodel.parse_bam(bammodel.bgzf_decompress(open(str(work / "u12.bam"), "rb").read()))
    _, _, o10 = bammodel.parse_bam(bammodel.bgzf_decompress(open(str(work / "u10.bam"), "rb").read()))
    assert [o["name"] for o in o10] == [o["name"] for o in o12] and len(o10) > 0.5 * n
    n_u8 = 0
    for a_, b_ in zip(o10, o12):
        ta, tb = {t: v for t, _ty, v in _parse_aux(a_["aux"])}, {t: v for t, _ty, v in _parse_aux(b_["aux"])}
        txt = lambda v: v.decode() if isinstance(v, bytes) else v  # noqa: E731
        assert ("U7" in ta) == ("U7" in tb)
        if "U7" in ta:
            assert len(txt(ta["U7"])) == 10 and txt(tb["U7"])[:10] == txt(ta["U7"])
        if "U8" in ta:
            assert len(txt(ta["U8"])) == 10
            n_u8 += 1
    assert n_u8 > 0.8 * len(o10)
    (work / "cfg16.xml").write_text("<Parameters><umis><umi_length>16</umi_length></umis></Parameters>")
    r = _run("$java -jar Jar/x.jar assignumis --inFileNanopore passed.bam -o u16.bam -c cfg16.xml", env, str(work))
    assert r.returncode == 1 and "umis/umi_length" in r.stderr


@pytest.mark.gpu
def test_accuracy_simulations_through_the_command_line(pkg, synth, tmp_path):
    """`scanfastq -e` (README.md:176: the specificity estimate) and `assignumis -f`: the same command lines with the simulation switch -- far fewer
    barcode-assigned reads, repeatable (the seed is fixed; SMI_RANDOM_SEED names another), and assignumis writes no BAM under -e / -f
    (UmiFinderWorker$BamWriters.java:L294, L412)"""
    import importlib

    import torch

    import bammodel

    run_files = importlib.import_module("sicelore_amd.run_files")
    dev = torch.device("cuda", 0)
    wl = synth.make_whitelist(30_000, seed=4601, device=dev)
    used = synth.pick_used(wl, 40, seed=4602)
    work = tmp_path / "run"
    work.mkdir()
    fastqdir = str(work / "fastq") + "/"
    n = run_files.write_synthetic_dir(synth, fastqdir, 2, 1500, used, dev, seed=4610, chimera_frac=0.05)
    keys = np.sort(wl.cpu().numpy().astype(np.uint64))
    with gzip.open(work / "3M-february-2018.txt.gz", "wt") as f:
        for k in keys:
            f.write("".join("AGCT"[(int(k) >> (2 * (15 - i))) & 3] for i in range(16)) + "-1\n")
    env = dict(os.environ, java=_wrapper(tmp_path), fastqdir=fastqdir)
    cmd = "$java -jar Jar/x.jar scanfastq -d $fastqdir -o {out} --bcEditDistance 1 --compress {sw}"

    def assigned(out):
        rows = [ln.split("\t") for ln in open(str(work / out / "BarcodesAssigned.tsv")).read().split("\n")[1:] if ln]
        return sum(int(c.replace(",", "")) for r in rows for c in r[1:] if c.replace(",", "").isdigit())

    for out, sw, extra in (("real", "", {}), ("sim", "-e", {}), ("sim2", "--randomBarcode", {}), ("sim3", "-e", {"SMI_RANDOM_SEED": "9"})):
        r = _run(cmd.format(out=out, sw=sw), dict(env, **extra), str(work))
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("Random barcode simulation" in r.stdout) == bool(sw)
    a_real, a_sim = assigned("real"), assigned("sim")
    assert a_real > 0.5 * n and a_sim < 0.02 * a_real
    tsv = lambda d: open(str(work / d / "BarcodesAssigned.tsv")).read()  # noqa: E731
    assert tsv("sim") == tsv("sim2")
    # ---- assignumis -f / -e on the real run's reads: counts only, no output file
    names, lens = [], []
    for p in sorted(os.listdir(str(work / "real" / "passed"))):
        lines = gzip.open(str(work / "real" / "passed" / p)).read().split(b"\n")
        names += [ln[1:].split(b" ")[0].decode() for ln in lines[0::4] if ln]
        lens += [len(ln) for ln in lines[1::4]]
    rng = np.random.default_rng(7)
    rows = sorted((int(20_000 + 4_000 * (i % 6) + rng.integers(0, 60)), nm, 16 if (i % 6) & 1 else 0, L) for i, (nm, L) in enumerate(zip(names, lens)))
    recs = [bammodel.bam_record(nm, fl, 0, p0, 30, [("M", L)], "C" * L) for p0, nm, fl, L in rows]
    header = bammodel.bam_bytes("@HD\tVN:1.6\tSO:coordinate\n", [("chr12", 10 ** 8)], [])
    with open(str(work / "passed.bam"), "wb") as f:
        f.write(bammodel.bgzf_compress(header + b"".join(recs), block=16384))
    r = _run("$java -jar Jar/x.jar assignumis --inFileNanopore passed.bam -o plain.bam", env, str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    n_real = int(r.stdout.split("records,")[1].split("in UMI clusters")[0])
    sims = {}
    for sw in ("-f", "--randomUMI", "-e"):
        r = _run(f"$java -jar Jar/x.jar assignumis --inFileNanopore passed.bam -o simu.bam {sw}", env, str(work))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "SIMULATION" in r.stdout and not os.path.exists(str(work / "simu.bam")) and not os.path.exists(str(work / "simu.genecounts.tsv"))
        n_sim = int(r.stdout.split("no BAM written;")[1].split("of")[0])
        sims[sw] = n_sim
    # (this generator gives every read a UMI of its own, so the real run's clusters are chance pairs too: the simulation's count is of that order;
    # -e leaves the UMIs alone; the random UMIs are a function of the seed: -f twice is the same run)
    assert sims["-e"] == n_real and sims["-f"] == sims["--randomUMI"] and sims["-f"] <= max(2 * n_real, 50)
