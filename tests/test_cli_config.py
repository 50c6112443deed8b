"""config.xml at run time (round 6), the host side -- no GPU: what sicelore-2.1_amd/cli.py makes of the file (smi_run_knobs for the
library, the finalize folds, mergeBCsED, the UMI stage's values), what it still refuses BY NAME because it is compiled in, and the reference's
output-file naming with the duplicate check (FastqWriterThreadPool$FastQoneFileThread.init L242-250)."""
import importlib
import re

import numpy as np
import pytest

SHIPPED = "/root/reference/Jar/config.xml"


@pytest.fixture(scope="module")
def cli(pkg):
    return importlib.import_module("sicelore_amd.cli")


@pytest.fixture(scope="module")
def lib(pkg):
    return importlib.import_module("sicelore_amd.lib")


def _config(tmp_path, body, name="config.xml"):
    p = tmp_path / name
    p.write_text("<Parameters>" + body + "</Parameters>")
    return str(p)


def test_shipped_defaults_are_the_knob_defaults(cli, lib):
    """smi_run_knobs_default == the shipped values written out as a file: no knob differs, so no knobs are set on the context"""
    k = lib.run_knobs().as_dict()
    assert (k["min_read_length"], k["min_mean_bc_qv"], k["min_mean_read_qv"], k["min_adapter_3p_matches"]) == (200, 8, 8, 8)
    assert (k["polya_len"], round(k["polya_frac"], 4), k["window_polya"], k["internal_pat_len"], round(k["internal_pat_frac"], 4)) == (15, 0.75, 150, 15, 0.70)
    assert (k["adapter3p"], k["adapter3p_complete"], k["adapter3p_max_mm"], k["adapter3p_complete_max_mm"]) == ("CTTCCGATCT", "CTACACGACGCTCTTCCGATCT", 3, 5)
    assert (k["adapter5p"], k["adapter5p_complete"], k["adapter5p_max_mm"], k["adapter5p_complete_max_mm"], k["adapter5p_window"]) == ("CTTCCGATCT", "CTACACGACGCTCTTCCGATCT", 3, 5, 110)
    assert (k["adapter3p5_complete"], k["adapter3p5_complete_max_mm"], k["tso_complete"], k["tso_complete_max_mm"], k["umi_length"]) == \
        ("AAGCAGTGGTATCAACGCAGAGTAC", 5, "AAGCAGTGGTATCAACGCAGAGTACAT", 6, 12)
    assert (k["tso_scan"], k["tso_scan_max_mm"], k["tso_scan_min_consec"], k["tso_scan_min_two_best"], k["tso_scan_window"]) == ("AACGCAGAGTACATGG", 5, 8, 12, 90)


def test_the_reference_shipped_file_parses_to_the_defaults(cli, lib):
    """(build container only: the reference's own Jar/config.xml) every element of the shipped file is accepted and changes nothing"""
    import os

    if not os.path.isfile(SHIPPED):
        pytest.skip("the reference checkout is not here")
    knobs = cli.read_config(SHIPPED)
    assert cli.run_knobs_from(knobs, SHIPPED) is None
    assert cli.host_knobs_from(knobs, cli.HOST_KNOBS_SCAN) == dict(min_count_fold=10, cells_fold_below_max=500)
    assert cli.host_knobs_from(knobs, cli.HOST_KNOBS_UMI) == dict(umi_length=12, complete_link_ed=2, single_link_ed=1, single_link_switch=3000, grouping_distance=100,
                                                                   max_dist=500, chunk_size=250000)
    for name in list(lib.KNOB_FIELDS) + list(cli.COMPILED_IN) + list(cli.NO_EFFECT):
        assert name in knobs or name in ("umis/pregroup_for_clustering_threshold",) or name in knobs, name


def test_run_time_knobs_reach_the_library_structure(cli, lib, tmp_path):
    """the README's printed defaults (QV 10, fold 20: /root/reference/README.md:480-489) and the other knob groups: file -> smi_run_knobs / call arguments"""
    path = _config(tmp_path, "<readscanner><minReadLength>150</minReadLength><minMeanBCqv>10</minMeanBCqv><minMeanReadqv>10</minMeanReadqv><minAdapter3pMatches>6</minAdapter3pMatches>"
                             "<minCountFold>20</minCountFold><cellsWithReadsnFoldBelowMaxToKeep>200</cellsWithReadsnFoldBelowMaxToKeep><mergeBCsED>2</mergeBCsED></readscanner>"
                             "<polyAT><internalpATlength>12</internalpATlength><internalFractionATInPolyAT>0.8</internalFractionATInPolyAT></polyAT>"
                             "<adapter_for3pBarcoding><sequence>AAGAGACAGT</sequence><sequence_complete>GTCAGATGTGTATAAGAGACAG</sequence_complete><maxNeedlemanMismatches>2</maxNeedlemanMismatches>"
                             "<maxCompleteSeqNeedlemanMismatches>4</maxCompleteSeqNeedlemanMismatches></adapter_for3pBarcoding>"
                             "<fiveprimeadapter_for5pBarcoding><AdapterSearchWindow>90</AdapterSearchWindow><maxNeedlemanMismatches>4</maxNeedlemanMismatches></fiveprimeadapter_for5pBarcoding>"
                             "<threeprimeadapter_for5pBarcoding><sequence>ACGT</sequence><maxCompleteSeqNeedlemanMismatches>7</maxCompleteSeqNeedlemanMismatches></threeprimeadapter_for5pBarcoding>"
                             "<tso_for3pBarcoding><sequence_complete>AAGCAGTGGTATCAACGCAGAGTGAAT</sequence_complete><maxCompleteSeqNeedlemanMismatches>8</maxCompleteSeqNeedlemanMismatches>"
                             "<offsetTSOend>3</offsetTSOend><sequence>AACGCAGAGTGAATGG</sequence><maxNeedlemanMismatches>4</maxNeedlemanMismatches>"
                             "<minTSO_NeedlemanConsecutiveMatches>7</minTSO_NeedlemanConsecutiveMatches><minTSO_TwoBestConsecutiveMatches>11</minTSO_TwoBestConsecutiveMatches>"
                             "<windowForTSOsearch>70</windowForTSOsearch></tso_for3pBarcoding>"
                             "<tso_for5pBarcoding><sequence>ACGTACGT</sequence></tso_for5pBarcoding>"
                             "<umis><umi_length>10</umi_length><umi_completelinkclusteringED>1</umi_completelinkclusteringED><umi_singlelinkclusteringED>0</umi_singlelinkclusteringED></umis>"
                             "<barcodes><distance_from_read_end_for_grouping>80</distance_from_read_end_for_grouping><max_GenomeDistance_forGrouping>300</max_GenomeDistance_forGrouping></barcodes>")
    knobs = cli.read_config(path)
    k = cli.run_knobs_from(knobs, path).as_dict()
    assert (k["min_read_length"], k["min_mean_bc_qv"], k["min_mean_read_qv"], k["min_adapter_3p_matches"]) == (150, 10, 10, 6)
    assert (k["internal_pat_len"], round(k["internal_pat_frac"], 4)) == (12, 0.8)
    assert (k["adapter3p"], k["adapter3p_complete"], k["adapter3p_max_mm"], k["adapter3p_complete_max_mm"]) == ("AAGAGACAGT", "GTCAGATGTGTATAAGAGACAG", 2, 4)
    assert (k["adapter5p_window"], k["adapter5p_max_mm"], k["adapter3p5_complete_max_mm"]) == (90, 4, 7)
    assert (k["tso_complete"], k["tso_complete_max_mm"], k["umi_length"]) == ("AAGCAGTGGTATCAACGCAGAGTGAAT", 8, 10)
    assert (k["tso_scan"], k["tso_scan_max_mm"], k["tso_scan_min_consec"], k["tso_scan_min_two_best"], k["tso_scan_window"]) == ("AACGCAGAGTGAATGG", 4, 7, 11, 70)
    assert cli.host_knobs_from(knobs, cli.HOST_KNOBS_SCAN) == dict(min_count_fold=20, cells_fold_below_max=200)
    assert cli.host_knobs_from(knobs, cli.HOST_KNOBS_UMI) == dict(umi_length=10, complete_link_ed=1, single_link_ed=0, grouping_distance=80, max_dist=300)
    # what the chunk workers derive from them, pass by pass (Parser.java:L99,L134-136)
    cfg1 = np.zeros(1, dtype=lib.SCAN_CONFIG_DTYPE)
    lib.load_library().smi_scan_config_from_knobs(__import__("ctypes").byref(cli.run_knobs_from(knobs, path)), 1, 0, 0, cfg1.ctypes.data)
    assert (int(cfg1["adapter_len"][0]), int(cfg1["max_mismatches"][0]), int(cfg1["min_read_length"][0]), int(cfg1["min_mean_bc_qv"][0])) == (22, 2, 150, 10)
    code = dict(A=1, G=2, C=4, T=8)   # the read scan's TSO as 4-bit codes, its window and limits
    assert [int(x) for x in cfg1["tso4"][0]] == [code[c] for c in "AACGCAGAGTGAATGG"]
    assert (int(cfg1["tso_window"][0]), int(cfg1["tso_max_mismatches"][0]), int(cfg1["tso_min_consec"][0]), int(cfg1["tso_min_two_best"][0])) == (70, 4, 7, 11)
    cfg5 = np.zeros(1, dtype=lib.SCAN_CONFIG_DTYPE)
    lib.load_library().smi_scan_config_from_knobs(__import__("ctypes").byref(cli.run_knobs_from(knobs, path)), 2, 1, 1, cfg5.ctypes.data)
    assert (int(cfg5["adapter_len"][0]), int(cfg5["max_mismatches"][0]), int(cfg5["adapter_search_window"][0]), int(cfg5["five_prime"][0]), int(cfg5["dont_search_polya"][0])) == (10, 5, 90, 1, 1)


@pytest.mark.parametrize("body,needle", [
    ("<readscanner><testPlusMinusPos>3</testPlusMinusPos></readscanner>", "readscanner/testPlusMinusPos"),
    ("<barcodes><cell_bc_length>14</cell_bc_length></barcodes>", "barcodes/cell_bc_length"),
    ("<readscanner><seq_prefix>SEQ=</seq_prefix></readscanner>", "readscanner/seq_prefix"),
    ("<readscanner><runningasdemon>true</runningasdemon></readscanner>", "readscanner/runningasdemon"),
])
def test_compiled_in_knobs_are_refused_by_name(cli, tmp_path, body, needle):
    with pytest.raises(cli.CliError) as e:
        cli.read_config(_config(tmp_path, body))
    assert needle in str(e.value) and "compiled into this build" in str(e.value)


@pytest.mark.parametrize("body,needle", [
    ("<adapter_for3pBarcoding><sequence>CTTCCGATCTA</sequence></adapter_for3pBarcoding>", "adapter_for3pBarcoding/sequence"),
    ("<tso_for3pBarcoding><sequence_complete>AAGCAGTGGTATCAACGCAGAGTACATGG</sequence_complete></tso_for3pBarcoding>", "tso_for3pBarcoding/sequence_complete"),
    ("<polyAT><internalpATlength>18</internalpATlength></polyAT>", "polyAT/internalpATlength"),
    ("<umis><umi_length>16</umi_length></umis>", "umis/umi_length"),
    ("<tso_for3pBarcoding><sequence>AACGCAGAGTACATGGG</sequence></tso_for3pBarcoding>", "tso_for3pBarcoding/sequence"),       # 17 bases: the build aligns 16
    ("<tso_for3pBarcoding><windowForTSOsearch>130</windowForTSOsearch></tso_for3pBarcoding>", "tso_for3pBarcoding/windowForTSOsearch"),
    ("<readscanner><minReadLength>many</minReadLength></readscanner>", "readscanner/minReadLength"),
    ("<fiveprimeadapter_for5pBarcoding><AdapterSearchWindow>170</AdapterSearchWindow></fiveprimeadapter_for5pBarcoding>", "AdapterSearchWindow"),
])
def test_values_without_a_kernel_are_refused_by_the_library_by_name(cli, tmp_path, body, needle):
    path = _config(tmp_path, body)
    with pytest.raises(cli.CliError) as e:
        cli.run_knobs_from(cli.read_config(path), path)
    assert needle in str(e.value), str(e.value)


def test_knobs_no_unit_of_the_path_reads_are_accepted(cli, tmp_path):
    path = _config(tmp_path, "<threeprimeadapter_for5pBarcoding><sequence>AAAA</sequence><maxNeedlemanMismatches>9</maxNeedlemanMismatches></threeprimeadapter_for5pBarcoding>"
                             "<tso_for5pBarcoding><sequence>CCCC</sequence><maxNeedlemanMismatches>1</maxNeedlemanMismatches></tso_for5pBarcoding>"
                             "<tso_for3pBarcoding><offsetTSOend>4</offsetTSOend></tso_for3pBarcoding><polyAT><internalMinPolyATlengthForReporting>30</internalMinPolyATlengthForReporting></polyAT>")
    assert cli.run_knobs_from(cli.read_config(path), path) is None


def test_output_names_follow_the_input_extension_and_duplicates_stop_the_run(pkg, tmp_path):
    """FastqWriterThreadPool$FastQoneFileThread.init L242-250: <base>_passed.<extension of the input>, `gz` taken together with the extension in front
    of it; two inputs with one name (two directories of a -d list) would share both outputs: stopped before anything is written"""
    run_files = importlib.import_module("sicelore_amd.run_files")
    lib = importlib.import_module("sicelore_amd.lib")
    assert run_files.out_base_ext("reads_0.fastq.gz") == ("reads_0", "fastq.gz")
    assert run_files.out_base_ext("reads_0.fastq") == ("reads_0", "fastq")
    assert run_files.out_base_ext("a.b.fastq.gz") == ("a.b", "fastq.gz")
    assert run_files.out_base_ext("x.GZ") == ("x", ".GZ")
    assert run_files.out_base_ext("plain") == ("plain", "")
    for d in ("dirA", "dirB"):
        (tmp_path / d).mkdir()
        (tmp_path / d / "reads_0.fastq").write_text("")
    (tmp_path / "dirA" / "reads_0.fastq.gz").write_text("")      # next to reads_0.fastq: other outputs (…_passed.fastq.gz), no clash
    files = run_files.find_fastqs(f"{tmp_path / 'dirA'}", pattern=run_files.FASTQ_PATTERN)
    run_files.check_output_names(files)
    files = run_files.find_fastqs(f"{tmp_path / 'dirA'},{tmp_path / 'dirB'}", pattern=run_files.FASTQ_PATTERN)
    with pytest.raises(lib.SmiError) as e:
        run_files.check_output_names(files)
    assert re.search(r"dirA/reads_0\.fastq and .*dirB/reads_0\.fastq have the same name", str(e.value)) and "reads_0_passed.fastq" in str(e.value)
