"""Scan statistics (smi_record_flags / smi_scan_stats_*): the whole 64-bit ReadFlags word of a record against the `flag` the
reference's own bytecode produced for every record of tests/golden/ref_exec_pass2_*.json, and the ReadFlags.print text against a Python
model of ReadFlags$Flags' print rules."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCAN_LEVEL = ("FAILED", "PASSED_FWD", "PASSED_REV", "POLY_T_5P", "POLY_A_3P", "POLY_A_NOT_FOUND", "POLY_T_5P_POLY_A_3P", "ADAPTER_5P", "ADAPTER_3P", "TSO_5P",
              "TSO_3P", "ADAPTER_SELECTED_DESP_ADAPTER_BOTH_SIDES", "READ_TOO_SHORT", "ADAPTER_5P_AND_3P", "TSO_5P_AND_3P")


@pytest.fixture
def libmod(pkg):
    from sicelore_amd import lib as libmod

    libmod.load_library()
    return libmod


@pytest.mark.parametrize("name", ["pass2_3p", "pass2_5p", "pass2_5p_polya", "pass2_3p_ed2"])
def test_record_flags_equal_reference_flag_word(libmod, name):
    """scan-level bits and barcode as the reference reports them in -> the reference's full flag word out (Parser.assignBarcode's BC_* bits,
    finalizeFlag's PASSED_TOTAL / PASSED_TOT_TSO / TSO_5P_AND_3P_FAILED)"""
    sec = json.load(open(os.path.join(GOLD, f"ref_exec_{name}.json")))["sections"][0]
    fv, five = sec["flag_values"], sec["five_prime"]
    assert [fv[n] for n in libmod.READ_FLAG_NAMES if n != "ALL_READS_AFTER_SPLIT"] == [1 << k for k in range(36)]   # the enum's values
    scan_mask = sum(fv[k] for k in SCAN_LEVEL)
    n = n_bc = 0
    for c in sec["cases"]:
        w = c["result"]
        if "throws" in w:
            continue
        sc = np.zeros(1, dtype=libmod.SCAN_RESULT_DTYPE)[0]
        sc["flags"] = (w["flag"] & scan_mask) & ~fv["FAILED"]         # FAILED is finalizeFlag's to set
        sc["found"] = 1 if w["adapter"] and w["adapter"][1] is not None else 0
        bc = None
        if w["barcode"] is not None:
            b = w["barcode"]
            bc = np.zeros(1, dtype=libmod.BC_RESULT_DTYPE)[0]
            ae = w["adapter"][1]
            bc["found"], bc["ed"], bc["ed_sec"] = 1, b["ed"], b["ed_second"]
            bc["offset"] = b["start"] - (ae + 1 if five else ae - 1)
            n_bc += 1
        assert libmod.record_flags(sc, bc) == w["flag"], (c["name"], hex(libmod.record_flags(sc, bc)), hex(w["flag"]))
        n += 1
    assert n >= 7 and n_bc >= 6
    # a fragment of a split read, and a discarded multi-chimeric read
    sc = np.zeros(1, dtype=libmod.SCAN_RESULT_DTYPE)[0]
    sc["flags"] = fv["PASSED_FWD"] | fv["TSO_5P"]
    assert libmod.record_flags(sc, None, from_split=True) == fv["PASSED_FWD"] | fv["TSO_5P"] | fv["READS_AFTER_SPLIT"] | fv["PASSED_TOTAL"] | fv["PASSED_TOT_TSO"]
    assert libmod.record_flags(sc, None, multi_chimeric=True) == fv["MULTI_CHIMERIC_READS_DISCARDED"] | fv["FAILED"]


def _percent(c, ref):
    r = float(np.float32(c) / np.float32(ref)) if ref else float("nan")
    if r != r:
        return "NaN"
    q = r * 100.0 * 10.0
    k = int(q)
    fr = q - k
    if fr > 0.5 or (fr == 0.5 and k & 1):
        k += 1
    return (str(k // 10) if k // 10 else "") + "." + str(k % 10) + " %"


def test_stats_text_follows_the_print_rules(libmod):
    names = libmod.READ_FLAG_NAMES
    st = np.zeros(libmod.N_SCAN_STATS, dtype=np.uint64)
    vals = dict(ALL_READS_AFTER_SPLIT=1_234_567, READS_AFTER_SPLIT=3000, PASSED_TOTAL=1_100_000, FAILED=134_567, PASSED_FWD=600_000, PASSED_REV=500_000,
                PASSED_TOT_TSO=900_001, POLY_A_3P=610_000, ADAPTER_5P=5, ADAPTER_3P=0, READ_TOO_SHORT=77, ADAPTER_5P_AND_3P=0, BC_FOUND=800_000,
                BC_FOUND_ED0=500_000, BC_FOUND_ED1=300_000, BC_OFFSET0=799_999, BC_OFFSET2=1, MULTI_CHIMERIC_READS_DISCARDED=12)
    for k, v in vals.items():
        st[names.index(k)] = v
    st[37], st[38], st[39] = 1_100_000 * 1234 + 7, 134_567 * 999, 1400      # sums of read lengths, split reads
    text = libmod.scan_stats_tsv(st)
    lines = text.split("\n")
    assert lines[0] == "=======  Scan Stats =======" and lines[1] == ""
    rows = {ln.split("\t")[0]: ln.split("\t")[1:] for ln in lines[2:] if ln}
    all_reads = 1_234_567 - (3000 - 1400)
    assert rows["All Reads"] == [f"{all_reads:,}", "", ""]
    assert rows["Chimeric reads split"] == ["1,400", _percent(1400, all_reads), "of All Reads"]
    assert rows["Multi Chimeric reads discarded n>3"][0] == "12"
    assert rows["Passed (Adapter found)"] == ["1,100,000", _percent(1_100_000, 1_234_567), "of Reads after chimera split"]
    assert "Adapter NOT found" not in rows                                 # FAILED is not printed (print = false)
    assert rows["Mean read length pA and Adapter found"] == ["1,234", "", ""] and rows["Mean read length pA and Adapter NOT found"] == ["999", "", ""]
    assert rows["Adapter at 3\\'(5\\' for 5p barcoding)"] == ["0", ".0 %", "of Reads after chimera split"]      # printed although zero
    assert "TSO at 5\\'" not in rows and "Barcode found ED= 2" not in rows                                        # printOnlyIfNonZero
    assert rows["Barcode found"] == ["800,000", _percent(800_000, 1_100_000), "of Passed (Adapter found)"]
    assert rows["Barcode found ED= 1"] == ["300,000", "37.5 %", "of Barcode found"]
    assert rows["Barcode offset from predicted pos=+/-2"] == ["1", ".0 %", "of Barcode found"]
    # merging is addition (ReadFlags.mergeStats); the derived rows are recomputed from the sums
    twice = libmod.scan_stats_tsv(st * np.uint64(2))
    assert f"{2 * all_reads:,}" in twice and "Mean read length pA and Adapter found\t1,234" in twice


def test_stats_text_equals_the_reference_print(libmod):
    """smi_scan_stats_tsv against what the reference's own ReadFlags.print wrote (tests/golden/ref_exec_stats_print.json: its bytecode executed on
    sets of 3 .. 20,000 flag words with read lengths, the sums added as Parser.call adds them): the same text, character for character.  With
    no failed (or no passed) read at all the reference's print throws an ArithmeticException -- its mean read length is an integer division by
    the count; the product writes 0 there."""
    sec = json.load(open(os.path.join(GOLD, "ref_exec_stats_print.json")))["sections"][0]
    n_equal = 0
    for c in sec["cases"]:
        st = np.zeros(libmod.N_SCAN_STATS, dtype=np.uint64)
        for k, nm in enumerate(libmod.READ_FLAG_NAMES):
            st[k] = c["n_records"] if nm == "ALL_READS_AFTER_SPLIT" else c["counts"][nm]
        st[37], st[38], st[39] = c["sum_len_passed"], c["sum_len_failed"], c["n_reads_split"]
        text = libmod.scan_stats_tsv(st)
        if c["throws"]:
            assert c["throws"] == "java/lang/ArithmeticException" and (c["counts"]["FAILED"] == 0 or c["counts"]["PASSED_TOTAL"] == 0)
            assert "Mean read length pA and Adapter NOT found\t0\t" in text
            continue
        assert text == c["text"], c["n_records"]
        n_equal += 1
    assert n_equal >= 7
    # ReadFlags.mergeStats (the `mergestats` sub-command): three of the sets merged by the reference, printed by the reference
    total = np.zeros(libmod.N_SCAN_STATS, dtype=np.uint64)
    for k in sec["merged"]["cases"]:
        c = sec["cases"][k]
        for i, nm in enumerate(libmod.READ_FLAG_NAMES):
            total[i] += c["n_records"] if nm == "ALL_READS_AFTER_SPLIT" else c["counts"][nm]
        total[37] += c["sum_len_passed"]
        total[38] += c["sum_len_failed"]
        total[39] += c["n_reads_split"]
    assert libmod.scan_stats_tsv(total) == sec["merged"]["text"]


def test_html_page_over_the_statistics(libmod, pkg, tmp_path):
    """ReadScanner.html (README.md:388, sicelore-nf/main.nf:24): a static page over ReadScanner.tsv's rows and BarcodesAssigned.tsv's counts --
    every row of the table with its numbers, the command line escaped, no script and no network reference"""
    import importlib
    import re

    run_files = importlib.import_module("sicelore_amd.run_files")
    stats = np.zeros(libmod.N_SCAN_STATS, dtype=np.uint64)
    stats[:libmod.SMI_N_READ_FLAGS if hasattr(libmod, "SMI_N_READ_FLAGS") else 37] = np.arange(37, dtype=np.uint64) * 13 + 5
    stats[37:] = (123456, 65432, 17)
    (tmp_path / "BarcodesAssigned.tsv").write_text("barcode\ted0\ted1\nACGTACGTACGTACGT\t7\t2\nTTTTACGTACGTACGT\t1\t0\n")
    run_files.write_stats(str(tmp_path), stats, command_line='scanfastq -d "in <x>" -o out & more')
    page = (tmp_path / "ReadScanner.html").read_text()
    tsv = (tmp_path / "ReadScanner.tsv").read_text()
    assert page.startswith("<!DOCTYPE html>") and "<script" not in page and "http" not in page
    assert "scanfastq -d &quot;in &lt;x&gt;&quot; -o out &amp; more" in page
    rows = [ln.split("\t") for ln in tsv.split("\n") if ln.strip()]
    assert len(rows) > 10
    import html as _html
    for r in rows:
        if len(r) > 1:
            assert f"<td>{_html.escape(r[0])}</td><td class=\"n\">{r[1]}</td>" in page, r
    assert "2 barcodes" in page and re.search(r'<td class="n">8</td>\s*<td class="n">2</td>', page)
    assert (tmp_path / "stats.tsv").exists()
