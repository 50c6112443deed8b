"""The account of how much of the reference's bytecode the executed fixtures reach (tests/golden/ref_exec_coverage.json, written by
`tools/make_ref_exec.py --coverage` from the per-section bitmaps in tests/golden/coverage/ and every method's LineNumberTable)."""
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_hot_path_lines_are_executed_or_explained():
    cov = json.load(open(os.path.join(GOLD, "ref_exec_coverage.json")))
    st = cov["scope_total"]                                   # the source-line ranges SURVEY 8a cites (tools/coverage_notes.json `scope`)
    assert st["lines"] >= 1700 and st["pct_hit"] >= 93.0, st
    assert st["pct_hit_or_annotated"] >= 99.0, st
    # every class in scope: what is neither executed nor explained is listed by name -- and is empty
    left = {c.split("/")[-1]: e["summary"]["scope"]["unexplained"] for c, e in cov["classes"].items() if e["summary"].get("scope")}
    assert len(left) >= 30 and all(not v for v in left.values()), {k: v for k, v in left.items() if v}
    # the sections that were folded in are the fixtures of this directory
    for sec in cov["sections_merged"]:
        assert os.path.isfile(os.path.join(GOLD, f"ref_exec_{sec}.json")), sec
        assert os.path.isfile(os.path.join(GOLD, "coverage", f"{sec}.json")), sec
    assert {"pass2w_3p", "pass2w_3p_ed2", "pass2w_5p", "pass2w_5p_polya", "pass2x_3p", "group2", "cluster_own2", "pass1_5p"} <= set(cov["sections_merged"])


def test_whole_record_fixtures_are_deep():
    """>= 500 input reads per configuration through the reference's own Parser.call (round 3 had 92 records in all)"""
    total = 0
    for name in ("pass2w_3p", "pass2w_3p_ed2", "pass2w_5p", "pass2w_5p_polya"):
        sec = json.load(open(os.path.join(GOLD, f"ref_exec_{name}.json")))["sections"][0]
        n_in = sum(len(c["reads"]) for c in sec["cases"])
        n_out = sum(len(c["result"].get("records", [])) for c in sec["cases"])
        assert n_in >= 500 and n_out >= 500 and sum(c["hash_orders_agree"] for c in sec["cases"]) >= 95, name
        total += n_out
    assert total >= 2100
