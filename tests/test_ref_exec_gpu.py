"""The HIP path against REFERENCE-EXECUTED vectors (tests/golden/ref_exec_pass2_*.json: whole records through the reference's
own pass 2, executed from its class files by tools/jvm_exec.py; see tests/test_ref_exec.py for the oracle side).

Every record goes through the C ABI's chunk worker (smi_scanfastq_pass2_chunk: FASTQ text in, passed / failed FASTQ text out,
K-FQ -> K-PACK -> K-SCAN -> K-WIN/K-BC -> K-WRITE on the device) and must come out byte for byte as the reference's
FastqRecordExt.getRecordForWriting wrote it.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_CODE = {"A": 0, "G": 1, "C": 2, "T": 3}


def _key(bc):
    v = 0
    for ch in bc:
        v = (v << 2) | _CODE[ch]
    return v


@pytest.mark.parametrize("packed", [False, True], ids=["text", "packed"])
@pytest.mark.parametrize("name", ["pass2_3p", "pass2_5p", "pass2_5p_polya", "pass2_3p_ed2"])
def test_chunk_worker_records_equal_reference_bytecode(pkg, gpu_ctx, name, packed):
    """packed: smi_scanfastq_pass2_chunk_packed (bit-planes up, decisions down, records written on the host) instead of the text worker"""
    with open(os.path.join(GOLD, f"ref_exec_{name}.json")) as f:
        sec = json.load(f)["sections"][0]
    keys = np.array([_key(b) for b in sec["barcodes"]], dtype=np.uint64)
    ranks = np.array(sec["ranks"], dtype=np.int32)
    order = np.argsort(keys)
    gpu_ctx.set_barcode_set(keys, mode=0)
    n_checked = n_passed = n_bc = 0
    for idx, c in enumerate(sec["cases"]):
        want = c["result"]
        if not c["hash_orders_agree"] or "throws" in want:
            continue
        text = f"@{c['name']}\n{c['seq']}\n+\n{c['qual']}\n".encode()
        passed, failed, info = gpu_ctx.scanfastq_pass2_chunk(text, max_ed=sec["ed"], five_prime=sec["five_prime"],
                                                             dont_search_polya=sec["dont_search_polya"], split_chimeras=False,
                                                             first_read_id=idx + 1, rank_keys=keys[order], rank_values=ranks[order],
                                                             packed=packed, n_threads=2)
        if packed:   # the reference's whole flag word from the results the device sent back (smi_record_flags; the counters behind ReadScanner.html)
            _p, _f, inf = gpu_ctx.scanfastq_pass2_chunk(text, max_ed=sec["ed"], five_prime=sec["five_prime"], dont_search_polya=sec["dont_search_polya"],
                                                        split_chimeras=False, first_read_id=idx + 1, rank_keys=keys[order], rank_values=ranks[order],
                                                        packed=True, n_threads=1, want_results=True)
            assert pkg.lib.record_flags(inf["scan"][0], inf["bc"][0]) == want["flag"], (c["name"], hex(want["flag"]))
            st = inf["stats"]
            assert int(st[3]) == 1 and int(st[5]) == int(want["passed"]) and int(st[25]) == int(want["barcode"] is not None)
        w = want["written"]
        exp = f"@{w['name']}\n{w['bases']}\n+{w['quality_header'] or ''}\n{w['qualities']}\n".encode()
        got = passed if want["passed"] else failed
        other = failed if want["passed"] else passed
        assert got == exp, (c["name"], got[:400], exp[:400])
        assert other == b""
        n_checked += 1
        n_passed += want["passed"]
        n_bc += want["barcode"] is not None
    assert n_checked >= len(sec["cases"]) * 0.8 and n_passed >= 3 and n_bc >= 2


@pytest.mark.parametrize("packed", [False, True], ids=["text", "packed"])
@pytest.mark.parametrize("name", ["pass2w_3p", "pass2w_3p_ed2", "pass2w_5p", "pass2w_5p_polya", "pass2x_3p", "pass2x_5p", "pass2p", "pass2k", "pass2t"])
def test_chunk_worker_equals_parser_call(pkg, gpu_ctx, name, packed):
    """whole chunks (five reads) through the chunk workers with the chimera splitter on, against the records the reference's
    Parser.call left in the chunk (tests/golden/ref_exec_pass2w_*.json: >= 500 input reads per configuration, fragments of split
    reads, multi-chimeric reads, failed reads): the passed and the failed text byte for byte"""
    with open(os.path.join(GOLD, f"ref_exec_{name}.json")) as f:
        secs = json.load(f)["sections"]
    for sec in secs:     # (pass2x_*: a section of targeted reads and one with --trimfastq; pass2p: -p / -f / -w, a 3' and a 5' section;
        #                   pass2k / pass2t, round 6: the reference started with other values of config.xml's knobs (pass2t: the read scan's TSO) -- the context gets the same: smi_ctx_set_knobs)
        if sec.get("polya"):
            gpu_ctx.set_polya(*sec["polya"])
        if sec.get("knobs"):
            gpu_ctx.set_knobs(pkg.lib.run_knobs(**sec["knobs"]))
        try:
            _chunk_section_through_worker(gpu_ctx, sec, packed, wide=name.startswith("pass2w"))
        finally:
            gpu_ctx.set_polya()
            gpu_ctx.set_knobs(None)


def _chunk_section_through_worker(gpu_ctx, sec, packed, wide):
    keys = np.array([_key(b) for b in sec["barcodes"]], dtype=np.uint64)
    ranks = np.array(sec["ranks"], dtype=np.int32)
    order = np.argsort(keys)
    gpu_ctx.set_barcode_set(keys, mode=0)
    n_in = n_out = n_passed = 0
    for c in sec["cases"]:
        if not c["hash_orders_agree"] or "throws" in c["result"]:
            continue
        text = "".join(f"@{r['name']}\n{r['seq']}\n+\n{r['qual']}\n" for r in c["reads"]).encode()
        passed, failed, info = gpu_ctx.scanfastq_pass2_chunk(text, max_ed=sec["ed"], five_prime=sec["five_prime"], dont_search_polya=sec["dont_search_polya"],
                                                             split_chimeras=sec["split_chimeras"], first_read_id=c["first_read_id"], trim_fastq=sec.get("trim_fastq", False),
                                                             rank_keys=keys[order], rank_values=ranks[order], packed=packed, n_threads=2)
        exp_p, exp_f = [], []
        for w in c["result"]["records"]:
            wr = w["written"]
            (exp_p if w["passed"] else exp_f).append(f"@{wr['name']}\n{wr['bases']}\n+{wr['quality_header'] or ''}\n{wr['qualities']}\n")
        assert passed == "".join(exp_p).encode(), (c["chunk"], passed[:300])
        assert failed == "".join(exp_f).encode(), (c["chunk"], failed[:300])
        n_in += len(c["reads"])
        n_out += len(c["result"]["records"])
        n_passed += len(exp_p)
    assert n_in >= (450 if wide else 50) and n_passed >= (300 if wide else 30)
    if sec["split_chimeras"] and wide:
        assert n_out >= n_in + 30   # fragments of split reads


@pytest.mark.parametrize("name", ["umi_3p", "umi_5p", "umi_3p_len10", "umi_5p_len10"])
def test_k_umi_distances_equal_reference_bytecode(pkg, gpu_ctx, name):
    """K-UMI on the windows the product cuts out of the read names == ClusteringEditDistanceBase.calcEditDistances executed from
    the reference's class files (3' and 5' / -p; *_len10: with <umi_length>10</umi_length> in config.xml -- the knob on the context,
    smi_ctx_set_knobs), and the U7 the UMI stage writes for every read == OneNanoporeResult.getPostBCUMIseqOffset(read, 0)"""
    import importlib

    import torch

    au = importlib.import_module("sicelore_amd.assignumis")
    with open(os.path.join(GOLD, f"ref_exec_{name}.json")) as f:
        sec = json.load(f)["sections"][0]
    five = sec["five_prime"]
    ul = sec.get("umi_length", 12)
    wins = []
    for nm in sec["names"]:
        d = au.scan_data_from_name(nm["name"])
        w = au.umi_window(d["x"], d["ae"], d["bc"]["end"], five, ul)
        assert w is not None
        wins.append(au.pack_window(w))
    n = len(wins)
    dev = torch.device("cuda", gpu_ctx.device)
    go, po, mo = gpu_ctx.umi_offsets([n])
    d_out = torch.zeros(int(mo[-1]), dtype=torch.uint8, device=dev)
    if ul != 12:
        gpu_ctx.set_knobs(pkg.lib.run_knobs(umi_length=ul))
    try:
        gpu_ctx.umi_dist_device(torch.from_numpy(np.array(wins, dtype=np.uint64).view(np.int64)).to(dev), torch.from_numpy(go.view(np.int32)).to(dev),
                                torch.from_numpy(po.view(np.int64)).to(dev), torch.from_numpy(mo.view(np.int64)).to(dev), 1, int(po[-1]), d_out)
        torch.cuda.synchronize()
        if ul != 12:
            # the whole stage on these names (device path and host path): U7 = the reference's getPostBCUMIseqOffset(read, 0), umi_length characters
            names = [nm["name"] for nm in sec["names"]]
            cig = [np.array([(1200 << 4) | 0], dtype=np.uint32)] * n
            for host in (False, True):
                if host:
                    os.environ["SMI_AU_HOST"] = "1"
                try:
                    for kw in (dict(), dict(umi_length=ul)):     # the context's knob / the chunk configuration's field
                        tags, n_done = gpu_ctx.assignumis_chunk(names, np.zeros(n, np.uint16), np.arange(n, dtype=np.int32) * 7, cig, five_prime=five, **kw)
                        assert n_done == n
                        for t, nm in zip(tags, sec["names"]):
                            assert t["flags"] & pkg.lib.UMI_HAS_U7 and t["u7"].decode() == nm["post_bc_umi"][1], (host, kw, nm["name"])
                finally:
                    os.environ.pop("SMI_AU_HOST", None)
    finally:
        if ul != 12:
            gpu_ctx.set_knobs(None)
    m = d_out.cpu().numpy().reshape(n, n)
    pos = {"MINUSONE": 0, "ZERO": 1, "PLUSONE": 2}
    for c in sec["cases"]:
        for (a, b), want in (((c["i"], c["j"]), c["distance"]), ((c["j"], c["i"]), c["reverse"])):
            r = int(m[a, b])
            if a > b:
                # the reference fills [v][i] with the TRANSPOSED copy of [i][v] (L213-216), not with calcEditDistances(v, i)
                want = {"ed": c["distance"]["ed"], "pos1": c["distance"]["pos2"], "pos2": c["distance"]["pos1"]}
            assert (r & 15, (r >> 4) & 3, (r >> 6) & 3) == (want["ed"], pos[want["pos1"]], pos[want["pos2"]]), (a, b, want, r)


def test_k_chim_equals_reference_bytecode(pkg, gpu_ctx):
    """K-PACKR + K-CHIM-A/B/C (+ fragment names) == ChimeraFindernew.findSplitPositions executed from the reference's class files"""
    import torch

    with open(os.path.join(GOLD, "ref_exec_chimera_3p.json")) as f:
        sec = json.load(f)["sections"][0]
    seqs = [c["seq"] for c in sec["cases"]]
    dev = torch.device("cuda", gpu_ctx.device)
    n = len(seqs)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    total = int(offs[-1])
    d_reads = torch.from_numpy(np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()).to(dev)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_planes = torch.zeros(gpu_ctx.read_planes_words(total, n), dtype=torch.int32, device=dev)
    gpu_ctx.pack_reads_device(d_reads, d_offs, n, total, d_planes)
    d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    gpu_ctx.chimera_device(d_planes, d_offs, n, total, gpu_ctx.chimera_config(False), d_out)
    torch.cuda.synchronize()
    res = d_out.cpu().numpy().view(pkg.CHIMERA_RESULT_DTYPE).reshape(-1)
    n_split = 0
    for i, c in enumerate(sec["cases"]):
        want = c["records"]
        k = int(res["n_split"][i])
        cuts = [0] + [int(res["pos"][i][j]) for j in range(k)] + [len(c["seq"])]
        assert len(want) == k + 1, (c["name"], k, [w["name"] for w in want])
        for j, w in enumerate(want):
            name = pkg.lib.chimera_fragment_name(c["name"], res[i], j) if k else c["name"]
            assert name == w["name"] and cuts[j + 1] - cuts[j] == w["length"], (c["name"], j, name, w["name"])
        multi = bool(res["flags"][i] & pkg.lib.CHIM_MULTI)
        assert multi == (len(want) == 1 and want[0]["flag"] != 0)
        n_split += k > 0
    assert n_split >= 8


def test_pass1_chunk_worker_equals_reference_bytecode(pkg, gpu_ctx):
    """smi_scanfastq_pass1_chunk (FASTQ text in, histogram increment on the device: K-FQ, K-PACK with qualities, K-SCAN<22> with the
    quality filter, K-HIST) against UsedCellBCListGenerator$Worker.call executed from the reference's class files
    (tests/golden/ref_exec_pass1.json): the counter map, and the per-record filter through the scan entry points"""
    import torch

    with open(os.path.join(GOLD, "ref_exec_pass1.json")) as f:
        sec = json.load(f)["sections"][0]
    assert sec["hash_orders_agree"]
    keys = np.sort(np.array([_key(b) for b in sec["whitelist"]], dtype=np.uint64))
    gpu_ctx.set_barcode_set(keys, mode=1)
    text = "".join(f"@{c['name']}\n{c['seq']}\n+\n{c['qual']}\n" for c in sec["cases"]).encode()
    hist = torch.zeros(keys.size, dtype=torch.int32, device="cuda")
    assert gpu_ctx.scanfastq_pass1_chunk(text, hist) == len(sec["cases"])
    h = hist.cpu().numpy()
    got = sorted([int(keys[i]), int(h[i])] for i in np.nonzero(h)[0])
    assert got == sec["histogram"] and len(got) >= 5
    hist2 = torch.zeros(keys.size, dtype=torch.int32, device="cuda")    # the packed worker: planes, quality tails and sums from the host
    assert gpu_ctx.scanfastq_pass1_chunk(text, hist2, packed=True, n_threads=2) == len(sec["cases"])
    assert (hist2.cpu().numpy() == h).all()
    # the filter per record: pack with qualities -> K-SCAN pass 1 -> pass1_ok
    seqs, quals = [c["seq"] for c in sec["cases"]], [c["qual"] for c in sec["cases"]]
    n = len(seqs)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(q) for q in seqs])
    d_reads = torch.from_numpy(np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()).cuda()
    d_quals = torch.from_numpy(np.frombuffer("".join(quals).encode(), dtype=np.uint8).copy()).cuda()
    d_offs = torch.from_numpy(offs).cuda()
    ends = torch.zeros((28, 2 * n), dtype=torch.int32, device="cuda")
    lens, qsum = torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
    qt = torch.zeros((n, 224), dtype=torch.uint8, device="cuda")
    gpu_ctx.pack_ends_device(d_reads, d_quals, d_offs, n, ends, lens, qt, qsum)
    out = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    win = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
    gpu_ctx.scan_device(ends, lens, n, gpu_ctx.scan_config(1), out, win, qt, qsum)
    torch.cuda.synchronize()
    res = out.cpu().numpy().view(pkg.SCAN_RESULT_DTYPE).reshape(-1)
    assert [bool(x) for x in res["pass1_ok"]] == [c["filter"] for c in sec["cases"]]


@pytest.mark.parametrize("name,five_prime", [("pass1_nowl", False), ("pass1_nowl_5p", True)])
def test_pass1_chunk_worker_without_whitelist_equals_reference_bytecode(pkg, gpu_ctx, name, five_prime):
    """`-a none`: smi_scanfastq_pass1_chunk_keys (the barcode of every read that passes the filter appended to a key list) + smi_count_keys_device
    (sorted, counted) against UsedCellBCListGenerator$Worker.call executed with allPossibleBarcodes == null -- reads with an N inside the
    barcode included: a clean 16-mer for 3' (reverseComplement keeps the low 32 bits), a long with its upper half set for 5'"""
    import torch

    with open(os.path.join(GOLD, f"ref_exec_{name}.json")) as f:
        sec = json.load(f)["sections"][0]
    assert sec["hash_orders_agree"] and sec["whitelist"] is None and sec["five_prime"] == five_prime
    text = "".join(f"@{c['name']}\n{c['seq']}\n+\n{c['qual']}\n" for c in sec["cases"]).encode()
    d_keys = torch.zeros(len(sec["cases"]) + 8, dtype=torch.int64, device="cuda")
    d_count = torch.zeros(1, dtype=torch.int64, device="cuda")
    # (the 5' fixture was executed with the polyA search on, as quickrun's 5' line without -y runs)
    assert gpu_ctx.scanfastq_pass1_chunk_keys(text, d_keys, d_count, five_prime=five_prime, dont_search_polya=False) == len(sec["cases"])
    m = int(d_count.item())
    assert m == sum(c for _, c in sec["histogram"]) == sum(1 for c in sec["cases"] if c.get("filter"))
    uk, uc = gpu_ctx.count_keys_device(d_keys[:m], m)
    assert [[int(k), int(c)] for k, c in zip(uk, uc)] == sec["histogram"]
    assert bool(int(uk.max()) >> 32) == five_prime


def test_count_keys_device_equals_numpy(pkg, gpu_ctx):
    import torch

    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 1000, 300_000):
        k = rng.integers(0, 500, n).astype(np.uint64) * np.uint64(7919) if n else np.zeros(0, dtype=np.uint64)
        if n > 10:
            k[::17] |= np.uint64(0xFFFFFFFF00000000)
        d = torch.from_numpy(k.view(np.int64).copy()).cuda()
        uk, uc = gpu_ctx.count_keys_device(d, n)
        ek, ec = np.unique(k, return_counts=True)
        assert (uk == ek).all() and (uc == ec).all() and uk.size == ek.size

