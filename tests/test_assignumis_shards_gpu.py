"""assignumis split over ranks by chromosome (sicelore-2.1_amd/assignumis.py assignumis_stream under torch.distributed): two and
three ranks over a BAM of three chromosomes and an unmapped tail write what one process writes -- every record with its tags, in the same
order, and the same genecounts.tsv / UMIdepths.tsv -- from the byte ranges the BAM index gives them.

(Chunks of 90 records over loci of about 30: where a locus is larger than what ReadGrouper holds back at a chunk's end -- a third of the
chunk, ReadGrouper.java:L171-184 -- the REFERENCE's own groups depend on where the chunk counter stands, i.e. on -v, and so do a rank's,
whose counter starts at its first chromosome.  With the shipped 250,000 that takes a locus of more than 83,000 reads.)"""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import bammodel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bai_extents_and_shard_plan(pkg):
    """CPU: the index reader (pseudo-bin and plain bins) and the dealing of whole references to ranks"""
    assignumis = importlib.import_module("sicelore_amd.assignumis")
    recs = [(0, 100, 900, 10 << 16, (10 << 16) | 500), (0, 5000, 5800, (10 << 16) | 500, 400 << 16), (2, 70, 800, 400 << 16, 900 << 16), (3, 10, 700, 900 << 16, 1000 << 16),
            (-1, -1, 0, 1000 << 16, 1010 << 16)]
    for meta in (True, False):
        ext = assignumis.bai_ref_extents_bytes(bammodel.bai_bytes(5, recs, meta=meta))
        assert ext == [(10 << 16, 400 << 16), None, (400 << 16, 900 << 16), (900 << 16, 1000 << 16), None]
    assert assignumis.plan_shards(ext, 1) == [(None, None)]
    assert assignumis.plan_shards(ext, 2) == [(None, 400 << 16), (400 << 16, None)] or assignumis.plan_shards(ext, 2) == [(None, 900 << 16), (900 << 16, None)]
    three = assignumis.plan_shards(ext, 3)
    assert three == [(None, 400 << 16), (400 << 16, 900 << 16), (900 << 16, None)]
    five = assignumis.plan_shards(ext, 5)      # more ranks than references with reads: the rest gets nothing
    assert [s for s in five if s != (0, 0)] == three and len(five) == 5


def _make_inputs(pkg, synth, gpu_ctx, tmp_path, meta):
    """a three-chromosome BAM of names from a real pass 2 (+ four unmapped records), its .bai, a refFlat with twelve genes per chromosome"""
    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    assignumis = importlib.import_module("sicelore_amd.assignumis")
    rng = np.random.default_rng(71)
    wl = synth.make_whitelist(40_000, seed=7101)
    used = synth.pick_used(wl, 6, seed=7102)
    n_mol, copies = 90, 4
    mol = synth.gen_reads(n_mol, used, seed=7103, err=0.0, q_mean=20.0)
    seqs, quals, mol_of = [], [], []
    for m in range(n_mol):
        s, q = synth.materialize(mol, m)
        for _ in range(copies):
            t = list(s)
            for p in rng.integers(0, len(t), max(1, len(t) // 50)):
                t[p] = "ACGT"[rng.integers(0, 4)]
            seqs.append("".join(t))
            quals.append(q)
            mol_of.append(m)
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    text = "".join(f"@read{i} runid=x\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(zip(seqs, quals))).encode()
    recs = scanfastq.ReadScanner(gpu_ctx, max_ed=1, split_chimeras=False).pass2_chunk(text)
    rows = []
    for r in recs:
        qname = r["name"].split(" ")[0]
        if "_FAILED" in qname:
            continue
        m = mol_of[r["source"]]
        ref = m % 3                                   # molecules dealt to three chromosomes, twelve loci each
        pos = 30_000 + 4_000 * (m // 3 % 12) + int(rng.integers(0, 80))
        rows.append((ref, pos, qname, 16 if m & 1 else 0, r["length"]))
    rows.sort()
    tail = [(-1, -1, f"unmapped{k}", 4, 300) for k in range(4)]
    block = 4096
    header = bammodel.bam_bytes("@HD\tVN:1.6\tSO:coordinate\n", [("chr1", 10 ** 6), ("chr2", 10 ** 6), ("chr3", 10 ** 6)], [])
    brecs = [bammodel.bam_record(nm, fl, ref, p0, 30, [("M", L)] if ref >= 0 else [], "C" * L) for ref, p0, nm, fl, L in rows + tail]
    data = header + b"".join(brecs)
    offs = bammodel.bgzf_block_offsets(data, block)
    at, idx = len(header), []
    for (ref, p0, _nm, _fl, L), b in zip(rows + tail, brecs):
        idx.append((ref, p0, p0 + L, bammodel.virtual_offset(at, block, offs), bammodel.virtual_offset(at + len(b), block, offs) or ((offs[-1] + 1) << 16)))
        at += len(b)
    in_bam = str(tmp_path / "in.bam")
    with open(in_bam, "wb") as f:
        f.write(bammodel.bgzf_compress(data, block=block))
    with open(in_bam + ".bai", "wb") as f:
        f.write(bammodel.bai_bytes(3, idx, meta=meta))
    refflat = str(tmp_path / "genes.refFlat")
    with open(refflat, "w") as f:
        for c in range(3):
            for g in range(12):
                a = 29_500 + 4_000 * g
                f.write(f"G{c}_{g}\tT{c}_{g}\tchr{c + 1}\t+\t{a}\t{a + 3000}\t{a}\t{a + 3000}\t1\t{a},\t{a + 3000},\n")
    return in_bam, refflat, rows, tail


@pytest.mark.gpu
@pytest.mark.parametrize("world,meta", [(2, True), (3, False), (4, True)])     # 4: more ranks than chromosomes, rank 3 has no range
def test_ranks_write_what_one_process_writes(pkg, synth, gpu_ctx, tmp_path, world, meta):
    assignumis = importlib.import_module("sicelore_amd.assignumis")
    in_bam, refflat, rows, tail = _make_inputs(pkg, synth, gpu_ctx, tmp_path, meta)
    one = str(tmp_path / "one")
    a = assignumis.assignumis_stream(gpu_ctx, in_bam, one, segment_bytes=9_000, chunk_size=90, n_threads=2, refflat=open(refflat).read())
    many = str(tmp_path / "many")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_assignumis_rank.py"), in_bam, many, refflat, "90"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert p.returncode == 0, p.stderr[-3000:]
    infos = [json.load(open(many + f".info_rank{r}.json")) for r in range(world)]
    assert infos[0]["records"] == a["records"] == len(rows) + len(tail) and infos[0]["gene_keys_order_dependent"] == 0
    assert all(i["records"] > 0 for i in infos[1:3]) and sum(i["records"] for i in infos[1:]) < a["records"]      # every rank had a share of its own
    assert all(i["records"] == 0 for i in infos[3:])                                                           # ... or none, and said so
    for name in (".bam", "_umifound_.bam"):
        got, want = bammodel.bgzf_decompress(open(many + name, "rb").read()), bammodel.bgzf_decompress(open(one + name, "rb").read())
        if got != want:
            from test_bam import _parse_aux
            _, _, ro = bammodel.parse_bam(want)
            _, _, rm = bammodel.parse_bam(got)
            diff = [(x["name"], [(t, v) for t, _ty, v in _parse_aux(x["aux"])], [(t, v) for t, _ty, v in _parse_aux(y["aux"])])
                    for x, y in zip(ro, rm) if x != y][:2]
            raise AssertionError((name, len(ro), len(rm), [x["name"] for x in ro] == [y["name"] for y in rm], diff))
        assert got == want, name                      # header, every record with its tags, the order
        assert not os.path.exists(many + name + ".shard0")
    for name in (".genecounts.tsv", ".UMIdepths.tsv"):
        assert open(many + name).read() == open(one + name).read(), name
    assert a["clustered"] > 50 and open(one + ".genecounts.tsv").read().count("\n") > 10


@pytest.mark.gpu
def test_command_line_under_torchrun_equals_one_process(pkg, synth, gpu_ctx, tmp_path):
    """the jar's `assignumis` line started once per rank (torch.distributed.run, two ranks on the box's one GPU, gloo): cli.py joins the process
    group it finds in the environment, the ranks take their chromosomes, rank 0 leaves the files one process leaves"""
    in_bam, refflat, rows, tail = _make_inputs(pkg, synth, gpu_ctx, tmp_path, True)
    pkg_dir = os.path.join(ROOT, "sicelore-2.1_amd")
    args = ["assignumis", "--inFileNanopore", in_bam, "--annotationFile", refflat]
    env = dict(os.environ, OMP_NUM_THREADS="2", SMI_DIST_BACKEND="gloo")
    one = subprocess.run([sys.executable, pkg_dir] + args + ["-o", str(tmp_path / "cli_one.bam")], capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          pkg_dir] + args + ["-o", str(tmp_path / "cli_two.bam")], capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    for suffix in (".bam", "_umifound_.bam"):
        a, b = (bammodel.bgzf_decompress(open(str(tmp_path / f"cli_{k}") + suffix, "rb").read()) for k in ("one", "two"))
        assert a == b and len(a) > 10_000, suffix
    for suffix in (".genecounts.tsv", ".UMIdepths.tsv"):
        assert open(str(tmp_path / "cli_one") + suffix).read() == open(str(tmp_path / "cli_two") + suffix).read(), suffix
    assert not any(".shard" in f for f in os.listdir(tmp_path))


@pytest.mark.gpu
def test_shards_without_a_process_group_then_merge_shards(pkg, synth, gpu_ctx, tmp_path):
    """shard=(rank, world) runs one after the other (a scheduler's jobs, no torch.distributed between them), merge_shards afterwards: the
    single process's files; three shards of a three-chromosome BAM, then five (two of them without a range)"""
    assignumis = importlib.import_module("sicelore_amd.assignumis")
    in_bam, refflat, rows, tail = _make_inputs(pkg, synth, gpu_ctx, tmp_path, False)
    text = open(refflat).read()
    one = str(tmp_path / "one")
    a = assignumis.assignumis_stream(gpu_ctx, in_bam, one, segment_bytes=9_000, chunk_size=90, n_threads=2, refflat=text)
    for world in (3, 5):
        many = str(tmp_path / f"many{world}")
        infos = [assignumis.assignumis_stream(gpu_ctx, in_bam, many, segment_bytes=9_000, chunk_size=90, n_threads=2, refflat=text, shard=(r, world))
                 for r in range(world)]
        assert sum(i["records"] for i in infos) == a["records"] and [i["records"] for i in infos[3:]] == [0] * (world - 3)
        with pytest.raises(pkg.SmiError, match="missing"):
            assignumis.merge_shards(many, world + 1)
        m = assignumis.merge_shards(many, world)
        assert m["gene_keys_order_dependent"] == 0
        for name in (".bam", "_umifound_.bam"):
            assert bammodel.bgzf_decompress(open(many + name, "rb").read()) == bammodel.bgzf_decompress(open(one + name, "rb").read()), name
        for name in (".genecounts.tsv", ".UMIdepths.tsv"):
            assert open(many + name).read() == open(one + name).read(), name
    assert not any(".shard" in f for f in os.listdir(tmp_path))


@pytest.mark.gpu
def test_stale_or_foreign_index_is_refused(pkg, synth, gpu_ctx, tmp_path):
    """a BAM shorter than its index says (truncated file, stale .bai) and an index of another reference count stop the run with a message
    instead of a reader that waits for bytes that never come"""
    assignumis = importlib.import_module("sicelore_amd.assignumis")
    in_bam, refflat, rows, tail = _make_inputs(pkg, synth, gpu_ctx, tmp_path, True)
    whole = open(in_bam, "rb").read()
    ext = assignumis.bai_ref_extents(in_bam + ".bai")
    cut = str(tmp_path / "cut.bam")
    with open(cut, "wb") as f:                       # ends inside the second chromosome: rank 0's range is whole, the index points behind the end
        f.write(whole[:(ext[1][0] >> 16) + 100])
    os.link(in_bam + ".bai", cut + ".bai")
    with pytest.raises(pkg.SmiError, match="behind the end|truncated"):
        assignumis.assignumis_stream(gpu_ctx, cut, str(tmp_path / "o1"), n_threads=2, shard=(0, 2))
    other = str(tmp_path / "other.bam")
    os.link(in_bam, other)
    with open(other + ".bai", "wb") as f:
        f.write(bammodel.bai_bytes(2, [], meta=False))
    with pytest.raises(pkg.SmiError, match="not this file's index"):
        assignumis.assignumis_stream(gpu_ctx, other, str(tmp_path / "o2"), n_threads=2, shard=(1, 2))
