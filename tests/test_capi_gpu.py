"""The C-ABI entries a single-process host binds: the RCCL histogram all-reduce inside the library and the host-buffer
forms of the scan and of the UMI distances, against the device entry points / the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_hist_allreduce_over_rccl(pkg, synth):
    """smi_hist_allreduce: one context per visible GPU (one on a single-GPU box: the collective still runs through RCCL)"""
    from sicelore_amd import lib as libmod

    n_gpu = torch.cuda.device_count()
    ctxs = [pkg.Context(d) for d in range(n_gpu)]
    n_keys = 50_000
    hists, total = [], np.zeros(n_keys, dtype=np.int64)
    for d in range(n_gpu):
        h = np.random.default_rng(10 + d).integers(0, 1000, n_keys).astype(np.int32)
        total += h
        hists.append(torch.from_numpy(h).to(f"cuda:{d}"))
    # no synchronize here: hist_allreduce orders every context's stream behind torch's stream on that device (smi_hist_allreduce_after)
    dev_before = torch.cuda.current_device()
    libmod.hist_allreduce(ctxs, hists)
    assert torch.cuda.current_device() == dev_before
    for d, h in enumerate(hists):
        torch.cuda.synchronize(d)
        assert (h.cpu().numpy().astype(np.int64) == total).all()
    libmod.hist_allreduce(ctxs, hists)          # the cached communicators serve the second exchange
    for h in hists:
        assert (h.cpu().numpy().astype(np.int64) == n_gpu * total).all()
    with pytest.raises(libmod.SmiError, match="one context per GPU"):
        libmod.hist_allreduce([ctxs[0], ctxs[0]], [hists[0], hists[0]])
    libmod.hist_allreduce_release()
    for c in ctxs:
        c.close()


def test_hist_allreduce_two_gpus_sum(pkg):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from sicelore_amd import lib as libmod

    ctxs = [pkg.Context(0), pkg.Context(1)]
    a = torch.arange(1000, dtype=torch.int32, device="cuda:0")
    b = torch.full((1000,), 7, dtype=torch.int32, device="cuda:1")
    libmod.hist_allreduce(ctxs, [a, b])
    for d in range(2):
        torch.cuda.synchronize(d)
    exp = np.arange(1000) + 7
    assert (a.cpu().numpy() == exp).all() and (b.cpu().numpy() == exp).all()


def test_scan_batch_from_host_buffers_equals_the_oracle(pkg, synth, sor, gpu_ctx):
    n = 600
    wl = synth.make_whitelist(20000, seed=71)
    reads = synth.gen_reads(n, synth.pick_used(wl, 50, seed=72), seed=73, n_rate=0.002)
    seqs, quals = zip(*(synth.materialize(reads, i) for i in range(n)))
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    ra = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    qa = np.frombuffer("".join(quals).encode(), dtype=np.uint8)
    for pass_no, with_q in ((2, False), (1, True)):
        scan, win = gpu_ctx.scan_batch(ra, qa if with_q else None, offs, gpu_ctx.scan_config(pass_no))
        adapter = synth.ADAPTER_3P_SHORT if pass_no == 2 else "CTACACGACGCTCTTCCGATCT"
        st, exp = sor.scan_batch_3p(ra, qa if with_q else None, offs, adapter, max_mm=3, n_threads=4)
        assert (scan["found"] == exp["adapter_found"]).all() and (scan["adapter_end"] == exp["adapter_end"]).all()
        assert (scan["flags"].astype(np.uint64) == exp["flags"]).all() and (scan["reverse"] == exp["reverse"]).all()
        if with_q:
            assert (scan["pass1_ok"] == exp["pass1_ok"]).all() and scan["pass1_ok"].sum() > 50
        assert win is not None and (win["flags"] & 1).sum() == (scan["found"] == 1).sum()


def test_umi_dist_batch_from_host_buffers_equals_the_oracle(pkg, sor, gpu_ctx):
    rng = np.random.default_rng(5)
    sizes = [1, 2, 7, 30, 3, 64, 5]
    wins = []
    for m in sizes:
        base = rng.choice([1, 2, 4, 8], size=14)
        for _ in range(m):
            w = base.copy()
            for p in rng.integers(0, 14, rng.integers(0, 3)):
                w[p] = rng.choice([1, 2, 4, 8, 15])
            wins.append(w.astype(np.uint8))
    packed = np.array([sum(int(c) << (4 * k) for k, c in enumerate(w)) for w in wins], dtype=np.uint64)
    go = np.zeros(len(sizes) + 1, dtype=np.uint32)
    go[1:] = np.cumsum(sizes)
    got = gpu_ctx.umi_dist_batch(packed, go)
    at = 0
    for g, m in enumerate(sizes):
        exp = sor.umi_matrix(np.array(wins[int(go[g]):int(go[g + 1])], dtype=np.uint8))
        assert (got[at:at + m * m].reshape(m, m) == exp).all(), g
        at += m * m
