#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json configs[1]:
10 M synthetic Nanopore reads (16-bp BC + 12-bp UMI, ~Q12 error profile), ed <= 1 against the 3.6 M whitelist,
one MI355X per rank.  A step = one pass of the hot path over the rank's batch, inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

# algorithmic bytes per read of the ed <= 1 matcher = 16 (window) + 16 (result) + 4 x 620 probes
ALG_BYTES_PER_READ_BC1 = 2512
# K-PA + K-AD scan: 2 x 175 bases (1 B/base as the reference holds them) + 32 B result
ALG_BYTES_PER_READ_SCAN = 382
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step (configs[1]: 10 M)")
    ap.add_argument("--whitelist", type=int, default=3_600_000)
    ap.add_argument("--cells", type=int, default=5000)
    ap.add_argument("--cpu-sample", type=int, default=200_000, help="reads of the same workload timed on the host")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e-reads", type=int, default=500_000,
                    help="reads of the bounded end-to-end leg (FASTQ text -> passed/failed text), reported beside `value`; 0 = skip")
    return ap.parse_args()


def end_to_end_leg(ctx, synth, dev, used, n):
    """Whole pass 2 of `scanfastq` for one chunk, FASTQ text resident in HBM -> finished `passed` / `failed` FASTQ text in
    HBM, every stage on the device: K-FQ (index + two gathers), K-PACKR + K-CHIM + fragment offsets, K-PACK, K-SCAN,
    K-BC1 (same 3.6 M whitelist as the step), K-WRITE.  Reported beside `value`, never part of it."""
    rd = synth.gen_reads(n, used, seed=77, device=dev)
    text, _buf, offs0 = synth.fastq_text_device(rd)
    del rd
    total_text, total_bases = int(text.numel()), int(offs0[-1])
    cap = n + 2
    i64 = lambda k: torch.zeros(k, dtype=torch.int64, device=dev)  # noqa: E731
    i32 = lambda k: torch.zeros(k, dtype=torch.int32, device=dev)  # noqa: E731
    u8 = lambda k: torch.zeros(k, dtype=torch.uint8, device=dev)  # noqa: E731
    line, ns, ss, qs, offs = i64(4 * cap + 8), i64(cap), i64(cap), i64(cap), i64(cap + 1)
    nl, sl = i32(cap), i32(cap)
    reads, quals = u8(total_bases), u8(total_bases)
    planes = i32(ctx.read_planes_words(total_bases, n))
    d_chim = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    scratch, nfrag, foffs, fsrc = i32((n + 1023) // 1024 + 1), i64(1), i64(3 * n + 1), i32(3 * n)
    m_cap = 3 * n
    ends = torch.zeros((28, 2 * m_cap), dtype=torch.int32, device=dev)
    lens, qsum = i32(m_cap), i32(m_cap)
    qt = torch.zeros((m_cap, 224), dtype=torch.uint8, device=dev)
    scan = torch.zeros((m_cap, 8), dtype=torch.int32, device=dev)
    win = torch.zeros((m_cap, 2), dtype=torch.int64, device=dev)
    bc = torch.zeros((m_cap, 4), dtype=torch.int32, device=dev)
    capw = 2 * total_bases + total_text + 320 * m_cap
    out_p, out_f = torch.empty(capw, dtype=torch.uint8, device=dev), torch.empty(capw, dtype=torch.uint8, device=dev)
    rec_off, is_p = i64(m_cap + 1), u8(m_cap)
    chim_cfg, scan_cfg = ctx.chimera_config(False), ctx.scan_config(2)
    state = {}

    def run():
        nr, err = ctx.fastq_index_device(text, total_text, line, ns, nl, ss, sl, qs, offs, cap)
        assert nr == n and err == 0
        ctx.fastq_gather_device(text, ss, offs, n, reads)
        ctx.fastq_gather_device(text, qs, offs, n, quals)
        ctx.pack_reads_device(reads, offs, n, total_bases, planes)
        ctx.chimera_device(planes, offs, n, total_bases, chim_cfg, d_chim)
        ctx.split_offsets_device(d_chim, offs, n, scratch, nfrag, foffs, fsrc)
        m = int(nfrag.item())
        ctx.pack_ends_device(reads, quals, foffs, m, ends, lens, qt, qsum)
        ctx.scan_device(ends, lens, m, scan_cfg, scan, win, qt, qsum)
        ctx.bc_match_device(win, bc, m, max_ed=1, five_prime=False)
        state["tot"] = ctx.fastq_write_device(text, line, reads, quals, foffs, fsrc, d_chim, scan, bc, None, m, 1, out_p, out_f,
                                              rec_off, is_p)
        state["m"] = m

    run()
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"reads": n, "records_out": state["m"], "passed": state["tot"][2], "text_in_bytes": total_text,
            "text_out_bytes": state["tot"][0] + state["tot"][1], "ms": dt * 1e3, "reads_per_s": n / dt,
            "stages": "K-FQ, K-PACKR, K-CHIM, fragment offsets, K-PACK, K-SCAN, K-BC1 (3.6M whitelist), K-WRITE; FASTQ text in HBM -> "
                      "passed/failed FASTQ text in HBM; host work between the launches included"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the hot path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    pkg = graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    ctx = pkg.Context(local_rank)

    # ---- inputs (synthetic, seeded; built on the device in chunks, resident in HBM before the timed region) ----
    n = args.reads
    wl = synth.make_whitelist(args.whitelist, seed=1, device=dev)           # same list on every rank
    used = synth.pick_used(wl, args.cells, seed=2)
    ctx.set_barcode_set_device(wl.to(torch.int32), mode=1)                   # -g semantics: search set = whole list
    ends = torch.empty((28, 2 * n), dtype=torch.int32, device=dev)           # packed read ends (bit-planes)
    lens = torch.empty(n, dtype=torch.int32, device=dev)
    truth = torch.empty(n, dtype=torch.int64, device=dev)
    chunk = 1_000_000
    cpu_reads = None
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        rd = synth.gen_reads(m, used, seed=1000 + 97 * rank + c0 // chunk, device=dev)   # each rank its own reads
        ends[:, 2 * c0:2 * (c0 + m)] = synth.pack_ends(rd["head"], rd["tail"])
        lens[c0:c0 + m] = (2 * synth.END_BASES + rd["mid_len"]).to(torch.int32)
        truth[c0:c0 + m] = rd["truth"]
        if c0 == 0 and rank == 0:
            k = min(args.cpu_sample, m)
            cpu_reads = {key: (v[:k].cpu() if torch.is_tensor(v) else v) for key, v in rd.items()}
        del rd
    scan_cfg = ctx.scan_config(2)
    scan_out = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    win = torch.zeros((n, 2), dtype=torch.int64, device=dev)
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step():
        # pass 2 of scanfastq for one batch: polyA/adapter scan -> barcode windows -> ed<=1 match + best/second rule
        ctx.scan_device(ends, lens, n, scan_cfg, scan_out, win)
        ctx.bc_match_device(win, out, n, max_ed=1, five_prime=False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.set_timing(True)
    scan_ms, match_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP events were recorded on the launch stream around each kernel; reading them waits for this step's
        # kernels (one sync per step inside the timed region -- conservative)
        scan_ms.append(ctx.kernel_ms(ctx.K_SCAN))
        match_ms.append(ctx.kernel_ms(ctx.K_BC_MATCH))
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ctx.set_timing(False)

    found = (out[:, 2] & 0xFF) == 1
    n_found = int(found.sum().item())
    acc = float(((out[:, 0].to(torch.int64) & 0xFFFFFFFF)[found] == truth[found]).float().mean().item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = n * world * args.steps / elapsed
    k_scan, k_match = float(np.mean(scan_ms)), float(np.mean(match_ms))
    # dominant kernel = the longer of the two.  K-SCAN (bit-parallel gates + Needleman-Wunsch cells on packed read ends) is
    # bound by integer VALU issue, a bound the contract's enum has no name for: its HBM figures are reported as asked
    # (algorithmic bytes) and the issue-rate figure beside them; K-BC1, the memory-side kernel the
    # metric is named after, is in `roofline.other` with the same fields.
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    tj = {}
    if os.path.exists(pmc):
        try:
            tj = json.load(open(pmc))
        except Exception:
            tj = {}

    def kernel_fields(name, key, ms, alg):
        ach = alg * n / (ms * 1e-3) / 1e9
        d = {"kernel": name, "kernel_ms": ms, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "alg_bytes_per_read": alg, "traffic": tj.get(key, {}).get("hbm_bytes_per_launch")}
        vi = tj.get(key, {}).get("valu_insts_per_launch")
        if vi and tj.get(key, {}).get("reads_per_launch") == n:
            # one wave-wide integer VALU instruction occupies a SIMD for 4 cycles (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.0
            # quad-cycle on these kernels): peak = 256 CUs x 4 SIMDs x 2.4 GHz / 4
            peak = 256 * 4 * 2.4 / 4
            d["valu_issue"] = {"achieved_ginst_s": vi / (ms * 1e-3) / 1e9, "peak_ginst_s": peak, "insts_per_launch": vi,
                               "frac": vi / (ms * 1e-3) / 1e9 / peak}
        return d

    f_scan = kernel_fields("k_scan<10>", "k_scan", k_scan, ALG_BYTES_PER_READ_SCAN)
    f_bc1 = kernel_fields("k_bc_match_ed1<1>", "k_bc_match_ed1", k_match, ALG_BYTES_PER_READ_BC1)
    f_scan["limiter"] = "integer VALU issue (bit-parallel gates, Needleman-Wunsch cells): not hbm, not mfma"
    f_bc1["limiter"] = "dependent 4/8-byte gathers into the barcode pyramid (L2 / Infinity Cache / HBM request rate)"
    dom, oth = (f_scan, f_bc1) if k_scan >= k_match else (f_bc1, f_scan)
    n_adapter = int(((scan_out[:, 6] >> 16) & 0xFF).eq(1).sum().item())
    res = {
        "metric": "Nanopore reads/sec BC-assigned at ed<=1, 3.6M whitelist",
        "value": value,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": f"synthetic ({synth.GENERATOR_VERSION}, seeds wl=1 used=2 reads=1000+97*rank+chunk, err=6.3% 40/30/30 sub/ins/del, "
                "50% reverse strand, read length 448..1948)",
        "config": {
            "workload": "configs[1]: 10M synthetic Nanopore reads, ed<=1 vs 3.6M whitelist (-g semantics), 3' protocol; "
                        "timed = pass 2 per read from packed read ends in HBM: polyA/T finder + k-mer gated NW adapter scan "
                        "+ TSO scan + strand decision (K-SCAN) -> 5-offset barcode match + best/second rule (K-BC1); "
                        "not in the step: FASTQ decode/packing, chimera split, UMI stage",
            "reads_per_gpu": n,
            "whitelist": int(wl.numel()),
            "cells": args.cells,
            "adapter_found_frac": n_adapter / n,
            "bc_assigned_frac": n_found / n,
            "bc_assigned_accuracy": acc,
        },
        "roofline": dict({"bound": "hbm"}, **dom, **{"kernels_ms": {"k_scan<10>": k_scan, "k_bc_match_ed1<1>": k_match},
                                                   "other": {oth["kernel"]: oth}, "probes_per_s_bc1": 620.0 * n / (k_match * 1e-3)}),
    }
    if world == 1 and args.e2e_reads > 0:
        res["end_to_end"] = end_to_end_leg(ctx, synth, dev, used, args.e2e_reads)
    if world == 1 and not args.no_cpu_baseline:
        sor = graft.load_oracle()
        sor.build()
        cores = os.cpu_count() or 1
        bset = sor.BarcodeSet(wl.cpu().numpy())
        m = int(cpu_reads["head"].shape[0])
        seqs, quals = zip(*(synth.materialize(cpu_reads, i) for i in range(m)))
        offs = np.zeros(m + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(s_) for s_ in seqs])
        ra = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
        t0 = time.perf_counter()
        st, sc = sor.scan_batch_3p(ra, None, offs, synth.ADAPTER_3P_SHORT, n_threads=cores)
        # stranded barcode regions for assignBarcode: reuse the oracle on windows cut out of the stranded read
        codes = np.full((m, 64), 4, dtype=np.uint8)
        ae = np.zeros(m, dtype=np.int32)
        lut = np.full(256, 4, dtype=np.uint8)
        for ch, v in ((b"A", 0), (b"G", 1), (b"C", 2), (b"T", 3)):
            lut[ch[0]] = v
        comp = bytes.maketrans(b"ACGTN", b"TGCAN")
        for i in range(m):
            if not sc["adapter_found"][i]:
                continue
            seq = bytes(ra[int(offs[i]):int(offs[i + 1])])
            stranded = seq.translate(comp)[::-1] if sc["reverse"][i] else seq
            a = int(sc["adapter_end"][i])
            lo = max(a - 30, 0)
            seg = np.frombuffer(stranded[lo:a + 2], dtype=np.uint8)
            codes[i, :seg.size] = lut[seg]
            ae[i] = a - lo
        st2, exp = sor.assign_batch(bset, codes, ae, max_ed=1, n_threads=cores)
        dt = time.perf_counter() - t0
        got = out[:m].cpu().numpy().view(pkg.BC_RESULT_DTYPE).reshape(-1)
        exp_found = np.where(sc["adapter_found"] == 1, np.where(st2 < 0, -1, exp["found"]), -1)
        same = bool((got["found"] == exp_found).all() and
                    (got["bc"][exp_found == 1] == exp["bc"][exp_found == 1].astype(np.uint32)).all())
        res["cpu_baseline"] = {
            "value": m / dt,
            "unit": "reads/s",
            "cores": cores,
            "kind": "port",
            "sample": f"first {m} reads of rank 0's batch (materialised as ASCII), oracle/sor_scan.c + sor_bc.c "
                      f"(C restatement, OpenMP x{cores}; includes a Python loop cutting the stranded windows); the Java "
                      "reference cannot run here (no JVM); README quotes 20.8k reads/s on 96 cores for the whole scan",
            "seconds": dt,
            "matches_gpu": same,
        }
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
