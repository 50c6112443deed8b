"""bench.py --gpus N starts its own ranks (no torchrun around it) and runs the path's exchange: here with world size 2 under gloo
on CPU tensors (`--exchange-only`: the launch, the pass-1 histogram all-reduce + finalize + broadcast, the BarcodesAssigned
counters' all-reduce; no kernels).  On the GPU box the same launcher runs the real two-pass leg over RCCL."""
import json
import os
import subprocess
import sys

import __graft_entry__ as graft


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(graft.ROOT, "bench.py")] + extra, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_launches_two_ranks_itself_and_exchanges():
    graft.build()
    r = _run(["--gpus", "2", "--exchange-only", "--backend", "gloo", "--whitelist", "50000", "--cells", "300"])
    assert r["n_gpus"] == 2 and r["backend"] == "gloo"
    assert r["same_used_list_on_all_ranks"] is True and 100 <= r["used_list"] <= 300
    # counters of rank 0 (1 per used barcode) + rank 1 (2 per used barcode) summed before the file is formatted
    assert r["assigned_rows"] == r["used_list"] and r["first_row_total"] == 3


def test_bench_launches_eight_ranks_and_counts_them():
    """the driver's first 8-GPU run rehearsed as far as a CPU box can: eight ranks (gloo), the process group reports 8, a collective sums over 8
    ranks, every rank holds the same used list, and each rank took its share of the host's CPU quota"""
    r = _run(["--gpus", "8", "--exchange-only", "--backend", "gloo", "--whitelist", "50000", "--cells", "300"])
    assert r["n_gpus"] == 8 and r["backend"] == "gloo"
    c = r["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 8 and c["ranks_counted"] == 8
    assert c["host_threads_per_rank"] >= 1
    assert r["same_used_list_on_all_ranks"] is True and r["first_row_total"] == sum(range(1, 9))


def test_single_rank_exchange_only_needs_no_process_group():
    r = _run(["--gpus", "1", "--exchange-only", "--whitelist", "20000", "--cells", "100"])
    assert r["n_gpus"] == 1 and r["backend"] == "none" and r["first_row_total"] == 1


def test_strong_scaling_split_sums_to_the_one_rank_run():
    """--total-reads: the same 200,000 reads (20 chunks with global seeds) on one rank and sharded over two: the all-reduced histogram, the
    used list after finalize and its ranks are identical (BASELINE configs[3]'s split, distributed.shard_range)"""
    common = ["--exchange-only", "--backend", "gloo", "--whitelist", "50000", "--cells", "300", "--total-reads", "200000"]
    one = _run(["--gpus", "1"] + common)
    two = _run(["--gpus", "2"] + common)
    three = _run(["--gpus", "3"] + common)
    assert one["scaling"] == two["scaling"] == "strong" and one["total_reads"] == 200_000
    assert one["hist_sum"] == two["hist_sum"] == three["hist_sum"] > 90_000
    assert one["used_list"] == two["used_list"] == three["used_list"] > 100
    assert one["used_list_digest"] == two["used_list_digest"] == three["used_list_digest"]
    assert two["same_used_list_on_all_ranks"] is True and three["same_used_list_on_all_ranks"] is True
