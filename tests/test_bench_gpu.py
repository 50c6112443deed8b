"""bench.py end to end at small sizes on the GPU: the three configurations it runs (BASELINE configs[1], [2], [4]) produce one JSON line
with the fields the driver's contract names, and the step's output equals the oracle where the line says so (`cpu_baseline.matches_gpu`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-2000:]
    line = p.stdout.strip().splitlines()[-1]
    return json.loads(line)


def _common(d, steps):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["value"] > 0 and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key


def test_bench_config1_small():
    d = _run("--reads", "300000", "--whitelist", "400000", "--steps", "2", "--warmup", "1", "--cpu-sample", "20000", "--e2e-reads", "30000",
             "--two-pass-reads", "20000")
    _common(d, 2)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["matches_gpu"] is True
    assert d["end_to_end"]["reads_per_s"] > 0 and d["value_full_pass2"] == d["end_to_end"]["reads_per_s"] == d["end_to_end"]["reads_per_s_one_lane"]
    assert d["value_full_pass2_lanes"] == d["end_to_end"]["reads_per_s_best_lanes"] >= 0.9 * d["value_full_pass2"]
    # the one-time set build beside the step: its time (warm / with allocations), the HBM it occupies, and the batch as a job of its own
    assert 0 < d["set_build_ms"] <= d["set_build_cold_ms"] * 1.5 and d["set_hbm_bytes"] > (1 << 30)
    assert 0 < d["value_one_shot"] < d["value"]
    assert d["two_pass"]["collective"]["ranks_counted"] == 1
    assert d["two_pass"]["same_used_list_on_all_ranks"] is True and d["two_pass"]["pass2_assigned"] > 0
    assert set(d["roofline"]["kernels_ms"]) == {"k_scan<10>", "k_bc_match_ed1<1>"}


def test_bench_pack_kernel_gives_the_same_step():
    """--pack-kernel: the step's inputs built by K-PACK from ASCII reads instead of the generator's packer -> the same assignments"""
    common = ("--reads", "200000", "--whitelist", "400000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--e2e-reads", "0", "--two-pass-reads", "0",
              "--umi-molecules", "0", "--h2h-reads", "0", "--f2f-reads", "0")
    a, b = _run(*common), _run(*common, "--pack-kernel")
    assert "K-PACK" in b["config"]["ends_packed_by"] and "torch" in a["config"]["ends_packed_by"]
    for key in ("bc_assigned_total", "adapter_found_frac", "bc_assigned_accuracy"):
        assert a["config"][key] == b["config"][key], key


def test_bench_overlap_gives_the_same_assignments():
    """--overlap: the steps as a two-stage pipeline on two streams (two window buffers, events between the stages) -> the same assignments"""
    common = ("--reads", "200000", "--whitelist", "400000", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--e2e-reads", "0", "--two-pass-reads", "0",
              "--umi-molecules", "0", "--h2h-reads", "0", "--f2f-reads", "0")
    a, b = _run(*common), _run(*common, "--overlap")
    for key in ("bc_assigned_total", "adapter_found_frac", "bc_assigned_accuracy"):
        assert a["config"][key] == b["config"][key], key
    assert b["steps"] == 5 and b["value"] > 0 and len(b["roofline"]["kernels_ms"]) == 2


def test_bench_config2_small():
    d = _run("--config", "2", "--reads", "400000", "--batch", "200000", "--whitelist", "400000", "--steps", "2", "--warmup", "1")
    _common(d, 2)
    assert d["roofline"]["kernel"] == "k_bc_match_ed2" and d["config"]["used_list"] > 1000 and d["config"]["bc_assigned_frac"] > 0.5


def test_bench_config4_small():
    d = _run("--config", "4", "--reads", "300000", "--steps", "2", "--warmup", "1")
    _common(d, 2)
    assert d["config"]["whitelist"] == 737_280 and d["config"]["bc_assigned_frac"] > 0.4 and d["config"]["bc_assigned_accuracy"] > 0.9
    assert d["umi"]["pairs_per_s"] > 0


def test_bench_two_ranks_on_one_gpu_weak_and_strong():
    """the N-rank launch (bench.py starts its ranks itself), sharding, barriers, max-over-ranks timing and both exchanges of the two-pass leg
    on the GPU with two ranks that share the box's one device (SMI_BENCH_SHARE_GPU, gloo: RCCL refuses two ranks on one device) --
    a rehearsal of the driver's multi-GPU run, not a measurement"""
    small = ["--whitelist", "400000", "--steps", "2", "--warmup", "1", "--e2e-reads", "0", "--umi-molecules", "0", "--h2h-reads", "0",
             "--f2f-reads", "0", "--two-pass-reads", "20000"]
    two = ["--gpus", "2", "--backend", "gloo"]
    env = {"SMI_BENCH_SHARE_GPU": "1"}
    w = _run("--reads", "200000", *small, *two, env=env)
    assert w["n_gpus"] == 2 and w["scaling"] == "weak" and w["config"]["reads_total"] == 400_000 and w["value"] > 0
    assert w["two_pass"]["ranks"] == 2 and w["two_pass"]["same_used_list_on_all_ranks"] is True and w["two_pass"]["backend"] == "gloo"
    assert w["two_pass"]["assigned_tsv_total"] >= w["two_pass"]["pass2_assigned"]          # counters summed over both ranks
    st = _run("--total-reads", "2000000", *small, *two, env=env)
    assert st["n_gpus"] == 2 and st["scaling"] == "strong" and st["config"]["reads_total"] == 2_000_000 and st["config"]["reads_per_gpu"] == 1_000_000
    one = _run("--total-reads", "2000000", *small)
    # the same reads whatever the world size: the number of assigned reads of the whole job is identical
    assert one["scaling"] == "strong" and one["config"]["bc_assigned_total"] == st["config"]["bc_assigned_total"] > 1_000_000
