#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + separate --pmc passes of the bench command, aggregated into
# profiles/<tag>_* by tools/pmc_aggregate.py.  usage: tools/profile_gpu.sh <tag> [bench args...]
# rocprofv3 gets `python3 bench.py ...` directly after `--` (no env / shell hop), one counter group per run.
set -u
TAG=${1:-r01}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --two-pass-reads 0 --e2e-reads 0 $*"
PROG=${PROFILE_PROG:-$ROOT/bench.py}   # e.g. PROFILE_PROG=$PWD/tools/microbench.py tools/profile_gpu.sh tag chimera
[ -n "${PROFILE_PROG:-}" ] && ARGS="$*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $PROG $ARGS > "$OUT/bench_trace.log" 2>&1
DEFAULT_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS;SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU;TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum;TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
IFS=';' read -ra GROUPS_ <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"   # PMC_GROUPS="FETCH_SIZE;WRITE_SIZE": just those passes
for grp in "${GROUPS_[@]}"; do
  name=$(echo "$grp" | tr ' ' '+')
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- python3 $PROG $ARGS > "$OUT/pmc_$name.log" 2>&1 || echo "pmc pass $name failed"
done
cd "$ROOT"
python3 tools/pmc_aggregate.py "$OUT" "$ROOT/gpurun_out/summary_$TAG" "$TAG"
tail -n 2 "$OUT"/bench_trace.log "$OUT"/pmc_FETCH_SIZE.log
# raw per-dispatch CSVs are large (torch kernel names): keep the summaries only
find "$OUT" -name "*.csv" -size +1M -delete
