#!/bin/bash
# per-kernel split of the set build (profiles/r06/set_build_kernel_stats.csv): TAG names the output
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r06}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/set_build_bench.py 3600000 4 > "$ROOT/gpurun_out/set_build_$TAG.json" 2> "$ROOT/gpurun_out/set_build_$TAG.err" || exit 1
rm -rf "$ROOT/gpurun_out/setb_prof"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/setb_prof" -- python3 $ROOT/tools/set_build_bench.py 3600000 3 > "$ROOT/gpurun_out/set_build_prof_$TAG.json" 2>> "$ROOT/gpurun_out/set_build_$TAG.err" || exit 1
f=$(find "$ROOT/gpurun_out/setb_prof" -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/gpurun_out/set_build_kernel_stats_$TAG.csv"
cat "$ROOT/gpurun_out/set_build_$TAG.json"
head -14 "$ROOT/gpurun_out/set_build_kernel_stats_$TAG.csv" | cut -c1-160
