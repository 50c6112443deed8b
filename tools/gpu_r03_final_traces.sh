#!/bin/bash
# round 3, last call: kernel traces of the final build -- the default bench command, the device UMI stage, the packed chunk worker, K-DEFLATE, K-INFLATE
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
trace() {  # tag, program, args...
  tag=$1; shift
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$tag" -- python3 "$@" > "$ROOT/gpurun_out/prof_$tag.log" 2>&1
  cd "$ROOT"
  f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (head -1 "$f"; grep -E "smi::|hipcub|rocprim" "$f") > gpurun_out/${tag}_kernel_stats.csv
  find gpurun_out/prof_$tag -name "*.csv" -size +1M -delete
  echo "== $tag"; cut -c1-160 gpurun_out/${tag}_kernel_stats.csv | head -8
}
trace r03f_bench $ROOT/bench.py --steps 5 --warmup 1 --f2f-reads 0
trace r03f_umi_stage $ROOT/tools/microbench.py assignumis
SMI_MB_READS=200000 trace r03f_packed $ROOT/tools/microbench.py packed
trace r03f_deflate $ROOT/tools/microbench.py deflate
SMI_MB_READS=100000 trace r03f_inflate $ROOT/tools/microbench.py inflate
