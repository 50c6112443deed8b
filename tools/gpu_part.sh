#!/bin/bash
# one part of the GPU work in one call -- usage: gpu_part.sh <part> ...
#   tests <pytest paths ...>   the named GPU tests (-x -q -m gpu)
#   cross                      the splitter's generations against each other (tools/chim_crosscheck.py, 0.27 M 5' and 0.9 M 3' reads), K-SCAN's shipped
#                              kernels against its generic ones (tools/scan_crosscheck.py, 2 M reads in each of four modes)
#   fuzz [minutes]             tools/fuzz_parity.py, legs bc + records
#   scancross                  tools/scan_crosscheck.py alone; fuzzscan [minutes [seed]]: the fuzz's scan leg
#   step                       the bench's timed step and its end-to-end leg, no other leg
#   e2e                        kernel trace of the end-to-end leg -> gpurun_out/e2e_kernel_stats.csv, e2e_timeline.json
#   own                        tools/own_cluster_bench.py 8000 with the clusterer's step timer -> gpurun_out/own_cluster_8000*.json
# Parts run in the order given; the call stops at the first one that fails.
set -u
TAG=${TAG:-r06}   # the round the outputs are named after (profiles/$TAG/ once copied there)
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
OFF="--two-pass-reads 0 --umi-molecules 0 --h2h-reads 0 --f2f-reads 0 --assignumis-file-records 0"
while [ $# -gt 0 ]; do
  part=$1; shift
  case $part in
    tests)
      paths=()
      while [ $# -gt 0 ] && [[ $1 == tests/* ]]; do paths+=("$1"); shift; done
      timeout -k 10 1000 python -m pytest "${paths[@]}" -x -q -m gpu > gpurun_out/${TAG}_part_tests.log 2>&1; rc=$?
      tail -6 gpurun_out/${TAG}_part_tests.log | cut -c1-400; [ $rc -eq 0 ] || exit $rc ;;
    cross)
      timeout -k 10 300 python tools/chim_crosscheck.py 300000 5p 2> gpurun_out/chim_cross5.err | cut -c1-700 || exit 1
      timeout -k 10 300 python tools/chim_crosscheck.py 1000000 2> gpurun_out/chim_cross3.err | cut -c1-700 || exit 1
      timeout -k 10 600 python tools/scan_crosscheck.py 2000000 2> gpurun_out/scan_cross.err | cut -c1-700 || exit 1 ;;
    scancross)   # K-SCAN's shipped kernels against its generic ones only
      timeout -k 10 600 python tools/scan_crosscheck.py 2000000 2> gpurun_out/scan_cross.err | cut -c1-700 || exit 1 ;;
    fuzzscan)    # the scan leg of the fuzz (random polyA windows; K-SCAN against the oracle): minutes, first seed
      minutes=3; seed=61000
      if [ $# -gt 0 ] && [[ $1 =~ ^[0-9.]+$ ]]; then minutes=$1; shift; fi
      if [ $# -gt 0 ] && [[ $1 =~ ^[0-9]+$ ]]; then seed=$1; shift; fi
      SMI_FUZZ_LEGS=scan timeout -k 10 900 python tools/fuzz_parity.py $minutes $seed > gpurun_out/${TAG}_part_fuzzscan.log 2>&1; rc=$?
      tail -2 gpurun_out/${TAG}_part_fuzzscan.log | cut -c1-300; [ $rc -eq 0 ] || exit $rc ;;
    fuzz)
      minutes=1.5
      if [ $# -gt 0 ] && [[ $1 =~ ^[0-9.]+$ ]]; then minutes=$1; shift; fi
      SMI_FUZZ_LEGS=r2 timeout -k 10 900 python tools/fuzz_parity.py $minutes 4000 > gpurun_out/${TAG}_part_fuzz.log 2>&1; rc=$?
      tail -3 gpurun_out/${TAG}_part_fuzz.log | cut -c1-300; [ $rc -eq 0 ] || exit $rc ;;
    step)
      timeout -k 10 600 python bench.py --steps 20 --warmup 3 $OFF > gpurun_out/${TAG}_part_bench.json 2> gpurun_out/${TAG}_part_bench.err || exit 1
      TAG=$TAG python3 - <<'PY'
import json, os
d = json.loads(open(f"gpurun_out/{os.environ['TAG']}_part_bench.json").read().strip().splitlines()[-1])
print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "k_scan_ms": d["roofline"]["kernel_ms"], "e2e_ms": d["end_to_end"]["ms"],
                  "cpu_matches": d["cpu_baseline"]["matches_gpu"]}))
PY
      ;;
    e2e)
      rm -rf gpurun_out/prof_e2e
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_e2e" -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-lanes 1 $OFF > "$ROOT/gpurun_out/prof_e2e.log" 2>&1) || exit 1
      f=$(find gpurun_out/prof_e2e -name "*kernel_stats.csv" | head -1)
      (head -1 "$f"; grep "smi::" "$f") > gpurun_out/e2e_kernel_stats.csv
      python3 tools/e2e_timeline.py gpurun_out/prof_e2e gpurun_out/e2e_timeline.json
      find gpurun_out/prof_e2e -name "*.csv" -size +1M -delete ;;
    own)
      SMI_AU_TIMING=1 timeout -k 10 300 python tools/own_cluster_bench.py 8000 > gpurun_out/own_cluster_8000.json 2> gpurun_out/own_cluster_8000.err || exit 1
      python3 tools/own_cluster_profile.py gpurun_out/own_cluster_8000 gpurun_out/own_cluster_8000_profile.json ;;
    *) echo "unknown part $part"; exit 2 ;;
  esac
done
