#!/bin/bash
# round 4: HBM traffic counters of the end-to-end leg's kernels (FETCH_SIZE / WRITE_SIZE, a pass each) -> gpurun_out/e2e_pmc.json
set -u
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
OFF="--umi-molecules 0 --h2h-reads 0 --f2f-reads 0 --assignumis-file-records 0"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$ROOT/gpurun_out/prof_e2e_pmc/$c" -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --two-pass-reads 0 $OFF > "$ROOT/gpurun_out/prof_e2e_pmc_$c.log" 2>&1 || echo "pass $c failed"
done
cd "$ROOT"
python3 - <<'PY'
import csv, glob, json, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/prof_e2e_pmc/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c or "smi::" not in r["Kernel_Name"]:
                continue
            nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smi::", "")
            acc[nm].append(float(r["Counter_Value"]))
    for nm, v in acc.items():
        v = v[len(v) // 2:]     # the later dispatches: the timed repetitions
        res[nm][c + "_KB_per_launch"] = sum(v) / len(v)
        res[nm]["launches"] = len(v)
json.dump(res, open("gpurun_out/e2e_pmc.json", "w"), indent=1)
for nm, d in sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE_KB_per_launch", 0) + kv[1].get("WRITE_SIZE_KB_per_launch", 0)))[:14]:
    print(f'{nm[:40]:40s} fetch {d.get("FETCH_SIZE_KB_per_launch", 0)/1e6:8.3f} GB  write {d.get("WRITE_SIZE_KB_per_launch", 0)/1e6:8.3f} GB')
PY
find gpurun_out/prof_e2e_pmc -name "*.csv" -size +1M -delete
