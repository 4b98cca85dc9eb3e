#!/bin/bash
# per-kernel durations of configs[1] under the given environment:  bash tools/kstats.sh <tag> [VAR=VALUE ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/ks_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-verify --no-legs > $OUT/run.log 2>&1
find $OUT -name "*_kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 - <<PY
import csv
for row in list(csv.DictReader(open("$OUT/kernel_stats.csv")))[:14]:
    print(f"{row['Name'][:60]:60s} {float(row['AverageNs'])/1e6:8.3f} ms x{row['Calls']}")
PY
