#!/bin/bash
# instruction counters (VALU + SALU per launch) and kernel durations of configs[1], one pass each
#   bash tools/insts.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
OUT=$R/gpurun_out/insts_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-verify --no-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- $CMD > $OUT/ks.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/insts -o insts -- $CMD > $OUT/insts.log 2>&1
find $OUT -name "*_kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/insts -name "*counter_collection.csv" -exec cp {} $OUT/pmc_insts_counter_collection.csv \;
python3 - <<PY
import csv, collections, json
tot, n = collections.defaultdict(lambda: [0.0, 0.0]), collections.defaultdict(int)
for row in csv.DictReader(open("$OUT/pmc_insts_counter_collection.csv")):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    i = 0 if row["Counter_Name"] == "SQ_INSTS_VALU" else 1
    tot[k][i] += float(row["Counter_Value"]); n[(k, i)] += 1
dur = {}
for row in csv.DictReader(open("$OUT/kernel_stats.csv")):
    dur[row["Name"].split("(")[0].replace("void ", "")] = float(row["AverageNs"]) / 1e6
for k, v in sorted(tot.items(), key=lambda kv: -dur.get(kv[0], 0)):
    if k.startswith("__amd"): continue
    va, sa = v[0] / max(n[(k, 0)], 1) / 1e9, v[1] / max(n[(k, 1)], 1) / 1e9
    print(f"{k:42s} {dur.get(k, 0):8.3f} ms   VALU {va:6.3f} G  SALU {sa:6.3f} G")
PY
