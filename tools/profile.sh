#!/bin/bash
# Profiles of the bench workload (configs[1]) on the GPU box, as MI355X_MICROARCH.md prescribes: one
# rocprofv3 run for the kernel trace / stats, separate --pmc runs for the counters.
#   bash tools/profile.sh <tag>      -> gpurun_out/prof_<tag>/{ks,fetch,write,insts}_*.csv + summaries
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-verify --no-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- $CMD > $OUT/ks.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/insts -o insts -- $CMD > $OUT/insts.log 2>&1
find $OUT -name "*_kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/fetch -name "*counter_collection.csv" -exec cp {} $OUT/pmc_fetch_counter_collection.csv \;
find $OUT/write -name "*counter_collection.csv" -exec cp {} $OUT/pmc_write_counter_collection.csv \;
find $OUT/insts -name "*counter_collection.csv" -exec cp {} $OUT/pmc_insts_counter_collection.csv \;
python3 $R/tools/pmc_summary.py $OUT/pmc_fetch_counter_collection.csv $OUT/pmc_write_counter_collection.csv $OUT/pmc_hbm_traffic.json "$CMD"
python3 - <<PY
import csv, collections, json
tot, n = collections.defaultdict(lambda: [0.0, 0.0]), collections.defaultdict(int)
for row in csv.DictReader(open("$OUT/pmc_insts_counter_collection.csv")):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    i = 0 if row["Counter_Name"] == "SQ_INSTS_VALU" else 1
    tot[k][i] += float(row["Counter_Value"]); n[(k, i)] += 1
out = {k: {"SQ_INSTS_VALU_per_launch": v[0] / max(n[(k, 0)], 1), "SQ_INSTS_SALU_per_launch": v[1] / max(n[(k, 1)], 1)} for k, v in tot.items() if not k.startswith("__amd")}
json.dump(out, open("$OUT/pmc_insts.json", "w"), indent=1)
print(json.dumps({k: round((v["SQ_INSTS_VALU_per_launch"] + v["SQ_INSTS_SALU_per_launch"]) / 1e9, 3) for k, v in out.items()}))
PY
head -30 $OUT/kernel_stats.csv
