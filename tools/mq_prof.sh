set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_mq
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export DAGCON_MERGE_Q=1 DAGCON_MERGE_SEGS=32
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/insts -o insts -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-verify --no-legs > $OUT/insts.log 2>&1
python3 - <<PY
import csv, collections
tot=collections.defaultdict(float); n=collections.defaultdict(int)
import glob
f=glob.glob("$OUT/insts/**/*counter_collection.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    k=row["Kernel_Name"].split("(")[0]
    if "merge" in k:
        tot[(k,row["Counter_Name"])]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
for k in sorted(tot): print(k, tot[k]/n[k]/1e9)
PY
