#!/bin/bash
# rocprofv3 kernel stats and instruction counters of the -a path (config-3 shape: 64 targets x 50 kb x 60x of .pre
# text through pbdagcon -a):   bash tools/profile_pre.sh <tag>  -> gpurun_out/prof_<tag>/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pre}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
E2E_KEEP=1 python3 $R/tools/e2e_pre.py 64 50000 60 > $OUT/e2e.log 2>&1
F=/dev/shm/e2e.pre
cd /tmp; export TMPDIR=/tmp
export PBDAGCON_TEARDOWN=1      # (the profiler writes its files from exit handlers: the command line must end in order)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- $R/pbdagcon_amd/bin/pbdagcon -a -c 8 -j 16 $F > $OUT/out.fa 2> $OUT/ks.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/insts -o insts -- $R/pbdagcon_amd/bin/pbdagcon -a -c 8 -j 16 $F > /dev/null 2> $OUT/insts.log
rm -f $F $OUT/out.fa
find $OUT -name "*_kernel_stats.csv" -exec cp {} $OUT/pre_kernel_stats.csv \;
find $OUT/insts -name "*counter_collection.csv" -exec cp {} $OUT/pre_pmc_insts_counter_collection.csv \;
head -8 $OUT/pre_kernel_stats.csv
