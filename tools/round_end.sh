#!/bin/bash
# What the numbers of a round's last commit come from, in one call on the GPU box: the profile passes of configs[1] and of the
# -a path (tools/profile.sh, tools/profile_pre.sh -> gpurun_out/prof_<tag>/) and the full bench line (gpurun_out/bench_<tag>.json).
#   bash tools/round_end.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-end}
bash $R/tools/profile.sh $TAG > $R/gpurun_out/profile_$TAG.log 2>&1
grep -A12 '^"Name"' $R/gpurun_out/profile_$TAG.log | cut -c1-110
bash $R/tools/profile_pre.sh $TAG > $R/gpurun_out/profile_pre_$TAG.log 2>&1
tail -4 $R/gpurun_out/profile_pre_$TAG.log | cut -c1-110
python3 $R/bench.py > $R/gpurun_out/bench_$TAG.json 2> $R/gpurun_out/bench_$TAG.err
python3 - "$R/gpurun_out/bench_$TAG.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"], d["bit_exact_vs_oracle"], d["targets_verified"])
for k in ("two_contexts", "streamed", "h2d_inclusive", "e2e", "e2e_4000", "e2e_pre", "config5_shape", "configs3_n1"):
    v = d.get(k)
    print(k, round(v["value"] / 1e6, 1) if isinstance(v, dict) and "value" in v else v,
          (v.get("digest_matches_oracle"), v.get("stage_ms")) if k == "config5_shape" else "")
print(d["gpu_over_cpu"], d["gpu_over_cpu_faithful"])
PY
