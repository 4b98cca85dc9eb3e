#!/bin/bash
# A/B of library builds on the GPU box: every var/lib_<tag>.so takes the place of the shipped library for one short run of
# configs[1] (bit-exactness checked against the oracle's digest) -- stage times side by side.   bash tools/variants.sh [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${1:-6}
cp $R/pbdagcon_amd/libdagcon_hip.so /tmp/lib_head.so
for f in /tmp/lib_head.so $R/var/lib_*.so; do
    cp $f $R/pbdagcon_amd/libdagcon_hip.so
    python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu --no-legs > /tmp/v.json 2> /tmp/v.err || { echo "$f FAILED"; tail -3 /tmp/v.err; continue; }
    python3 - "$f" <<'PY'
import json, sys
d = json.loads(open("/tmp/v.json").read().strip().splitlines()[-1])
st = d.get("stage_ms") or {}
print(f"{sys.argv[1].split('/')[-1]:24s} ms/step {d['ms_per_step']:7.2f}  {d['value']/1e6:6.1f} M  exact={d.get('bit_exact_vs_oracle')}  " + " ".join(f"{k[3:]}={v:.2f}" for k, v in st.items()))
PY
done
cp /tmp/lib_head.so $R/pbdagcon_amd/libdagcon_hip.so
