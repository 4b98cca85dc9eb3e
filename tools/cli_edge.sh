# edge cases of the command line on the GPU box: empty input, one target with several workers, stdin
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from pbdagcon_amd import synth
b = synth.make_batch(3, 1500, 12, seed=5)
open("/tmp/t3.m5", "wb").write(synth.to_m5(b))
open("/tmp/empty.m5", "wb").write(b"")
PY
timeout 60 pbdagcon_amd/bin/pbdagcon --contexts 3 /tmp/empty.m5 | wc -c
timeout 60 pbdagcon_amd/bin/pbdagcon --contexts 3 --devices 0,0 /tmp/t3.m5 | grep -c ">"
timeout 60 pbdagcon_amd/bin/pbdagcon --contexts 2 --batch-targets 1 - < /tmp/t3.m5 | grep -c ">"
timeout 60 pbdagcon_amd/bin/pbdagcon -a --contexts 2 /tmp/empty.m5 | wc -c
echo ok
