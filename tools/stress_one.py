"""One round of tools/stress.py under the environment given on the command line (bisecting a mismatch):
    python tools/stress_one.py <seed> <round> [VAR=VALUE ...]   -- default settings only (kws[0], no emit shift)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import stress
from pbdagcon_amd import capi
from util import oracle_batch
seed, rnd = int(sys.argv[1]), int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("="); os.environ[k] = v
b, desc, min_cov, min_len, trim, kws = stress.make_round(seed, rnd)
exp = oracle_batch(b, min_cov, min_len, trim)
ctx = capi.Context(min_cov=min_cov, min_len=min_len, trim=trim)
try:
    got = ctx.consensus(b, strict=False)
    st = ctx.target_status.tolist()
except capi.DagconError as e:
    got, st = str(e), None
ctx.close()
bad = got if isinstance(got, str) else [t for t in range(len(exp)) if got[t] != exp[t]]
print(seed, rnd, desc, sys.argv[3:], "status", st, "bad targets", bad, flush=True)
