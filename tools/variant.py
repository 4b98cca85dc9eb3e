import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbdagcon_amd import capi, synth
capi.LIB_PATH = os.path.join(ROOT, "pbdagcon_amd", sys.argv[1])
b = synth.make_batch(1000, 10000, 40, seed=1000)
ctx = capi.Context(min_cov=6, min_len=500, trim=50)
ctx.upload(b); ctx.run(); r0 = ctx.fetch(); ctx.run(); r1 = ctx.fetch()
assert r0 == r1
t = ctx.timings()
import hashlib
hh = hashlib.sha256(repr(r1).encode()).hexdigest()[:16]
print(sys.argv[1], os.environ.get("DAGCON_MERGE_SEGS"), os.environ.get("DAGCON_PF_AHEAD"), hh, {k: round(v, 2) for k, v in t.items() if k.startswith("ms_")}, sum(len(s) for x in r1 for _, _, s in x))
