"""Device time at configs[1] against the number of merge pieces per target."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbdagcon_amd import capi, synth
b = synth.make_batch(1000, 10000, 40, seed=1000)
for ms in (8, 10, 12, 13, 16, 24, 32):
    ctx = capi.Context(min_cov=6, min_len=500, trim=50, max_segments=ms)
    ctx.upload(b); ctx.run(); ctx.fetch(); ctx.run(); ctx.fetch(); ctx.run(); ctx.fetch()
    t = ctx.timings()
    print(ms, {k: round(v, 2) for k, v in t.items() if k.startswith("ms_")}, "segments", t["merge_segments"], flush=True)
    ctx.close()
