"""config-5 shape, 400 targets: sensitivity of the stage times to the number of merge / bestPath pieces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pbdagcon_amd import capi, synth
tl = np.random.default_rng(5).integers(2000, 40000, 400)
b5 = synth.make_batch(400, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True)
for ms, msl in ((0, 0), (32, 0), (64, 0), (64, 384), (32, 384)):
    ctx = capi.Context(min_cov=6, min_len=500, trim=10, max_segments=ms, min_segment_len=msl)
    ctx.upload(b5); ctx.run(); ctx.fetch(); ctx.run(); r = ctx.fetch()
    t = ctx.timings()
    bases = sum(len(s) for x in r for _, _, s in x)
    print("max_segments", ms, "min_segment_len", msl, {k: round(v, 2) for k, v in t.items() if k.startswith("ms_")}, "segments", t["merge_segments"],
          f"{bases / t['ms_total'] / 1e3:.1f} M bases/s", flush=True)
    ctx.close()
