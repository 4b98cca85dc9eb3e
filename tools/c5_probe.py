"""config-5 shape (partial spans): merge / bestPath time against the number of pieces per target."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pbdagcon_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
tl = np.random.default_rng(5).integers(2000, 40000, n)
b = synth.make_batch(n, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True)
ref = None
for kw in ({}, dict(max_segments=32, min_segment_len=512), dict(max_segments=48, min_segment_len=384), dict(max_segments=64, min_segment_len=256),
           dict(max_segments=64, min_segment_len=128)):
    ctx = capi.Context(min_cov=6, min_len=500, trim=10, **kw)
    ctx.upload(b); ctx.run(); r = ctx.fetch(); ctx.run(); r = ctx.fetch()
    if ref is None: ref = r
    t = ctx.timings()
    print(n, kw or "auto", {k: round(v, 2) for k, v in t.items() if k in ("ms_total", "ms_merge", "ms_bestpath")}, "segments", t["merge_segments"],
          "same" if r == ref else "DIFFERENT", flush=True)
    ctx.close()
