"""config-5 shape: bestPath pieces per target (DAGCON_BP_SEGS) against batch size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pbdagcon_amd import capi, synth
for n in (400, 4000):
    tl = np.random.default_rng(5).integers(2000, 40000, n)
    b = synth.make_batch(n, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True)
    ref = None
    for segs in ("", "1", "2", "4", "8", "16"):
        if segs: os.environ["DAGCON_BP_SEGS"] = segs
        else: os.environ.pop("DAGCON_BP_SEGS", None)
        ctx = capi.Context(min_cov=6, min_len=500, trim=10)
        ctx.upload(b); ctx.run(); r = ctx.fetch(); ctx.run(); r = ctx.fetch()
        if ref is None: ref = r
        t = ctx.timings()
        print(n, "bp pieces", segs or "default", {k: round(v, 2) for k, v in t.items() if k in ("ms_total", "ms_bestpath")}, "same" if r == ref else "DIFFERENT", flush=True)
        ctx.close()
