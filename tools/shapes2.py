"""config-5 shape (mixed lengths, partial spans, real backbone) at growing batch sizes: device time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pbdagcon_amd import capi, synth
def run(name, b, **kw):
    ctx = capi.Context(**kw)
    ctx.upload(b); ctx.run(); ctx.fetch(); ctx.run(); r = ctx.fetch()
    t = ctx.timings()
    bases = sum(len(s) for x in r for _, _, s in x)
    print(name, {k: round(v, 2) for k, v in t.items() if k.startswith("ms_")}, "segments", t["merge_segments"],
          "bases", bases, f"{bases / t['ms_total'] / 1e3:.1f} M bases/s", flush=True)
    ctx.close()
for n in [int(x) for x in sys.argv[1:]] or [400, 1600, 4000]:
    tl = np.random.default_rng(5).integers(2000, 40000, n)
    b5 = synth.make_batch(n, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True)
    run(f"config5 shape: {n} x 2-40 kb x 30x, spans >= 60 %", b5, min_cov=6, min_len=500, trim=10)
