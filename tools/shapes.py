"""Device time of the other BASELINE shapes (aligned strings inward), for DESIGN.md."""
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
run("config3 shape: 8 x 50 kb x 60x full span", synth.make_batch(8, 50000, 60, seed=7000), min_cov=8, min_len=500, trim=50)
run("config3 shape: 128 x 50 kb x 60x full span (chip full)", synth.make_batch(128, 50000, 60, seed=7000), min_cov=8, min_len=500, trim=50)
run("config3 shape, one sweep per target", synth.make_batch(8, 50000, 60, seed=7000), min_cov=8, min_len=500, trim=50, max_segments=1)
tl = np.random.default_rng(5).integers(2000, 40000, 400)
b5 = synth.make_batch(400, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True)
run("config5 shape: 400 x 2-40 kb x 30x, spans >= 60 %", b5, min_cov=6, min_len=500, trim=10)
tl = np.random.default_rng(5).integers(2000, 40000, 4000)
run("config5 shape: 4000 x 2-40 kb x 30x, spans >= 60 %", synth.make_batch(4000, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True), min_cov=6, min_len=500, trim=10)
run("configs[1] at 4000 targets", synth.make_batch(4000, 10000, 40, seed=1000), min_cov=6, min_len=500, trim=50)
