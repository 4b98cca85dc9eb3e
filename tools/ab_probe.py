"""A / B of environment knobs on configs[1] (and optionally the config-5 shape): stage times of the device pipeline.
    python tools/ab_probe.py DAGCON_FOLD=0 DAGCON_FOLD=1 [--c5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hashlib
import numpy as np
from pbdagcon_amd import capi, synth
sets = [a for a in sys.argv[1:] if "=" in a] or [""]
shapes = [("configs[1] 1000x10kx40", synth.make_batch(1000, 10000, 40, seed=1000), dict(min_cov=6, min_len=500, trim=50))]
if "--c5" in sys.argv:
    tl = np.random.default_rng(5).integers(2000, 40000, 400)
    shapes.append(("config5 400 mixed partial", synth.make_batch(400, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True),
                   dict(min_cov=6, min_len=500, trim=10)))
for name, b, kw in shapes:
    for st in sets:
        for kv in st.split(","):
            if kv:
                k, v = kv.split("="); os.environ[k] = v
        ctx = capi.Context(**kw)
        ctx.upload(b); ctx.run(); ctx.fetch()
        best = None
        for _ in range(4):
            ctx.run(); r = ctx.fetch(); t = ctx.timings()
            if best is None or t["ms_total"] < best["ms_total"]:
                best = t
        print(name, st, {k: round(v, 2) for k, v in best.items() if k.startswith("ms_")}, "segs", best["merge_segments"],
              hashlib.sha256(repr(r).encode()).hexdigest()[:12], flush=True)
        ctx.close()
        for kv in st.split(","):
            if kv:
                os.environ.pop(kv.split("=")[0], None)
