"""configs[1] and other full-span shapes: k_merge (a wave per segment) against k_merge_q (four segments per wave)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbdagcon_amd import capi, synth
for name, b, kw in (("configs[1]", synth.make_batch(1000, 10000, 40, seed=1000), dict(min_cov=6, min_len=500, trim=50)),
                    ("250 x 10 kb x 40x", synth.make_batch(250, 10000, 40, seed=1000), dict(min_cov=6, min_len=500, trim=50)),
                    ("4000 x 10 kb x 40x", synth.make_batch(4000, 10000, 40, seed=1000), dict(min_cov=6, min_len=500, trim=50)),
                    ("128 x 50 kb x 60x", synth.make_batch(128, 50000, 60, seed=7000), dict(min_cov=8, min_len=500, trim=50)),
                    ("8 x 50 kb x 60x", synth.make_batch(8, 50000, 60, seed=7000), dict(min_cov=8, min_len=500, trim=50))):
    ref = None
    for q in ("0", "1"):
        os.environ["DAGCON_MERGE_Q"] = q
        ctx = capi.Context(**kw)
        ctx.upload(b); ctx.run(); r = ctx.fetch(); ctx.run(); r = ctx.fetch()
        if ref is None: ref = r
        t = ctx.timings()
        print(name, "merge_q", q, {k: round(v, 2) for k, v in t.items() if k in ("ms_total", "ms_merge", "ms_bestpath")},
              "segments", t["merge_segments"], "same" if r == ref else "DIFFERENT", flush=True)
        ctx.close()
