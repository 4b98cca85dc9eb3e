"""config-5 shape, 400 targets, three runs (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pbdagcon_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
tl = np.random.default_rng(5).integers(2000, 40000, n)
b = synth.make_batch(n, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True)
ctx = capi.Context(min_cov=6, min_len=500, trim=10)
ctx.upload(b)
for _ in range(3):
    ctx.run(); ctx.fetch()
print({k: round(v, 2) for k, v in ctx.timings().items() if k.startswith("ms_")})
