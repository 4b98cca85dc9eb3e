"""Experiment: lane-per-stretch merge (DAGCON_LANE_MERGE=1) vs wave-per-segment: parity + device time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pbdagcon_amd import capi, synth
from util import oracle_batch

def timed(name, b, reps=3, **kw):
    ctx = capi.Context(**kw)
    ctx.upload(b)
    for _ in range(reps):
        ctx.run(); r = ctx.fetch()
    t = ctx.timings()
    print(name, {k: round(v, 2) for k, v in t.items() if k.startswith("ms_")}, "segments", t["merge_segments"], "reruns", t["reruns"], flush=True)
    ctx.close()
    return r

small = synth.make_batch(16, 3000, 24, seed=500)
exp = oracle_batch(small, 6, 500, 50)
part = synth.make_batch(6, 0, 24, seed=32000, min_span=0.5, tlens=np.array([5000, 7000, 9000, 3000, 12000, 6000]))
exp_p = oracle_batch(part, 6, 500, 10)
big = synth.make_batch(1000, 10000, 40, seed=1000)
for mode, space in ((0, 16), (1, 8), (1, 16), (1, 32), (1, 64)):
    os.environ["DAGCON_LANE_MERGE"] = str(mode); os.environ["DAGCON_LN_SPACE"] = str(space)
    print("---- lane", mode, "space", space, flush=True)
    r = timed("small", small, 1, min_cov=6, min_len=500, trim=50)
    print("  parity small:", r == exp, flush=True)
    r = timed("part", part, 1, min_cov=6, min_len=500, trim=10)
    print("  parity part:", r == exp_p, flush=True)
    rb = timed("configs[1]", big, 3, min_cov=6, min_len=500, trim=50)
    if mode == 0: ref = rb
    else: print("  same as wave kernel:", rb == ref, flush=True)
