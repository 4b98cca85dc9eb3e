"""LDS-tile merge (DAGCON_TILES=1, default) vs wave-per-segment (DAGCON_TILES=0): parity + device time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pbdagcon_amd import capi, synth
from util import oracle_batch, batch_from_targets, random_target

def timed(name, b, reps=3, **kw):
    ctx = capi.Context(**kw)
    ctx.upload(b)
    for _ in range(reps):
        ctx.run(); r = ctx.fetch()
    t = ctx.timings()
    print(name, {k: round(v, 2) for k, v in t.items() if k.startswith("ms_")}, "segments", t["merge_segments"], "reruns", t["reruns"], flush=True)
    ctx.close()
    return r

rng = np.random.default_rng(11)
targets = []
for i in range(120):
    tl = int(rng.integers(20, 400))
    alph = [b"AC", b"ACGT", b"A"][i % 3]
    alns, bb = random_target(rng, tl, int(rng.integers(2, 10)), alphabet=alph, sub=float(rng.uniform(0, 0.1)),
                             ins=float(rng.uniform(0, 0.25)), dele=float(rng.uniform(0, 0.12)), full_span=(i % 4 != 0))
    targets.append((tl, alns, bb))
adv = batch_from_targets(targets)
exp_adv = oracle_batch(adv, 0, 0, 0, 0)
small = synth.make_batch(16, 3000, 24, seed=500)
exp = oracle_batch(small, 6, 500, 50)
part = synth.make_batch(6, 0, 24, seed=32000, min_span=0.5, tlens=np.array([5000, 7000, 9000, 3000, 12000, 6000]))
exp_p = oracle_batch(part, 6, 500, 10)
big = synth.make_batch(1000, 10000, 40, seed=1000)
# mode = "<tiles>:<gcuts>"
modes = [("0", "0"), ("0", "1")]
if len(sys.argv) > 1:
    modes = [tuple(a.split(":")) for a in sys.argv[1:]]
for mode, gc in modes:
    os.environ["DAGCON_TILES"] = mode
    os.environ["DAGCON_GCUTS"] = gc
    print("---- tiles", mode, "gcuts", gc, flush=True)
    r = timed("adv", adv, 1, min_cov=0, min_len=0, trim=0, min_weight=0)
    print("  parity adv:", r == exp_adv, flush=True)
    r = timed("small", small, 1, min_cov=6, min_len=500, trim=50)
    print("  parity small:", r == exp, flush=True)
    r = timed("part", part, 1, min_cov=6, min_len=500, trim=10)
    print("  parity part:", r == exp_p, flush=True)
    rb = timed("configs[1]", big, 3, min_cov=6, min_len=500, trim=50)
    if (mode, gc) == ("0", "0"): ref = rb
    else: print("  same as wave kernel:", rb == ref, flush=True)
