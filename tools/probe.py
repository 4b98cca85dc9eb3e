import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from pbdagcon_amd import capi, synth
from util import oracle_batch
case = sys.argv[1]
if case == "a": b = synth.make_batch(1, 20000, 60, seed=7000); kw = dict(min_cov=8, min_len=500, trim=50)
elif case == "b": b = synth.make_batch(1, 50000, 60, seed=7000); kw = dict(min_cov=8, min_len=500, trim=50)
elif case == "c":
    tl = np.random.default_rng(5).integers(2000, 40000, 6)
    b = synth.make_batch(6, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True); kw = dict(min_cov=6, min_len=500, trim=10)
print("case", case, "generated", b.qstr.size, flush=True)
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = capi.Context(flags=flags, **kw)
t0 = time.time(); ctx.upload(b); print("uploaded", time.time() - t0, flush=True)
ctx.run(); ctx.sync(); print("ran", time.time() - t0, flush=True)
r = ctx.fetch(); print("fetched", time.time() - t0, ctx.timings(), flush=True)
if not flags:
    e = oracle_batch(b, kw["min_cov"], kw["min_len"], kw["trim"])
    print("oracle", time.time() - t0, r == e, flush=True)
