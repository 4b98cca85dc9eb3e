"""Throughput with one context (run, fetch, run, ...) against two contexts whose batches are in
flight together (own stream each): consecutive steps overlap on the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbdagcon_amd import capi, synth
b = synth.make_batch(1000, 10000, 40, seed=1000)
opts = dict(min_cov=6, min_len=500, trim=50)
K = 12
for nctx in (1, 2, 3):
    ctxs = [capi.Context(**opts) for _ in range(nctx)]
    for c in ctxs:
        c.upload(b); c.run(); ref = c.fetch()
    t0 = time.perf_counter()
    inflight = []
    done = 0
    for i in range(K):
        c = ctxs[i % nctx]
        if len(inflight) == nctx:
            r = inflight.pop(0).fetch(); done += 1
            assert r == ref
        c.run(); inflight.append(c)
    while inflight:
        r = inflight.pop(0).fetch(); done += 1
        assert r == ref
    dt = time.perf_counter() - t0
    print(f"{nctx} context(s): {dt / K * 1e3:.2f} ms per step, {9.9e6 * K / dt / 1e6:.1f} M bases/s", flush=True)
    for c in ctxs:
        c.close()
