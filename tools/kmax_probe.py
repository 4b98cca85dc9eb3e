"""Where the row sweep (k_merge_q) stops paying: full-span batches of growing depth, k_merge against k_merge_q
(DAGCON_MERGE_Q=0 / 1 with the depth limit lifted: DAGCON_MERGE_SEGS forces the row kernel).   python tools/kmax_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbdagcon_amd import capi, synth
for cov in (40, 50, 60, 70, 80, 100):
    b = synth.make_batch(600, 6000, cov, seed=4000 + cov)
    for q, segs in (("0", None), ("1", "32")):
        os.environ["DAGCON_MERGE_Q"] = q
        if segs: os.environ["DAGCON_MERGE_SEGS"] = segs
        else: os.environ.pop("DAGCON_MERGE_SEGS", None)
        ctx = capi.Context(min_cov=6, min_len=500, trim=50)
        ctx.upload(b); ctx.run(); ctx.fetch(); ctx.run(); ctx.fetch()
        t = ctx.timings()
        print(f"600 x 6 kb x {cov}x  merge_q={q} segs={segs}", {k: round(v, 2) for k, v in t.items() if k in ("ms_total", "ms_merge", "ms_bestpath")}, "pieces", t["merge_segments"], flush=True)
        ctx.close()
