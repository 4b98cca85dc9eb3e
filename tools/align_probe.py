"""Throughput of the -a stage (dagcon_align): n pairs of L-base reads against their targets (synthetic edits)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pbdagcon_amd import capi, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 8
b = synth.make_batch(nt, L, cov, seed=7000)
pairs = []
for t in range(b.n_targets):
    for start, q, tt in b.target_alignments(t):
        pairs.append((q.replace(b"-", b""), tt.replace(b"-", b"")))
ctx = capi.Context(min_cov=8, min_len=500, trim=50)
ctx.align(pairs[:4])
for rep in range(2):
    t0 = time.perf_counter()
    out = ctx.align(pairs)
    dt = time.perf_counter() - t0
    cells = sum((2 * capi_w + 1) * len(q) for (q, t2), capi_w in ((p, min(480, 32 + 4 * int((15 * max(len(p[0]), len(p[1])) + 99) // 100) ** 0.5 // 1)) for p in pairs))
    bases = sum(len(q) for q, _ in pairs)
    print(f"{len(pairs)} pairs of ~{L} bases: {dt * 1e3:.1f} ms, {bases / dt / 1e6:.1f} M read bases/s, ~{cells / dt / 1e9:.1f} G cells/s", flush=True)
ctx.upload(b); ctx.run(); ctx.fetch(); ctx.run(); ctx.fetch()
print("consensus of the same batch on the device:", round(ctx.timings()["ms_total"], 2), "ms")
