"""Diagnostic build only (make stamps): cycle stamps of k_merge_tile, tile 2 of target 0."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbdagcon_amd import capi, synth
capi.LIB_PATH = os.path.join(ROOT, "pbdagcon_amd", "libdagcon_hip_stamps.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
b = synth.make_batch(n, 10000, 40, seed=1000)
ctx = capi.Context(min_cov=6, min_len=500, trim=50, flags=capi.FLAG_STOP_AFTER_MERGE)
ctx.upload(b); ctx.run(); ctx.fetch(); ctx.run(); ctx.fetch()
t = ctx.timings()
print(n, "targets", {k: round(v, 2) for k, v in t.items() if k.startswith("ms_")}, "segments", t["merge_segments"])
d = ctx.debug_counters()
print("cycles: load %d, translate %d, stretches %d, sweep %d (lane0 %d), barrier %d, writeback %d" % (d[0], d[1], d[2], d[3], d[14], d[4], d[5]))
print("tile: %d stretches, %d vertices, visits max/lane %d, sum %d, mergeIn calls %d, mergeOut calls %d" % (d[12], d[13], d[8], d[9], d[10], d[11]))
if d[8]:
    print("sweep cycles per visit of the longest lane: %.0f" % (d[3] / d[8]))
