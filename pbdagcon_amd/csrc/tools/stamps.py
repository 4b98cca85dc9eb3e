import sys, os
sys.path.insert(0, os.getcwd())
from pbdagcon_amd import capi, synth
capi.LIB_PATH = os.path.join(os.getcwd(), "pbdagcon_amd", "libdagcon_hip_stamps.so")
b = synth.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 10000, 40, seed=1000)
ctx = capi.Context(min_cov=6, min_len=500, trim=50)
ctx.upload(b); ctx.run(); ctx.fetch(); ctx.run(); ctx.fetch()
print(ctx.timings())
d = ctx.debug_counters()
print("dbg", d)
