"""Diagnostic build only (make stamps): in-kernel cycle counters of k_merge, segment 0 of target 0."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbdagcon_amd import capi, synth
capi.LIB_PATH = os.path.join(ROOT, "pbdagcon_amd", "libdagcon_hip_stamps.so")
b = synth.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 10000, 40, seed=1000)
ctx = capi.Context(min_cov=6, min_len=500, trim=50, flags=capi.FLAG_STOP_AFTER_MERGE)
ctx.upload(b); ctx.run(); ctx.fetch(); ctx.run(); ctx.fetch()
t = ctx.timings()
print({k: round(v, 2) for k, v in t.items() if k.startswith("ms_")}, "segments", t["merge_segments"])
d = ctx.debug_counters()
names = ["n_fast", "n_slow", "c_fast", "c_slow", "n_scalar", "c_a", "c_b", "c_c", "c_grp", "ng_in", "ng_out",
         "c_odd", "q_a", "q_b", "q_c", "q_d"]
dd = dict(zip(names, d))
print(dd)
nf, ns = max(dd["n_fast"], 1), max(dd["n_slow"], 1)
print("boring visit: %.0f cycles (pop %.0f, rec %.0f, list %.0f, nbr %.0f, rest %.0f)" % (
    dd["c_fast"] / nf, dd["c_odd"] / nf, dd["q_a"] / nf, dd["q_b"] / nf, dd["q_c"] / nf, dd["q_d"] / nf))
print("merge visit: %.0f cycles; groups: %d in + %d out, %.0f cycles each" % (
    dd["c_slow"] / ns, dd["ng_in"], dd["ng_out"], dd["c_grp"] / max(dd["ng_in"] + dd["ng_out"], 1)))
