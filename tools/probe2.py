import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DAGCON_EMIT_SHIFT"] = "4"
import numpy as np
import oracle
from pbdagcon_amd import capi
from util import batch_from_targets, random_target
import test_gpu_parity as T
rng = np.random.default_rng(41)
targets = []
for i in range(40):
    tl = int(rng.integers(20, 200))
    alph = [b"ACGT", b"AC", b"A"][i % 3]
    alns, bb = random_target(rng, tl, int(rng.integers(1, 10)), alphabet=alph, sub=0.05,
                             ins=float(rng.uniform(0.05, 0.3)), dele=float(rng.uniform(0.05, 0.35)),
                             dots=(i % 6 == 0), full_span=(i % 2 == 0))
    targets.append((tl, alns, bb))
batch = batch_from_targets(targets)
ctx = capi.Context(min_cov=0, min_len=0, trim=1, min_weight=0, flags=capi.FLAG_STOP_AFTER_BUILD)
ctx.consensus(batch)
t = 27
got = ctx.debug_graph(t)
exp, o2d = T._oracle_graph(batch, t, 0, 1, False)
d2o = {d: o for o, d in enumerate(o2d)}
blen = int(batch.tlen[t])
print("blen", blen, "reads", len(targets[t][1]))
for o, (eb, ew, ec, ed, eoe, eie) in enumerate(exp):
    g = got[o2d[o]]
    ei = [o2d[s] for s, _ in eie]
    eo = [(o2d[d], c) for d, c in eoe]
    if g["inn"] != ei or g["out"] != eo:
        print("vertex oracle", o, "dev", o2d[o], "bbpos", g["bbpos"], "backbone", g["backbone"])
        print("  in  dev", g["inn"], "exp", ei)
        print("  out dev", g["out"], "exp", eo)
        for s in set(ei) ^ set(g["inn"]):
            gs = got[s]
            print("   src", s, "bbpos", gs["bbpos"], "backbone", gs["backbone"], "out", gs["out"])
for k, (s, q, tt) in enumerate(targets[t][1]):
    qn, tn = oracle.normalize_gaps(q, tt)
    gq, gt, gs = oracle.trim_aln(qn, tn, s, 1)
    print("read", k, "start", gs, "len", len(gq))
    print("  q", gq.decode())
    print("  t", gt.decode())
