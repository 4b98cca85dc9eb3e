"""Is it the merge?  One (seed, round, setting) of tools/stress.py, N runs that stop after mergeNodes; the merged graph of
every target against the oracle's, vertex by vertex; the first vertices that differ, with the cuts around them.
    python tools/merge_check.py <seed> <round> <setting> [N]      (DAGCON_NO_FLOOR=1 lifts the floor under partial-span pieces)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import stress
from pbdagcon_amd import capi
from test_gpu_parity import _oracle_graph
seed, rnd, k = (int(a) for a in sys.argv[1:4])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 6
b, desc, min_cov, min_len, trim, kws = stress.make_round(seed, rnd)
print(desc, kws[k], "trim", trim, "min_len", min_len, flush=True)
exp = {}
for rep in range(N):
    ctx = capi.Context(min_cov=min_cov, min_len=min_len, trim=trim, flags=capi.FLAG_STOP_AFTER_MERGE, **kws[k])
    try:
        ctx.consensus(b, strict=False)
        st = ctx.target_status.tolist()
    except capi.DagconError as e:
        print("run", rep, "error:", e, flush=True); ctx.close(); continue
    nbad = 0
    for t in range(b.n_targets):
        if st[t] != 0:
            print("run", rep, "target", t, "status", st[t]); nbad += 1; continue
        got = ctx.debug_graph(t)
        if t not in exp: exp[t] = _oracle_graph(b, t, min_len, trim, True)
        eg, o2d = exp[t]
        diffs = []
        for o, (eb, ew, ec, ed, eoe, eie) in enumerate(eg):
            g = got[o2d[o]]
            if g["deleted"] != ed: diffs.append((o2d[o], "deleted", ed, g["deleted"])); continue
            if ed: continue
            goe = [(d, c) for d, c in g["out"]]; eoe2 = [(o2d[d], c) for d, c in eoe]
            gie = list(g["inn"]); eie2 = [o2d[x[0]] if isinstance(x, tuple) else o2d[x] for x in eie]
            if g["weight"] != ew or goe != eoe2 or gie != eie2:
                diffs.append((o2d[o], "w", ew, g["weight"], "out", eoe2, goe, "in", eie2, gie))
        if diffs:
            nbad += 1
            diffs.sort()
            print("run", rep, "target", t, "N", len(got), "differing vertices", len(diffs), "first:", diffs[:4], flush=True)
    print("run", rep, "targets that differ:", nbad, "segments", ctx.timings()["merge_segments"], flush=True)
    ctx.close()
