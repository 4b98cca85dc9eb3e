"""Which edges of a target's merged graph leave their bestPath piece other than through its upper cut or to exit?
(one round / setting of tools/stress.py; the library's DAGCON_DUMP leaves the target's graph and cuts in a file)
    python tools/bp_pieces.py <seed> <round> <setting> <target>"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import stress
from pbdagcon_amd import capi
seed, rnd, k, t = (int(a) for a in sys.argv[1:5])
b, desc, min_cov, min_len, trim, kws = stress.make_round(seed, rnd)
path = "/tmp/dg_dump.bin"
os.environ["DAGCON_DUMP"] = f"{t}:{path}"
ctx = capi.Context(min_cov=min_cov, min_len=min_len, trim=trim, **kws[k])
try:
    ctx.consensus(b)
    print("consensus ok")
except capi.DagconError as e:
    print("consensus:", e)
raw = open(path, "rb").read()
N, bp_max, psz, seg_max = struct.unpack_from("4I", raw, 0)
off = 16
cuts = np.frombuffer(raw, np.uint32, bp_max + 2, off); off += 4 * (bp_max + 2)
nd = np.frombuffer(raw, np.uint8, 32 * N, off).reshape(N, 32); off += 32 * N
pool = np.frombuffer(raw, np.uint32, psz, off); off += 4 * psz
sc = np.frombuffer(raw, np.float32, 2 * N, off).reshape(N, 2); off += 8 * N
best = np.frombuffer(raw, np.int32, N, off)
nseg = int(cuts[0]); cs = [int(x) for x in cuts[1:1 + nseg]]
print("N", N, "bp_max", bp_max, "pieces", nseg, "cuts", cs[:12], "...")
out_len = nd[:, 0:2].copy().view(np.uint16)[:, 0]; flags = nd[:, 5]; base = nd[:, 4]
out_off = nd[:, 16:20].copy().view(np.uint32)[:, 0]
DELETED, DEFER, SHARED, BACKBONE = 2, 8, 4, 1
piece_of = np.searchsorted(np.array(cs), np.arange(N), side="right") - 1
X = N - 1
nbad = 0
for v in range(N):
    if flags[v] & (DELETED | DEFER): continue          # (deferred vertices are scored on demand, or by k_bp_defer)
    pv = piece_of[v]
    lo = cs[pv]; hi = cs[pv + 1] if pv + 1 < nseg else N - 1
    for e in range(int(out_len[v])):
        d = int(pool[out_off[v] + 2 * e])
        if d == X or (lo <= d <= hi) or sc[d, 1] == 2.0: continue      # (2.0: the exit tree, scored first by k_bp_xtree)
        nbad += 1
        if nbad <= 25:
            print(f"edge {v} (piece {pv} [{lo},{hi}], base {chr(base[v])}, flags {flags[v]:#x}, final {sc[v,1]}) -> {d} (piece {piece_of[d]}, flags {flags[d]:#x}, final {sc[d,1]}, out {int(out_len[d])})")
print("edges that leave their piece otherwise:", nbad, "; unscored live vertices:", int(sum(1 for v in range(N) if not (flags[v] & DELETED) and sc[v, 1] < 1.0)))
