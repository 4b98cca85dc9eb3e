"""The merge's pieces of one target (partial-span worklist) and the vertices around given ids, from a DAGCON_DUMP.
    python tools/merge_cuts.py <seed> <round> <setting> <target> <id> [<id> ...]"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import stress
from pbdagcon_amd import capi
seed, rnd, k, t = (int(a) for a in sys.argv[1:5])
ids = [int(a) for a in sys.argv[5:]]
b, desc, min_cov, min_len, trim, kws = stress.make_round(seed, rnd)
path = "/tmp/dg_dump.bin"
os.environ["DAGCON_DUMP"] = f"{t}:{path}"
ctx = capi.Context(min_cov=min_cov, min_len=min_len, trim=trim, flags=capi.FLAG_STOP_AFTER_MERGE, **kws[k])
ctx.consensus(b, strict=False)
raw = open(path, "rb").read()
N, bp_max, psz, seg_max = struct.unpack_from("4I", raw, 0)
off = 16 + 4 * (bp_max + 2)
nd = np.frombuffer(raw, np.uint8, 32 * N, off).reshape(N, 32); off += 32 * N
pool = np.frombuffer(raw, np.uint32, psz, off); off += 4 * psz + 8 * N + 4 * N
nl = struct.unpack_from("I", raw, off)[0]; off += 4
wl = np.frombuffer(raw, np.uint32, 3 * nl, off).reshape(nl, 3)
mine = [(int(a), int(c)) for tt, a, c in wl if tt == t]
print("target", t, "N", N, "pieces", len(mine))
flags = nd[:, 5]; base = nd[:, 4]; bbpos = nd[:, 28:32].copy().view(np.int32)[:, 0]
out_len = nd[:, 0:2].copy().view(np.uint16)[:, 0]; in_len = nd[:, 2:4].copy().view(np.uint16)[:, 0]
weight = nd[:, 8:12].copy().view(np.int32)[:, 0]
lo, hi = min(ids) - 12, max(ids) + 12
print("pieces around:", [(a, c) for a, c in mine if (c == 0xFFFFFFFF or c >= lo) and a <= hi])
for v in range(max(lo, 0), min(hi, N - 1) + 1):
    print(v, chr(base[v]), "flags", hex(flags[v]), "bbpos", bbpos[v], "w", weight[v], "out", out_len[v], "in", in_len[v])
