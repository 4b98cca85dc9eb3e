#!/usr/bin/env python3
"""End-to-end rate of the pbdagcon CLI host (file -> FASTA): .m5 text of N targets x 10 kb x 40x
written to /tmp, then `pbdagcon_amd/bin/pbdagcon -j 1 file.m5 > out.fa`, wall-clock timed."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pbdagcon_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b = synth.make_batch(n, 10000, 40, seed=1000)
path = ("/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp") + "/e2e.m5"
t0 = time.time()
have = os.environ.get("E2E_KEEP") and os.path.exists(path + f".{n}")
with open(os.devnull if have else path, "wb") as f:
    for t in range(b.n_targets):
        tid = b.ids[t]
        for k, (start, q, tt) in enumerate(b.target_alignments(t)):
            qa, ta = np.frombuffer(q, np.uint8), np.frombuffer(tt, np.uint8)
            nq, nt = int(np.count_nonzero(qa != 45)), int(np.count_nonzero(ta != 45))
            match = np.where(qa == ta, np.uint8(124), np.uint8(42)).tobytes()
            f.write(b"q%07d_%d/0_%d %d 0 %d + %s %d %d %d + -1000 0 0 0 0 254 " % (
                t, k, nq, nq, nq, tid.encode(), int(b.tlen[t]), start - 1, start - 1 + nt))
            f.write(q); f.write(b" "); f.write(match); f.write(b" "); f.write(tt); f.write(b"\n")
rep_n = int(os.environ.get("E2E_REPEAT", "1"))
if rep_n > 1 and not have:            # the same text again and again: ids repeat, but never side by side
    blk = open(path, "rb").read()
    with open(path, "ab") as f:
        for _ in range(rep_n - 1): f.write(blk)
    del blk
if not have and os.environ.get("E2E_KEEP"): open(path + f".{n}", "w").close()
size = os.path.getsize(path)
print(f"wrote {size / 1e6:.1f} MB of .m5 in {time.time() - t0:.1f} s", flush=True)
exe = os.path.join(ROOT, "pbdagcon_amd", "bin", "pbdagcon")
for rep in range(2):
    t0 = time.time()
    out = subprocess.run([exe, "-j", os.environ.get("E2E_J", "8")] + os.environ.get("E2E_ARGS", "").split() + [path],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, PBDAGCON_TIMING="1"))
    dt = time.time() - t0
    bases = len(out.stdout) - sum(len(l) + 1 for l in out.stdout.split(b"\n") if l.startswith(b">")) - out.stdout.count(b"\n") + out.stdout.count(b"\n>") + (1 if out.stdout.startswith(b">") else 0)
    bases = sum(len(l) for l in out.stdout.split(b"\n") if l and not l.startswith(b">"))
    print(f"run {rep}: rc {out.returncode}, {dt:.2f} s wall, {bases} consensus bases, "
          f"{bases / dt / 1e6:.2f} M bases/s end to end, {size / dt / 1e6:.0f} MB/s of .m5", flush=True)
    print(out.stderr.decode()[-500:].strip())
