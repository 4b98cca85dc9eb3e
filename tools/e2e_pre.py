#!/usr/bin/env python3
"""End-to-end rate of `pbdagcon -a` (HGAP surface, config-3 shape): .pre text of N targets x L x cov
(unaligned q / t substrings, Alignment.cpp:82-112) on tmpfs -> re-alignment -> consensus -> FASTA."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbdagcon_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
cov = int(sys.argv[3]) if len(sys.argv) > 3 else 60
b = synth.make_batch(n, L, cov, seed=7000)
path = ("/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp") + "/e2e.pre"
with open(path, "wb") as f:
    for t in range(b.n_targets):
        tid = b.ids[t].encode()
        for k, (start, q, tt) in enumerate(b.target_alignments(t)):
            qs, ts = q.replace(b"-", b""), tt.replace(b"-", b"")
            f.write(b"q%07d_%d %s + %d %d %d %s %s\n" % (t, k, tid, int(b.tlen[t]), start - 1, start - 1 + len(ts), qs, ts))
size = os.path.getsize(path)
print(f"wrote {size / 1e6:.1f} MB of .pre", flush=True)
exe = os.path.join(ROOT, "pbdagcon_amd", "bin", "pbdagcon")
for rep in range(2):
    t0 = time.time()
    out = subprocess.run([exe, "-a", "-c", "8", "-j", os.environ.get("E2E_J", "16")] + os.environ.get("E2E_ARGS", "").split() + [path],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, PBDAGCON_TIMING="1"))
    dt = time.time() - t0
    bases = sum(len(l) for l in out.stdout.split(b"\n") if l and not l.startswith(b">"))
    print(f"run {rep}: rc {out.returncode}, {dt:.2f} s wall, {bases} consensus bases, {bases / dt / 1e6:.2f} M bases/s end to end, "
          f"{size / dt / 1e6:.0f} MB/s of .pre", flush=True)
    print(out.stderr.decode()[-700:].strip())
if not os.environ.get("E2E_KEEP"): os.unlink(path)
