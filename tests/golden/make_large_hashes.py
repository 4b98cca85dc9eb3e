#!/usr/bin/env python3
"""Golden SHA-256 of the consensus of the large BASELINE shapes, computed by the CPU oracle
(oracle/dagcon_oracle.c) in the build container.  The GPU tests hash what the device returns
the same way and compare: whole-batch parity at full size without running the oracle on the
GPU box.  (These are oracle outputs, not reference outputs: the graph stages of the oracle are
pinned to the reference by its own known-answer tests only)

    python tests/golden/make_large_hashes.py        # rewrites tests/golden/large_hashes.json
"""
import hashlib
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np          # noqa: E402
import oracle               # noqa: E402
from pbdagcon_amd import synth  # noqa: E402


def shapes():
    """name -> (batch factory, options): the same calls the GPU tests make."""
    return {
        "configs1_1000x10kx40": (lambda: synth.make_batch(1000, 10000, 40, seed=1000), dict(min_cov=6, min_len=500, trim=50)),
        "config3_2x50kx60": (lambda: synth.make_batch(2, 50000, 60, seed=7000), dict(min_cov=8, min_len=500, trim=50)),
        "config5_6xmixedx30_partial": (lambda: synth.make_batch(
            6, 0, 30, seed=8000, min_span=0.6, tlens=np.random.default_rng(5).integers(2000, 40000, 6), with_backbone=True),
            dict(min_cov=6, min_len=500, trim=10)),
        "config5_400xmixedx30_partial": (lambda: synth.make_batch(
            400, 0, 30, seed=8000, min_span=0.6, tlens=np.random.default_rng(5).integers(2000, 40000, 400), with_backbone=True),
            dict(min_cov=6, min_len=500, trim=10)),
    }


def digest(results):
    """SHA-256 over the per-target segment lists, in target order."""
    h = hashlib.sha256()
    for t, segs in enumerate(results):
        h.update(b"T%d:%d\n" % (t, len(segs)))
        for r0, r1, seq in segs:
            h.update(b"%d %d " % (r0, r1)); h.update(seq); h.update(b"\n")
    return h.hexdigest()


def oracle_all(batch, min_cov, min_len, trim, threads=None):
    threads = threads or len(os.sched_getaffinity(0))

    def one(t):
        a0, a1 = int(batch.aln_begin[t]), int(batch.aln_begin[t + 1])
        if a1 - a0 == 0 or a1 - a0 < min_cov:
            return []
        bb = None
        if batch.backbone is not None:
            o = int(batch.backbone_off[t])
            bb = batch.backbone[o:o + int(batch.tlen[t])].tobytes()
        return oracle.consensus_target_blob(
            int(batch.tlen[t]), batch.aln_start[a0:a1].copy(), batch.aln_off[a0:a1].copy(),
            batch.aln_len[a0:a1].copy(), batch.qstr, batch.tstr, min_len, trim, min_cov, bb)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(one, range(batch.n_targets)))


if __name__ == "__main__":
    oracle.build()
    out = {}
    for name, (make, opts) in shapes().items():
        b = make()
        res = oracle_all(b, **opts)
        out[name] = {"sha256": digest(res), "targets": b.n_targets,
                     "consensus_bases": int(sum(len(s) for segs in res for _, _, s in segs)), "options": opts}
        print(name, out[name], flush=True)
    json.dump(out, open(os.path.join(HERE, "large_hashes.json"), "w"), indent=1)
