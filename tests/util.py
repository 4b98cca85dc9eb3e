"""Shared helpers for the parity tests."""
import numpy as np

import oracle


def oracle_batch(batch, min_cov=6, min_len=500, trim=50, min_weight=None):
    """main.cpp:66-72,118 (coverage filter) + main.cpp:130-138 per target, on the CPU oracle."""
    if min_weight is None or min_weight < 0:
        min_weight = min_cov
    out = []
    for t in range(batch.n_targets):
        a0, a1 = int(batch.aln_begin[t]), int(batch.aln_begin[t + 1])
        k = a1 - a0
        if k == 0 or k < min_cov:
            out.append([])
            continue
        bb = None
        if batch.backbone is not None:
            o = int(batch.backbone_off[t])
            bb = batch.backbone[o:o + int(batch.tlen[t])].tobytes()
        out.append(oracle.consensus_target_blob(
            int(batch.tlen[t]), batch.aln_start[a0:a1].copy(), batch.aln_off[a0:a1].copy(),
            batch.aln_len[a0:a1].copy(), batch.qstr, batch.tstr, min_len, trim, min_weight, bb))
    return out


def random_target(rng, tlen, n_reads, alphabet=b"ACGT", sub=0.03, ins=0.12, dele=0.06,
                  ins_ext=0.3, full_span=False, dots=False):
    """A random backbone and n_reads raw alignments to it: [(start, q, t)], backbone bytes.
    Small alphabets and high error rates make equal-base neighbours (merge work) frequent."""
    bb = bytes(alphabet[i] for i in rng.integers(0, len(alphabet), tlen))
    alns = []
    for _ in range(n_reads):
        if full_span or tlen < 4:
            s0, span = 0, tlen
        else:
            span = int(rng.integers(max(1, tlen // 3), tlen + 1))
            s0 = int(rng.integers(0, tlen - span + 1))
        q, t = bytearray(), bytearray()
        if rng.random() < 0.3:                      # leading insertion run
            for _ in range(int(rng.integers(1, 4))):
                q.append(alphabet[rng.integers(0, len(alphabet))]); t.append(0x2D)
        for i in range(s0, s0 + span):
            b = bb[i]
            u = rng.random()
            if u < dele:
                q.append(0x2D); t.append(b)
            elif u < dele + sub:
                x = alphabet[rng.integers(0, len(alphabet))]
                q.append(x); t.append(b)            # may equal b: then it is a match
            else:
                q.append(b); t.append(b)
            if rng.random() < ins:
                while True:
                    q.append(alphabet[rng.integers(0, len(alphabet))]); t.append(0x2D)
                    if rng.random() >= ins_ext:
                        break
        if dots:
            for i in range(len(q)):
                if q[i] == 0x2D and rng.random() < 0.3:
                    q[i] = 0x2E
                if t[i] == 0x2D and rng.random() < 0.3:
                    t[i] = 0x2E
        alns.append((s0 + 1, bytes(q), bytes(t)))
    return alns, bb


def batch_from_targets(targets, with_backbone=False):
    """targets = [(tlen, [(start,q,t)...], backbone)] -> pbdagcon_amd.capi.HostBatch."""
    from pbdagcon_amd.capi import HostBatch
    tlen, begins, starts, offs, lens = [], [0], [], [], []
    qs, ts, bbs, bb_off = [], [], [], []
    pos = bpos = 0
    for tl, alns, bb in targets:
        tlen.append(tl)
        for s, q, t in alns:
            starts.append(s); offs.append(pos); lens.append(len(q))
            qs.append(q); ts.append(t)
            pos += len(q)
        begins.append(len(starts))
        bb_off.append(bpos)
        if with_backbone:
            bbs.append(bb); bpos += tl
    return HostBatch(np.array(tlen, np.uint32), np.array(begins, np.uint64), np.array(starts, np.uint32),
                     np.array(offs, np.uint64), np.array(lens, np.uint32),
                     b"".join(qs) or b"", b"".join(ts) or b"",
                     (b"".join(bbs) or b"N") if with_backbone else None,
                     np.array(bb_off, np.uint64) if with_backbone else None)


def concat_batches(batches):
    """Several HostBatches (no backbone) as one: targets in order, blobs concatenated."""
    from pbdagcon_amd.capi import HostBatch
    tlen = np.concatenate([b.tlen for b in batches])
    starts = np.concatenate([b.aln_start for b in batches])
    lens = np.concatenate([b.aln_len for b in batches])
    offs, begins, ids = [], [np.zeros(1, np.uint64)], []
    pos = na = 0
    for b in batches:
        offs.append(b.aln_off + np.uint64(pos))
        begins.append(b.aln_begin[1:] + np.uint64(na))
        pos += int(b.qstr.size); na += b.n_alns
        ids.extend(b.ids or ["x%d" % i for i in range(b.n_targets)])
    return HostBatch(tlen, np.concatenate(begins), starts, np.concatenate(offs), lens,
                     np.concatenate([b.qstr for b in batches]), np.concatenate([b.tstr for b in batches]), None, None, ids)
