"""Host-side mirror of the reference's interface for the hot path.

Same names and argument meaning as the reference so that the parity tests read
like test/cpp/*.cpp:

    Alignment            dagcon::Alignment           src/cpp/Alignment.hpp:12-42
    normalizeGaps        src/cpp/Alignment.cpp:131-217   (on the device, stage a1)
    trimAln              src/cpp/Alignment.cpp:219-242   (on the device, stage a1)
    AlnGraphBoost        src/cpp/AlnGraphBoost.hpp:74-143: addAln / mergeNodes /
                         consensus -- alignments are collected on the host and the
                         whole graph pipeline (stages a2, b, c) runs on the GPU when
                         a consensus is asked for.
    CnsResult            src/cpp/AlnGraphBoost.hpp:63-66

Everything computes through libdagcon_hip.so; there is no CPU path here.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import capi


@dataclass
class Alignment:
    tlen: int = 0
    start: int = 0          # conforming offsets are 1-based (Alignment.hpp:21-22)
    end: int = 0
    id: str = ""
    sid: str = ""
    strand: str = "+"
    qstr: bytes = b""
    tstr: bytes = b""


@dataclass
class CnsResult:
    range: tuple = (0, 0)
    seq: bytes = b""


_CTX = {}


def _ctx(device=0) -> capi.Context:
    if device not in _CTX:
        _CTX[device] = capi.Context(min_cov=0, min_len=0, trim=0, min_weight=0, device=device)
    return _CTX[device]


def normalizeGaps(aln: Alignment, push: bool = True, device=0) -> Alignment:
    """Alignment.cpp:131-217.  `push=False` is not offered by the device path
    (the reference never calls it that way: main.cpp:133, dazcon.cpp:79)."""
    if not push:
        raise NotImplementedError("normalizeGaps(push=false) is unused by the reference's hot path")
    (start, q, t), = _ctx(device).normalize([(aln.start, aln.qstr, aln.tstr)], trim=0)
    # the reference copies id/sid/start/tlen/strand, not `end` (Alignment.cpp:204-208)
    return Alignment(tlen=aln.tlen, start=aln.start, end=0, id=aln.id, sid=aln.sid,
                     strand=aln.strand, qstr=q, tstr=t)


def trimAln(aln: Alignment, trimLen: int = 50, device=0) -> None:
    """Alignment.cpp:219-242, in place, on the device (k_normalize with the
    normalisation switched off)."""
    (start, q, t), = _ctx(device).normalize([(aln.start, aln.qstr, aln.tstr)], trim=trimLen, raw=True)
    aln.qstr, aln.tstr, aln.start = q, t, start


class AlnGraphBoost:
    """AlnGraphBoost.hpp:74-143 for this path."""

    def __init__(self, backbone, device=0):
        if isinstance(backbone, (bytes, bytearray)):
            self._backbone, self._blen = bytes(backbone), len(backbone)   # AlnGraphBoost.cpp:16-39
        else:
            self._backbone, self._blen = None, int(backbone)              # AlnGraphBoost.cpp:41-62
        self._alns = []
        self._device = device
        self._merged = False

    def addAln(self, aln: Alignment) -> None:
        """AlnGraphBoost.cpp:64-107: strings are used as they are (raw mode)."""
        if len(aln.qstr) != len(aln.tstr):
            raise ValueError("query and target strings must be equal length")
        self._alns.append((aln.start, aln.qstr, aln.tstr))

    def mergeNodes(self) -> None:
        """AlnGraphBoost.cpp:129-160 (runs on the device together with consensus)."""
        self._merged = True

    def _run(self, minWeight, minLen, flags=capi.FLAG_RAW_ALIGNMENTS):
        if not self._merged:
            raise RuntimeError("mergeNodes() must be called before consensus()")
        n = len(self._alns)
        lens = np.array([len(a[1]) for a in self._alns], dtype=np.uint32)
        offs = np.zeros(n, dtype=np.uint64)
        if n:
            offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        batch = capi.HostBatch(
            [self._blen], [0, n], [a[0] for a in self._alns], offs, lens,
            b"".join(a[1] for a in self._alns), b"".join(a[2] for a in self._alns),
            self._backbone, None if self._backbone is None else [0])
        ctx = capi.Context(min_cov=0, min_len=0, trim=0, min_weight=minWeight, device=self._device,
                           flags=flags)
        # min_len doubles as the alignment pre-filter in the ABI (main.cpp:132);
        # this class takes alignments unfiltered, so segments are filtered here
        try:
            segs = ctx.consensus(batch)[0]
        finally:
            ctx.close()
        return [CnsResult((r0, r1), s) for r0, r1, s in segs if r1 - r0 >= minLen]

    def consensus(self, seqs=None, minWeight: int = 0, minLength: int = 500):
        """consensus(minWeight) -> longest run (AlnGraphBoost.cpp:285-325) when
        `seqs` is None; consensus(seqs, minWeight, minLength) fills `seqs`
        with every qualifying run (AlnGraphBoost.cpp:327-373)."""
        if seqs is None:
            best = b""
            for r in self._run(minWeight, 0):
                if len(r.seq) > len(best):     # strict '>': first longest wins (:309,:319)
                    best = r.seq
            return best
        seqs[:] = self._run(minWeight, minLength)
        return None
