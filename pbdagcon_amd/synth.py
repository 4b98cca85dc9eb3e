"""Deterministic synthetic pileups in the C-ABI batch layout.

The generator itself is C (csrc/dagcon_synth.c, libdagcon_synth.so): seed =
base_seed + target_index, so a target's data does not depend on how a batch is
sharded over ranks.
"""
from __future__ import annotations

import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .capi import HostBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SynthParams(C.Structure):
    _fields_ = [("tlen", C.c_uint32), ("coverage", C.c_uint32), ("sub_ppm", C.c_uint32),
                ("ins_open_ppm", C.c_uint32), ("ins_ext_ppm", C.c_uint32), ("del_ppm", C.c_uint32),
                ("min_span_ppm", C.c_uint32), ("reserved", C.c_uint32)]


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdagcon_synth.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run __graft_entry__.build()")
        L = C.CDLL(path)
        L.dagcon_synth_target.restype = C.c_uint64
        L.dagcon_synth_target.argtypes = [C.POINTER(SynthParams), C.c_uint64, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def make_batch(n_targets, tlen, coverage, seed=1000, first_target=0, sub=0.01, ins=0.10, dele=0.04,
               ins_ext=0.20, min_span=1.0, with_backbone=False, tlens=None, threads=8) -> HostBatch:
    """n_targets targets of `coverage` reads each.  `ins` is the expected number
    of inserted bases per target base (run open probability = ins * (1 - ins_ext)).
    `tlens` (optional array) gives per-target lengths (mixed-length batches)."""
    L = _lib()
    tl = np.full(n_targets, tlen, dtype=np.uint32) if tlens is None else np.asarray(tlens, dtype=np.uint32)
    ps = []
    for t in range(n_targets):
        p = SynthParams()
        p.tlen, p.coverage = int(tl[t]), coverage
        p.sub_ppm, p.del_ppm = int(sub * 1e6), int(dele * 1e6)
        p.ins_open_ppm, p.ins_ext_ppm = int(ins * (1.0 - ins_ext) * 1e6), int(ins_ext * 1e6)
        p.min_span_ppm = int(min_span * 1e6)
        ps.append(p)
    A = n_targets * coverage
    starts, lens, offs = np.zeros(A, np.uint32), np.zeros(A, np.uint32), np.zeros(A, np.uint64)
    sizes = np.zeros(n_targets, dtype=np.uint64)

    def size_of(t):
        sizes[t] = L.dagcon_synth_target(C.byref(ps[t]), seed + first_target + t, None, None, None, None, 0, None, None)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(size_of, range(n_targets)))
    base = np.zeros(n_targets + 1, dtype=np.uint64)
    base[1:] = np.cumsum(sizes)
    total = int(base[-1])
    q, tt = np.empty(max(total, 1), np.uint8), np.empty(max(total, 1), np.uint8)
    bb = np.empty(int(tl.sum()) if with_backbone else 0, np.uint8)
    bb_off = np.zeros(n_targets, dtype=np.uint64)
    if with_backbone and n_targets:
        bb_off[1:] = np.cumsum(tl[:-1].astype(np.uint64))

    def fill(t):
        a0 = t * coverage
        L.dagcon_synth_target(
            C.byref(ps[t]), seed + first_target + t,
            bb.ctypes.data + int(bb_off[t]) if with_backbone else None,
            starts.ctypes.data + 4 * a0, lens.ctypes.data + 4 * a0, offs.ctypes.data + 8 * a0,
            int(base[t]), q.ctypes.data + int(base[t]), tt.ctypes.data + int(base[t]))

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(fill, range(n_targets)))
    aln_begin = np.arange(n_targets + 1, dtype=np.uint64) * coverage
    ids = [f"t{first_target + t:07d}/0_{int(tl[t])}" for t in range(n_targets)]
    return HostBatch(tl, aln_begin, starts, offs, lens, q[:total], tt[:total],
                     bb if with_backbone else None, bb_off if with_backbone else None, ids)


def to_m5(batch: HostBatch) -> bytes:
    """The batch as BLASR -m 5 text ('+' strand), the layout parseM5 reads
    (reference Alignment.cpp:44-80): 19 space-separated fields."""
    lines = []
    for t in range(batch.n_targets):
        tid = batch.ids[t] if batch.ids else f"t{t:07d}/0_{int(batch.tlen[t])}"
        for k, (start, q, tt) in enumerate(batch.target_alignments(t)):
            nq = sum(1 for c in q if c != 0x2D)
            nt = sum(1 for c in tt if c != 0x2D)
            match = "".join("|" if a == b else "*" for a, b in zip(q, tt))
            lines.append(
                f"q{t:07d}_{k}/0_{nq} {nq} 0 {nq} + {tid} {int(batch.tlen[t])} {start - 1} {start - 1 + nt} + "
                f"-1000 0 0 0 0 254 {q.decode()} {match} {tt.decode()}")
    return ("\n".join(lines) + "\n").encode()
