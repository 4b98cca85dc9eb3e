"""ctypes binding of the C ABI in include/dagcon.h (libdagcon_hip.so).

There is no fallback: if the HIP extension is missing or no GPU is present the
calls raise.  Nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdagcon_hip.so")

DAGCON_OK = 0
ERR_NAMES = {
    -1: "DAGCON_ERR_INVALID_ARG", -2: "DAGCON_ERR_NO_DEVICE", -3: "DAGCON_ERR_HIP",
    -4: "DAGCON_ERR_NONCONFORMING", -5: "DAGCON_ERR_UNSUPPORTED", -6: "DAGCON_ERR_WORKSPACE",
    -7: "DAGCON_ERR_INTERNAL", -8: "DAGCON_ERR_STATE",
}
FLAG_RAW_ALIGNMENTS = 1
FLAG_STOP_AFTER_BUILD = 2
FLAG_STOP_AFTER_MERGE = 4
FLAG_DEBUG_RESWEEP = 16
MAX_COVERAGE = 4094

EXPORTS = [
    "dagcon_abi_version", "dagcon_default_opts", "dagcon_create", "dagcon_destroy",
    "dagcon_last_error", "dagcon_consensus", "dagcon_upload", "dagcon_run", "dagcon_sync",
    "dagcon_fetch", "dagcon_get_timings", "dagcon_normalize", "dagcon_debug_graph",
    "dagcon_debug_counters", "dagcon_host_alloc", "dagcon_host_free", "dagcon_align",
    "dagcon_consensus_pre", "dagcon_debug_plan", "dagcon_align_dropped",
]
ABI_VERSION = 2


class DagconError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class Opts(C.Structure):
    _fields_ = [("min_cov", C.c_uint32), ("min_len", C.c_uint32), ("trim", C.c_uint32),
                ("min_weight", C.c_int32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("max_segments", C.c_uint32), ("min_segment_len", C.c_uint32)]


class Batch(C.Structure):
    _fields_ = [("n_targets", C.c_uint32), ("tlen", C.c_void_p), ("aln_begin", C.c_void_p),
                ("aln_start", C.c_void_p), ("aln_off", C.c_void_p), ("aln_len", C.c_void_p),
                ("qstr", C.c_void_p), ("tstr", C.c_void_p), ("blob_bytes", C.c_uint64),
                ("backbone", C.c_void_p), ("backbone_off", C.c_void_p)]


class PreBatch(C.Structure):
    _fields_ = [("n_targets", C.c_uint32), ("tlen", C.c_void_p), ("rec_begin", C.c_void_p),
                ("tstart", C.c_void_p), ("strand", C.c_void_p), ("q_off", C.c_void_p), ("q_len", C.c_void_p),
                ("t_off", C.c_void_p), ("t_len", C.c_void_p), ("q_blob", C.c_void_p), ("q_bytes", C.c_uint64),
                ("t_blob", C.c_void_p), ("t_bytes", C.c_uint64)]


class Results(C.Structure):
    _fields_ = [("n_targets", C.c_uint32), ("n_segments", C.c_uint64),
                ("seg_begin", C.POINTER(C.c_uint64)), ("range0", C.POINTER(C.c_int32)),
                ("range1", C.POINTER(C.c_int32)), ("seq_off", C.POINTER(C.c_uint64)),
                ("seq_len", C.POINTER(C.c_uint32)), ("seq_blob", C.c_void_p),
                ("seq_bytes", C.c_uint64), ("target_status", C.POINTER(C.c_int32)),
                ("n_failed", C.c_uint32)]


class Timings(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_normalize", C.c_float), ("ms_build", C.c_float),
                ("ms_merge", C.c_float), ("ms_bestpath", C.c_float),
                ("algorithmic_bytes", C.c_uint64), ("consensus_bases", C.c_uint64),
                ("n_alignments", C.c_uint64), ("n_columns", C.c_uint64), ("n_nodes", C.c_uint64),
                ("reruns", C.c_uint32), ("merge_segments", C.c_uint32)]


class GraphDump(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("base", C.POINTER(C.c_uint8)),
                ("weight", C.POINTER(C.c_int32)), ("coverage", C.POINTER(C.c_int32)),
                ("deleted", C.POINTER(C.c_uint8)), ("backbone", C.POINTER(C.c_uint8)),
                ("bbpos", C.POINTER(C.c_int32)), ("out_begin", C.POINTER(C.c_uint32)),
                ("out_dst", C.POINTER(C.c_int32)), ("out_count", C.POINTER(C.c_int32)),
                ("in_begin", C.POINTER(C.c_uint32)), ("in_src", C.POINTER(C.c_int32))]


_LIB = None


def load() -> C.CDLL:
    """Load libdagcon_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  pbdagcon_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.dagcon_abi_version.restype = C.c_int
    L.dagcon_default_opts.argtypes = [C.POINTER(Opts)]
    L.dagcon_create.argtypes = [C.POINTER(Opts), C.POINTER(vp)]
    L.dagcon_destroy.argtypes = [vp]
    L.dagcon_destroy.restype = None
    L.dagcon_last_error.argtypes = [vp]
    L.dagcon_last_error.restype = C.c_char_p
    L.dagcon_consensus.argtypes = [vp, C.POINTER(Batch), C.POINTER(Results)]
    L.dagcon_upload.argtypes = [vp, C.POINTER(Batch)]
    L.dagcon_run.argtypes = [vp]
    L.dagcon_sync.argtypes = [vp]
    L.dagcon_fetch.argtypes = [vp, C.POINTER(Results)]
    L.dagcon_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.dagcon_normalize.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint32,
                                   C.c_uint32, vp, vp, vp, vp, vp]
    L.dagcon_debug_graph.argtypes = [vp, C.c_uint32, C.POINTER(GraphDump)]
    L.dagcon_align.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, vp]
    L.dagcon_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.dagcon_host_free.argtypes = [vp, vp]
    L.dagcon_host_free.restype = None
    _LIB = L
    return L


def default_opts() -> Opts:
    o = Opts()
    load().dagcon_default_opts(C.byref(o))
    return o


class HostBatch:
    """numpy view of a dagcon_batch (structure-of-arrays blobs)."""

    def __init__(self, tlen, aln_begin, aln_start, aln_off, aln_len, qstr, tstr,
                 backbone=None, backbone_off=None, ids=None):
        self.tlen = np.ascontiguousarray(tlen, dtype=np.uint32)
        self.aln_begin = np.ascontiguousarray(aln_begin, dtype=np.uint64)
        self.aln_start = np.ascontiguousarray(aln_start, dtype=np.uint32)
        self.aln_off = np.ascontiguousarray(aln_off, dtype=np.uint64)
        self.aln_len = np.ascontiguousarray(aln_len, dtype=np.uint32)
        self.qstr = np.ascontiguousarray(np.frombuffer(qstr, dtype=np.uint8)
                                         if isinstance(qstr, (bytes, bytearray)) else qstr, dtype=np.uint8)
        self.tstr = np.ascontiguousarray(np.frombuffer(tstr, dtype=np.uint8)
                                         if isinstance(tstr, (bytes, bytearray)) else tstr, dtype=np.uint8)
        assert self.qstr.size == self.tstr.size
        self.backbone = None if backbone is None else np.ascontiguousarray(
            np.frombuffer(backbone, dtype=np.uint8) if isinstance(backbone, (bytes, bytearray)) else backbone,
            dtype=np.uint8)
        self.backbone_off = None if backbone_off is None else np.ascontiguousarray(backbone_off, dtype=np.uint64)
        self.ids = ids

    @property
    def n_targets(self):
        return int(self.tlen.size)

    @property
    def n_alns(self):
        return int(self.aln_len.size)

    def c_struct(self) -> Batch:
        b = Batch()
        b.n_targets = self.n_targets
        b.tlen = self.tlen.ctypes.data
        b.aln_begin = self.aln_begin.ctypes.data
        b.aln_start = self.aln_start.ctypes.data
        b.aln_off = self.aln_off.ctypes.data
        b.aln_len = self.aln_len.ctypes.data
        b.qstr = self.qstr.ctypes.data
        b.tstr = self.tstr.ctypes.data
        b.blob_bytes = self.qstr.size
        b.backbone = None if self.backbone is None else self.backbone.ctypes.data
        b.backbone_off = None if self.backbone_off is None else self.backbone_off.ctypes.data
        return b

    def target_alignments(self, t):
        """[(start, qstr, tstr)] of target t, as bytes."""
        out = []
        for a in range(int(self.aln_begin[t]), int(self.aln_begin[t + 1])):
            o, n = int(self.aln_off[a]), int(self.aln_len[a])
            out.append((int(self.aln_start[a]), self.qstr[o:o + n].tobytes(), self.tstr[o:o + n].tobytes()))
        return out

    def select(self, targets):
        """A new batch holding only the given targets (blobs are shared)."""
        targets = list(targets)
        begins = [0]
        idx = []
        for t in targets:
            a0, a1 = int(self.aln_begin[t]), int(self.aln_begin[t + 1])
            idx.extend(range(a0, a1))
            begins.append(len(idx))
        idx = np.asarray(idx, dtype=np.int64)
        return HostBatch(self.tlen[targets], begins, self.aln_start[idx], self.aln_off[idx],
                         self.aln_len[idx], self.qstr, self.tstr,
                         self.backbone, None if self.backbone_off is None else self.backbone_off[targets],
                         None if self.ids is None else [self.ids[t] for t in targets])


class Context:
    """dagcon_ctx handle.  One per GPU; single-owner."""

    def __init__(self, min_cov=6, min_len=500, trim=50, min_weight=-1, device=0, flags=0, max_segments=0,
                 min_segment_len=0):
        self.L = load()
        o = Opts()
        o.min_cov, o.min_len, o.trim, o.min_weight = min_cov, min_len, trim, min_weight
        o.device, o.flags, o.max_segments = device, flags, max_segments
        o.min_segment_len = min_segment_len
        self.opts = o
        self.h = C.c_void_p()
        rc = self.L.dagcon_create(C.byref(o), C.byref(self.h))
        if rc != DAGCON_OK:
            raise DagconError(rc, "dagcon_create failed (is a gfx950 GPU visible?)")
        self._keep = None
        self._pinned = []
        self.target_status = None      # per-target dagcon_status of the last fetch (ABI 2)

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            for p in self._pinned:
                self.L.dagcon_host_free(self.h, p)
            self._pinned = []
            self.L.dagcon_destroy(self.h)
            self.h = C.c_void_p()

    def host_array(self, nbytes):
        """uint8 numpy array over page-locked host memory (dagcon_host_alloc); freed by close()."""
        p = C.c_void_p()
        self._chk(self.L.dagcon_host_alloc(self.h, max(int(nbytes), 1), C.byref(p)))
        self._pinned.append(p)
        return np.ctypeslib.as_array((C.c_uint8 * max(int(nbytes), 1)).from_address(p.value))[:int(nbytes)]

    def pin_batch(self, batch: "HostBatch") -> "HostBatch":
        """A copy of the batch whose string blobs live in page-locked memory."""
        q = self.host_array(batch.qstr.size); q[:] = batch.qstr
        t = self.host_array(batch.tstr.size); t[:] = batch.tstr
        bb = None
        if batch.backbone is not None:
            bb = self.host_array(batch.backbone.size); bb[:] = batch.backbone
        return HostBatch(batch.tlen, batch.aln_begin, batch.aln_start, batch.aln_off, batch.aln_len, q, t,
                         bb, batch.backbone_off, batch.ids)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != DAGCON_OK:
            raise DagconError(rc, (self.L.dagcon_last_error(self.h) or b"").decode())

    def upload(self, batch: HostBatch):
        self._keep = batch
        b = batch.c_struct()
        self._chk(self.L.dagcon_upload(self.h, C.byref(b)))

    def run(self):
        self._chk(self.L.dagcon_run(self.h))

    def sync(self):
        self._chk(self.L.dagcon_sync(self.h))

    def _status(self, r, strict):
        """ABI 2: a failure is confined to its target.  strict (the default) raises for the first failed
        target, as a caller that cannot use a partial batch wants; strict=False returns the batch with
        [] for the failed targets and leaves their codes in self.target_status."""
        self.target_status = np.ctypeslib.as_array(r.target_status, shape=(r.n_targets,)).copy() if r.n_targets else np.zeros(0, np.int32)
        if strict and r.n_failed:
            t = int(np.flatnonzero(self.target_status)[0])
            raise DagconError(int(self.target_status[t]), (self.L.dagcon_last_error(self.h) or b"").decode())

    def fetch(self, strict=True):
        r = Results()
        self._chk(self.L.dagcon_fetch(self.h, C.byref(r)))
        self._status(r, strict)
        return _results_to_py(r)

    def fetch_raw(self):
        """dagcon_fetch without the conversion to Python objects: the returned struct points into
        host memory the context owns until its next fetch (dagcon_run does not touch it), so a
        caller can start the next run first and convert (results_to_py) meanwhile."""
        r = Results()
        self._chk(self.L.dagcon_fetch(self.h, C.byref(r)))
        self._status(r, True)
        return r

    results_to_py = staticmethod(lambda r: _results_to_py(r))

    def consensus(self, batch: HostBatch, strict=True):
        """Per target: [(range0, range1, seq_bytes)]."""
        self._keep = batch
        b = batch.c_struct()
        r = Results()
        self._chk(self.L.dagcon_consensus(self.h, C.byref(b), C.byref(r)))
        self._status(r, strict)
        return _results_to_py(r)

    def timings(self) -> dict:
        t = Timings()
        self._chk(self.L.dagcon_get_timings(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timings._fields_ if k != "reserved"}

    def normalize(self, alns, trim=0, raw=False):
        """alns = [(start, qstr, tstr)] -> [(start', qnorm, tnorm)] on the device:
        normalizeGaps then trimAln(trim); raw=True gives trimAln alone."""
        n = len(alns)
        starts = np.array([a[0] for a in alns], dtype=np.uint32)
        lens = np.array([len(a[1]) for a in alns], dtype=np.uint32)
        offs = np.zeros(n, dtype=np.uint64)
        if n:
            offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        q = np.frombuffer(b"".join(a[1] for a in alns) or b"\0", dtype=np.uint8)
        t = np.frombuffer(b"".join(a[2] for a in alns) or b"\0", dtype=np.uint8)
        out_off = 2 * offs
        total = int(2 * lens.sum()) + 1
        qout, tout = np.zeros(total, dtype=np.uint8), np.zeros(total, dtype=np.uint8)
        out_len, out_start = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
        self._chk(self.L.dagcon_normalize(
            self.h, n, starts.ctypes.data, offs.ctypes.data, lens.ctypes.data, q.ctypes.data,
            t.ctypes.data, int(lens.sum()), trim, FLAG_RAW_ALIGNMENTS if raw else 0, out_off.ctypes.data, qout.ctypes.data,
            tout.ctypes.data, out_len.ctypes.data, out_start.ctypes.data))
        res = []
        for a in range(n):
            o, m = int(out_off[a]), int(out_len[a])
            res.append((int(out_start[a]), qout[o:o + m].tobytes(), tout[o:o + m].tobytes()))
        return res

    def align(self, pairs):
        """pairs = [(qseq, tseq)] of unaligned sequences -> [(qaln, taln)] (the -a stage, SimpleAligner.cpp:25-63)."""
        n = len(pairs)
        if n == 0:
            return []
        ql = np.array([len(q) for q, _ in pairs], dtype=np.uint32)
        tl = np.array([len(t) for _, t in pairs], dtype=np.uint32)
        qo, to, oo = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        qo[1:] = np.cumsum(ql[:-1], dtype=np.uint64)
        to[1:] = np.cumsum(tl[:-1], dtype=np.uint64)
        oo[1:] = np.cumsum((ql[:-1].astype(np.uint64) + tl[:-1]), dtype=np.uint64)
        qb = np.frombuffer(b"".join(q for q, _ in pairs) or b"\0", dtype=np.uint8)
        tb = np.frombuffer(b"".join(t for _, t in pairs) or b"\0", dtype=np.uint8)
        total = int(ql.sum()) + int(tl.sum()) + 1
        qa, ta = np.zeros(total, np.uint8), np.zeros(total, np.uint8)
        ln = np.zeros(n, np.uint32)
        self._chk(self.L.dagcon_align(self.h, n, qo.ctypes.data, ql.ctypes.data, to.ctypes.data, tl.ctypes.data,
                                      qb.ctypes.data, int(ql.sum()), tb.ctypes.data, int(tl.sum()), oo.ctypes.data,
                                      qa.ctypes.data, ta.ctypes.data, ln.ctypes.data))
        return [(qa[int(oo[a]):int(oo[a]) + int(ln[a])].tobytes(), ta[int(oo[a]):int(oo[a]) + int(ln[a])].tobytes())
                for a in range(n)]

    def consensus_pre(self, targets, strict=True):
        """targets = [(tlen, [(tstart, strand, qseq, tseq)])]: .pre records per target (Alignment.cpp:82-112)
        -> per target [(range0, range1, seq_bytes)] (dagcon_consensus_pre: main.cpp:117-145 with -a)."""
        recs = [r for _, rs in targets for r in rs]
        n = len(recs)
        tlen = np.array([t for t, _ in targets], dtype=np.uint32)
        begin = np.zeros(len(targets) + 1, np.uint64)
        begin[1:] = np.cumsum([len(rs) for _, rs in targets], dtype=np.uint64)
        ts = np.array([r[0] for r in recs] or [0], dtype=np.uint32)
        strand = np.frombuffer(b"".join(r[1] for r in recs) or b"+", dtype=np.uint8)
        ql = np.array([len(r[2]) for r in recs] or [0], dtype=np.uint32)
        tl = np.array([len(r[3]) for r in recs] or [0], dtype=np.uint32)
        qo, to = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64)
        qo[1:] = np.cumsum(ql[:-1], dtype=np.uint64)
        to[1:] = np.cumsum(tl[:-1], dtype=np.uint64)
        qb = np.frombuffer(b"".join(r[2] for r in recs) or b"\0", dtype=np.uint8)
        tb = np.frombuffer(b"".join(r[3] for r in recs) or b"\0", dtype=np.uint8)
        pb = PreBatch(len(targets), tlen.ctypes.data, begin.ctypes.data, ts.ctypes.data, strand.ctypes.data,
                      qo.ctypes.data, ql.ctypes.data, to.ctypes.data, tl.ctypes.data, qb.ctypes.data,
                      int(ql.sum()) if n else 0, tb.ctypes.data, int(tl.sum()) if n else 0)
        r = Results()
        self.L.dagcon_consensus_pre.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.L.dagcon_consensus_pre(self.h, C.byref(pb), C.byref(r)))
        self._status(r, strict)
        return _results_to_py(r)

    def debug_counters(self):
        a = (C.c_ulonglong * 16)()
        self.L.dagcon_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
        self._chk(self.L.dagcon_debug_counters(self.h, a))
        return list(a)

    def debug_graph(self, target=0):
        """Per vertex (device ids, backbone-position order):
        dict(base, weight, coverage, deleted, backbone, bbpos, out=[(dst,count)], inn=[src])."""
        d = GraphDump()
        self._chk(self.L.dagcon_debug_graph(self.h, target, C.byref(d)))
        out = []
        for v in range(d.n_nodes):
            oe = [(d.out_dst[i], d.out_count[i]) for i in range(d.out_begin[v], d.out_begin[v + 1])]
            ie = [d.in_src[i] for i in range(d.in_begin[v], d.in_begin[v + 1])]
            out.append(dict(base=chr(d.base[v]), weight=d.weight[v], coverage=d.coverage[v],
                            deleted=bool(d.deleted[v]), backbone=bool(d.backbone[v]), bbpos=d.bbpos[v],
                            out=oe, inn=ie))
        return out


def _results_to_py(r: Results):
    T = r.n_targets
    blob = C.string_at(r.seq_blob, r.seq_bytes) if r.seq_bytes else b""
    out = []
    for t in range(T):
        segs = []
        for s in range(r.seg_begin[t], r.seg_begin[t + 1]):
            o, n = r.seq_off[s], r.seq_len[s]
            segs.append((r.range0[s], r.range1[s], blob[o:o + n]))
        out.append(segs)
    return out
