"""oracle -- CPU checker for the DAGCon hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (pbdagcon_amd) never does.

`lib()` loads oracle/liboracle.so (built by oracle/Makefile from
dagcon_oracle.c); `ref_lib()` loads oracle/_ref/libref_alignment.so (the
reference's own Alignment.cpp compiled in place) when it exists.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


class Segment(C.Structure):
    _fields_ = [("range0", C.c_int32), ("range1", C.c_int32), ("seq", C.c_char_p)]


class Opts(C.Structure):
    _fields_ = [("min_len", C.c_uint32), ("trim", C.c_uint32), ("min_weight", C.c_int32)]


def build(force: bool = False) -> None:
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dagcon_oracle.c")
    ref_so = os.path.join(_HERE, "_ref", "libref_alignment.so")
    cf = os.path.join(_HERE, "libcpu_faithful.so")
    need = (force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
            or not os.path.exists(cf)
            or os.path.getmtime(cf) < os.path.getmtime(os.path.join(_HERE, "cpu_faithful.cpp")))
    need_ref = os.path.exists("/root/reference/src/cpp/Alignment.cpp") and (
        force or not os.path.exists(ref_so))
    if need or need_ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    sz, u32, i32, cp, vp = C.c_size_t, C.c_uint32, C.c_int32, C.c_char_p, C.c_void_p
    L.og_revcomp.argtypes = [C.c_char_p, sz]
    L.og_normalize_gaps.restype = sz
    L.og_normalize_gaps.argtypes = [cp, cp, sz, C.c_int, cp, cp]
    L.og_trim_aln.argtypes = [cp, sz, C.c_int, C.POINTER(sz), C.POINTER(sz), C.POINTER(u32)]
    L.og_graph_new_seq.restype = vp
    L.og_graph_new_seq.argtypes = [cp, sz]
    L.og_graph_new_len.restype = vp
    L.og_graph_new_len.argtypes = [sz]
    L.og_graph_free.argtypes = [vp]
    L.og_add_aln.argtypes = [vp, u32, cp, cp, sz]
    L.og_merge_nodes.argtypes = [vp]
    L.og_best_path.restype = sz
    L.og_best_path.argtypes = [vp, C.POINTER(C.POINTER(i32))]
    L.og_consensus_longest.restype = vp
    L.og_consensus_longest.argtypes = [vp, C.c_int]
    L.og_consensus_all.restype = sz
    L.og_consensus_all.argtypes = [vp, C.c_int, sz, C.POINTER(C.POINTER(Segment))]
    L.og_free_segments.argtypes = [C.POINTER(Segment), sz]
    L.og_dangling_nodes.argtypes = [vp]
    L.og_num_nodes.restype = sz
    L.og_num_nodes.argtypes = [vp]
    L.og_num_live_nodes.restype = sz
    L.og_num_live_nodes.argtypes = [vp]
    L.og_num_live_edges.restype = sz
    L.og_num_live_edges.argtypes = [vp]
    L.og_node_info.argtypes = [vp, sz, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.og_out_edges.restype = sz
    L.og_out_edges.argtypes = [vp, sz, C.POINTER(i32), C.POINTER(i32), sz]
    L.og_in_edges.restype = sz
    L.og_in_edges.argtypes = [vp, sz, C.POINTER(i32), C.POINTER(i32), sz]
    L.og_consensus_target_blob.restype = C.c_long
    L.og_consensus_target_blob.argtypes = [
        u32, cp, sz, C.POINTER(u32), C.POINTER(C.c_uint64), C.POINTER(u32), vp, vp,
        C.POINTER(Opts), C.POINTER(C.POINTER(Segment)), C.POINTER(C.c_long)]
    L.og_parse_m5.argtypes = [cp, sz, C.c_int, vp]
    L.og_free_parsed.argtypes = [vp]
    _LIB = L
    return L


def ref_lib():
    """The reference's Alignment.cpp, or None when oracle/_ref was not built."""
    global _REF
    if _REF is not None:
        return _REF
    so = os.path.join(_HERE, "_ref", "libref_alignment.so")
    if not os.path.exists(so):
        return None
    R = C.CDLL(so)
    sz, u32, cp = C.c_size_t, C.c_uint32, C.c_char_p
    R.ref_normalize_gaps.restype = sz
    R.ref_normalize_gaps.argtypes = [cp, cp, sz, C.c_int, cp, cp]
    R.ref_trim_aln.restype = sz
    R.ref_trim_aln.argtypes = [cp, cp, sz, C.c_int, C.POINTER(u32)]
    R.ref_parse_m5.argtypes = [cp, sz, C.c_int, cp, cp, cp, cp,
                               C.POINTER(u32), C.POINTER(u32), cp]
    R.ref_revcomp.argtypes = [cp, sz]
    _REF = R
    return R


# ---- convenience wrappers --------------------------------------------------

def normalize_gaps(q: bytes, t: bytes, push: bool = True):
    L = lib()
    n = len(q)
    qo, to = C.create_string_buffer(2 * n + 1), C.create_string_buffer(2 * n + 1)
    m = L.og_normalize_gaps(q, t, n, int(push), qo, to)
    return qo.raw[:m], to.raw[:m]


def ref_normalize_gaps(q: bytes, t: bytes, push: bool = True):
    R = ref_lib()
    n = len(q)
    qo, to = C.create_string_buffer(2 * n + 1), C.create_string_buffer(2 * n + 1)
    m = R.ref_normalize_gaps(q, t, n, int(push), qo, to)
    return qo.raw[:m], to.raw[:m]


def trim_aln(q: bytes, t: bytes, start: int, trim: int):
    L = lib()
    lo, ro, lb = C.c_size_t(), C.c_size_t(), C.c_uint32()
    L.og_trim_aln(t, len(t), trim, C.byref(lo), C.byref(ro), C.byref(lb))
    return q[lo.value:ro.value], t[lo.value:ro.value], start + lb.value


def ref_trim_aln(q: bytes, t: bytes, start: int, trim: int):
    R = ref_lib()
    qb, tb = C.create_string_buffer(q, len(q) + 1), C.create_string_buffer(t, len(t) + 1)
    s = C.c_uint32(start)
    m = R.ref_trim_aln(qb, tb, len(q), trim, C.byref(s))
    return qb.raw[:m], tb.raw[:m], s.value


class Parsed(C.Structure):
    _fields_ = [("id", C.c_char_p), ("sid", C.c_char_p), ("qstr", C.c_char_p),
                ("tstr", C.c_char_p), ("tlen", C.c_uint32), ("start", C.c_uint32),
                ("strand", C.c_char)]


def parse_m5(line: bytes, group_by_target: bool = True):
    L = lib()
    p = Parsed()
    rc = L.og_parse_m5(line, len(line), int(group_by_target), C.byref(p))
    if rc != 1:
        return None
    out = dict(id=p.id, sid=p.sid, qstr=p.qstr, tstr=p.tstr, tlen=p.tlen, start=p.start,
               strand=p.strand)
    L.og_free_parsed(C.byref(p))
    return out


def parse_pre(line: bytes):
    """Alignment.cpp:82-112 (restatement): dict with start as parsed (no +1) and end."""
    L = lib()
    p = Parsed()
    end = C.c_uint32()
    L.og_parse_pre.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Parsed), C.POINTER(C.c_uint32)]
    rc = L.og_parse_pre(line, len(line), C.byref(p), C.byref(end))
    if rc != 1:
        return None if rc == 0 else rc
    out = dict(id=p.id, sid=p.sid, qstr=p.qstr, tstr=p.tstr, tlen=p.tlen, start=p.start, end=end.value,
               strand=p.strand)
    L.og_free_parsed(C.byref(p))
    return out


def ref_parse_pre(line: bytes):
    """The reference's own parsePre (oracle/_ref)."""
    R = ref_lib()
    n = len(line) + 1
    bufs = [C.create_string_buffer(n) for _ in range(4)]
    tlen, start, end, strand = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.create_string_buffer(2)
    R.ref_parse_pre(line, len(line), bufs[0], bufs[1], bufs[2], bufs[3], C.byref(tlen), C.byref(start),
                    C.byref(end), strand)
    return dict(id=bufs[0].value, sid=bufs[1].value, qstr=bufs[2].value, tstr=bufs[3].value,
                tlen=tlen.value, start=start.value, end=end.value, strand=strand.raw[:1])


def banded_align(q: bytes, t: bytes):
    """The -a re-aligner's restatement (parity unpinned beyond SimpleAlignerTest.cpp:8-21): (qaln, taln)."""
    L = lib()
    L.og_banded_align.restype = C.c_size_t
    L.og_banded_align.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p]
    qa, ta = C.create_string_buffer(len(q) + len(t) + 1), C.create_string_buffer(len(q) + len(t) + 1)
    n = L.og_banded_align(q, len(q), t, len(t), qa, ta)
    return qa.raw[:n], ta.raw[:n]


def align_halfwidth(qlen: int, tlen: int) -> int:
    L = lib()
    L.og_align_halfwidth.restype = C.c_uint32
    L.og_align_halfwidth.argtypes = [C.c_uint32, C.c_uint32]
    return L.og_align_halfwidth(qlen, tlen)


def simple_align(start: int, tlen: int, strand: bytes, qseq: bytes, tseq: bytes):
    """SimpleAligner::align (SimpleAligner.cpp:25-63) on a parsePre record: (start, end, qstr, tstr)."""
    L = lib()
    qa, ta = banded_align(qseq, tseq)
    qb, tb = C.create_string_buffer(qa, len(qa) + 1), C.create_string_buffer(ta, len(ta) + 1)
    s, e = C.c_uint32(start), C.c_uint32(0)
    L.og_simple_aligner_finish.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_char,
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.og_simple_aligner_finish.restype = None
    L.og_simple_aligner_finish(qb, tb, len(qa), len(tseq), tlen, strand[:1], C.byref(s), C.byref(e))
    return s.value, e.value, qb.raw[:len(qa)], tb.raw[:len(ta)]


def ref_parse_m5(line: bytes, group_by_target: bool = True):
    R = ref_lib()
    n = len(line) + 1
    bufs = [C.create_string_buffer(n) for _ in range(4)]
    tlen, start, strand = C.c_uint32(), C.c_uint32(), C.create_string_buffer(2)
    R.ref_parse_m5(line, len(line), int(group_by_target), bufs[0], bufs[1], bufs[2], bufs[3],
                   C.byref(tlen), C.byref(start), strand)
    return dict(id=bufs[0].value, sid=bufs[1].value, qstr=bufs[2].value, tstr=bufs[3].value,
                tlen=tlen.value, start=start.value, strand=strand.raw[:1])


class Graph:
    """Thin handle on og_graph for the unit-level KATs."""

    def __init__(self, backbone: bytes | None = None, blen: int | None = None):
        self.L = lib()
        if backbone is not None:
            self.g = self.L.og_graph_new_seq(backbone, len(backbone))
        else:
            self.g = self.L.og_graph_new_len(blen)

    def __del__(self):
        if getattr(self, "g", None):
            self.L.og_graph_free(self.g)
            self.g = None

    def add_aln(self, start: int, q: bytes, t: bytes):
        assert len(q) == len(t)
        self.L.og_add_aln(self.g, start, q, t, len(q))

    def merge_nodes(self) -> int:
        return self.L.og_merge_nodes(self.g)

    def consensus_longest(self, min_weight: int = 0) -> bytes:
        p = self.L.og_consensus_longest(self.g, min_weight)
        s = C.string_at(p)
        _libc_free(p)
        return s

    def consensus_all(self, min_weight: int = 0, min_len: int = 500):
        segs = C.POINTER(Segment)()
        n = self.L.og_consensus_all(self.g, min_weight, min_len, C.byref(segs))
        out = [(segs[i].range0, segs[i].range1, segs[i].seq) for i in range(n)]
        self.L.og_free_segments(segs, n)
        return out

    def best_path(self):
        p = C.POINTER(C.c_int32)()
        n = self.L.og_best_path(self.g, C.byref(p))
        out = [p[i] for i in range(n)]
        _libc_free(p)
        return out

    def dangling_nodes(self) -> bool:
        return bool(self.L.og_dangling_nodes(self.g))

    def num_nodes(self):
        return self.L.og_num_nodes(self.g)

    def live_counts(self):
        return self.L.og_num_live_nodes(self.g), self.L.og_num_live_edges(self.g)

    def adjacency(self):
        """Same shape as pymodel.AlnGraph.adjacency()."""
        out = []
        cap = 4096
        a, b = (C.c_int32 * cap)(), (C.c_int32 * cap)()
        for v in range(self.num_nodes()):
            base = C.create_string_buffer(1)
            w, cv, d, bb, bm = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int64()
            self.L.og_node_info(self.g, v, base, C.byref(w), C.byref(cv), C.byref(d),
                                C.byref(bb), C.byref(bm))
            n = self.L.og_out_edges(self.g, v, a, b, cap)
            oe = [(a[i], b[i]) for i in range(n)]
            n = self.L.og_in_edges(self.g, v, a, b, cap)
            ie = [(a[i], b[i]) for i in range(n)]
            out.append((base.raw.decode("latin1"), w.value, cv.value, bool(d.value), oe, ie))
        return out


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _libc_free(p):
    _libc.free(C.cast(p, C.c_void_p))


def consensus_target(tlen: int, alns, min_len=500, trim=50, min_weight=6, backbone=None):
    """main.cpp:130-138 for one target.  alns = [(start, qstr, tstr)] (bytes).
    Returns [(range0, range1, seq)] or raises ValueError on a non-conforming alignment."""
    import numpy as np
    L = lib()
    n = len(alns)
    starts = np.array([a[0] for a in alns], dtype=np.uint32)
    lens = np.array([len(a[1]) for a in alns], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    if n:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    qblob = b"".join(a[1] for a in alns)
    tblob = b"".join(a[2] for a in alns)
    return consensus_target_blob(tlen, starts, offs, lens, qblob, tblob, min_len, trim,
                                 min_weight, backbone)


_CF = None


def faithful_lib():
    """libcpu_faithful.so: the algorithm on the reference's container classes."""
    global _CF
    if _CF is None:
        so = os.path.join(_HERE, "libcpu_faithful.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libcpu_faithful.so"])
        F = C.CDLL(so)
        F.cf_consensus_target_blob.restype = C.c_long
        F.cf_consensus_target_blob.argtypes = [
            C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
            C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.POINTER(Opts),
            C.POINTER(C.POINTER(Segment)), C.POINTER(C.c_long)]
        _CF = F
    return _CF


def consensus_target_blob(tlen, starts, offs, lens, qblob, tblob, min_len=500, trim=50,
                          min_weight=6, backbone=None, faithful=False):
    import numpy as np
    L = lib()
    o = Opts(min_len, trim, min_weight)
    segs = C.POINTER(Segment)()
    bad = C.c_long(-1)
    qp = qblob.ctypes.data if isinstance(qblob, np.ndarray) else C.cast(C.c_char_p(qblob), C.c_void_p)
    tp = tblob.ctypes.data if isinstance(tblob, np.ndarray) else C.cast(C.c_char_p(tblob), C.c_void_p)
    fn = faithful_lib().cf_consensus_target_blob if faithful else L.og_consensus_target_blob
    rc = fn(
        tlen, backbone, len(starts),
        starts.ctypes.data_as(C.POINTER(C.c_uint32)),
        offs.ctypes.data_as(C.POINTER(C.c_uint64)),
        lens.ctypes.data_as(C.POINTER(C.c_uint32)),
        qp, tp, C.byref(o), C.byref(segs), C.byref(bad))
    if rc == -2:
        raise ValueError(f"non-conforming alignment #{bad.value}")
    if rc < 0:
        raise RuntimeError("oracle hit a state the reference treats as undefined")
    out = [(segs[i].range0, segs[i].range1, segs[i].seq) for i in range(rc)]
    L.og_free_segments(segs, rc)
    return out
