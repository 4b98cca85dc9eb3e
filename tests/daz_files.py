"""This build's own WRITER of the two binary inputs of dazcon -- a DAZZ_DB database (.db stub + hidden .idx and .bps)
and a DALIGNER .las file -- for the round-trip tests of pbdagcon_amd/csrc/host/daz_io.h.  Layout as daz_io.h states
it (PARITY UNPINNED: no file DALIGNER or DAZZ_DB wrote is available, the reference holds no fixture at this
boundary; what the reference's own code shows is the .las header and the trace element width,
DazAlnProvider.cpp:48-63)."""
import os
import struct

DB_BEST = 0x0800


def write_db(path, reads, cutoff=0, all_=1, flags=None):
    """reads: list of ACGT strings (untrimmed order).  flags[i]: HITS_READ.flags (DB_BEST marks the read Trim_DB keeps
    of its well when `all` is 0)."""
    assert path.endswith(".db")
    d, root = os.path.dirname(path), os.path.basename(path)[:-3]
    flags = flags or [DB_BEST] * len(reads)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    bps, recs = bytearray(), []
    for i, s in enumerate(reads):
        boff = len(bps)
        for k in range(0, len(s), 4):
            q = s[k:k + 4].ljust(4, "A")
            bps.append(code[q[0]] << 6 | code[q[1]] << 4 | code[q[2]] << 2 | code[q[3]])
        # HITS_READ: origin, rlen, fpulse, (pad), boff, coff, flags, (pad)
        recs.append(struct.pack("<iiiiqqii", i, len(s), 0, 0, boff, -1, flags[i], 0))
    totlen, maxlen = sum(len(s) for s in reads), max([len(s) for s in reads] or [0])
    treads = sum(1 for i, s in enumerate(reads) if len(s) >= cutoff and (all_ or flags[i] & DB_BEST))
    # HITS_DB: ureads, treads, cutoff, all, freq[4], maxlen, (pad), totlen, nreads, trimmed, part, ufirst, tfirst, (pad),
    #          path, loaded, (pad), bases, reads, tracks
    hdr = struct.pack("<iiii4fiiqiiiiiiQiiQQQ", len(reads), treads, cutoff, all_, 0.25, 0.25, 0.25, 0.25, maxlen, 0, totlen,
                      len(reads), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    assert len(hdr) == 112
    open(os.path.join(d, "." + root + ".idx"), "wb").write(hdr + b"".join(recs))
    open(os.path.join(d, "." + root + ".bps"), "wb").write(bytes(bps))
    with open(path, "w") as f:
        f.write("files = %9d\n" % 1)
        f.write("  %9d %s %s\n" % (len(reads), "synth", "synth"))
        f.write("blocks = %9d\n" % 1)
        f.write("size = %10d cutoff = %9d all = %1d\n" % (200, cutoff, all_))
        f.write(" %9d %9d\n" % (0, 0))
        f.write(" %9d %9d\n" % (len(reads), treads))


def write_las(path, overlaps, tspace=100):
    """overlaps: dicts aread, bread (0-based, trimmed numbering), flags, abpos, aepos, bbpos, bepos, diffs, and
    optionally trace (a list of ints: per panel of tspace A bases a pair differences, B bases)."""
    tb = 1 if tspace <= 125 else 2
    with open(path, "wb") as f:
        f.write(struct.pack("<qi", len(overlaps), tspace))
        for o in overlaps:
            tr = o.get("trace")
            if tr is None:
                # plausible trace points: panels end at multiples of tspace on A; B bases spread evenly
                tr = []
                a, b = o["abpos"], o["bbpos"]
                while a < o["aepos"]:
                    na = min((a // tspace + 1) * tspace, o["aepos"])
                    nb = o["bbpos"] + (o["bepos"] - o["bbpos"]) * (na - o["abpos"]) // max(o["aepos"] - o["abpos"], 1)
                    tr += [0, min(nb - b, 255 if tb == 1 else 65535)]
                    a, b = na, nb
            f.write(struct.pack("<iiiiiiIiii", len(tr), o["diffs"], o["abpos"], o["bbpos"], o["aepos"], o["bepos"], o["flags"],
                                o["aread"], o["bread"], 0))
            f.write(struct.pack("<%d%s" % (len(tr), "B" if tb == 1 else "H"), *tr))
