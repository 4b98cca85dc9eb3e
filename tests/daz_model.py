"""Python restatement of dazcon's first-party front-end arithmetic (test infrastructure):
TargetHit::add / computeOvlScore (DazAlnProvider.cpp:171-211), Target::addRecord / sortHits /
getAlignments (:264-369), decodeAlignment (:383-417).  float32 where the reference uses float."""
import numpy as np

F = np.float32
ToU = "ACGT.[]-"


class Hit:
    def __init__(self, aread, bread, flags, alen, blen):
        self.aread, self.bread, self.flags, self.alen, self.blen = aread, bread, flags, alen, blen
        self.records, self.ovl, self.cov = [], F(0), F(0)

    def belongs(self, r):
        return (self.aread, self.bread, self.flags) == (r["aread"], r["bread"], r["flags"])

    def add(self, r):                                   # :171-188
        if not self.records:
            self.records.append(r); return
        prev = self.records[-1]
        if r["abpos"] > prev["aepos"]:
            self.records.append(r)
        elif r["aepos"] - r["abpos"] > prev["aepos"] - prev["abpos"]:
            self.records[-1] = r

    def score(self, proper=False):                      # :190-211
        ah = bh = diff = 0
        for r in self.records:
            ah += r["aepos"] - r["abpos"]; bh += r["bepos"] - r["bbpos"]
            diff += abs(ah - bh) + r["diffs"]
        with np.errstate(divide="ignore", invalid="ignore"):
            self.ovl = (F(1) - F(diff) / F(ah)) * F(ah)
        if proper:
            f, b = self.records[0], self.records[-1]
            if f["abpos"] != 0 and b["bbpos"] != 0:
                self.ovl = F(0)
            if f["aepos"] != self.alen and b["bepos"] != self.blen:
                self.ovl = F(0)


def group_hits(records, alen, blens, proper=False):     # Target::firstRecord / addRecord :229-283
    hits = []
    for r in records:
        if hits and hits[-1].belongs(r):
            hits[-1].add(r); hits[-1].score(proper); continue
        h = Hit(r["aread"], r["bread"], r["flags"], alen, blens[r["bread"]])
        h.add(r); h.score(proper); hits.append(h)
    return hits


def sort_hits(hits, alen, sort_cov):                    # :285-302 (stable where std::sort leaves ties open)
    hits = sorted(hits, key=lambda h: -float(h.ovl))
    if not sort_cov:
        return hits
    cov = [0] * alen
    for h in hits:
        for r in h.records:
            acc = 0.0
            for i in range(r["abpos"], r["aepos"]):
                cov[i] += 1
            for i in range(r["abpos"], r["aepos"]):
                acc = float(F(acc) + F(1) / F(cov[i]))   # invertedSum(float, unsigned) through a double accumulator
            h.cov = F(acc)
    return sorted(hits, key=lambda h: -float(h.cov))


def decode(a, b, r):                                    # :383-417; past a read's end the buffer terminator '.' is read
    A = lambda i: a[i] if 0 <= i < len(a) else "."
    B = lambda j: b[j] if 0 <= j < len(b) else "."
    t, q, i, j = [], [], r["abpos"], r["bbpos"]
    for p in r["trace"]:
        if p < 0:
            p = -p
            while i != p:
                t.append(A(i)); q.append(B(j)); i += 1; j += 1
            t.append("-"); q.append(B(j)); j += 1
        else:
            while j != p:
                t.append(A(i)); q.append(B(j)); i += 1; j += 1
            t.append(A(i)); q.append("-"); i += 1
    while i <= r["aepos"]:
        t.append(A(i)); q.append(B(j)); i += 1; j += 1
    return "".join(t), "".join(q)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def synth_dataset(rng, n_targets=3, tlen=(1500, 2600), n_b=(8, 14), raw_fraction=0.5):
    """Reads plus overlap records whose alignments are known: returns (reads dict 1-based, record lines,
    per target [(start, qstr, tstr)] as dazcon must hand them to the consensus, given the default flags)."""
    reads, lines, model = {}, [], {}
    rid = 0
    for _ in range(n_targets):
        rid += 1
        aid = rid
        a = "".join("ACGT"[k] for k in rng.integers(0, 4, int(rng.integers(*tlen))))
        reads[aid] = a
        recs = []
        for _ in range(int(rng.integers(*n_b))):
            rid += 1
            abpos = int(rng.integers(1, len(a) // 4)); aepos = int(rng.integers(3 * len(a) // 4, len(a) - 1))
            comp = bool(rng.integers(0, 2))
            lead = "".join("ACGT"[k] for k in rng.integers(0, 4, int(rng.integers(1, 30))))
            # walk a[abpos .. aepos] (aepos inclusive: decodeAlignment's last loop runs to i <= aepos)
            bseq, trace, cols_t, cols_q = list(lead), [], [], []
            i, j = abpos, len(lead)
            diffs = 0
            while i <= aepos:
                u = rng.random()
                if u < 0.05 and i < aepos and i > abpos:                    # a base of A with no partner
                    trace.append(j); cols_t.append(a[i]); cols_q.append("-"); i += 1; diffs += 1
                elif u < 0.12 and i > abpos and i < aepos:                 # a base of B with no partner
                    c = "ACGT"[rng.integers(0, 4)]
                    trace.append(-i); cols_t.append("-"); cols_q.append(c); bseq.append(c); j += 1; diffs += 1
                else:
                    c = a[i] if rng.random() > 0.02 else "ACGT"[rng.integers(0, 4)]
                    diffs += c != a[i]
                    cols_t.append(a[i]); cols_q.append(c); bseq.append(c); i += 1; j += 1
            bbpos, bepos = len(lead), j - 1
            b_oriented = "".join(bseq) + "".join("ACGT"[k] for k in rng.integers(0, 4, int(rng.integers(1, 30))))
            reads[rid] = revcomp(b_oriented) if comp else b_oriented         # the .db holds the read as sequenced
            r = dict(aread=aid - 1, bread=rid - 1, flags=int(comp), abpos=abpos, aepos=aepos, bbpos=bbpos, bepos=bepos,
                     diffs=diffs, trace=trace, tstr="".join(cols_t), qstr="".join(cols_q), b_oriented=b_oriented)
            recs.append(r)
        model[aid] = recs
        for r in recs:
            head = f"{r['aread'] + 1} {r['bread'] + 1} {r['flags']} {r['abpos']} {r['aepos']} {r['bbpos']} {r['bepos']} {r['diffs']}"
            if rng.random() < raw_fraction:
                lines.append(f"R {head} " + (",".join(map(str, r["trace"])) if r["trace"] else "-"))
            else:
                lines.append(f"O {head} {r['tstr']} {r['qstr']}")
    return reads, lines, model
