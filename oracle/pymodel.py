"""pymodel.py -- literal pure-Python model of the pbdagcon consensus hot path.

TEST INFRASTRUCTURE ONLY (small cases; it is slow on purpose).  An independent
restatement used to cross-check oracle/dagcon_oracle.c: it keeps explicit edge
objects with `visited` flags and Python lists for the ordered adjacency of
boost::adjacency_list<vecS,vecS,bidirectionalS>, and follows the reference
statement by statement (file:line relative to /root/reference).
"""
from __future__ import annotations

FLT_MAX = 3.4028234663852886e38


# --- Alignment.cpp ---------------------------------------------------------

def normalize_gaps(q: str, t: str, push: bool = True):
    """Alignment.cpp:131-217."""
    assert len(q) == len(t)
    q = q.replace(".", "-")
    t = t.replace(".", "-")
    qn, tn = [], []
    for qb, tb in zip(q, t):
        if qb != tb and qb != "-" and tb != "-":
            qn += ["-", qb]
            tn += [tb, "-"]
        else:
            qn.append(qb)
            tn.append(tb)
    n = len(qn)
    if push and n > 0:
        for i in range(n - 1):
            if tn[i] == "-":
                j = i + 1
                while j < n:
                    c = tn[j]
                    if c != "-":
                        if c == qn[i]:
                            tn[i] = c
                            tn[j] = "-"
                        break
                    j += 1
            if qn[i] == "-":
                j = i + 1
                while j < n:
                    c = qn[j]
                    if c != "-":
                        if c == tn[i]:
                            qn[i] = c
                            qn[j] = "-"
                        break
                    j += 1
    qo = "".join(a for a, b in zip(qn, tn) if a != "-" or b != "-")
    to = "".join(b for a, b in zip(qn, tn) if a != "-" or b != "-")
    return qo, to


def trim_aln(q: str, t: str, start: int, trim_len: int = 50):
    """Alignment.cpp:219-242."""
    n = len(t)
    lbases = loffs = 0
    while lbases < trim_len and loffs < n:
        if t[loffs] != "-":
            lbases += 1
        loffs += 1
    rbases = 0
    roffs = n
    while rbases < trim_len and roffs > loffs:
        roffs -= 1
        if t[roffs] != "-":
            rbases += 1
    return q[loffs:roffs], t[loffs:roffs], start + lbases


# --- AlnGraphBoost.cpp -------------------------------------------------------

class Edge:
    __slots__ = ("src", "dst", "count", "visited")

    def __init__(self, src, dst):
        self.src, self.dst, self.count, self.visited = src, dst, 0, False


class Node:
    __slots__ = ("base", "coverage", "weight", "backbone", "deleted", "out", "inn")

    def __init__(self):
        self.base, self.coverage, self.weight = "N", 0, 0
        self.backbone = self.deleted = False
        self.out, self.inn = [], []


class AlnGraph:
    def __init__(self, backbone=None, blen=None):
        """AlnGraphBoost.cpp:16-39 / :41-62."""
        if backbone is not None:
            blen = len(backbone)
        self.nodes = [Node() for _ in range(blen + 2)]
        self.bbmap = {}
        for i in range(blen + 1):
            self._add_edge(i, i + 1)
        self.enter, self.exit = 0, blen + 1
        self.nodes[0].base, self.nodes[0].backbone = "^", True
        for i in range(blen):
            n = self.nodes[i + 1]
            n.backbone, n.weight = True, 1
            n.base = backbone[i] if backbone is not None else "N"
            self.bbmap[i + 1] = i + 1
        self.nodes[blen + 1].base, self.nodes[blen + 1].backbone = "$", True

    def _add_edge(self, u, v):
        e = Edge(u, v)
        self.nodes[u].out.append(e)
        self.nodes[v].inn.append(e)
        return e

    def _bb(self, v):
        return self.bbmap.setdefault(v, 0)   # std::map operator[]

    def add_edge(self, u, v):
        """AlnGraphBoost.cpp:109-127."""
        exists = False
        for e in self.nodes[v].inn:
            if e.src == u:
                e.count += 1
                exists = True
        if not exists:
            self._add_edge(u, v).count += 1

    def add_aln(self, start, q, t):
        """AlnGraphBoost.cpp:64-107."""
        bbpos, prev = start, self.enter
        for qb, tb in zip(q, t):
            curr = bbpos
            if qb == tb:
                bb = self.nodes[self._bb(curr)]
                bb.coverage += 1
                bb.base = tb
                self.nodes[curr].weight += 1
                self.add_edge(prev, curr)
                bbpos += 1
                prev = curr
            elif qb == "-" and tb != "-":
                bb = self.nodes[self._bb(curr)]
                bb.coverage += 1
                bb.base = tb
                bbpos += 1
            elif qb != "-" and tb == "-":
                self.nodes.append(Node())
                nv = len(self.nodes) - 1
                self.nodes[nv].base = qb
                self.nodes[nv].weight += 1
                self.bbmap[nv] = bbpos
                self.add_edge(prev, nv)
                prev = nv
        self.add_edge(prev, self.exit)

    def _find_edge(self, u, v):
        for e in self.nodes[u].out:
            if e.dst == v:
                return e
        return None

    def _reap(self, n):
        """AlnGraphBoost.cpp:269-273 + boost::clear_vertex."""
        nd = self.nodes[n]
        nd.deleted = True
        for e in nd.out:
            lst = self.nodes[e.dst].inn
            lst[:] = [x for x in lst if x is not e]
        for e in nd.inn:
            lst = self.nodes[e.src].out
            lst[:] = [x for x in lst if x is not e]
        nd.out, nd.inn = [], []

    def merge_in(self, n):
        """AlnGraphBoost.cpp:162-215."""
        groups = {}
        for e in self.nodes[n].inn:
            s = e.src
            if len(self.nodes[s].out) == 1:
                groups.setdefault(self.nodes[s].base, []).append(s)
        for base in sorted(groups):
            nodes = list(groups[base])
            if len(nodes) <= 1:
                continue
            an = nodes[0]
            for ni in nodes[1:]:
                self.nodes[an].out[0].count += self.nodes[ni].out[0].count
                self.nodes[an].weight += self.nodes[ni].weight
            for v in nodes[1:]:
                for ie in list(self.nodes[v].inn):
                    n1 = ie.src
                    e = self._find_edge(n1, an)
                    if e is not None:
                        e.count += ie.count
                    else:
                        ne = self._add_edge(n1, an)
                        ne.count, ne.visited = ie.count, ie.visited
                self._reap(v)
            self.merge_in(an)

    def merge_out(self, n):
        """AlnGraphBoost.cpp:217-267."""
        groups = {}
        for e in self.nodes[n].out:
            d = e.dst
            if len(self.nodes[d].inn) == 1:
                groups.setdefault(self.nodes[d].base, []).append(d)
        for base in sorted(groups):
            nodes = list(groups[base])
            if len(nodes) <= 1:
                continue
            an = nodes[0]
            for ni in nodes[1:]:
                self.nodes[an].inn[0].count += self.nodes[ni].inn[0].count
                self.nodes[an].weight += self.nodes[ni].weight
            for v in nodes[1:]:
                for oe in list(self.nodes[v].out):
                    n2 = oe.dst
                    e = self._find_edge(an, n2)
                    if e is not None:
                        e.count += oe.count
                    else:
                        ne = self._add_edge(an, n2)
                        ne.count, ne.visited = oe.count, oe.visited
                self._reap(v)

    def merge_nodes(self):
        """AlnGraphBoost.cpp:129-160."""
        queue = [self.enter]
        head = 0
        while head < len(queue):
            u = queue[head]
            head += 1
            self.merge_in(u)
            self.merge_out(u)
            for e in self.nodes[u].out:
                e.visited = True
                v = e.dst
                if all(x.visited for x in self.nodes[v].inn):
                    queue.append(v)
        return queue

    def best_path(self):
        """AlnGraphBoost.cpp:375-459.  Scores are kept as Python floats; every
        value is a multiple of 0.5 far below 2**23, so fp32 == fp64 here."""
        for nd in self.nodes:
            for e in nd.out:
                e.visited = False
        best_edge, score = {}, {self.exit: 0.0}
        queue, head = [self.exit], 0
        while head < len(queue):
            n = queue[head]
            head += 1
            found, best, best_e = False, -FLT_MAX, None
            for oe in self.nodes[n].out:
                tn = self.nodes[oe.dst]
                s = score.setdefault(oe.dst, 0.0)
                if tn.backbone and tn.weight == 1:
                    ns = s - 10.0
                else:
                    bb = self.nodes[self._bb(oe.dst)]
                    ns = oe.count - bb.coverage * 0.5 + s
                if ns > best:
                    best, best_e, found = ns, oe, True
            if found:
                score[n] = best
                best_edge[n] = best_e
            for ie in self.nodes[n].inn:
                ie.visited = True
                if all(x.visited for x in self.nodes[ie.src].out):
                    queue.append(ie.src)
        path, prev = [], self.enter
        while True:
            path.append(prev)
            if prev not in best_edge:
                break
            prev = best_edge[prev].dst
        return path

    def consensus_all(self, min_weight=0, min_len=500):
        """AlnGraphBoost.cpp:327-373."""
        path = self.best_path()
        eb, xb = self.nodes[self.enter].base, self.nodes[self.exit].base
        cns, segs = [], []
        offs = idx = 0
        met = False
        for v in path:
            n = self.nodes[v]
            if n.base == eb or n.base == xb:
                continue
            cns.append(n.base)
            if not met and n.weight >= min_weight:
                offs, met = idx, True
            elif met and n.weight < min_weight:
                met = False
                if idx - offs >= min_len:
                    segs.append((offs, idx, "".join(cns[offs:idx])))
            idx += 1
        if met and idx - offs >= min_len:
            segs.append((offs, idx, "".join(cns[offs:idx])))
        return segs

    def consensus_longest(self, min_weight=0):
        """AlnGraphBoost.cpp:285-325."""
        path = self.best_path()
        eb, xb = self.nodes[self.enter].base, self.nodes[self.exit].base
        cns = []
        offs = best_offs = length = idx = 0
        met = False
        for v in path:
            n = self.nodes[v]
            if n.base == eb or n.base == xb:
                continue
            cns.append(n.base)
            if not met and n.weight >= min_weight:
                offs, met = idx, True
            elif met and n.weight < min_weight:
                if idx - offs > length:
                    best_offs, length = offs, idx - offs
                met = False
            idx += 1
        if met and idx - offs > length:
            best_offs, length = offs, idx - offs
        return "".join(cns[best_offs:best_offs + length])

    def dangling_nodes(self):
        """AlnGraphBoost.cpp:468-487."""
        eb, xb = self.nodes[self.enter].base, self.nodes[self.exit].base
        found = False
        for n in self.nodes:
            if n.deleted or n.base == eb or n.base == xb:
                continue
            if n.out and n.inn:
                continue
            found = True
        return found

    def adjacency(self):
        """(base, weight, coverage, deleted, [(dst,count)...], [(src,count)...]) per node."""
        return [
            (n.base, n.weight, n.coverage, n.deleted,
             [(e.dst, e.count) for e in n.out], [(e.src, e.count) for e in n.inn])
            for n in self.nodes
        ]


def consensus_target(tlen, alns, min_len=500, trim=50, min_weight=6, backbone=None):
    """main.cpp:130-138 for one target.  alns = [(start, qstr, tstr), ...]."""
    g = AlnGraph(backbone=backbone, blen=tlen)
    for start, q, t in alns:
        if len(q) < min_len:
            continue
        qn, tn = normalize_gaps(q, t)
        qn, tn, s = trim_aln(qn, tn, start, trim)
        g.add_aln(s, qn, tn)
    g.merge_nodes()
    return g.consensus_all(min_weight, min_len), g
