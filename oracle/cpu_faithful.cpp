// cpu_faithful.cpp -- CPU restatement of the hot path with the reference's CONTAINER CLASSES.
//
// TEST INFRASTRUCTURE ONLY (see dagcon_oracle.h).  dagcon_oracle.c restates the algorithm on
// flat arrays, which is considerably faster than what the reference runs on.  This file is the
// second flavour SURVEY.md section 8(d) / BASELINE.md section 3 ask for: the same algorithm on the
// same kind of containers as the reference, so that a CPU baseline timed on it says what the
// reference's data structures cost:
//   * the graph is an adjacency list in the shape of boost::adjacency_list<vecS,vecS,
//     bidirectionalS>: per-vertex std::vector out / in lists of (neighbour, iterator into a global
//     std::list of edge properties); clear_vertex erases with remove_if from the neighbours' vectors
//     (src/cpp/AlnGraphBoost.hpp:16,51);
//   * _bbMap, nodeScore, bestNodeScoreEdge are std::map (AlnGraphBoost.hpp:141, .cpp:380-381);
//   * merge groups are std::map<char, std::vector<>> built per visit (AlnGraphBoost.cpp:163,218);
//   * std::queue for both sweeps; bestPath returns copies of the vertex properties (:402,:407,:445);
//   * normalizeGaps builds std::string with += and copies (Alignment.cpp:131-217).
// It is written from the behaviour of the reference, not from its text, and is checked against
// dagcon_oracle.c in tests/test_oracle.py.
#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <list>
#include <map>
#include <queue>
#include <string>
#include <vector>

#include "dagcon_oracle.h"

namespace {

struct EdgeProp { int count = 0; bool visited = false; };
typedef std::list<EdgeProp>::iterator EdgeIt;
struct Stored { size_t other; EdgeIt prop; };            // an out entry (target) or in entry (source)
struct VertexProp { char base = 'N'; int coverage = 0, weight = 0; bool backbone = false, deleted = false; };
struct Vertex { VertexProp p; std::vector<Stored> out, in; };
struct EdgeDesc { size_t src = 0, dst = 0; EdgeIt prop; };

struct Graph {
    std::vector<Vertex> v;
    std::list<EdgeProp> edges;
    std::map<size_t, size_t> bbMap;
    size_t enterVtx = 0, exitVtx = 0;

    size_t add_vertex() { v.push_back(Vertex()); return v.size() - 1; }
    EdgeDesc add_edge(size_t s, size_t d) {
        edges.push_back(EdgeProp());
        EdgeIt it = --edges.end();
        v[s].out.push_back(Stored{d, it});
        v[d].in.push_back(Stored{s, it});
        EdgeDesc e; e.src = s; e.dst = d; e.prop = it;
        return e;
    }
    bool find_edge(size_t s, size_t d, EdgeDesc &e) {
        for (auto &o : v[s].out) if (o.other == d) { e.src = s; e.dst = d; e.prop = o.prop; return true; }
        return false;
    }
    void clear_vertex(size_t n) {
        for (auto &o : v[n].out) {
            auto &lst = v[o.other].in;
            lst.erase(std::remove_if(lst.begin(), lst.end(), [&](const Stored &x) { return x.other == n; }), lst.end());
            edges.erase(o.prop);
        }
        for (auto &i : v[n].in) {
            auto &lst = v[i.other].out;
            lst.erase(std::remove_if(lst.begin(), lst.end(), [&](const Stored &x) { return x.other == n; }), lst.end());
            edges.erase(i.prop);
        }
        v[n].out.clear(); v[n].in.clear();
    }
};

// AlnGraphBoost.cpp:16-62
void graph_init(Graph &g, const char *backbone, size_t blen) {
    g.v.assign(blen + 2, Vertex());
    for (size_t i = 0; i < blen + 1; i++) g.add_edge(i, i + 1);
    g.enterVtx = 0;
    g.v[0].p.base = '^'; g.v[0].p.backbone = true;
    for (size_t i = 0; i < blen; i++) {
        Vertex &n = g.v[i + 1];
        n.p.backbone = true; n.p.weight = 1;
        n.p.base = backbone ? backbone[i] : 'N';
        g.bbMap[i + 1] = i + 1;
    }
    g.exitVtx = blen + 1;
    g.v[blen + 1].p.base = '$'; g.v[blen + 1].p.backbone = true;
}

// AlnGraphBoost.cpp:109-127
void add_edge_counted(Graph &g, size_t u, size_t v) {
    bool exists = false;
    for (auto &i : g.v[v].in) if (i.other == u) { i.prop->count++; exists = true; }
    if (!exists) { EdgeDesc e = g.add_edge(u, v); e.prop->count++; }
}

// AlnGraphBoost.cpp:64-107
void add_aln(Graph &g, uint32_t start, const std::string &q, const std::string &t) {
    uint32_t bbPos = start;
    size_t prev = g.enterVtx;
    for (size_t i = 0; i < q.length(); i++) {
        char qb = q[i], tb = t[i];
        size_t cur = bbPos;
        if (qb == tb) {
            g.v[g.bbMap[cur]].p.coverage++;
            g.v[g.bbMap[cur]].p.base = tb;
            g.v[cur].p.weight++;
            add_edge_counted(g, prev, cur);
            bbPos++;
            prev = cur;
        } else if (qb == '-' && tb != '-') {
            g.v[g.bbMap[cur]].p.coverage++;
            g.v[g.bbMap[cur]].p.base = tb;
            bbPos++;
        } else if (qb != '-' && tb == '-') {
            size_t nv = g.add_vertex();
            g.v[nv].p.base = qb;
            g.v[nv].p.weight++;
            g.bbMap[nv] = bbPos;
            add_edge_counted(g, prev, nv);
            prev = nv;
        }
    }
    add_edge_counted(g, prev, g.exitVtx);
}

void mark_for_reaper(Graph &g, size_t n) { g.v[n].p.deleted = true; g.clear_vertex(n); }

// AlnGraphBoost.cpp:162-215
void merge_in_nodes(Graph &g, size_t n) {
    std::map<char, std::vector<size_t>> groups;
    for (auto &i : g.v[n].in) {
        size_t s = i.other;
        if (g.v[s].out.size() == 1) groups[g.v[s].p.base].push_back(s);
    }
    for (auto kv = groups.cbegin(); kv != groups.cend(); ++kv) {
        std::vector<size_t> nodes = kv->second;
        if (nodes.size() <= 1) continue;
        size_t an = nodes[0];
        for (size_t k = 1; k < nodes.size(); k++) {
            g.v[an].out[0].prop->count += g.v[nodes[k]].out[0].prop->count;
            g.v[an].p.weight += g.v[nodes[k]].p.weight;
        }
        for (size_t k = 1; k < nodes.size(); k++) {
            size_t vv = nodes[k];
            for (size_t x = 0; x < g.v[vv].in.size(); x++) {
                Stored ie = g.v[vv].in[x];
                EdgeDesc e;
                if (g.find_edge(ie.other, an, e)) e.prop->count += ie.prop->count;
                else {
                    EdgeDesc ne = g.add_edge(ie.other, an);
                    ne.prop->count = ie.prop->count;
                    ne.prop->visited = ie.prop->visited;
                }
            }
            mark_for_reaper(g, vv);
        }
        merge_in_nodes(g, an);
    }
}

// AlnGraphBoost.cpp:217-267
void merge_out_nodes(Graph &g, size_t n) {
    std::map<char, std::vector<size_t>> groups;
    for (auto &o : g.v[n].out) {
        size_t d = o.other;
        if (g.v[d].in.size() == 1) groups[g.v[d].p.base].push_back(d);
    }
    for (auto kv = groups.cbegin(); kv != groups.cend(); ++kv) {
        std::vector<size_t> nodes = kv->second;
        if (nodes.size() <= 1) continue;
        size_t an = nodes[0];
        for (size_t k = 1; k < nodes.size(); k++) {
            g.v[an].in[0].prop->count += g.v[nodes[k]].in[0].prop->count;
            g.v[an].p.weight += g.v[nodes[k]].p.weight;
        }
        for (size_t k = 1; k < nodes.size(); k++) {
            size_t vv = nodes[k];
            for (size_t x = 0; x < g.v[vv].out.size(); x++) {
                Stored oe = g.v[vv].out[x];
                EdgeDesc e;
                if (g.find_edge(an, oe.other, e)) e.prop->count += oe.prop->count;
                else {
                    EdgeDesc ne = g.add_edge(an, oe.other);
                    ne.prop->count = oe.prop->count;
                    ne.prop->visited = oe.prop->visited;
                }
            }
            mark_for_reaper(g, vv);
        }
    }
}

// AlnGraphBoost.cpp:129-160
void merge_nodes(Graph &g) {
    std::queue<size_t> seeds;
    seeds.push(g.enterVtx);
    while (!seeds.empty()) {
        size_t u = seeds.front();
        seeds.pop();
        merge_in_nodes(g, u);
        merge_out_nodes(g, u);
        for (size_t x = 0; x < g.v[u].out.size(); x++) {
            Stored &o = g.v[u].out[x];
            o.prop->visited = true;
            size_t v = o.other;
            int notVisited = 0;
            for (auto &i : g.v[v].in) if (!i.prop->visited) notVisited++;
            if (notVisited == 0) seeds.push(v);
        }
    }
}

// AlnGraphBoost.cpp:375-459: returns copies of the path's vertex properties
std::vector<VertexProp> best_path(Graph &g) {
    for (auto &e : g.edges) e.visited = false;
    std::map<size_t, EdgeDesc> bestEdge;
    std::map<size_t, float> nodeScore;
    std::queue<size_t> seeds;
    seeds.push(g.exitVtx);
    nodeScore[g.exitVtx] = 0.0f;
    while (!seeds.empty()) {
        size_t n = seeds.front();
        seeds.pop();
        bool found = false;
        float best = -FLT_MAX;
        EdgeDesc bestE;
        for (auto &o : g.v[n].out) {
            VertexProp outNode = g.v[o.other].p;                  // a copy, as the reference makes
            float newScore, score = nodeScore[o.other];
            if (outNode.backbone && outNode.weight == 1) newScore = score - 10.0f;
            else {
                VertexProp bbNode = g.v[g.bbMap[o.other]].p;      // another copy
                newScore = o.prop->count - bbNode.coverage * 0.5f + score;
            }
            if (newScore > best) { best = newScore; bestE.src = n; bestE.dst = o.other; bestE.prop = o.prop; found = true; }
        }
        if (found) { nodeScore[n] = best; bestEdge[n] = bestE; }
        for (auto &i : g.v[n].in) {
            i.prop->visited = true;
            size_t s = i.other;
            int notVisited = 0;
            for (auto &o : g.v[s].out) if (!o.prop->visited) notVisited++;
            if (notVisited == 0) seeds.push(s);
        }
    }
    std::vector<VertexProp> path;
    size_t prev = g.enterVtx;
    while (true) {
        path.push_back(g.v[prev].p);
        if (bestEdge.count(prev) == 0) break;
        prev = bestEdge[prev].dst;
        if (path.size() > g.v.size() + 1) break;
    }
    return path;
}

// Alignment.cpp:131-217 with std::string, as the reference builds it
void normalize_gaps(const std::string &q0, const std::string &t0, std::string &qo, std::string &to) {
    size_t len = q0.length();
    std::string qNorm, tNorm;
    qNorm.reserve(len + 100); tNorm.reserve(len + 100);
    std::string qstr = q0, tstr = t0;
    for (size_t i = 0; i < len; i++) { if (qstr[i] == '.') qstr[i] = '-'; if (tstr[i] == '.') tstr[i] = '-'; }
    for (size_t i = 0; i < len; i++) {
        char qb = qstr[i], tb = tstr[i];
        if (qb != tb && qb != '-' && tb != '-') { qNorm += '-'; qNorm += qb; tNorm += tb; tNorm += '-'; }
        else { qNorm += qb; tNorm += tb; }
    }
    len = qNorm.length();
    if (len > 0) {
        for (size_t i = 0; i < len - 1; i++) {
            if (tNorm[i] == '-') {
                size_t j = i;
                while (++j < len) { char c = tNorm[j]; if (c != '-') { if (c == qNorm[i]) { tNorm[i] = c; tNorm[j] = '-'; } break; } }
            }
            if (qNorm[i] == '-') {
                size_t j = i;
                while (++j < len) { char c = qNorm[j]; if (c != '-') { if (c == tNorm[i]) { qNorm[i] = c; qNorm[j] = '-'; } break; } }
            }
        }
    }
    qo.clear(); to.clear();
    for (size_t i = 0; i < len; i++)
        if (qNorm[i] != '-' || tNorm[i] != '-') { qo += qNorm[i]; to += tNorm[i]; }
}

}  // namespace

// main.cpp:130-138 for one target on the reference's kind of containers; same contract as
// og_consensus_target_blob.
extern "C" long cf_consensus_target_blob(uint32_t tlen, const char *backbone, size_t n_alns,
                                         const uint32_t *starts, const uint64_t *offs,
                                         const uint32_t *lens, const char *qblob, const char *tblob,
                                         const og_opts *opts, og_segment **segs_out, long *bad_aln) {
    Graph g;
    graph_init(g, backbone, tlen);
    for (size_t a = 0; a < n_alns; a++) {
        std::string q(qblob + offs[a], lens[a]), t(tblob + offs[a], lens[a]);
        if (q.length() < opts->min_len) continue;
        std::string qn, tn;
        normalize_gaps(q, t, qn, tn);
        // Alignment.cpp:219-242 trimAln
        int lbases = 0, rbases = 0;
        size_t loffs = 0, roffs = tn.length();
        while (lbases < (int)opts->trim && loffs < tn.length()) { if (tn[loffs++] != '-') lbases++; }
        while (rbases < (int)opts->trim && roffs > loffs) { if (tn[--roffs] != '-') rbases++; }
        uint32_t start = starts[a] + (uint32_t)lbases;
        qn = qn.substr(loffs, roffs - loffs);
        tn = tn.substr(loffs, roffs - loffs);
        uint64_t tb = 0;
        for (char c : tn) tb += (c != '-');
        if (!tn.empty() && (start < 1 || (uint64_t)start - 1 + tb > (uint64_t)tlen)) {
            if (bad_aln) *bad_aln = (long)a;
            return -2;
        }
        add_aln(g, start, qn, tn);
    }
    merge_nodes(g);
    std::vector<VertexProp> path = best_path(g);
    // AlnGraphBoost.cpp:327-373
    std::string cns;
    std::vector<og_segment> segs;
    int offs_i = 0, idx = 0;
    bool met = false;
    char eb = g.v[g.enterVtx].p.base, xb = g.v[g.exitVtx].p.base;
    auto emit = [&](int a, int b) {
        size_t length = (size_t)(b - a);
        if (length >= opts->min_len) {
            og_segment s;
            s.range0 = a; s.range1 = b;
            s.seq = (char *)malloc(length + 1);
            memcpy(s.seq, cns.data() + a, length);
            s.seq[length] = 0;
            segs.push_back(s);
        }
    };
    for (auto &n : path) {
        if (n.base == eb || n.base == xb) continue;
        cns += n.base;
        if (!met && n.weight >= opts->min_weight) { offs_i = idx; met = true; }
        else if (met && n.weight < opts->min_weight) { met = false; emit(offs_i, idx); }
        idx++;
    }
    if (met) emit(offs_i, idx);
    og_segment *out = (og_segment *)malloc((segs.size() + 1) * sizeof(og_segment));
    for (size_t i = 0; i < segs.size(); i++) out[i] = segs[i];
    *segs_out = out;
    return (long)segs.size();
}
