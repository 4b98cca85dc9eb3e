/*
 * dagcon_oracle.c -- CPU restatement of the pbdagcon consensus hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see dagcon_oracle.h).  Written from the behaviour
 * of the reference, with flat arrays in place of Boost.Graph; the container
 * semantics it reproduces are those of
 * boost::adjacency_list<vecS,vecS,bidirectionalS>: stable vertex ids, ordered
 * per-vertex out- and in-lists, append on add_edge, stable erase on
 * clear_vertex, edge(u,v) = first match in out[u].
 */
#include "dagcon_oracle.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* small int vector                                                         */
/* ------------------------------------------------------------------------ */
typedef struct ivec { int32_t *v; size_t n, cap; } ivec;

static void iv_push(ivec *a, int32_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 2;
        a->v = (int32_t *)realloc(a->v, a->cap * sizeof(int32_t));
    }
    a->v[a->n++] = x;
}
static void iv_free(ivec *a) { free(a->v); a->v = NULL; a->n = a->cap = 0; }
/* stable removal of every entry equal to x */
static void iv_remove_all(ivec *a, int32_t x) {
    size_t w = 0;
    for (size_t r = 0; r < a->n; r++)
        if (a->v[r] != x) a->v[w++] = a->v[r];
    a->n = w;
}

/* ------------------------------------------------------------------------ */
/* Alignment.cpp                                                            */
/* ------------------------------------------------------------------------ */

/* Alignment.cpp:15-26 */
void og_revcomp(char *seq, size_t len) {
    for (size_t i = 0; i < len; i++) {
        char c = seq[i];
        seq[i] = c == 'T' ? 'A' : c == 'G' ? 'C' : c == 'A' ? 'T' : c == 'C' ? 'G' : c;
    }
    for (size_t i = 0, j = len; i + 1 < j; i++) {
        j--;
        char c = seq[i]; seq[i] = seq[j]; seq[j] = c;
    }
}

/* Alignment.cpp:131-217 */
size_t og_normalize_gaps(const char *q, const char *t, size_t len, int push,
                         char *qout, char *tout) {
    char *qn = (char *)malloc(2 * len + 1), *tn = (char *)malloc(2 * len + 1);
    size_t n = 0;
    /* :142-145 dots to dashes, :148-159 mismatches to indels */
    for (size_t i = 0; i < len; i++) {
        char qb = q[i] == '.' ? '-' : q[i];
        char tb = t[i] == '.' ? '-' : t[i];
        if (qb != tb && qb != '-' && tb != '-') {
            qn[n] = '-'; tn[n] = tb; n++;
            qn[n] = qb;  tn[n] = '-'; n++;
        } else {
            qn[n] = qb; tn[n] = tb; n++;
        }
    }
    /* :165-198 push gaps right; the loop bound is len-1 on an unsigned len,
     * so an empty alignment would wrap: the reference never gets one here
     * (main.cpp:132 filters on minLen) and we treat it as no work. */
    if (push && n > 0) {
        for (size_t i = 0; i < n - 1; i++) {
            if (tn[i] == '-') {
                size_t j = i;
                while (++j < n) {
                    char c = tn[j];
                    if (c != '-') {
                        if (c == qn[i]) { tn[i] = c; tn[j] = '-'; }
                        break;
                    }
                }
            }
            if (qn[i] == '-') {
                size_t j = i;
                while (++j < n) {
                    char c = qn[j];
                    if (c != '-') {
                        if (c == tn[i]) { qn[i] = c; qn[j] = '-'; }
                        break;
                    }
                }
            }
        }
    }
    /* :209-214 drop all-gap columns */
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        if (qn[i] != '-' || tn[i] != '-') { qout[m] = qn[i]; tout[m] = tn[i]; m++; }
    }
    qout[m] = 0; tout[m] = 0;
    free(qn); free(tn);
    return m;
}

/* Alignment.cpp:219-242 */
void og_trim_aln(const char *t, size_t len, int trim_len,
                 size_t *loffs_out, size_t *roffs_out, uint32_t *lbases_out) {
    int lbases = 0, rbases = 0;
    size_t loffs = 0, roffs = len;
    while (lbases < trim_len && loffs < len) {
        if (t[loffs++] != '-') lbases++;
    }
    while (rbases < trim_len && roffs > loffs) {
        if (t[--roffs] != '-') rbases++;
    }
    *loffs_out = loffs; *roffs_out = roffs; *lbases_out = (uint32_t)lbases;
}

static char *dup_range(const char *s, size_t n) {
    char *r = (char *)malloc(n + 1);
    memcpy(r, s, n); r[n] = 0;
    return r;
}

/* istringstream >> uint32_t on a token: leading digits, 0 on failure
 * (Alignment.cpp:63-66); overflow saturates like libstdc++ (UINT32_MAX). */
static uint32_t parse_u32(const char *s, size_t n) {
    size_t i = 0;
    uint64_t v = 0;
    int any = 0, over = 0, neg = 0;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    for (; i < n && s[i] >= '0' && s[i] <= '9'; i++) {
        if (!over) v = v * 10 + (uint64_t)(s[i] - '0');
        if (v > 0xFFFFFFFFull) over = 1;
        any = 1;
    }
    if (!any) return 0;
    if (over) return 0xFFFFFFFFu;
    return neg ? (uint32_t)(0u - (uint32_t)v) : (uint32_t)v;
}

/* Alignment.cpp:44-80 */
int og_parse_m5(const char *line, size_t len, int group_by_target, og_parsed *out) {
    const char *f[32]; size_t fl[32]; int nf = 0;
    size_t i = 0;
    while (i <= len) {
        size_t j = i;
        while (j < len && line[j] != ' ') j++;
        if (j > i && nf < 32) { f[nf] = line + i; fl[nf] = j - i; nf++; }
        i = j + 1;
    }
    if (nf == 0) return 0;
    if (nf < 19) return -1; /* the reference indexes fields[18]: UB */
    /* :58 baseQid = fields[0] up to the last '/' */
    size_t slash = fl[0];
    for (size_t k = fl[0]; k > 0; k--) if (f[0][k - 1] == '/') { slash = k - 1; break; }
    free(out->sid); free(out->id); free(out->qstr); free(out->tstr);
    out->sid = dup_range(f[0], fl[0]);
    out->id = group_by_target ? dup_range(f[5], fl[5]) : dup_range(f[0], slash);
    out->tlen = group_by_target ? parse_u32(f[6], fl[6]) : parse_u32(f[1], fl[1]);
    out->start = (group_by_target ? parse_u32(f[7], fl[7]) : parse_u32(f[2], fl[2])) + 1;
    out->strand = f[9][0];
    if (out->strand == '-' && group_by_target) {
        out->qstr = dup_range(f[16], fl[16]); og_revcomp(out->qstr, fl[16]);
        out->tstr = dup_range(f[18], fl[18]); og_revcomp(out->tstr, fl[18]);
    } else {
        out->qstr = group_by_target ? dup_range(f[16], fl[16]) : dup_range(f[18], fl[18]);
        out->tstr = group_by_target ? dup_range(f[18], fl[18]) : dup_range(f[16], fl[16]);
    }
    return 1;
}

void og_free_parsed(og_parsed *p) {
    free(p->id); free(p->sid); free(p->qstr); free(p->tstr);
    memset(p, 0, sizeof(*p));
}

/* Alignment.cpp:82-112 */
int og_parse_pre(const char *line, size_t len, og_parsed *out, uint32_t *end_out) {
    const char *f[16]; size_t fl[16]; int nf = 0;
    size_t i = 0;
    while (i <= len) {
        size_t j = i;
        while (j < len && line[j] != ' ') j++;
        if (j > i && nf < 16) { f[nf] = line + i; fl[nf] = j - i; nf++; }
        i = j + 1;
    }
    if (nf == 0) return 0;
    if (nf < 8) return -1;
    free(out->sid); free(out->id); free(out->qstr); free(out->tstr);
    out->sid = dup_range(f[0], fl[0]);          /* :99 */
    out->id = dup_range(f[1], fl[1]);           /* :100 */
    out->strand = f[2][0];                      /* :101 */
    out->tlen = parse_u32(f[3], fl[3]);         /* :103-104 */
    out->start = parse_u32(f[4], fl[4]);        /* :106-107 (no +1 here: SimpleAligner.cpp:61 adds it) */
    if (end_out) *end_out = parse_u32(f[5], fl[5]);   /* :109-110 */
    out->qstr = dup_range(f[6], fl[6]);         /* :112 */
    out->tstr = dup_range(f[7], fl[7]);
    return 1;
}

/* ---- the -a re-aligner: see dagcon_oracle.h (parity unpinned beyond the reference's one KAT) ---- */
static uint32_t isqrt_u64(uint64_t x) {
    uint64_t r = 0, b = 1ull << 62;
    while (b > x) b >>= 2;
    while (b) {
        if (x >= r + b) { x -= r + b; r = (r >> 1) + b; } else r >>= 1;
        b >>= 2;
    }
    return (uint32_t)r;
}

uint32_t og_align_halfwidth(uint32_t qlen, uint32_t tlen) {
    /* a random walk of indels at ~15 % leaves the scaled diagonal by ~sqrt(0.15 L); four of those, and room */
    const uint64_t L = qlen > tlen ? qlen : tlen;
    uint32_t w = 32u + 4u * isqrt_u64((15ull * L + 99ull) / 100ull);
    if (w > 480u) w = 480u;                     /* band of at most 961 cells per row */
    return w;
}

/* The band tried first: two of those.  The alignment found in it stands if its path keeps
 * OG_AL_MARGIN cells away from both edges of the band on every row; else the pair is done again in the full band. */
uint32_t og_align_halfwidth_first(uint32_t qlen, uint32_t tlen) {
    const uint64_t L = qlen > tlen ? qlen : tlen;
    uint32_t w = 32u + 2u * isqrt_u64((15ull * L + 99ull) / 100ull);
    const uint32_t full = og_align_halfwidth(qlen, tlen);
    return w > full ? full : w;
}
#define OG_AL_MARGIN 8

#define OG_AL_MATCH (-5)
#define OG_AL_MISMATCH 6
#define OG_AL_INS 4
#define OG_AL_DEL 5
#define OG_AL_INF (1 << 28)

/* the global alignment inside the band of half-width W; *touched: its path came within OG_AL_MARGIN cells of an edge */
static size_t og_banded_align_w(const char *q, uint32_t n, const char *t, uint32_t m, uint32_t W, char *qaln, char *taln,
                                int *touched) {
    const uint32_t B = 2 * W + 1;
    *touched = 0;
    /* row i holds columns j = c_i - W + k, k = 0..B-1, c_i = i * m / n */
    int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * B), *cur = (int32_t *)malloc(sizeof(int32_t) * B);
    uint8_t *dir = (uint8_t *)malloc((size_t)(n + 1) * B);
    for (uint32_t i = 0; i <= n; i++) {
        const int64_t ci = (int64_t)((uint64_t)i * m / n), cp = i ? (int64_t)((uint64_t)(i - 1) * m / n) : 0;
        for (uint32_t k = 0; k < B; k++) {
            const int64_t j = ci - (int64_t)W + k;
            int32_t best = OG_AL_INF; uint8_t d = 3;
            if (j >= 0 && j <= (int64_t)m) {
                if (i == 0 && j == 0) { best = 0; d = 3; }
                if (i > 0 && j > 0) {                       /* diagonal: (i-1, j-1) */
                    const int64_t kp = j - 1 - (cp - (int64_t)W);
                    if (kp >= 0 && kp < (int64_t)B && prev[kp] < OG_AL_INF) {
                        const int32_t v = prev[kp] + (q[i - 1] == t[j - 1] ? OG_AL_MATCH : OG_AL_MISMATCH);
                        if (v < best) { best = v; d = 0; }
                    }
                }
                if (i > 0) {                                /* insertion: (i-1, j), gap in the target */
                    const int64_t kp = j - (cp - (int64_t)W);
                    if (kp >= 0 && kp < (int64_t)B && prev[kp] < OG_AL_INF) {
                        const int32_t v = prev[kp] + OG_AL_INS;
                        if (v < best) { best = v; d = 1; }
                    }
                }
                if (j > 0 && k > 0 && cur[k - 1] < OG_AL_INF) {    /* deletion: (i, j-1), gap in the query */
                    const int32_t v = cur[k - 1] + OG_AL_DEL;
                    if (v < best) { best = v; d = 2; }
                }
            }
            cur[k] = best;
            dir[(size_t)i * B + k] = d;
        }
        int32_t *x = prev; prev = cur; cur = x;
    }
    /* traceback from (n, m) */
    size_t len = 0;
    char *rq = (char *)malloc((size_t)n + m + 1), *rt = (char *)malloc((size_t)n + m + 1);
    uint32_t i = n, j = m;
    while (i > 0 || j > 0) {
        const int64_t ci = (int64_t)((uint64_t)i * m / n);
        const int64_t kk = (int64_t)j - (ci - (int64_t)W);
        if (kk < OG_AL_MARGIN || kk > (int64_t)B - 1 - OG_AL_MARGIN) *touched = 1;
        const uint8_t d = dir[(size_t)i * B + (size_t)kk];
        if (d == 0) { rq[len] = q[--i]; rt[len] = t[--j]; }
        else if (d == 1) { rq[len] = q[--i]; rt[len] = '-'; }
        else if (d == 2) { rq[len] = '-'; rt[len] = t[--j]; }
        else break;                                   /* cannot happen: (0,0) is inside the band */
        len++;
    }
    for (size_t k = 0; k < len; k++) { qaln[k] = rq[len - 1 - k]; taln[k] = rt[len - 1 - k]; }
    qaln[len] = taln[len] = 0;
    free(prev); free(cur); free(dir); free(rq); free(rt);
    return len;
}

/* The band tried before either of those: OG_AL_WA cells to either side of a centre that FOLLOWS the alignment -- row i's
 * band starts s_i = clamp(a + 1 - OG_AL_WA, 0, 2) columns to the right of row i - 1's, a = the (first) cell of row i - 1
 * with the smallest score: the band's centre sits on the diagonal successor of the best prefix alignment so far.  Returns
 * the alignment, or *fallback = 1 when the band lost the corner (n, m), a row had no reachable cell, or the path came
 * within OG_AL_MARGIN cells of an edge: the static bands decide then. */
#define OG_AL_WA 56
static size_t og_adaptive_align(const char *q, uint32_t n, const char *t, uint32_t m, char *qaln, char *taln, int *fallback) {
    const int32_t W = OG_AL_WA, B = 2 * W + 1;
    *fallback = 1;
    int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * B), *cur = (int32_t *)malloc(sizeof(int32_t) * B);
    uint8_t *dir = (uint8_t *)malloc((size_t)(n + 1) * B), *sh = (uint8_t *)malloc((size_t)n + 1);
    int64_t lo = -(int64_t)W;
    size_t len = 0;
    int ok = 1;
    for (uint32_t i = 0; i <= n && ok; i++) {
        int32_t s = 0;
        if (i > 0) {
            int32_t a = -1, best = OG_AL_INF;
            for (int32_t k = 0; k < B; k++) if (prev[k] < best) { best = prev[k]; a = k; }
            if (a < 0) { ok = 0; break; }
            s = a + 1 - W; if (s < 0) s = 0; if (s > 2) s = 2;
            lo += s;
        }
        sh[i] = (uint8_t)s;
        for (int32_t k = 0; k < B; k++) {
            const int64_t j = lo + k;
            int32_t best = OG_AL_INF; uint8_t d = 3;
            if (j >= 0 && j <= (int64_t)m) {
                if (i == 0 && j == 0) { best = 0; d = 3; }
                if (i > 0 && j > 0) {
                    const int32_t kp = k + s - 1;
                    if (kp >= 0 && kp < B && prev[kp] < OG_AL_INF) {
                        const int32_t v = prev[kp] + (q[i - 1] == t[j - 1] ? OG_AL_MATCH : OG_AL_MISMATCH);
                        if (v < best) { best = v; d = 0; }
                    }
                }
                if (i > 0) {
                    const int32_t kp = k + s;
                    if (kp >= 0 && kp < B && prev[kp] < OG_AL_INF) {
                        const int32_t v = prev[kp] + OG_AL_INS;
                        if (v < best) { best = v; d = 1; }
                    }
                }
                if (j > 0 && k > 0 && cur[k - 1] < OG_AL_INF) {
                    const int32_t v = cur[k - 1] + OG_AL_DEL;
                    if (v < best) { best = v; d = 2; }
                }
            }
            cur[k] = best;
            dir[(size_t)i * B + k] = d;
        }
        int32_t *x = prev; prev = cur; cur = x;
    }
    if (ok) {
        const int64_t kend = (int64_t)m - lo;
        if (kend < 0 || kend >= B || prev[kend] >= OG_AL_INF) ok = 0;
    }
    if (ok) {
        char *rq = (char *)malloc((size_t)n + m + 1), *rt = (char *)malloc((size_t)n + m + 1);
        uint32_t i = n, j = m;
        int touched = 0;
        while (i > 0 || j > 0) {
            const int64_t kk = (int64_t)j - lo;
            if (kk < 0 || kk >= B) { touched = 1; break; }
            if (kk < OG_AL_MARGIN || kk > (int64_t)B - 1 - OG_AL_MARGIN) touched = 1;
            const uint8_t d = dir[(size_t)i * B + (size_t)kk];
            if (d == 0) { rq[len] = q[i - 1]; rt[len] = t[j - 1]; lo -= sh[i]; i--; j--; }
            else if (d == 1) { rq[len] = q[i - 1]; rt[len] = '-'; lo -= sh[i]; i--; }
            else if (d == 2) { rq[len] = '-'; rt[len] = t[j - 1]; j--; }
            else { touched = 1; break; }
            len++;
        }
        if (!touched) {
            for (size_t k = 0; k < len; k++) { qaln[k] = rq[len - 1 - k]; taln[k] = rt[len - 1 - k]; }
            qaln[len] = taln[len] = 0;
            *fallback = 0;
        }
        free(rq); free(rt);
    }
    free(prev); free(cur); free(dir); free(sh);
    return *fallback ? 0 : len;
}

size_t og_banded_align(const char *q, uint32_t n, const char *t, uint32_t m, char *qaln, char *taln) {
    if (n == 0 || m == 0) {
        size_t k = 0;
        for (uint32_t i = 0; i < n; i++) { qaln[k] = q[i]; taln[k] = '-'; k++; }
        for (uint32_t j = 0; j < m; j++) { qaln[k] = '-'; taln[k] = t[j]; k++; }
        qaln[k] = taln[k] = 0;
        return k;
    }
    const uint32_t w1 = og_align_halfwidth_first(n, m), w2 = og_align_halfwidth(n, m);
    int touched = 0;
    if (w1 > OG_AL_WA && !getenv("OG_NO_ADAPTIVE")) {          /* (a static band that narrow already: nothing to gain) */
        int fb = 0;
        const size_t la = og_adaptive_align(q, n, t, m, qaln, taln, &fb);
        if (!fb) return la;
    }
    size_t len = og_banded_align_w(q, n, t, m, w1, qaln, taln, &touched);
    /* near an edge, or no path at all inside the narrow band: the full band decides */
    if (w1 < w2 && (touched || len == 0)) len = og_banded_align_w(q, n, t, m, w2, qaln, taln, &touched);
    return len;
}

/* SimpleAligner.cpp:51-62 */
void og_simple_aligner_finish(char *qaln, char *taln, size_t n, uint32_t tseq_len, uint32_t tlen,
                              char strand, uint32_t *start, uint32_t *end) {
    *start += 0;                                /* :52  refinedAln.GenomicTBegin(): the global alignment starts at 0 */
    *end = *start + tseq_len;                   /* :53  + GenomicTEnd() */
    if (strand == '-') {
        *start = tlen - *end;                   /* :56 */
        og_revcomp(qaln, n);                    /* :57-58 */
        og_revcomp(taln, n);
    }
    *start += 1;                                /* :62 */
}

/* ------------------------------------------------------------------------ */
/* AlnGraphBoost                                                            */
/* ------------------------------------------------------------------------ */

typedef struct og_node {
    char base;
    int coverage, weight;
    int backbone, deleted;
    ivec out, in;          /* edge ids, list order */
    int64_t bbmap;         /* std::map<VtxDesc,VtxDesc>: absent key reads as 0 */
} og_node;

typedef struct og_edge { int32_t src, dst; int count; int visited; int alive; } og_edge;

struct og_graph {
    og_node *nodes; size_t nn, ncap;
    og_edge *edges; size_t ne, ecap;
    int32_t enter, exit_;
    int error;
};

static int32_t g_add_vertex(og_graph *g) {
    if (g->nn == g->ncap) {
        g->ncap = g->ncap ? g->ncap * 2 : 64;
        g->nodes = (og_node *)realloc(g->nodes, g->ncap * sizeof(og_node));
    }
    og_node *n = &g->nodes[g->nn];
    memset(n, 0, sizeof(*n));
    n->base = 'N';                      /* AlnGraphBoost.hpp:30-36 */
    return (int32_t)g->nn++;
}

static int32_t g_add_edge(og_graph *g, int32_t u, int32_t v) {
    if (g->ne == g->ecap) {
        g->ecap = g->ecap ? g->ecap * 2 : 64;
        g->edges = (og_edge *)realloc(g->edges, g->ecap * sizeof(og_edge));
    }
    og_edge *e = &g->edges[g->ne];
    e->src = u; e->dst = v; e->count = 0; e->visited = 0; e->alive = 1; /* hpp:43-46 */
    int32_t id = (int32_t)g->ne++;
    iv_push(&g->nodes[u].out, id);
    iv_push(&g->nodes[v].in, id);
    return id;
}

/* boost::edge(u,v,g): first entry of out[u] whose target is v */
static int32_t g_find_edge(og_graph *g, int32_t u, int32_t v) {
    ivec *o = &g->nodes[u].out;
    for (size_t i = 0; i < o->n; i++)
        if (g->edges[o->v[i]].dst == v) return o->v[i];
    return -1;
}

/* boost::clear_vertex: stable erase from the neighbours' lists */
static void g_clear_vertex(og_graph *g, int32_t n) {
    og_node *nd = &g->nodes[n];
    for (size_t i = 0; i < nd->out.n; i++) {
        int32_t e = nd->out.v[i];
        iv_remove_all(&g->nodes[g->edges[e].dst].in, e);
        g->edges[e].alive = 0;
    }
    for (size_t i = 0; i < nd->in.n; i++) {
        int32_t e = nd->in.v[i];
        iv_remove_all(&g->nodes[g->edges[e].src].out, e);
        g->edges[e].alive = 0;
    }
    nd->out.n = 0; nd->in.n = 0;
}

static og_graph *graph_new(const char *backbone, size_t blen) {
    og_graph *g = (og_graph *)calloc(1, sizeof(og_graph));
    for (size_t i = 0; i < blen + 2; i++) g_add_vertex(g);
    for (size_t i = 0; i < blen + 1; i++) g_add_edge(g, (int32_t)i, (int32_t)(i + 1));
    g->enter = 0;
    g->nodes[0].base = '^'; g->nodes[0].backbone = 1;
    for (size_t i = 0; i < blen; i++) {
        og_node *n = &g->nodes[i + 1];
        n->backbone = 1; n->weight = 1;
        n->base = backbone ? backbone[i] : 'N';
        n->bbmap = (int64_t)(i + 1);
    }
    g->exit_ = (int32_t)(blen + 1);
    g->nodes[blen + 1].base = '$'; g->nodes[blen + 1].backbone = 1;
    return g;
}

/* AlnGraphBoost.cpp:16-39 */
og_graph *og_graph_new_seq(const char *backbone, size_t blen) { return graph_new(backbone, blen); }
/* AlnGraphBoost.cpp:41-62 */
og_graph *og_graph_new_len(size_t blen) { return graph_new(NULL, blen); }

void og_graph_free(og_graph *g) {
    if (!g) return;
    for (size_t i = 0; i < g->nn; i++) { iv_free(&g->nodes[i].out); iv_free(&g->nodes[i].in); }
    free(g->nodes); free(g->edges); free(g);
}

/* AlnGraphBoost.cpp:109-127 */
static void g_add_edge_counted(og_graph *g, int32_t u, int32_t v) {
    int exists = 0;
    ivec *in = &g->nodes[v].in;
    for (size_t i = 0; i < in->n; i++) {
        og_edge *e = &g->edges[in->v[i]];
        if (e->src == u) { e->count++; exists = 1; }
    }
    if (!exists) {
        int32_t e = g_add_edge(g, u, v);
        g->edges[e].count++;
    }
}

/* AlnGraphBoost.cpp:64-107 */
void og_add_aln(og_graph *g, uint32_t start, const char *q, const char *t, size_t len) {
    uint32_t bbpos = start;
    int32_t prev = g->enter;
    for (size_t i = 0; i < len; i++) {
        char qb = q[i], tb = t[i];
        int32_t curr = (int32_t)bbpos;          /* index[bbPos] is the identity */
        if (qb == tb) {                         /* :75-85 match */
            og_node *bb = &g->nodes[g->nodes[curr].bbmap];
            bb->coverage++;
            bb->base = tb;
            g->nodes[curr].weight++;
            g_add_edge_counted(g, prev, curr);
            bbpos++;
            prev = curr;
        } else if (qb == '-' && tb != '-') {    /* :87-93 deletion */
            og_node *bb = &g->nodes[g->nodes[curr].bbmap];
            bb->coverage++;
            bb->base = tb;
            bbpos++;
        } else if (qb != '-' && tb == '-') {    /* :95-104 insertion */
            int32_t nv = g_add_vertex(g);
            g->nodes[nv].base = qb;
            g->nodes[nv].weight++;
            g->nodes[nv].bbmap = (int64_t)bbpos;
            g_add_edge_counted(g, prev, nv);
            prev = nv;
        }
    }
    g_add_edge_counted(g, prev, g->exit_);      /* :106 */
}

/* AlnGraphBoost.cpp:269-273 */
static void g_mark_for_reaper(og_graph *g, int32_t n) {
    g->nodes[n].deleted = 1;
    g_clear_vertex(g, n);
}

/* Collect the distinct bases of a candidate list in ascending char order
 * (std::map<char,...> iteration order, AlnGraphBoost.cpp:163,174,218,227). */
static size_t distinct_bases(const og_graph *g, const ivec *cand, char *bases) {
    size_t nb = 0;
    for (size_t i = 0; i < cand->n; i++) {
        char b = g->nodes[cand->v[i]].base;
        size_t k = 0;
        while (k < nb && bases[k] != b) k++;
        if (k == nb) bases[nb++] = b;
    }
    for (size_t i = 1; i < nb; i++) {           /* insertion sort, char compare */
        char b = bases[i]; size_t j = i;
        while (j > 0 && bases[j - 1] > b) { bases[j] = bases[j - 1]; j--; }
        bases[j] = b;
    }
    return nb;
}

/* AlnGraphBoost.cpp:162-215 */
static void g_merge_in_nodes(og_graph *g, int32_t n) {
    ivec cand = {0};
    for (size_t i = 0; i < g->nodes[n].in.n; i++) {     /* :166-171 */
        int32_t in_node = g->edges[g->nodes[n].in.v[i]].src;
        if (g->nodes[in_node].out.n == 1) iv_push(&cand, in_node);
    }
    char *bases = (char *)malloc(cand.n + 1);
    size_t nb = distinct_bases(g, &cand, bases);
    /* group membership is fixed here; bases are read now, as the map keys are */
    char *cbase = (char *)malloc(cand.n + 1);
    for (size_t i = 0; i < cand.n; i++) cbase[i] = g->nodes[cand.v[i]].base;

    for (size_t b = 0; b < nb; b++) {                   /* :174 */
        ivec nodes = {0};
        for (size_t i = 0; i < cand.n; i++) if (cbase[i] == bases[b]) iv_push(&nodes, cand.v[i]);
        if (nodes.n <= 1) { iv_free(&nodes); continue; }
        int32_t an = nodes.v[0];
        /* :183-190 accumulate out edge information */
        for (size_t k = 1; k < nodes.n; k++) {
            int32_t ni = nodes.v[k];
            if (g->nodes[an].out.n == 0 || g->nodes[ni].out.n == 0) { g->error = 1; continue; }
            g->edges[g->nodes[an].out.v[0]].count += g->edges[g->nodes[ni].out.v[0]].count;
            g->nodes[an].weight += g->nodes[ni].weight;
        }
        /* :193-212 accumulate in edge information, merge nodes */
        for (size_t k = 1; k < nodes.n; k++) {
            int32_t v = nodes.v[k];
            for (size_t i = 0; i < g->nodes[v].in.n; i++) {
                int32_t ie = g->nodes[v].in.v[i];
                int32_t n1 = g->edges[ie].src;
                int32_t e = g_find_edge(g, n1, an);
                if (e >= 0) {
                    g->edges[e].count += g->edges[ie].count;
                } else {
                    int32_t ne = g_add_edge(g, n1, an);
                    g->edges[ne].count = g->edges[ie].count;
                    g->edges[ne].visited = g->edges[ie].visited;
                }
            }
            g_mark_for_reaper(g, v);
        }
        g_merge_in_nodes(g, an);                        /* :213 */
        iv_free(&nodes);
    }
    free(bases); free(cbase); iv_free(&cand);
}

/* AlnGraphBoost.cpp:217-267 */
static void g_merge_out_nodes(og_graph *g, int32_t n) {
    ivec cand = {0};
    for (size_t i = 0; i < g->nodes[n].out.n; i++) {    /* :220-225 */
        int32_t out_node = g->edges[g->nodes[n].out.v[i]].dst;
        if (g->nodes[out_node].in.n == 1) iv_push(&cand, out_node);
    }
    char *bases = (char *)malloc(cand.n + 1);
    size_t nb = distinct_bases(g, &cand, bases);
    char *cbase = (char *)malloc(cand.n + 1);
    for (size_t i = 0; i < cand.n; i++) cbase[i] = g->nodes[cand.v[i]].base;

    for (size_t b = 0; b < nb; b++) {                   /* :227 */
        ivec nodes = {0};
        for (size_t i = 0; i < cand.n; i++) if (cbase[i] == bases[b]) iv_push(&nodes, cand.v[i]);
        if (nodes.n <= 1) { iv_free(&nodes); continue; }
        int32_t an = nodes.v[0];
        /* :236-243 accumulate inner edge information */
        for (size_t k = 1; k < nodes.n; k++) {
            int32_t ni = nodes.v[k];
            if (g->nodes[an].in.n == 0 || g->nodes[ni].in.n == 0) { g->error = 1; continue; }
            g->edges[g->nodes[an].in.v[0]].count += g->edges[g->nodes[ni].in.v[0]].count;
            g->nodes[an].weight += g->nodes[ni].weight;
        }
        /* :246-265 accumulate and merge outer edge information */
        for (size_t k = 1; k < nodes.n; k++) {
            int32_t v = nodes.v[k];
            for (size_t i = 0; i < g->nodes[v].out.n; i++) {
                int32_t oe = g->nodes[v].out.v[i];
                int32_t n2 = g->edges[oe].dst;
                int32_t e = g_find_edge(g, an, n2);
                if (e >= 0) {
                    g->edges[e].count += g->edges[oe].count;
                } else {
                    int32_t ne = g_add_edge(g, an, n2);
                    g->edges[ne].count = g->edges[oe].count;
                    g->edges[ne].visited = g->edges[oe].visited;
                }
            }
            g_mark_for_reaper(g, v);
        }
        iv_free(&nodes);
    }
    free(bases); free(cbase); iv_free(&cand);
}

/* AlnGraphBoost.cpp:129-160 */
int og_merge_nodes(og_graph *g) {
    size_t qcap = g->nn + 16, qh = 0, qt = 0;
    int32_t *queue = (int32_t *)malloc(qcap * sizeof(int32_t));
    queue[qt++] = g->enter;
    while (qh < qt) {
        int32_t u = queue[qh++];
        g_merge_in_nodes(g, u);
        g_merge_out_nodes(g, u);
        for (size_t i = 0; i < g->nodes[u].out.n; i++) {
            og_edge *e = &g->edges[g->nodes[u].out.v[i]];
            e->visited = 1;
            int32_t v = e->dst;
            int not_visited = 0;
            for (size_t k = 0; k < g->nodes[v].in.n; k++)
                if (!g->edges[g->nodes[v].in.v[k]].visited) not_visited++;
            if (not_visited == 0) {
                if (qt == qcap) { qcap *= 2; queue = (int32_t *)realloc(queue, qcap * sizeof(int32_t)); }
                queue[qt++] = v;
            }
        }
    }
    free(queue);
    return g->error ? -1 : 0;
}

/* AlnGraphBoost.cpp:375-459 */
size_t og_best_path(og_graph *g, int32_t **path_out) {
    for (size_t e = 0; e < g->ne; e++) g->edges[e].visited = 0;       /* :376-378 */
    int32_t *best_edge = (int32_t *)malloc(g->nn * sizeof(int32_t));
    float *score = (float *)calloc(g->nn, sizeof(float));             /* map default 0.0f */
    for (size_t i = 0; i < g->nn; i++) best_edge[i] = -1;
    size_t qcap = g->nn + 16, qh = 0, qt = 0;
    int32_t *queue = (int32_t *)malloc(qcap * sizeof(int32_t));
    queue[qt++] = g->exit_;
    score[g->exit_] = 0.0f;
    while (qh < qt) {
        int32_t n = queue[qh++];
        int found = 0;
        float best = -FLT_MAX;
        int32_t best_e = -1;
        for (size_t i = 0; i < g->nodes[n].out.n; i++) {              /* :399-416 */
            int32_t oe = g->nodes[n].out.v[i];
            int32_t t = g->edges[oe].dst;
            const og_node *tn = &g->nodes[t];
            float s = score[t], ns;
            if (tn->backbone && tn->weight == 1) {
                ns = s - 10.0f;
            } else {
                const og_node *bb = &g->nodes[tn->bbmap];
                ns = (float)g->edges[oe].count - (float)bb->coverage * 0.5f + s;
            }
            if (ns > best) { best = ns; best_e = oe; found = 1; }
        }
        if (found) { score[n] = best; best_edge[n] = best_e; }
        for (size_t i = 0; i < g->nodes[n].in.n; i++) {               /* :423-439 */
            og_edge *ie = &g->edges[g->nodes[n].in.v[i]];
            ie->visited = 1;
            int32_t s = ie->src;
            int not_visited = 0;
            for (size_t k = 0; k < g->nodes[s].out.n; k++)
                if (!g->edges[g->nodes[s].out.v[k]].visited) not_visited++;
            if (not_visited == 0) {
                if (qt == qcap) { qcap *= 2; queue = (int32_t *)realloc(queue, qcap * sizeof(int32_t)); }
                queue[qt++] = s;
            }
        }
    }
    /* :443-456 */
    size_t pcap = 64, pn = 0;
    int32_t *path = (int32_t *)malloc(pcap * sizeof(int32_t));
    int32_t prev = g->enter;
    for (;;) {
        if (pn == pcap) { pcap *= 2; path = (int32_t *)realloc(path, pcap * sizeof(int32_t)); }
        path[pn++] = prev;
        if (best_edge[prev] < 0) break;
        prev = g->edges[best_edge[prev]].dst;
        if (pn > g->nn + 1) { g->error = 1; break; }   /* cycle guard: non-conforming input */
    }
    free(queue); free(score); free(best_edge);
    *path_out = path;
    return pn;
}

/* AlnGraphBoost.cpp:285-325 */
char *og_consensus_longest(og_graph *g, int min_weight) {
    int32_t *path; size_t pn = og_best_path(g, &path);
    char *cns = (char *)malloc(pn + 1);
    int offs = 0, best_offs = 0, length = 0, idx = 0, met = 0;
    char eb = g->nodes[g->enter].base, xb = g->nodes[g->exit_].base;
    for (size_t i = 0; i < pn; i++) {
        const og_node *n = &g->nodes[path[i]];
        if (n->base == eb || n->base == xb) continue;
        cns[idx] = n->base;
        if (!met && n->weight >= min_weight) { offs = idx; met = 1; }
        else if (met && n->weight < min_weight) {
            if ((idx - offs) > length) { best_offs = offs; length = idx - offs; }
            met = 0;
        }
        idx++;
    }
    if (met && (idx - offs) > length) { best_offs = offs; length = idx - offs; }
    char *r = dup_range(cns + best_offs, (size_t)length);
    free(cns); free(path);
    return r;
}

/* AlnGraphBoost.cpp:327-373 */
size_t og_consensus_all(og_graph *g, int min_weight, size_t min_len, og_segment **segs_out) {
    int32_t *path; size_t pn = og_best_path(g, &path);
    char *cns = (char *)malloc(pn + 1);
    size_t ns = 0, scap = 4;
    og_segment *segs = (og_segment *)malloc(scap * sizeof(og_segment));
    int offs = 0, idx = 0, met = 0;
    char eb = g->nodes[g->enter].base, xb = g->nodes[g->exit_].base;
    for (size_t i = 0; i <= pn; i++) {
        int close = 0;
        if (i < pn) {
            const og_node *n = &g->nodes[path[i]];
            if (n->base == eb || n->base == xb) continue;
            cns[idx] = n->base;
            if (!met && n->weight >= min_weight) { offs = idx; met = 1; }
            else if (met && n->weight < min_weight) { met = 0; close = 1; }
        } else if (met) {
            close = 1;                                   /* :363-371 end of sequence */
        }
        if (close) {
            size_t length = (size_t)(idx - offs);
            if (length >= min_len) {
                if (ns == scap) { scap *= 2; segs = (og_segment *)realloc(segs, scap * sizeof(og_segment)); }
                segs[ns].range0 = offs; segs[ns].range1 = idx;
                segs[ns].seq = dup_range(cns + offs, length);
                ns++;
            }
        }
        if (i < pn) idx++;
    }
    free(cns); free(path);
    *segs_out = segs;
    return ns;
}

void og_free_segments(og_segment *segs, size_t n) {
    if (!segs) return;
    for (size_t i = 0; i < n; i++) free(segs[i].seq);
    free(segs);
}

/* AlnGraphBoost.cpp:468-487 (the reference's variable names are swapped; the
 * test is symmetric) */
int og_dangling_nodes(og_graph *g) {
    int found = 0;
    char eb = g->nodes[g->enter].base, xb = g->nodes[g->exit_].base;
    for (size_t v = 0; v < g->nn; v++) {
        const og_node *n = &g->nodes[v];
        if (n->deleted) continue;
        if (n->base == eb || n->base == xb) continue;
        if (n->out.n > 0 && n->in.n > 0) continue;
        found = 1;
    }
    return found;
}

/* ------------------------------------------------------------------------ */
/* introspection                                                            */
/* ------------------------------------------------------------------------ */
size_t og_num_nodes(const og_graph *g) { return g->nn; }
size_t og_num_live_nodes(const og_graph *g) {
    size_t n = 0;
    for (size_t i = 0; i < g->nn; i++) n += !g->nodes[i].deleted;
    return n;
}
size_t og_num_live_edges(const og_graph *g) {
    size_t n = 0;
    for (size_t i = 0; i < g->ne; i++) n += g->edges[i].alive;
    return n;
}
void og_node_info(const og_graph *g, size_t v, char *base, int *weight, int *coverage,
                  int *deleted, int *backbone, int64_t *bbmap) {
    const og_node *n = &g->nodes[v];
    *base = n->base; *weight = n->weight; *coverage = n->coverage;
    *deleted = n->deleted; *backbone = n->backbone; *bbmap = n->bbmap;
}
size_t og_out_edges(const og_graph *g, size_t v, int32_t *dst, int32_t *count, size_t cap) {
    const ivec *o = &g->nodes[v].out;
    for (size_t i = 0; i < o->n && i < cap; i++) {
        dst[i] = g->edges[o->v[i]].dst; count[i] = g->edges[o->v[i]].count;
    }
    return o->n;
}
size_t og_in_edges(const og_graph *g, size_t v, int32_t *src, int32_t *count, size_t cap) {
    const ivec *o = &g->nodes[v].in;
    for (size_t i = 0; i < o->n && i < cap; i++) {
        src[i] = g->edges[o->v[i]].src; count[i] = g->edges[o->v[i]].count;
    }
    return o->n;
}

/* ------------------------------------------------------------------------ */
/* main.cpp:117-148 for one target                                          */
/* ------------------------------------------------------------------------ */

/* An alignment is conforming when addAln stays inside the backbone: start>=1
 * and every target base lands on a backbone vertex 1..tlen. */
static int aln_conforms(uint32_t tlen, uint32_t start, const char *t, size_t len) {
    if (len == 0) return 1;            /* only adds enter->exit; start unused */
    if (start < 1) return 0;
    uint64_t tb = 0;
    for (size_t i = 0; i < len; i++) tb += (t[i] != '-');
    return (uint64_t)start - 1 + tb <= (uint64_t)tlen;
}

static long consensus_target_impl(uint32_t tlen, const char *backbone, size_t n_alns,
                                  const uint32_t *starts, const char *const *qstrs,
                                  const char *const *tstrs, const uint64_t *offs,
                                  const uint32_t *lens32, const size_t *lens,
                                  const char *qblob, const char *tblob,
                                  const og_opts *opts, og_segment **segs, long *bad_aln) {
    og_graph *g = backbone ? og_graph_new_seq(backbone, tlen) : og_graph_new_len(tlen);
    size_t cap = 0;
    char *qn = NULL, *tn = NULL;
    for (size_t a = 0; a < n_alns; a++) {
        const char *q = qstrs ? qstrs[a] : qblob + offs[a];
        const char *t = tstrs ? tstrs[a] : tblob + offs[a];
        size_t len = lens ? lens[a] : (size_t)lens32[a];
        if (len < opts->min_len) continue;                       /* main.cpp:132 */
        if (2 * len + 1 > cap) {
            cap = 2 * len + 1;
            qn = (char *)realloc(qn, cap); tn = (char *)realloc(tn, cap);
        }
        size_t nl = og_normalize_gaps(q, t, len, 1, qn, tn);     /* main.cpp:133 */
        size_t lo, ro; uint32_t lb;
        og_trim_aln(tn, nl, (int)opts->trim, &lo, &ro, &lb);     /* main.cpp:134 */
        uint32_t start = starts[a] + lb;
        if (!aln_conforms(tlen, start, tn + lo, ro - lo)) {
            if (bad_aln) *bad_aln = (long)a;
            free(qn); free(tn); og_graph_free(g);
            return -2;
        }
        og_add_aln(g, start, qn + lo, tn + lo, ro - lo);         /* main.cpp:135 */
    }
    free(qn); free(tn);
    int rc = og_merge_nodes(g);                                  /* main.cpp:137 */
    long ns = -1;
    if (rc == 0) {
        ns = (long)og_consensus_all(g, opts->min_weight, opts->min_len, segs); /* :138 */
        if (g->error) { og_free_segments(*segs, (size_t)ns); *segs = NULL; ns = -1; }
    }
    og_graph_free(g);
    return ns;
}

long og_consensus_target(uint32_t tlen, const char *backbone, size_t n_alns,
                         const uint32_t *starts, const char *const *qstrs,
                         const char *const *tstrs, const size_t *lens,
                         const og_opts *opts, og_segment **segs, long *bad_aln) {
    return consensus_target_impl(tlen, backbone, n_alns, starts, qstrs, tstrs, NULL, NULL,
                                 lens, NULL, NULL, opts, segs, bad_aln);
}

long og_consensus_target_blob(uint32_t tlen, const char *backbone, size_t n_alns,
                              const uint32_t *starts, const uint64_t *offs,
                              const uint32_t *lens, const char *qblob,
                              const char *tblob, const og_opts *opts,
                              og_segment **segs, long *bad_aln) {
    return consensus_target_impl(tlen, backbone, n_alns, starts, NULL, NULL, offs, lens,
                                 NULL, qblob, tblob, opts, segs, bad_aln);
}
