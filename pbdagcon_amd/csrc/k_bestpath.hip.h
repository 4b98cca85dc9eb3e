// k_bestpath.hip.h -- stage (c): bestPath (AlnGraphBoost.cpp:375-459) and the
// consensus segmentation (AlnGraphBoost.cpp:327-373).
//
// The DP is order-independent: score[n] = max over out_edges(n), IN LIST ORDER with
// strict '>' (first maximum wins, :411), of
//     score[t] - 10                                   if t.backbone && t.weight == 1
//     count(e) - coverage[bbMap[t]]*0.5 + score[t]    otherwise          (:404-409)
// so any order that scores a vertex after its successors gives the reference's
// result.  Both cases are  w(e) + score[t]  with a per-edge term
//     w(e) = -10                                (x - 10 and -10 + x are the same fp32 value)
//     w(e) = (float)count - coverage*0.5f       (the reference's left-to-right order)
// that does not depend on any score.
//
//   k_bp_terms    thread per vertex: the target-side term of w(e), one float per vertex.
//                 (w(e) itself is formed from it and the edge's count when k_bp_sweep unpacks a
//                 chunk: the terms are gathered a chunk ahead, like the edges.)
//   k_bp_sweep    one wave per (target, segment between two cut vertices of k_cuts); what
//                 is left is the bare recurrence.  Vertex ids
//                 are in backbone-position order, a topological order except for the few
//                 edges the merge turned around.  The wave STREAMS the vertices from the
//                 exit downwards: the records of 64 vertices at a time are requested two
//                 chunks ahead (plain loads whose latency nobody waits for) and unpacked
//                 into an LDS ring (edges: target, w); finished scores live in a second
//                 LDS ring.  A step reads the vertex's edges and its successors' scores
//                 from LDS, one edge per lane, takes the first maximum and stores it.
//                 An edge to a vertex that has no score yet (a turned-around edge) puts
//                 that vertex on a small stack and it is scored first.  Whatever is not
//                 in the rings (long lists, far-away successors) comes from HBM.
//   k_bp_check    one wave per target: checks that the segmented sweep was exact, else
//                 sweeps the target in one piece.
//   k_bp_walk     one wave per (target, segment): the best-edge walk (:443-456) from one
//                 cut vertex to the next on LDS-staged (best, base, weight) triples.
//   k_bp_join     one wave per target: the segmentation (:327-373) over the joined
//                 pieces, and the output.
//
// fp32 throughout; every value is a multiple of 0.5 below 2^23, so the arithmetic is
// exact (-ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "dagcon_dev.h"

// what an edge needs to know about its target t: -10 applies (backbone && weight == 1, :404-405),
// else coverage[bbMap[t]] * 0.5f (:407-408).  One float per vertex, so that the per-edge pass
// gathers 4 bytes from a compact array instead of the target's 32-byte record and its coverage.
#define DG_TT_TEN 1.0e30f
__global__ __launch_bounds__(256) void k_bp_terms(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint64_t nb = p.node_base[t];
    const uint32_t N = p.n_nodes[t];
    const int32_t *cov = p.cov + p.bbv_base[t];
    const DgNode *nd = p.nodes + nb;
    for (uint32_t v = blockIdx.y * 256 + threadIdx.x; v < N; v += gridDim.y * 256) {
        const uint4 h = *reinterpret_cast<const uint4 *>(&nd[v]);
        const uint4 h2 = *(reinterpret_cast<const uint4 *>(&nd[v]) + 1);
        float tt;
        if (((h.y >> 8) & DG_NF_BACKBONE) && (int)h.z == 1) tt = DG_TT_TEN;
        else tt = (float)cov[(int)h2.w] * 0.5f;
        p.bp_tt[nb + v] = tt;
        p.score[nb + v] = make_float2(0.0f, 0.0f);     // (score, 1 = final); absent key reads as 0
        p.best[nb + v] = -1;
    }
}

// LDS per sweep wave is what bounds the waves in flight (8 segments x 1000 targets want
// ~8 KB each): only the chunk being swept needs its edges staged, and successors are
// almost always within a few hundred ids
#ifndef DG_SR
#define DG_SR 256            // finished scores (id & 255)
#define DG_BOUT 6            // out edges kept in a staged slot
#endif
#define DG_BR 64             // staged vertices (id & 63): the chunk being swept
#define DG_WR 256            // the walk's staging ring
#define DG_BSTK 64           // evaluation stack (LDS part)
#define DG_BDEF 32           // vertices that wait for a far successor (def_v: who, def_b: for whom)
#define DG_BFAR 192          // a successor this many ids below the stream is "far"
#define DG_BL_HBM  0x40000000  // lens: the edges did not fit the slot
#define DG_BL_DONE 0x20000000  // lens: scored already

struct DgBpShared {
    int stag[DG_SR];
    float sval[DG_SR];
    int tag[DG_BR];
    int lens[DG_BR];                     // out_len | flags << 16 | DG_BL_*
    int out_dst[DG_BR * DG_BOUT];
    float out_w[DG_BR * DG_BOUT];
    int stk[DG_BSTK];
    int def_v[DG_BDEF], def_b[DG_BDEF];
    int rbest[DG_BR];
    float rscore[DG_BR];                 // results of the chunk being swept, flushed at its end
};
struct DgBpSharedB {                     // the AB sweep's second value of a vertex (B next to A): k_bp_sweep_ab only, so that the
    float svalb[DG_SR], rscoreb[DG_BR];  // other sweeps keep their LDS footprint (it bounds their waves in flight)
};
struct DgWalkShared {
    unsigned char wbuf[64];              // consensus bases of the walk, flushed 64 at a time
    int wtag[DG_WR], wbest[DG_WR], wbase[DG_WR], wweight[DG_WR];   // the walk's staging ring
};

// Scores the vertices v_top .. v_bot (descending ids) of one target.  c_top >= 0: the
// vertex every path of this stretch ends in (the next cut, see k_cuts); it counts as
// score 0 and is not evaluated, so the scores are relative to it.  amax = largest
// |score| seen.
// Partial-span pileups (p.gcuts): a segment between two cuts of k_cuts2 may hold vertices with an edge
// to the exit vertex (reads end there), so a path can leave it without passing its upper cut.  The
// recurrence is linear in (max, +): with A[x] = best x -> upper cut and B[x] = best x -> exit past it,
// score[x] = max(A[x] + score[cut], B[x]).  The same sweep gives A with (cut, exit) worth (0, -inf), B with
// (-inf, 0) and the absolute scores with (score[cut], 0): ctv / xv are those two values, xid the exit
// vertex (-1: it is a vertex of the stretch itself).  skip_def: the stream passes over the vertices flagged
// DG_NF_DEFER -- enter, and the few vertices the merge's prologue visited whose successors lie in another
// segment than their own id (chains reads begin with, united at enter), with their ancestors.  One of
// them whose successors all lie in ONE segment is scored when a vertex of that segment asks for it (the
// evaluation stack); what nobody asks for is scored by k_bp_defer, after the sweeps.
// The other end: mergeInNodes(exit) unites the vertices reads END with (one out-edge, to exit, equal bases),
// wherever on the backbone they lie, and its recursion their predecessors: edges from anywhere into that
// little tree.  Its scores depend on nothing but exit, so k_bp_xtree has them first (final flag 2.0) and an
// edge into it is an escape like an edge to exit itself: -inf for A, its absolute score for B and the rest.
#define DG_BP_PIECES 256          // most pieces of a target's bestPath sweep (full-span path; 64 with p.gcuts)
#define DG_BP_NINF (-1.0e9f)
#define DG_BP_ONE 0xFFFFFFFFu      // DgParams::defer[0]: the target is swept in one piece; bp_end of the first piece: it ended at exit
// AB: one sweep carries both values of the (max, +) decomposition, A[x] = best x -> upper cut (cut worth 0, exit
// -inf) in score[].x and B[x] = best x -> exit past it (cut -inf, exit and the exit tree worth their absolute scores)
// in score_b[]; no choices are recorded (best[] is k_bp_choose's, from the absolute scores k_bp_abs makes of them).
template <bool AB = false>
__device__ __forceinline__ void dg_bp_sweep(DgBpShared &S, const DgNode *nd, int32_t *best, float2 *score,
                                            const uint32_t *pool, const float *tt, const int v_top, const int v_bot,
                                            const int c_top, int32_t *gstk, const int gstk_cap,
                                            const int lane, float &amax, bool &bad, bool &stuck,
                                            const float ctv = 0.0f, const int xid = -1, const float xv = 0.0f,
                                            const bool skip_def = false, float *score_b = nullptr, DgBpSharedB *SB = nullptr) {
    const int dead_mask = (int)DG_NF_DELETED | (skip_def ? (int)DG_NF_DEFER : 0);
    for (int i = lane; i < DG_SR; i += 64) S.stag[i] = -1;
    for (int i = lane; i < DG_BR; i += 64) S.tag[i] = -1;
    if (c_top >= 0 && lane == 0) {
        S.stag[c_top & (DG_SR - 1)] = c_top; S.sval[c_top & (DG_SR - 1)] = ctv;
        if constexpr (AB) SB->svalb[c_top & (DG_SR - 1)] = DG_BP_NINF;
    }

    // ---- staging registers: r_* = records of a chunk, e_* = its edges ----
    uint4 r_lo, r_hi, n_lo, n_hi;
    int e_dst[DG_BOUT], e_w[DG_BOUT];          // e_w: the edge's count; its score term is made of it and
    float e_t[DG_BOUT];                        // e_t, the target-side term (k_bp_terms), when the chunk is unpacked
    // chunk c holds ids [v_top-64c-63, v_top-64c]; lane l -> id v_top-64c-l
#define DG_LOAD_REC(C, LO, HI)                                                             \
    do {                                                                                    \
        const int v_ = v_top - 64 * (C) - lane;                                             \
        LO = make_uint4(0, 0, 0, 0); HI = make_uint4(0, 0, 0, 0);                           \
        if (v_ >= v_bot) {                                                                      \
            LO = *reinterpret_cast<const uint4 *>(&nd[v_]);                                 \
            HI = *(reinterpret_cast<const uint4 *>(&nd[v_]) + 1);                           \
        }                                                                                   \
    } while (0)
#define DG_LOAD_ENT(LO, HI)                                                                 \
    do {                                                                                    \
        const int ol_ = (int)((LO).x & 0xffffu);                                            \
        _Pragma("unroll") for (int k_ = 0; k_ < DG_BOUT; k_++) {                            \
            e_dst[k_] = 0; e_w[k_] = 0;                                                     \
            if (ol_ <= DG_BOUT && k_ < ol_) { e_dst[k_] = (int)pool[(HI).x + 2 * k_]; e_w[k_] = (int)pool[(HI).x + 2 * k_ + 1]; } \
        }                                                                                   \
    } while (0)
#define DG_LOAD_TT()                                                                        \
    do {                                                                                    \
        _Pragma("unroll") for (int k_ = 0; k_ < DG_BOUT; k_++) e_t[k_] = tt[e_dst[k_]];     \
    } while (0)
#define DG_WRITE_CHUNK(C, LO)                                                               \
    do {                                                                                    \
        const int v_ = v_top - 64 * (C) - lane;                                             \
        if (v_ >= v_bot) {                                                                      \
            const int x_ = v_ & (DG_BR - 1);                                                \
            const int ol_ = (int)((LO).x & 0xffffu);                                        \
            int ln_ = (ol_ <= DG_BOUT ? ol_ : DG_BL_HBM) | (int)(((LO).y >> 8) & 0xffu) << 16; \
            S.tag[x_] = v_;                                                                 \
            S.lens[x_] = ln_;                                                               \
            if (ol_ <= DG_BOUT) {                                                           \
                _Pragma("unroll") for (int k_ = 0; k_ < DG_BOUT; k_++) {                    \
                    S.out_dst[x_ * DG_BOUT + k_] = e_dst[k_];                               \
                    S.out_w[x_ * DG_BOUT + k_] = e_t[k_] == DG_TT_TEN ? -10.0f : (float)e_w[k_] - e_t[k_]; /* :404-408 */ \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
    } while (0)

    const int n_chunks = (v_top - v_bot + 1 + 63) / 64;
    // prologue: chunk 0 into the ring; chunk 1 records + edges, chunk 2 records in flight
    DG_LOAD_REC(0, r_lo, r_hi);
    DG_LOAD_ENT(r_lo, r_hi);
    DG_LOAD_TT();
    DG_WRITE_CHUNK(0, r_lo);
    DG_LOAD_REC(1, r_lo, r_hi);
    DG_LOAD_ENT(r_lo, r_hi);
    DG_LOAD_REC(2, n_lo, n_hi);
    DG_LOAD_TT();                                      // (chunk 1's; from here on at the end of every chunk)

    unsigned long long guard = 0;
    const unsigned long long guard_max = 64ull * (unsigned long long)(v_top - v_bot + 1) + 1000000ull;

    // A successor FAR below the stream (partial-span reads: the vertices merged at `enter` reach all
    // over the backbone) is not chased: that would score everything in between on the slow path
    // below.  The vertex WAITS for it instead (so does every vertex on the evaluation stack under
    // it), the stream goes on, and when the successor has its score the waiting vertex is looked at
    // again.  Few vertices ever wait (DG_BDEF; the stack takes over when the list is full).
    int ndef = 0;
    // x has its score now: the vertices that wait for it go onto the evaluation stack (entries
    // q0, q0+1, ...; entry q lives in S.stk[q] or, beyond DG_BSTK, in the HBM scratch) and leave
    // the waiting list.  Returns how many.
    auto collect = [&](const int x, const int q0) -> int {
        const bool in = lane < ndef;
        const int wv = in ? S.def_v[lane] : 0, wb = in ? S.def_b[lane] : -1;
        const unsigned long long hit = __ballot(in && wb == x);
        if (!hit) return 0;
        const unsigned long long below = (1ull << lane) - 1ull;
        if ((hit >> lane) & 1ull) {
            const int q = q0 + __popcll(hit & below);
            if (q < DG_BSTK) S.stk[q] = wv; else gstk[q - DG_BSTK] = wv;
        } else if (in) {
            const unsigned long long stay = __ballot(in) & ~hit;       // (same value in every lane)
            const int k = __popcll(stay & below);
            S.def_v[k] = wv; S.def_b[k] = wb;
        }
        ndef -= __popcll(hit);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        return __popcll(hit);
    };
    bool tt_due = false;
    for (int c = 0; c < n_chunks && !bad; c++) {
        if (c > 0) {
            // chunk c: its records and edges were requested a whole chunk ago
            DG_WRITE_CHUNK(c, r_lo);
            r_lo = n_lo; r_hi = n_hi;
            DG_LOAD_ENT(r_lo, r_hi);                       // chunk c+1
            DG_LOAD_REC(c + 2, n_lo, n_hi);                // chunk c+2
            tt_due = true;
        }
        const int v_hi = v_top - 64 * c;
        const int v_lo = v_hi - 63 < v_bot ? v_bot : v_hi - 63;

        // (lens, edges) of the next vertex are read one step ahead: their LDS latency overlaps
        // the current step
        const int el = lane < DG_BOUT ? lane : 0;
        // the vertices of the chunk that are still in the graph (lane l <-> vertex v_hi - l): the
        // ~40 % the merge deleted are never looked at
        unsigned long long live;
        {
            const int vl = v_hi - lane;
            const int lnl = vl >= v_lo ? S.lens[vl & (DG_BR - 1)] : (int)DG_BL_DONE;
            live = __ballot(!(lnl & DG_BL_DONE) && !((lnl >> 16) & dead_mask));
        }
        int pf_ln = 0, pf_d = 0;
        float pf_w = 0.0f;
        if (live) {
            const int x0 = (v_hi - (__ffsll((long long)live) - 1)) & (DG_BR - 1);
            pf_ln = S.lens[x0]; pf_d = S.out_dst[x0 * DG_BOUT + el]; pf_w = S.out_w[x0 * DG_BOUT + el];
        }
        while (live && !bad) {
            const int v = v_hi - (__ffsll((long long)live) - 1);
            live &= live - 1ull;
            // ---- straight-line step for the common case: the vertex of the stream is staged,
            // its edges fit the slot and every successor already has its score in the ring.
            // Two LDS trips: (lens, edges) then (successor scores). ----
            int woken = 0;
            {
                const int xs = v & (DG_BR - 1);
                const int ln = __builtin_amdgcn_readfirstlane(pf_ln);          // chunk c is resident: tag == v
                const int d = pf_d;
                const float w = pf_w;
                if (live) {
                    const int xn = (v_hi - (__ffsll((long long)live) - 1)) & (DG_BR - 1);
                    pf_ln = S.lens[xn]; pf_d = S.out_dst[xn * DG_BOUT + el]; pf_w = S.out_w[xn * DG_BOUT + el];
                }
                if (ln & DG_BL_DONE) continue;                                  // scored early, as somebody's successor
                if (!(ln & DG_BL_HBM)) {
                    const int ol = ln & 0xffff;
                    const int y = d & (DG_SR - 1);
                    const int stg = S.stag[y];
                    const float ns = w + S.sval[y];
                    float nsb = 0.0f;
                    if constexpr (AB) nsb = w + SB->svalb[y];
                    const bool ok = lane >= ol || stg == d;
                    if (__all(ok)) {
                        // :399-416 first maximum in list order, strict '>'
                        float mx = 0.0f, mxb = 0.0f;
                        int bd = -1;
                        if (ol > 0) {
                            mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ns), 0));
                            bd = __builtin_amdgcn_readlane(d, 0);
                            if constexpr (AB) mxb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nsb), 0));
                            for (int i = 1; i < ol; i++) {
                                const float xsx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ns), i));
                                if (xsx > mx) { mx = xsx; bd = __builtin_amdgcn_readlane(d, i); }
                                if constexpr (AB) mxb = fmaxf(mxb, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nsb), i)));
                            }
                        }
                        if (lane == 0) {
                            S.stag[v & (DG_SR - 1)] = v;
                            S.sval[v & (DG_SR - 1)] = mx;
                            S.rscore[xs] = mx; S.rbest[xs] = bd;
                            S.lens[xs] = ln | DG_BL_DONE;
                            if constexpr (AB) { SB->svalb[v & (DG_SR - 1)] = mxb; SB->rscoreb[xs] = mxb; }
                        }
                        if (mx > 0.5f * DG_BP_NINF) amax = fmaxf(amax, fabsf(mx));
                        if constexpr (AB) if (mxb > 0.5f * DG_BP_NINF) amax = fmaxf(amax, fabsf(mxb));
                        if (ndef) woken = collect(v, 0);          // (rare: somebody far above waits for v)
                        if (!woken) continue;
                    }
                }
            }
            // ---- general step: the vertex of the stream is the bottom of the evaluation stack
            // and stays in a register; entries above it (turned-around edges, vertices that
            // waited) live in LDS, then HBM ----
            int sp = 1 + woken;
            while (sp > 0) {
                // every wave must leave this loop: the work is bounded by (vertices + edges), far
                // below this budget; running out of it means a broken graph, not a long one
                if (++guard > guard_max) { bad = true; stuck = true; break; }
                int n;
                if (sp == 1) n = v;
                else if (sp - 2 < DG_BSTK) n = __builtin_amdgcn_readfirstlane(S.stk[sp - 2]);
                else n = __builtin_amdgcn_readfirstlane(gstk[sp - 2 - DG_BSTK]);
                const int x = n & (DG_BR - 1);
                const int tg_ = S.tag[x], ln_ = S.lens[x];
                const bool in_ring = __builtin_amdgcn_readfirstlane(tg_) == n;
                int lens;
                if (in_ring) lens = __builtin_amdgcn_readfirstlane(ln_);
                else {
                    // not staged (a far successor reached through a turned-around edge)
                    const uint4 lo = *reinterpret_cast<const uint4 *>(&nd[n]);
                    const float fin = score[n].y;
                    lens = __builtin_amdgcn_readfirstlane((int)DG_BL_HBM | (int)((lo.y >> 8) & 0xffu) << 16 |
                                                          (fin >= 1.0f ? (int)DG_BL_DONE : 0));
                }
                if ((lens & DG_BL_DONE) || ((lens >> 16) & DG_NF_DELETED)) { sp--; continue; }   // (a deferred vertex is scored when somebody asks for it)
                int out_len = lens & 0xffff;
                const bool hbm = (lens & DG_BL_HBM) != 0;
                uint32_t out_off = 0;
                if (hbm) {
                    const DgNode nn = nd[n];
                    out_off = __builtin_amdgcn_readfirstlane(nn.out_off);
                    out_len = __builtin_amdgcn_readfirstlane((int)nn.out_len);
                }
                float mx = 0.0f, mxb = 0.0f;   // no out edge (the exit vertex): score 0, the map default
                int bd = -1;
                bool again = false;
                for (int e0 = 0; e0 < out_len; e0 += 64) {
                    const int idx = e0 + lane;
                    const bool valid = idx < out_len;
                    int d = 0;
                    float w = 0.0f;
                    if (valid) {
                        if (!hbm) { d = S.out_dst[x * DG_BOUT + idx]; w = S.out_w[x * DG_BOUT + idx]; }
                        else {
                            d = (int)pool[out_off + 2 * idx];
                            const float td = tt[d];
                            w = td == DG_TT_TEN ? -10.0f : (float)(int)pool[out_off + 2 * idx + 1] - td;   // :404-408
                        }
                    }
                    const int y = d & (DG_SR - 1);
                    const int stg = S.stag[y];
                    float sc = S.sval[y], scb = 0.0f;
                    if constexpr (AB) scb = SB->svalb[y];
                    bool have = valid && stg == d;
                    if (valid && !have) {
                        // the score ring holds the last 1024 finished ids only (a long turned-around
                        // edge makes thousands finish early): a vertex of a resident chunk keeps its
                        // result in its slot until the chunk's row store, everything else is in HBM
                        const int yd = d & (DG_BR - 1);
                        if (d == c_top) { have = true; sc = ctv; if constexpr (AB) scb = DG_BP_NINF; }   // the segment's reference point
                        else if (d == xid) { have = true; sc = xv; if constexpr (AB) scb = 0.0f; }      // an edge to the exit vertex, which lies beyond the stretch
                        else if (S.tag[yd] == d && (S.lens[yd] & DG_BL_DONE) && d >= v_lo) {
                            have = true; sc = S.rscore[yd];
                            if constexpr (AB) scb = SB->rscoreb[yd];
                        } else {
                            const float2 sg = score[d];
                            if (sg.y >= 1.0f) {
                                have = true; sc = (sg.y == 2.0f && xv < 0.5f * DG_BP_NINF) ? xv : sg.x;    // (2.0: k_bp_xtree)
                                if constexpr (AB) scb = sg.y == 2.0f ? sg.x : score_b[d];
                            }
                        }
                    }
                    const unsigned long long miss = __ballot(valid && !have);
                    if (miss) {
                        const unsigned long long far = __ballot(valid && !have && d < v - DG_BFAR);
                        if (far && ndef + sp <= DG_BDEF) {
                            // n waits for its far successor, every entry under it for the entry above
                            const int dfar = __builtin_amdgcn_readlane(d, __ffsll((long long)far) - 1);
                            int above = dfar;
                            for (int q = sp - 1; q >= 0; q--) {
                                int who;
                                if (q == 0) who = v;
                                else if (q - 1 < DG_BSTK) who = __builtin_amdgcn_readfirstlane(S.stk[q - 1]);
                                else who = __builtin_amdgcn_readfirstlane(gstk[q - 1 - DG_BSTK]);
                                if (lane == 0) { S.def_v[ndef] = who; S.def_b[ndef] = above; }
                                ndef++;
                                above = who;
                            }
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                            sp = 0;
                            again = true;
                            break;
                        }
                        // turned-around edges: score those successors first, then come back
                        if (sp + __popcll(miss) > DG_BSTK + gstk_cap) { bad = true; break; }
                        if (valid && !have) {
                            const int si = sp - 1 + __popcll(miss & ((1ull << lane) - 1ull));
                            if (si < DG_BSTK) S.stk[si] = d; else gstk[si - DG_BSTK] = d;
                        }
                        sp += __popcll(miss);
                        again = true;
                        break;
                    }
                    // :399-416 first maximum in list order, strict '>'
                    const float ns = w + sc;
                    const int cnt = out_len - e0 < 64 ? out_len - e0 : 64;
                    if (e0 == 0) { mx = -FLT_MAX; mxb = -FLT_MAX; }
                    for (int i = 0; i < cnt; i++) {
                        const float xs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ns), i));
                        if (xs > mx) { mx = xs; bd = __builtin_amdgcn_readlane(d, i); }
                    }
                    if constexpr (AB) {
                        const float nsb = w + scb;
                        for (int i = 0; i < cnt; i++) mxb = fmaxf(mxb, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nsb), i)));
                    }
                }
                if (bad) break;
                if (again) continue;
                if (lane == 0) {
                    S.stag[n & (DG_SR - 1)] = n;
                    S.sval[n & (DG_SR - 1)] = mx;
                    if constexpr (AB) SB->svalb[n & (DG_SR - 1)] = mxb;
                    if (in_ring && n >= v_lo && n <= v_hi) {   // this chunk: HBM gets it at the end of the chunk
                        S.rscore[x] = mx; S.rbest[x] = bd;
                        if constexpr (AB) SB->rscoreb[x] = mxb;
                        S.lens[x] = lens | DG_BL_DONE;
                    } else {
                        score[n] = make_float2(mx, 1.0f);
                        if constexpr (AB) score_b[n] = mxb; else best[n] = bd;
                        if (in_ring) S.lens[x] = lens | DG_BL_DONE;
                    }
                }
                // (a vertex scored before its chunk is unpacked is scored again at its turn:
                // same successors, same result)
                if (mx > 0.5f * DG_BP_NINF) amax = fmaxf(amax, fabsf(mx));
                if constexpr (AB) if (mxb > 0.5f * DG_BP_NINF) amax = fmaxf(amax, fabsf(mxb));
                sp--;
                if (ndef) {                                // those that waited for n are next
                    if (sp + 1 + ndef > DG_BSTK + gstk_cap) { bad = true; break; }
                    const int base = sp > 0 ? sp : 1;      // (the finished stream vertex stays the bottom)
                    const int k = collect(n, base - 1);
                    if (k) sp = base + k;
                }
            }
        }
        // the chunk's results leave as two row stores
        {
            const int id = v_hi - lane;
            if (id >= v_lo) {
                const int xr = id & (DG_BR - 1);
                if (S.lens[xr] & DG_BL_DONE) {
                    score[id] = make_float2(S.rscore[xr], 1.0f);
                    if constexpr (AB) score_b[id] = SB->rscoreb[xr]; else best[id] = S.rbest[xr];
                }
            }
        }
        // the target terms of the edges that wait in e_* (requested a chunk ago: they are here)
        if (tt_due) { DG_LOAD_TT(); tt_due = false; }
    }
#undef DG_LOAD_REC
#undef DG_LOAD_ENT
#undef DG_LOAD_TT
#undef DG_WRITE_CHUNK
}

// ---- the recurrence, one wave per (target, segment of k_cuts) -------------------
// Every path crosses every cut vertex, so inside a segment
//     score[x] = (best path x -> next cut) + score[next cut]:
// the segments are swept concurrently, each relative to its own upper cut, and the
// first-maximum choices (all that the consensus uses) are the reference's as long as
// the arithmetic is exact, which k_bp_check verifies from the figures left here.
__global__ __launch_bounds__(64) void k_bp_sweep(DgParams p) {
    const uint32_t t = blockIdx.x / p.bp_max, seg = blockIdx.x % p.bp_max;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg >= nseg) return;
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    __shared__ DgBpShared S;
    const int N = (int)p.n_nodes[t];
    const int c_bot = (int)crow[1 + seg];
    const int c_top = seg + 1 < nseg ? (int)crow[2 + seg] : -1;
    // (p.bp_lane: k_bp_sweep_l has been over every piece; what it gave up is marked)
    if (p.bp_lane && !(p.bp_stat[2 * (uint64_t)blockIdx.x] < 0.0f)) return;
    float amax = 0.0f;
    bool bad = false, stuck = false;
    dg_bp_sweep(S, p.nodes + nb, p.best + nb, p.score + nb, p.pool + p.pool_base[t], p.bp_tt + nb,
                c_top >= 0 ? c_top - 1 : N - 1, c_bot, c_top,
                p.stk + (uint64_t)blockIdx.x * p.stk_words, (int)p.stk_words, lane, amax, bad, stuck);
    if (bad) {                                                    // STACK: the host grows the scratch and re-runs
        if (lane == 0) { if (stuck) dg_fail_target(p, t, DG_E_INTERNAL); else { dg_fail(p, DG_E_STACK); p.st->bad_target = t; } }
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const float a = p.score[nb + c_bot].x;                        // this segment's first vertex, relative to c_top
    if (lane == 0) { p.bp_stat[2 * (uint64_t)blockIdx.x] = amax; p.bp_stat[2 * (uint64_t)blockIdx.x + 1] = a; }
}

// ---- the same recurrence with a ROW of eight lanes per piece (full-span pileups, round 3) ----------------------------
// k_bp_sweep gives a wave to a piece and scores one vertex at a time with the out-edges on its lanes: 1.5 lanes busy, 116
// instructions issued per live vertex, and the kernel is bound by exactly that.  Here a wave sweeps eight pieces at once,
// one per row of eight lanes (as k_merge_q does for the merge).  A step scores one vertex per row: five words of the vertex
// (lens, flags, out_off, the final flag of its score, its own target term) fetched by lanes 0 .. 4 in one load and read off
// those lanes; its out-edges one per lane; the successors' (score, target term) from a 32-entry ring of the row's finished
// vertices in LDS (99 % of the forward edges reach at most 17 ids ahead), else from HBM; then the first maximum in list
// order with a strict '>' (AlnGraphBoost.cpp:399-416) as the row's maximum and the lowest lane that holds it, whose value
// is taken as it stands.  The fp32 operations per edge are dg_bp_sweep's, so are the bits.  A successor that has no score
// yet (an edge the merge turned around: 6 % of the live vertices have one) is scored first, from a small stack of the
// row's own in LDS; a row that runs out of stack, meets a vertex with more out-edges than it has lanes, or spends its step
// budget marks its piece (bp_stat < 0) and k_bp_sweep does that piece over.  Finished scores are held back in LDS and
// written eight at a time, one per lane (a step that waits for its loads would wait for the previous step's stores with
// them: gfx950 counts both in vmcnt); the stream's next vertex has its edges on the way and the one behind it its words;
// lanes 5 .. 7 of the fetch bring the flags of the three ids below, and the stream steps over the ones the merge deleted.
// Scores are relative to the piece's upper cut, as there (k_bp_check verifies the exactness of that from bp_stat).
// What steers a row lives in vector registers (see k_emit for why).
// (A LANE per piece -- 64 pieces a wave, plain sequential code -- was built first: exact, and 8 - 14 ms instead of 4.7:
// every load of a wave is 64 requests for 64 different lines, and the vector cache fills at a line every other clock.)
#define DG_BL_STK 16
#define DG_BRW 8                   // lanes of a row
#define DG_BRR 32                  // a row's ring of finished vertices in LDS: (id, score, target term) of id & 31
#define DG_BRP 8                   // finished vertices a row holds back before it writes them to HBM, one per lane
__device__ __forceinline__ uint32_t dbr_ballot(bool q) { return (uint32_t)((__ballot(q) >> (threadIdx.x & (64u - DG_BRW))) & 0xffull); }
// maximum over the row's eight lanes: two quad permutations and the half-row mirror, in registers (DPP)
__device__ __forceinline__ float dbr_max(float m) {
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xf, 0xf, true)));     // quad_perm [1,0,3,2]
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xf, 0xf, true)));     // quad_perm [2,3,0,1]
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x141, 0xf, 0xf, true)));    // row_half_mirror
    return m;
}
__global__ __launch_bounds__(64) void k_bp_sweep_l(DgParams p) {
    constexpr int ROWS = 64 / DG_BRW;
    __shared__ int s_stk[ROWS * DG_BL_STK];
    __shared__ int s_rtag[ROWS * DG_BRR];
    __shared__ float s_rsc[ROWS * DG_BRR], s_rtt[ROWS * DG_BRR];
    __shared__ int s_pid[ROWS * DG_BRP], s_pbd[ROWS * DG_BRP];
    __shared__ float s_pmx[ROWS * DG_BRP];
    if (dg_failed(p)) return;
    const int l = threadIdx.x & (DG_BRW - 1);
    const uint32_t row = threadIdx.x / DG_BRW;
    const uint32_t piece = blockIdx.x * (uint32_t)ROWS + row;
    const uint32_t t = piece / p.bp_max, seg = piece % p.bp_max;
    if (t >= p.T || dg_tskip(p, t)) return;
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg >= nseg) return;
    const uint64_t nb = p.node_base[t];
    const DgNode *nd = p.nodes + nb;
    float2 *score = p.score + nb;
    int32_t *best = p.best + nb;
    const uint32_t *pool = p.pool + p.pool_base[t];
    const float *tt = p.bp_tt + nb;
    const int N = (int)p.n_nodes[t];
    const int c_bot = (int)crow[1 + seg];
    const int c_top = seg + 1 < nseg ? (int)crow[2 + seg] : -1;
    const int v_top = c_top >= 0 ? c_top - 1 : N - 1;
    int *stk = s_stk + row * DG_BL_STK;
    int *rtag = s_rtag + row * DG_BRR;
    float *rsc = s_rsc + row * DG_BRR, *rtt = s_rtt + row * DG_BRR;
    int *pid = s_pid + row * DG_BRP, *pbd = s_pbd + row * DG_BRP;
    float *pmx = s_pmx + row * DG_BRP;
    for (int i = l; i < DG_BRR; i += DG_BRW) rtag[i] = -1;
    const int stk_cap = (int)p.bl_stk;
    float amax = 0.0f;
    int bad = 0;
    uint32_t guard = 0;
    const uint32_t guard_max = 64u * (uint32_t)(v_top - c_bot + 1) + 1000000u;
    // What a step needs of its vertex x -- out_len | in_len, base | flags, out_off, the final flag of its score, its own
    // target term -- is five words at three places: lanes 0 .. 4 of the row fetch one each (one load instruction), and the
    // row reads them off those lanes.  Lanes 5 .. 7 fetch the flags of x - 1, x - 2, x - 3 with them: a third of the ids are
    // vertices the merge deleted, and the stream steps over those it knows of (DBR_BELOW).
    // (a lane's word of vertex x sits at g_base + (x - g_back) << g_shift: the three places differ in base and stride only)
    const char *g_base = reinterpret_cast<const char *>(nd) + ((l == 1 || l >= 5) ? 4 : l == 2 ? 16 : 0);
    uint32_t g_shift = 5;
    if (l == 3) { g_base = reinterpret_cast<const char *>(score) + 4; g_shift = 3; }
    if (l == 4) { g_base = reinterpret_cast<const char *>(tt); g_shift = 2; }
    const int g_back = l >= 5 ? l - 4 : 0;
#define DBR_GATHER(X, OUT)                                                                               \
    do {                                                                                                 \
        const int xl_ = (X) - g_back;                                                                    \
        OUT = *reinterpret_cast<const uint32_t *>(g_base + ((uint64_t)(uint32_t)(xl_ < 0 ? 0 : xl_) << g_shift)); \
    } while (0)
    // the stream's vertex behind X, G = X's gathered words: the first of X - 1, X - 2, X - 3 that is not deleted, else X - 4
    // (-1: the piece ends first; its lowest vertex is a cut vertex, which nobody deletes)
#define DBR_BELOW(G, X, OUT)                                                                             \
    do {                                                                                                 \
        const uint32_t dm_ = dbr_ballot(l >= 5 && (((G) >> 8) & DG_NF_DELETED));                          \
        const int nx_ = (X) - (!(dm_ & 32u) ? 1 : !(dm_ & 64u) ? 2 : !(dm_ & 128u) ? 3 : 4);              \
        OUT = nx_ >= c_bot ? nx_ : -1;                                                                   \
    } while (0)
#define DBR_DECODE(G, LOX, LOY, OFF, FIN, TTN)                                                           \
    do {                                                                                                 \
        LOX = (uint32_t)__shfl((int)(G), 0, DG_BRW); LOY = (uint32_t)__shfl((int)(G), 1, DG_BRW);         \
        OFF = (uint32_t)__shfl((int)(G), 2, DG_BRW);                                                     \
        FIN = __int_as_float(__shfl((int)(G), 3, DG_BRW)); TTN = __int_as_float(__shfl((int)(G), 4, DG_BRW)); \
    } while (0)
#define DBR_EDGES(LOX, LOY, OFF, D, CNT)                                                                 \
    do {                                                                                                 \
        D = c_top; CNT = 0;                                                                              \
        if (l < (int)((LOX) & 0xffffu) && !(((LOY) >> 8) & DG_NF_DELETED)) {                             \
            D = (int)pool[(OFF) + 2u * (uint32_t)l]; CNT = (int)pool[(OFF) + 2u * (uint32_t)l + 1u];     \
        }                                                                                                \
    } while (0)
    // the row's finished vertices go to HBM, one per lane (nobody waits for these stores: the ring has the scores)
    int npend = 0;
#define DBR_FLUSH()                                                                                      \
    do {                                                                                                 \
        if (l < npend) { const int id_ = pid[l]; score[id_] = make_float2(pmx[l], 1.0f); best[id_] = pbd[l]; } \
        npend = 0;                                                                                       \
    } while (0)
    // The stream runs two steps ahead of itself: while vertex v is scored the out-edges of v - 1 and the record of v - 2
    // are on their way (c_*: record and edges of vertex cv, the next one of the stream; raw: the words of cv - 1).
    int v = v_top, n = v_top, sp = 0;
    int going = v_top >= c_bot ? 1 : 0;
    int cv = -1, rv = -1, c_d = 0, c_cnt = 0;          // cv: the vertex whose record and edges c_* hold; rv: the one whose words raw holds
    uint32_t c_lox = 0, c_loy = 0, c_off = 0, raw = 0;
    float c_fin = 0.0f, c_ttn = 0.0f;
    if (going) {
        uint32_t g0;
        DBR_GATHER(v_top, g0);
        DBR_BELOW(g0, v_top, rv);
        if (rv >= 0) DBR_GATHER(rv, raw);
        DBR_DECODE(g0, c_lox, c_loy, c_off, c_fin, c_ttn);
        DBR_EDGES(c_lox, c_loy, c_off, c_d, c_cnt);
        cv = v_top;
    }
    while (going) {
        // every row must leave this loop: the work is bounded by (vertices + edges), far below this budget
        if (++guard > guard_max) { bad = 1; going = 0; continue; }
        uint32_t lox, loy, out_off;
        float fin, ttn;
        int d, cnt;
        if (n == cv) {
            // the stream's vertex: everything is at hand; the pipeline moves on
            lox = c_lox; loy = c_loy; out_off = c_off; fin = c_fin; ttn = c_ttn; d = c_d; cnt = c_cnt;
            // (copied here and now: the loads below then land in the c_* registers themselves, and nobody has to wait for
            // them before the next step)
            asm volatile("" : "+v"(lox), "+v"(loy), "+v"(out_off), "+v"(fin), "+v"(ttn), "+v"(d), "+v"(cnt));
            if (rv >= 0) {
                DBR_DECODE(raw, c_lox, c_loy, c_off, c_fin, c_ttn);
                DBR_EDGES(c_lox, c_loy, c_off, c_d, c_cnt);
                cv = rv;
                DBR_BELOW(raw, cv, rv);
                if (rv >= 0) DBR_GATHER(rv, raw);
            } else cv = -1;
        } else {
            // a vertex off the stack (or the stream's vertex once more, behind one): fetched now
            uint32_t g;
            DBR_GATHER(n, g);
            DBR_DECODE(g, lox, loy, out_off, fin, ttn);
            DBR_EDGES(lox, loy, out_off, d, cnt);
            asm volatile("" : "+v"(d), "+v"(cnt));            // (waited for here: behind the join nothing of this path is in flight)
        }
        const int out_len = (int)(lox & 0xffffu);
        // (a final flag fetched ahead may be out of date -- the vertex was scored since, off the stack: the ring knows, and
        // if it has forgotten the vertex is scored once more, to the same result)
        const int live = (int)!(fin >= 1.0f) & (int)!((loy >> 8) & DG_NF_DELETED) & (int)(rtag[n & (DG_BRR - 1)] != n);
        int next = 1;                                    // 1: n is done with (scored, or nothing to score): the row moves on
        if (live) {
            if (out_len > DG_BRW) { bad = 1; going = 0; continue; }
            // ---- its out-edges, one per lane; their successors' scores and target terms: the ring, else HBM ----
            const bool valid = l < out_len;
            const int x = d & (DG_BRR - 1);
            const bool ref = d == c_top;                     // the piece's reference point: score 0
            const bool hit = valid && rtag[x] == d;
            float sc = rsc[x], tv = rtt[x];
            int have = (int)hit;
            if (dbr_ballot(valid && !hit)) {
                // (what the row holds back goes out first, and is waited for: HBM then has every score the row has made)
                DBR_FLUSH();
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if (valid && !hit) {
                    const float2 sg = score[d];
                    tv = tt[d];
                    sc = sg.x;
                    have = (int)(sg.y >= 1.0f);
                    asm volatile("" : "+v"(sc), "+v"(tv), "+v"(have));       // (waited for here, not behind the join)
                }
            }
            if (ref) { sc = 0.0f; have = 1; }
            const uint32_t miss = dbr_ballot(valid && !have);
            if (miss) {
                // a turned-around edge: that successor first, then this vertex again
                if (sp >= stk_cap) { bad = 1; going = 0; continue; }
                if (l == 0) stk[sp] = n;
                sp++;
                n = __shfl(d, __ffs((int)miss) - 1, DG_BRW);
                next = 0;
            } else {
                const float w = tv == DG_TT_TEN ? -10.0f : (float)cnt - tv;          // :404-408
                const float ns = w + sc;
                float mx = 0.0f;                             // no out edge (the exit vertex): score 0, the map default
                int bd = -1;
                if (out_len > 0) {
                    // :399-416 first maximum in list order, strict '>': the row's maximum, the lowest lane that has it
                    const float m = dbr_max(valid ? ns : -FLT_MAX);
                    const int f = __ffs((int)dbr_ballot(valid && ns == m)) - 1;
                    mx = __shfl(ns, f, DG_BRW);
                    bd = __shfl(d, f, DG_BRW);
                }
                if (l == 0) {
                    pid[npend] = n; pmx[npend] = mx; pbd[npend] = bd;
                    rtag[n & (DG_BRR - 1)] = n; rsc[n & (DG_BRR - 1)] = mx; rtt[n & (DG_BRR - 1)] = ttn;
                }
                npend++;
                if (npend == DG_BRP) DBR_FLUSH();
                if (mx > 0.5f * DG_BP_NINF) amax = fmaxf(amax, fabsf(mx));
            }
        }
        if (next) {
            if (sp > 0) { sp--; n = stk[sp]; }
            else { v = cv; n = v; if (v < 0) going = 0; }     // (the stream's next vertex is the one the pipeline holds)
        }
    }
    DBR_FLUSH();
#undef DBR_GATHER
#undef DBR_BELOW
#undef DBR_DECODE
#undef DBR_EDGES
#undef DBR_FLUSH
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const float a = bad ? 0.0f : score[c_bot].x;                      // this piece's first vertex, relative to c_top
    if (l == 0) {
        p.bp_stat[2 * (uint64_t)piece] = bad ? -1.0f : amax;
        p.bp_stat[2 * (uint64_t)piece + 1] = a;
    }
}

// ---- exactness check of the segmented sweep, one wave per target -------------------
__global__ __launch_bounds__(64) void k_bp_check(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    // Every score is a multiple of 0.5, so fp32 is exact below 2^23.  A vertex of segment i has
    // the absolute score rel + abs(cut i+1), and a candidate adds one edge term
    // (|w| <= max(10, reads)); if all of that stays below 2^22 both the reference's absolute
    // arithmetic and the relative one of k_bp_sweep are exact and pick the same first maxima.
    // Otherwise the target is swept again in one piece.
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const int nseg = (int)crow[0];
    if (nseg <= 1) return;
    bool redo = (p.flags & DG_F_RESWEEP) != 0;
    const float K = (float)(uint32_t)(p.aln_begin[t + 1] - p.aln_begin[t]);
    const float wmax = K > 10.0f ? K : 10.0f;
    float acc = 0.0f;
    for (int i = nseg - 1; i >= 0; i--) {
        const float m = p.bp_stat[2 * ((uint64_t)t * p.bp_max + i)];
        const float a = p.bp_stat[2 * ((uint64_t)t * p.bp_max + i) + 1];
        if (!(m + fabsf(acc) + wmax < 4194304.0f)) redo = true;
        acc += a;
    }
    if (!redo) return;
    __shared__ DgBpShared S;
    const int N = (int)p.n_nodes[t];
    float2 *score = p.score + nb;
    for (int i = lane; i < N; i += 64) score[i] = make_float2(0.0f, 0.0f);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    float amax = 0.0f;
    bool bad = false, stuck = false;
    dg_bp_sweep(S, p.nodes + nb, p.best + nb, score, p.pool + p.pool_base[t], p.bp_tt + nb, N - 1, 0, -1,
                p.stk + (uint64_t)t * p.bp_max * p.stk_words, (int)p.stk_words, lane, amax, bad, stuck);
    if (bad && lane == 0) { if (stuck) dg_fail_target(p, t, DG_E_INTERNAL); else { dg_fail(p, DG_E_STACK); p.st->bad_target = t; } }
}

// ---- the best-edge walk (:443-456), one wave per (target, segment) ----------------
// The best path passes through every cut vertex, so its stretch from one cut to the next is
// walked by its own wave.  A segment leaves one byte per path vertex in its own stretch of
// the scratch (ids of a segment are contiguous): the base, bit 7 = weight >= minWeight.
__global__ __launch_bounds__(64) void k_bp_walk(DgParams p) {
    const uint32_t t = blockIdx.x / p.bp_max, seg = blockIdx.x % p.bp_max;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg >= nseg) return;
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    __shared__ DgWalkShared W;
    const DgNode *nd = p.nodes + nb;
    const int32_t *best = p.best + nb;
    const int N = (int)p.n_nodes[t];
    for (int i = lane; i < DG_WR; i += 64) W.wtag[i] = -1;
    const int c0 = (int)crow[1 + seg];
    const int c1 = seg + 1 < nseg ? (int)crow[2 + seg] : -1;
    const uint8_t eb = nd[0].base, xb = nd[N - 1].base;
    const int minw = p.min_weight;
    uint8_t *tmp = p.cns_tmp + nb + c0;
    int v = c0, cs = c0 >> 6, idx = 0;
    uint32_t steps = 0;
    bool bad = false;
    for (;;) {
        if (v == c1) break;                                  // the next segment starts here
        while (64 * cs < N && 64 * cs < v + 192) {           // (best, base, weight) of 64 ids ahead
            const int id = 64 * cs + lane;
            if (id < N) {
                const uint4 h = *reinterpret_cast<const uint4 *>(&nd[id]);
                const int b = best[id];
                const int xw = id & (DG_WR - 1);
                W.wtag[xw] = id; W.wbest[xw] = b; W.wbase[xw] = (int)(h.y & 0xffu); W.wweight[xw] = (int)h.z;
            }
            cs++;
        }
        const int xw = v & (DG_WR - 1);
        int nxt, w;
        uint8_t base;
        const int wt = W.wtag[xw], wb = W.wbest[xw], wa = W.wbase[xw], ww = W.wweight[xw];
        if (__builtin_amdgcn_readfirstlane(wt) == v) {
            nxt = __builtin_amdgcn_readfirstlane(wb); base = (uint8_t)__builtin_amdgcn_readfirstlane(wa);
            w = __builtin_amdgcn_readfirstlane(ww);
        } else {
            const uint4 h = *reinterpret_cast<const uint4 *>(&nd[v]);
            nxt = __builtin_amdgcn_readfirstlane(best[v]);
            base = (uint8_t)__builtin_amdgcn_readfirstlane((int)(h.y & 0xffu));
            w = __builtin_amdgcn_readfirstlane((int)h.z);
        }
        if (!(base == eb || base == xb)) {
            if (lane == 0) W.wbuf[idx & 63] = (unsigned char)(base | (w >= minw ? 0x80u : 0u));
            if ((idx & 63) == 63) tmp[(idx & ~63) + lane] = W.wbuf[lane];
            idx++;
        }
        if (nxt < 0) break;
        v = nxt;
        if (++steps > (uint32_t)N) { bad = true; break; }
    }
    if (lane < (idx & 63)) tmp[(idx & ~63) + lane] = W.wbuf[lane];     // the last, partial row
    if (lane == 0) {
        if (bad) dg_fail_target(p, t, DG_E_INTERNAL);
        p.bp_len[blockIdx.x] = (uint32_t)idx;
    }
}

// ---- the same walk with a row of eight lanes per (target, segment): eight pieces per wave (round 3; where k_bp_sweep_l
// runs).  A step is one dependent round trip -- best[v], the vertex's base and weight: three words, fetched by lanes
// 0 .. 2 in one load -- so what counts is how many walks are in flight; the bytes leave one at a time.
__global__ __launch_bounds__(64) void k_bp_walk_r(DgParams p) {
    if (dg_failed(p)) return;
    const int l = threadIdx.x & (DG_BRW - 1);
    const uint32_t piece = blockIdx.x * (64u / DG_BRW) + threadIdx.x / DG_BRW;
    const uint32_t t = piece / p.bp_max, seg = piece % p.bp_max;
    if (t >= p.T || dg_tskip(p, t)) return;
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg >= nseg) return;
    const uint64_t nb = p.node_base[t];
    const DgNode *nd = p.nodes + nb;
    const int32_t *best = p.best + nb;
    const int N = (int)p.n_nodes[t];
    const int c0 = (int)crow[1 + seg];
    const int c1 = seg + 1 < nseg ? (int)crow[2 + seg] : -1;
    const uint32_t eb = nd[0].base, xb = nd[N - 1].base;
    const int minw = p.min_weight;
    uint8_t *tmp = p.cns_tmp + nb + c0;
    int v = c0, idx = 0, steps = 0, bad = 0;
    int go = v != c1 ? 1 : 0;
    while (go) {
        const char *a = l == 1 ? reinterpret_cast<const char *>(&nd[v]) + 4 : l == 2 ? reinterpret_cast<const char *>(&nd[v]) + 8
                                                                            : reinterpret_cast<const char *>(&best[v]);
        const uint32_t g = *reinterpret_cast<const uint32_t *>(a);
        const int nxt = __shfl((int)g, 0, DG_BRW);
        const uint32_t base = (uint32_t)__shfl((int)g, 1, DG_BRW) & 0xffu;
        const int w = __shfl((int)g, 2, DG_BRW);
        if (!(base == eb || base == xb)) {
            if (l == 0) tmp[idx] = (uint8_t)(base | (w >= minw ? 0x80u : 0u));
            idx++;
        }
        if (nxt < 0) go = 0;
        else {
            v = nxt;
            if (v == c1) go = 0;                             // the next segment starts here
            else if (++steps > N) { bad = 1; go = 0; }
        }
    }
    if (l == 0) {
        if (bad) dg_fail_target(p, t, DG_E_INTERNAL);
        p.bp_len[piece] = (uint32_t)idx;
    }
}

// ---- consensus segmentation (:327-373) and output, one wave per target ------------
__global__ __launch_bounds__(64) void k_bp_join(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);     // (k_cuts, or k_cuts2 when p.gcuts)
    const uint32_t nseg = crow[0];
    __shared__ uint32_t s_off[DG_BP_PIECES + 1], s_c0[DG_BP_PIECES];
    {
        // where each segment's piece of the path goes: one lane per segment, 64 segments at a time
        // (p.gcuts: at most 64 pieces, one round)
        uint32_t carry = 0;
        for (uint32_t s0 = 0; s0 < nseg || s0 == 0; s0 += 64) {
            const uint32_t sg = s0 + (uint32_t)lane;
            uint32_t len = sg < nseg ? p.bp_len[(uint64_t)t * p.bp_max + sg] : 0u;
            if (p.gcuts) {
                // the pieces that are on the path: the one from enter, which ends at cut number a (or at exit), then
                // a, a + 1, ... up to the first one that ends at exit instead of at the next cut
                const uint32_t e = (uint32_t)lane < nseg ? p.bp_end[(uint64_t)t * p.bp_max + lane] : 1u;
                const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)e, 0);
                const unsigned long long ends = __ballot((uint32_t)lane >= 1u && (uint32_t)lane < nseg && e == 0u);
                bool on = lane == 0;
                if (a != DG_BP_ONE && a < 64u && (ends >> a)) {
                    const uint32_t fe = a + (uint32_t)__ffsll((long long)(ends >> a)) - 1u;
                    on |= (uint32_t)lane >= a && (uint32_t)lane <= fe;
                }
                if (!on) len = 0;
            }
            uint32_t incl = len;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if (lane >= o) incl += up; }
            if (sg < DG_BP_PIECES) { s_off[sg] = carry + incl - len; s_c0[sg] = sg < nseg ? crow[1 + sg] : 0u; }
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) s_off[nseg < DG_BP_PIECES ? nseg : DG_BP_PIECES] = carry;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const uint32_t total = s_off[nseg < DG_BP_PIECES ? nseg : DG_BP_PIECES];
    const uint8_t *tmp = p.cns_tmp + nb;
    int32_t *segs = p.stk + (uint64_t)t * p.bp_max * p.stk_words;      // (range0, range1) pairs
    const uint32_t seg_cap = p.stk_words / 2;
    const uint32_t minlen = p.min_len;
    // every maximal run of path vertices with weight >= minWeight that is long enough
    bool met = false, ovf = false;
    uint32_t offs = 0, nout = 0, keep = 0;
    for (uint32_t s = 0; s < nseg; s++) {
        const uint32_t len = s_off[s + 1] - s_off[s], g0s = s_off[s];
        const uint8_t *src = (p.gcuts && s == 0) ? p.cns_tmp0 + nb : tmp + s_c0[s];
        for (uint32_t j0 = 0; j0 < len; j0 += 64) {
            const uint32_t n = len - j0 < 64 ? len - j0 : 64;
            const bool valid = (uint32_t)lane < n;
            const uint32_t b = valid ? src[j0 + lane] : 0u;
            const unsigned long long vm = n == 64 ? ~0ull : ((1ull << n) - 1ull);
            const unsigned long long m = __ballot(valid && (b >> 7));
            uint32_t pos = 0;
            while (pos < n) {
                const unsigned long long look = (met ? (~m & vm) : m) >> pos;
                if (!look) break;
                const uint32_t f = pos + (uint32_t)__ffsll((long long)look) - 1u;
                const uint32_t g = g0s + j0 + f;
                if (!met) { offs = g; met = true; }
                else {
                    met = false;
                    if (g - offs >= minlen) {
                        if (nout < seg_cap) { if (lane == 0) { segs[2 * nout] = (int)offs; segs[2 * nout + 1] = (int)g; } nout++; keep = g; }
                        else ovf = true;
                    }
                }
                pos = f + 1;
            }
        }
    }
    if (met && total - offs >= minlen) {
        if (nout < seg_cap) { if (lane == 0) { segs[2 * nout] = (int)offs; segs[2 * nout + 1] = (int)total; } nout++; keep = total; }
        else ovf = true;
    }
    // only the bases some segment covers are shipped
    __shared__ unsigned long long s_co, s_so;
    if (lane == 0) {
        if (ovf) dg_fail(p, DG_E_STACK);
        unsigned long long co = atomicAdd(&p.st->cns_top, (unsigned long long)keep);
        unsigned long long so = atomicAdd(&p.st->seg_top, (unsigned long long)nout);
        if (co + keep > p.cns_cap || so + nout > p.seg_cap) { dg_fail(p, DG_E_OUT_OVF); s_co = ~0ull; }
        else s_co = co;
        s_so = so;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const unsigned long long co = s_co, so = s_so;
    if (co == ~0ull) { keep = 0; nout = 0; }
    if (lane == 0) {
        p.cns_off[t] = co == ~0ull ? 0ull : co; p.cns_len[t] = keep;
        p.seg_first[t] = so; p.n_seg[t] = nout;
    }
    if (co == ~0ull) return;
    uint8_t *out = p.cns + co;
    for (uint32_t s = 0; s < nseg; s++) {
        const uint32_t g0s = s_off[s], len = s_off[s + 1] - s_off[s];
        const uint8_t *src = (p.gcuts && s == 0) ? p.cns_tmp0 + nb : tmp + s_c0[s];
        for (uint32_t j = lane; j < len && g0s + j < keep; j += 64) out[g0s + j] = (uint8_t)(src[j] & 0x7fu);
    }
    for (uint32_t i = lane; i < nout; i += 64) {
        p.seg_r0[so + i] = segs[2 * i];
        p.seg_r1[so + i] = segs[2 * i + 1];
    }
}


// ====================================================================================
// bestPath on the pieces of k_cuts2 (partial-span pileups, p.gcuts): see dg_bp_sweep.
//   k_bp_sweep_g<0>  A of every piece but the last (cut worth 0, exit -inf); the last piece, which ends
//                    in the exit vertex, gets its absolute scores at once
//   k_bp_sweep_g<1>  B of every piece but the last (cut -inf, exit 0)
//   k_bp_comb        per target: the absolute score of every cut, from the last one down:
//                    score[c_k] = max(A_k + score[c_k+1], B_k); the exactness bound; deferred vertices reset
//   k_bp_sweep_g<2>  the absolute scores and first-maximum choices of every piece but the last
//                    (a target that failed the bound, or has too many deferred vertices: one sweep, as a whole)
//   k_bp_defer       the deferred vertices nobody asked for (enter is the last of them)
//   k_bp_walk_g      the best-edge walk in pieces: from every cut to the next cut or to exit, and from
//                    enter to the first cut it meets;  k_bp_join chains the pieces that are on the path
// ====================================================================================

__global__ __launch_bounds__(64) void k_bp_reset_def(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t *dl = p.defer + (uint64_t)t * (DG_DEFER_MAX + 1u);
    const uint32_t n = dl[0];
    if (n == DG_BP_ONE) return;
    const uint64_t nb = p.node_base[t];
    for (uint32_t i = threadIdx.x; i < n; i += 64)
        if (p.score[nb + dl[1 + i]].y != 2.0f) { p.score[nb + dl[1 + i]] = make_float2(0.0f, 0.0f); p.best[nb + dl[1 + i]] = -1; }
}

// the tree of vertices that lead to exit and nowhere else (see dg_bp_sweep): absolute scores, final flag 2.0
__global__ __launch_bounds__(64) void k_bp_xtree(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t) || threadIdx.x != 0) return;
    const uint64_t nb = p.node_base[t];
    DgNode *nd = p.nodes + nb;
    const uint32_t *pool = p.pool + p.pool_base[t];
    const float *tt = p.bp_tt + nb;
    const int N = (int)p.n_nodes[t];
    int32_t *q = p.queue + nb;                              // (the merge is over: its FIFO is free)
    int qh = 0, qt = 0;
    q[qt++] = N - 1;
    while (qh < qt) {
        const int z = q[qh++];
        const float sz = z == N - 1 ? 0.0f : p.score[nb + z].x;
        const float td = tt[z];
        const DgNode nz = nd[z];
        for (uint32_t i = 0; i < nz.in_len; i++) {
            const int sv = (int)pool[nz.in_off + i];
            if (sv == 0) continue;                          // (enter is k_bp_defer's)
            const DgNode ns = nd[sv];
            if ((ns.flags & DG_NF_DELETED) || ns.out_len != 1 || p.score[nb + sv].y == 2.0f) continue;
            const float w = td == DG_TT_TEN ? -10.0f : (float)(int)pool[ns.out_off + 1u] - td;       // :404-408
            p.score[nb + sv] = make_float2(w + sz, 2.0f);
            p.best[nb + sv] = z;
            nd[sv].flags |= DG_NF_DEFER;                    // the segments' streams pass over it
            if (qt < N) q[qt++] = sv;
        }
    }
}

template <int PASS>
__global__ __launch_bounds__(64) void k_bp_sweep_g(DgParams p) {
    const uint32_t t = blockIdx.x / p.bp_max, seg = blockIdx.x % p.bp_max;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg >= nseg) return;
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    __shared__ DgBpShared S;
    const int N = (int)p.n_nodes[t];
    float2 *score = p.score + nb;
    const bool one = p.defer[(uint64_t)t * (DG_DEFER_MAX + 1u)] == DG_BP_ONE || nseg <= 1;
    float amax = 0.0f;
    bool bad = false, stuck = false;
    if (one) {
        // in one piece, as the reference does it (every vertex in the stream, the deferred ones too)
        if (PASS != 2 || seg != 0) return;
        for (int i = lane; i < N; i += 64) score[i] = make_float2(0.0f, 0.0f);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        dg_bp_sweep(S, p.nodes + nb, p.best + nb, score, p.pool + p.pool_base[t], p.bp_tt + nb, N - 1, 0, -1,
                    p.stk + (uint64_t)blockIdx.x * p.stk_words, (int)p.stk_words, lane, amax, bad, stuck);
    } else {
        if (p.bp_fused) return;                                // (k_bp_sweep_ab / k_bp_abs / k_bp_choose did the pieces)
        const bool last = seg + 1 == nseg;
        if (last && PASS != 0) return;
        const int c_bot = (int)crow[1 + seg];
        const int c_top = last ? -1 : (int)crow[2 + seg];
        const int v_top = last ? N - 1 : c_top - 1;
        if (PASS != 0) {
            for (int i = c_bot + lane; i <= v_top; i += 64) if (score[i].y != 2.0f) score[i] = make_float2(0.0f, 0.0f);   // (the pass before left its own)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        float *ab = p.bp_ab + 4ull * blockIdx.x;
        const float ctv = PASS == 0 ? 0.0f : PASS == 1 ? DG_BP_NINF : ab[2];
        const float xv = PASS == 0 && !last ? DG_BP_NINF : 0.0f;   // (the last piece ends in exit: its scores are absolute at once)
        dg_bp_sweep(S, p.nodes + nb, p.best + nb, score, p.pool + p.pool_base[t], p.bp_tt + nb, v_top, c_bot, c_top,
                    p.stk + (uint64_t)blockIdx.x * p.stk_words, (int)p.stk_words, lane, amax, bad, stuck,
                    ctv, last ? -1 : N - 1, xv, true);
        if (!bad) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            const float a = score[c_bot].x;                    // (piece 0 begins with enter, which is deferred: not used)
            if (lane == 0) {
                if (PASS == 0) { ab[0] = a; ab[3] = amax; }
                else if (PASS == 1) { ab[1] = a; ab[3] = fmaxf(ab[3], amax); }
                else ab[3] = fmaxf(ab[3], amax);
            }
        }
    }
    if (bad && lane == 0) { if (stuck) dg_fail_target(p, t, DG_E_INTERNAL); else { dg_fail(p, DG_E_STACK); p.st->bad_target = t; } }
}

// ---- the pieces in ONE sweep (p.bp_fused) -------------------------------------------------------------------------
// A and B of every vertex come out of the same pass (dg_bp_sweep<true>): they are the same recurrence on the same edges
// with different boundary values, and nothing is decided in it.  k_bp_comb turns the pieces' (A, B) into the absolute
// score of every cut as before; then the absolute score of every vertex is an elementwise max(A + score[cut], B)
// (k_bp_abs) and its first-maximum choice a loop over its own out-list (k_bp_choose): vertex-parallel kernels instead of
// a second and a third sequential sweep.  Exact for the reason the three-sweep form is (all values multiples of 0.5
// below 2^22: k_bp_comb's bound); k_bp_choose checks it on the way -- the maximum it finds must BE the vertex's
// absolute score, else the target is swept again in one piece.
__global__ __launch_bounds__(64) void k_bp_sweep_ab(DgParams p) {
    const uint32_t t = blockIdx.x / p.bp_max, seg = blockIdx.x % p.bp_max;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg >= nseg) return;
    if (p.defer[(uint64_t)t * (DG_DEFER_MAX + 1u)] == DG_BP_ONE || nseg <= 1) return;       // k_bp_sweep_g<2> has it, whole
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    __shared__ DgBpShared S;
    __shared__ DgBpSharedB SB;
    const int N = (int)p.n_nodes[t];
    float2 *score = p.score + nb;
    float amax = 0.0f;
    bool bad = false, stuck = false;
    const bool last = seg + 1 == nseg;
    const int c_bot = (int)crow[1 + seg];
    const int c_top = last ? -1 : (int)crow[2 + seg];
    const int v_top = last ? N - 1 : c_top - 1;
    float *ab = p.bp_ab + 4ull * blockIdx.x;
    if (last)       // the last piece ends in exit: its scores are absolute at once, its choices final
        dg_bp_sweep<false>(S, p.nodes + nb, p.best + nb, score, p.pool + p.pool_base[t], p.bp_tt + nb, v_top, c_bot, c_top,
                           p.stk + (uint64_t)blockIdx.x * p.stk_words, (int)p.stk_words, lane, amax, bad, stuck,
                           0.0f, -1, 0.0f, true);
    else
        dg_bp_sweep<true>(S, p.nodes + nb, p.best + nb, score, p.pool + p.pool_base[t], p.bp_tt + nb, v_top, c_bot, c_top,
                          p.stk + (uint64_t)blockIdx.x * p.stk_words, (int)p.stk_words, lane, amax, bad, stuck,
                          0.0f, N - 1, DG_BP_NINF, true, p.score_b + nb, &SB);
    if (!bad) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const float a = score[c_bot].x;                    // (piece 0 begins with enter, which is deferred: not used)
        const float b = last ? 0.0f : p.score_b[nb + c_bot];
        if (lane == 0) { ab[0] = a; ab[1] = b; ab[3] = amax; }
    }
    if (bad && lane == 0) { if (stuck) dg_fail_target(p, t, DG_E_INTERNAL); else { dg_fail(p, DG_E_STACK); p.st->bad_target = t; } }
}

// absolute scores of the pieces' vertices: max(A + score[upper cut], B); one block per (target, piece)
__global__ __launch_bounds__(256) void k_bp_abs(DgParams p) {
    const uint32_t t = blockIdx.x / p.bp_max, seg = blockIdx.x % p.bp_max;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg + 1 >= nseg) return;                                       // (the last piece is absolute already)
    if (p.defer[(uint64_t)t * (DG_DEFER_MAX + 1u)] == DG_BP_ONE) return;
    const uint64_t nb = p.node_base[t];
    const float s = p.bp_ab[4ull * blockIdx.x + 2];
    const int c_bot = (int)crow[1 + seg], c_top = (int)crow[2 + seg];
    for (int v = c_bot + (int)threadIdx.x; v < c_top; v += 256) {
        const float2 sg = p.score[nb + v];
        if (sg.y != 1.0f) continue;                                    // not scored (deleted, deferred) or the exit tree's (2.0: absolute)
        const float via_a = sg.x + s, via_b = p.score_b[nb + v];
        p.score[nb + v] = make_float2(via_a > via_b ? via_a : via_b, 1.0f);
    }
}

// the first-maximum choice of every vertex of the pieces (AlnGraphBoost.cpp:399-416), from absolute scores
__global__ __launch_bounds__(256) void k_bp_choose(DgParams p) {
    const uint32_t t = blockIdx.x / p.bp_max, seg = blockIdx.x % p.bp_max;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg + 1 >= nseg) return;
    uint32_t *dl = p.defer + (uint64_t)t * (DG_DEFER_MAX + 1u);
    if (dl[0] == DG_BP_ONE) return;
    const uint64_t nb = p.node_base[t];
    const DgNode *nd = p.nodes + nb;
    const uint32_t *pool = p.pool + p.pool_base[t];
    const float *tt = p.bp_tt + nb;
    const float2 *score = p.score + nb;
    const int c_bot = (int)crow[1 + seg], c_top = (int)crow[2 + seg];
    bool wrong = false;
    for (int v = c_bot + (int)threadIdx.x; v < c_top; v += 256) {
        const float2 sv = score[v];
        if (sv.y != 1.0f) continue;
        const uint4 h = *reinterpret_cast<const uint4 *>(&nd[v]);
        const uint32_t out_len = h.x & 0xffffu, out_off = (reinterpret_cast<const uint4 *>(&nd[v]) + 1)->x;
        float mx = 0.0f;
        int bd = -1;
        for (uint32_t e = 0; e < out_len; e++) {
            const int d = (int)pool[out_off + 2u * e];
            const float2 sd = score[d];
            const float td = tt[d];
            const float w = td == DG_TT_TEN ? -10.0f : (float)(int)pool[out_off + 2u * e + 1u] - td;     // :404-408
            const float ns = w + sd.x;
            // (the exit vertex has no score of its own: 0, the map default; every other successor has one by now)
            if (sd.y < 1.0f && d != (int)p.n_nodes[t] - 1) wrong = true;
            if (e == 0 || ns > mx) { mx = ns; bd = d; }                 // strict '>': the first maximum wins
        }
        if (mx != sv.x) wrong = true;                                   // the recurrence must hold with the absolute scores
        p.best[nb + v] = bd;
    }
    if (wrong) dl[0] = DG_BP_ONE;                                       // (cannot happen below the 2^22 bound: swept again, whole)
}

__global__ __launch_bounds__(64) void k_bp_comb(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const int nseg = (int)crow[0];
    uint32_t *dl = p.defer + (uint64_t)t * (DG_DEFER_MAX + 1u);
    if (nseg <= 1 || dl[0] == DG_BP_ONE) return;
    const uint64_t nb = p.node_base[t];
    if (threadIdx.x == 0) {
        // every score is a multiple of 0.5: fp32 is exact below 2^23.  What the pieces' relative and the
        // reference's absolute arithmetic can form stays below 2^22 if every piece's largest value, the
        // absolute score of its upper cut and one edge term do (k_bp_check has the argument); else one piece
        bool redo = (p.flags & DG_F_RESWEEP) != 0;
        const float K = (float)(uint32_t)(p.aln_begin[t + 1] - p.aln_begin[t]);
        const float wmax = K > 10.0f ? K : 10.0f;
        float *ab = p.bp_ab + 4ull * ((uint64_t)t * p.bp_max);
        float abs_ = ab[4 * (nseg - 1)];                                   // the last piece's first vertex, absolute (also when k_bp_xtree scored it)
        if (!(ab[4 * (nseg - 1) + 3] + wmax < 4194304.0f)) redo = true;
        for (int k = nseg - 2; k >= 0; k--) {
            ab[4 * k + 2] = abs_;
            if (!(ab[4 * k + 3] + fabsf(abs_) + wmax < 4194304.0f)) redo = true;
            if (k >= 1) {
                const float2 sc = p.score[nb + crow[1 + k]];
                const float viaA = ab[4 * k] + abs_, viaB = ab[4 * k + 1];
                abs_ = sc.y == 2.0f ? sc.x : viaA > viaB ? viaA : viaB;       // (a cut inside the exit tree has its score already)
            }
        }
        if (redo) dl[0] = DG_BP_ONE;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const uint32_t n = dl[0];
    if (n == DG_BP_ONE) return;
    for (uint32_t i = threadIdx.x; i < n; i += 64)
        if (p.score[nb + dl[1 + i]].y != 2.0f) { p.score[nb + dl[1 + i]] = make_float2(0.0f, 0.0f); p.best[nb + dl[1 + i]] = -1; }
}

// the deferred vertices that no sweep was asked for: each once all its successors have their scores
__global__ __launch_bounds__(64) void k_bp_defer(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t *dl = p.defer + (uint64_t)t * (DG_DEFER_MAX + 1u);
    const uint32_t n = dl[0];
    if (n == DG_BP_ONE || threadIdx.x != 0) return;
    const uint64_t nb = p.node_base[t];
    const DgNode *nd = p.nodes + nb;
    const uint32_t *pool = p.pool + p.pool_base[t];
    const float *tt = p.bp_tt + nb;
    float2 *score = p.score + nb;
    int32_t *best = p.best + nb;
    uint32_t left = n;
    for (uint32_t round = 0; round <= n && left; round++) {
        left = 0;
        for (uint32_t i = 0; i < n; i++) {
            const int v = (int)dl[1 + i];
            if (score[v].y >= 1.0f || (nd[v].flags & DG_NF_DELETED)) continue;
            const DgNode nv = nd[v];
            bool ready = true;
            float mx = 0.0f;
            int bd = -1;
            for (uint32_t e = 0; e < nv.out_len; e++) {
                const int d = (int)pool[nv.out_off + 2u * e];
                const float2 sd = score[d];
                if (sd.y < 1.0f) { ready = false; break; }
                const float td = tt[d];
                const float w = td == DG_TT_TEN ? -10.0f : (float)(int)pool[nv.out_off + 2u * e + 1u] - td;     // :404-408
                const float ns = w + sd.x;
                if (e == 0 || ns > mx) { mx = ns; bd = d; }                 // :399-416 first maximum, strict '>'
            }
            if (!ready) { left++; continue; }
            score[v] = make_float2(mx, 1.0f);
            best[v] = bd;
        }
    }
    if (left) dg_fail_target(p, t, DG_E_INTERNAL);
}

__global__ __launch_bounds__(64) void k_bp_walk_g(DgParams p) {
    const uint32_t t = blockIdx.x / p.bp_max, seg = blockIdx.x % p.bp_max;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg >= nseg) return;
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    __shared__ DgWalkShared W;
    const DgNode *nd = p.nodes + nb;
    const int32_t *best = p.best + nb;
    const int N = (int)p.n_nodes[t];
    for (int i = lane; i < DG_WR; i += 64) W.wtag[i] = -1;
    const int c0 = (int)crow[1 + seg];
    const int c1 = seg + 1 < nseg ? (int)crow[2 + seg] : -1;
    const int mycut = (uint32_t)lane >= 1u && (uint32_t)lane < nseg ? (int)crow[1 + lane] : -1;   // piece 0 stops at any cut
    const uint8_t eb = nd[0].base, xb = nd[N - 1].base;
    const int minw = p.min_weight;
    uint8_t *tmp = seg == 0 ? p.cns_tmp0 + nb : p.cns_tmp + nb + c0;
    int v = c0, cs = c0 >> 6, idx = 0;
    uint32_t steps = 0, endk = 0;                          // endk: piece 0: cut index reached (DG_BP_ONE: exit); others: 1 = the next cut
    bool bad = false;
    if (seg == 0) endk = DG_BP_ONE;
    for (;;) {
        if (seg != 0) { if (v == c1) { endk = 1; break; } }
        else if (v != c0) {
            const unsigned long long hit = __ballot(mycut == v);
            if (hit) { endk = (uint32_t)__ffsll((long long)hit) - 1u; break; }
        }
        if (v > 64 * cs + 512) cs = v >> 6;                  // (a jump: enter's edges lead anywhere)
        while (64 * cs < N && 64 * cs < v + 192) {           // (best, base, weight) of 64 ids ahead
            const int id = 64 * cs + lane;
            if (id < N) {
                const uint4 h = *reinterpret_cast<const uint4 *>(&nd[id]);
                const int b = best[id];
                const int xw = id & (DG_WR - 1);
                W.wtag[xw] = id; W.wbest[xw] = b; W.wbase[xw] = (int)(h.y & 0xffu); W.wweight[xw] = (int)h.z;
            }
            cs++;
        }
        const int xw = v & (DG_WR - 1);
        int nxt, w;
        uint8_t base;
        const int wt = W.wtag[xw], wb = W.wbest[xw], wa = W.wbase[xw], ww = W.wweight[xw];
        if (__builtin_amdgcn_readfirstlane(wt) == v) {
            nxt = __builtin_amdgcn_readfirstlane(wb); base = (uint8_t)__builtin_amdgcn_readfirstlane(wa);
            w = __builtin_amdgcn_readfirstlane(ww);
        } else {
            const uint4 h = *reinterpret_cast<const uint4 *>(&nd[v]);
            nxt = __builtin_amdgcn_readfirstlane(best[v]);
            base = (uint8_t)__builtin_amdgcn_readfirstlane((int)(h.y & 0xffu));
            w = __builtin_amdgcn_readfirstlane((int)h.z);
        }
        if (!(base == eb || base == xb)) {
            if (lane == 0) W.wbuf[idx & 63] = (unsigned char)(base | (w >= minw ? 0x80u : 0u));
            if ((idx & 63) == 63) tmp[(idx & ~63) + lane] = W.wbuf[lane];
            idx++;
        }
        if (nxt < 0) break;
        v = nxt;
        if (++steps > (uint32_t)N) { bad = true; break; }
    }
    if (lane < (idx & 63)) tmp[(idx & ~63) + lane] = W.wbuf[lane];     // the last, partial row
    if (lane == 0) {
        if (bad) dg_fail_target(p, t, DG_E_INTERNAL);
        p.bp_len[blockIdx.x] = (uint32_t)idx;
        p.bp_end[blockIdx.x] = endk;
    }
}
