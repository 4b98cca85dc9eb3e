// k_merge_q.hip.h -- mergeNodes (AlnGraphBoost.cpp:129-273) with EIGHT segments per wave (round 2: four).
//
// k_merge (k_merge.hip.h) gives a wave to one segment between two cut vertices and uses, on a typical visit, 2 - 4 of
// its 64 lanes: the kernel is bound by instruction issue, and almost all of the instructions are bookkeeping that
// does not care how many lanes take part.  Here a wave sweeps DQ_ROWS segments at once, one per row of DQ_W lanes (8 x 8 since round 3; 4 x 16 in round 2): the code
// is the same sweep, written for a row -- a list entry per lane of the row, ballots cut down to the row's bits,
// cross-lane reads through the row (ds_bpermute), per-row state in vector registers -- so one instruction serves
// DQ_ROWS visits as long as the rows do the same thing.  When one row merges and the others do not, the others wait
// (the wave executes the union of its rows' paths); lists longer than a row take the reference-literal
// single-lane path (dgg_*), as lists longer than a wave do in k_merge.
//
// Exactness is inherited: every row runs exactly the sweep of dg_merge_segment on its own segment, and segments
// touch disjoint state (the cut argument above k_cuts).  Full-span pileups only (p.gcuts == 0).
#pragma once
#include <hip/hip_runtime.h>
#include "dagcon_dev.h"

#ifndef DQ_W
#define DQ_W 8                       // lanes of a row (round 3: 8 rows of 8 lanes; 4 x 16 / 2 x 32 -> merge 13.3 / 18.9 ms against 11.2 at configs[1])
#endif
#define DQ_ROWS (64 / DQ_W)          // rows (segments) of a wave
#define DQ_ALL ((uint32_t)((1ull << DQ_W) - 1ull))
#define DQ_LO ((1u << (DQ_W / 2)) - 1u)   // the lower half of a row: in entries of the one-look path
#ifndef DQ_RING
#define DQ_RING 16                   // the youngest queue entries of a row, in LDS: (id, out_len | in_len << 16, out_off, in_off)
#endif
typedef uint32_t qmask;              // one bit per lane of the row

__device__ __forceinline__ qmask dq_ballot(bool p) { return (qmask)((__ballot(p) >> (threadIdx.x & (64u - DQ_W))) & (unsigned long long)DQ_ALL); }
__device__ __forceinline__ int dq_rl(int v, int l) { return __shfl(v, l, DQ_W); }          // lane l of the caller's row
#define DQ_LT(lane) ((1u << (lane)) - 1u)
// a row's earlier stores before its later loads (a single lane's, on the literal path)
#define DQ_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#define DQ_LANE0(G, BODY)                                          \
    do {                                                           \
        bool e_ = false;                                           \
        if (lane == 0) { DgGraph gs = (G); BODY; e_ = gs.err; }    \
        DQ_FENCE();                                                \
        (G).err = dq_ballot(e_) != 0;                              \
    } while (0)

__device__ __forceinline__ void dq_fail(DgGraph &g, uint32_t bit, int lane) {
    if (lane == 0) {
        if (bit & DG_E_TARGET_MASK) atomicOr(g.tfail, bit);
        else { atomicOr(&g.st->err_flags, bit); g.st->bad_target = g.t; }
    }
    g.err = true;
}
__device__ __forceinline__ int dq_sum_masked(int v, qmask m) {
    int acc = 0;
    while (m) {
        const int f = __ffs((int)m) - 1;
        acc += dq_rl(v, f);
        m &= m - 1u;
    }
    return acc;
}
// smallest base > last that at least two candidate lanes share; returns the base (or 256) and the mask of its lanes
__device__ __forceinline__ int dq_pick_group(qmask cand, int base, int last, int lane, qmask *mask) {
    int best = 256;
    qmask bm = 0;
    const int key = ((cand >> lane) & 1u) ? base : -1 - lane;
    while (cand) {
        const int f = __ffs((int)cand) - 1;
        const int b = dq_rl(key, f);
        const qmask same = dq_ballot(key == b);
        if (b > last && b < best && __popc(same) >= 2) { best = b; bm = same; }
        cand &= ~same;
    }
    *mask = bm;
    return best;
}
__device__ __forceinline__ uint32_t dq_alloc(DgGraph &g, uint32_t words, int lane) {
    uint32_t off = 0;
    if (lane == 0) { DgGraph gs = g; off = dgg_alloc(gs, words); }
    off = (uint32_t)dq_rl((int)off, 0);
    if (off == 0xFFFFFFFFu) g.err = true;
    return off;
}

// Removes from in[v] every source held (in `vid`) by a lane of vm (stable), then appends `app` when app >= 0.
// pend_delta is added to v's pending counter.  Requires in_len(v) <= DQ_W.
__device__ inline void dq_in_rewrite(DgGraph &g, int v, int vid, qmask vm, int app, int pend_delta, int lane) {
    const uint4 h = dg_lo16(&DG_NV(g, v)), h2 = dg_hi16(&DG_NV(g, v));
    const int len = DG_H_INLEN(h);
    uint32_t off = DG_H2_INOFF(h2);
    int cap = DG_H2_INCAP(h2);
    int e = -1;
    if (lane < len) e = (int)DG_PW(g, off + lane);
    bool rm = false;
    for (qmask m = vm; m; m &= m - 1u) rm |= (e == dq_rl(vid, __ffs((int)m) - 1));
    const bool keep = lane < len && !rm;
    const qmask km = dq_ballot(keep);
    int nlen = __popc(km);
    const int nidx = __popc(km & DQ_LT(lane));
    if (app >= 0 && nlen + 1 > cap) {
        uint32_t ncap = 2u * (uint32_t)(nlen + 1);
        if (ncap < 4) ncap = 4;
        const uint32_t noff = dq_alloc(g, ncap, lane);
        if (noff == 0xFFFFFFFFu) return;
        off = noff; cap = (int)ncap;
    }
    if (keep) DG_PW(g, off + nidx) = (uint32_t)e;
    if (app >= 0) { if (lane == 0) DG_PW(g, off + nlen) = (uint32_t)app; nlen++; }
    if (lane == 0) {
        DgNode *n = &DG_NV(g, v);
        n->in_len = (uint16_t)nlen; n->in_off = off; n->in_cap = (uint16_t)cap;
        if (pend_delta) n->pending = DG_H_PEND(h) + pend_delta;
    }
}

// ---- mergeOutNodes, one group (AlnGraphBoost.cpp:229-266); as dgw_merge_out_group, on a row ----------
// Lanes of M hold out entries of u: d (target), cnt, h (first half of the target's record); survivor = lowest lane.
// Returns false, with nothing modified, when a list involved is longer than a row.
__device__ inline bool dq_merge_out_group(DgGraph &g, int u, uint32_t u_out_off, qmask M, int d, int &cnt, uint4 h,
                                          bool valid_out, qmask out_lanes, int lane) {
    const int an_lane = __ffs((int)M) - 1;
    const qmask vm = M & ~(1u << an_lane);
    const int an = dq_rl(d, an_lane);
    const bool member = (M >> lane) & 1u;
    uint4 h2 = make_uint4(0, 0, 0, 0);
    if (member) h2 = dg_hi16(&DG_NV(g, d));
    // members' out entries flattened onto lanes 0..L-1: survivor's first, then victims in order
    int L = 0, src = -1, e = 0;
    uint32_t src_off = 0;
    {
        qmask mm = M;
        while (mm) {
            const int ml = __ffs((int)mm) - 1;
            mm &= mm - 1u;
            const int mlen = dq_rl(DG_H_OUTLEN(h), ml);
            const uint32_t moff = (uint32_t)dq_rl((int)DG_H2_OUTOFF(h2), ml);
            if (lane >= L && lane < L + mlen) { src = ml; e = lane - L; src_off = moff; }
            L += mlen;
        }
    }
    if (L > DQ_W) return false;
    const bool fl = lane < L;
    int n2 = -1, c2 = 0;
    if (fl) { n2 = (int)DG_PW(g, src_off + 2 * e); c2 = (int)DG_PW(g, src_off + 2 * e + 1); }
    const bool vic_entry = fl && src != an_lane;
    uint4 hn2 = make_uint4(0, 0, 0, 0);
    if (vic_entry) hn2 = dg_lo16(&DG_NV(g, n2));
    if (dq_ballot(vic_entry && DG_H_INLEN(hn2) > DQ_W)) return false;
    if (g.sh && dq_ballot(vic_entry && ((hn2.y >> 8) & DG_NF_SHARED))) return false;      // in[exit] is shared: literal path

    // ---- nothing has been modified up to here ----
    const int add_cnt = dq_sum_masked(cnt, vm);
    const int add_w = dq_sum_masked(DG_H_WEIGHT(h), vm);
    // :246-265 fold the victims' out edges into the survivor's, first occurrence order
    qmask rem = dq_ballot(fl), first_m = 0;
    const qmask vics = dq_ballot(vic_entry);
    int newcnt = c2;
    while (rem) {
        const int f = __ffs((int)rem) - 1;
        const int x = dq_rl(n2, f);
        const qmask same = dq_ballot(fl && n2 == x);
        rem &= ~same;
        first_m |= 1u << f;
        const int tot = dq_sum_masked(c2, same);
        if (lane == f) newcnt = tot;
        const qmask vsame = same & vics;
        const int nv = __popc(vsame);
        if (nv) {
            const bool is_new = (vsame >> f) & 1u;         // survivor had no edge to x
            dq_in_rewrite(g, x, d, vm, is_new ? an : -1, -(nv - (is_new ? 1 : 0)), lane);
            if (g.err) return true;
        }
    }
    // survivor's new out list
    {
        const int nlen = __popc(first_m);
        uint32_t off = (uint32_t)dq_rl((int)DG_H2_OUTOFF(h2), an_lane);
        int cap = dq_rl(DG_H2_OUTCAP(h2), an_lane);
        if (nlen > cap) {
            uint32_t ncap = 2u * (uint32_t)(nlen + 1);
            if (ncap < 4) ncap = 4;
            const uint32_t noff = dq_alloc(g, 2u * ncap, lane);
            if (noff == 0xFFFFFFFFu) return true;
            off = noff; cap = (int)ncap;
        }
        if ((first_m >> lane) & 1u) {
            const int idx = __popc(first_m & DQ_LT(lane));
            DG_PW(g, off + 2 * idx) = (uint32_t)n2;
            DG_PW(g, off + 2 * idx + 1) = (uint32_t)newcnt;
        }
        const int an_w = dq_rl(DG_H_WEIGHT(h), an_lane);
        if (lane == 0) {
            DgNode *a = &DG_NV(g, an);
            a->out_len = (uint16_t)nlen; a->out_off = off; a->out_cap = (uint16_t)cap;
            a->weight = an_w + add_w;
        }
    }
    // u's out list without the victims (stable), survivor's edge count updated.  out_lanes: the lanes that hold
    // u's out entries, in list order
    {
        const bool keep = valid_out && !((vm >> lane) & 1u);
        const qmask km = dq_ballot(keep);
        if (keep) {
            const int idx = __popc(km & DQ_LT(lane));
            DG_PW(g, u_out_off + 2 * idx) = (uint32_t)d;
            if (lane == an_lane) cnt += add_cnt;
            DG_PW(g, u_out_off + 2 * idx + 1) = (uint32_t)cnt;
        }
        if (lane == 0) DG_NV(g, u).out_len = (uint16_t)__popc(km);
    }
    (void)out_lanes;
    // AlnGraphBoost.cpp:269-273 for every victim
    if ((vm >> lane) & 1u) {
        DgNode *vn = &DG_NV(g, d);
        vn->out_len = 0; vn->in_len = 0; vn->flags |= DG_NF_DELETED;
    }
    return true;
}

// ---- mergeInNodes, one group (AlnGraphBoost.cpp:176-212); as dgw_merge_in_group, on a row ---------------
// Lanes 0.. hold n's in entries: s (source), h (first half of its record).  *an_out = survivor.
__device__ inline bool dq_merge_in_group(DgGraph &g, int n, uint32_t n_in_off, qmask M, int s, uint4 h, bool valid_in,
                                         int lane, int *an_out) {
    const int an_lane = __ffs((int)M) - 1;
    const qmask vm = M & ~(1u << an_lane);
    const int an = dq_rl(s, an_lane);
    *an_out = an;
    const bool member = (M >> lane) & 1u;
    const bool victim = (vm >> lane) & 1u;
    uint4 h2 = make_uint4(0, 0, 0, 0);
    if (member) h2 = dg_hi16(&DG_NV(g, s));
    int c0 = 0;
    if (member) c0 = (int)DG_PW(g, DG_H2_OUTOFF(h2) + 1);     // count of its single out edge (-> n)
    // victims' in entries flattened onto lanes 0..L-1, victims in order
    int L = 0, e = 0;
    uint32_t src_off = 0;
    {
        qmask mm = vm;
        while (mm) {
            const int ml = __ffs((int)mm) - 1;
            mm &= mm - 1u;
            const int mlen = dq_rl(DG_H_INLEN(h), ml);
            const uint32_t moff = (uint32_t)dq_rl((int)DG_H2_INOFF(h2), ml);
            if (lane >= L && lane < L + mlen) { e = lane - L; src_off = moff; }
            L += mlen;
        }
    }
    if (L > DQ_W) return false;
    const bool fl = lane < L;
    int n1 = -1;
    if (fl) n1 = (int)DG_PW(g, src_off + e);
    uint4 hn1 = make_uint4(0, 0, 0, 0);
    if (fl) hn1 = dg_lo16(&DG_NV(g, n1));
    if (dq_ballot(fl && DG_H_OUTLEN(hn1) > DQ_W)) return false;
    if (g.sh && dq_ballot(fl && ((hn1.y >> 8) & DG_NF_SHARED))) return false;            // a shared out-list: literal path
    const int a_in_len0 = dq_rl(DG_H_INLEN(h), an_lane);

    // ---- nothing has been modified up to here ----
    const int add_cnt = dq_sum_masked(c0, vm);
    const int add_w = dq_sum_masked(DG_H_WEIGHT(h), vm);
    if (lane == an_lane) {
        DG_PW(g, DG_H2_OUTOFF(h2) + 1) = (uint32_t)(c0 + add_cnt);
        DG_NV(g, an).weight = DG_H_WEIGHT(h) + add_w;
    }
    // :193-212 re-point the victims' in edges to the survivor, in order
    uint32_t a_in_off = (uint32_t)dq_rl((int)DG_H2_INOFF(h2), an_lane);
    int a_in_cap = dq_rl(DG_H2_INCAP(h2), an_lane);
    int a_in_len = a_in_len0;
    bool a_dirty = false;
    qmask rem = dq_ballot(fl);
    while (rem) {
        const int f = __ffs((int)rem) - 1;
        const int x = dq_rl(n1, f);
        rem &= ~dq_ballot(fl && n1 == x);
        // out[x]: drop the entries that point at victims, fold their counts into x->an
        const uint4 hx2 = dg_hi16(&DG_NV(g, x));
        const int xlen = dq_rl(DG_H_OUTLEN(hn1), f);
        const uint32_t xoff = DG_H2_OUTOFF(hx2);
        int dst = -1, c = 0;
        if (lane < xlen) { dst = (int)DG_PW(g, xoff + 2 * lane); c = (int)DG_PW(g, xoff + 2 * lane + 1); }
        bool isv = false;
        for (qmask m = vm; m; m &= m - 1u) isv |= (dst == dq_rl(s, __ffs((int)m) - 1));
        const qmask vmask = dq_ballot(lane < xlen && isv);
        const int csum = dq_sum_masked(c, vmask);
        const qmask apos = dq_ballot(lane < xlen && dst == an);
        const bool keep = lane < xlen && !isv;
        const qmask km = dq_ballot(keep);
        int nlen = __popc(km);
        if (keep) {
            const int idx = __popc(km & DQ_LT(lane));
            DG_PW(g, xoff + 2 * idx) = (uint32_t)dst;
            DG_PW(g, xoff + 2 * idx + 1) = (uint32_t)(((apos >> lane) & 1u) ? c + csum : c);
        }
        if (!apos) {
            // new edge x->an: END of out[x] (room is there: at least one entry was dropped)
            if (lane == 0) { DG_PW(g, xoff + 2 * nlen) = (uint32_t)an; DG_PW(g, xoff + 2 * nlen + 1) = (uint32_t)csum; }
            nlen++;
            // ... and END of in[an]
            if (a_in_len + 1 > a_in_cap) {
                uint32_t ncap = 2u * (uint32_t)(a_in_len + 1);
                if (ncap < 4) ncap = 4;
                const uint32_t noff = dq_alloc(g, ncap, lane);
                if (noff == 0xFFFFFFFFu) return true;
                for (int i = lane; i < a_in_len; i += DQ_W) DG_PW(g, noff + i) = DG_PW(g, a_in_off + i);
                a_in_off = noff; a_in_cap = (int)ncap;
            }
            if (lane == 0) DG_PW(g, a_in_off + a_in_len) = (uint32_t)x;
            a_in_len++;
            a_dirty = true;
        }
        if (lane == 0) DG_NV(g, x).out_len = (uint16_t)nlen;
    }
    if (a_dirty && lane == 0) {
        DgNode *a = &DG_NV(g, an);
        a->in_len = (uint16_t)a_in_len; a->in_off = a_in_off; a->in_cap = (uint16_t)a_in_cap;
    }
    // in[n] without the victims (stable)
    {
        const bool keep = valid_in && !victim;
        const qmask km = dq_ballot(keep);
        if (keep) DG_PW(g, n_in_off + __popc(km & DQ_LT(lane))) = (uint32_t)s;
        if (lane == 0) DG_NV(g, n).in_len = (uint16_t)__popc(km);
    }
    if (victim) {
        DgNode *vn = &DG_NV(g, s);
        vn->out_len = 0; vn->in_len = 0; vn->flags |= DG_NF_DELETED;
    }
    return true;
}

#define DQ_IN_STACK 48
// what a row is doing (dq_merge_segment)
#define DQ_ST_RUN 0                  // visits that merge nothing, one after the other
#define DQ_ST_NEED 1                 // its visit has a merge group or a long list: the generic code
#define DQ_ST_END 2                  // the segment is done (or has failed)

// One segment [c_start, c_end] of target t, swept by the calling ROW (dg_merge_segment, mode DG_MM_WORKER).
// c_end = 0x7fffffff: the segment runs to the exit vertex.
// GC: a segment of k_cuts2 (partial-span pileups, p.gcuts): enter's out-list, exit's in-list and the out-lists of the few
// vertices flagged DG_NF_SHARED are touched by every segment's worker (the protocol at DgGraph::sh) -- a visit next to one
// of them goes the reference-literal way, where the protocol lives; exit is nobody's to enqueue; the segment that starts
// at enter goes on from the queue k_merge_pro left.  me / wlo: the entry's index in the worklist, its target's first.
template <bool GC>
__device__ __forceinline__ void dq_merge_segment(const DgParams &p, const uint32_t t, const int c_start, const int c_end,
                                                 int32_t *stk_base, int *s_stk, int4 *s_ring, const uint32_t me = 0,
                                                 const uint32_t wlo = 0) {
    const int lane = threadIdx.x & (DQ_W - 1);
    const uint64_t nb = p.node_base[t];
    const uint32_t NT = p.n_nodes[t];
    const bool has_end = c_end != 0x7fffffff;
    const int c_hi = has_end ? c_end : (int)NT - 1;
    DgGraph g;
    const bool own_q = GC && c_start == 0;                 // (see DgParams::queue0)
    g.nd = p.nodes + nb; g.queue = own_q ? p.queue0 + nb : p.queue + nb + c_start;
    g.pool = p.pool + p.pool_base[t]; g.pool_size = p.pool_size[t]; g.pool_top = p.pool_top + t;
    g.stk = stk_base; g.stk_words = p.stk_words;
    g.st = p.st; g.tfail = p.tfail + t; g.t = t; g.err = false;
    g.sh = 0; g.X = -1; g.sh_tab = nullptr; g.seg = 0; g.lg_cap = 0; g.lg_cnt = nullptr;
    if (GC) {
        g.sh = 1; g.X = (int)NT - 1;
        g.sh_tab = p.sh_cnt + (uint64_t)t * (2u + 2u * DG_SH_MAX); g.seg = me - wlo;
        g.lg_cap = p.sh_log; g.lg_cnt = p.seg_done + (uint64_t)me * (DG_SH_MAX + 1u);
    }
    const int X = GC ? (int)NT - 1 : -1;                   // exit, where it must be left alone
    const uint32_t N = own_q ? NT : (uint32_t)(c_hi - c_start + 1);     // vertices this worker can dequeue
    uint32_t qh = 0, qt = 1;
    // (a ring entry carries what the visit needs of its vertex's record when the one who queued the vertex had it at hand
    // -- lens = -1: not so.  The record of a queued vertex does not change before its visit: whoever could touch its lists
    // is one of its predecessors, or has one of them for a predecessor, and those have all been visited)
    if (own_q) {
        // the prologue has visited enter (and what hangs on it alone): go on from its queue
        qh = p.pro_state[4u * t]; qt = p.pro_state[4u * t + 1u];
        // (the ring holds the youngest DQ_RING entries)
        for (uint32_t i = (qt - qh > DQ_RING ? qt - DQ_RING : qh) + (uint32_t)lane; i < qt; i += DQ_W) s_ring[i & (DQ_RING - 1)] = make_int4(g.queue[i], -1, 0, 0);
    } else if (lane == 0) { g.queue[0] = c_start; s_ring[0] = make_int4(c_start, -1, 0, 0); }
    DQ_FENCE();
    // Two phases per round, so that the rows of a wave spend their time on the same code: (A) every row runs
    // through the visits that merge nothing (one look, the FIFO bookkeeping) until it meets a visit that has a merge
    // group, or a list longer than half a row, to deal with -- rows that have met theirs wait; (B) those visits, by
    // the generic code, all rows at once.
    // (What steers a row lives in a vector register, `st`, and the loop of phase A has no way out but its condition:
    // a boolean that differs from row to row is a lane mask to the compiler, and every such mask that lives across a
    // branch costs three scalar instructions at every join behind it -- with five of them and breaks from four levels
    // deep, 115 of the 265 instructions of a visit that merges nothing were mask bookkeeping.)
    int st = DQ_ST_RUN;
    int u = -1;
    for (;;) {
        while (st == DQ_ST_RUN && qh < qt) {
            int4 qe = make_int4(0, -1, 0, 0);
            if (qt - qh <= DQ_RING) qe = s_ring[qh & (DQ_RING - 1)];
            else qe.x = g.queue[qh];
            u = qe.x;
            qh++;
            const bool in_only = u == c_end;               // the next segment's worker does the rest of that visit
            if (u < c_start || u > c_hi || (in_only && qh != qt)) {       // cannot happen (see k_cuts)
                dq_fail(g, DG_E_INTERNAL, lane);
                st = DQ_ST_END;
            } else {
                const bool skip_in = c_start != 0 && u == c_start;        // the previous segment's worker merges in[u]
                // (k_merge asks for the next vertex's record before this visit's stores go out; here the 8 registers that
                // takes cost more, in waves per SIMD, than the round trip: 17.3 ms with, 16.1 without at configs[1])
                // u's lens and list offsets: from the ring (the visit that queued u had just read them), else from its record
                uint32_t u_lens = (uint32_t)qe.y, u_out = (uint32_t)qe.z, u_inn = (uint32_t)qe.w;
                if (qe.y == -1) {
                    const uint4 ul = dg_lo16(&DG_NV(g, u)), uh = dg_hi16(&DG_NV(g, u));
                    u_lens = ul.x; u_out = DG_H2_OUTOFF(uh); u_inn = DG_H2_INOFF(uh);
                }
                const int eff_in = skip_in ? 0 : (int)(u_lens >> 16), eff_out = in_only ? 0 : (int)(u_lens & 0xffffu);
                if (eff_in > DQ_W / 2 || eff_out > DQ_W / 2) st = DQ_ST_NEED;
                else {
                    const bool is_in = lane < DQ_W / 2;
                    const int idx = lane & (DQ_W / 2 - 1);
                    const bool valid = is_in ? idx < eff_in : idx < eff_out;
                    const uint32_t ea = is_in ? u_inn + (uint32_t)idx : u_out + 2u * (uint32_t)idx;
                    int nbr = 0;
                    if (valid) nbr = (int)DG_PW(g, ea);
                    uint4 h = make_uint4(0, 0, 0, 0);
                    uint2 ho = make_uint2(0, 0);                  // out_off, in_off of an out-neighbour: for the ring, should this visit queue it
                    if (valid) h = dg_lo16(&DG_NV(g, nbr));
                    if (valid && !is_in) ho = *(reinterpret_cast<const uint2 *>(&DG_NV(g, nbr)) + 2);
                    // in lanes: out_len == 1 (low half of h.x), out lanes: in_len == 1 (high half)
                    const qmask cand = dq_ballot(valid && ((h.x >> (is_in ? 0u : 16u)) & 0xffffu) == 1u);
                    // a merge group = two candidates of one side with the same base
                    int work = 0;
                    // (GC: next to enter / exit / a shared vertex the visit goes the literal way, where their lists are
                    // handled by the protocol; '^' and '$' are nobody else's base, so none of them is ever a group's member)
                    if (GC && dq_ballot(valid && ((h.y >> 8) & DG_NF_SHARED))) work = 2;
                    else if (__popc(cand & DQ_LO) >= 2 || __popc(cand & (DQ_ALL & ~DQ_LO)) >= 2) {
                        const int key = ((cand >> lane) & 1u) ? (DG_H_BASE(h) | (is_in ? 0 : 256)) : -1 - lane;
                        qmask m = cand;
                        while (m) {
                            const int kf = dq_rl(key, __ffs((int)m) - 1);
                            const qmask same = dq_ballot(key == kf);
                            if (__popc(same) >= 2) { work = 2; m = 0; }
                            else m &= ~same;
                        }
                    }
                    if (work) st = DQ_ST_NEED;
                    else if (in_only) st = DQ_ST_END;      // nothing to merge in front of the cut: segment done
                    else {
                        // AlnGraphBoost.cpp:143-158
                        const bool live = valid && !is_in;
                        const int pend = DG_H_PEND(h) - 1;
                        const qmask rm = dq_ballot(live && pend == 0);
                        if (live) DG_NV(g, nbr).pending = pend;
                        if (live && pend == 0) {
                            const uint32_t pos = qt + (uint32_t)__popc(rm & DQ_LT(lane));
                            if (pos < N) { g.queue[pos] = nbr; s_ring[pos & (DQ_RING - 1)] = make_int4(nbr, (int)h.x, (int)ho.x, (int)ho.y); }
                        }
                        qt += (uint32_t)__popc(rm);
                        if (qt > N) { dq_fail(g, DG_E_INTERNAL, lane); st = DQ_ST_END; }
                    }
                }
            }
        }
        if (st != DQ_ST_NEED) break;                       // the queue ran dry, the cut was reached, or something failed
        const bool skip_in = c_start != 0 && u == c_start, in_only = u == c_end;
        bool scalar = false;
        if (GC) {
            // is u a neighbour of enter, exit or a shared vertex?  (whatever the lengths of its lists)
            const uint4 ql = dg_lo16(&DG_NV(g, u)), qhh = dg_hi16(&DG_NV(g, u));
            bool f = false;
            if (!skip_in) for (int i = lane; i < DG_H_INLEN(ql); i += DQ_W) f |= ((dg_lo16(&DG_NV(g, (int)DG_PW(g, DG_H2_INOFF(qhh) + i))).y >> 8) & DG_NF_SHARED) != 0;
            if (!in_only) for (int i = lane; i < DG_H_OUTLEN(ql); i += DQ_W) f |= (int)DG_PW(g, DG_H2_OUTOFF(qhh) + 2 * i) == X;
            if (dq_ballot(f)) scalar = true;              // the literal path handles their lists
        }

        // ---------------- mergeInNodes(u), recursion on an explicit stack ----------------
        int sp = skip_in ? 0 : 1;
        int fr_n = u, fr_last = -1;                       // top frame lives in registers
        while (sp > 0 && !scalar) {
            const uint4 nl = dg_lo16(&DG_NV(g, fr_n)), nh = dg_hi16(&DG_NV(g, fr_n));
            if (DG_H_INLEN(nl) > DQ_W) { scalar = true; break; }
            const bool valid = lane < DG_H_INLEN(nl);
            int s = 0;
            if (valid) s = (int)DG_PW(g, DG_H2_INOFF(nh) + lane);
            uint4 h = make_uint4(0, 0, 0, 0);
            if (valid) h = dg_lo16(&DG_NV(g, s));
            const qmask cand = dq_ballot(valid && DG_H_OUTLEN(h) == 1 && !(GC && ((h.y >> 8) & DG_NF_SHARED)));
            qmask M = 0;
            int b = 256;
            if (__popc(cand) >= 2) b = dq_pick_group(cand, DG_H_BASE(h), fr_last, lane, &M);
            if (b == 256) {                               // frame done: pop
                sp--;
                if (sp > 0) { fr_n = s_stk[2 * (sp - 1)]; fr_last = s_stk[2 * (sp - 1) + 1]; }
                continue;
            }
            if (sp >= DQ_IN_STACK) { scalar = true; break; }
            int an = -1;
            if (!dq_merge_in_group(g, fr_n, DG_H2_INOFF(nh), M, s, h, valid, lane, &an)) { scalar = true; break; }
            if (g.err) break;
            fr_last = b;
            if (lane == 0) { s_stk[2 * (sp - 1)] = fr_n; s_stk[2 * (sp - 1) + 1] = fr_last; }
            sp++;                                         // :213 recurse on the survivor
            fr_n = an; fr_last = -1;
        }
        if (scalar && !g.err && sp > 0) {
            // finish every open frame, deepest first, on the reference-literal path; groups already merged are gone,
            // so re-evaluating a frame from scratch is exact
            DQ_LANE0(g, {
                dgg_merge_in(gs, fr_n);
                for (int f = sp - 2; f >= 0 && !gs.err; f--) dgg_merge_in(gs, s_stk[2 * f]);
            });
        }
        if (in_only) break;                               // the next segment's worker does the rest of this visit
        // ---------------- mergeOutNodes(u) + FIFO bookkeeping ----------------
        bool done = false;
        if (scalar) {
            if (!g.err) DQ_LANE0(g, { dgg_merge_out(gs, u); });
        }
        int last_out = -1;
        while (!done && !g.err) {
            const uint4 ul = dg_lo16(&DG_NV(g, u)), uh = dg_hi16(&DG_NV(g, u));
            const int out_len = DG_H_OUTLEN(ul);
            if (out_len > DQ_W) break;                    // bookkeeping by the single-lane loop below
            const bool valid = lane < out_len;
            int d = 0, cnt = 0;
            if (valid) { d = (int)DG_PW(g, DG_H2_OUTOFF(uh) + 2 * lane); cnt = (int)DG_PW(g, DG_H2_OUTOFF(uh) + 2 * lane + 1); }
            uint4 h = make_uint4(0, 0, 0, 0);
            if (valid) h = dg_lo16(&DG_NV(g, d));
            if (!scalar) {
                const qmask cand = dq_ballot(valid && DG_H_INLEN(h) == 1 && d != X);
                qmask M = 0;
                int b = 256;
                if (__popc(cand) >= 2) b = dq_pick_group(cand, DG_H_BASE(h), last_out, lane, &M);
                if (b != 256) {
                    if (dq_merge_out_group(g, u, DG_H2_OUTOFF(uh), M, d, cnt, h, valid, DQ_ALL, lane)) {
                        last_out = b;
                        continue;                         // re-read u's list, look for the next group
                    }
                    DQ_LANE0(g, { dgg_merge_out(gs, u); });   // a list longer than a row: literal path
                    scalar = true;
                    continue;
                }
            }
            // AlnGraphBoost.cpp:143-158
            const int pend = DG_H_PEND(h) - 1;
            const bool bk = valid && d != X;              // (the exit vertex is k_merge_fin's)
            if (bk) DG_NV(g, d).pending = pend;
            const qmask rm = dq_ballot(bk && pend == 0);
            if (bk && pend == 0) {
                const uint32_t pos = qt + (uint32_t)__popc(rm & DQ_LT(lane));
                if (pos < N) { g.queue[pos] = d; s_ring[pos & (DQ_RING - 1)] = make_int4(d, -1, 0, 0); }
            }
            qt += (uint32_t)__popc(rm);
            if (qt > N) dq_fail(g, DG_E_INTERNAL, lane);
            done = true;
        }
        if (!done && !g.err) {
            // out list longer than a row
            uint32_t nqt = qt;
            DQ_LANE0(g, {
                if (!scalar) dgg_merge_out(gs, u);
                const uint32_t off = gs.nd[u].out_off;
                const int len = gs.nd[u].out_len;
                for (int i = 0; i < len && !gs.err; i++) {
                    const int v = (int)gs.pool[off + 2 * i];
                    if (v == X) continue;
                    const int pend = gs.nd[v].pending - 1;
                    gs.nd[v].pending = pend;
                    if (pend == 0) {
                        if (nqt >= N) { dgg_fail(gs, DG_E_INTERNAL); break; }
                        s_ring[nqt & (DQ_RING - 1)] = make_int4(v, -1, 0, 0);
                        gs.queue[nqt++] = v;
                    }
                }
            });
            qt = (uint32_t)dq_rl((int)nqt, 0);
        }
        if (g.err) break;
        st = DQ_ST_RUN;
    }
}

// mergeNodes, four (target, segment of p.cuts) pairs per wave: row r of block b sweeps pair 4 b + r.  (Rows that take
// their pairs off a ticket counter, one after the other, were 2 - 3 ms slower at configs[1] than this grid with the
// number of pieces chosen so that the waves fill the chip a whole number of times: dagcon_upload.)
#ifndef DQ_WAVES
#define DQ_WAVES 6
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DQ_WAVES, DQ_WAVES))) void k_merge_q(DgParams p) {
    __shared__ int s_stk[DQ_ROWS][2 * DQ_IN_STACK];
    __shared__ int4 s_ring[DQ_ROWS][DQ_RING];
    const uint32_t row = threadIdx.x / DQ_W;
    const uint32_t pair = blockIdx.x * DQ_ROWS + row;
    const uint32_t t = pair / p.seg_max, seg = pair % p.seg_max;
    if (t >= p.T) return;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    // a row holds 8 + 8 list entries: a target far deeper than that is swept by k_merge (a wave per segment), launched
    // beside this kernel when the batch has such targets (DgParams::q_kmax; 0: every target is taken here)
    if (p.q_kmax && (uint32_t)(p.aln_begin[t + 1] - p.aln_begin[t]) > p.q_kmax) return;
    const uint32_t *crow = p.cuts + (uint64_t)t * (p.seg_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg >= nseg) return;
    const int c_start = (int)crow[1 + seg];
    const int c_end = seg + 1 < nseg ? (int)crow[2 + seg] : 0x7fffffff;
    dq_merge_segment<false>(p, t, c_start, c_end, p.stk + (uint64_t)pair * p.stk_words, s_stk[row], s_ring[row]);
}

#ifdef DG_EXPERIMENTS
// (make experiments; DAGCON_MERGE_LIST_Q=1.  Exact on the whole suite, and no faster than a wave per entry: config-5 shape,
// 400 / 1,000 / 2,000 targets: merge 22.0 / 33.0 / 56.4 ms against k_merge_list's 13.4 / 29.1 / 55.7 -- see DESIGN.md)
// The worklist of k_cuts2 (partial-span pileups; see k_merge_list), eight entries per wave: a wave takes eight consecutive
// entries by ticket, row r the r-th of them, until the list is done.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DQ_WAVES, DQ_WAVES))) void k_merge_list_q(DgParams p) {
    __shared__ int s_stk[DQ_ROWS][2 * DQ_IN_STACK];
    __shared__ int4 s_ring[DQ_ROWS][DQ_RING];
    if (dg_failed(p)) return;
    const uint32_t row = threadIdx.x / DQ_W;
    const uint32_t n = p.tile_list[0] < p.tile_list_cap ? p.tile_list[0] : p.tile_list_cap;
    for (;;) {
        uint32_t i0 = 0;
        if (threadIdx.x == 0) i0 = atomicAdd(&p.tile_list[2], (uint32_t)DQ_ROWS);
        i0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i0);
        if (i0 >= n) break;
        const uint32_t i = i0 + row;
        if (i < n) {
            const uint32_t t = p.tile_list[4 + 3 * i];
            const int c_start = (int)p.tile_list[5 + 3 * i];
            const uint32_t ce = p.tile_list[6 + 3 * i];
            if (!dg_tskip(p, t)) {
                dq_merge_segment<true>(p, t, c_start, ce == DG_NOSEG_END ? 0x7fffffff : (int)ce, p.stk + (uint64_t)(blockIdx.x * DQ_ROWS + row) * p.stk_words,
                                       s_stk[row], s_ring[row], i, p.wl_first[t]);
                if ((threadIdx.x & (DQ_W - 1)) == 0) atomicAdd(&p.st->n_mseg, 1u);
            }
        }
        DQ_FENCE();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.tile_list[0] > p.tile_list_cap) dg_fail(p, DG_E_LIST_OVF);
}

#endif
