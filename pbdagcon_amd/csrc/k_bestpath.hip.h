// k_bestpath.hip.h -- stage (c): bestPath (AlnGraphBoost.cpp:375-459) and the
// consensus segmentation (AlnGraphBoost.cpp:327-373).
//
// The DP itself is order-independent: score[n] = max over out_edges(n), IN LIST
// ORDER with strict '>' (first maximum wins, :411), of
//     score[t] - 10                          if t.backbone && t.weight == 1
//     count(e) - coverage[bbMap[t]]*0.5 + score[t]   otherwise          (:404-409)
// so any reverse-topological order gives the reference's scores.
//
//   k_bp_prepare  thread per vertex: the per-target term of the edge score
//                 (pen = coverage[bbMap[t]]*0.5, or the "-10" marker), score = 0
//                 (std::map default), pending = out-degree
//   k_bestpath    one wave per target.  Vertex ids are in backbone-position order,
//                 which is a topological order of the graph except for the few edges
//                 the merge turned around.  The wave therefore STREAMS the vertices
//                 from the exit downwards: 64 records at a time are loaded two chunks
//                 ahead (plain loads whose latency nobody waits for), unpacked into an
//                 LDS ring of the last 256 vertices (score, pen, pending, the adjacency
//                 entries), and every step works on LDS only: the successors' (score,
//                 pen) come from the ring, the pending counters of the predecessors are
//                 decremented in the ring.  A vertex whose pending counter is not zero
//                 when the stream reaches it (a turned-around edge) waits and is taken
//                 up the moment its last successor is scored (Kahn's rule), so the
//                 order is always reverse-topological.  Whatever does not fit the ring
//                 (long lists, far-away neighbours) goes through HBM.
//                 Then the best-edge walk and the segmentation.
//
// fp32 throughout; every value is a multiple of 0.5 below 2^23, so the
// arithmetic is exact and the expression order of the reference is kept
// (-ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "dagcon_dev.h"

#define DG_PEN_BACKBONE_ONLY (-1.0f)    // real penalties are >= 0

__global__ __launch_bounds__(256) void k_bp_prepare(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || !p.tactive[t]) return;
    const uint64_t nb = p.node_base[t];
    const uint32_t N = p.n_nodes[t];
    const int32_t *cov = p.cov + p.bbv_base[t];
    for (uint32_t v = blockIdx.y * 256 + threadIdx.x; v < N; v += gridDim.y * 256) {
        DgNode *n = &p.nodes[nb + v];
        const uint4 h = *reinterpret_cast<const uint4 *>(n);
        const uint4 h2 = *(reinterpret_cast<const uint4 *>(n) + 1);
        const uint32_t flags = (h.y >> 8) & 0xffu;
        float pen;
        if ((flags & DG_NF_BACKBONE) && (int)h.z == 1) pen = DG_PEN_BACKBONE_ONLY;   // :404
        else pen = (float)cov[(int)h2.w] * 0.5f;                                      // :407-408
        n->pending = (int)(h.x & 0xffffu);         // out-edges not yet visited (:423-439)
        p.score[nb + v] = make_float2(0.0f, pen);  // std::map<VtxDesc,float>: absent key reads as 0
        p.best[nb + v] = -1;
    }
}

#define DG_BR 256            // ring slots (vertex id & 255)
#define DG_BOUT 6            // out entries kept in a slot
#define DG_BIN 8             // in entries kept in a slot
#define DG_BDEF 64           // waiting vertices that became ready
#define DG_BL_HBM 0x80000000 // lens: the lists did not fit the slot, read them from HBM

struct DgBpShared {
    int tag[DG_BR];
    float score[DG_BR];
    float pen[DG_BR];
    int pend[DG_BR];
    int lens[DG_BR];                     // out_len | in_len << 8 | flags << 16 | DG_BL_HBM
    int out_dst[DG_BR * DG_BOUT];
    unsigned short out_cnt[DG_BR * DG_BOUT];
    int in_src[DG_BR * DG_BIN];
    int delta[DG_BR];                    // decrements for vertices that are not in the ring yet
    int defer[DG_BDEF];
};

// one predecessor loses an unvisited out-edge (AlnGraphBoost.cpp:423-439); returns true when
// it became ready although the stream has already passed it
__device__ __forceinline__ bool dg_bp_release(DgBpShared &S, DgNode *nd, int s, int stream_pos) {
    const int x = s & (DG_BR - 1);
    const int tg = S.tag[x], pv = S.pend[x];               // one LDS trip for both
    if (tg == s) {
        const int pnd = pv - 1;
        S.pend[x] = pnd;
        return pnd == 0 && s > stream_pos;
    }
    if (s <= stream_pos && s > stream_pos - DG_BR) {       // about to be staged: remember
        S.delta[x] += 1;
        return false;
    }
    const int pnd = nd[s].pending - 1;                      // far away: HBM
    nd[s].pending = pnd;
    return pnd == 0 && s > stream_pos;
}

__global__ __launch_bounds__(64) void k_bestpath(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || !p.tactive[t]) return;
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    __shared__ DgBpShared S;
    __shared__ uint32_t s_len, s_nseg;
    DgNode *nd = p.nodes + nb;
    int32_t *best = p.best + nb;
    float2 *score = p.score + nb;
    const uint32_t *pool = p.pool + p.pool_base[t];
    const int N = (int)p.n_nodes[t];
    const int exitv = N - 1;

    for (int i = lane; i < DG_BR; i += 64) { S.tag[i] = -1; S.delta[i] = 0; S.pend[i] = 0; }

    // ---- staging registers: r_* = records of a chunk, e_* = its list entries ----
    uint4 r_lo, r_hi, n_lo, n_hi;
    float r_pen, n_pen;
    int e_out[DG_BOUT], e_cnt[DG_BOUT], e_in[DG_BIN];
    // chunk c holds ids [N-1-64c-63, N-1-64c]; lane l -> id N-1-64c-l
#define DG_LOAD_REC(C, LO, HI, PEN)                                                        \
    do {                                                                                    \
        const int v_ = N - 1 - 64 * (C) - lane;                                             \
        LO = make_uint4(0, 0, 0, 0); HI = make_uint4(0, 0, 0, 0); PEN = 0.0f;               \
        if (v_ >= 0) {                                                                      \
            LO = *reinterpret_cast<const uint4 *>(&nd[v_]);                                 \
            HI = *(reinterpret_cast<const uint4 *>(&nd[v_]) + 1);                           \
            PEN = score[v_].y;                                                              \
        }                                                                                   \
    } while (0)
#define DG_LOAD_ENT(LO, HI)                                                                 \
    do {                                                                                    \
        const int ol_ = (int)((LO).x & 0xffffu), il_ = (int)((LO).x >> 16);                 \
        const bool fit_ = ol_ <= DG_BOUT && il_ <= DG_BIN;                                  \
        _Pragma("unroll") for (int k_ = 0; k_ < DG_BOUT; k_++) {                            \
            e_out[k_] = 0; e_cnt[k_] = 0;                                                   \
            if (fit_ && k_ < ol_) { e_out[k_] = (int)pool[(HI).x + 2 * k_]; e_cnt[k_] = (int)pool[(HI).x + 2 * k_ + 1]; } \
        }                                                                                   \
        _Pragma("unroll") for (int k_ = 0; k_ < DG_BIN; k_++) {                             \
            e_in[k_] = 0;                                                                   \
            if (fit_ && k_ < il_) e_in[k_] = (int)pool[(HI).y + k_];                        \
        }                                                                                   \
    } while (0)
    // unpack chunk C (records LO/PEN + entries) into the ring; a slot that still holds a
    // vertex waiting for its successors hands its counter back to HBM
#define DG_WRITE_CHUNK(C, LO, PEN)                                                          \
    do {                                                                                    \
        const int v_ = N - 1 - 64 * (C) - lane;                                             \
        if (v_ >= 0) {                                                                      \
            const int x_ = v_ & (DG_BR - 1);                                                \
            const int old_ = S.tag[x_];                                                     \
            if (old_ >= 0 && S.pend[x_] > 0) nd[old_].pending = S.pend[x_];                 \
            const int ol_ = (int)((LO).x & 0xffffu), il_ = (int)((LO).x >> 16);             \
            const bool fit_ = ol_ <= DG_BOUT && il_ <= DG_BIN;                              \
            S.tag[x_] = v_;                                                                 \
            S.score[x_] = 0.0f;                                                             \
            S.pen[x_] = PEN;                                                                \
            S.pend[x_] = (int)(LO).w - S.delta[x_];                                         \
            S.delta[x_] = 0;                                                                \
            S.lens[x_] = (fit_ ? (ol_ | (il_ << 8)) : (int)DG_BL_HBM) | (int)(((LO).y >> 8) & 0xffu) << 16; \
            if (fit_) {                                                                     \
                _Pragma("unroll") for (int k_ = 0; k_ < DG_BOUT; k_++) {                    \
                    S.out_dst[x_ * DG_BOUT + k_] = e_out[k_];                               \
                    S.out_cnt[x_ * DG_BOUT + k_] = (unsigned short)e_cnt[k_];               \
                }                                                                           \
                _Pragma("unroll") for (int k_ = 0; k_ < DG_BIN; k_++) S.in_src[x_ * DG_BIN + k_] = e_in[k_]; \
            }                                                                               \
        }                                                                                   \
    } while (0)

    const int n_chunks = (N + 63) / 64;
    // prologue: chunk 0 into the ring; chunk 1 records + entries, chunk 2 records in flight
    DG_LOAD_REC(0, r_lo, r_hi, r_pen);
    DG_LOAD_ENT(r_lo, r_hi);
    DG_WRITE_CHUNK(0, r_lo, r_pen);
    DG_LOAD_REC(1, r_lo, r_hi, r_pen);
    DG_LOAD_ENT(r_lo, r_hi);
    DG_LOAD_REC(2, n_lo, n_hi, n_pen);

    int n_defer = 0;
    bool bad = false;
    for (int c = 0; c < n_chunks && !bad; c++) {
        if (c > 0) {
            // chunk c: its records and entries were requested a whole chunk ago
            DG_WRITE_CHUNK(c, r_lo, r_pen);
            r_lo = n_lo; r_hi = n_hi; r_pen = n_pen;
            DG_LOAD_ENT(r_lo, r_hi);                       // chunk c+1
            DG_LOAD_REC(c + 2, n_lo, n_hi, n_pen);         // chunk c+2
        }
        const int v_hi = N - 1 - 64 * c;
        const int v_lo = v_hi - 63 < 0 ? 0 : v_hi - 63;
        int v = v_hi;
        while (v >= v_lo || n_defer > 0) {
            int n, from_defer;
            if (n_defer > 0) { n = S.defer[--n_defer]; from_defer = 1; }
            else { n = v--; from_defer = 0; }
            n = __builtin_amdgcn_readfirstlane(n);
            const int stream_pos = v;                     // ids > v have had their turn
            const int x = n & (DG_BR - 1);
            // tag, lens and pending in one LDS trip
            const int tg_ = S.tag[x], ln_ = S.lens[x], pd_ = S.pend[x];
            const bool in_ring = __builtin_amdgcn_readfirstlane(tg_) == n;
            int lens, pend_n;
            if (in_ring) {
                lens = __builtin_amdgcn_readfirstlane(ln_);
                pend_n = __builtin_amdgcn_readfirstlane(pd_);
            } else {
                // a waiting vertex that fell out of the ring: everything from HBM
                const uint4 lo = *reinterpret_cast<const uint4 *>(&nd[n]);
                lens = __builtin_amdgcn_readfirstlane((int)DG_BL_HBM | (int)((lo.y >> 8) & 0xffu) << 16);
                pend_n = __builtin_amdgcn_readfirstlane((int)lo.w);
            }
            if ((lens >> 16) & DG_NF_DELETED) continue;
            if (!from_defer && pend_n != 0) continue;     // waits for a turned-around edge
            int out_len = lens & 0xff, in_len = (lens >> 8) & 0xff;
            const bool hbm = lens < 0;
            uint32_t out_off = 0, in_off = 0;
            if (hbm) {
                const DgNode nn = nd[n];
                out_off = __builtin_amdgcn_readfirstlane(nn.out_off);
                in_off = __builtin_amdgcn_readfirstlane(nn.in_off);
                out_len = __builtin_amdgcn_readfirstlane((int)nn.out_len);
                in_len = __builtin_amdgcn_readfirstlane((int)nn.in_len);
            }
            if (out_len <= 32 && in_len <= 32) {
                const bool is_out = lane < 32;
                const int idx = lane & 31;
                const bool valid = is_out ? idx < out_len : idx < in_len;
                int nbr = 0, cnt = 0;
                if (valid) {
                    if (!hbm) {
                        if (is_out) { nbr = S.out_dst[x * DG_BOUT + idx]; cnt = S.out_cnt[x * DG_BOUT + idx]; }
                        else nbr = S.in_src[x * DG_BIN + idx];
                    } else {
                        if (is_out) { nbr = (int)pool[out_off + 2 * idx]; cnt = (int)pool[out_off + 2 * idx + 1]; }
                        else nbr = (int)pool[in_off + idx];
                    }
                }
                float ns = -FLT_MAX;
                if (valid && is_out) {
                    const int y = nbr & (DG_BR - 1);
                    const int tg = S.tag[y];
                    float sc = S.score[y], pn = S.pen[y];           // one LDS trip for all three
                    if (tg != nbr) { const float2 sp = score[nbr]; sc = sp.x; pn = sp.y; }
                    if (pn == DG_PEN_BACKBONE_ONLY) ns = sc - 10.0f;
                    else ns = (float)cnt - pn + sc;
                }
                // :399-416 first maximum in list order: the out entries sit on lanes
                // 0..out_len-1 in list order; a scalar walk with strict '>' is the reference loop
                if (out_len > 0) {
                    float mx = -FLT_MAX;
                    int bd = -1;
                    for (int i = 0; i < out_len; i++) {
                        const float xs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ns), i));
                        if (xs > mx) { mx = xs; bd = __builtin_amdgcn_readlane(nbr, i); }
                    }
                    if (lane == 0 && bd >= 0) {
                        if (in_ring) S.score[x] = mx;
                        score[n].x = mx;
                        best[n] = bd;
                    }
                }
                bool rdy = false;
                if (valid && !is_out) rdy = dg_bp_release(S, nd, nbr, stream_pos);
                const unsigned long long rm = __ballot(rdy);
                if (rm) {
                    if (n_defer + __popcll(rm) > DG_BDEF) { bad = true; break; }
                    if (rdy) S.defer[n_defer + __popcll(rm & ((1ull << lane) - 1ull))] = nbr;
                    n_defer += __popcll(rm);
                }
            } else {
                // a list longer than half a wave: literal loops on lane 0
                int nd_new = n_defer;
                if (lane == 0) {
                    float bs = -FLT_MAX;
                    int bd = -1;
                    for (int i = 0; i < out_len; i++) {
                        const int d = (int)pool[out_off + 2 * i];
                        const int cc = (int)pool[out_off + 2 * i + 1];
                        const int y = d & (DG_BR - 1);
                        float sc, pn;
                        if (S.tag[y] == d) { sc = S.score[y]; pn = S.pen[y]; }
                        else { const float2 sp = score[d]; sc = sp.x; pn = sp.y; }
                        const float nsx = pn == DG_PEN_BACKBONE_ONLY ? sc - 10.0f : (float)cc - pn + sc;
                        if (nsx > bs) { bs = nsx; bd = d; }
                    }
                    if (bd >= 0) { if (in_ring) S.score[x] = bs; score[n].x = bs; best[n] = bd; }
                    for (int i = 0; i < in_len; i++) {
                        const int s = (int)pool[in_off + i];
                        if (dg_bp_release(S, nd, s, stream_pos)) {
                            if (nd_new < DG_BDEF) S.defer[nd_new] = s;
                            nd_new++;
                        }
                    }
                }
                n_defer = __builtin_amdgcn_readlane(nd_new, 0);
                if (n_defer > DG_BDEF) { bad = true; break; }
            }
        }
    }
#undef DG_LOAD_REC
#undef DG_LOAD_ENT
#undef DG_WRITE_CHUNK
    if (bad) { if (lane == 0) dg_fail(p, DG_E_INTERNAL); return; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

    // :443-456 walk the best edges from enter; :327-373 segmentation.  The walk
    // is a pointer chase; lane 0 does it and keeps the consensus in cns_tmp.
    uint8_t *tmp = p.cns_tmp + nb;
    int32_t *segs = p.stk + (uint64_t)t * p.stk_words;      // (range0, range1) pairs
    if (lane == 0) {
        const uint8_t eb = nd[0].base, xb = nd[exitv].base;
        const int minw = p.min_weight;
        const uint32_t minlen = p.min_len;
        const uint32_t seg_cap = p.stk_words / 2;
        int v = 0;
        int offs = 0, idx = 0;
        bool met = false;
        uint32_t nseg = 0, steps = 0;
        bool ovf = false;
        for (;;) {
            const uint4 h = *reinterpret_cast<const uint4 *>(&nd[v]);
            const int nxt = best[v];
            const uint8_t base = (uint8_t)(h.y & 0xffu);
            if (!(base == eb || base == xb)) {
                tmp[idx] = base;
                const int w = (int)h.z;
                if (!met && w >= minw) { offs = idx; met = true; }
                else if (met && w < minw) {
                    met = false;
                    if ((uint32_t)(idx - offs) >= minlen) {
                        if (nseg < seg_cap) { segs[2 * nseg] = offs; segs[2 * nseg + 1] = idx; nseg++; }
                        else ovf = true;
                    }
                }
                idx++;
            }
            if (nxt < 0) break;
            v = nxt;
            if (++steps > (uint32_t)N) { ovf = true; break; }
        }
        if (met && (uint32_t)(idx - offs) >= minlen) {
            if (nseg < seg_cap) { segs[2 * nseg] = offs; segs[2 * nseg + 1] = idx; nseg++; }
            else ovf = true;
        }
        if (ovf) dg_fail(p, DG_E_STACK);
        // only the bases some segment covers are shipped
        uint32_t keep = nseg ? (uint32_t)segs[2 * (nseg - 1) + 1] : 0u;
        const unsigned long long co = atomicAdd(&p.st->cns_top, (unsigned long long)keep);
        const unsigned long long so = atomicAdd(&p.st->seg_top, (unsigned long long)nseg);
        if (co + keep > p.cns_cap || so + nseg > p.seg_cap) { dg_fail(p, DG_E_OUT_OVF); keep = 0; nseg = 0; }
        p.cns_off[t] = co; p.cns_len[t] = keep;
        p.seg_first[t] = so; p.n_seg[t] = nseg;
        s_len = keep; s_nseg = nseg;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const uint32_t keep = s_len, nseg = s_nseg;
    uint8_t *out = p.cns + p.cns_off[t];
    for (uint32_t i = lane; i < keep; i += 64) out[i] = tmp[i];
    const uint64_t so = p.seg_first[t];
    for (uint32_t i = lane; i < nseg; i += 64) {
        p.seg_r0[so + i] = segs[2 * i];
        p.seg_r1[so + i] = segs[2 * i + 1];
    }
}
