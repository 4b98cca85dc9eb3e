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
//   k_bestpath    wave per target: Kahn sweep from the exit vertex, one vertex per
//                 step, one list entry per lane (out entries on lanes 0-31, in
//                 entries on lanes 32-63): gather (score, pen) of the successors,
//                 wave max with lowest-lane tie-break (= first maximum in list
//                 order), release the predecessors with plain stores (one wave
//                 owns the target, no atomics); then the best-edge walk and the
//                 segmentation.
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

__global__ __launch_bounds__(128) void k_bestpath(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || !p.tactive[t]) return;
    const int lane = threadIdx.x & 63;
    const uint64_t nb = p.node_base[t];
    __shared__ int s_prog;
    __shared__ uint32_t s_len, s_nseg;
    if (threadIdx.x == 0) s_prog = (int)p.n_nodes[t] - 1;
    __syncthreads();
    if (threadIdx.x >= 64) {                             // wave 1: prefetcher, exit -> enter
        dg_prefetch_wave(p.nodes + nb, p.pool + p.pool_base[t], p.pool_size[t], (int)p.n_nodes[t], &s_prog, lane, -1, (int)p.pf_ahead);
        return;
    }
    DgNode *nd = p.nodes + nb;
    int32_t *best = p.best + nb;
    int32_t *queue = p.queue + nb;
    float2 *score = p.score + nb;
    const uint32_t *pool = p.pool + p.pool_base[t];
    const uint32_t N = p.n_nodes[t];
    const int exitv = (int)N - 1;

    if (lane == 0) queue[0] = exitv;
    DG_WAVE_FENCE();
    uint32_t qh = 0, qt = 1;
    bool bad = false;
    int prog = exitv;
    while (qh < qt) {
        const int n = queue[qh++];
        if (n < prog) { prog = n; if (lane == 0) *(volatile int *)&s_prog = n; }
        const DgNode nn = nd[n];
        const int out_len = nn.out_len, in_len = nn.in_len;
        if (out_len <= 32 && in_len <= 32) {
            const bool is_out = lane < 32;
            const int idx = lane & 31;
            const bool valid = is_out ? idx < out_len : idx < in_len;
            int nbr = 0, cnt = 0;
            if (valid) {
                if (is_out) { nbr = (int)pool[nn.out_off + 2 * idx]; cnt = (int)pool[nn.out_off + 2 * idx + 1]; }
                else nbr = (int)pool[nn.in_off + idx];
            }
            float ns = -FLT_MAX;
            int pend = 0;
            if (valid) {
                if (is_out) {
                    const float2 sp = score[nbr];
                    if (sp.y == DG_PEN_BACKBONE_ONLY) ns = sp.x - 10.0f;
                    else ns = (float)cnt - sp.y + sp.x;
                } else {
                    pend = nd[nbr].pending - 1;
                    nd[nbr].pending = pend;
                }
            }
            // :399-416 first maximum in list order: the out entries sit on lanes 0..out_len-1 in
            // list order, so a scalar walk over them with strict '>' is the reference loop
            if (out_len > 0) {
                float mx = -FLT_MAX;
                int bd = -1;
                for (int i = 0; i < out_len; i++) {
                    const float x = __int_as_float(DG_RL(__float_as_int(ns), i));
                    if (x > mx) { mx = x; bd = DG_RL(nbr, i); }
                }
                if (lane == 0 && bd >= 0) { score[n].x = mx; best[n] = bd; }
            }
            const unsigned long long rm = __ballot(valid && !is_out && pend == 0);
            if (valid && !is_out && pend == 0) {
                const uint32_t pos = qt + (uint32_t)__popcll(rm & ((1ull << lane) - 1ull));
                if (pos < N) queue[pos] = nbr;
            }
            qt += (uint32_t)__popcll(rm);
        } else {
            // a list longer than half a wave: literal loops on lane 0
            uint32_t nqt = qt;
            if (lane == 0) {
                float bs = -FLT_MAX;
                int bd = -1;
                for (int i = 0; i < out_len; i++) {
                    const int d = (int)pool[nn.out_off + 2 * i];
                    const int c = (int)pool[nn.out_off + 2 * i + 1];
                    const float2 sp = score[d];
                    const float ns = sp.y == DG_PEN_BACKBONE_ONLY ? sp.x - 10.0f : (float)c - sp.y + sp.x;
                    if (ns > bs) { bs = ns; bd = d; }
                }
                if (bd >= 0) { score[n].x = bs; best[n] = bd; }
                for (int i = 0; i < in_len; i++) {
                    const int s = (int)pool[nn.in_off + i];
                    const int pend = nd[s].pending - 1;
                    nd[s].pending = pend;
                    if (pend == 0 && nqt < N) queue[nqt++] = s;
                }
            }
            DG_WAVE_FENCE();
            qt = (uint32_t)DG_RL(nqt, 0);
        }
        if (qt > N) { bad = true; break; }
    }
    if (lane == 0) *(volatile int *)&s_prog = DG_PROG_DONE;
    if (bad) { if (lane == 0) dg_fail(p, DG_E_INTERNAL); return; }

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
            if (++steps > N) { ovf = true; break; }
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
    DG_WAVE_FENCE();
    const uint32_t keep = s_len, nseg = s_nseg;
    uint8_t *out = p.cns + p.cns_off[t];
    for (uint32_t i = lane; i < keep; i += 64) out[i] = tmp[i];
    const uint64_t so = p.seg_first[t];
    for (uint32_t i = lane; i < nseg; i += 64) {
        p.seg_r0[so + i] = segs[2 * i];
        p.seg_r1[so + i] = segs[2 * i + 1];
    }
}
