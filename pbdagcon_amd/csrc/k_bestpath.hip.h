// k_bestpath.hip.h -- stage (c): bestPath (AlnGraphBoost.cpp:375-459) and the
// consensus segmentation (AlnGraphBoost.cpp:327-373).
//
// The DP itself is order-independent: score[n] = max over out_edges(n), IN LIST
// ORDER with strict '>' (first maximum wins, :411), of
//     score[t] - 10                          if t.backbone && t.weight == 1
//     count(e) - coverage[bbMap[t]]*0.5 + score[t]   otherwise          (:404-409)
// so any reverse-topological order gives the reference's scores.  One wave
// per target sweeps the DAG as a wavefront: every lane takes one vertex whose
// successors are all scored, scores it, then releases its predecessors
// (Kahn's algorithm run 64 vertices at a time).  fp32 throughout; every value
// is a multiple of 0.5 below 2^23, so the arithmetic is exact and the
// expression order of the reference is kept (-ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "dagcon_dev.h"

__global__ __launch_bounds__(64) void k_bestpath(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || !p.tactive[t]) return;
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    DgNode *nd = p.nodes + nb;
    int32_t *best = p.best + nb;
    int32_t *queue = p.queue + nb;
    float *score = p.score + nb;
    const int32_t *cov = p.cov + p.bbv_base[t];
    const uint32_t *pool = p.pool + p.pool_base[t];
    const uint32_t N = p.n_nodes[t];
    const int exitv = (int)N - 1;
    __shared__ uint32_t s_qt;

    for (uint32_t v = lane; v < N; v += 64) {
        nd[v].pending = nd[v].out_len;   // out-edges not yet visited (:423-439)
        best[v] = -1;
        score[v] = 0.0f;              // std::map<VtxDesc,float>: absent key reads as 0
    }
    if (lane == 0) { queue[0] = exitv; s_qt = 1; }
    __syncthreads();
    uint32_t qh = 0, qt = 1;
    while (qh < qt) {
        const uint32_t m = min(64u, qt - qh);
        int n = -1;
        if ((uint32_t)lane < m) {
            n = queue[qh + lane];
            const uint32_t off = nd[n].out_off;
            const int len = nd[n].out_len;
            float bs = -FLT_MAX;
            int bd = -1;
            for (int i = 0; i < len; i++) {
                const int d = (int)pool[off + 2 * i];
                const int cnt = (int)pool[off + 2 * i + 1];
                const DgNode h = nd[d];
                const float s = score[d];
                float ns;
                if ((h.flags & DG_NF_BACKBONE) && h.weight == 1) {
                    ns = s - 10.0f;
                } else {
                    const int c = cov[h.bbpos];
                    ns = (float)cnt - (float)c * 0.5f + s;
                }
                if (ns > bs) { bs = ns; bd = d; }
            }
            if (bd >= 0) { score[n] = bs; best[n] = bd; }
        }
        __syncthreads();              // scores of this wavefront land before anyone is released
        if (n >= 0) {
            const uint32_t off = nd[n].in_off;
            const int len = nd[n].in_len;
            for (int i = 0; i < len; i++) {
                const int s = (int)pool[off + i];
                if (atomicSub(&nd[s].pending, 1) == 1) {
                    const uint32_t pos = atomicAdd(&s_qt, 1u);
                    if (pos < N) queue[pos] = s;
                }
            }
        }
        __syncthreads();
        qh += m;
        qt = s_qt;
        if (qt > N) { if (lane == 0) dg_fail(p, DG_E_INTERNAL); return; }
    }

    // :443-456 walk the best edges from enter; :327-373 segmentation.  The walk
    // is a pointer chase; lane 0 does it and keeps the consensus in cns_tmp.
    __shared__ uint32_t s_len, s_nseg;
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
            const DgNode h = nd[v];
            const int nxt = best[v];
            if (!(h.base == eb || h.base == xb)) {
                tmp[idx] = h.base;
                const int w = h.weight;
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
    __syncthreads();
    const uint32_t keep = s_len, nseg = s_nseg;
    uint8_t *out = p.cns + p.cns_off[t];
    for (uint32_t i = lane; i < keep; i += 64) out[i] = tmp[i];
    const uint64_t so = p.seg_first[t];
    for (uint32_t i = lane; i < nseg; i += 64) {
        p.seg_r0[so + i] = segs[2 * i];
        p.seg_r1[so + i] = segs[2 * i + 1];
    }
}
