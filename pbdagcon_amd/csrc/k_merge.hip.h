// k_merge.hip.h -- stage (b): mergeNodes (AlnGraphBoost.cpp:129-273).
//
// The reference's merge is an order-dependent sequential algorithm: survivor
// identity and adjacency ORDER depend on the FIFO visiting order and on the
// state of the graph at each visit, and the consensus depends on them through
// the first-wins tie-breaks of bestPath.  It is therefore executed in exactly
// the reference's order, one wave per target; parallelism across the chip
// comes from the targets in flight, and inside a visit from the 64 lanes:
//
//   * a visit gathers the records of all in- and out-neighbours of the
//     dequeued vertex at once (lanes 0-31 in-entries, lanes 32-63 out-entries)
//     and decides with ballots whether any merge group exists; when none does
//     (the common case) the FIFO bookkeeping of AlnGraphBoost.cpp:143-158 is
//     done by the out lanes in one step (pending counters, ballot-ranked queue
//     positions);
//   * a visit that has merge work, or lists longer than 32, takes the
//     reference-literal single-lane path below (mergeInNodes with its recursion
//     made explicit, mergeOutNodes), on the same ordered slot lists.
//
// All list primitives keep the container semantics of
// boost::adjacency_list<vecS,vecS,bidirectionalS> (append on add_edge, stable
// erase on clear_vertex, edge(u,v) = first match).
//
// `visited` flags are not stored.  Invariant of the reference: an edge is
// visited iff its source has been dequeued and processed (new edges copy the
// flag of the edge they replace, whose source has the same processed state).
// So "all in-edges of v visited" == pending[v] == 0 where pending counts the
// in-edges whose source is unprocessed; it only changes when a source is
// processed or when mergeOutNodes folds an unprocessed victim's out-edge into
// an existing edge of the survivor.
#pragma once
#include <hip/hip_runtime.h>
#include "dagcon_dev.h"

struct DgGraph {
    DgNode *nd;
    int32_t *queue;
    uint32_t *pool;
    uint32_t pool_size;
    uint32_t *pool_top;
    int32_t *stk;
    uint32_t stk_words;
    DgStatus *st;
    uint32_t t;
    bool err;
};

__device__ __forceinline__ void dgg_fail(DgGraph &g, uint32_t bit) {
    if (!g.err) { atomicOr(&g.st->err_flags, bit); g.st->bad_target = g.t; }
    g.err = true;
}

// ---- ordered slot lists (single lane) --------------------------------------
__device__ inline int dgg_out_find(DgGraph &g, int v, int dst) {
    const uint32_t off = g.nd[v].out_off;
    const int n = g.nd[v].out_len;
    for (int i = 0; i < n; i++)
        if ((int)g.pool[off + 2 * i] == dst) return i;
    return -1;
}
__device__ inline int dgg_in_find(DgGraph &g, int v, int src) {
    const uint32_t off = g.nd[v].in_off;
    const int n = g.nd[v].in_len;
    for (int i = 0; i < n; i++)
        if ((int)g.pool[off + i] == src) return i;
    return -1;
}
__device__ inline void dgg_out_erase(DgGraph &g, int v, int idx) {
    const uint32_t off = g.nd[v].out_off;
    const int n = g.nd[v].out_len;
    for (int i = idx; i + 1 < n; i++) {
        g.pool[off + 2 * i] = g.pool[off + 2 * i + 2];
        g.pool[off + 2 * i + 1] = g.pool[off + 2 * i + 3];
    }
    g.nd[v].out_len = (uint16_t)(n - 1);
}
__device__ inline void dgg_in_erase(DgGraph &g, int v, int idx) {
    const uint32_t off = g.nd[v].in_off;
    const int n = g.nd[v].in_len;
    for (int i = idx; i + 1 < n; i++) g.pool[off + i] = g.pool[off + i + 1];
    g.nd[v].in_len = (uint16_t)(n - 1);
}
__device__ inline uint32_t dgg_alloc(DgGraph &g, uint32_t words) {
    const uint32_t off = *g.pool_top;
    if ((uint64_t)off + words > g.pool_size) { dgg_fail(g, DG_E_POOL_TGT); return 0xFFFFFFFFu; }
    *g.pool_top = off + words;
    return off;
}
__device__ inline void dgg_out_append(DgGraph &g, int v, int dst, int count) {
    uint32_t off = g.nd[v].out_off;
    const int n = g.nd[v].out_len;
    if (n >= g.nd[v].out_cap) {
        uint32_t ncap = 2u * (uint32_t)(n + 1);
        if (ncap < 4) ncap = 4;
        if (ncap > 65535u) { dgg_fail(g, DG_E_INTERNAL); return; }
        const uint32_t noff = dgg_alloc(g, 2u * ncap);
        if (noff == 0xFFFFFFFFu) return;
        for (int i = 0; i < 2 * n; i++) g.pool[noff + i] = g.pool[off + i];
        off = noff;
        g.nd[v].out_off = noff; g.nd[v].out_cap = (uint16_t)ncap;
    }
    g.pool[off + 2 * n] = (uint32_t)dst;
    g.pool[off + 2 * n + 1] = (uint32_t)count;
    g.nd[v].out_len = (uint16_t)(n + 1);
}
__device__ inline void dgg_in_append(DgGraph &g, int v, int src) {
    uint32_t off = g.nd[v].in_off;
    const int n = g.nd[v].in_len;
    if (n >= g.nd[v].in_cap) {
        uint32_t ncap = 2u * (uint32_t)(n + 1);
        if (ncap < 4) ncap = 4;
        if (ncap > 65535u) { dgg_fail(g, DG_E_INTERNAL); return; }
        const uint32_t noff = dgg_alloc(g, ncap);
        if (noff == 0xFFFFFFFFu) return;
        for (int i = 0; i < n; i++) g.pool[noff + i] = g.pool[off + i];
        off = noff;
        g.nd[v].in_off = noff; g.nd[v].in_cap = (uint16_t)ncap;
    }
    g.pool[off + n] = (uint32_t)src;
    g.nd[v].in_len = (uint16_t)(n + 1);
}

// boost::clear_vertex + deleted flag (AlnGraphBoost.cpp:269-273)
__device__ inline void dgg_reap(DgGraph &g, int v) {
    const uint32_t ooff = g.nd[v].out_off, ioff = g.nd[v].in_off;
    const int no = g.nd[v].out_len, ni = g.nd[v].in_len;
    for (int i = 0; i < no; i++) {
        const int d = (int)g.pool[ooff + 2 * i];
        const int k = dgg_in_find(g, d, v);
        if (k >= 0) dgg_in_erase(g, d, k);
    }
    for (int i = 0; i < ni; i++) {
        const int s = (int)g.pool[ioff + i];
        const int k = dgg_out_find(g, s, v);
        if (k >= 0) dgg_out_erase(g, s, k);
    }
    g.nd[v].out_len = 0; g.nd[v].in_len = 0;
    g.nd[v].flags |= DG_NF_DELETED;
}

// ---- mergeInNodes (AlnGraphBoost.cpp:162-215), recursion made explicit ------
// Frame in g.stk: [prev_fp, ncand, last_base, ids[ncand], bases[ncand]].
// Group membership and keys are fixed when the frame is made (:166-171).
__device__ inline int dgg_push_in_frame(DgGraph &g, int sp, int fp, int n) {
    const uint32_t off = g.nd[n].in_off;
    const int len = g.nd[n].in_len;
    int nc = 0;
    for (int i = 0; i < len; i++) {
        const int s = (int)g.pool[off + i];
        if (g.nd[s].out_len == 1) nc++;
    }
    if (nc < 2) return -1;
    if ((uint32_t)(sp + 3 + 2 * nc) > g.stk_words) { dgg_fail(g, DG_E_STACK); return -1; }
    g.stk[sp] = fp; g.stk[sp + 1] = nc; g.stk[sp + 2] = -1;
    int k = 0;
    for (int i = 0; i < len; i++) {
        const int s = (int)g.pool[off + i];
        if (g.nd[s].out_len == 1) { g.stk[sp + 3 + k] = s; g.stk[sp + 3 + nc + k] = g.nd[s].base; k++; }
    }
    return sp + 3 + 2 * nc;
}

__device__ inline void dgg_merge_in(DgGraph &g, int n0) {
    int fp = -1;
    int sp = dgg_push_in_frame(g, 0, -1, n0);
    if (sp < 0) return;
    fp = 0;
    while (fp >= 0 && !g.err) {
        const int nc = g.stk[fp + 1];
        const int last = g.stk[fp + 2];
        int *ids = g.stk + fp + 3, *bases = ids + nc;
        // next key in ascending char order (std::map<char,...>) with >= 2 members
        int b = 256;
        for (int i = 0; i < nc; i++) {
            const int bi = bases[i];
            if (bi > last && bi < b) {
                int cnt = 0;
                for (int j = 0; j < nc; j++) cnt += (bases[j] == bi);
                if (cnt >= 2) b = bi;
            }
        }
        if (b == 256) {               // frame done: pop
            sp = fp;
            fp = g.stk[fp];
            continue;
        }
        g.stk[fp + 2] = b;
        int an = -1;
        // :183-190 accumulate out edge information
        for (int i = 0; i < nc; i++) {
            if (bases[i] != b) continue;
            if (an < 0) { an = ids[i]; continue; }
            const int v = ids[i];
            if (g.nd[an].out_len == 0 || g.nd[v].out_len == 0) { dgg_fail(g, DG_E_INTERNAL); return; }
            g.pool[g.nd[an].out_off + 1] += g.pool[g.nd[v].out_off + 1];
            g.nd[an].weight += g.nd[v].weight;
        }
        // :193-212 accumulate in edge information, merge nodes
        bool first = true;
        for (int i = 0; i < nc; i++) {
            if (bases[i] != b) continue;
            if (first) { first = false; continue; }
            const int v = ids[i];
            const uint32_t voff = g.nd[v].in_off;
            const int vin = g.nd[v].in_len;
            for (int k = 0; k < vin; k++) {
                const int n1 = (int)g.pool[voff + k];
                const int kv = dgg_out_find(g, n1, v);
                if (kv < 0) { dgg_fail(g, DG_E_INTERNAL); return; }
                const int c = (int)g.pool[g.nd[n1].out_off + 2 * kv + 1];
                const int ka = dgg_out_find(g, n1, an);
                if (ka >= 0) {
                    g.pool[g.nd[n1].out_off + 2 * ka + 1] += (uint32_t)c;
                    dgg_out_erase(g, n1, kv);
                } else {
                    // new edge n1->an goes to the END of out[n1] and in[an];
                    // erasing v's entry first or last gives the same order
                    dgg_out_erase(g, n1, kv);
                    dgg_out_append(g, n1, an, c);
                    dgg_in_append(g, an, n1);
                }
            }
            g.nd[v].in_len = 0;        // its in-edges are gone from the sources' lists already
            dgg_reap(g, v);            // removes v from in[n] (its single out-edge)
            if (g.err) return;
        }
        // :213 recurse on the survivor
        const int nsp = dgg_push_in_frame(g, sp, fp, an);
        if (nsp >= 0) { fp = sp; sp = nsp; }
    }
}

// ---- mergeOutNodes (AlnGraphBoost.cpp:217-267) ------------------------------
__device__ inline void dgg_merge_out(DgGraph &g, int n) {
    const uint32_t off = g.nd[n].out_off;
    const int len = g.nd[n].out_len;
    int nc = 0;
    for (int i = 0; i < len; i++) {
        const int d = (int)g.pool[off + 2 * i];
        if (g.nd[d].in_len == 1) nc++;
    }
    if (nc < 2) return;
    if ((uint32_t)(2 * nc) > g.stk_words) { dgg_fail(g, DG_E_STACK); return; }
    int *ids = g.stk, *bases = g.stk + nc;
    {
        int k = 0;
        for (int i = 0; i < len; i++) {
            const int d = (int)g.pool[off + 2 * i];
            if (g.nd[d].in_len == 1) { ids[k] = d; bases[k] = g.nd[d].base; k++; }
        }
    }
    int last = -1;
    while (!g.err) {
        int b = 256;
        for (int i = 0; i < nc; i++) {
            const int bi = bases[i];
            if (bi > last && bi < b) {
                int cnt = 0;
                for (int j = 0; j < nc; j++) cnt += (bases[j] == bi);
                if (cnt >= 2) b = bi;
            }
        }
        if (b == 256) break;
        last = b;
        int an = -1;
        // :236-243 accumulate inner edge information: in[an][0] += in[v][0]
        for (int i = 0; i < nc; i++) {
            if (bases[i] != b) continue;
            if (an < 0) { an = ids[i]; continue; }
            const int v = ids[i];
            if (g.nd[an].in_len == 0 || g.nd[v].in_len == 0) { dgg_fail(g, DG_E_INTERNAL); return; }
            const int sa = (int)g.pool[g.nd[an].in_off], sv = (int)g.pool[g.nd[v].in_off];
            const int ka = dgg_out_find(g, sa, an), kv = dgg_out_find(g, sv, v);
            if (ka < 0 || kv < 0) { dgg_fail(g, DG_E_INTERNAL); return; }
            g.pool[g.nd[sa].out_off + 2 * ka + 1] += g.pool[g.nd[sv].out_off + 2 * kv + 1];
            g.nd[an].weight += g.nd[v].weight;
        }
        // :246-265 accumulate and merge outer edge information
        bool first = true;
        for (int i = 0; i < nc; i++) {
            if (bases[i] != b) continue;
            if (first) { first = false; continue; }
            const int v = ids[i];
            const uint32_t voff = g.nd[v].out_off;
            const int vout = g.nd[v].out_len;
            for (int k = 0; k < vout; k++) {
                const int n2 = (int)g.pool[voff + 2 * k];
                const int c = (int)g.pool[voff + 2 * k + 1];
                const int ka = dgg_out_find(g, an, n2);
                const int kin = dgg_in_find(g, n2, v);
                if (kin < 0) { dgg_fail(g, DG_E_INTERNAL); return; }
                dgg_in_erase(g, n2, kin);
                if (ka >= 0) {
                    g.pool[g.nd[an].out_off + 2 * ka + 1] += (uint32_t)c;
                    g.nd[n2].pending -= 1;   // the victim's unvisited in-edge disappears
                } else {
                    dgg_out_append(g, an, n2, c);
                    dgg_in_append(g, n2, an);
                }
            }
            g.nd[v].out_len = 0;
            dgg_reap(g, v);                  // removes v from out[n]
            if (g.err) return;
        }
    }
}

// ---- mergeNodes (AlnGraphBoost.cpp:129-160): one wave per target ------------
__device__ __forceinline__ bool dg_has_group(unsigned long long cand, uint32_t base) {
    // does any base occur twice among the candidate lanes?
    while (cand) {
        const int f = __ffsll((long long)cand) - 1;
        const uint32_t b = (uint32_t)__shfl((int)base, f);
        const unsigned long long same = __ballot(((cand >> (threadIdx.x & 63)) & 1ull) && base == b);
        if (__popcll(same) >= 2) return true;
        cand &= ~same;
    }
    return false;
}

__global__ __launch_bounds__(64) void k_merge(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || !p.tactive[t]) return;
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    DgGraph g;
    g.nd = p.nodes + nb; g.queue = p.queue + nb;
    g.pool = p.pool + p.pool_base[t]; g.pool_size = p.pool_size[t]; g.pool_top = p.pool_top + t;
    g.stk = p.stk + (uint64_t)t * p.stk_words; g.stk_words = p.stk_words;
    g.st = p.st; g.t = t; g.err = false;
    const uint32_t N = p.n_nodes[t];
    uint32_t qh = 0, qt = 1;
    if (lane == 0) g.queue[0] = 0;                       // enter vertex
    __syncthreads();
    int failed = 0;
    while (qh < qt && !failed) {
        const int u = g.queue[qh++];
        const DgNode nu = g.nd[u];
        const int in_len = nu.in_len, out_len = nu.out_len;
        bool slow = in_len > 32 || out_len > 32;
        if (!slow) {
            const bool is_in = lane < 32;
            const int idx = lane & 31;
            const bool valid = is_in ? idx < in_len : idx < out_len;
            int nbr = 0;
            if (valid) nbr = (int)(is_in ? g.pool[nu.in_off + idx] : g.pool[nu.out_off + 2 * idx]);
            uint4 h = make_uint4(0, 0, 0, 0);
            if (valid) h = *reinterpret_cast<const uint4 *>(&g.nd[nbr]);   // lens, base, weight, pending
            const uint32_t n_out = h.x & 0xffffu, n_in = h.x >> 16, base = h.y & 0xffu;
            const bool elig = valid && (is_in ? n_out == 1u : n_in == 1u);
            const unsigned long long em = __ballot(elig);
            const unsigned long long em_in = em & 0xffffffffull, em_out = em & ~0xffffffffull;
            bool work = false;
            if (__popcll(em_in) >= 2) work = dg_has_group(em_in, base);
            if (!work && __popcll(em_out) >= 2) work = dg_has_group(em_out, base);
            if (work) slow = true;
            else {
                // AlnGraphBoost.cpp:143-158: mark out-edges visited, enqueue targets
                // whose in-edges are now all visited, in out-list order
                const bool outl = valid && !is_in;
                const int pend = (int)h.w - 1;
                if (outl) g.nd[nbr].pending = pend;
                const unsigned long long rm = __ballot(outl && pend == 0);
                if (outl && pend == 0) {
                    const uint32_t pos = qt + (uint32_t)__popcll(rm & ((1ull << lane) - 1ull));
                    if (pos < N) g.queue[pos] = nbr;
                }
                qt += (uint32_t)__popcll(rm);
                if (qt > N) { if (lane == 0) dgg_fail(g, DG_E_INTERNAL); failed = 1; }
            }
        }
        if (slow) {
            uint32_t nqt = qt;
            if (lane == 0) {
                dgg_merge_in(g, u);
                dgg_merge_out(g, u);
                const uint32_t off = g.nd[u].out_off;
                const int len = g.nd[u].out_len;
                for (int i = 0; i < len && !g.err; i++) {
                    const int v = (int)g.pool[off + 2 * i];
                    const int pend = g.nd[v].pending - 1;
                    g.nd[v].pending = pend;
                    if (pend == 0) {
                        if (nqt >= N) { dgg_fail(g, DG_E_INTERNAL); break; }
                        g.queue[nqt++] = v;
                    }
                }
            }
            __syncthreads();                             // lane 0's stores before anybody's loads
            qt = (uint32_t)__shfl((int)nqt, 0);
            failed = __shfl((int)g.err, 0);
        }
    }
}
