// k_merge.hip.h -- stage (b): mergeNodes (AlnGraphBoost.cpp:129-273).
//
// The reference's merge is an order-dependent sequential algorithm: survivor
// identity and adjacency ORDER depend on the FIFO visiting order and on the
// state of the graph at each visit, and the consensus depends on them through
// the first-wins tie-breaks of bestPath.  It is therefore executed in exactly
// the reference's order, one wave per target; parallelism across the chip
// comes from the targets in flight, and inside a visit from the 64 lanes:
//
//   * a visit gathers the records of all in- (then out-) neighbours of the
//     dequeued vertex at once, one list entry per lane, and picks the next merge
//     group with ballots (smallest base shared by >= 2 eligible neighbours);
//   * a group is merged cooperatively (dgw_merge_in_group / dgw_merge_out_group):
//     the victims' edges are flattened onto lanes, deduplicated in first-
//     occurrence order with ballot / popcount, and every list that changes is
//     rewritten by one compaction; after a group the vertex is re-evaluated
//     from the live graph (groups already merged are gone, no new group can
//     form, so this equals the reference's frozen candidate lists);
//   * the FIFO bookkeeping of AlnGraphBoost.cpp:143-158 is one step for all
//     out-edges (pending counters, ballot-ranked queue positions);
//   * lists longer than a wave take the reference-literal single-lane path
//     (dgg_*), on the same ordered slot lists.
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
    uint32_t *tfail;               // this target's word of DgParams::tfail
    uint32_t t;
    int err;                       // (an int: in k_merge_q it differs from row to row, and a boolean there is a lane mask with merges at every join)
    // Partial-span pileups (k_cuts2): reads that start or end inside the target put enter -> x and
    // x -> exit edges across the cut vertices, so enter's out-list and exit's in-list are touched by
    // every segment's worker, and so are the out-lists of the few vertices the prologue has visited
    // whose out-edges lead into more than one segment (insertion chains that reads begin with, united
    // by mergeOutNodes(enter) wherever on the backbone they lead): the vertices flagged DG_NF_SHARED.
    // Entries of such a list belong to the segment of the vertex they name,
    // and a worker only ever looks for, changes or erases entries of its own segment; what the
    // reference's order fixes is the ORDER OF APPENDS, and all visits of segment i precede all visits
    // of segment i + 1 in the reference's FIFO (the cut argument above k_cuts).  Hence the protocol
    // (sh != 0): an erase leaves a tombstone (nobody's entries move); an append goes to the worker's
    // own stretch of slots behind the list -- k_cuts2 lays the list out as [entries][slots of segment 0]
    // [slots of segment 1] ..., all tombstones to begin with -- so appends need no waiting and stand in
    // segment order, each segment's in its own order; k_merge_fin squeezes the tombstones out before it
    // visits the exit vertex.  No shared vertex can be a member of a merge group ('^' and '$' are nobody
    // else's base; a vertex with out-edges into two segments has more than one out-edge whatever is
    // merged, since no worker ever removes the last entry that points into its own segment).
    int sh;                        // 1: the protocol is on (workers of k_merge_list)
    int X;                         // the exit vertex
    const uint32_t *sh_tab;        // this target's row of DgParams::sh_cnt
    uint32_t seg;                  // index of the worker's segment in its target
    uint32_t lg_cap;               // slots per stretch
    uint32_t *lg_cnt;              // [DG_SH_MAX + 1] slots of its stretches this worker has used
};

__device__ __forceinline__ void dgg_fail(DgGraph &g, uint32_t bit) {
    if (!g.err) {
        if (bit & DG_E_TARGET_MASK) atomicOr(g.tfail, bit);      // the target is dropped, the batch goes on
        else { atomicOr(&g.st->err_flags, bit); g.st->bad_target = g.t; }
    }
    g.err = true;
}
// The wave-level code keeps every value that steers control flow provably wave-uniform (the
// compiler then keeps it in SGPRs and branches without exec masks): g.err of the wave's DgGraph
// is only ever assigned uniform values.  Single-lane stretches work on a private copy (DG_LANE0).
__device__ __forceinline__ void dgw_fail(DgGraph &g, uint32_t bit, int lane) {
    if (lane == 0) {
        if (bit & DG_E_TARGET_MASK) atomicOr(g.tfail, bit);
        else { atomicOr(&g.st->err_flags, bit); g.st->bad_target = g.t; }
    }
    g.err = true;
}
#define DG_LANE0(G, BODY)                                          \
    do {                                                           \
        bool e_ = false;                                           \
        if (lane == 0) { DgGraph gs = (G); BODY; e_ = gs.err; }    \
        DG_WAVE_FENCE();                                           \
        (G).err = __any((int)e_) != 0;                             \
    } while (0)

// ---- ordered slot lists (single lane) --------------------------------------
#define DGG_SH_IN(g, v) ((g).sh && (v) == (g).X)
// index of v among the target's shared out-lists, -1 if out[v] is v's own business
__device__ inline int dgg_sh_out(const DgGraph &g, int v) {
    if (!g.sh || !(g.nd[v].flags & DG_NF_SHARED) || v == g.X) return -1;
    const uint32_t n = g.sh_tab[0];
    for (uint32_t i = 0; i < n; i++)
        if ((int)g.sh_tab[2u + 2u * i] == v) return (int)i;
    return -1;
}
#define DGG_SH_BASE_OUT(g, i) ((g).sh_tab[3u + 2u * (uint32_t)(i)])
#define DGG_SH_BASE_IN(g) ((g).sh_tab[1])
__device__ inline int dgg_out_find(DgGraph &g, int v, int dst) {
    const uint32_t off = g.nd[v].out_off;
    const int si = dgg_sh_out(g, v);
    const int n = si >= 0 ? (int)DGG_SH_BASE_OUT(g, si) : (int)g.nd[v].out_len;
    for (int i = 0; i < n; i++)
        if ((int)g.pool[off + 2 * i] == dst) return i;
    if (si >= 0) {                                         // ... and what this worker has appended
        const uint32_t lo = (uint32_t)n + g.seg * g.lg_cap;
        for (uint32_t i = lo, e = lo + g.lg_cnt[si]; i < e; i++)
            if ((int)g.pool[off + 2 * i] == dst) return (int)i;
    }
    return -1;
}
__device__ inline int dgg_in_find(DgGraph &g, int v, int src) {
    const uint32_t off = g.nd[v].in_off;
    const int n = DGG_SH_IN(g, v) ? (int)DGG_SH_BASE_IN(g) : (int)g.nd[v].in_len;
    for (int i = 0; i < n; i++)
        if ((int)g.pool[off + i] == src) return i;
    if (DGG_SH_IN(g, v)) {
        const uint32_t lo = (uint32_t)n + g.seg * g.lg_cap;
        for (uint32_t i = lo, e = lo + g.lg_cnt[DG_SH_MAX]; i < e; i++)
            if ((int)g.pool[off + i] == src) return (int)i;
    }
    return -1;
}
__device__ inline void dgg_out_erase(DgGraph &g, int v, int idx) {
    const uint32_t off = g.nd[v].out_off;
    if (dgg_sh_out(g, v) >= 0) { g.pool[off + 2 * idx] = DG_TOMB; g.pool[off + 2 * idx + 1] = 0u; return; }
    const int n = g.nd[v].out_len;
    for (int i = idx; i + 1 < n; i++) {
        g.pool[off + 2 * i] = g.pool[off + 2 * i + 2];
        g.pool[off + 2 * i + 1] = g.pool[off + 2 * i + 3];
    }
    g.nd[v].out_len = (uint16_t)(n - 1);
}
__device__ inline void dgg_in_erase(DgGraph &g, int v, int idx) {
    const uint32_t off = g.nd[v].in_off;
    if (DGG_SH_IN(g, v)) { g.pool[off + idx] = DG_TOMB; return; }
    const int n = g.nd[v].in_len;
    for (int i = idx; i + 1 < n; i++) g.pool[off + i] = g.pool[off + i + 1];
    g.nd[v].in_len = (uint16_t)(n - 1);
}
__device__ inline uint32_t dgg_alloc(DgGraph &g, uint32_t words) {
    const uint32_t off = atomicAdd(g.pool_top, words);   // shared by the segment workers of the target
    if ((uint64_t)off + words > g.pool_size) { dgg_fail(g, DG_E_POOL_TGT); return 0xFFFFFFFFu; }
    return off;
}
__device__ inline void dgg_out_append(DgGraph &g, int v, int dst, int count) {
    uint32_t off = g.nd[v].out_off;
    const int si = dgg_sh_out(g, v);
    if (si >= 0) {
        const uint32_t k = g.lg_cnt[si];
        if (k >= g.lg_cap) { dgg_fail(g, DG_E_LOG_OVF); return; }               // (re-run with longer stretches)
        const uint32_t i = DGG_SH_BASE_OUT(g, si) + g.seg * g.lg_cap + k;
        g.pool[off + 2 * i] = (uint32_t)dst; g.pool[off + 2 * i + 1] = (uint32_t)count;
        g.lg_cnt[si] = k + 1u;
        return;
    }
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
    if (DGG_SH_IN(g, v)) {
        const uint32_t k = g.lg_cnt[DG_SH_MAX];
        if (k >= g.lg_cap) { dgg_fail(g, DG_E_LOG_OVF); return; }
        g.pool[off + DGG_SH_BASE_IN(g) + g.seg * g.lg_cap + k] = (uint32_t)src;
        g.lg_cnt[DG_SH_MAX] = k + 1u;
        return;
    }
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
        if (g.nd[s].out_len == 1 && !(g.sh && (g.nd[s].flags & DG_NF_SHARED))) nc++;
    }
    if (nc < 2) return -1;
    if ((uint32_t)(sp + 3 + 2 * nc) > g.stk_words) { dgg_fail(g, DG_E_STACK); return -1; }
    g.stk[sp] = fp; g.stk[sp + 1] = nc; g.stk[sp + 2] = -1;
    int k = 0;
    for (int i = 0; i < len; i++) {
        const int s = (int)g.pool[off + i];
        if (g.nd[s].out_len == 1 && !(g.sh && (g.nd[s].flags & DG_NF_SHARED))) { g.stk[sp + 3 + k] = s; g.stk[sp + 3 + nc + k] = g.nd[s].base; k++; }
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
        if (g.nd[d].in_len == 1 && !DGG_SH_IN(g, d)) nc++;
    }
    if (nc < 2) return;
    if ((uint32_t)(2 * nc) > g.stk_words) { dgg_fail(g, DG_E_STACK); return; }
    int *ids = g.stk, *bases = g.stk + nc;
    {
        int k = 0;
        for (int i = 0; i < len; i++) {
            const int d = (int)g.pool[off + 2 * i];
            if (g.nd[d].in_len == 1 && !DGG_SH_IN(g, d)) { ids[k] = d; bases[k] = g.nd[d].base; k++; }
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
                    if (!DGG_SH_IN(g, n2)) g.nd[n2].pending -= 1;   // the victim's unvisited in-edge disappears
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

// ============================================================================
// Prefetch wave.  The sweeps of stages (b) and (c) are dependent pointer chases:
// what they cost is the latency of each access, and a first touch of a cache
// line is an HBM miss.  Vertex ids are in backbone-position order, so the sweep
// moves through the vertex records and their lists monotonically; a second
// wave of the same workgroup (same CU, same L1) runs ahead of the worker and
// touches the records and list heads it is about to need.  It only loads; it
// never changes what the worker computes.
// ============================================================================
#define DG_PROG_DONE 0x7fffffff
#define DG_PF_AHEAD 256
#define DG_PF_CHUNK 64

__device__ inline void dg_prefetch_wave(const DgNode *nd, const uint32_t *pool, uint32_t pool_size,
                                        int lo, int N, int *s_prog, int lane, int dir, int ahead) {
    if (ahead <= 0) return;
    int next = dir > 0 ? lo : N - 1;
    unsigned spins = 0;
    uint32_t sink = 0;
    for (;;) {
        const int cur = __hip_atomic_load(s_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == DG_PROG_DONE) break;
        const bool work = dir > 0 ? (next < N && next < cur + ahead) : (next >= lo && next > cur - ahead);
        if (!work) {
            if ((dir > 0 ? next >= N : next < lo) || ++spins > 400000000u) break;
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        const int v = next + dir * lane;
        if (v >= lo && v < N) {
            const uint4 h2 = *(reinterpret_cast<const uint4 *>(&nd[v]) + 1);
            if (h2.x < pool_size) sink ^= pool[h2.x];
            if (h2.y < pool_size) sink ^= pool[h2.y];
        }
        next += dir * DG_PF_CHUNK;
    }
    asm volatile("" ::"v"(sink));
}

// ============================================================================
// Wave-cooperative merge.  Everything below is executed by all 64 lanes with
// wave-uniform control flow; a lane holds one list entry.
// ============================================================================
// pool word / vertex record through a 32-bit BYTE offset from the (wave-uniform) base: the
// compiler can then address with the base in an SGPR pair and the offset in one VGPR instead of
// building a 64-bit address per lane (a target's pool is < 2^30 words, its ids < 2^25)
#define DG_PW(G, OFF) (*reinterpret_cast<uint32_t *>(reinterpret_cast<char *>((G).pool) + (((uint32_t)(OFF)) << 2)))
#define DG_NV(G, V) (*reinterpret_cast<DgNode *>(reinterpret_cast<char *>((G).nd) + (((uint32_t)(V)) << 5)))
#define DG_LT(lane) ((1ull << (lane)) - 1ull)
// orders this wave's earlier stores (a single lane's, on the literal path) before its
// later loads; the worker is one wave, so no s_barrier is involved
#define DG_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")

__device__ __forceinline__ uint4 dg_lo16(const DgNode *n) { return *reinterpret_cast<const uint4 *>(n); }
__device__ __forceinline__ uint4 dg_hi16(const DgNode *n) { return *(reinterpret_cast<const uint4 *>(n) + 1); }
// fields of the two halves of a DgNode
#define DG_H_OUTLEN(h) ((int)((h).x & 0xffffu))
#define DG_H_INLEN(h)  ((int)((h).x >> 16))
#define DG_H_BASE(h)   ((int)((h).y & 0xffu))
#define DG_H_WEIGHT(h) ((int)(h).z)
#define DG_H_PEND(h)   ((int)(h).w)
#define DG_H2_OUTOFF(h) ((h).x)
#define DG_H2_INOFF(h)  ((h).y)
#define DG_H2_OUTCAP(h) ((int)((h).z & 0xffffu))
#define DG_H2_INCAP(h)  ((int)((h).z >> 16))

// v_readlane: the source lane is wave-uniform everywhere it is used here (it comes from a
// ballot), so no cross-lane permute through the LDS crossbar is needed
#define DG_RL(v, l) __builtin_amdgcn_readlane((int)(v), (int)(l))

__device__ __forceinline__ int dg_wave_incl_scan(int v, int lane) {
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(v, o);
        if (lane >= o) v += up;
    }
    return v;
}
__device__ __forceinline__ int dg_wave_sum_masked(int v, unsigned long long m) {
    int acc = 0;
    while (m) {
        const int f = __ffsll((long long)m) - 1;
        acc += DG_RL(v, f);
        m &= m - 1ull;
    }
    return acc;
}

// smallest base > last that at least two candidate lanes share; returns the
// base (or 256) and the mask of its lanes
__device__ __forceinline__ int dg_pick_group(unsigned long long cand, int base, int last, int lane,
                                             unsigned long long *mask) {
    int best = 256;
    unsigned long long bm = 0;
    // candidates carry their base, every other lane a value no base equals: one compare per round
    const int key = ((cand >> lane) & 1ull) ? base : -1 - lane;
    while (cand) {
        const int f = __ffsll((long long)cand) - 1;
        const int b = DG_RL(key, f);
        const unsigned long long same = __ballot(key == b);
        if (b > last && b < best && __popcll(same) >= 2) { best = b; bm = same; }
        cand &= ~same;
    }
    *mask = bm;
    return best;
}

__device__ __forceinline__ uint32_t dg_wave_alloc(DgGraph &g, uint32_t words, int lane) {
    uint32_t off = 0;
    if (lane == 0) { DgGraph gs = g; off = dgg_alloc(gs, words); }
    off = (uint32_t)DG_RL(off, 0);
    if (off == 0xFFFFFFFFu) g.err = true;                // every lane: keeps control flow uniform
    return off;
}

// Removes from in[v] every source held (in `vid`) by a lane of vm (stable), then
// appends `app` when app >= 0.  pend_delta is added to v's pending counter.
// Requires in_len(v) <= 64.
__device__ inline void dgw_in_rewrite(DgGraph &g, int v, int vid, unsigned long long vm, int app,
                                      int pend_delta, int lane) {
    const uint4 h = dg_lo16(&DG_NV(g, v)), h2 = dg_hi16(&DG_NV(g, v));
    const int len = DG_H_INLEN(h);
    uint32_t off = DG_H2_INOFF(h2);
    int cap = DG_H2_INCAP(h2);
    int e = -1;
    if (lane < len) e = (int)DG_PW(g, off + lane);
    bool rm = false;
    for (unsigned long long m = vm; m; m &= m - 1ull) rm |= (e == DG_RL(vid, __ffsll((long long)m) - 1));
    const bool keep = lane < len && !rm;
    const unsigned long long km = __ballot(keep);
    int nlen = __popcll(km);
    const int nidx = __popcll(km & DG_LT(lane));
    if (app >= 0 && nlen + 1 > cap) {
        uint32_t ncap = 2u * (uint32_t)(nlen + 1);
        if (ncap < 4) ncap = 4;
        const uint32_t noff = dg_wave_alloc(g, ncap, lane);
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

// ---- mergeOutNodes, one group (AlnGraphBoost.cpp:229-266) ---------------------
// Lanes 32..63 hold u's out entries: d (target), cnt, h (first half of the
// target's record).  M = lanes of the group (survivor = lowest lane).
// Returns false, with nothing modified, when a list involved is longer than a wave.
__device__ inline bool dgw_merge_out_group(DgGraph &g, int u, const DgNode &nu, unsigned long long M,
                                           int d, int &cnt, uint4 h, bool valid_out, int lane) {
    const int an_lane = __ffsll((long long)M) - 1;
    const unsigned long long vm = M & ~(1ull << an_lane);
    const int an = DG_RL(d, an_lane);
    const bool member = (M >> lane) & 1ull;
    uint4 h2 = make_uint4(0, 0, 0, 0);
    if (member) h2 = dg_hi16(&DG_NV(g, d));
    // members' out entries flattened onto lanes 0..L-1: survivor's first, then victims in order
    int L = 0, src = -1, e = 0;
    uint32_t src_off = 0;
    {
        unsigned long long mm = M;
        while (mm) {
            const int ml = __ffsll((long long)mm) - 1;
            mm &= mm - 1ull;
            const int mlen = DG_RL(DG_H_OUTLEN(h), ml);
            const uint32_t moff = (uint32_t)DG_RL(DG_H2_OUTOFF(h2), ml);
            if (lane >= L && lane < L + mlen) { src = ml; e = lane - L; src_off = moff; }
            L += mlen;
        }
    }
    if (L > 64) return false;
    const bool fl = lane < L;
    int n2 = -1, c2 = 0;
    if (fl) { n2 = (int)DG_PW(g, src_off + 2 * e); c2 = (int)DG_PW(g, src_off + 2 * e + 1); }
    const bool vic_entry = fl && src != an_lane;
    uint4 hn2 = make_uint4(0, 0, 0, 0);
    if (vic_entry) hn2 = dg_lo16(&DG_NV(g, n2));
    if (__ballot(vic_entry && DG_H_INLEN(hn2) > 64)) return false;
    if (g.sh && __ballot(vic_entry && ((hn2.y >> 8) & DG_NF_SHARED))) return false;      // in[exit] is shared: literal path

    // ---- nothing has been modified up to here ----
    // :236-243 count(u->an) += counts of u->victims, weight[an] += weights
    const int add_cnt = dg_wave_sum_masked(cnt, vm);
    const int add_w = dg_wave_sum_masked(DG_H_WEIGHT(h), vm);
    // :246-265 fold the victims' out edges into the survivor's, first occurrence order
    unsigned long long rem = __ballot(fl), first_m = 0;
    int newcnt = c2;
    while (rem) {
        const int f = __ffsll((long long)rem) - 1;
        const int x = DG_RL(n2, f);
        const unsigned long long same = __ballot(fl && n2 == x);
        rem &= ~same;
        first_m |= 1ull << f;
        const int tot = dg_wave_sum_masked(c2, same);
        if (lane == f) newcnt = tot;
        const unsigned long long vsame = same & __ballot(vic_entry);
        const int nv = __popcll(vsame);
        if (nv) {
            const bool is_new = (vsame >> f) & 1ull;       // survivor had no edge to x
            dgw_in_rewrite(g, x, d, vm, is_new ? an : -1, -(nv - (is_new ? 1 : 0)), lane);
            if (g.err) return true;
        }
    }
    // survivor's new out list
    {
        const int nlen = __popcll(first_m);
        uint32_t off = (uint32_t)DG_RL(DG_H2_OUTOFF(h2), an_lane);
        int cap = DG_RL(DG_H2_OUTCAP(h2), an_lane);
        if (nlen > cap) {
            uint32_t ncap = 2u * (uint32_t)(nlen + 1);
            if (ncap < 4) ncap = 4;
            const uint32_t noff = dg_wave_alloc(g, 2u * ncap, lane);
            if (noff == 0xFFFFFFFFu) return true;
            off = noff; cap = (int)ncap;
        }
        if ((first_m >> lane) & 1ull) {
            const int idx = __popcll(first_m & DG_LT(lane));
            DG_PW(g, off + 2 * idx) = (uint32_t)n2;
            DG_PW(g, off + 2 * idx + 1) = (uint32_t)newcnt;
        }
        const int an_w = DG_RL(DG_H_WEIGHT(h), an_lane);
        if (lane == 0) {
            DgNode *a = &DG_NV(g, an);
            a->out_len = (uint16_t)nlen; a->out_off = off; a->out_cap = (uint16_t)cap;
            a->weight = an_w + add_w;
        }
    }
    // u's out list without the victims (stable), survivor's edge count updated
    {
        const bool keep = valid_out && !((vm >> lane) & 1ull);
        const unsigned long long km = __ballot(keep);
        if (keep) {
            const int idx = __popcll(km & DG_LT(lane));
            DG_PW(g, nu.out_off + 2 * idx) = (uint32_t)d;
            if (lane == an_lane) cnt += add_cnt;
            DG_PW(g, nu.out_off + 2 * idx + 1) = (uint32_t)cnt;
        }
        if (lane == 0) DG_NV(g, u).out_len = (uint16_t)__popcll(km);
    }
    // AlnGraphBoost.cpp:269-273 for every victim
    if ((vm >> lane) & 1ull) {
        DgNode *vn = &DG_NV(g, d);
        vn->out_len = 0; vn->in_len = 0; vn->flags |= DG_NF_DELETED;
    }
    return true;
}

// ---- mergeInNodes, one group (AlnGraphBoost.cpp:176-212) ----------------------
// Lanes 0..31 hold n's in entries: s (source), h (first half of its record).
// M = lanes of the group (survivor = lowest lane).  *an_out = survivor.
// Returns false, with nothing modified, when a list involved is longer than a wave.
__device__ inline bool dgw_merge_in_group(DgGraph &g, int n, const DgNode &nn, unsigned long long M,
                                          int s, uint4 h, bool valid_in, int lane, int *an_out) {
    const int an_lane = __ffsll((long long)M) - 1;
    const unsigned long long vm = M & ~(1ull << an_lane);
    const int an = DG_RL(s, an_lane);
    *an_out = an;
    const bool member = (M >> lane) & 1ull;
    const bool victim = (vm >> lane) & 1ull;
    uint4 h2 = make_uint4(0, 0, 0, 0);
    if (member) h2 = dg_hi16(&DG_NV(g, s));
    int c0 = 0;
    if (member) c0 = (int)DG_PW(g, DG_H2_OUTOFF(h2) + 1);     // count of its single out edge (-> n)
    // victims' in entries flattened onto lanes 0..L-1, victims in order
    int L = 0, e = 0;
    uint32_t src_off = 0;
    {
        unsigned long long mm = vm;
        while (mm) {
            const int ml = __ffsll((long long)mm) - 1;
            mm &= mm - 1ull;
            const int mlen = DG_RL(DG_H_INLEN(h), ml);
            const uint32_t moff = (uint32_t)DG_RL(DG_H2_INOFF(h2), ml);
            if (lane >= L && lane < L + mlen) { e = lane - L; src_off = moff; }
            L += mlen;
        }
    }
    if (L > 64) return false;
    const bool fl = lane < L;
    int n1 = -1;
    if (fl) n1 = (int)DG_PW(g, src_off + e);
    uint4 hn1 = make_uint4(0, 0, 0, 0);
    if (fl) hn1 = dg_lo16(&DG_NV(g, n1));
    if (__ballot(fl && DG_H_OUTLEN(hn1) > 64)) return false;
    if (g.sh && __ballot(fl && ((hn1.y >> 8) & DG_NF_SHARED))) return false;    // a shared out-list: literal path

    // ---- nothing has been modified up to here ----
    // :183-190 survivor's out edge count and weight
    const int add_cnt = dg_wave_sum_masked(c0, vm);
    const int add_w = dg_wave_sum_masked(DG_H_WEIGHT(h), vm);
    if (lane == an_lane) {
        DG_PW(g, DG_H2_OUTOFF(h2) + 1) = (uint32_t)(c0 + add_cnt);
        DG_NV(g, an).weight = DG_H_WEIGHT(h) + add_w;
    }
    // :193-212 re-point the victims' in edges to the survivor, in order
    uint32_t a_in_off = (uint32_t)DG_RL(DG_H2_INOFF(h2), an_lane);
    int a_in_cap = DG_RL(DG_H2_INCAP(h2), an_lane);
    int a_in_len = DG_RL(DG_H_INLEN(h), an_lane);
    bool a_dirty = false;
    unsigned long long rem = __ballot(fl);
    while (rem) {
        const int f = __ffsll((long long)rem) - 1;
        const int x = DG_RL(n1, f);
        rem &= ~__ballot(fl && n1 == x);
        // out[x]: drop the entries that point at victims, fold their counts into x->an
        const uint4 hx2 = dg_hi16(&DG_NV(g, x));
        const int xlen = DG_RL(DG_H_OUTLEN(hn1), f);
        const uint32_t xoff = DG_H2_OUTOFF(hx2);
        int dst = -1, c = 0;
        if (lane < xlen) { dst = (int)DG_PW(g, xoff + 2 * lane); c = (int)DG_PW(g, xoff + 2 * lane + 1); }
        bool isv = false;
        for (unsigned long long m = vm; m; m &= m - 1ull) isv |= (dst == DG_RL(s, __ffsll((long long)m) - 1));
        const unsigned long long vmask = __ballot(lane < xlen && isv);
        const int csum = dg_wave_sum_masked(c, vmask);
        const unsigned long long apos = __ballot(lane < xlen && dst == an);
        const bool keep = lane < xlen && !isv;
        const unsigned long long km = __ballot(keep);
        int nlen = __popcll(km);
        if (keep) {
            const int idx = __popcll(km & DG_LT(lane));
            DG_PW(g, xoff + 2 * idx) = (uint32_t)dst;
            DG_PW(g, xoff + 2 * idx + 1) = (uint32_t)(((apos >> lane) & 1ull) ? c + csum : c);
        }
        if (!apos) {
            // new edge x->an: END of out[x] (room is there: at least one entry was dropped)
            if (lane == 0) { DG_PW(g, xoff + 2 * nlen) = (uint32_t)an; DG_PW(g, xoff + 2 * nlen + 1) = (uint32_t)csum; }
            nlen++;
            // ... and END of in[an]
            if (a_in_len + 1 > a_in_cap) {
                uint32_t ncap = 2u * (uint32_t)(a_in_len + 1);
                if (ncap < 4) ncap = 4;
                const uint32_t noff = dg_wave_alloc(g, ncap, lane);
                if (noff == 0xFFFFFFFFu) return true;
                if (lane < a_in_len) DG_PW(g, noff + lane) = DG_PW(g, a_in_off + lane);
                for (int i = 64 + lane; i < a_in_len; i += 64) DG_PW(g, noff + i) = DG_PW(g, a_in_off + i);
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
        const unsigned long long km = __ballot(keep);
        if (keep) DG_PW(g, nn.in_off + __popcll(km & DG_LT(lane))) = (uint32_t)s;
        if (lane == 0) DG_NV(g, n).in_len = (uint16_t)__popcll(km);
    }
    if (victim) {
        DgNode *vn = &DG_NV(g, s);
        vn->out_len = 0; vn->in_len = 0; vn->flags |= DG_NF_DELETED;
    }
    return true;
}

// ---- cut vertices: where the sweep of a target can be split exactly -----------
// A backbone vertex v that EVERY read of the target passes through (weight - 1 ==
// reads threaded into the graph) is a cut of the alignment graph: vertex ids rise
// along every edge addAln creates (k_emit numbers the vertices in backbone order),
// so no edge jumps over v, every vertex before v is an ancestor of v and every
// vertex after it a descendant, and merging (which only unifies vertices of one
// side: v can never be a member of a merge group, that would close a cycle)
// keeps it so.  The reference's FIFO therefore holds exactly {v} when v is
// dequeued: everything before v is finished, nothing after v has started, and
// the two sides touch disjoint state (of v itself: the in list and pending
// counter belong to the side before, the out list to the side after).  The
// segments between consecutive cuts are swept concurrently, each in exactly the
// reference's order: a segment starts with the visit of its cut vertex minus
// mergeInNodes (the previous segment's worker does that as its last act) and
// ends when it dequeues the next cut vertex.
// row of p.cuts: [0] = number of segments, [1 + s] = first vertex of segment s.
__global__ __launch_bounds__(64) void k_cuts(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const int lane = threadIdx.x;
    uint32_t *row = p.cuts + (uint64_t)t * (p.seg_max + 2u);
    const uint32_t blen = p.tlen[t];
    const DgNode *nd = p.nodes + p.node_base[t];
    const uint32_t *pool = p.pool + p.pool_base[t];
    const uint32_t *bid = p.bid + p.bbv_base[t];
    // reads threaded into the graph = uses of enter's out-edges (AlnGraphBoost.cpp:60,106)
    const DgNode en = nd[0];
    int kg = 0;
    for (int i = lane; i < en.out_len; i += 64) kg += (int)pool[en.out_off + 2 * i + 1];
    for (int o = 32; o; o >>= 1) kg += __shfl_xor(kg, o);
    // two sets of cuts: the merge's (seg_max pieces: heavy waves, about a chip's worth of them) and a
    // finer one for bestPath (bp_max pieces: light waves)
    for (int which = 0; which < 2; which++) {
        const uint32_t smax = which == 0 ? p.seg_max : p.bp_max;
        uint32_t *out = which == 0 ? row : p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
        const uint32_t smin = which == 0 ? p.seg_min : (p.seg_min + 2u) / 3u;
        uint32_t want = blen / (smin ? smin : 1u);
        if (want > smax) want = smax;
        if (want < 1) want = 1;
        uint32_t nseg = 1;
        if (lane == 0) out[1] = 0;
        for (uint32_t s = 1; s < want; s++) {
            const uint32_t p0 = 1u + (uint32_t)((uint64_t)s * blen / want);
            const uint32_t span = blen / want / 2u;          // stays below the next ideal position
            uint32_t found = 0;
            for (uint32_t o = 0; o < span && !found; o += 64) {
                const uint32_t pos = p0 + o + (uint32_t)lane;
                uint32_t v = 0;
                bool ok = false;
                if (o + (uint32_t)lane < span && pos >= 1 && pos <= blen) {
                    v = bid[pos];
                    ok = nd[v].weight - 1 == kg;
                }
                const unsigned long long m = __ballot(ok);
                if (m) found = (uint32_t)DG_RL(v, __ffsll((long long)m) - 1);
            }
            if (found) { if (lane == 0) out[1 + nseg] = found; nseg++; }
        }
        if (lane == 0) { out[0] = nseg; if (which == 0 && !p.tile_pos && !p.gcuts) atomicAdd(&p.st->n_mseg, nseg); }
    }
}

// ---- mergeNodes (AlnGraphBoost.cpp:129-160): one wave per (target, segment) ----
#define DG_IN_STACK 48
#define DG_QRING 256

// PF: a second wave runs ahead of the worker and pulls the records and lists it is about to
// need into L2 (pays when the chip has idle wave slots: few targets x segments in flight)
#define DG_PROG_SET(x) __hip_atomic_store(&s_prog, (x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// One segment [c_start, c_end] of target t, swept by the calling wave (PF: by the first wave of the
// block, the second prefetches).  c_end = 0x7fffffff: the segment runs to the exit vertex.
// mode DG_MM_WORKER: a segment between two cuts (with p.gcuts: enter / exit shared, exit not visited, the
//                      segment that starts at enter resumes where the prologue stopped);
//      DG_MM_PROLOGUE: from enter, every visit up to FIFO level `lvl_stop` (the enter visit and the
//                      chains that hang on enter alone), wherever the vertices lie; the queue state is left
//                      in p.pro_state for the first worker;
//      DG_MM_FINISH:   the exit vertex alone (mergeInNodes(exit) and its recursion), after every worker.
#define DG_MM_WORKER 0
#define DG_MM_PROLOGUE 1
#define DG_MM_FINISH 2
// GC: compiled for p.gcuts (the full-span path, k_merge, carries none of the checks for shared lists)
template <bool PF, bool GC = false>
__device__ __forceinline__ void dg_merge_segment(const DgParams &p, const uint32_t t, const int c_start, const int c_end,
                                                 int32_t *stk_base, const int mode = DG_MM_WORKER, const uint32_t me = 0,
                                                 const uint32_t wlo = 0, const int lvl_stop = 0) {
    const int lane = threadIdx.x & 63;
    const uint64_t nb = p.node_base[t];
    const uint32_t NT = p.n_nodes[t];
    const bool has_end = c_end != 0x7fffffff;
    const int c_hi = has_end ? c_end : (int)NT - 1;
    __shared__ int s_prog;
    if (PF) {
        if (threadIdx.x == 0) s_prog = c_start;
        __syncthreads();
    }
    if (PF && threadIdx.x >= 64) {                       // wave 1: prefetcher
        dg_prefetch_wave(p.nodes + nb, p.pool + p.pool_base[t], p.pool_size[t], c_start, c_hi + 1, &s_prog, lane, +1, (int)p.pf_ahead);
        return;
    }
    DgGraph g;
    const bool own_q = GC && p.gcuts && c_start == 0 && mode != DG_MM_FINISH;     // (see DgParams::queue0)
    g.nd = p.nodes + nb; g.queue = own_q ? p.queue0 + nb : p.queue + nb + c_start;          // the segment's own stretch of the queue
    g.pool = p.pool + p.pool_base[t]; g.pool_size = p.pool_size[t]; g.pool_top = p.pool_top + t;
    g.stk = stk_base; g.stk_words = p.stk_words;
    g.st = p.st; g.tfail = p.tfail + t; g.t = t; g.err = false;
    const bool sh = GC && p.gcuts && mode == DG_MM_WORKER;
    g.sh = sh ? 1 : 0; g.X = (int)NT - 1;
    g.sh_tab = p.sh_cnt + (uint64_t)t * (2u + 2u * DG_SH_MAX); g.seg = me - wlo;
    g.lg_cap = p.sh_log; g.lg_cnt = p.seg_done + (uint64_t)me * (DG_SH_MAX + 1u);
    const int X = sh ? (int)NT - 1 : -1;                               // exit, where it must be left alone
    const uint32_t N = own_q ? NT : (uint32_t)(c_hi - c_start + 1);     // vertices this worker can dequeue
    __shared__ int s_stk[2 * DG_IN_STACK];
    __shared__ int s_ring[DG_QRING];                     // the youngest DG_QRING queue entries
    uint32_t qh = 0, qt = 1;
    if (sh && c_start == 0) {
        // the prologue has visited enter (and what hangs on it alone): go on from its queue
        qh = p.pro_state[4u * t]; qt = p.pro_state[4u * t + 1u];
        for (uint32_t i = qh + (uint32_t)lane; i < qt; i += 64) s_ring[i & (DG_QRING - 1)] = g.queue[i];
    } else if (lane == 0) { g.queue[0] = c_start; s_ring[0] = c_start; }    // enter vertex / cut vertex
    DG_WAVE_FENCE();
    // FIFO levels (prologue): entries [.., lvl_end) belong to level lvl
    int lvl = 0;
    uint32_t lvl_end = 1;
    int failed = 0;
    int prog = c_start;
    int u_next = 0;
    bool have_next = false;
    DgNode nu_next;                 // record of u_next, requested before the previous visit's stores
    nu_next.out_len = 0; nu_next.in_len = 0;
#ifdef DG_STAMPS
    unsigned long long c_fast = 0, c_slow = 0, n_fast = 0, n_slow = 0, n_scalar = 0, t_prev = clock64(), c_a = 0, c_b = 0, c_c = 0, c_grp = 0, ng_in = 0, ng_out = 0, c_odd = 0, n_odd = 0, q_a = 0, q_b = 0, q_c = 0, q_d = 0;
#endif
    while (qh < qt && !failed) {
        if (mode == DG_MM_PROLOGUE) {
            if (qh == lvl_end) { lvl++; lvl_end = qt; }
            if (lvl > lvl_stop) break;
            have_next = false;                            // (the look-ahead below assumes nothing about levels)
        }
        // an entry still in the LDS ring has not been overwritten: pushes so far are < qt <= qh + DG_QRING
        int u;
        const bool pre = have_next;
        if (have_next) u = u_next;
        else if (qt - qh <= DG_QRING) u = __builtin_amdgcn_readfirstlane(s_ring[qh & (DG_QRING - 1)]);
        else u = __builtin_amdgcn_readfirstlane(g.queue[qh]);
        have_next = false;
        qh++;
        bool scalar = false, merged = false;
        (void)merged;
#ifdef DG_STAMPS
        unsigned long long ts_pre = 0, ts_in = 0, acc_grp = 0, n_grp_in = 0, n_grp_out = 0;
        const unsigned long long ts0 = clock64();
#endif
        if (PF && u > prog + 15) { prog = u; if (lane == 0) DG_PROG_SET(u); }
        if (mode != DG_MM_PROLOGUE && (u < c_start || u > c_hi)) {    // cannot happen (see k_cuts): refuse rather than race
            dgw_fail(g, DG_E_INTERNAL, lane);
            break;
        }
        const bool skip_in = mode == DG_MM_WORKER && c_start != 0 && u == c_start; // the previous segment's worker merges in[u]
        const bool in_only = u == c_end;                  // ... which is this, for the next segment
        if (in_only && qh != qt) { dgw_fail(g, DG_E_INTERNAL, lane); break; }
        int adj = sh ? -1 : 0;                            // 1: u is a neighbour of the shared enter / exit vertex (-1: not looked yet)
        if (mode == DG_MM_PROLOGUE && u == (int)NT - 1 && lane == 0) p.pro_state[4u * t + 3u] |= 1u;   // exit visited already

        // ---------------- the common case in one look: no merge group on either side --------
        {
#ifdef DG_STAMPS
            const unsigned long long tq0 = clock64();
#endif
            // gfx950 counts stores in vmcnt, so a load issued behind the previous visit's stores
            // waits for them to reach L2: the record was requested before those stores went out
            DgNode nu;
            if (pre) nu = nu_next; else nu = DG_NV(g, u);
            const int eff_in = skip_in ? 0 : (int)nu.in_len, eff_out = in_only ? 0 : (int)nu.out_len;
            if (eff_in <= 32 && eff_out <= 32) {
#ifdef DG_STAMPS
                const unsigned long long tq1 = clock64();
#endif
                const bool is_in = lane < 32;
                const int idx = lane & 31;
                const bool valid = is_in ? idx < eff_in : idx < eff_out;
                // one address per lane, two independent loads (an in lane's second word is not used)
                const uint32_t ea = is_in ? nu.in_off + (uint32_t)idx : nu.out_off + 2u * (uint32_t)idx;
                int nbr = 0, cnt = 0;
                if (valid) { nbr = (int)DG_PW(g, ea); cnt = (int)DG_PW(g, ea + 1); }
#ifdef DG_STAMPS
                asm volatile("" ::"v"(nbr), "v"(cnt));
                const unsigned long long tq2 = clock64();
#endif
                uint4 h = make_uint4(0, 0, 0, 0);
                if (valid) h = dg_lo16(&DG_NV(g, nbr));
                // in lanes: out_len == 1 (low half of h.x), out lanes: in_len == 1 (high half); with shared
                // enter / exit neither is a candidate ('^' / '$' are nobody else's base) and a visit next to
                // them goes the literal way, where their lists are handled by the protocol
                const bool shn = sh && valid && ((h.y >> 8) & DG_NF_SHARED);
                adj = __ballot(shn) ? 1 : 0;
                if (adj) goto generic;
                const unsigned long long cand = __ballot(((h.x >> (is_in ? 0u : 16u)) & 0xffffu) == 1u);
#ifdef DG_STAMPS
                const unsigned long long tq3 = clock64();
#endif
                const unsigned long long c_in = cand & 0xffffffffull;
                unsigned long long M = 0;
                bool in_work = false;
                if (__popcll(c_in) >= 2) in_work = dg_pick_group(c_in, DG_H_BASE(h), -1, lane, &M) != 256;
                if (!in_work && in_only) break;           // nothing to merge in front of the cut: segment done
                if (!in_work) {
                    // mergeOutNodes(u) on the resident out entries (lanes 32..63), group by group
                    bool live = valid && !is_in;
                    int last_out = -1;
                    bool bail = false;
                    for (;;) {
                        const unsigned long long c_out = __ballot(live && DG_H_INLEN(h) == 1);
                        int b = 256;
                        if (__popcll(c_out) >= 2) b = dg_pick_group(c_out, DG_H_BASE(h), last_out, lane, &M);
                        if (b == 256) break;
#ifdef DG_STAMPS
                        const unsigned long long tg0 = clock64();
#endif
                        const bool okg = dgw_merge_out_group(g, u, nu, M, nbr, cnt, h, live, lane);
#ifdef DG_STAMPS
                        c_grp += clock64() - tg0; ng_out++;
#endif
                        if (!okg) { bail = true; break; }
                        merged = true;
                        if (g.err) break;
                        last_out = b;
                        const int an_lane = __ffsll((long long)M) - 1;
                        if (((M >> lane) & 1ull) && lane != an_lane) live = false;     // victims are gone
                        if (live) h = dg_lo16(&DG_NV(g, nbr));                             // pending / lens may have moved
                    }
                    if (!bail) {
                        if (!g.err) {
                            // AlnGraphBoost.cpp:143-158
                            const int pend = DG_H_PEND(h) - 1;
                            const unsigned long long rm = __ballot(live && pend == 0);
                            // the next vertex to visit is known now: the queue's head, or the first
                            // vertex this visit enqueues
                            if (qh < qt) {
                                if (qt - qh <= DG_QRING) { u_next = __builtin_amdgcn_readfirstlane(s_ring[qh & (DG_QRING - 1)]); have_next = true; }
                            } else if (rm) {
                                u_next = DG_RL(nbr, __ffsll((long long)rm) - 1); have_next = true;
                            }
                            // (a vertex this visit releases still shows pending 1 in this copy; the
                            // visit of a vertex never reads its own pending counter)
                            if (have_next) nu_next = DG_NV(g, u_next);
                            if (live) DG_NV(g, nbr).pending = pend;
                            if (live && pend == 0) {
                                const uint32_t pos = qt + (uint32_t)__popcll(rm & DG_LT(lane));
                                if (pos < N) { g.queue[pos] = nbr; s_ring[pos & (DG_QRING - 1)] = nbr; }
                            }
                            qt += (uint32_t)__popcll(rm);
                            if (qt > N) dgw_fail(g, DG_E_INTERNAL, lane);
                        }
                        failed = g.err;
#ifdef DG_STAMPS
                        { unsigned long long now = clock64(); if (merged) { c_slow += now - t_prev; n_slow++; } else { c_fast += now - t_prev; n_fast++; c_odd += tq0 - t_prev; q_a += tq1 - tq0; q_b += tq2 - tq1; q_c += tq3 - tq2; q_d += now - tq3; } t_prev = now; }
#endif
                        continue;
                    }
                    // a list longer than a wave is involved: nothing was modified by the refused
                    // group; the generic path below re-reads the vertex
                }
            }
        }

    generic:
#ifdef DG_STAMPS
        ts_pre = clock64();
#endif
        // ---------------- mergeInNodes(u), recursion on an explicit stack ----------------
        int sp = skip_in ? 0 : 1;
        int fr_n = u, fr_last = -1;                       // top frame lives in registers
        if (adj < 0) {
            // long lists: look for the shared vertices among u's neighbours
            const DgNode nq = DG_NV(g, u);
            bool f = false;
            if (!skip_in) for (int i = lane; i < nq.in_len; i += 64) f |= (DG_NV(g, DG_PW(g, nq.in_off + i)).flags & DG_NF_SHARED) != 0;
            if (!in_only) for (int i = lane; i < nq.out_len; i += 64) f |= (int)DG_PW(g, nq.out_off + 2 * i) == X;
            adj = __ballot(f) ? 1 : 0;
        }
        if (adj) scalar = true;                           // next to enter / exit: the literal path handles their lists
        while (sp > 0 && !scalar) {
            const DgNode nn = DG_NV(g, fr_n);
            if (nn.in_len > 32) { scalar = true; break; }
            const bool valid = lane < nn.in_len;
            int s = 0;
            if (valid) s = (int)DG_PW(g, nn.in_off + lane);
            uint4 h = make_uint4(0, 0, 0, 0);
            if (valid) h = dg_lo16(&DG_NV(g, s));
            const unsigned long long cand = __ballot(valid && DG_H_OUTLEN(h) == 1 && !(sh && ((h.y >> 8) & DG_NF_SHARED)));
            unsigned long long M = 0;
            int b = 256;
            if (__popcll(cand) >= 2) b = dg_pick_group(cand, DG_H_BASE(h), fr_last, lane, &M);
            if (b == 256) {                               // frame done: pop
                sp--;
                if (sp > 0) { fr_n = s_stk[2 * (sp - 1)]; fr_last = s_stk[2 * (sp - 1) + 1]; }
                continue;
            }
            if (sp >= DG_IN_STACK) { scalar = true; break; }
            int an = -1;
#ifdef DG_STAMPS
            const unsigned long long tg0 = clock64();
#endif
            if (!dgw_merge_in_group(g, fr_n, nn, M, s, h, valid, lane, &an)) { scalar = true; break; }
#ifdef DG_STAMPS
            acc_grp += clock64() - tg0; n_grp_in++;
#endif
            merged = true;
            if (g.err) break;
            fr_last = b;
            if (lane == 0) { s_stk[2 * (sp - 1)] = fr_n; s_stk[2 * (sp - 1) + 1] = fr_last; }
            sp++;                                         // :213 recurse on the survivor
            fr_n = an; fr_last = -1;
        }
        if (scalar && !g.err && sp > 0) {
            // finish every open frame, deepest first, on the reference-literal path;
            // groups already merged are gone, so re-evaluating a frame from scratch is exact
            DG_LANE0(g, {
                dgg_merge_in(gs, fr_n);
                for (int f = sp - 2; f >= 0 && !gs.err; f--) dgg_merge_in(gs, s_stk[2 * f]);
            });
        }

#ifdef DG_STAMPS
        ts_in = clock64();
#endif
        if (in_only) break;                               // the next segment's worker does the rest of this visit
        // ---------------- mergeOutNodes(u) + FIFO bookkeeping ----------------
        bool done = false;
        if (scalar) {
            if (!g.err) DG_LANE0(g, { dgg_merge_out(gs, u); });
        }
        int last_out = -1;
        while (!done && !g.err) {
            const DgNode nu = DG_NV(g, u);
            const int out_len = nu.out_len;
            if (out_len > 64) break;                      // bookkeeping by the single-lane loop below
            const bool valid = lane < out_len;
            int d = 0, cnt = 0;
            if (valid) { d = (int)DG_PW(g, nu.out_off + 2 * lane); cnt = (int)DG_PW(g, nu.out_off + 2 * lane + 1); }
            uint4 h = make_uint4(0, 0, 0, 0);
            if (valid) h = dg_lo16(&DG_NV(g, d));
            if (!scalar) {
                const unsigned long long cand = __ballot(valid && DG_H_INLEN(h) == 1 && d != X);
                unsigned long long M = 0;
                int b = 256;
                if (__popcll(cand) >= 2) b = dg_pick_group(cand, DG_H_BASE(h), last_out, lane, &M);
                if (b != 256) {
#ifdef DG_STAMPS
                    const unsigned long long tg0 = clock64();
                    const bool okg = dgw_merge_out_group(g, u, nu, M, d, cnt, h, valid, lane);
                    acc_grp += clock64() - tg0; n_grp_out++;
                    if (okg) {
#else
                    if (dgw_merge_out_group(g, u, nu, M, d, cnt, h, valid, lane)) {
#endif
                        merged = true;
                        last_out = b;
                        continue;                         // re-read u's list, look for the next group
                    }
                    DG_LANE0(g, { dgg_merge_out(gs, u); });   // a list longer than a wave: literal path
                    scalar = true;
                    continue;
                }
            }
            // AlnGraphBoost.cpp:143-158: mark out-edges visited, enqueue targets whose
            // in-edges are now all visited, in out-list order
            const int pend = DG_H_PEND(h) - 1;
            const bool bk = valid && d != X;              // (the exit vertex is k_merge_fin's)
            if (bk) DG_NV(g, d).pending = pend;
            const unsigned long long rm = __ballot(bk && pend == 0);
            if (bk && pend == 0) {
                const uint32_t pos = qt + (uint32_t)__popcll(rm & DG_LT(lane));
                if (pos < N) { g.queue[pos] = d; s_ring[pos & (DG_QRING - 1)] = d; }
            }
            qt += (uint32_t)__popcll(rm);
            if (qt > N) dgw_fail(g, DG_E_INTERNAL, lane);
            done = true;
        }
        if (!done && !g.err) {
            // out list longer than a wave
            uint32_t nqt = qt;
            DG_LANE0(g, {
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
                        s_ring[nqt & (DG_QRING - 1)] = v;
                        gs.queue[nqt++] = v;
                    }
                }
            });
            qt = (uint32_t)DG_RL(nqt, 0);
        }
        failed = g.err;
#ifdef DG_STAMPS
        { unsigned long long now = clock64(); if (merged || scalar) { c_slow += now - t_prev; n_slow++; n_scalar += scalar; c_a += ts_pre - ts0; c_b += ts_in - ts_pre; c_c += now - ts_in; c_grp += acc_grp; ng_in += n_grp_in; ng_out += n_grp_out; } else { c_fast += now - t_prev; n_fast++; c_odd += now - t_prev; n_odd++; } t_prev = now; }
#endif
    }
    if (PF && lane == 0) DG_PROG_SET(DG_PROG_DONE);
    if (mode == DG_MM_PROLOGUE && lane == 0) {
        p.pro_state[4u * t] = qh; p.pro_state[4u * t + 1u] = qt; p.pro_state[4u * t + 2u] = qh;
    }
#ifdef DG_STAMPS
    if (t == 0 && c_start == 0 && lane == 0) { p.st->dbg[0] = n_fast; p.st->dbg[1] = n_slow; p.st->dbg[2] = c_fast; p.st->dbg[3] = c_slow; p.st->dbg[4] = n_scalar; p.st->dbg[5] = c_a; p.st->dbg[6] = c_b; p.st->dbg[7] = c_c; p.st->dbg[8] = c_grp; p.st->dbg[9] = ng_in; p.st->dbg[10] = ng_out; p.st->dbg[11] = c_odd; p.st->dbg[12] = q_a; p.st->dbg[13] = q_b; p.st->dbg[14] = q_c; p.st->dbg[15] = q_d; }
#endif
}

// mergeNodes (AlnGraphBoost.cpp:129-160): one wave per (target, segment of p.cuts)
template <bool PF>
__global__ __launch_bounds__(PF ? 128 : 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_merge(DgParams p) {
    const uint32_t t = blockIdx.x / p.seg_max, seg = blockIdx.x % p.seg_max;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    // (a batch that k_merge_q sweeps leaves its deep targets to this kernel: DgParams::q_kmax)
    if (p.q_kmax && (uint32_t)(p.aln_begin[t + 1] - p.aln_begin[t]) <= p.q_kmax) return;
    const uint32_t *crow = p.cuts + (uint64_t)t * (p.seg_max + 2u);
    const uint32_t nseg = crow[0];
    if (seg >= nseg) return;
    const int c_start = (int)crow[1 + seg];
    const int c_end = seg + 1 < nseg ? (int)crow[2 + seg] : 0x7fffffff;     // the next segment's cut vertex
    dg_merge_segment<PF>(p, t, c_start, c_end, p.stk + (uint64_t)blockIdx.x * p.stk_words);
}

// A worklist of segments (p.tile_list: [0] = entries, [2] = ticket cursor, then (target, first vertex,
// last vertex or DG_NOSEG_END) triples, a target's segments in a row and in order): one wave per
// entry.  Entries are taken by ticket, so a worker that waits for the earlier segments of its target
#define DG_NOSEG_END 0xFFFFFFFFu
#ifndef DG_ML_WAVES
#define DG_ML_WAVES 7           // (config-5 shape, 400 targets: 5 / 6 / 7 / 8 waves per SIMD -> merge 15.5 / 14.1 / 13.7 / 14.1 ms; 74 / 74 / 109 / 180 SGPR spills)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DG_ML_WAVES, DG_ML_WAVES))) void k_merge_list(DgParams p) {
    if (dg_failed(p)) return;
    const uint32_t n = p.tile_list[0] < p.tile_list_cap ? p.tile_list[0] : p.tile_list_cap;
    for (;;) {
        uint32_t i = 0;
        if (threadIdx.x == 0) i = atomicAdd(&p.tile_list[2], 1u);
        i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        if (i >= n) break;
        const uint32_t t = p.tile_list[4 + 3 * i];
        const int c_start = (int)p.tile_list[5 + 3 * i];
        const uint32_t ce = p.tile_list[6 + 3 * i];
        if (dg_tskip(p, t)) continue;
        dg_merge_segment<false, true>(p, t, c_start, ce == DG_NOSEG_END ? 0x7fffffff : (int)ce, p.stk + (uint64_t)blockIdx.x * p.stk_words,
                                      DG_MM_WORKER, i, p.gcuts ? p.wl_first[t] : 0u);
        if (threadIdx.x == 0) atomicAdd(&p.st->n_mseg, 1u);
        DG_WAVE_FENCE();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.tile_list[0] > p.tile_list_cap) dg_fail(p, DG_E_LIST_OVF);
}

// ---- mergeNodes for pileups of partial-span reads: prologue, cuts, epilogue ------------------------
// A read that starts or ends inside the target puts an enter -> x or x -> exit edge across every
// backbone vertex it does not reach, so "every read passes through v" (k_cuts) leaves such pileups in
// one piece.  What is needed of a cut vertex v is weaker: when the reference dequeues v its FIFO holds
// nothing else, everything in front of v is done and nothing behind it has been touched.  That holds for
// a backbone vertex v at position p when
//   (1) every read that COVERS p passes through v (weight - 1 == coverage): no edge of a read jumps over v;
//   (2) the vertices that hang on enter alone -- the insertion chains reads begin with; mergeOutNodes(enter)
//       unites the ones with equal bases, wherever on the backbone they lead -- have all been visited
//       before anything behind v: the FIFO works level by level (level = longest path from enter), such a
//       chain ends at level `lead` at the latest and v's level is at least p, so k_merge_pro visits levels
//       0 .. maxlead + 1 first (that IS the reference's order) and cuts lie at p > 2 (maxlead + 2);
//   (3) a vertex k_merge_pro has visited and whose out-edges lead to both sides of v has its out-list
//       rewritten by two workers: such lists are shared like enter's (k_cuts2 finds them);
//   (4) the dead ends in front of v (the insertion run a read ends with leads to exit only) are done before
//       v: a read that ends at e < p with a trailing run of r vertices needs p - e > r + 1;
//   (5) no read STARTS behind v up to and including v's backbone successors (round 3).  Behind a full-span cut every vertex descends from v, so
//       v has an out-edge towards every predecessor of its successors and can never be a member of a merge group
//       there.  A read that starts at position q with a leading insertion puts a vertex c that hangs on enter in front
//       of bb(q); if bb(q) is v's successor (q = p + 1, or further on behind deletions), every read through v goes on to
//       it and c has v's base, mergeInNodes(bb(q)) unites v and c -- the worker behind the cut rewrites v's in-list (or
//       deletes v) while the worker in front of it is still to run mergeInNodes(v): tools/stress.py seeds 417 / 463
//       found it at the end of round 3, with pieces so short that the two workers met (a different merged graph from
//       run to run, bestPath stuck or out of bounds on it) -- and with long pieces the worker behind always came
//       first, which is not the reference's order either.  The same with a read that starts in between, where the reads
//       through v all delete a stretch (p, q): its vertex bb(q - 1) is a predecessor of bb(q) that does not descend from v
//       either.  So: for every backbone successor bb(q) of v, no read has its first match in (p, q] -- then every
//       predecessor of v's successors descends from v, v has an out-edge towards it, and v cannot be a member of a
//       group behind the cut, the full-span argument.  (v's successors only ever get fewer: merging re-points an edge to
//       a victim at the survivor, which is a successor already);
// and the two vertices every segment touches, enter (out-list) and exit (in-list), follow the protocol
// described at DgGraph::sh; exit itself is visited last of all, by k_merge_fin.  Every worker still
// checks that it dequeues no vertex of another segment and that its FIFO is empty when it reaches its
// end (DG_E_INTERNAL for the target otherwise).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_merge_pro(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const int lane = threadIdx.x;
    const uint64_t ab = p.aln_begin[t];
    const uint32_t K = (uint32_t)(p.aln_begin[t + 1] - ab);
    uint32_t maxlead = 0;
    for (uint32_t r = lane; r < K; r += 64) { const uint32_t l = p.rd_lead[ab + r]; maxlead = l > maxlead ? l : maxlead; }
    for (int o = 32; o; o >>= 1) { const uint32_t x = __shfl_xor(maxlead, o); maxlead = x > maxlead ? x : maxlead; }
    if (lane == 0) p.pro_state[4u * t + 3u] = 0u;
    DG_WAVE_FENCE();
    dg_merge_segment<false, true>(p, t, 0, 0x7fffffff, p.stk + (uint64_t)blockIdx.x * p.stk_words, DG_MM_PROLOGUE, 0, 0, (int)maxlead + 1);
}

// cuts for k_merge_list: up to p.seg_max pieces per target, conditions (1) - (4) above
#define DG_CUT_STARTS 2048u         // reads of a target whose first positions k_cuts2 holds (deeper targets: one piece)
__global__ __launch_bounds__(64) void k_cuts2(DgParams p) {
    __shared__ uint32_t s_dead[2 * 256];
    __shared__ uint32_t s_start[DG_CUT_STARTS];
    __shared__ uint32_t s_cut[72];
    const uint32_t t = blockIdx.x;
    if (dg_failed(p)) return;
    const int lane = threadIdx.x;
    uint32_t nseg = 1;
    const bool skip = dg_tskip(p, t);
    if (!skip) {
        const uint32_t blen = p.tlen[t];
        const DgNode *nd = p.nodes + p.node_base[t];
        const uint32_t *bid = p.bid + p.bbv_base[t];
        const int32_t *cov = p.cov + p.bbv_base[t];
        const uint64_t ab = p.aln_begin[t];
        const uint32_t K = (uint32_t)(p.aln_begin[t + 1] - ab);
        bool allow = p.pro_state[4u * t] < p.pro_state[4u * t + 1u];
        // (4) dead zones behind read ends, (2) the longest chain a read begins with
        uint32_t maxlead = 0, ndead = 0;
        for (uint32_t r0 = 0; r0 < K; r0 += 64) {
            const uint32_t r = r0 + (uint32_t)lane;
            uint32_t tr = 0, e = 0;
            if (r < K) {
                const uint32_t l = p.rd_lead[ab + r];
                maxlead = l > maxlead ? l : maxlead;
                tr = p.rd_trail[ab + r]; e = p.rd_e[ab + r];
                if (r < DG_CUT_STARTS) s_start[r] = p.rd_s[ab + r];            // (5)
            }
            const unsigned long long m = __ballot(tr > 0);
            const uint32_t k = ndead + (uint32_t)__popcll(m & DG_LT(lane));
            if (tr > 0 && k < 256u) { s_dead[2 * k] = e; s_dead[2 * k + 1] = e + tr + 1u; }
            ndead += (uint32_t)__popcll(m);
        }
        for (int o = 32; o; o >>= 1) { const uint32_t x = __shfl_xor(maxlead, o); maxlead = x > maxlead ? x : maxlead; }
        if (ndead > 256u || K > DG_CUT_STARTS) allow = false;
        const uint32_t nstart = K < DG_CUT_STARTS ? K : DG_CUT_STARTS;
        __syncthreads();
        const uint32_t pmin = 2u * (maxlead + 2u);
        uint32_t want = blen / (p.seg_min ? p.seg_min : 1u);
        if (want > p.seg_max) want = p.seg_max;
        if (want > 64u) want = 64u;
        if (want < 1 || !allow) want = 1;
        if (lane == 0) s_cut[0] = 0;
        for (uint32_t s = 1; s < want; s++) {
            const uint32_t p0 = 1u + (uint32_t)((uint64_t)s * blen / want);
            const uint32_t span = blen / want / 2u;          // stays below the next ideal position
            uint32_t found = 0;
            for (uint32_t o = 0; o < span && !found; o += 64) {
                const uint32_t pos = p0 + o + (uint32_t)lane;
                uint32_t v = 0;
                bool ok = false;
                if (o + (uint32_t)lane < span && pos > pmin && pos <= blen) {
                    v = bid[pos];
                    ok = nd[v].weight - 1 == cov[pos];
                    for (uint32_t d = 0; d < ndead && ok; d++) ok = !(pos > s_dead[2 * d] && pos <= s_dead[2 * d + 1]);
                    if (ok) {                                                                      // (5)
                        const DgNode nv = nd[v];
                        const uint32_t *pl_ = p.pool + p.pool_base[t];
                        for (uint32_t e = 0; e < nv.out_len && ok; e++) {
                            const uint32_t dd = pl_[nv.out_off + 2u * e];
                            if (dd == p.n_nodes[t] - 1u || !(nd[dd].flags & DG_NF_BACKBONE)) continue;
                            const uint32_t q = (uint32_t)nd[dd].bbpos;
                            for (uint32_t d = 0; d < nstart && ok; d++) ok = !(s_start[d] > pos && s_start[d] <= q);
                        }
                    }
                }
                const unsigned long long m = __ballot(ok);
                if (m) found = (uint32_t)DG_RL(v, __ffsll((long long)m) - 1);
            }
            if (found) { if (lane == 0) s_cut[nseg] = found; nseg++; }
        }
        __syncthreads();
        // bestPath sweeps finer pieces (its waves are light; every piece is swept three times, dg_bp_sweep):
        // the same conditions, three times as many wanted
        {
            uint32_t *brow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);
            const uint32_t smin_b = p.bp_seg_min;
            uint32_t want_b = blen / (smin_b ? smin_b : 1u);
            if (want_b > p.bp_max) want_b = p.bp_max;
            if (want_b > 64u) want_b = 64u;
            if (want_b < 1 || !allow) want_b = 1;
            uint32_t nb_ = 1;
            if (lane == 0) brow[1] = 0;
            for (uint32_t s = 1; s < want_b; s++) {
                const uint32_t p0 = 1u + (uint32_t)((uint64_t)s * blen / want_b);
                const uint32_t span = blen / want_b / 2u;
                uint32_t found = 0;
                for (uint32_t o = 0; o < span && !found; o += 64) {
                    const uint32_t pos = p0 + o + (uint32_t)lane;
                    uint32_t v = 0;
                    bool ok = false;
                    if (o + (uint32_t)lane < span && pos > pmin && pos <= blen) {
                        v = bid[pos];
                        ok = nd[v].weight - 1 == cov[pos];
                        for (uint32_t d = 0; d < ndead && ok; d++) ok = !(pos > s_dead[2 * d] && pos <= s_dead[2 * d + 1]);
                    }
                    const unsigned long long m = __ballot(ok);
                    if (m) found = (uint32_t)DG_RL(v, __ffsll((long long)m) - 1);
                }
                if (found) { if (lane == 0) brow[1 + nb_] = found; nb_++; }
            }
            if (lane == 0) brow[0] = nb_;
        }
        __syncthreads();
        // shared out-lists: enter's, and those of the vertices the prologue has visited whose out-edges
        // lead into more than one segment; shared in-list: exit's
        DgNode *ndw = p.nodes + p.node_base[t];
        uint32_t *pool = p.pool + p.pool_base[t];
        const uint32_t NT = p.n_nodes[t];
        const int32_t *q0 = p.queue0 + p.node_base[t];
        const uint32_t nvis = p.pro_state[4u * t + 2u];
        uint32_t *tab = p.sh_cnt + (uint64_t)t * (2u + 2u * DG_SH_MAX);
        __shared__ uint32_t s_sh[DG_SH_MAX];
        __shared__ uint32_t s_nsh;
        for (int attempt = 0; attempt < 2; attempt++) {
            if (lane == 0) { s_sh[0] = 0u; s_nsh = 1u; }
            __syncthreads();
            if (nseg > 1)
                for (uint32_t i0 = 1; i0 < nvis; i0 += 64) {
                    const uint32_t i = i0 + (uint32_t)lane;
                    bool multi = false;
                    int y = 0;
                    if (i < nvis) {
                        y = q0[i];
                        const DgNode ny = ndw[y];
                        if (!(ny.flags & DG_NF_DELETED) && y != (int)NT - 1) {
                            uint32_t sa = 0xFFFFFFFFu;
                            for (uint32_t e = 0; e < ny.out_len; e++) {
                                const uint32_t d = pool[ny.out_off + 2u * e];
                                if (d == NT - 1u) continue;
                                uint32_t sg = 0;                     // segment of d: cuts at or below it
                                for (uint32_t c = 1; c < nseg; c++) sg += s_cut[c] <= d;
                                if (sa == 0xFFFFFFFFu) sa = sg; else multi |= sg != sa;
                            }
                        }
                    }
                    const unsigned long long m = __ballot(multi);
                    if (multi) {
                        const uint32_t k = s_nsh + (uint32_t)__popcll(m & DG_LT(lane));
                        if (k < DG_SH_MAX) s_sh[k] = (uint32_t)y;
                    }
                    __syncthreads();
                    if (lane == 0) s_nsh += (uint32_t)__popcll(m);
                    __syncthreads();
                }
            if (s_nsh <= DG_SH_MAX) break;
            nseg = 1;                                        // too many shared lists: one piece (exact, slower)
            __syncthreads();
        }
        const uint32_t nsh = s_nsh;
        if (lane == 0) tab[0] = nsh;
        // every shared list moves to where each segment has p.sh_log slots of its own behind the entries
        // (tombstones to begin with): [entries][slots of segment 0][slots of segment 1] ...
        for (uint32_t li = 0; li <= nsh; li++) {
            const bool is_in = li == nsh;                    // the last one: in[exit]
            const int v = is_in ? (int)NT - 1 : (int)s_sh[li];
            const DgNode nv = ndw[v];
            const uint32_t len = is_in ? nv.in_len : nv.out_len;
            const uint32_t cap = len + nseg * p.sh_log;
            const uint32_t words = is_in ? cap : 2u * cap;
            uint32_t off = 0;
            if (lane == 0) {
                off = cap > 65535u ? 0xFFFFFFFFu : atomicAdd(&p.pool_top[t], words);
                if (off == 0xFFFFFFFFu || (uint64_t)off + words > p.pool_size[t]) { dg_fail(p, DG_E_POOL_TGT); p.st->bad_target = t; off = 0xFFFFFFFFu; }
            }
            off = (uint32_t)DG_RL(off, 0);
            if (off == 0xFFFFFFFFu) break;
            const uint32_t old = is_in ? nv.in_off : nv.out_off;
            const uint32_t lw = is_in ? len : 2u * len;
            for (uint32_t i = lane; i < words; i += 64) {
                uint32_t w = i < lw ? pool[old + i] : DG_TOMB;
                if (!is_in && i >= lw && (i & 1u)) w = 0u;         // (count word of an empty slot)
                pool[off + i] = w;
            }
            if (lane == 0) {
                if (is_in) { ndw[v].in_off = off; ndw[v].in_cap = (uint16_t)cap; tab[1] = len; }
                else { ndw[v].out_off = off; ndw[v].out_cap = (uint16_t)cap; tab[2u + 2u * li] = (uint32_t)v; tab[3u + 2u * li] = len; }
                ndw[v].flags |= DG_NF_SHARED;
            }
        }
    }
    if (skip && lane == 0) { p.cuts_bp[(uint64_t)t * (p.bp_max + 2u)] = 1u; p.cuts_bp[(uint64_t)t * (p.bp_max + 2u) + 1u] = 0u; }
    // the target's segments, in a row and in order
    __shared__ uint32_t s_base;
    if (lane == 0) {
        s_base = atomicAdd(&p.tile_list[0], nseg);
        p.wl_first[t] = s_base;
    }
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t s = lane; s < nseg; s += 64) {
        const uint32_t i = base + s;
        if (i < p.tile_list_cap) {
            p.tile_list[4 + 3 * i] = t;
            p.tile_list[5 + 3 * i] = skip ? 0u : s_cut[s];
            p.tile_list[6 + 3 * i] = s + 1 < nseg ? s_cut[s + 1] : DG_NOSEG_END;
        }
    }
}

// after every worker: tombstones out of the shared lists, then the visit of the exit vertex
// (mergeInNodes(exit) and its recursion: the last visit of the reference's sweep)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_merge_fin(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const int lane = threadIdx.x;
    const uint64_t nb = p.node_base[t];
    DgNode *nd = p.nodes + nb;
    uint32_t *pool = p.pool + p.pool_base[t];
    const uint32_t NT = p.n_nodes[t];
    const uint32_t *tab = p.sh_cnt + (uint64_t)t * (2u + 2u * DG_SH_MAX);
    const uint32_t nsh = tab[0] <= DG_SH_MAX ? tab[0] : DG_SH_MAX;
    for (uint32_t li = 0; li <= nsh; li++) {
        const bool is_in = li == nsh;
        const int v = is_in ? (int)NT - 1 : (int)tab[2u + 2u * li];
        const uint32_t off = is_in ? nd[v].in_off : nd[v].out_off;
        const uint32_t n = is_in ? nd[v].in_cap : nd[v].out_cap;             // entries + every segment's slots
        uint32_t w = 0;                                    // entries kept so far
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            uint32_t a = DG_TOMB, b = 0;
            if (i < n) { a = is_in ? pool[off + i] : pool[off + 2u * i]; if (!is_in) b = pool[off + 2u * i + 1u]; }
            const bool keep = a != DG_TOMB;
            const unsigned long long m = __ballot(keep);
            DG_WAVE_FENCE();                               // (all loads of the round before its stores: w <= i0)
            const uint32_t k = w + (uint32_t)__popcll(m & DG_LT(lane));
            if (keep) { if (is_in) pool[off + k] = a; else { pool[off + 2u * k] = a; pool[off + 2u * k + 1u] = b; } }
            w += (uint32_t)__popcll(m);
            DG_WAVE_FENCE();
        }
        if (lane == 0) {
            if (is_in) nd[v].in_len = (uint16_t)w; else nd[v].out_len = (uint16_t)w;
            nd[v].flags &= (uint8_t)~DG_NF_SHARED;
        }
    }
    DG_WAVE_FENCE();
    if (!(p.pro_state[4u * t + 3u] & 1u))                 // (a tiny target: the prologue came as far as exit)
        dg_merge_segment<false, true>(p, t, (int)NT - 1, 0x7fffffff, p.stk + (uint64_t)blockIdx.x * p.stk_words, DG_MM_FINISH);
    DG_WAVE_FENCE();
    // what bestPath's segment sweeps must leave alone (DG_NF_DEFER, see dg_bp_sweep): enter, the vertices the
    // prologue visited that have a successor in another segment than their own id, and their ancestors
    {
        const uint32_t *crow = p.cuts_bp + (uint64_t)t * (p.bp_max + 2u);     // (bestPath's pieces, k_cuts2)
        const uint32_t nseg = crow[0];
        const int32_t *q0 = p.queue0 + nb;
        const uint32_t nvis = p.pro_state[4u * t + 2u];
        uint32_t *dl = p.defer + (uint64_t)t * (DG_DEFER_MAX + 1u);
        uint32_t nd_ = 0;
        bool ovf = false;
        if (lane == 0) {
            nd[0].flags |= DG_NF_DEFER; dl[1] = 0u; nd_ = 1;
            if (nseg > 1) {
                for (bool more = true; more && !ovf;) {
                    more = false;
                    for (uint32_t i = 1; i < nvis && !ovf; i++) {
                        const int y = q0[i];
                        const DgNode ny = nd[y];
                        if ((ny.flags & (DG_NF_DELETED | DG_NF_DEFER)) || y == (int)NT - 1) continue;
                        uint32_t sy = 0;
                        for (uint32_t c = 1; c < nseg; c++) sy += crow[1 + c] <= (uint32_t)y;
                        bool f = false;
                        for (uint32_t e = 0; e < ny.out_len && !f; e++) {
                            const uint32_t d = pool[ny.out_off + 2u * e];
                            if (d == NT - 1u) continue;
                            if (nd[d].flags & DG_NF_DEFER) { f = true; break; }
                            uint32_t sd = 0;
                            for (uint32_t c = 1; c < nseg; c++) sd += crow[1 + c] <= d;
                            f = sd != sy;
                        }
                        if (f) {
                            if (nd_ >= DG_DEFER_MAX) { ovf = true; break; }
                            nd[y].flags |= DG_NF_DEFER; dl[1 + nd_] = (uint32_t)y; nd_++;
                            more = true;
                        }
                    }
                }
            }
            dl[0] = ovf ? 0xFFFFFFFFu : nd_;                 // (too many: bestPath sweeps the target in one piece)
        }
    }
}

