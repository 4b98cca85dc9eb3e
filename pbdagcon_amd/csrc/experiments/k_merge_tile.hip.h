// k_merge_tile.hip.h -- stage (b), fast path: mergeNodes on LDS-resident tiles.
//
// A cut vertex (k_cutmap: a backbone vertex every read passes through) splits the sweep of
// mergeNodes exactly (argument above k_cuts in k_merge.hip.h): the stretch between two
// consecutive cuts is swept in the reference's FIFO order and touches no state of any other
// stretch.  At 40x about one position in five is a cut, so a target falls into a couple of
// thousand stretches of a few dozen vertices.  One wave per stretch wastes the wave (a visit
// occupies one to four lanes) and one lane per stretch on HBM wastes the memory system (64
// dependent 4-byte accesses to 64 different lines per instruction).  Here a workgroup takes a
// TILE -- the stretches between the first cut at or behind backbone position k*G and the first cut
// at or behind (k+1)*G -- copies the tile's vertex records and adjacency lists into LDS with
// coalesced 16-byte loads (vertex ids are in backbone-position order, so the records are one
// contiguous range and the lists two), lets every lane sweep a stretch of its own with the
// reference's algorithm at LDS latency, and copies the tile back.  Nothing is written to HBM
// before the whole tile is done: a tile that does not fit (LDS budget, recursion depth) or trips
// an invariant is simply handed to the wave-per-segment kernel (k_merge_list) untouched.
//
// LDS image of a tile (word offsets into dg_smem):
//   [o_nodes)  8 words per vertex of [va, vb]            (DgNode, list offsets rewritten to LDS offsets)
//   [o_ins)    3 words per inserted vertex of the tile    (out dst, out count, in src)
//   [o_bb)     3*capb words per backbone position [pa, pb]
//   [o_q)      1 word per vertex: the FIFO of each stretch (stretch i uses the words of its own vertices)
//   [o_stk)    2*DG_T_DEPTH words per lane: mergeInNodes' recursion (node, last key)
//   [o_grow)   lists that outgrew their slots (bump allocated; flushed to the target's growth region)
#pragma once
#include <hip/hip_runtime.h>
#include "../dagcon_dev.h"

#define DG_T_DEPTH 8            // frames of mergeInNodes' recursion kept per lane (deeper: fallback)
#define DG_T_NONE 0xFFFFFFFFu
#define DG_T_LANES 64

extern __shared__ uint32_t dg_smem[];

// ---- k_cutmap: cut flags and "next cut at or behind position p" per target ----------------------
// nextcut[p] = smallest cut position >= p (blen + 1, the exit, when there is none); p in 0..blen+1.
// Position 0 (enter) counts as a cut: it starts the first stretch.
__global__ __launch_bounds__(1024) void k_cutmap(DgParams p) {
    __shared__ uint32_t s_min[1024];
    __shared__ uint32_t s_carry;
    __shared__ int s_kg[16];
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t blen = p.tlen[t];
    const uint32_t np = blen + 2;
    const DgNode *nd = p.nodes + p.node_base[t];
    const uint32_t *pool = p.pool + p.pool_base[t];
    const uint32_t *bid = p.bid + p.bbv_base[t];
    uint32_t *nextcut = p.nextcut + p.bbv_base[t];
    // reads threaded into the graph = uses of enter's out-edges (AlnGraphBoost.cpp:60,106)
    const DgNode en = nd[0];
    int kg = 0;
    for (uint32_t i = tid; i < en.out_len; i += 1024) kg += (int)pool[en.out_off + 2 * i + 1];
    for (int o = 32; o; o >>= 1) kg += __shfl_xor(kg, o);
    if ((tid & 63) == 0) s_kg[tid >> 6] = kg;
    if (tid == 0) s_carry = blen + 1;
    __syncthreads();
    kg = 0;
    for (int i = 0; i < 16; i++) kg += s_kg[i];
    // chunks of 1024 positions from the top down; inside a chunk a suffix-min scan
    const uint32_t nchunk = (np + 1023) / 1024;
    for (uint32_t c = nchunk; c-- > 0;) {
        const uint32_t pos = c * 1024 + tid;
        uint32_t v = 0xFFFFFFFFu;
        if (pos < np) {
            bool cut = pos == 0;
            if (pos >= 1 && pos <= blen) cut = nd[bid[pos]].weight - 1 == kg;
            if (cut) v = pos;
        }
        s_min[tid] = v;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            const uint32_t x = tid + o < 1024 ? s_min[tid + o] : 0xFFFFFFFFu;
            __syncthreads();
            if (x < s_min[tid]) s_min[tid] = x;
            __syncthreads();
        }
        const uint32_t carry = s_carry;
        if (pos < np) nextcut[pos] = s_min[tid] != 0xFFFFFFFFu ? s_min[tid] : carry;
        __syncthreads();
        if (tid == 0 && s_min[0] != 0xFFFFFFFFu) s_carry = s_min[0];
        __syncthreads();
    }
}

// ---- LDS image accessors ---------------------------------------------------------------------
struct DgTile {
    int va;                      // first vertex of the tile
    uint32_t o_nodes;            // word offset of the vertex records
    uint32_t o_stk;              // this lane's recursion stack
    uint32_t grow_end;           // end of the growth region
    uint32_t i_top;              // word that holds the growth region's bump cursor
    uint32_t i_abort;            // word that is set when the tile gives up
    bool err;
};

#define DGT_W(T, v, k) dg_smem[(T).o_nodes + ((uint32_t)((v) - (T).va) << 3) + (k)]
#define DGT_H(T, v, h) (reinterpret_cast<uint16_t *>(dg_smem))[(((T).o_nodes + ((uint32_t)((v) - (T).va) << 3)) << 1) + (h)]
// halves: 0 out_len, 1 in_len, 2 base|flags<<8, 12 out_cap, 13 in_cap.  Two lanes may work on the two
// sides of one cut vertex at the same time (in-side: the stretch in front, out-side: the stretch
// behind), so the 16-bit fields are stored as 16-bit LDS stores, never as read-modify-write of a word
#define DGT_OUTLEN(T, v) ((int)DGT_H(T, v, 0))
#define DGT_INLEN(T, v) ((int)DGT_H(T, v, 1))
#define DGT_BASE(T, v) ((int)(DGT_H(T, v, 2) & 0xffu))
#define DGT_WEIGHT(T, v) DGT_W(T, v, 2)
#define DGT_PEND(T, v) DGT_W(T, v, 3)
#define DGT_OUTOFF(T, v) DGT_W(T, v, 4)
#define DGT_INOFF(T, v) DGT_W(T, v, 5)
#define DGT_OUTCAP(T, v) ((int)DGT_H(T, v, 12))
#define DGT_INCAP(T, v) ((int)DGT_H(T, v, 13))
#define DGT_P(off) dg_smem[(off)]

__device__ __forceinline__ void dgt_fail(DgTile &T) {
    T.err = true;
    dg_smem[T.i_abort] = 1u;
}
__device__ __forceinline__ bool dgt_in_tile(const DgTile &T, int v, int vb) { return v >= T.va && v <= vb; }

__device__ __forceinline__ int dgt_out_find(DgTile &T, int v, int dst) {
    const uint32_t off = DGT_OUTOFF(T, v);
    const int n = DGT_OUTLEN(T, v);
    for (int i = 0; i < n; i++)
        if ((int)DGT_P(off + 2 * i) == dst) return i;
    return -1;
}
__device__ __forceinline__ int dgt_in_find(DgTile &T, int v, int src) {
    const uint32_t off = DGT_INOFF(T, v);
    const int n = DGT_INLEN(T, v);
    for (int i = 0; i < n; i++)
        if ((int)DGT_P(off + i) == src) return i;
    return -1;
}
__device__ __forceinline__ void dgt_out_erase(DgTile &T, int v, int idx) {
    const uint32_t off = DGT_OUTOFF(T, v);
    const int n = DGT_OUTLEN(T, v);
    for (int i = idx; i + 1 < n; i++) {
        DGT_P(off + 2 * i) = DGT_P(off + 2 * i + 2);
        DGT_P(off + 2 * i + 1) = DGT_P(off + 2 * i + 3);
    }
    DGT_H(T, v, 0) = (uint16_t)(n - 1);
}
__device__ __forceinline__ void dgt_in_erase(DgTile &T, int v, int idx) {
    const uint32_t off = DGT_INOFF(T, v);
    const int n = DGT_INLEN(T, v);
    for (int i = idx; i + 1 < n; i++) DGT_P(off + i) = DGT_P(off + i + 1);
    DGT_H(T, v, 1) = (uint16_t)(n - 1);
}
__device__ __forceinline__ uint32_t dgt_alloc(DgTile &T, uint32_t words) {
    const uint32_t off = atomicAdd(&dg_smem[T.i_top], words);
    if (off + words > T.grow_end) { dgt_fail(T); return DG_T_NONE; }
    return off;
}
__device__ __forceinline__ void dgt_out_append(DgTile &T, int v, int dst, int count) {
    uint32_t off = DGT_OUTOFF(T, v);
    const int n = DGT_OUTLEN(T, v);
    if (n >= DGT_OUTCAP(T, v)) {
        uint32_t ncap = 2u * (uint32_t)(n + 1);
        if (ncap < 4) ncap = 4;
        if (ncap > 65535u) { dgt_fail(T); return; }
        const uint32_t noff = dgt_alloc(T, 2u * ncap);
        if (noff == DG_T_NONE) return;
        for (int i = 0; i < 2 * n; i++) DGT_P(noff + i) = DGT_P(off + i);
        off = noff;
        DGT_OUTOFF(T, v) = noff; DGT_H(T, v, 12) = (uint16_t)ncap;
    }
    DGT_P(off + 2 * n) = (uint32_t)dst;
    DGT_P(off + 2 * n + 1) = (uint32_t)count;
    DGT_H(T, v, 0) = (uint16_t)(n + 1);
}
__device__ __forceinline__ void dgt_in_append(DgTile &T, int v, int src) {
    uint32_t off = DGT_INOFF(T, v);
    const int n = DGT_INLEN(T, v);
    if (n >= DGT_INCAP(T, v)) {
        uint32_t ncap = 2u * (uint32_t)(n + 1);
        if (ncap < 4) ncap = 4;
        if (ncap > 65535u) { dgt_fail(T); return; }
        const uint32_t noff = dgt_alloc(T, ncap);
        if (noff == DG_T_NONE) return;
        for (int i = 0; i < n; i++) DGT_P(noff + i) = DGT_P(off + i);
        off = noff;
        DGT_INOFF(T, v) = noff; DGT_H(T, v, 13) = (uint16_t)ncap;
    }
    DGT_P(off + n) = (uint32_t)src;
    DGT_H(T, v, 1) = (uint16_t)(n + 1);
}
// boost::clear_vertex + deleted flag (AlnGraphBoost.cpp:269-273)
__device__ __forceinline__ void dgt_reap(DgTile &T, int v) {
    const uint32_t ooff = DGT_OUTOFF(T, v), ioff = DGT_INOFF(T, v);
    const int no = DGT_OUTLEN(T, v), ni = DGT_INLEN(T, v);
    for (int i = 0; i < no; i++) {
        const int d = (int)DGT_P(ooff + 2 * i);
        const int k = dgt_in_find(T, d, v);
        if (k >= 0) dgt_in_erase(T, d, k);
    }
    for (int i = 0; i < ni; i++) {
        const int s = (int)DGT_P(ioff + i);
        const int k = dgt_out_find(T, s, v);
        if (k >= 0) dgt_out_erase(T, s, k);
    }
    DGT_H(T, v, 0) = 0; DGT_H(T, v, 1) = 0;
    DGT_H(T, v, 2) = (uint16_t)(DGT_H(T, v, 2) | (DG_NF_DELETED << 8));
}

// Smallest key > last that at least two eligible neighbours of n share (std::map<char,...> order,
// AlnGraphBoost.cpp:163-174 / :218-227), 256 when there is none.  IN: in-neighbours with one
// out-edge; else out-neighbours with one in-edge.  The lists are re-read from the live graph
// after every merged group: groups already merged are gone and no new group can form (an
// ineligible neighbour keeps a second edge whatever is merged around it), so this equals the
// reference's candidate lists, which are fixed before the first group.
template <bool IN>
__device__ __forceinline__ int dgt_next_key(DgTile &T, int n, int last) {
    const uint32_t off = IN ? DGT_INOFF(T, n) : DGT_OUTOFF(T, n);
    const int len = IN ? DGT_INLEN(T, n) : DGT_OUTLEN(T, n);
    int best = 256;
    for (int i = 0; i < len; i++) {
        const int x = (int)DGT_P(off + (IN ? i : 2 * i));
        if ((IN ? DGT_OUTLEN(T, x) : DGT_INLEN(T, x)) != 1) continue;
        const int b = DGT_BASE(T, x);
        if (b <= last || b >= best) continue;
        int cnt = 0;
        for (int j = 0; j < len; j++) {
            const int y = (int)DGT_P(off + (IN ? j : 2 * j));
            cnt += (IN ? DGT_OUTLEN(T, y) : DGT_INLEN(T, y)) == 1 && DGT_BASE(T, y) == b;
        }
        if (cnt >= 2) best = b;
    }
    return best;
}

// mergeInNodes(n), the group with key b (AlnGraphBoost.cpp:176-212): the survivor is the first
// member in list order; the others are folded into it one by one, in list order (the reference
// first sums all the victims' out-edge counts and weights, :183-190, and then re-points their
// in-edges, :193-212; the two parts touch different words, so victim by victim gives the same)
__device__ __forceinline__ int dgt_merge_in_group(DgTile &T, int n, int b) {
    int an = -1;
    for (;;) {
        const uint32_t off = DGT_INOFF(T, n);
        const int len = DGT_INLEN(T, n);
        int v = -1;
        an = -1;
        for (int i = 0; i < len; i++) {
            const int s = (int)DGT_P(off + i);
            if (DGT_OUTLEN(T, s) == 1 && DGT_BASE(T, s) == b) {
                if (an < 0) an = s; else { v = s; break; }
            }
        }
        if (v < 0) break;
        DGT_P(DGT_OUTOFF(T, an) + 1) += DGT_P(DGT_OUTOFF(T, v) + 1);
        DGT_WEIGHT(T, an) += DGT_WEIGHT(T, v);
        const uint32_t voff = DGT_INOFF(T, v);
        const int vin = DGT_INLEN(T, v);
        for (int k = 0; k < vin; k++) {
            const int n1 = (int)DGT_P(voff + k);
            const int kv = dgt_out_find(T, n1, v);
            if (kv < 0) { dgt_fail(T); return an; }
            const int c = (int)DGT_P(DGT_OUTOFF(T, n1) + 2 * kv + 1);
            const int ka = dgt_out_find(T, n1, an);
            if (ka >= 0) {
                DGT_P(DGT_OUTOFF(T, n1) + 2 * ka + 1) += (uint32_t)c;
                dgt_out_erase(T, n1, kv);
            } else {
                // new edge n1->an goes to the END of out[n1] and in[an]
                dgt_out_erase(T, n1, kv);
                dgt_out_append(T, n1, an, c);
                dgt_in_append(T, an, n1);
            }
            if (T.err) return an;
        }
        DGT_H(T, v, 1) = 0;             // its in-edges are gone from the sources' lists already
        dgt_reap(T, v);                 // removes v from in[n] (its single out-edge)
    }
    return an;
}

// mergeInNodes (AlnGraphBoost.cpp:162-215), recursion (:213) on a small explicit stack
__device__ __forceinline__ void dgt_merge_in(DgTile &T, int n0) {
    int sp = 0, fr_n = n0, fr_last = -1;
    for (;;) {
        const int b = dgt_next_key<true>(T, fr_n, fr_last);
        if (b == 256) {
            if (sp == 0) break;
            sp--;
            fr_n = (int)dg_smem[T.o_stk + 2 * sp]; fr_last = (int)dg_smem[T.o_stk + 2 * sp + 1];
            continue;
        }
        const int an = dgt_merge_in_group(T, fr_n, b);
        if (T.err) return;
        if (sp >= DG_T_DEPTH) { dgt_fail(T); return; }
        dg_smem[T.o_stk + 2 * sp] = (uint32_t)fr_n; dg_smem[T.o_stk + 2 * sp + 1] = (uint32_t)b;
        sp++;
        fr_n = an; fr_last = -1;
    }
}

// mergeOutNodes (AlnGraphBoost.cpp:217-267)
__device__ __forceinline__ void dgt_merge_out(DgTile &T, int u) {
    int last = -1;
    for (;;) {
        const int b = dgt_next_key<false>(T, u, last);
        if (b == 256) break;
        last = b;
        for (;;) {
            const uint32_t off = DGT_OUTOFF(T, u);
            const int len = DGT_OUTLEN(T, u);
            int an = -1, v = -1, ka0 = -1, kv0 = -1;
            for (int i = 0; i < len; i++) {
                const int d = (int)DGT_P(off + 2 * i);
                if (DGT_INLEN(T, d) == 1 && DGT_BASE(T, d) == b) {
                    if (an < 0) { an = d; ka0 = i; } else { v = d; kv0 = i; break; }
                }
            }
            if (v < 0) break;
            // :236-243 (the single in-edge of both is u's)
            DGT_P(off + 2 * ka0 + 1) += DGT_P(off + 2 * kv0 + 1);
            DGT_WEIGHT(T, an) += DGT_WEIGHT(T, v);
            // :246-265
            const uint32_t voff = DGT_OUTOFF(T, v);
            const int vout = DGT_OUTLEN(T, v);
            for (int k = 0; k < vout; k++) {
                const int n2 = (int)DGT_P(voff + 2 * k);
                const int c = (int)DGT_P(voff + 2 * k + 1);
                const int ka = dgt_out_find(T, an, n2);
                const int kin = dgt_in_find(T, n2, v);
                if (kin < 0) { dgt_fail(T); return; }
                dgt_in_erase(T, n2, kin);
                if (ka >= 0) {
                    DGT_P(DGT_OUTOFF(T, an) + 2 * ka + 1) += (uint32_t)c;
                    DGT_PEND(T, n2) -= 1;      // the victim's unvisited in-edge disappears
                } else {
                    dgt_out_append(T, an, n2, c);
                    dgt_in_append(T, n2, an);
                }
                if (T.err) return;
            }
            DGT_H(T, v, 0) = 0;
            dgt_reap(T, v);                    // removes v from out[u]
        }
    }
}

// ---- the tile kernel ---------------------------------------------------------------------------
// grid (T, tiles per target); block DG_T_LANES lanes; dynamic LDS p.tile_words words.
// A tile whose image does not fit, or that gives up, is appended to p.tile_list for k_merge_list.
__global__ __launch_bounds__(DG_T_LANES) void k_merge_tile(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const int lane = threadIdx.x;
    const uint32_t blen = p.tlen[t];
    const uint32_t G = p.tile_pos;
    const uint32_t k = blockIdx.y;
    if ((uint64_t)k * G > blen) return;
    const uint32_t *nextcut = p.nextcut + p.bbv_base[t];
    const uint32_t *bid = p.bid + p.bbv_base[t];
    // tile = [first cut at or behind k*G, first cut at or behind (k+1)*G]; the exit ends the last one
    const uint32_t pa = nextcut[k * G];
    const uint64_t e0 = (uint64_t)(k + 1) * G;
    const uint32_t pb = e0 > blen ? blen + 1 : nextcut[e0];
    if (pa >= pb || pa > blen) return;                     // (the tile in front reaches over this one)
    if (k > 0 && pa >= e0) return;
    const uint64_t nb = p.node_base[t];
    const uint32_t NT = p.n_nodes[t];
    const int va = (int)bid[pa], vb = (int)bid[pb];
    const bool last_tile = pb == blen + 1;
    const uint32_t K = (uint32_t)(p.aln_begin[t + 1] - p.aln_begin[t]);
    const uint32_t capb = dg_capb(K);
    const uint32_t nv = (uint32_t)(vb - va + 1);
    const uint32_t npos = pb - pa + 1;
    // inserted vertices with an id below v = v - (backbone vertices below v)
    const uint32_t ins_lo = 3u * ((uint32_t)va - pa), ins_n = 3u * (nv - npos);      // pool words of the tile's inserted vertices
    const uint32_t bb0 = 3u * p.t_nins[t];
    const uint32_t bb_lo = bb0 + 3u * capb * pa, bb_n = 3u * capb * npos;
    // LDS layout
    const uint32_t o_ctl = 0;                               // [0] growth cursor, [1] abort, [2] unfit, [3] stretches
    const uint32_t o_cuts = 4;                              // first vertex of each lane's stretch + one
    const uint32_t o_stk = o_cuts + DG_T_LANES + 4;
    const uint32_t o_nodes = (o_stk + DG_T_LANES * 2 * DG_T_DEPTH + 3u) & ~3u;
    const uint32_t o_ins = o_nodes + 8u * nv;
    const uint32_t o_bb = o_ins + ins_n;
    const uint32_t o_q = o_bb + bb_n;
    const uint32_t o_grow = o_q + nv + 1u;
    const uint32_t total = p.tile_words;
    uint32_t *list = p.tile_list;
    DgNode *gn = p.nodes + nb;
    uint32_t *gpool = p.pool + p.pool_base[t];
    const bool fits = (uint64_t)o_grow + (uint64_t)(nv / 2u + 64u) <= total && vb < (int)NT && va < vb;
    if (!fits) {
        if (lane == 0) {
            const uint32_t i = atomicAdd(&list[0], 1u);
            if (i < p.tile_list_cap) { list[4 + 3 * i] = t; list[5 + 3 * i] = (uint32_t)va; list[6 + 3 * i] = last_tile ? DG_T_NONE : (uint32_t)vb; }
        }
        return;
    }
#ifdef DG_STAMPS
    const bool stamp = t == 0 && k == 2;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ts[0] = clock64();
#endif
    if (lane < 4) dg_smem[o_ctl + lane] = lane == 0 ? o_grow : 0u;
    // ---- load: records (two 16-byte loads per vertex), the two list ranges ----
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(gn + va);
        uint4 *dst = reinterpret_cast<uint4 *>(dg_smem + o_nodes);
        for (uint32_t i = lane; i < 2u * nv; i += DG_T_LANES) dst[i] = src[i];
        for (uint32_t i = lane; i < ins_n; i += DG_T_LANES) dg_smem[o_ins + i] = gpool[ins_lo + i];
        for (uint32_t i = lane; i < bb_n; i += DG_T_LANES) dg_smem[o_bb + i] = gpool[bb_lo + i];
    }
    __syncthreads();
#ifdef DG_STAMPS
    ts[1] = clock64();
#endif
    // list offsets -> LDS offsets; a list that lives in the target's growth region (k_lists put it
    // there: more than capb entries) is copied into the tile's growth region
    {
        const uint32_t need_in_first = 0;                   // in-list of va belongs to the tile in front
        (void)need_in_first;
        for (uint32_t i = lane; i < nv; i += DG_T_LANES) {
            uint32_t *w = dg_smem + o_nodes + 8u * i;
            const uint32_t out_len = w[0] & 0xffffu, in_len = w[0] >> 16;
            for (int dir = 0; dir < 2; dir++) {
                const uint32_t off = w[4 + dir];
                const uint32_t len = dir == 0 ? out_len : in_len;
                const uint32_t words = dir == 0 ? 2u * len : len;
                uint32_t noff;
                if (off - ins_lo < ins_n) noff = o_ins + (off - ins_lo);
                else if (off - bb_lo < bb_n) noff = o_bb + (off - bb_lo);
                else if ((i == 0 && dir == 1 && pa != 0) || (i == nv - 1 && dir == 0 && !last_tile)) noff = off;   // not ours
                else {
                    const uint32_t cap = dir == 0 ? (w[6] & 0xffffu) : (w[6] >> 16);
                    const uint32_t cw = dir == 0 ? 2u * cap : cap;
                    noff = atomicAdd(&dg_smem[o_ctl], cw);
                    if (noff + cw > total || off + words > p.pool_size[t]) { dg_smem[o_ctl + 2] = 1u; noff = o_grow; }
                    else for (uint32_t x = 0; x < words; x++) dg_smem[noff + x] = gpool[off + x];
                }
                w[4 + dir] = noff;
            }
        }
    }
    __syncthreads();
#ifdef DG_STAMPS
    ts[2] = clock64();
#endif
    // ---- stretches: the cut vertices inside the tile, at most one per lane ----
    // (a cut = backbone vertex whose weight - 1 equals the reads threaded: nextcut[pos] == pos)
    {
        // count the cuts in (pa, pb); lane l takes cuts [l * n / L, ...): evenly thinned when there are more than lanes
        uint32_t ncut = 0;
        for (uint32_t base = pa + 1; base < pb; base += DG_T_LANES) {
            const uint32_t pos = base + lane;
            const bool c = pos < pb && nextcut[pos] == pos;
            ncut += (uint32_t)__popcll(__ballot(c));
        }
        const uint32_t nst = ncut + 1 < DG_T_LANES ? ncut + 1 : DG_T_LANES;      // stretches
        // stretch s starts at cut number floor(s * (ncut + 1) / nst) (cut 0 = va)
        uint32_t seen = 0;
        if (lane == 0) { dg_smem[o_cuts] = (uint32_t)va; dg_smem[o_ctl + 3] = nst; }
        for (uint32_t base = pa + 1; base < pb; base += DG_T_LANES) {
            const uint32_t pos = base + lane;
            const bool c = pos < pb && nextcut[pos] == pos;
            const unsigned long long m = __ballot(c);
            if (c) {
                const uint32_t idx = seen + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) + 1u;      // cut number (1-based)
                // is it the first cut of a stretch?  s = ceil(idx * nst / (ncut + 1))
                const uint32_t s = (uint32_t)(((uint64_t)idx * nst + ncut) / (ncut + 1u));
                if (s < nst && (uint32_t)(((uint64_t)s * (ncut + 1u)) / nst) == idx) dg_smem[o_cuts + s] = bid[pos];
            }
            seen += (uint32_t)__popcll(m);
        }
        if (lane == 0) dg_smem[o_cuts + nst] = last_tile ? DG_T_NONE : (uint32_t)vb;
    }
    __syncthreads();
    const uint32_t nst = dg_smem[o_ctl + 3];
    bool unfit = dg_smem[o_ctl + 2] != 0;
#ifdef DG_STAMPS
    ts[3] = clock64();
    unsigned long long n_vis = 0, n_min = 0, n_mout = 0;
#endif
    // ---- sweep: one stretch per lane, the reference's FIFO order (AlnGraphBoost.cpp:129-160) ----
    if (!unfit && (uint32_t)lane < nst) {
        DgTile T;
        T.va = va; T.o_nodes = o_nodes; T.o_stk = o_stk + (uint32_t)lane * 2u * DG_T_DEPTH;
        T.grow_end = total; T.i_top = o_ctl; T.i_abort = o_ctl + 1; T.err = false;
        const int c_start = (int)dg_smem[o_cuts + lane];
        const uint32_t ce = dg_smem[o_cuts + lane + 1];
        const int c_end = ce != DG_T_NONE ? (int)ce : 0x7fffffff;
        const int c_hi = ce != DG_T_NONE ? c_end : (int)NT - 1;
        const uint32_t N = (uint32_t)(c_hi - c_start + 1);
        const uint32_t qb = o_q + (uint32_t)(c_start - va);
        uint32_t qh = 0, qt = 1;
        dg_smem[qb] = (uint32_t)c_start;
        while (qh < qt && !T.err) {
            if (dg_smem[T.i_abort]) break;
            const int u = (int)dg_smem[qb + qh];
            qh++;
#ifdef DG_STAMPS
            n_vis++;
#endif
            if (u < c_start || u > c_hi) { dgt_fail(T); break; }      // cannot happen (see k_cuts): refuse rather than race
            const bool skip_in = c_start != 0 && u == c_start;        // the stretch in front merges in[u]
            const bool in_only = u == c_end;                           // ... which is this, for the stretch behind
            if (in_only && qh != qt) { dgt_fail(T); break; }
            if (!skip_in) {
                // mergeInNodes(u) can only find a group among >= 2 in-neighbours with one out-edge
                const uint32_t off = DGT_INOFF(T, u);
                const int n_in = DGT_INLEN(T, u);
                int nc = 0;
                for (int i = 0; i < n_in; i++) nc += DGT_OUTLEN(T, (int)DGT_P(off + i)) == 1;
#ifdef DG_STAMPS
                n_min += nc >= 2;
#endif
                if (nc >= 2) { dgt_merge_in(T, u); if (T.err) break; }
            }
            if (in_only) break;
            {
                const uint32_t off = DGT_OUTOFF(T, u);
                const int n_out = DGT_OUTLEN(T, u);
                int nc = 0;
                for (int i = 0; i < n_out; i++) nc += DGT_INLEN(T, (int)DGT_P(off + 2 * i)) == 1;
#ifdef DG_STAMPS
                n_mout += nc >= 2;
#endif
                if (nc >= 2) { dgt_merge_out(T, u); if (T.err) break; }
            }
            // AlnGraphBoost.cpp:143-158
            const uint32_t off = DGT_OUTOFF(T, u);
            const int n_out = DGT_OUTLEN(T, u);
            for (int i = 0; i < n_out; i++) {
                const int d = (int)DGT_P(off + 2 * i);
                const uint32_t pend = DGT_PEND(T, d) - 1u;
                DGT_PEND(T, d) = pend;
                if (pend == 0) {
                    if (qt >= N) { dgt_fail(T); break; }
                    dg_smem[qb + qt] = (uint32_t)d;
                    qt++;
                }
            }
        }
    }
#ifdef DG_STAMPS
    ts[4] = clock64();
    if (stamp) {
        unsigned long long mx = n_vis, sm = n_vis, mi = n_min, mo = n_mout;
        for (int o = 32; o; o >>= 1) { const unsigned long long x = __shfl_xor(mx, o); mx = x > mx ? x : mx; sm += __shfl_xor(sm, o); mi += __shfl_xor(mi, o); mo += __shfl_xor(mo, o); }
        if (lane == 0) { p.st->dbg[8] = mx; p.st->dbg[9] = sm; p.st->dbg[10] = mi; p.st->dbg[11] = mo; p.st->dbg[12] = nst; p.st->dbg[13] = nv; p.st->dbg[14] = ts[4] - ts[3]; }
    }
#endif
    __syncthreads();
#ifdef DG_STAMPS
    ts[5] = clock64();
#endif
    if (unfit || dg_smem[o_ctl + 1]) {
        // nothing of this tile has reached HBM: hand it to the wave-per-segment kernel as it was
        if (lane == 0) {
            const uint32_t i = atomicAdd(&list[0], 1u);
            if (i < p.tile_list_cap) { list[4 + 3 * i] = t; list[5 + 3 * i] = (uint32_t)va; list[6 + 3 * i] = last_tile ? DG_T_NONE : (uint32_t)vb; }
            atomicAdd(&list[1], 1u);
        }
        return;
    }
    // ---- write back ----
    const uint32_t gused = dg_smem[o_ctl] - o_grow;
    if (lane == 0) {
        uint32_t gb = 0;
        if (gused) {
            gb = atomicAdd(&p.pool_top[t], gused);
            if ((uint64_t)gb + gused > p.pool_size[t]) { dg_fail(p, DG_E_POOL_TGT); p.st->bad_target = t; gb = DG_T_NONE; }
        }
        dg_smem[o_ctl + 2] = gb;
        atomicAdd(&p.st->n_mseg, nst);
    }
    __syncthreads();
    const uint32_t gb = dg_smem[o_ctl + 2];
    if (gb == DG_T_NONE) return;
    for (uint32_t i = lane; i < gused; i += DG_T_LANES) gpool[gb + i] = dg_smem[o_grow + i];
    // list offsets back to pool offsets
    for (uint32_t i = lane; i < nv; i += DG_T_LANES) {
        uint32_t *w = dg_smem + o_nodes + 8u * i;
        for (int dir = 0; dir < 2; dir++) {
            const uint32_t off = w[4 + dir];
            if ((i == 0 && dir == 1 && pa != 0) || (i == nv - 1 && dir == 0 && !last_tile)) continue;
            uint32_t noff;
            if (off >= o_grow) noff = gb + (off - o_grow);
            else if (off >= o_bb) noff = bb_lo + (off - o_bb);
            else noff = ins_lo + (off - o_ins);
            w[4 + dir] = noff;
        }
    }
    __syncthreads();
    {
        // interior records whole; of the first cut the out side, of the last cut the in side
        uint4 *dst = reinterpret_cast<uint4 *>(gn + va);
        const uint4 *src = reinterpret_cast<const uint4 *>(dg_smem + o_nodes);
        const uint32_t i0 = pa != 0 ? 2u : 0u, i1 = last_tile ? 2u * nv : 2u * nv - 2u;
        for (uint32_t i = i0 + lane; i < i1; i += DG_T_LANES) dst[i] = src[i];
        if (lane == 0 && pa != 0) {
            const uint32_t *w = dg_smem + o_nodes;
            gn[va].out_len = (uint16_t)(w[0] & 0xffffu); gn[va].out_off = w[4]; gn[va].out_cap = (uint16_t)(w[6] & 0xffffu);
        }
        if (lane == 1 && !last_tile) {
            const uint32_t *w = dg_smem + o_nodes + 8u * (nv - 1);
            gn[vb].in_len = (uint16_t)(w[0] >> 16); gn[vb].pending = (int32_t)w[3];
            gn[vb].in_off = w[5]; gn[vb].in_cap = (uint16_t)(w[6] >> 16);
        }
        for (uint32_t i = lane; i < ins_n; i += DG_T_LANES) gpool[ins_lo + i] = dg_smem[o_ins + i];
        // backbone blocks: of position pa only the out half, of position pb only the in half
        const uint32_t b0 = pa != 0 ? 2u * capb : 0u;          // skip pa's in half?  no: layout is [out 2*capb][in capb]
        (void)b0;
        for (uint32_t i = lane; i < bb_n; i += DG_T_LANES) {
            const uint32_t blk = i / (3u * capb), r = i - blk * 3u * capb;
            const bool is_in = r >= 2u * capb;
            if (blk == 0 && is_in && pa != 0) continue;
            if (blk == npos - 1 && !is_in && !last_tile) continue;
            gpool[bb_lo + i] = dg_smem[o_bb + i];
        }
    }
#ifdef DG_STAMPS
    ts[6] = clock64();
    if (stamp && lane == 0) { for (int i = 0; i < 6; i++) p.st->dbg[i] = ts[i + 1] - ts[i]; }
#endif
}
