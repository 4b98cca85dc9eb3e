// k_emit2.hip.h -- addAln (AlnGraphBoost.cpp:64-107) with ONE THREAD PER COLUMN.
//
// k_emit (k_build.hip.h) gives a lane to a read and walks the backbone in lock step: 40 of 64 lanes busy, a divergent
// loop per lane and position for the insertion columns, 410 instructions per position and wave -- two thirds of them
// scalar mask bookkeeping.  What a column contributes to the graph depends on very little besides itself:
//     its backbone position   = start + (match / deletion columns in front of it)          a prefix count
//     an inserted vertex's id = first id of its position's group + the reads in front of it there + its rank in the run
//     the vertex before / after it on the read's path = the nearest match / insertion column to the left / right
// so a wave takes 64 consecutive columns of one read, a lane a column: the counts are ballots and popcounts over the
// wave, the neighbours' vertices come over the lanes (ds_bpermute), and what lies outside the wave's own 64 columns is
// read from the 64 columns before and after them (rarely further: a uniform loop).  Every column writes its own cells
// and records; nothing loops over positions.
//
//   k_blockscan   wave per alignment: backbone position at the start of every 64-column block
//   k_emit2       wave per 64 columns: arrival / departure cells ([read][position], so that consecutive lanes store
//                 consecutive cells), inserted vertices' records and slots, and for the match column that closes a short
//                 insertion chain the chain's key (a byte: k_dedupe)
//   k_dedupe      wave per 8 positions, lane per read: chains with equal keys are duplicates -- the later reads' fold
//                 into the first one's (the argument is above dg_emit_fold in k_build.hip.h)
//
// EXPERIMENT (make experiments, DAGCON_EMIT2=1), measured and not adopted: exact on the whole GPU suite at the first run,
// but a wave of 64 columns costs ~500 instructions (a dozen 64-bit masks, their leading / trailing-bit searches, six
// cross-lane reads, three row stores), 3.8 G per launch against k_emit's 4.1 G: k_emit2 7.2 ms + k_blockscan 0.5 +
// k_dedupe 1.6 + k_lists 4.9 (strided cell reads) = build 15.6 ms against 12.4 at configs[1].
#pragma once
#include <hip/hip_runtime.h>
#include "../dagcon_dev.h"

#define E2_N 0   // no vertex, no advance (outside the window; a raw column addAln skips)
#define E2_M 1
#define E2_D 2
#define E2_I 3
__device__ __forceinline__ int e2_class(const uint16_t c, const bool in) {
    const uint8_t qb = DG_Q(c), tb = DG_T(c);
    if (!in) return E2_N;
    if (qb == tb) return E2_M;                 // AlnGraphBoost.cpp:75
    if (qb == DG_GAP) return E2_D;             // :87
    if (tb == DG_GAP) return E2_I;             // :95
    return E2_N;
}
__device__ __forceinline__ int e2_hi(const unsigned long long m) { return 63 - __clzll((long long)m); }     // m != 0
__device__ __forceinline__ unsigned long long e2_below(const int l) { return l >= 64 ? ~0ull : (1ull << l) - 1ull; }
__device__ __forceinline__ unsigned long long e2_from(const int l) { return l >= 64 ? 0ull : ~0ull << l; }       // lanes >= l

// ---- k_blockscan -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_blockscan(DgParams p) {
    const uint32_t a = blockIdx.x;
    if (a >= p.A || dg_failed(p) || dg_askip(p, a)) return;
    const int lane = threadIdx.x;
    const uint32_t lo = p.n_lo[a], hi = p.n_hi[a];
    if (hi == DG_REDO || hi <= lo) return;
    const uint16_t *buf = p.norm + p.norm_off[a];
    uint32_t *out = p.bbstart + (uint64_t)a * p.bs_stride;
    const uint32_t nblk = (hi - lo + 63u) / 64u;
    uint32_t run = p.n_start[a], mine = 0;
    for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t i = lo + 64u * b + (uint32_t)lane;
        const int cls = e2_class(i < hi ? buf[i] : (uint16_t)0, i < hi);
        if ((uint32_t)lane == (b & 63u)) mine = run;
        run += (uint32_t)__popcll(__ballot(cls == E2_M || cls == E2_D));
        if ((b & 63u) == 63u || b + 1 == nblk) {
            const uint32_t bb = (b & ~63u) + (uint32_t)lane;
            if (bb <= b) out[bb] = mine;
        }
    }
}

// ---- the rare ways out of the 192-column window: uniform loops with scalar loads ---------------------------------
// insertion columns between the last match / deletion column in front of column `iend` and iend (the read starts at lo)
__device__ __noinline__ uint32_t e2_count_ins_back(const uint16_t *buf, const uint32_t lo, uint32_t iend) {
    uint32_t n = 0;
    while (iend > lo) {
        const int cls = e2_class(buf[--iend], true);
        if (cls == E2_M || cls == E2_D) break;
        n += cls == E2_I;
    }
    return n;
}
struct E2Vtx { uint32_t v, pos; bool bb, none; };
// the vertex of the nearest match / insertion column in front of column iend, whose backbone cursor is bb_end
__device__ __noinline__ E2Vtx e2_find_prev(const uint16_t *buf, const uint32_t lo, uint32_t iend, uint32_t bb_end,
                                           const uint32_t *bid, const uint32_t *gbase, const uint32_t *Cm) {
    E2Vtx r; r.v = 0; r.pos = 0; r.bb = true; r.none = true;
    while (iend > lo) {
        const int cls = e2_class(buf[--iend], true);
        if (cls == E2_M) { bb_end--; r.v = bid[bb_end]; r.pos = bb_end; r.bb = true; r.none = false; return r; }
        if (cls == E2_D) { bb_end--; continue; }
        if (cls == E2_I) {
            r.v = gbase[bb_end] + Cm[bb_end] + e2_count_ins_back(buf, lo, iend); r.pos = bb_end; r.bb = false; r.none = false;
            return r;
        }
    }
    return r;                                      // the read's first vertex: enter is in front of it
}
// the vertex of the first match / insertion column at or behind column ibeg (backbone cursor bb_beg, `carry` insertion
// columns of that position already passed); none: the read ends, exit is next
__device__ __noinline__ E2Vtx e2_find_next(const uint16_t *buf, const uint32_t hi, uint32_t ibeg, uint32_t bb_beg, uint32_t carry,
                                           const uint32_t *bid, const uint32_t *gbase, const uint32_t *Cm) {
    E2Vtx r; r.v = 0; r.pos = 0; r.bb = true; r.none = true;
    for (; ibeg < hi; ibeg++) {
        const int cls = e2_class(buf[ibeg], true);
        if (cls == E2_M) { r.v = bid[bb_beg]; r.pos = bb_beg; r.none = false; return r; }
        if (cls == E2_D) { bb_beg++; carry = 0; continue; }
        if (cls == E2_I) { r.v = gbase[bb_beg] + Cm[bb_beg] + carry; r.pos = bb_beg; r.bb = false; r.none = false; return r; }
    }
    return r;
}

__device__ __forceinline__ int e2_base_code(const uint32_t b) { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : -1; }
#define E2_KEY_NINS(k) ((((uint32_t)(k)) >> 6) + 1u)
#define E2_KEY_DELTA(k) ((((uint32_t)(k)) >> 4) & 3u)

// ---- k_emit2 -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emit2(DgParams p) {
    const uint32_t a = blockIdx.y;
    if (dg_failed(p) || dg_askip(p, a)) return;
    const uint32_t t = p.aln_tgt[a];
    if (dg_tskip(p, t)) return;
    const uint32_t lo = p.n_lo[a], hi = p.n_hi[a];
    const uint32_t n = hi > lo ? hi - lo : 0u;
    const uint32_t nblk = (n + 63u) / 64u;
    const uint32_t blk = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (blk >= (nblk ? nblk : 1u)) return;
    const int lane = threadIdx.x & 63;
    const uint32_t r = (uint32_t)(a - p.aln_begin[t]);
    const uint32_t blen = p.tlen[t], exitpos = blen + 1u;
    const uint64_t bv = p.bbv_base[t];
    const uint32_t *bid = p.bid + bv, *gbase = p.gbase + bv;
    const uint32_t stride = p.matc_stride[t];
    const uint64_t rowoff = p.matc_base[t] + (uint64_t)r * stride;
    const uint32_t *Cm = p.matC + rowoff;
    uint32_t *Am = p.matA + rowoff, *Dm = p.matD + rowoff;
    uint8_t *Km = p.matK + rowoff;
    uint32_t *pool = p.pool + p.pool_base[t];
    DgNode *ndt = p.nodes + p.node_base[t];
    const uint16_t *buf = p.norm + p.norm_off[a];
    const uint32_t start = p.n_start[a];
    const uint32_t exit_id = bid[exitpos];

    if (n == 0) {
        // AlnGraphBoost.cpp:106 alone: enter -> exit; every cell of the row is empty
        for (uint32_t x = lane; x <= exitpos; x += 64) { Am[x] = x == exitpos ? 1u : 0u; Dm[x] = x == 0 ? exit_id + 1u : 0u; Km[x] = 0; }
        return;
    }
    const uint32_t i0 = lo + 64u * blk, i = i0 + (uint32_t)lane;
    const bool in = i < hi;
    const uint16_t c = in ? buf[i] : (uint16_t)0;
    const bool has_back = blk > 0;
    const uint16_t cbk = has_back ? buf[i - 64u] : (uint16_t)0;
    const bool ina = i + 64u < hi;
    const uint16_t cah = ina ? buf[i + 64u] : (uint16_t)0;
    const uint32_t bb0 = p.bbstart[(uint64_t)a * p.bs_stride + blk];
    const int cls = e2_class(c, in), clsb = e2_class(cbk, has_back), clsa = e2_class(cah, ina);
    const unsigned long long advM = __ballot(cls == E2_M || cls == E2_D), IM = __ballot(cls == E2_I),
                             VM = __ballot(cls == E2_M || cls == E2_I), MM = __ballot(cls == E2_M);
    const unsigned long long advB = __ballot(clsb == E2_M || clsb == E2_D), IB = __ballot(clsb == E2_I),
                             VB = __ballot(clsb == E2_M || clsb == E2_I), MB = __ballot(clsb == E2_M);
    const unsigned long long advA = __ballot(clsa == E2_M || clsa == E2_D), IA = __ballot(clsa == E2_I),
                             VA = __ballot(clsa == E2_M || clsa == E2_I), MA = __ballot(clsa == E2_M);
    const unsigned long long below = e2_below(lane);
    const uint32_t bbpos = bb0 + (uint32_t)__popcll(advM & below);

    // insertion columns between the last match / deletion column in front of the wave and the wave
    uint32_t carry_in = 0;
    if (has_back) {
        if (advB) { const int la = e2_hi(advB); carry_in = (uint32_t)__popcll(IB & e2_from(la + 1)); }
        else carry_in = (uint32_t)__popcll(IB) + (blk > 1 ? e2_count_ins_back(buf, lo, i0 - 64u) : 0u);
    }
    // an insertion column's rank among its position's insertion columns of this read (AlnGraphBoost.cpp:96: vertices
    // are numbered in (read, column) order)
    uint32_t idx = 0;
    {
        const unsigned long long lowadv = advM & below;
        if (lowadv) idx = (uint32_t)__popcll(IM & below & e2_from(e2_hi(lowadv) + 1));
        else idx = (uint32_t)__popcll(IM & below) + carry_in;
    }
    uint32_t v = 0;
    if (cls == E2_M) v = bid[bbpos];
    else if (cls == E2_I) v = gbase[bbpos] + Cm[bbpos] + idx;

    // ---- the vertex in front of the wave's first vertex ----
    uint32_t pin_v = 0, pin_pos = 0;
    bool pin_bb = true, pin_enter = true;
    if (has_back && VM) {
        if (VB) {
            const int lv = e2_hi(VB);
            const uint32_t bb_lv = bb0 - (uint32_t)__popcll(advB & e2_from(lv));
            pin_enter = false; pin_pos = bb_lv;
            if ((MB >> lv) & 1ull) { pin_v = bid[bb_lv]; pin_bb = true; }
            else {
                const unsigned long long lowadvB = advB & e2_below(lv);
                uint32_t idxI;
                if (lowadvB) idxI = (uint32_t)__popcll(IB & e2_below(lv) & e2_from(e2_hi(lowadvB) + 1));
                else idxI = (uint32_t)__popcll(IB & e2_below(lv)) + (blk > 1 ? e2_count_ins_back(buf, lo, i0 - 64u) : 0u);
                pin_v = gbase[bb_lv] + Cm[bb_lv] + idxI; pin_bb = false;
            }
        } else if (blk > 1) {
            const E2Vtx f = e2_find_prev(buf, lo, i0 - 64u, bb0 - (uint32_t)__popcll(advB), bid, gbase, Cm);
            pin_v = f.v; pin_pos = f.pos; pin_bb = f.bb; pin_enter = f.none;
        }
    }
    // ---- the vertex behind the wave's last vertex ----
    uint32_t nout_v = exit_id;
    bool nout_exit = true;
    if (VM) {
        const uint32_t bb_after = bb0 + (uint32_t)__popcll(advM);
        uint32_t tail;                              // insertion columns behind the wave's last match / deletion column
        if (advM) tail = (uint32_t)__popcll(IM & e2_from(e2_hi(advM) + 1)); else tail = (uint32_t)__popcll(IM) + carry_in;
        if (VA) {
            const int fv = __ffsll((long long)VA) - 1;
            const unsigned long long lowadvA = advA & e2_below(fv);
            const uint32_t bbA = bb_after + (uint32_t)__popcll(lowadvA);
            nout_exit = false;
            if ((MA >> fv) & 1ull) nout_v = bid[bbA];
            else nout_v = gbase[bbA] + Cm[bbA] + (lowadvA ? 0u : tail);
        } else if (i0 + 128u < hi) {
            const E2Vtx f = e2_find_next(buf, hi, i0 + 128u, bb_after + (uint32_t)__popcll(advA), advA ? 0u : tail, bid, gbase, Cm);
            if (!f.none) { nout_v = f.v; nout_exit = false; }
        }
    }

    // ---- every column writes what is its own ----
    const bool isV = cls == E2_M || cls == E2_I;
    uint32_t prev = 0, nxt = exit_id, prev_pos = 0;
    bool prev_bb = true, first_v = false, last_v = false;
    {
        const unsigned long long lowV = VM & below;
        const int pl = lowV ? e2_hi(lowV) : 0;
        const uint32_t sv = (uint32_t)__shfl((int)v, pl), sp = (uint32_t)__shfl((int)bbpos, pl);
        if (lowV) { prev = sv; prev_pos = sp; prev_bb = (MM >> pl) & 1ull; }
        else { prev = pin_v; prev_pos = pin_pos; prev_bb = pin_bb; first_v = pin_enter; }
        const unsigned long long highV = lane == 63 ? 0ull : VM & ~e2_below(lane + 1);
        const int nl = highV ? __ffsll((long long)highV) - 1 : 0;
        const uint32_t sn = (uint32_t)__shfl((int)v, nl);
        if (highV) nxt = sn; else { nxt = nout_v; last_v = nout_exit; }
    }
    // the key of a short insertion chain, at the match column that closes it (k_dedupe): one or two inserted bases
    // right in front of this column, a backbone vertex one to three positions back in front of them
    uint32_t key = 0;
    {
        const unsigned long long nonI = ~IM & below;
        const int s = nonI ? e2_hi(nonI) + 1 : 0;                     // first lane of the insertion run that ends at lane - 1
        const int nins = lane - s;
        const unsigned long long lowV2 = VM & e2_below(s);
        const int pl2 = lowV2 ? e2_hi(lowV2) : 0;
        const uint32_t apos = (uint32_t)__shfl((int)bbpos, pl2);
        const uint32_t b1 = (uint32_t)__shfl((int)DG_Q(c), s), b2 = (uint32_t)__shfl((int)DG_Q(c), s + 1 < 64 ? s + 1 : 63);
        if (p.fold && cls == E2_M && nins >= 1 && nins <= 2 && s > 0 && lowV2 && ((MM >> pl2) & 1ull)) {
            const uint32_t delta = bbpos - apos;
            const int c1 = e2_base_code(b1), c2 = nins == 2 ? e2_base_code(b2) : 0;
            if (delta >= 1u && delta <= 3u && c1 >= 0 && c2 >= 0)
                key = ((uint32_t)(nins - 1) << 6) | (delta << 4) | ((uint32_t)c1 << 2) | (uint32_t)c2;
        }
    }
    if (cls == E2_M) {
        Am[bbpos] = ((uint32_t)DG_T(c) << 25) | (prev + 1u);           // :75-85
        Dm[bbpos] = nxt + 1u;
        Km[bbpos] = (uint8_t)key;
    } else if (cls == E2_D) {
        Am[bbpos] = ((uint32_t)DG_T(c) << 25) | DG_CELL_DEL;           // :87-93
        Dm[bbpos] = 0u;
        Km[bbpos] = 0;
    } else if (cls == E2_I) {                                          // :95-104
        const uint32_t rk = v - bbpos;                                 // bbpos backbone vertices precede group bbpos
        DgNode nd;
        nd.out_len = 1; nd.in_len = 1; nd.base = DG_Q(c); nd.flags = 0; nd.pad = 0;
        nd.weight = 1; nd.pending = 1;
        nd.out_off = 3u * rk; nd.in_off = 3u * rk + 2u; nd.out_cap = 1; nd.in_cap = 1;
        nd.bbpos = (int32_t)bbpos;
        ndt[v] = nd;
        pool[3u * rk] = nxt; pool[3u * rk + 1u] = 1u; pool[3u * rk + 2u] = prev;
    }
    if (isV && first_v) Dm[0] = v + 1u;                                // the read's first vertex: enter's out-edge
    if (isV && last_v) Am[exitpos] = v + 1u;                           // :106
    // the rest of the row: no cell in front of the read's first position, none behind its last
    if (blk == 0) {
        for (uint32_t x = (uint32_t)lane; x < start && x <= exitpos; x += 64) { Am[x] = 0u; if (x) Dm[x] = 0u; Km[x] = 0; }
        // (a read without a vertex -- nothing but deletions: enter -> exit)
        if (!VM) {
            bool none = !VA;
            if (none && i0 + 128u < hi) none = e2_find_next(buf, hi, i0 + 128u, 0u, 0u, bid, gbase, Cm).none;
            if (none && lane == 0) { Dm[0] = exit_id + 1u; Am[exitpos] = 1u; }
        }
    }
    if (blk + 1 == nblk) {
        const uint32_t endp = bb0 + (uint32_t)__popcll(advM);          // first position behind the read
        for (uint32_t x = endp + (uint32_t)lane; x <= exitpos; x += 64) { if (x < exitpos) Am[x] = 0u; Dm[x] = 0u; Km[x] = 0; }
    }
}

// ---- k_dedupe ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dedupe(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blen = p.tlen[t];
    const uint32_t K = (uint32_t)(p.aln_begin[t + 1] - p.aln_begin[t]);
    const uint32_t stride = p.matc_stride[t];
    const uint64_t mb = p.matc_base[t];
    uint32_t *pool = p.pool + p.pool_base[t];
    DgNode *ndt = p.nodes + p.node_base[t];
    const uint32_t pos0 = (blockIdx.y * 4 + wave) * 8u;
    if (pos0 > blen) return;
    for (uint32_t r0 = 0; r0 < K; r0 += 64) {
        const uint32_t r = r0 + (uint32_t)lane;
        const uint64_t row = mb + (uint64_t)r * stride;
        uint2 kk = make_uint2(0u, 0u);
        if (r < K) kk = *reinterpret_cast<const uint2 *>(p.matK + row + pos0);       // rows are 32-byte aligned, pos0 a multiple of 8
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t pos = pos0 + (uint32_t)j;
            const uint32_t key = ((j < 4 ? kk.x : kk.y) >> (8 * (j & 3))) & 0xFFu;
            unsigned long long em = __ballot(key != 0u);
            if (!(em & (em - 1ull)) || pos > blen) continue;
            while (em & (em - 1ull)) {
                const int f = __ffsll((long long)em) - 1;
                const uint32_t kf = (uint32_t)__builtin_amdgcn_readlane((int)key, f);
                const unsigned long long same = __ballot(key == kf);
                em &= ~same;
                const uint32_t n = (uint32_t)__popcll(same);
                if (n < 2 || !((same >> lane) & 1ull)) continue;
                const uint32_t nins = E2_KEY_NINS(key), delta = E2_KEY_DELTA(key);
                const uint32_t ac = p.matA[row + pos];
                const uint32_t leaf = DG_CELL_ID(ac) - 1u, first = leaf - (nins - 1u);
                if (lane == f) {
                    for (uint32_t k = 0; k < nins; k++) {
                        const uint32_t id = first + k, rk = id - pos;
                        ndt[id].weight = (int32_t)n;
                        pool[3u * rk + 1u] = n;
                    }
                    p.matD[row + pos - delta] = (first + 1u) | ((n - 1u) << 25);
                } else {
                    for (uint32_t k = 0; k < nins; k++) {
                        const uint32_t code = nins == 2 && k == 0 ? (key >> 2) & 3u : nins == 2 ? key & 3u : (key >> 2) & 3u;
                        const uint32_t base = code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : 'T';
                        *reinterpret_cast<uint2 *>(&ndt[first + k]) = make_uint2(0u, base | (DG_NF_DELETED << 8));   // :269-273
                    }
                    p.matD[row + pos - delta] = 0u;
                    p.matA[row + pos] = (ac & 0xFE000000u) | DG_CELL_DUP;
                }
            }
        }
    }
}
