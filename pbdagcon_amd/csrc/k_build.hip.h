// k_build.hip.h -- stage (a): alignment strings -> alignment DAG in HBM.
//
//   k_norm_*     a1  normalizeGaps (Alignment.cpp:131-217) + trimAln (:219-242) in chunks of ~1 k
//                    input columns, one lane per chunk; also records the insertion run length
//                    per (position, read).  k_normalize_slow: whole alignments, sequential.
//   k_carve      a2  exact vertex / pool needs per target, exclusive scans -> arena offsets
//   k_groups     a2  per backbone position: exclusive scan of insertion run lengths over reads
//   k_gscan      a2  per target: exclusive scan over positions -> position-ordered vertex ids
//   k_emit       a2  addAln (AlnGraphBoost.cpp:64-107): one lane per alignment walks its columns;
//                    plain stores only (arrival / departure cells, inserted vertex records)
//   k_lists      a2  backbone vertices (AlnGraphBoost.cpp:16-62), addEdge dedupe (:109-127) + coverage / weight / base:
//                    one wave per backbone position turns its arrival / departure row into
//                    ordered adjacency lists (ballot / popcount peeling)
#pragma once
#include <hip/hip_runtime.h>
#include "dagcon_dev.h"

#define DG_WAVE 64
#ifndef DG_LPW
#define DG_LPW 8u             // positions per wave of k_lists / k_groups
#endif

__device__ __forceinline__ bool dg_failed(const DgParams &p) {
    return __hip_atomic_load(&p.st->err_flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
__device__ __forceinline__ void dg_fail(const DgParams &p, uint32_t bit) {
    atomicOr(&p.st->err_flags, bit);
}
// a failure that is confined to target t (the reference's asserts / undefined behaviour hit one
// worker's one target, AlnGraphBoost.cpp:71-72): the target is dropped, the batch completes
__device__ __forceinline__ void dg_fail_target(const DgParams &p, uint32_t t, uint32_t bit) {
    atomicOr(&p.tfail[t], bit);
}
__device__ __forceinline__ bool dg_tskip(const DgParams &p, uint32_t t) {
    return !p.tactive[t] || __hip_atomic_load(&p.tfail[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
// alignment-level kernels: alignments of a failed target are not worth another instruction
__device__ __forceinline__ bool dg_askip(const DgParams &p, uint32_t a) {
    return !(p.flags & DG_F_A1_ONLY) && __hip_atomic_load(&p.tfail[p.aln_tgt[a]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
__device__ __forceinline__ void dg_fail_aln(const DgParams &p, uint32_t a, uint32_t bit) {
    if (p.flags & DG_F_A1_ONLY) dg_fail(p, bit);       // dagcon_normalize: no targets, the call fails
    else dg_fail_target(p, p.aln_tgt[a], bit);
    p.st->bad_aln = a;
}

// ---------------------------------------------------------------------------
// normalizeGaps.  Columns are uint16: low byte = query char, high byte = target char.
//
// The rewrite streams: 16 input columns per 16-byte load, expanded into a per-lane LDS
// window; the push loop (Alignment.cpp:165-198) runs on the window with its two monotone
// look-ahead cursors and pauses when a look-ahead reaches the end of the window (its steps
// are idempotent, see dg_norm_run); finished columns leave 8 at a time.  An alignment whose
// look-ahead outgrows the window (a gap run of ~100 columns) is flagged and redone by
// k_normalize_slow, the same algorithm on HBM.
// ---------------------------------------------------------------------------
#define DG_COL(qb, tb) ((uint16_t)((uint16_t)(qb) | ((uint16_t)(tb) << 8)))
#define DG_Q(c) ((uint8_t)((c) & 0xff))
#define DG_T(c) ((uint8_t)((c) >> 8))
#define DG_NW 64u             // LDS window of the first pass: columns per lane (power of two); the lane
                              // row is NW + 2 uint16 = an odd number of dwords, so lanes spread over banks
#define DG_NW_BIG 512u        // window of the second pass, for the chunks whose look-ahead outgrew the first
#define DG_REDO 0xFFFFFFFFu   // n_hi marker: redo on the slow path

// trimAln, column counts, conformity, insertion runs: what follows normalizeGaps for
// both kernels.  buf holds the m final columns.
__device__ inline void dg_finish_alignment(const DgParams &p, uint32_t a, uint16_t *buf, uint32_t m) {
    // Alignment.cpp:219-242 trimAln (a no-op for trim == 0)
    const uint32_t trim = p.trim;
    uint32_t lbases = 0, rbases = 0, lo = 0, hi = m;
    uint32_t start = p.aln_start[a];
    while (lbases < trim && lo < m) {
        if (DG_T(buf[lo++]) != DG_GAP) lbases++;
    }
    while (rbases < trim && hi > lo) {
        if (DG_T(buf[--hi]) != DG_GAP) rbases++;
    }
    start += lbases;
    // what addAln will do with the window: insertions create vertices, matches and
    // deletions advance the backbone cursor (AlnGraphBoost.cpp:75-104); the insertion run
    // length per (position, read) numbers the inserted vertices
    const bool graph = !(p.flags & DG_F_A1_ONLY);
    uint32_t t_idx = 0, r = 0;
    uint32_t *Cm = nullptr;
    if (graph) {
        t_idx = p.aln_tgt[a];
        if (p.tactive[t_idx]) {
            const uint64_t ab = p.aln_begin[t_idx];
            r = (uint32_t)(a - ab);
            Cm = p.matC + p.matc_base[t_idx] + (uint64_t)r * p.matc_stride[t_idx];   // the read's row
        }
    }
    const uint32_t tlen = graph ? p.tlen[t_idx] : 0xFFFFFFFFu;
    uint32_t n_ins = 0, n_del = 0, adv = 0, run = 0;
    bool conf = start >= 1;
    uint32_t i = lo;
    // 8 columns per 16-byte load once the index is a multiple of 8 (buffers are 16-byte aligned)
#define DG_FIN_COL(c, ci_)                                                                \
    do {                                                                                  \
        const uint8_t qb_ = DG_Q(c), tb_ = DG_T(c);                                       \
        if (qb_ == tb_ || qb_ == DG_GAP) {                                                \
            if (Cm && conf && ((start + adv) & ((1u << p.emit_shift) - 1u)) == 0 && start + adv <= tlen + 1) \
                ck[(start + adv) >> p.emit_shift] = ci_ - run;                            \
            if (run) { if (Cm && conf && start + adv <= tlen + 1) Cm[start + adv] = run; run = 0; } \
            adv++; n_del += (qb_ != tb_);                                                 \
            if ((uint64_t)start - 1 + adv > (uint64_t)tlen) conf = false;                 \
        } else if (tb_ == DG_GAP) { n_ins++; run++; }                                     \
    } while (0)
    uint32_t *ck = p.ckpt + p.ck_base[a];
    while (i < hi && (i & 7u)) { const uint16_t c = buf[i]; DG_FIN_COL(c, i); i++; }
    while (i + 8 <= hi) {
        const uint4 v = *reinterpret_cast<const uint4 *>(buf + i);
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint16_t c0 = (uint16_t)(w4[k] & 0xffffu), c1 = (uint16_t)(w4[k] >> 16);
            DG_FIN_COL(c0, i + 2 * k);
            DG_FIN_COL(c1, i + 2 * k + 1);
        }
        i += 8;
    }
    while (i < hi) { const uint16_t c = buf[i]; DG_FIN_COL(c, i); i++; }
    if (run && Cm && conf && ((start + adv) & ((1u << p.emit_shift) - 1u)) == 0 && start + adv <= tlen + 1)
        ck[(start + adv) >> p.emit_shift] = hi - run;     // a trailing insertion run: the position after the read's last
    if (run && Cm && conf && start + adv <= tlen + 1) Cm[start + adv] = run;
#undef DG_FIN_COL
    p.n_lo[a] = lo; p.n_hi[a] = hi; p.n_start[a] = start;
    p.n_ins[a] = n_ins; p.n_del[a] = n_del;
    atomicAdd(&p.st->n_columns, (unsigned long long)(hi - lo));
    if (graph && hi > lo && !conf) dg_fail_aln(p, a, DG_E_NONCONF);
}

// ---------------------------------------------------------------------------
// Chunked normalizeGaps: k_norm_chunk -> k_norm_scan -> k_norm_finish.
//
// The gap push only ever changes columns to the right of the one it is working on, so the
// rewrite of an alignment can be cut at any column k that no earlier step has touched when
// its turn comes ("clean"): steps >= k then see exactly the input.  Whether k is clean is
// only known to the run that arrives there, so every chunk starts cold at a column chosen
// from the input alone (three matches in a row, the last two with different bases: a gap in
// flight rarely gets through those) and the chunk in front of it checks, at its end, that it
// has not written past it.  By induction from chunk 0 all chunks are then exact.  A chunk
// that did write past its end is run again with the next chunk taken in (its output goes to
// the re-run region), and the swallowed chunk's own result is dropped by k_norm_scan.
//   k_norm_chunk   lane per chunk: the streaming push loop (dg_norm_run) on [k0, k1)
//   k_norm_scan    lane per alignment: offsets of the chunks, trimAln (:219-242)
//   k_norm_finish  lane per chunk: columns to their final place; what dg_finish_alignment
//                  does, on the chunk's share of the trimmed window
// ---------------------------------------------------------------------------
#ifndef DG_NCH
#define DG_NCH 512u            // input columns per window (1024: +1.4 ms at configs[1], 256: the same, 128: +1.3 ms)
#endif
#define DG_CH_NONE 0xFFFFFFFFu

__device__ __forceinline__ bool dg_match_col(uint8_t qb, uint8_t tb) { return qb == tb && qb != DG_GAP && qb != '.'; }

// column a chunk starts at inside window c of the alignment, DG_CH_NONE if there is none
__device__ inline uint32_t dg_chunk_start(const uint8_t *q, const uint8_t *t, uint32_t len, uint32_t c) {
    if (c == 0) return 0;
    const uint64_t w0 = (uint64_t)c * DG_NCH;
    if (w0 >= len) return DG_CH_NONE;
    const uint32_t hi = (uint64_t)len < w0 + DG_NCH ? len : (uint32_t)(w0 + DG_NCH);
    for (uint32_t k = (uint32_t)w0; k < hi; k++) {
        const uint8_t b = q[k];
        if (dg_match_col(b, t[k]) && b != q[k - 1] && dg_match_col(q[k - 1], t[k - 1]) && dg_match_col(q[k - 2], t[k - 2]))
            return k;
    }
    return DG_CH_NONE;
}

struct DgChunkRun { uint32_t w, tb; bool dirty, overflow, badchar; };

// normalizeGaps (Alignment.cpp:142-214) on the input columns [k0, k1) of an alignment, started
// cold; the look-ahead may read (and, reported as `dirty`, write) beyond k1.
template <uint32_t NW>
__device__ inline DgChunkRun dg_norm_run(const uint8_t *q, const uint8_t *t, const uint32_t len, const uint32_t k0,
                                         const uint32_t k1, uint16_t *win, uint16_t *out) {
#define DG_W(x) win[(x) & (NW - 1u)]
    DgChunkRun r;
    r.w = 0; r.tb = 0; r.dirty = false; r.overflow = false; r.badchar = false;
    if (k0 >= k1) return r;
    uint32_t badw = 0;                                     // bit 7 of a byte set: a byte outside 33..126 was read
    uint32_t ip = k0, e = 0, i = 0, w = 0, tb = 0, jt = 0, jq = 0;
    uint32_t e_end = 0xFFFFFFFFu;                          // window index of input column k1, once known
    // finished columns collect in a 128-bit shift register and leave 8 at a time (out is
    // 16-byte aligned): one store request instead of eight
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;
    // (flags that differ from lane to lane are kept as integers in vector registers, not as booleans: see k_emit)
    uint32_t in_done = 0, dirty = 0, overflow = 0;
    for (;;) {
        asm volatile("" : "+v"(in_done), "+v"(dirty));
        // ---- refill: Alignment.cpp:142-159 on the next (up to) 16 input columns ----
        if (!in_done && (e - i) + 32u > NW) { overflow = 1; break; }
        if (!in_done) {
            uint32_t take = len - ip;
            if (take > 16u) take = 16u;
            if (ip < k1 && take > k1 - ip) take = k1 - ip;  // land on the chunk's end exactly
#define DG_EXPAND(QB, TB)                                                      \
            do {                                                               \
                uint8_t qb_ = (QB), tb_ = (TB);                                \
                if (qb_ == '.') qb_ = DG_GAP;                                  \
                if (tb_ == '.') tb_ = DG_GAP;                                  \
                /* a mismatch becomes (-, t) (q, -): no branch -- the second slot is written whatever the column is */ \
                /* (the next column overwrites it; the refill has 32 free slots for its 16 columns) */                   \
                const bool mm_ = qb_ != tb_ && qb_ != DG_GAP && tb_ != DG_GAP; \
                DG_W(e) = mm_ ? DG_COL(DG_GAP, tb_) : DG_COL(qb_, tb_);        \
                DG_W(e + 1u) = DG_COL(qb_, DG_GAP);                            \
                e += mm_ ? 2u : 1u;                                            \
            } while (0)
            if (take == 16u && (((uintptr_t)(q + ip)) & 15u) == 0) {
                // the common case, unrolled: bytes come out of the two 16-byte registers with
                // constant shifts
                const uint4 qv = *reinterpret_cast<const uint4 *>(q + ip);
                const uint4 tv = *reinterpret_cast<const uint4 *>(t + ip);
                const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w}, tw[4] = {tv.x, tv.y, tv.z, tv.w};
                // every byte has to be printable ASCII (33..126): four at a time
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    badw |= ((qw[k] - 0x21212121u) & ~qw[k]) | ((qw[k] + 0x01010101u) | qw[k]);
                    badw |= ((tw[k] - 0x21212121u) & ~tw[k]) | ((tw[k] + 0x01010101u) | tw[k]);
                }
#pragma unroll
                for (int k = 0; k < 16; k++)
                    DG_EXPAND((uint8_t)(qw[k >> 2] >> (8 * (k & 3))), (uint8_t)(tw[k >> 2] >> (8 * (k & 3))));
            } else {
                // head (up to the next 16-byte boundary) and tail: byte loads
                const uint32_t to_align = (uint32_t)((16u - (((uintptr_t)(q + ip)) & 15u)) & 15u);
                if (to_align && take > to_align) take = to_align;
                for (uint32_t k = 0; k < take; k++) {
                    const uint8_t qb0 = q[ip + k], tb0 = t[ip + k];
                    if (qb0 < 33 || qb0 > 126 || tb0 < 33 || tb0 > 126) badw = 0x80u;
                    DG_EXPAND(qb0, tb0);
                }
            }
#undef DG_EXPAND
            ip += take;
            if (ip == k1) e_end = e;
            if (ip == len) in_done = 1;
        }
        // ---- Alignment.cpp:165-198 push gaps to the right, as far as the window reaches.
        // jt / jq only move forward (a column left of a cursor is never turned back into
        // a base).  A step that runs out of window stores what it has done to column i
        // and is restarted after the refill: the t-pass of a restarted step either finds
        // its column already filled or repeats the same fruitless look-up. ----
        while (i < e && i < e_end) {
            if (i + 1 == e && !in_done) break;
            const uint16_t c = DG_W(i);
            uint8_t qi = DG_Q(c), ti = DG_T(c);
            uint32_t more = 0;
            if (i + 1 < e) {
                if (ti == DG_GAP) {
                    if (jt <= i) jt = i + 1;
                    while (jt < e && DG_T(DG_W(jt)) == DG_GAP) jt++;
                    if (jt < e) {
                        const uint16_t cj = DG_W(jt);
                        if (DG_T(cj) == qi) { ti = qi; DG_W(jt) = DG_COL(DG_Q(cj), DG_GAP); dirty |= (uint32_t)(jt >= e_end); }
                    } else if (!in_done) more = 2;
                }
                if (!more && qi == DG_GAP) {
                    if (jq <= i) jq = i + 1;
                    while (jq < e && DG_Q(DG_W(jq)) == DG_GAP) jq++;
                    if (jq < e) {
                        const uint16_t cj = DG_W(jq);
                        if (DG_Q(cj) == ti) { qi = ti; DG_W(jq) = DG_COL(DG_GAP, DG_T(cj)); dirty |= (uint32_t)(jq >= e_end); }
                    } else if (!in_done) more = 2;
                }
            }
            if (more) { DG_W(i) = DG_COL(qi, ti); break; }
            if (qi != DG_GAP || ti != DG_GAP) {                                // :209-214
                o0 = (o0 >> 16) | (o1 << 16); o1 = (o1 >> 16) | (o2 << 16); o2 = (o2 >> 16) | (o3 << 16);
                o3 = (o3 >> 16) | ((uint32_t)DG_COL(qi, ti) << 16);
                w++;
                tb += (ti != DG_GAP);
                if ((w & 7u) == 0) *reinterpret_cast<uint4 *>(out + w - 8) = make_uint4(o0, o1, o2, o3);
            }
            i++;
        }
        if (i == e_end) break;
    }
#undef DG_W
    // the last, partial group: its columns sit at the top of the register
    for (uint32_t k = w & 7u, x = w - (w & 7u); k > 0; k--, x++) {
        const uint32_t sh = 8u - k;                        // column x is sh places from the bottom
        const uint32_t word = sh >> 1;
        const uint32_t v = word == 0 ? o0 : word == 1 ? o1 : word == 2 ? o2 : o3;
        out[x] = (uint16_t)((sh & 1u) ? v >> 16 : v & 0xffffu);
    }
    r.w = w; r.tb = tb; r.dirty = dirty != 0; r.overflow = overflow != 0; r.badchar = (badw & 0x80808080u) != 0;
    return r;
}

// NW = 64 keeps the LDS footprint of the first pass at 8 KB per wave (the kernel is bound by waves in
// flight); a chunk whose look-ahead outgrows that window (a gap run of ~30 columns) is done again
// by the second pass (RETRY: LANES = 32 lanes per block, a 512-column window each), and only what
// outgrows that one too sends its alignment to k_normalize_slow.
template <uint32_t NW, uint32_t LANES, bool RETRY>
__global__ __launch_bounds__(LANES) void k_norm_chunk(DgParams p) {
    __shared__ uint16_t s_win[LANES * (NW + 2u)];
    // neighbouring chunks are ~1 KB of input (4 KB of scratch) apart: lanes of a wave take chunks
    // a whole grid apart instead, or their lines fight for the same few L1 sets and L2 channels
    const uint32_t g = threadIdx.x * gridDim.x + blockIdx.x;
    if (g >= p.n_chunks) return;
    if (dg_failed(p)) return;
    if (RETRY && p.ch_flag[g] != 2u) return;
    const uint32_t a = p.ch_aln[g];
    if (dg_askip(p, a)) { if (!RETRY) p.ch_flag[g] = 0; return; }
    const uint32_t c = g - p.ch_base[a], nwin = p.ch_base[a + 1] - p.ch_base[a];
    const uint64_t off = p.aln_off[a];
    const uint32_t len = p.aln_len[a];
    const uint8_t *q = p.q + off, *t = p.t + off;
    uint32_t flag = 0;
    DgChunkRun r;
    r.w = 0; r.tb = 0; r.dirty = false; r.overflow = false; r.badchar = false;
    uint32_t k0 = DG_CH_NONE, cn = c + 1;
    uint64_t src = 0;
    if (p.flags & DG_F_RAW) {                              // raw columns: the slow kernel copies them
        if (c == 0) { k0 = 0; cn = nwin; flag = 1; }
    } else {
        k0 = dg_chunk_start(q, t, len, c);
        if (k0 != DG_CH_NONE) {
            uint16_t *win = s_win + threadIdx.x * (NW + 2u);
            src = (2ull * (off + k0) + 8ull * g + 7ull) & ~7ull;     // 16-byte aligned, regions stay disjoint
            for (;;) {
                uint32_t k1 = len;
                for (; cn < nwin; cn++) {
                    const uint32_t s = dg_chunk_start(q, t, len, cn);
                    if (s != DG_CH_NONE) { k1 = s; break; }
                }
                r = dg_norm_run<NW>(q, t, len, k0, k1, win, p.norm_tmp + src);
                if (r.badchar) dg_fail_aln(p, a, DG_E_BADCHAR);
                if (r.overflow) { flag = RETRY ? 1u : 2u; break; }
                if (!r.dirty) break;
                // a gap in flight got past the end: once more with the next chunk taken in,
                // into a stretch of the re-run region
                cn++;
                uint32_t k2 = len;
                for (uint32_t x = cn; x < nwin; x++) {
                    const uint32_t s = dg_chunk_start(q, t, len, x);
                    if (s != DG_CH_NONE) { k2 = s; break; }
                }
                const unsigned long long need = (2ull * (k2 - k0) + 15ull) & ~7ull;
                const unsigned long long o = atomicAdd(&p.st->ovf_top, need);
                if (p.tmp_main + o + need > p.tmp_cap) { flag = 1; break; }      // out of room: slow path
                src = p.tmp_main + o;
            }
        }
    }
    p.ch_k0[g] = k0; p.ch_next[g] = cn; p.ch_w[g] = r.w; p.ch_tb[g] = r.tb;
    p.ch_flag[g] = flag; p.ch_src[g] = src;
}

__global__ __launch_bounds__(64) void k_norm_scan(DgParams p) {
    const uint32_t a = blockIdx.x * 64 + threadIdx.x;
    if (a >= p.A) return;
    if (dg_failed(p)) return;
    if (dg_askip(p, a)) return;
    const uint32_t g0 = p.ch_base[a], g1 = p.ch_base[a + 1];
    bool redo = false;
    uint32_t m = 0, tbt = 0;
    for (uint32_t g = g0; g < g1;) {
        if (p.ch_k0[g] == DG_CH_NONE) { p.ch_out[g] = DG_CH_NONE; g++; continue; }
        redo |= p.ch_flag[g] != 0;                       // (1: too long even for the second pass, or no scratch left)
        p.ch_out[g] = m; p.ch_adv[g] = tbt;
        m += p.ch_w[g]; tbt += p.ch_tb[g];
        uint32_t nx = g0 + p.ch_next[g];
        if (nx > g1) nx = g1;
        for (uint32_t x = g + 1; x < nx; x++) p.ch_out[x] = DG_CH_NONE;       // empty or swallowed windows
        g = nx;
    }
    if (redo) { p.n_hi[a] = DG_REDO; return; }
    // Alignment.cpp:219-242 trimAln: whole chunks by their counts, the last one column by column
    const uint32_t trim = p.trim;
    uint32_t lbases = 0, rbases = 0, lo = 0, hi = m;
    for (uint32_t g = g0; g < g1 && lbases < trim && lo < m; g++) {
        if (p.ch_out[g] == DG_CH_NONE) continue;
        const uint32_t w = p.ch_w[g], tb = p.ch_tb[g];
        if (lbases + tb < trim) { lo += w; lbases += tb; continue; }
        const uint16_t *src = p.norm_tmp + p.ch_src[g];
        for (uint32_t x = 0; x < w && lbases < trim; x++) {
            if (DG_T(src[x]) != DG_GAP) lbases++;
            lo++;
        }
    }
    for (uint32_t g = g1; g > g0 && rbases < trim && hi > lo;) {
        g--;
        const uint32_t o = p.ch_out[g];
        if (o == DG_CH_NONE) continue;
        const uint32_t tb = p.ch_tb[g];
        if (o >= lo && rbases + tb < trim) { hi = o; rbases += tb; continue; }
        const uint16_t *src = p.norm_tmp + p.ch_src[g];
        while (rbases < trim && hi > lo && hi > o) {
            if (DG_T(src[--hi - o]) != DG_GAP) rbases++;
        }
    }
    p.n_lo[a] = lo; p.n_hi[a] = hi; p.n_start[a] = p.aln_start[a] + lbases; p.n_lb[a] = lbases;
    p.n_ins[a] = 0; p.n_del[a] = 0;
    atomicAdd(&p.st->n_columns, (unsigned long long)(hi - lo));
}

// 8 columns starting h columns into the 16 columns (A, B)
__device__ __forceinline__ uint4 dg_funnel_cols(const uint4 A, const uint4 B, const uint32_t h) {
    const bool w1 = (h >> 1) & 1u, w2 = (h >> 2) & 1u, half = h & 1u;
    const uint32_t u0 = w1 ? A.y : A.x, u1 = w1 ? A.z : A.y, u2 = w1 ? A.w : A.z, u3 = w1 ? B.x : A.w,
                   u4 = w1 ? B.y : B.x, u5 = w1 ? B.z : B.y, u6 = w1 ? B.w : B.z;
    const uint32_t t0 = w2 ? u2 : u0, t1 = w2 ? u3 : u1, t2 = w2 ? u4 : u2, t3 = w2 ? u5 : u3, t4 = w2 ? u6 : u4;
    uint4 r;
    r.x = half ? __builtin_amdgcn_alignbit(t1, t0, 16) : t0;
    r.y = half ? __builtin_amdgcn_alignbit(t2, t1, 16) : t1;
    r.z = half ? __builtin_amdgcn_alignbit(t3, t2, 16) : t2;
    r.w = half ? __builtin_amdgcn_alignbit(t4, t3, 16) : t3;
    return r;
}

__global__ __launch_bounds__(64) void k_norm_finish(DgParams p) {
    // neighbouring chunks are ~1 KB of input (4 KB of scratch) apart: lanes of a wave take chunks
    // a whole grid apart instead, or their lines fight for the same few L1 sets and L2 channels
    const uint32_t g = threadIdx.x * gridDim.x + blockIdx.x;
    if (g >= p.n_chunks) return;
    if (dg_failed(p)) return;
    const uint32_t a = p.ch_aln[g];
    if (dg_askip(p, a)) return;
    const uint32_t o = p.ch_out[g];
    if (o == DG_CH_NONE) return;
    const uint32_t hi = p.n_hi[a];
    if (hi == DG_REDO) return;
    const uint32_t lo = p.n_lo[a], start = p.n_start[a], lb = p.n_lb[a];
    const uint32_t w = p.ch_w[g], adv0 = p.ch_adv[g];
    const uint16_t *src = p.norm_tmp + p.ch_src[g];                  // 16-byte aligned
    const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
    uint16_t *dst = p.norm + p.norm_off[a] + o;                       // norm_off is a multiple of 8 columns
    // what addAln will do with the window (see dg_finish_alignment)
    const bool graph = !(p.flags & DG_F_A1_ONLY);
    uint32_t t_idx = 0, r = 0;
    uint32_t *Cm = nullptr;
    if (graph) {
        t_idx = p.aln_tgt[a];
        if (p.tactive[t_idx]) {
            const uint64_t ab = p.aln_begin[t_idx];
            r = (uint32_t)(a - ab);
            Cm = p.matC + p.matc_base[t_idx] + (uint64_t)r * p.matc_stride[t_idx];   // the read's row
        }
    }
    const uint32_t tlen = graph ? p.tlen[t_idx] : 0xFFFFFFFFu;
    uint32_t *ck = p.ckpt + p.ck_base[a];
    const uint32_t ck_mask = (1u << p.emit_shift) - 1u;
    // target bases between the trimmed start and this chunk (a chunk in front of lo has none)
    uint32_t adv = adv0 > lb ? adv0 - lb : 0u;
    uint32_t n_ins = 0, n_del = 0, run = 0;
    bool conf = start >= 1 && !((uint64_t)start - 1 + adv > (uint64_t)tlen && adv > 0);
    // the chunk's share of the trimmed window, as chunk-relative column numbers
    const uint32_t f0 = lo > o ? lo - o : 0u;
    const uint32_t f1 = hi > o ? (hi - o < w ? hi - o : w) : 0u;
    const bool any = f1 > f0;
    const uint32_t fspan = any ? f1 - f0 : 0u;

    // columns go to their final place as aligned 16-byte stores: h single columns up to the
    // destination's next 16-byte line, then blocks funnelled out of two source vectors
    const uint32_t h0 = (8u - (o & 7u)) & 7u;
    const uint32_t h = h0 < w ? h0 : w;
    const uint32_t nb = (w - h) / 8u;
    for (uint32_t x = 0; x < h; x++) dst[x] = src[x];
    uint4 *dst4 = reinterpret_cast<uint4 *>(dst + h);
    const uint32_t nvec = (w + 7u) / 8u;
    uint4 prev = make_uint4(0, 0, 0, 0);
    // a whole 128-byte line of the chunk (8 vectors) is requested at once: taking it 16 bytes at
    // a time, with every lane on a line of its own, each line came in from L2 eight times
    uint4 lin[8];
    for (uint32_t v = 0; v < nvec; v++) {
        if ((v & 7u) == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) lin[k] = v + k < nvec ? src4[v + k] : make_uint4(0, 0, 0, 0);
        }
        uint4 cur = lin[0];
#pragma unroll
        for (int k = 1; k < 8; k++) if ((v & 7u) == (uint32_t)k) cur = lin[k];
        if (v >= 1 && v - 1 < nb) dst4[v - 1] = dg_funnel_cols(prev, cur, h);
        prev = cur;
        if (8u * v + 8u <= f0 || 8u * v >= f1) continue;         // nothing of the window in this vector
        const uint32_t w4[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint16_t c = (uint16_t)((k & 1) ? w4[k >> 1] >> 16 : w4[k >> 1] & 0xffffu);
            if (8u * v + (uint32_t)k - f0 >= fspan) continue;
            const uint8_t qb = DG_Q(c), tb = DG_T(c);
            if (qb == tb || qb == DG_GAP) {
                // (the first column of a chunk inside the window: the chunk in front records it,
                // it may end in the insertion run that belongs to this position)
                if (Cm && conf && ((start + adv) & ck_mask) == 0 && start + adv <= tlen + 1 && !(8u * v + (uint32_t)k == 0 && o > lo))
                    ck[(start + adv) >> p.emit_shift] = o + 8u * v + (uint32_t)k - run;
                if (run) { if (Cm && conf && start + adv <= tlen + 1) Cm[start + adv] = run; run = 0; }
                adv++; n_del += (qb != tb);
                if ((uint64_t)start - 1 + adv > (uint64_t)tlen) conf = false;
            } else if (tb == DG_GAP) { n_ins++; run++; }
        }
    }
    if (nvec >= 1 && nb == nvec) dst4[nb - 1] = dg_funnel_cols(prev, make_uint4(0, 0, 0, 0), h);   // h == 0, w % 8 == 0
    for (uint32_t x = h + 8u * nb; x < w; x++) dst[x] = src[x];
    // the column after the chunk is a match (the next chunk's first) or the end of the window;
    // a trailing insertion run of the read belongs to the position after its last one
    if (any && Cm && conf && ((start + adv) & ck_mask) == 0 && start + adv <= tlen + 1 && (o + f1 < hi || run))
        ck[(start + adv) >> p.emit_shift] = o + f1 - run;
    if (run && Cm && conf && start + adv <= tlen + 1) Cm[start + adv] = run;
    if (n_ins) atomicAdd(&p.n_ins[a], n_ins);
    if (n_del) atomicAdd(&p.n_del[a], n_del);
    if (graph && any && !conf) dg_fail_aln(p, a, DG_E_NONCONF);
}

// exclusive prefix sum over the lanes of a wave: four DPP row shifts inside the rows of 16 lanes, two row broadcasts
// across them (lanes without a source add 0); no LDS
__device__ __forceinline__ uint32_t dg_wave_excl(const uint32_t v, const int lane) {
    int incl = (int)v;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);     // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);     // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);     // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);     // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    (void)lane;
    return (uint32_t)incl - v;
}

// ---------------------------------------------------------------------------
// k_norm_finish2: what k_norm_finish does, a WAVE per chunk (round 3).  With a lane per chunk every lane streams lines
// of its own, a whole grid apart: 16-byte accesses that share nothing, half a million chunks in flight, their partly
// written lines (matC cells above all: 4 bytes here, 4 bytes there along a row) pushed out of the L2s before the next
// store to them arrives -- 4.7 GB of traffic for a 0.9 GB copy.  Here the 64 lanes take consecutive 16-byte pieces of ONE
// chunk (8 columns each, 512 per pass): the copy is coalesced loads and funnelled, aligned, coalesced stores; how
// many target bases lie in front of a lane's columns is a prefix sum over the lanes, how long the insertion run in
// front of them is a segmented one (reset at every lane that holds a match / deletion column); then every lane walks its
// own 8 columns as k_norm_finish walks them all.
// ---------------------------------------------------------------------------
#ifndef DG_NF2_V
#define DG_NF2_V 1            // (2: one pass per chunk instead of two, and no faster: 6.0 against 5.9 ms of normalize)
#endif
__global__ __launch_bounds__(256) void k_norm_finish2(DgParams p) {
    const uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= p.n_chunks) return;
    if (dg_failed(p)) return;
    const int lane = threadIdx.x & 63;
    const uint32_t a = p.ch_aln[g];
    if (dg_askip(p, a)) return;
    const uint32_t o = p.ch_out[g];
    if (o == DG_CH_NONE) return;
    const uint32_t hi = p.n_hi[a];
    if (hi == DG_REDO) return;
    const uint32_t lo = p.n_lo[a], start = p.n_start[a], lb = p.n_lb[a];
    const uint32_t w = p.ch_w[g], adv0 = p.ch_adv[g];
    const uint16_t *src = p.norm_tmp + p.ch_src[g];                  // 16-byte aligned
    const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
    uint16_t *dst = p.norm + p.norm_off[a] + o;                       // norm_off is a multiple of 8 columns
    const bool graph = !(p.flags & DG_F_A1_ONLY);
    uint32_t t_idx = 0;
    uint32_t *Cm = nullptr;
    if (graph) {
        t_idx = p.aln_tgt[a];
        if (p.tactive[t_idx]) Cm = p.matC + p.matc_base[t_idx] + (uint64_t)(a - p.aln_begin[t_idx]) * p.matc_stride[t_idx];   // the read's row
    }
    const uint32_t tlen = graph ? p.tlen[t_idx] : 0xFFFFFFFFu;
    uint32_t *ck = p.ckpt + p.ck_base[a];
    const uint32_t ck_mask = (1u << p.emit_shift) - 1u;
    const uint32_t adv_init = adv0 > lb ? adv0 - lb : 0u;            // target bases between the trimmed start and this chunk
    const uint32_t f0 = lo > o ? lo - o : 0u;
    const uint32_t f1 = hi > o ? (hi - o < w ? hi - o : w) : 0u;
    const bool any = f1 > f0;
    const uint32_t fspan = any ? f1 - f0 : 0u;
    // conformity as k_norm_finish keeps it: true while start - 1 + (target bases so far) <= tlen
#define DG_NF2_CONF(A) (start >= 1u && ((A) == 0u || (uint64_t)start - 1u + (A) <= (uint64_t)tlen))
    const uint32_t h0 = (8u - (o & 7u)) & 7u;
    const uint32_t h = h0 < w ? h0 : w;
    const uint32_t nb = (w - h) / 8u, nvec = (w + 7u) / 8u;
    // head and tail of the copy: single columns
    if ((uint32_t)lane < h) dst[lane] = src[lane];
    { const uint32_t x = h + 8u * nb + (uint32_t)lane; if (lane < 8 && x < w) dst[x] = src[x]; }
    uint4 *dst4 = reinterpret_cast<uint4 *>(dst + h);
    uint32_t adv_tile = adv_init, run_tile = 0, n_ins = 0, n_del = 0;
    // a lane takes DG_NF2_V consecutive 16-byte pieces: a chunk of 512 input columns is ~550 columns here, one pass of 1024
    constexpr uint32_t NV = DG_NF2_V, NC = 8u * NV;
    for (uint32_t v0 = 0; v0 < nvec; v0 += 64u * NV) {
        const uint32_t vl = v0 + NV * (uint32_t)lane;                // the lane's first piece
        uint4 pc[NV + 1];
#pragma unroll
        for (uint32_t j = 0; j <= NV; j++) pc[j] = vl + j < nvec ? src4[vl + j] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t j = 0; j < NV; j++) if (vl + j < nb) dst4[vl + j] = dg_funnel_cols(pc[j], pc[j + 1], h);
        if (!any || 8u * v0 >= f1) continue;                          // (uniform) nothing of the window from here on
        // ---- the lane's columns: classes, inside the trimmed window only ----
        uint32_t advm = 0, insm = 0, delm = 0;                        // bit k: column k advances the cursor / is an insertion / a deletion
#pragma unroll
        for (uint32_t k = 0; k < NC; k++) {
            const uint4 q4 = pc[k >> 3];
            const uint32_t wd = ((k >> 1) & 3u) == 0 ? q4.x : ((k >> 1) & 3u) == 1 ? q4.y : ((k >> 1) & 3u) == 2 ? q4.z : q4.w;
            const uint16_t c = (uint16_t)((k & 1u) ? wd >> 16 : wd & 0xffffu);
            const bool inw = 8u * vl + k < 8u * nvec && 8u * vl + k - f0 < fspan;
            const uint8_t qb = DG_Q(c), tb = DG_T(c);
            const bool adv = inw && (qb == tb || qb == DG_GAP);
            advm |= adv ? 1u << k : 0u;
            delm |= (adv && qb != tb) ? 1u << k : 0u;
            insm |= (inw && !adv && tb == DG_GAP) ? 1u << k : 0u;
        }
        const uint32_t nadv = (uint32_t)__popc(advm), nI = (uint32_t)__popc(insm);
        n_ins += nI; n_del += (uint32_t)__popc(delm);
        // target bases in front of the lane's columns
        const uint32_t adv_lane = adv_tile + dg_wave_excl(nadv, lane);
        // insertion columns since the last advancing column in front of the lane's columns: a scan of (reset, value)
        // pairs, (f1, v1) then (f2, v2) = (f1 | f2, f2 ? v2 : v1 + v2)
        uint32_t sf = advm ? 1u : 0u;
        uint32_t sv = advm ? (uint32_t)__popc(insm & ~((2u << (31 - __clz((int)advm))) - 1u)) : nI;      // behind its last advancing column
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t pf = (uint32_t)__shfl_up((int)sf, d), pv = (uint32_t)__shfl_up((int)sv, d);
            if (lane >= d) { sv = sf ? sv : pv + sv; sf |= pf; }
        }
        // exclusive: what the lane in front ends with (the pass's first lane: what the pass before ended with)
        uint32_t ef = (uint32_t)__shfl_up((int)sf, 1), ev = (uint32_t)__shfl_up((int)sv, 1);
        if (lane == 0) { ef = 0; ev = 0; }
        uint32_t run = ef ? ev : ev + run_tile;
        const uint32_t lf = (uint32_t)__builtin_amdgcn_readlane((int)sf, 63), lv = (uint32_t)__builtin_amdgcn_readlane((int)sv, 63);
        run_tile = lf ? lv : lv + run_tile;
        adv_tile += (uint32_t)__builtin_amdgcn_readlane((int)(adv_lane - adv_tile + nadv), 63);
        // ---- the lane walks its columns (k_norm_finish's loop body) ----
        uint32_t adv = adv_lane;
        if (Cm && (advm | insm)) {
#pragma unroll
            for (uint32_t k = 0; k < NC; k++) {
                if ((advm >> k) & 1u) {
                    const bool conf = DG_NF2_CONF(adv);
                    const uint32_t x = 8u * vl + k;
                    // (the first column of a chunk inside the window: the chunk in front records it,
                    // it may end in the insertion run that belongs to this position)
                    if (conf && ((start + adv) & ck_mask) == 0 && start + adv <= tlen + 1 && !(x == 0 && o > lo))
                        ck[(start + adv) >> p.emit_shift] = o + x - run;
                    if (run && conf && start + adv <= tlen + 1) Cm[start + adv] = run;
                    run = 0;
                    adv++;
                } else if ((insm >> k) & 1u) run++;
            }
        }
    }
    // the column after the chunk is a match (the next chunk's first) or the end of the window;
    // a trailing insertion run of the read belongs to the position after its last one
    {
        const uint32_t adv = adv_tile, run = run_tile;
        const bool conf = DG_NF2_CONF(adv);
        if (lane == 0) {
            if (any && Cm && conf && ((start + adv) & ck_mask) == 0 && start + adv <= tlen + 1 && (o + f1 < hi || run))
                ck[(start + adv) >> p.emit_shift] = o + f1 - run;
            if (run && Cm && conf && start + adv <= tlen + 1) Cm[start + adv] = run;
            if (graph && any && !conf) dg_fail_aln(p, a, DG_E_NONCONF);
        }
    }
    // totals of the chunk
    {
        const uint32_t ti = dg_wave_excl(n_ins, lane) + n_ins, td = dg_wave_excl(n_del, lane) + n_del;
        if (lane == 63) { if (ti) atomicAdd(&p.n_ins[a], ti); if (td) atomicAdd(&p.n_del[a], td); }
    }
#undef DG_NF2_CONF
}

// The same algorithm with the whole expanded alignment in HBM: raw mode and
// alignments whose look-ahead outgrew the LDS window.
__global__ __launch_bounds__(64) void k_normalize_slow(DgParams p) {
    const uint32_t a = blockIdx.x * 64 + threadIdx.x;
    if (a >= p.A) return;
    if (dg_failed(p)) return;
    if (dg_askip(p, a)) return;
    if (p.n_hi[a] != DG_REDO) return;
    const uint64_t off = p.aln_off[a];
    const uint32_t len = p.aln_len[a];
    const uint8_t *q = p.q + off, *t = p.t + off;
    uint16_t *buf = p.norm + p.norm_off[a];
    const bool raw = (p.flags & DG_F_RAW) != 0;
    uint32_t n = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint8_t qb = q[i], tb = t[i];
        if (qb < 33 || qb > 126 || tb < 33 || tb > 126) dg_fail_aln(p, a, DG_E_BADCHAR);
        if (!raw) {
            if (qb == '.') qb = DG_GAP;
            if (tb == '.') tb = DG_GAP;
            if (qb != tb && qb != DG_GAP && tb != DG_GAP) {
                buf[n++] = DG_COL(DG_GAP, tb);
                buf[n++] = DG_COL(qb, DG_GAP);
                continue;
            }
        }
        buf[n++] = DG_COL(qb, tb);
    }
    uint32_t m = n;
    if (!raw) {
        uint32_t w = 0, jt = 0, jq = 0;
        for (uint32_t i = 0; i < n; i++) {
            uint16_t c = buf[i];
            uint8_t qi = DG_Q(c), ti = DG_T(c);
            if (i + 1 < n) {
                if (ti == DG_GAP) {
                    if (jt <= i) jt = i + 1;
                    while (jt < n && DG_T(buf[jt]) == DG_GAP) jt++;
                    if (jt < n) {
                        uint16_t cj = buf[jt];
                        if (DG_T(cj) == qi) { ti = qi; buf[jt] = DG_COL(DG_Q(cj), DG_GAP); }
                    }
                }
                if (qi == DG_GAP) {
                    if (jq <= i) jq = i + 1;
                    while (jq < n && DG_Q(buf[jq]) == DG_GAP) jq++;
                    if (jq < n) {
                        uint16_t cj = buf[jq];
                        if (DG_Q(cj) == ti) { qi = ti; buf[jq] = DG_COL(DG_GAP, DG_T(cj)); }
                    }
                }
            }
            if (qi != DG_GAP || ti != DG_GAP) buf[w++] = DG_COL(qi, ti);
        }
        m = w;
    }
    dg_finish_alignment(p, a, buf, m);
}

// ---------------------------------------------------------------------------
// k_carve: a single 1024-thread block.  Exact sizes per target, then
// exclusive scans over targets for the arena offsets.
// Pool of a target: [inserted vertices: 3 words each, id order]
//                   [backbone positions: 3*capb words each] [growth region]
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dg_pool_words(uint32_t tlen, uint32_t n_ins, uint32_t n_del,
                                                  uint32_t k, uint32_t growth_pct, uint32_t *fixed_words) {
    const uint64_t capb = dg_capb(k);
    const uint64_t fixed = 3ull * n_ins + (uint64_t)(tlen + 2) * 3ull * capb;
    // rows that outgrow capb move to the growth region: every entry beyond the constructor
    // edge comes from an insertion run, a deletion run or a read end
    const uint64_t spill = 3ull * ((uint64_t)n_ins + n_del + 2ull * k) + 64ull;
    const uint64_t growth = (fixed * growth_pct) / 100ull + spill + 256ull;
    *fixed_words = (uint32_t)fixed;
    const uint64_t tot = fixed + growth;
    return tot > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)tot;
}

__global__ __launch_bounds__(1024) void k_carve(DgParams p) {
    __shared__ unsigned long long s_scan[1024];
    __shared__ unsigned long long s_carry_nodes, s_carry_pool;
    if (dg_failed(p)) return;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { s_carry_nodes = 0; s_carry_pool = 0; }
    __syncthreads();
    for (uint32_t t0 = 0; t0 < p.T; t0 += 1024) {
        const uint32_t t = t0 + tid;
        unsigned long long nodes = 0, poolw = 0;
        uint32_t fixed_words = 0, ins = 0;
        if (t < p.T && !dg_tskip(p, t)) {
            uint32_t del = 0;
            const uint64_t b = p.aln_begin[t], e = p.aln_begin[t + 1];
            for (uint64_t a = b; a < e; a++) { ins += p.n_ins[a]; del += p.n_del[a]; }
            nodes = (unsigned long long)p.tlen[t] + 2ull + ins;
            poolw = dg_pool_words(p.tlen[t], ins, del, (uint32_t)(e - b), p.growth_pct, &fixed_words);
            // (the sweeps address a target's pool with 32-bit byte offsets, arrival cells hold 25-bit ids)
            if (nodes > DG_MAX_NODES || poolw > 0x3FFFFFFFull) {
                dg_fail_target(p, t, DG_E_TOO_BIG);
                nodes = 0; poolw = 0; fixed_words = 0;
            }
        }
        s_scan[tid] = nodes;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            unsigned long long v = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        unsigned long long node_excl = s_scan[tid] - nodes + s_carry_nodes;
        unsigned long long node_tot = s_scan[1023];
        __syncthreads();
        s_scan[tid] = poolw;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            unsigned long long v = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        unsigned long long pool_excl = s_scan[tid] - poolw + s_carry_pool;
        unsigned long long pool_tot = s_scan[1023];
        __syncthreads();
        if (t < p.T) {
            p.node_base[t] = node_excl;
            p.n_nodes[t] = (uint32_t)nodes;
            p.pool_base[t] = pool_excl;
            p.pool_size[t] = (uint32_t)poolw;
            p.pool_top[t] = fixed_words;
            p.t_nins[t] = ins;
        }
        if (tid == 0) { s_carry_nodes += node_tot; s_carry_pool += pool_tot; }
        __syncthreads();
    }
    if (tid == 0) {
        p.st->node_need = s_carry_nodes;
        p.st->pool_need = s_carry_pool;
        if (s_carry_nodes > p.node_cap) dg_fail(p, DG_E_NODE_OVF);
        if (s_carry_pool > p.pool_cap) dg_fail(p, DG_E_POOL_OVF);
    }
}

// ---------------------------------------------------------------------------
// k_readspan: lane per alignment.  What the cuts of mergeNodes need to know about partial-span reads:
//   rd_lead  insertion columns in front of the read's first MATCH column: those vertices hang on enter
//            alone (addAln's `prev` is still enter: deletion columns in between do not move it);
//   rd_e     backbone position of the read's last match column, rd_trail the insertion columns behind it:
//            a chain that leads to exit only;  rd_s: position of the first match column.
// A read without a match column is one chain from enter to exit (all of it `lead`).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_readspan(DgParams p) {
    const uint32_t a = blockIdx.x * 64 + threadIdx.x;
    if (a >= p.A) return;
    if (dg_failed(p)) return;
    const uint32_t t = p.aln_tgt[a];
    uint32_t s = 0, e = 0, lead = 0, trail = 0;
    if (!dg_tskip(p, t)) {
        const uint32_t lo = p.n_lo[a], hi = p.n_hi[a];
        if (hi > lo && hi != DG_REDO) {
            const uint16_t *col = p.norm + p.norm_off[a];
            const uint32_t ins = p.n_ins[a];
            uint32_t i = lo, dels = 0;
            for (; i < hi; i++) {
                const uint8_t qb = DG_Q(col[i]), tb = DG_T(col[i]);
                if (qb == tb) break;
                if (tb == DG_GAP) lead++; else if (qb == DG_GAP) dels++;
            }
            s = p.n_start[a] + dels;
            if (i < hi) {
                uint32_t k = hi, tdels = 0;
                while (k > i + 1) {
                    const uint8_t qb = DG_Q(col[k - 1]), tb = DG_T(col[k - 1]);
                    if (qb == tb) break;
                    if (tb == DG_GAP) trail++; else if (qb == DG_GAP) tdels++;
                    k--;
                }
                e = p.n_start[a] + (hi - lo - ins) - 1u - tdels;
            } else { e = 0; s = 0; }
        }
    }
    p.rd_s[a] = s; p.rd_e[a] = e; p.rd_lead[a] = lead; p.rd_trail[a] = trail;
}

// ---------------------------------------------------------------------------
// k_groups: one wave per 8 backbone positions.  matC column -> exclusive prefix over
// reads (in place); gcount[p] = inserted vertices whose _bbMap is p.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_groups(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blen = p.tlen[t];
    const uint32_t K = (uint32_t)(p.aln_begin[t + 1] - p.aln_begin[t]);
    uint32_t *col0 = p.matC + p.matc_base[t];
    const uint32_t cstride = p.matc_stride[t];              // a multiple of 8: rows start 32-byte aligned
    uint32_t *gcount = p.gcount + p.bbv_base[t];
    // a wave takes 8 consecutive positions: a lane (= a read) loads and stores them as two 16-byte
    // pieces of its row
    const uint32_t pos0 = (blockIdx.y * 4 + wave) * 8u;
    if (pos0 >= blen + 2) return;
    uint32_t carry[8];
#pragma unroll
    for (int j = 0; j < 8; j++) carry[j] = 0;
    for (uint32_t r0 = 0; r0 < K; r0 += DG_WAVE) {
        const uint32_t r = r0 + lane;
        uint4 *cell = reinterpret_cast<uint4 *>(col0 + (uint64_t)r * cstride + pos0);
        uint4 a = make_uint4(0, 0, 0, 0), b = a;
        if (r < K) { a = cell[0]; b = cell[1]; }
        uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t ex[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t incl = v[j];
            for (int o = 1; o < DG_WAVE; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            ex[j] = carry[j] + incl - v[j];
            carry[j] += __shfl(incl, DG_WAVE - 1);
        }
        if (r < K) {
            cell[0] = make_uint4(ex[0], ex[1], ex[2], ex[3]);
            cell[1] = make_uint4(ex[4], ex[5], ex[6], ex[7]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (lane == j && pos0 + j < blen + 2) gcount[pos0 + j] = carry[j];
}

// ---------------------------------------------------------------------------
// k_gscan: one 1024-thread block per target: gbase[p] = sum_{p'<p} (gcount[p'] + 1).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_gscan(DgParams p) {
    __shared__ uint32_t s_scan[1024];
    __shared__ uint32_t s_carry;
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t np = p.tlen[t] + 2;
    const uint64_t bv = p.bbv_base[t];
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t p0 = 0; p0 < np; p0 += 1024) {
        const uint32_t pos = p0 + tid;
        const uint32_t gc = pos < np ? p.gcount[bv + pos] : 0u;
        const uint32_t v = pos < np ? gc + 1u : 0u;
        s_scan[tid] = v;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            uint32_t x = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += x;
            __syncthreads();
        }
        const uint32_t excl = s_scan[tid] - v + s_carry;
        const uint32_t tot = s_scan[1023];
        __syncthreads();
        if (pos < np) { p.gbase[bv + pos] = excl; p.bid[bv + pos] = excl + gc; }
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    if (tid == 0 && s_carry != p.n_nodes[t]) dg_fail_target(p, t, DG_E_INTERNAL);
}

// ---------------------------------------------------------------------------
// k_emit: addAln.  One wave per (target, group of 64 reads, stretch of 1 << emit_shift
// backbone positions), one lane per read.
//
// A lane enters its stretch at the column k_norm_finish recorded for the stretch's first
// position (or at the read's start) and finds the vertex in front of it by looking back
// over the deletion columns; it writes the arrival side of every vertex of the stretch and
// the departure side of every vertex it created (`own`): for the last one it looks ahead,
// past the stretch's end, for the vertex the read goes to next.
//
// The lanes walk the backbone in lock step, DG_EB positions per batch.  In a batch a
// lane first emits the insertion columns in front of a position and then its match
// / deletion column; the arrival cell of the position stays in a register and the
// departure cells in LDS, and the whole batch is written at its end as 16 + 16
// row stores (K consecutive cells each: coalesced), with no load in between.
// gfx950 counts stores in vmcnt, so a load behind a store waits for the store to
// reach L2: keeping the stores of 16 positions together takes that wait off every
// position.  Columns come 8 at a time (16-byte loads) into two 64-bit registers,
// the backbone ids of the batch in four 16-byte loads.
// ---------------------------------------------------------------------------
// ---- duplicates folded while the graph is built (DgParams::fold) -------------------------------------------------
// Two reads that leave the same backbone vertex u, insert the SAME bases and match the same next backbone position
// v are merged by the reference when it visits u and the vertices of the first read's chain (mergeOutNodes groups
// u's children by base, AlnGraphBoost.cpp:217-267): the later read's chain folds into the earlier one's vertex by
// vertex and contributes nothing but counts -- at every level its vertex is a victim of the earlier read's, its out-edge
// goes where the survivor's own already goes, its entries in out[u] and in[v] are erased with the others keeping their
// order.  So the result of the sweep is the same when the later chain is never linked in (its vertices born deleted) and
// its weights and edge counts are added to the earlier chain's at once (checked against the literal Python model on
// thousands of adversarial pileups before any of it ran here; tests: the merged graph vertex by vertex with the fold on
// and off).  About 55 % of all vertices mergeNodes reaps at configs[1] go this way, and 40 % of its merge visits.
//
// k_emit sees such a pair at one place: the lanes walk the backbone in lock step, and a chain that hangs between a
// backbone vertex and the match column of position `pos` (insertion columns in front of pos only) is complete when the
// lanes have done position pos.  Lanes of `em` hold: key (up to four inserted bases), anc (the vertex in front),
// first (id of the chain's first vertex), nins.  Returns true for a lane whose chain is folded into an earlier lane's
// (its arrival cell then becomes DG_CELL_DUP: the column counts as a match, the read brings no in-edge).
#define DG_CELL_DUP 0x1FFFFFEu
#define DG_EDONE 0xFFFFFFFFu      // k_emit: a lane that has nothing (left) to do in its stretch
#define DG_EF_BB 1u               // k_emit, flags of the previous vertex on the read's path: a backbone vertex
#define DG_EF_OWN 2u              //         created by this wave
// key of a chain that closed at position pos: its (at most three) inserted bases, and in the top byte how far back the
// backbone vertex in front of it lies (1: the usual insertion, 2: the insertion half of a substitution, ...); 0 = no
// chain to compare.  Chains at one position with equal keys are duplicates.
__device__ __forceinline__ uint32_t dg_emit_fold(unsigned long long em, const uint32_t key, const uint32_t first, const uint32_t pos,
                                                 DgNode *ndt, uint32_t *pool, uint32_t *n_out, const int lane) {
    uint32_t victim = 0;
    *n_out = 0;
    while (em & (em - 1ull)) {                            // two or more chains left to compare
        const int f = __ffsll((long long)em) - 1;
        const uint32_t kf = (uint32_t)__builtin_amdgcn_readlane((int)key, f);
        const unsigned long long same = __ballot(key == kf);
        em &= ~same;
        const uint32_t n = (uint32_t)__popcll(same);
        if (n < 2 || !((same >> lane) & 1ull)) continue;
        const uint32_t nins = (key & 0xFF0000u) ? 3u : (key & 0xFF00u) ? 2u : 1u;
        if (lane == f) {
            // the survivor: every vertex of the chain weighs n, every edge along it counts n (u's out-entry for its
            // first vertex gets its count from the departure cell: k_lists)
            *n_out = n;
            for (uint32_t k = 0; k < nins; k++) {
                const uint32_t id = first + k, rk = id - pos;
                reinterpret_cast<DgNode *>(reinterpret_cast<char *>(ndt) + (id << 5))->weight = (int32_t)n;
                pool[3u * rk + 1u] = n;
            }
        } else {
            victim = 1;
            for (uint32_t k = 0; k < nins; k++) {
                // out_len = in_len = 0, flags = deleted (AlnGraphBoost.cpp:269-273); the base stays
                const uint32_t base = (key >> (8u * (nins - 1u - k))) & 0xFFu;
                *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(ndt) + ((first + k) << 5)) = make_uint2(0u, base | (DG_NF_DELETED << 8));
            }
        }
    }
    return victim;
}

// ---------------------------------------------------------------------------
// k_gsum (p.emit_scan): gcount[p] = inserted vertices whose _bbMap is p = the column sum of the reads' run lengths; a
// thread per position, the reads' rows read side by side (k_groups does the same and the prefix over the reads with it,
// in place: where every target has at most a wave of reads k_emit takes that prefix itself)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gsum(DgParams p) {
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const uint32_t pos = blockIdx.y * 256u + threadIdx.x;
    if (pos >= p.tlen[t] + 2u) return;
    const uint32_t K = (uint32_t)(p.aln_begin[t + 1] - p.aln_begin[t]);
    const uint32_t *col = p.matC + p.matc_base[t] + pos;
    const uint32_t stride = p.matc_stride[t];
    uint32_t sum = 0;
    for (uint32_t r = 0; r < K; r++) sum += col[(uint64_t)r * stride];
    p.gcount[p.bbv_base[t] + pos] = sum;
}

#ifndef DG_EB
#define DG_EB 8             // positions per batch (round 3, with the fold inside: 4 / 8 / 16 -> 73 / 86 / 122 VGPRs, 6 / 5 / 4 waves per
                            // SIMD, build 11.47 / 11.21 / 11.43 ms at configs[1])
#endif
#ifndef DG_ERPW
#define DG_ERPW 64          // reads per wave (16 or 32 were tried: no faster)
#endif
#ifndef DG_ECOLS
#define DG_ECOLS 40u          // columns staged in LDS per lane and batch (5 x 16 bytes)
#define DG_ECOLS_STRIDE 42u   // 21 dwords per lane row: odd, lanes spread over banks
#define DG_ENEED 16u          // columns a batch wants staged when it starts (restaged in place if it needs more)
#endif
// (LDS, not registers, bounds the waves in flight here: 2 KB of departure cells + the staged columns.  48 / 40 / 32 / 24
// columns -> 8.4 / 7.4 / 6.4 / 5.4 KB, 86 / 82 / 78 / 74 VGPRs, 5 / 5 / 6 / 6 waves per SIMD: build 10.08 / 9.85 / 9.97 / 10.01 ms)
__global__ __launch_bounds__(64) void k_emit(DgParams p) {
    __shared__ uint32_t s_D[DG_EB * 64];
    __shared__ uint16_t s_col[64 * DG_ECOLS_STRIDE];
    if (dg_failed(p)) return;
    const uint32_t t = blockIdx.x;
    if (dg_tskip(p, t)) return;
    const int lane = threadIdx.x;
    uint16_t *colw = s_col + lane * DG_ECOLS_STRIDE;
    const uint64_t ab = p.aln_begin[t];
    const uint32_t K = (uint32_t)(p.aln_begin[t + 1] - ab);
    const uint32_t r = blockIdx.y * DG_ERPW + lane;
    if (blockIdx.y * DG_ERPW >= K) return;
    const bool idle = r >= K || lane >= DG_ERPW;   // idle lanes stay in the wave (shuffles, votes)
    const uint32_t a = (uint32_t)(ab + (idle ? 0 : r));
    const uint32_t blen = p.tlen[t];
    const uint32_t exitpos = blen + 1;
    if ((blockIdx.z << p.emit_shift) > exitpos) return;      // the grid is sized for the longest target
    const uint64_t nb = p.node_base[t];
    const uint64_t bv = p.bbv_base[t];
    const uint32_t *bid = p.bid + bv, *gbase = p.gbase + bv;
    uint32_t *Am = p.matA + p.mat_base[t];
    uint32_t *Dm = p.matD + p.mat_base[t];
    const uint32_t *Cm = p.matC + p.matc_base[t] + (uint64_t)(idle ? 0 : r) * p.matc_stride[t];   // the read's row
    uint32_t *pool = p.pool + p.pool_base[t];
    DgNode *ndt = p.nodes + nb;
    // pool words / vertex records / matrix cells: uniform base + 32-bit byte offset (see k_merge)
#define DG_EPW(OFF) (*reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(pool) + (((uint32_t)(OFF)) << 2)))
#define DG_ECELL(M, POS) (*reinterpret_cast<uint32_t *>(reinterpret_cast<char *>((M) + (uint64_t)(POS) * K) + (r << 2)))
    const uint16_t *buf = p.norm + p.norm_off[a];
    const uint32_t lo = idle ? 0 : p.n_lo[a], hi = idle ? 0 : p.n_hi[a];
    // What a lane carries along its read lives in vector registers, flags included: a boolean that differs from lane to
    // lane is a lane mask to the compiler, and every such mask that lives across a branch costs three scalar instructions
    // at every join behind it -- with `done`, `own`, `prev_bb`, `have`, `ins_open` ... kept that way a third of the
    // kernel's instructions were mask bookkeeping (2.7 G scalar against 1.6 G vector instructions per launch).
    uint32_t bbpos = idle ? DG_EDONE : p.n_start[a];   // the backbone position the lane's next column belongs to; DG_EDONE: nothing (left) here
    uint32_t prev = 0;            // vertex id of the previous vertex on the read's path
    uint32_t prev_pos = 0;        // its backbone position (_bbMap for an inserted vertex)
    uint32_t fl = DG_EF_BB | DG_EF_OWN;   // prev is a backbone vertex; prev was created by this wave (the enter vertex: by the read's first stretch)
    uint32_t i = lo;
    const uint32_t P0 = blockIdx.z << p.emit_shift, P1 = P0 + (1u << p.emit_shift);
    const bool rvalid = !idle;                     // the lane has a read of its own (its matC row is its own)
    bool pfx_entry = false;
    if (!idle) {
        if (bbpos >= P1) bbpos = DG_EDONE;                  // the read starts in a later stretch
        else if (bbpos < P0) {
            const uint32_t ck = p.ckpt[p.ck_base[a] + blockIdx.z];
            if (ck == DG_CK_NONE || ck >= hi) bbpos = DG_EDONE;  // the read ended in an earlier stretch
            else {
                i = ck; bbpos = P0; fl = DG_EF_BB;
                // the vertex in front: back over deletion columns (and columns addAln skips)
                uint32_t x = ck, d = 0;
                while (x > lo) {
                    const uint16_t c = buf[--x];
                    const uint8_t qb = DG_Q(c), tb = DG_T(c);
                    if (qb == tb) { prev_pos = P0 - 1u - d; prev = bid[prev_pos]; break; }
                    if (qb == DG_GAP) { d++; continue; }
                    if (tb == DG_GAP) {                     // last vertex of the insertion run of position P0 - d
                        prev_pos = P0 - d;
                        if (p.emit_scan) pfx_entry = true;       // (the prefix over the reads is taken below, by the wave)
                        else {
                            const uint32_t c1 = r + 1 < K ? Cm[p.matc_stride[t] + prev_pos] : p.gcount[bv + prev_pos];
                            prev = gbase[prev_pos] + c1 - 1u;
                        }
                        fl = 0;
                        break;
                    }
                }
            }
        }
    }
    // p.emit_scan (targets of at most a wave of reads): matC holds the run lengths themselves and the wave takes the
    // prefix over the reads where it needs it -- here for the lanes that enter the stretch behind an insertion run
    // (one position after the other; few lanes, few positions), per batch of positions below
    if (p.emit_scan) {
        for (unsigned long long m = __ballot(pfx_entry); m; m = __ballot(pfx_entry)) {
            const uint32_t pp = (uint32_t)__builtin_amdgcn_readlane((int)prev_pos, __ffsll((long long)m) - 1);
            const uint32_t x = rvalid ? Cm[pp] : 0u;
            const uint32_t e = dg_wave_excl(x, lane);
            if (pfx_entry && prev_pos == pp) { prev = gbase[pp] + e + x - 1u; pfx_entry = false; }
        }
    }
    uint32_t c_base = 0x80000000u;                // first column staged in colw[] (nothing yet)
    // columns [c_base, c_base + DG_ECOLS) of the lane sit in LDS; a batch normally needs
    // ~18 of them, a long insertion run restages in place
#define DG_STAGE(I)                                                                         \
    do {                                                                                    \
        c_base = (I) & ~7u;                                                                 \
        const uint4 *src_ = reinterpret_cast<const uint4 *>(buf + c_base);                  \
        uint4 v_[DG_ECOLS / 8];                                                             \
        _Pragma("unroll") for (int k_ = 0; k_ < (int)(DG_ECOLS / 8); k_++)                  \
            v_[k_] = (c_base + 8u * k_ < hi) ? src_[k_] : make_uint4(0, 0, 0, 0);           \
        _Pragma("unroll") for (int k_ = 0; k_ < (int)(DG_ECOLS / 8); k_++) {                \
            uint32_t *dst_ = reinterpret_cast<uint32_t *>(colw) + 4 * k_;                   \
            dst_[0] = v_[k_].x; dst_[1] = v_[k_].y; dst_[2] = v_[k_].z; dst_[3] = v_[k_].w; \
        }                                                                                   \
    } while (0)
#define DG_COLUMN(I, OUT)                                                                   \
    do {                                                                                    \
        if ((I) - c_base >= DG_ECOLS) DG_STAGE(I);                                          \
        OUT = colw[(I) - c_base];                                                           \
    } while (0)
    // departure of `prev` to `nxt`: a matrix cell for backbone vertices (staged in LDS when
    // its row belongs to this batch), the single out slot for an inserted vertex (its rank
    // among inserted vertices is id - position)
#define DG_DEPART(NXT)                                                       \
    do {                                                                     \
        if (!(fl & DG_EF_OWN)) break;                                        \
        if (fl & DG_EF_BB) {                                                 \
            if (prev_pos >= pos0) s_D[(prev_pos - pos0) * 64 + lane] = (NXT) + 1u;   \
            else DG_ECELL(Dm, prev_pos) = (NXT) + 1u;                \
        } else {                                                             \
            const uint32_t _rk = prev - prev_pos;                            \
            DG_EPW(3u * _rk) = (NXT);                                        \
            DG_EPW(3u * _rk + 1u) = 1u;                                      \
        }                                                                    \
    } while (0)

    // Every row of the stretch is written (cells no read touches as zeros), so the matrices need no
    // clearing beforehand: the wave starts at the stretch's first position whether or not a read of
    // its is there yet, and goes on to its end (and past position tlen while a read still has to
    // reach the exit: every read ends by position tlen + 1).
    uint32_t pos0 = P0;                            // a multiple of 16
    while (pos0 < P1 && (pos0 <= blen || (!__all(bbpos == DG_EDONE) && pos0 <= blen + 2u * DG_EB))) {
        // backbone ids of the batch: bid[pos0 .. pos0+15] (reads past tlen+1 stay inside the arena)
        uint32_t bidv[DG_EB];
        {
            const uint4 *bp = reinterpret_cast<const uint4 *>(bid + pos0);
#pragma unroll
            for (int k = 0; k < DG_EB / 4; k++) {
                const uint4 v = bp[k];
                bidv[4 * k] = v.x; bidv[4 * k + 1] = v.y; bidv[4 * k + 2] = v.z; bidv[4 * k + 3] = v.w;
            }
        }
        uint32_t gbv[DG_EB], cmv[DG_EB];
        {
            const uint4 *gp = reinterpret_cast<const uint4 *>(gbase + pos0);
#pragma unroll
            for (int k = 0; k < DG_EB / 4; k++) {
                const uint4 v = gp[k];
                gbv[4 * k] = v.x; gbv[4 * k + 1] = v.y; gbv[4 * k + 2] = v.z; gbv[4 * k + 3] = v.w;
            }
            // (the row is padded to a multiple of 8 cells and pos0 is a multiple of 4: whole
            // 16-byte pieces; what lies past the exit position is not used)
            const uint4 *cp = reinterpret_cast<const uint4 *>(Cm + pos0);
#pragma unroll
            for (int k = 0; k < DG_EB / 4; k++) {
                const uint4 v = cp[k];
                cmv[4 * k] = v.x; cmv[4 * k + 1] = v.y; cmv[4 * k + 2] = v.z; cmv[4 * k + 3] = v.w;
            }
        }
        if (p.emit_scan) {
#pragma unroll
            for (int j = 0; j < DG_EB; j++) {
                const uint32_t x = rvalid ? cmv[j] : 0u;
                cmv[j] = __ballot(x != 0u) ? dg_wave_excl(x, lane) : 0u;
            }
        }
        if (bbpos != DG_EDONE && i < hi && (i - c_base) + DG_ENEED > DG_ECOLS) DG_STAGE(i);
        uint32_t acell[DG_EB];
#pragma unroll
        for (int j = 0; j < DG_EB; j++) { acell[j] = 0; s_D[j * 64 + lane] = 0; }
        uint32_t exit_val = 0;                      // (not 0: the arrival cell of the exit vertex)
#pragma unroll
        for (int j = 0; j < DG_EB; j++) {
            const uint32_t pos = pos0 + j;
            uint32_t f_apos = 0, f_key = 0, f_first = 0;
            asm volatile("" : "+v"(fl));            // (the flags stay a register: no lane masks rebuilt from them across positions)
            if (bbpos == pos && pos < P1) {
                // columns in front of this position that do not advance the backbone cursor:
                // insertions (AlnGraphBoost.cpp:95-104); raw columns that match no branch are skipped
                uint32_t ins_id = 0;                  // next id of the read's insertion run here (0: no insertion column yet)
                uint32_t kacc = 0;                    // its bases
                uint16_t c = 0;
                f_apos = prev_pos;
                const bool f_anc_ok = fl == (DG_EF_BB | DG_EF_OWN) && pos - prev_pos < 16u;
                for (;;) {
                    if (i < hi) DG_COLUMN(i, c);
                    const uint8_t qb = DG_Q(c), tb = DG_T(c);
                    if (i >= hi || qb == tb || qb == DG_GAP) break;
                    if (tb == DG_GAP) {
                        if (!ins_id) { ins_id = gbv[j] + cmv[j]; f_first = ins_id; }
                        kacc = (kacc << 8) | qb;
                        const uint32_t id = ins_id++;
                        const uint32_t rk = id - bbpos;   // bbpos backbone vertices precede group bbpos
                        DgNode nd;
                        nd.out_len = 1; nd.in_len = 1; nd.base = qb; nd.flags = 0; nd.pad = 0;
                        nd.weight = 1; nd.pending = 1;
                        nd.out_off = 3u * rk; nd.in_off = 3u * rk + 2u; nd.out_cap = 1; nd.in_cap = 1;
                        nd.bbpos = (int32_t)bbpos;
                        *reinterpret_cast<DgNode *>(reinterpret_cast<char *>(ndt) + (id << 5)) = nd;
                        DG_EPW(3u * rk + 2u) = prev;
                        DG_DEPART(id);
                        prev = id; prev_pos = bbpos; fl = DG_EF_OWN;
                    }
                    i++;
                }
                if (i >= hi) {                            // :106 the read ends: edge to the exit vertex
                    // (behind nothing but deletions of this stretch the earlier stretch's look-ahead
                    // has been here already)
                    exit_val = (fl & DG_EF_OWN) ? prev + 1u : 0u;
                    const uint32_t ex = bid[exitpos];
                    DG_DEPART(ex);
                    bbpos = DG_EDONE;
                } else {
                    const uint8_t qb = DG_Q(c), tb = DG_T(c);
                    if (qb == tb) {                       // match (:75-85)
                        const uint32_t cur = bidv[j];
                        acell[j] = ((uint32_t)tb << 25) | (prev + 1u);
                        // (a chain of one to three vertices between a backbone vertex and this match: dg_emit_fold)
                        if (f_anc_ok && kacc != 0u && kacc < (1u << 24)) f_key = kacc | ((pos - f_apos) << 24);
                        DG_DEPART(cur);
                        prev = cur; prev_pos = bbpos; fl = DG_EF_BB | DG_EF_OWN;
                    } else {                              // deletion (:87-93)
                        acell[j] = ((uint32_t)tb << 25) | DG_CELL_DEL;
                    }
                    bbpos++;
                    i++;
                }
            }
            if (p.fold) {
                // chains that closed at this position, two or more of them: fold the duplicates (dg_emit_fold)
                const unsigned long long em = __ballot(f_key != 0u);
                if (em & (em - 1ull)) {
                    uint32_t fn;
                    const uint32_t vic = dg_emit_fold(em, f_key, f_first, pos, ndt, pool, &fn, lane);
                    if (vic | fn) {
                        // the departure cell of the vertex in front: the survivor's carries the reads folded into its
                        // chain (k_lists counts the out-edge with them), a victim's is empty
                        const uint32_t dv = vic ? 0u : (f_first + 1u) | ((fn - 1u) << 25);
                        if (f_apos >= pos0) s_D[(f_apos - pos0) * 64 + lane] = dv;
                        // (enter's row, position 0, leaves a batch only where a cell is not 0 -- its cells belong to
                        // whichever stretch a read starts in -- so a victim's empty cell there is stored here; nobody
                        // clears the matrix, and a stale cell would be a vertex that does not exist)
                        if (f_apos < pos0 || f_apos == 0u) DG_ECELL(Dm, f_apos) = dv;
                        if (vic) acell[j] = (acell[j] & 0xFE000000u) | DG_CELL_DUP;
                    }
                }
            }
        }
        // the batch leaves as row stores; rows past the exit row do not exist
        if (r < K && lane < DG_ERPW) {
#pragma unroll
            for (int j = 0; j < DG_EB; j++) {
                const uint32_t pos = pos0 + j;
                if (pos <= blen && pos >= P0 && pos < P1) {
                    DG_ECELL(Am, pos) = acell[j];
                    // (the enter vertex's row is shared: a read that starts in a later stretch
                    // has its cell written by that stretch's wave)
                    const uint32_t dv = s_D[j * 64 + lane];
                    if (pos > 0 || dv) DG_ECELL(Dm, pos) = dv;
                }
            }
            if (exit_val) DG_ECELL(Am, exitpos) = exit_val;
        }
        pos0 += DG_EB;
    }
    // the read goes on beyond the stretch: the departure of the last vertex created here needs
    // the next vertex of the path (the next stretch's wave writes that vertex's arrival side)
    {
        const bool ahead = bbpos != DG_EDONE && (fl & DG_EF_OWN);
        uint32_t q = bbpos, nxt = 0;
        bool found = false, pfx_next = false;
        if (ahead) {
            while (i < hi) {
                const uint16_t c = buf[i];
                const uint8_t qb = DG_Q(c), tb = DG_T(c);
                if (qb == tb) { nxt = bid[q]; found = true; break; }
                if (qb == DG_GAP) { q++; i++; continue; }
                if (tb == DG_GAP) {
                    if (p.emit_scan) pfx_next = true; else nxt = gbase[q] + Cm[q];
                    found = true; break;
                }
                i++;
            }
        }
        if (p.emit_scan) {
            // (the first vertex of the read's insertion run at position q: the prefix over the reads in front, by the wave)
            for (unsigned long long m = __ballot(pfx_next); m; m = __ballot(pfx_next)) {
                const uint32_t pp = (uint32_t)__builtin_amdgcn_readlane((int)q, __ffsll((long long)m) - 1);
                const uint32_t x = rvalid ? Cm[pp] : 0u;
                const uint32_t e = dg_wave_excl(x, lane);
                if (pfx_next && q == pp) { nxt = gbase[pp] + e; pfx_next = false; }
            }
        }
        if (ahead) {
            if (!found) {                                     // nothing but deletions to the end (:106)
                nxt = bid[exitpos];
                Am[(uint64_t)exitpos * K + r] = prev + 1u;
            }
            const uint32_t pos0 = 0xFFFFFFFFu;                // no row of this departure is staged any more
            DG_DEPART(nxt);
        }
    }
#undef DG_EPW
#undef DG_ECELL
#undef DG_DEPART
#undef DG_COLUMN
#undef DG_STAGE
}

// ---------------------------------------------------------------------------
// k_lists: one wave per backbone position.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int dg_lds_find(volatile int32_t *vals, int n, int x, int lane) {
    for (int b = 0; b < n; b += DG_WAVE) {
        int i = b + lane;
        bool hit = (i < n) && (vals[i] == x);
        unsigned long long m = __ballot(hit);
        if (m) return b + (__ffsll((long long)m) - 1);
    }
    return -1;
}

__device__ __forceinline__ uint32_t dg_pool_alloc(const DgParams &p, uint32_t t, uint32_t words, int lane) {
    uint32_t off = 0;
    if (lane == 0) {
        off = atomicAdd(&p.pool_top[t], words);
        if ((uint64_t)off + words > p.pool_size[t]) { dg_fail(p, DG_E_POOL_TGT); p.st->bad_target = t; off = 0xFFFFFFFFu; }
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)off, 0);
}

__global__ __launch_bounds__(256) void k_lists(DgParams p) {
    extern __shared__ int32_t s_tmp[];      // per wave: vals[max_k+2], cnts[max_k+2]
    const uint32_t t = blockIdx.x;
    if (dg_failed(p) || dg_tskip(p, t)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blen = p.tlen[t];
    const uint32_t K = (uint32_t)(p.aln_begin[t + 1] - p.aln_begin[t]);
    const uint32_t stride = p.max_k + 2;
    volatile int32_t *vals = s_tmp + (size_t)wave * 2 * stride;
    volatile int32_t *cnts = vals + stride;
    const uint64_t nb = p.node_base[t];
    const uint64_t bv = p.bbv_base[t];
    uint32_t *pool = p.pool + p.pool_base[t];
    // cells of read r at position pos: [position][read] from k_emit, [read][position] (rows as matC's) from k_emit2
    const uint32_t *Am0 = p.matA + (p.emit2 ? p.matc_base[t] : p.mat_base[t]), *Dm0 = p.matD + (p.emit2 ? p.matc_base[t] : p.mat_base[t]);
    const uint32_t rstep = p.emit2 ? p.matc_stride[t] : 1u;
    const uint32_t capb = dg_capb(K);
    const uint32_t fixed0 = 3u * p.t_nins[t];
    // a wave takes DG_LPW consecutive positions: the set-up above is paid once for them
    for (uint32_t pi = 0; pi < DG_LPW; pi++) {
    const uint32_t pos = (blockIdx.y * 4 + wave) * DG_LPW + pi;
    if (pos >= blen + 2) return;
    const uint32_t v = p.bid[bv + pos];
    const uint32_t *Am = Am0 + (p.emit2 ? (uint64_t)pos : (uint64_t)pos * K);
    const uint32_t *Dm = Dm0 + (p.emit2 ? (uint64_t)pos : (uint64_t)pos * K);
    const uint32_t fixed = fixed0 + pos * 3u * capb;
    uint32_t out_off = fixed, in_off = fixed + 2u * capb;
    uint32_t out_cap = capb, in_cap = capb;
    uint32_t out_len = 0, in_len = 0;
    uint32_t n_match = 0, n_cov = 0, last_base = 0;

    for (int dir = 0; dir < 2; dir++) {
        // dir 0: out list from departures; dir 1: in list from arrivals
        if (dir == 0 && pos == blen + 1) continue;
        if (dir == 1 && pos == 0) continue;
        const uint32_t *row = dir == 0 ? Dm : Am;
        if (K <= DG_WAVE) {
            // ---- one read per lane.  Entry 0 is the constructor's neighbour; a neighbour that is an
            // inserted vertex of the adjacent group belongs to one read only, so it is an entry of
            // its own with no comparing; only the other backbone vertices (deletion jumps, the exit,
            // the enter) are grouped by peeling.  Entries go straight to the pool in the order of
            // their first read. ----
            const uint32_t cell = (uint32_t)lane < K ? row[(uint64_t)lane * rstep] : 0u;
            // departure: neighbour id + 1 (0 = none) and, above bit 25, the reads k_emit folded into this one's chain
            int32_t val = (int32_t)DG_CELL_ID(cell);
            const int32_t extra = (int32_t)(cell >> 25);
            if (dir == 1) {
                const uint32_t idf = DG_CELL_ID(cell);
                const unsigned long long covered = __ballot(cell != 0u);
                if (pos <= blen) {
                    n_cov += (uint32_t)__popcll(covered);
                    n_match += (uint32_t)__popcll(__ballot(cell != 0u && idf != DG_CELL_DEL));
                    if (covered) last_base = DG_CELL_BASE((uint32_t)__builtin_amdgcn_readlane((int)cell, 63 - __clzll((long long)covered)));
                }
                val = (cell != 0u && idf != DG_CELL_DEL && idf != DG_CELL_DUP) ? (int32_t)idf : 0;
            }
            const uint32_t gp = dir == 0 ? pos + 1 : pos;
            const int32_t chain = (int32_t)p.bid[bv + (dir == 0 ? pos + 1 : pos - 1)] + 1;
            const int32_t ulo = (int32_t)p.gbase[bv + gp] + 1, uhi = (int32_t)p.bid[bv + gp] + 1;
            const unsigned long long cm = __ballot(val == chain);
            const bool uniq = val >= ulo && val < uhi;
            unsigned long long firsts = __ballot(uniq);
            int32_t cnt = dir == 0 ? 1 + extra : 1;
            unsigned long long rem = __ballot(val != 0 && !uniq) & ~cm;
            while (rem) {
                const int first = __ffsll((long long)rem) - 1;
                const int32_t x = __builtin_amdgcn_readlane(val, first);
                const unsigned long long same = __ballot(val == x);
                rem &= ~same;
                if (lane == first) cnt = __popcll(same) + (dir == 0 ? extra : 0);
                firsts |= 1ull << first;
            }
            const uint32_t n = 1u + (uint32_t)__popcll(firsts);
            uint32_t off = dir == 0 ? out_off : in_off;
            if (n > capb) {                                     // rare: move to the growth region
                const uint32_t cap = n + 2u;
                off = dg_pool_alloc(p, t, dir == 0 ? 2u * cap : cap, lane);
                if (off == 0xFFFFFFFFu) return;
                if (dir == 0) { out_off = off; out_cap = cap; } else { in_off = off; in_cap = cap; }
            }
            const uint32_t idx = 1u + (uint32_t)__popcll(firsts & ((1ull << lane) - 1ull));
            if (dir == 0) {
                if (lane == 0) { pool[off] = (uint32_t)(chain - 1); pool[off + 1] = (uint32_t)__popcll(cm); }
                if ((firsts >> lane) & 1ull) { pool[off + 2 * idx] = (uint32_t)(val - 1); pool[off + 2 * idx + 1] = (uint32_t)cnt; }
                out_len = n;
            } else {
                if (lane == 0) pool[off] = (uint32_t)(chain - 1);
                if ((firsts >> lane) & 1ull) pool[off + idx] = (uint32_t)(val - 1);
                in_len = n;
            }
            continue;
        }
        // ---- more than a wave of reads: the list under construction has entry i on lane i while it
        // fits a wave, spilled to LDS beyond that ----
        int n = 1;
        int32_t lv = dir == 0 ? (int32_t)p.bid[bv + pos + 1] : (int32_t)p.bid[bv + pos - 1];
        int32_t lc = 0;
        bool in_lds = false;
        for (uint32_t r0 = 0; r0 < K; r0 += DG_WAVE) {
            const uint32_t r = r0 + lane;
            const uint32_t cell = r < K ? row[(uint64_t)r * rstep] : 0u;
            int32_t val = (int32_t)DG_CELL_ID(cell);           // neighbour id + 1, 0 = none
            const int32_t extra = dir == 0 ? (int32_t)(cell >> 25) : 0;       // reads folded into this one's chain (k_emit)
            if (dir == 1) {
                const uint32_t idf = DG_CELL_ID(cell);
                const unsigned long long covered = __ballot(cell != 0u);
                if (pos <= blen) {
                    n_cov += (uint32_t)__popcll(covered);
                    n_match += (uint32_t)__popcll(__ballot(cell != 0u && idf != DG_CELL_DEL));
                    if (covered) last_base = DG_CELL_BASE((uint32_t)__builtin_amdgcn_readlane((int)cell, 63 - __clzll((long long)covered)));
                }
                val = (cell != 0u && idf != DG_CELL_DEL && idf != DG_CELL_DUP) ? (int32_t)idf : 0;
            }
            unsigned long long rem = __ballot(val != 0);
            if (!in_lds) {
                // the constructor's neighbour (entry 0) takes most of the row: no peeling round for it
                const int32_t chain = __builtin_amdgcn_readlane(lv, 0) + 1;
                const unsigned long long cm = __ballot(val == chain);
                if (lane == 0) lc += __popcll(cm);
                rem &= ~cm;
            }
            while (rem) {
                const int first = __ffsll((long long)rem) - 1;
                const int32_t x = __builtin_amdgcn_readlane(val, first);
                const unsigned long long same = __ballot(val == x);
                rem &= ~same;
                const int c = __popcll(same) + __builtin_amdgcn_readlane(extra, first);
                if (!in_lds) {
                    const unsigned long long hit = __ballot(lane < n && lv == x - 1);
                    if (hit) {
                        if (lane == __ffsll((long long)hit) - 1) lc += c;
                    } else if (n < DG_WAVE) {
                        if (lane == n) { lv = x - 1; lc = c; }
                        n++;
                    } else {
                        vals[lane] = lv; cnts[lane] = lc;      // 64 entries so far: continue in LDS
                        in_lds = true;
                    }
                }
                if (in_lds) {
                    const int idx = dg_lds_find(vals, n, x - 1, lane);
                    if (idx >= 0) {
                        if (lane == 0) cnts[idx] += c;
                    } else {
                        if (lane == 0) { vals[n] = x - 1; cnts[n] = c; }
                        n++;
                    }
                }
            }
        }
        if (n > 65000) { if (lane == 0) dg_fail_target(p, t, DG_E_INTERNAL); return; }
        uint32_t off = dir == 0 ? out_off : in_off;
        if ((uint32_t)n > capb) {                               // rare: move to the growth region
            const uint32_t cap = (uint32_t)n + 2u;
            off = dg_pool_alloc(p, t, dir == 0 ? 2u * cap : cap, lane);
            if (off == 0xFFFFFFFFu) return;
            if (dir == 0) { out_off = off; out_cap = cap; } else { in_off = off; in_cap = cap; }
        }
        if (!in_lds) {
            if (lane < n) {
                if (dir == 0) { pool[off + 2 * lane] = (uint32_t)lv; pool[off + 2 * lane + 1] = (uint32_t)lc; }
                else pool[off + lane] = (uint32_t)lv;
            }
        } else {
            for (int i = lane; i < n; i += DG_WAVE) {
                if (dir == 0) { pool[off + 2 * i] = (uint32_t)vals[i]; pool[off + 2 * i + 1] = (uint32_t)cnts[i]; }
                else pool[off + i] = (uint32_t)vals[i];
            }
        }
        if (dir == 0) out_len = (uint32_t)n; else in_len = (uint32_t)n;
    }
    if (lane == 0) {
        // the whole record of the backbone vertex (AlnGraphBoost.cpp:16-62 + what addAln made of it)
        const bool inner = pos >= 1 && pos <= blen;
        DgNode nd;
        nd.out_len = (uint16_t)out_len;
        nd.in_len = (uint16_t)in_len;
        nd.flags = DG_NF_BACKBONE; nd.pad = 0;
        if (pos == 0) nd.base = '^';
        else if (pos == blen + 1) nd.base = '$';
        else if (n_cov) nd.base = (uint8_t)last_base;           // last read to cover the position (:79,:90)
        else nd.base = p.bb ? p.bb[p.bb_off[t] + (pos - 1)] : (uint8_t)'N';
        nd.weight = inner ? 1 + (int32_t)n_match : 0;           // :81
        nd.pending = (int32_t)in_len;
        nd.out_off = out_off; nd.in_off = in_off;
        nd.out_cap = (uint16_t)out_cap; nd.in_cap = (uint16_t)in_cap;
        nd.bbpos = inner ? (int32_t)pos : 0;                    // _bbMap: absent key (enter, exit) reads as 0
        p.cov[bv + pos] = inner ? (int32_t)n_cov : 0;           // :76,:87
        p.nodes[nb + v] = nd;
    }
    }
}
