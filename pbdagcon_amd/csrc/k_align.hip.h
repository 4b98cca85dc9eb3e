// k_align.hip.h -- the `-a` stage: re-alignment of .pre records (SimpleAligner.cpp:25-63).
//
// The reference re-aligns every .pre record (unaligned query / target substrings,
// Alignment.cpp:82-112) with blasr_libcpp's SDPAlign + GuidedAlign before normalizeGaps
// (main.cpp:127-128).  That library is not in the tree: PARITY UNPINNED except for the
// reference's one known-answer test (test/cpp/SimpleAlignerTest.cpp:8-21), which this stage
// reproduces.  What is computed: the global alignment that minimises blasr's distance score with
// SimpleAligner's parameters (SimpleAligner.cpp:10-23: match -5, mismatch +6, insertion 4,
// deletion 5), ties resolved diagonal first, then insertion (gap in the target), then deletion,
// inside a band of half-width dg_align_halfwidth() around the length-scaled diagonal
// j = i * tlen / qlen (the role of GuidedAlign's band around the SDP chain); narrower bands are tried first and
// stand when the path keeps away from their edges: a band that follows the alignment (k_align_adapt), then a static
// one (dg_align_halfwidth_first).  The tests hold a bit-exact CPU twin of all of it.
//
// One wave per alignment, the band of a row in REGISTERS: B = 2 W + 1 cells right-aligned on 64 lanes x C
// cells (C = 2 .. 16, one kernel instance per C), lane l owning cells l C .. l C + C - 1 with their previous-row
// scores and their target characters.  When the band's centre moves on by one column, scores and characters
// move one cell to the left (C register moves and two DPP wave shifts; the entering character comes out of a
// 64-character prefetch register), so a cell's diagonal and upper neighbours are always the cell to its left
// and itself: no LDS, no barrier in the row loop.  The dependency on the cell to the left inside a row
// (deletions) is a running minimum, S[k] = min_{k' <= k} (A[k'] - 5 k') + 5 k with A = min(diagonal,
// insertion): C sequential steps per lane plus one DPP prefix minimum over the lanes.  Two bits of direction
// per cell go to HBM as one coalesced 256-byte store per row.  The walk back from (qlen, tlen) runs on scalar
// registers, rows of directions staged through LDS 16 at a time, and leaves one code per step; the
// characters are then filled in 64 steps at a time (ballot prefix counts give every step its i and j).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define DG_AL_MATCH (-5)
#define DG_AL_MISMATCH 6
#define DG_AL_INS 4
#define DG_AL_DEL 5
#define DG_AL_BIG (1 << 28)      // unreachable: every score >= DG_AL_LIM is (reachable scores stay below 2^23)
#define DG_AL_LIM (1 << 27)
#define DG_AL_MAXW 480u
#define DG_AL_ROWS 16            // rows of directions staged in LDS during the walk back

__host__ __device__ inline uint32_t dg_isqrt64(uint64_t x) {
    uint64_t r = 0, b = 1ull << 62;
    while (b > x) b >>= 2;
    while (b) {
        if (x >= r + b) { x -= r + b; r = (r >> 1) + b; } else r >>= 1;
        b >>= 2;
    }
    return (uint32_t)r;
}
// a random walk of indels at ~15 % leaves the scaled diagonal by ~sqrt(0.15 L): four of those, and room
__host__ __device__ inline uint32_t dg_align_halfwidth(uint32_t qlen, uint32_t tlen) {
    const uint64_t L = qlen > tlen ? qlen : tlen;
    uint32_t w = 32u + 4u * dg_isqrt64((15ull * L + 99ull) / 100ull);
    return w > DG_AL_MAXW ? DG_AL_MAXW : w;
}
// The band tried first: two of those.  The alignment found in it stands if its path keeps DG_AL_MARGIN
// cells away from both edges of the band on every row; else the pair is done again in the full band
// (dagcon_align: second pass).  Hardly a pair needs the full band: 1.8 x fewer cells, half the directions.
__host__ __device__ inline uint32_t dg_align_halfwidth_first(uint32_t qlen, uint32_t tlen) {
    const uint64_t L = qlen > tlen ? qlen : tlen;
    const uint32_t w = 32u + 2u * dg_isqrt64((15ull * L + 99ull) / 100ull), full = dg_align_halfwidth(qlen, tlen);
    return w > full ? full : w;
}
#define DG_AL_MARGIN 8
#define DG_AL_RETRY 0xFFFFFFFFu   // aln_len of a pair whose path came too near an edge of the first band
// cells per lane a band of half-width w needs, rounded up to a kernel instance (2, 4, 6, 8, 12, 16)
__host__ __device__ inline uint32_t dg_align_cells(uint32_t w) {
    const uint32_t c = (2u * w + 1u + 63u) / 64u;
    return c <= 2 ? 2u : c <= 4 ? 4u : c <= 6 ? 6u : c <= 8 ? 8u : c <= 12 ? 12u : 16u;
}
// 256-byte units a pair takes in the direction buffer: q_len + 1 rows of 64 direction words (16-bit words up to
// 8 cells per lane), then one code per step of the walk
__host__ __device__ inline uint64_t dg_align_rows(uint32_t qlen, uint32_t tlen, uint32_t cells) {
    const uint64_t row_bytes = cells <= 8u ? 128ull : 256ull;
    return (((uint64_t)qlen + 1ull) * row_bytes + 255ull) / 256ull + ((uint64_t)qlen + tlen + 255ull) / 256ull + 1ull;
}

struct DgAlignParams {
    const uint8_t *q, *t;          // sequence blobs
    const uint64_t *q_off, *t_off;
    const uint32_t *q_len, *t_len;
    const uint64_t *out_off;       // per alignment: room for q_len + t_len columns in qaln / taln
    uint8_t *qaln, *taln;          // written from the front of each alignment's room
    uint32_t *aln_len;
    uint32_t *dirs;                // direction words: 64 per row
    const uint64_t *dir_off;       // per alignment, in 256-byte units (dg_align_rows of them)
    const uint32_t *idx;           // the alignments of this launch
    const uint32_t *halfw;         // per alignment: half-width of the band of this pass
    uint32_t first_pass;           // 1: a path near an edge of the band gives DG_AL_RETRY instead of an alignment
    uint32_t n;
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dg_al_dpp(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
// s_waitcnt vmcnt(0) where a (rare) block has loaded from HBM: the compiler then knows that nothing is pending
// when the row loop goes on, and does not make every row wait for the previous row's direction store
#define DG_AL_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)
#define DG_DPP_ROW_SHR(n) (0x110 + (n))
#define DG_DPP_WAVE_SHL1 0x130     // lane l reads lane l + 1
#define DG_DPP_WAVE_SHR1 0x138     // lane l reads lane l - 1
#define DG_DPP_BCAST15 0x142
#define DG_DPP_BCAST31 0x143

template <int C>
__global__ __launch_bounds__(64) void k_align_band(DgAlignParams p) {
    typedef typename std::conditional<(C <= 8), uint16_t, uint32_t>::type DirT;      // 2 bits per cell
    __shared__ DirT s_dir[DG_AL_ROWS * 64];
    // the characters the row loop consumes one at a time (q[i - 1]; the t character that enters the band on its
    // right) come out of two 128-character LDS windows, refilled 64 at a time: a global load inside the row loop
    // would make every row wait for the previous row's direction store (gfx950 counts stores in vmcnt)
    __shared__ uint8_t s_qw[128], s_tw[128];
    const uint32_t a = p.idx[blockIdx.x];
    const int lane = threadIdx.x;
    const uint32_t n = p.q_len[a], m = p.t_len[a];
    const uint8_t *q = p.q + p.q_off[a], *t = p.t + p.t_off[a];
    uint8_t *qo = p.qaln + p.out_off[a], *to = p.taln + p.out_off[a];
    if (n == 0 || m == 0) {
        // all gaps, in order: query columns first
        for (uint32_t i = lane; i < n; i += 64) { qo[i] = q[i]; to[i] = '-'; }
        for (uint32_t j = lane; j < m; j += 64) { qo[n + j] = '-'; to[n + j] = t[j]; }
        if (lane == 0) p.aln_len[a] = n + m;
        return;
    }
    const int W = (int)p.halfw[a], B = 2 * W + 1;
    const int off = 64 * C - B;                             // dead cells in front of the band (host: B <= 64 C)
    DirT *dirs = reinterpret_cast<DirT *>(p.dirs + p.dir_off[a] * 64ull);
    // one code per step of the walk, behind the rows (on a 256-byte boundary)
    uint8_t *path = reinterpret_cast<uint8_t *>(p.dirs + p.dir_off[a] * 64ull) + ((((uint64_t)n + 1ull) * 64ull * sizeof(DirT) + 255ull) & ~255ull);

    // ---- forward: row i holds columns j = c_i - W + k, k = 0 .. B - 1, c_i = i m / n ----
    int P[C], T[C];
    int jl = -W - off + lane * C;                           // column of the lane's first cell
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int j = jl + c;
        P[c] = DG_AL_BIG;
        T[c] = (j >= 1 && j <= (int)m) ? (int)t[j - 1] : 0;
    }
    // window r -> slot r & 127: s_tw holds t[tw0 + r] (tw0 = W: the first character to enter), s_qw holds q[r]
    int tw0 = W;
    uint32_t tr = 0;                                        // characters of the window consumed so far
    for (int x = lane; x < 128; x += 64) {
        s_tw[x] = (uint32_t)(tw0 + x) < m ? t[tw0 + x] : (uint8_t)0;
        s_qw[x] = (uint32_t)x < n ? q[x] : (uint8_t)0;
    }
    DG_AL_WAIT_LOADS();
    uint32_t num = 0;                                       // i m = c_i n + num
    int ci = 0;
    int tcur = (int)s_tw[0];                                // the next character to enter
    int qc = -1, qcn = (int)s_qw[0];                        // q[i - 1] of this row, of the next
    // the band moves on to row i: scores and characters follow, q[i - 1] is fetched
    auto advance = [&](const uint32_t i) {
        {
            num += m;
            uint32_t shift = 0;
            while (num >= n) { num -= n; shift++; }
            if (shift > (uint32_t)B) {
                // the band jumps by more than its width (tlen >> qlen): nothing of the previous row is in reach of this one
                ci += (int)shift; jl += (int)shift;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const int j = jl + c;
                    P[c] = DG_AL_BIG;
                    T[c] = (j >= 1 && j <= (int)m) ? (int)t[j - 1] : 0;
                }
                tw0 = ci + W; tr = 0;
                for (int x = lane; x < 128; x += 64) s_tw[x] = (uint32_t)(tw0 + x) < m ? t[tw0 + x] : (uint8_t)0;
                DG_AL_WAIT_LOADS();
                tcur = (int)s_tw[0];
            } else {
                for (uint32_t s = 0; s < shift; s++) {
                    ci++; jl++;
                    const int newc = tcur;                                          // t[c_i + W - 1] (0 past the end of t)
                    tr++;
                    if ((tr & 63u) == 0 && tr >= 64u) {                             // the half behind is finished with
                        const uint32_t r = tr + 64u + (uint32_t)lane;
                        s_tw[r & 127u] = (uint32_t)tw0 + r < m ? t[(uint32_t)tw0 + r] : (uint8_t)0;
                        DG_AL_WAIT_LOADS();
                    }
                    tcur = (int)s_tw[tr & 127u];
                    int pin = dg_al_dpp<DG_DPP_WAVE_SHL1, 0xf>(DG_AL_BIG, P[0]);
                    int tin = dg_al_dpp<DG_DPP_WAVE_SHL1, 0xf>(0, T[0]);
                    if (lane == 63) { pin = DG_AL_BIG; tin = newc; }
#pragma unroll
                    for (int c = 0; c + 1 < C; c++) { P[c] = P[c + 1]; T[c] = T[c + 1]; }
                    P[C - 1] = pin; T[C - 1] = tin;
                }
            }
            qc = qcn;
            if ((i & 63u) == 0 && i >= 64u) {                                       // q[i .. i + 63] went out of use
                const uint32_t r = i + 64u + (uint32_t)lane;
                s_qw[r & 127u] = r < n ? q[r] : (uint8_t)0;
                DG_AL_WAIT_LOADS();
            }
            qcn = (int)s_qw[i & 127u];                                              // q[i]: the next row's
        }
    };
    // row i.  EDGE: the band hangs over an end of t on this row (columns < 0 or > m are not cells); rows in the
    // middle, the bulk, skip the test
    auto row = [&](const uint32_t i, auto first_row, auto edge_row) {
        constexpr bool FIRST = decltype(first_row)::value, EDGE = decltype(edge_row)::value;
        // A = min(diagonal, insertion) of the lane's cells
        const int left = dg_al_dpp<DG_DPP_WAVE_SHR1, 0xf>(DG_AL_BIG, P[C - 1]);     // the lane in front: its last cell
        int A[C];
        uint32_t dbits = 0;                                 // bit c: insertion wins over the diagonal
        int lm = DG_AL_BIG;
        const int kb = lane * C;                            // (scores go through X = A - 5 k: any common origin of k does)
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int j = jl + c;
            const bool valid = (!EDGE || (uint32_t)j <= m) && kb + c >= off;
            int best;
            if constexpr (FIRST) best = j == 0 ? 0 : DG_AL_BIG;
            else {
                const int dg = (c == 0 ? left : P[c - 1]) + (T[c] == qc ? DG_AL_MATCH : DG_AL_MISMATCH);
                const int up = P[c] + DG_AL_INS;
                const bool ins = up < dg;
                best = ins ? up : dg;
                dbits |= ins ? 1u << c : 0u;
            }
            best = valid ? best : DG_AL_BIG;
            A[c] = best;
            const int x = best - DG_AL_DEL * (kb + c);
            lm = x < lm ? x : lm;
        }
        // exclusive prefix minimum over the lanes in front
        int incl = lm, v;
        v = dg_al_dpp<DG_DPP_ROW_SHR(1), 0xf>(0x7fffffff, incl); incl = v < incl ? v : incl;
        v = dg_al_dpp<DG_DPP_ROW_SHR(2), 0xf>(0x7fffffff, incl); incl = v < incl ? v : incl;
        v = dg_al_dpp<DG_DPP_ROW_SHR(4), 0xf>(0x7fffffff, incl); incl = v < incl ? v : incl;
        v = dg_al_dpp<DG_DPP_ROW_SHR(8), 0xf>(0x7fffffff, incl); incl = v < incl ? v : incl;
        v = dg_al_dpp<DG_DPP_BCAST15, 0xa>(0x7fffffff, incl); incl = v < incl ? v : incl;
        v = dg_al_dpp<DG_DPP_BCAST31, 0xc>(0x7fffffff, incl); incl = v < incl ? v : incl;
        int pm = dg_al_dpp<DG_DPP_WAVE_SHR1, 0xf>(0x7fffffff, incl);
        uint32_t word = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int j = jl + c;
            const bool valid = (!EDGE || (uint32_t)j <= m) && kb + c >= off;
            const int x = A[c] - DG_AL_DEL * (kb + c);
            // deletion (left) wins only when strictly better than diagonal / insertion
            const bool del = j > 0 && pm < x;
            int sc = del ? pm + DG_AL_DEL * (kb + c) : A[c];
            const uint32_t d = del ? 2u : (dbits >> c) & 1u;
            pm = x < pm ? x : pm;
            P[c] = valid ? sc : DG_AL_BIG;
            word |= d << (2 * c);
        }
        dirs[(uint64_t)i * 64ull + (uint64_t)lane] = (DirT)word;
    };
    row(0u, std::true_type{}, std::true_type{});
    for (uint32_t i = 1; i <= n; i++) {
        advance(i);
        if (ci >= W && ci + W <= (int)m) row(i, std::false_type{}, std::false_type{});
        else row(i, std::false_type{}, std::true_type{});
    }
    // (n, m) is cell k = W of the last row
    const int kend = W + off;
    int fin = DG_AL_BIG;
#pragma unroll
    for (int c = 0; c < C; c++) if (kend % C == c) fin = P[c];
    fin = __builtin_amdgcn_readlane(fin, kend / C);
    if (fin >= DG_AL_LIM) {                                 // the band does not connect (0, 0) with (n, m)
        if (lane == 0) p.aln_len[a] = p.first_pass ? DG_AL_RETRY : 0u;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

    // ---- walk back from (n, m): uniform, directions through LDS, one code per step ----
    uint32_t i = n, j = m, len = 0;
    int wci = (int)m;                                       // c_i of the row the walk is on
    uint32_t wnum = 0;
    int r0 = (int)n + 1;                                    // rows [r0, ..] are staged
    int codes = 0;
    const uint32_t cap = n + m;
    bool bad = false, near_edge = false;
    while (i > 0 || j > 0) {
        if ((int)i < r0) {
            __syncthreads();
            r0 = (int)i >= DG_AL_ROWS - 1 ? (int)i - (DG_AL_ROWS - 1) : 0;
            for (uint32_t x = lane; x < ((uint32_t)((int)i - r0) + 1u) * 64u; x += 64) s_dir[x] = dirs[(uint64_t)r0 * 64ull + x];
            __syncthreads();
        }
        const int k = (int)j - (wci - W) + off;
        if (k < off || k >= 64 * C || len >= cap) { bad = true; break; }     // cannot happen on a connected band
        near_edge |= k - off < DG_AL_MARGIN || k - off > B - 1 - DG_AL_MARGIN;
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_dir[((int)i - r0) * 64 + k / C]);
        const uint32_t d = (w >> (2 * (k % C))) & 3u;
        if (d == 3u) { bad = true; break; }
        codes = (uint32_t)lane == (len & 63u) ? (int)d : codes;
        len++;
        if ((len & 63u) == 0) path[len - 64u + (uint32_t)lane] = (uint8_t)codes;
        if (d != 2u) {
            if (i == 0) { bad = true; break; }
            i--;
            while (wnum < m) { wnum += n; wci--; }
            wnum -= m;
        }
        if (d != 1u) {
            if (j == 0) { bad = true; break; }
            j--;
        }
    }
    if (bad) { if (lane == 0) p.aln_len[a] = p.first_pass ? DG_AL_RETRY : 0u; return; }
    if (near_edge && p.first_pass) { if (lane == 0) p.aln_len[a] = DG_AL_RETRY; return; }
    if ((uint32_t)lane < (len & 63u)) path[(len & ~63u) + (uint32_t)lane] = (uint8_t)codes;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    // ---- the characters, 64 steps at a time: step s consumed q[i_s - 1] and / or t[j_s - 1] ----
    uint32_t iq = n, jt = m;
    for (uint32_t s0 = 0; s0 < len; s0 += 64) {
        const uint32_t s = s0 + (uint32_t)lane;
        const bool on = s < len;
        const uint32_t d = on ? path[s] : 3u;
        const bool uq = on && d != 2u, ut = on && d != 1u;
        const unsigned long long mq = __ballot(uq), mt = __ballot(ut);
        const unsigned long long lt = (1ull << lane) - 1ull;
        const uint32_t myi = iq - (uint32_t)__popcll(mq & lt), myj = jt - (uint32_t)__popcll(mt & lt);
        if (on) {
            qo[len - 1u - s] = uq ? q[myi - 1u] : (uint8_t)'-';
            to[len - 1u - s] = ut ? t[myj - 1u] : (uint8_t)'-';
        }
        iq -= (uint32_t)__popcll(mq); jt -= (uint32_t)__popcll(mt);
    }
    if (lane == 0) p.aln_len[a] = len;
}

// ---- the band tried before the static ones: DG_AL_WA cells to either side of a centre that FOLLOWS the alignment ----
// Row i's band starts s_i = clamp(a + 1 - DG_AL_WA, 0, 2) columns to the right of row i - 1's, a = the first cell of row
// i - 1 with the smallest score: the centre sits on the diagonal successor of the best prefix alignment so far.  113
// cells per row instead of 300 - 960: two per lane (the structure of k_align_band<2>), 4 bits of direction per lane
// (a byte; lane 0, whose cells lie in front of the band, keeps s_i there for the walk back).  DG_AL_RETRY when the
// band loses the corner (n, m), a row has no reachable cell, or the path comes within DG_AL_MARGIN cells of an edge:
// the static bands decide then (dagcon_align's later passes).  On the synthetic pairs the alignment is the static
// band's, byte for byte, and no pair falls back.
#define DG_AL_WA 56
__host__ __device__ inline uint64_t dg_align_rows_adapt(uint32_t qlen, uint32_t tlen) {
    return (((uint64_t)qlen + 1ull) * 64ull + 255ull) / 256ull + ((uint64_t)qlen + tlen + 255ull) / 256ull + 1ull;
}
__global__ __launch_bounds__(64) void k_align_adapt(DgAlignParams p) {
    constexpr int C = 2, W = DG_AL_WA, B = 2 * W + 1, off = 64 * C - B;
    __shared__ uint8_t s_dir[DG_AL_ROWS * 64];
    __shared__ uint8_t s_qw[128], s_tw[128];
    const uint32_t a = p.idx[blockIdx.x];
    const int lane = threadIdx.x;
    const uint32_t n = p.q_len[a], m = p.t_len[a];
    const uint8_t *q = p.q + p.q_off[a], *t = p.t + p.t_off[a];
    uint8_t *qo = p.qaln + p.out_off[a], *to = p.taln + p.out_off[a];
    if (n == 0 || m == 0) {
        for (uint32_t i = lane; i < n; i += 64) { qo[i] = q[i]; to[i] = '-'; }
        for (uint32_t j = lane; j < m; j += 64) { qo[n + j] = '-'; to[n + j] = t[j]; }
        if (lane == 0) p.aln_len[a] = n + m;
        return;
    }
    uint8_t *dirs = reinterpret_cast<uint8_t *>(p.dirs + p.dir_off[a] * 64ull);
    uint8_t *path = dirs + ((((uint64_t)n + 1ull) * 64ull + 255ull) & ~255ull);
    int P[C], T[C];
    int jl = -W - off + lane * C;                           // column of the lane's first cell
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int j = jl + c;
        P[c] = DG_AL_BIG;
        T[c] = (j >= 1 && j <= (int)m) ? (int)t[j - 1] : 0;
    }
    const int tw0 = W;
    uint32_t tr = 0;
    for (int x = lane; x < 128; x += 64) {
        s_tw[x] = (uint32_t)(tw0 + x) < m ? t[tw0 + x] : (uint8_t)0;
        s_qw[x] = (uint32_t)x < n ? q[x] : (uint8_t)0;
    }
    DG_AL_WAIT_LOADS();
    int tcur = (int)s_tw[0];
    int qc = -1, qcn = (int)s_qw[0];
    int lo = -W;                                            // column of the band's first cell
    bool lost = false;
    const int kb = lane * C;
    uint32_t shift = 0;
    // the band moves on to row i: where the previous row's best cell is (its first one) says how far
    auto advance = [&](const uint32_t i) {
        {
            int mn = P[0] < P[1] ? P[0] : P[1], v;
            v = dg_al_dpp<DG_DPP_ROW_SHR(1), 0xf>(0x7fffffff, mn); mn = v < mn ? v : mn;
            v = dg_al_dpp<DG_DPP_ROW_SHR(2), 0xf>(0x7fffffff, mn); mn = v < mn ? v : mn;
            v = dg_al_dpp<DG_DPP_ROW_SHR(4), 0xf>(0x7fffffff, mn); mn = v < mn ? v : mn;
            v = dg_al_dpp<DG_DPP_ROW_SHR(8), 0xf>(0x7fffffff, mn); mn = v < mn ? v : mn;
            v = dg_al_dpp<DG_DPP_BCAST15, 0xa>(0x7fffffff, mn); mn = v < mn ? v : mn;
            v = dg_al_dpp<DG_DPP_BCAST31, 0xc>(0x7fffffff, mn); mn = v < mn ? v : mn;
            const int best = __builtin_amdgcn_readlane(mn, 63);
            if (best >= DG_AL_LIM) { lost = true; return; }
            const unsigned long long b0 = __ballot(P[0] == best), b1 = __ballot(P[1] == best);
            const int k0 = b0 ? 2 * (__ffsll((long long)b0) - 1) : 1 << 20, k1 = b1 ? 2 * (__ffsll((long long)b1) - 1) + 1 : 1 << 20;
            const int am = (k0 < k1 ? k0 : k1) - off;
            int sft = am + 1 - W;
            sft = sft < 0 ? 0 : sft > 2 ? 2 : sft;
            shift = (uint32_t)sft;
            for (uint32_t s = 0; s < shift; s++) {
                lo++; jl++;
                const int newc = tcur;
                tr++;
                if ((tr & 63u) == 0 && tr >= 64u) {
                    const uint32_t r = tr + 64u + (uint32_t)lane;
                    s_tw[r & 127u] = (uint32_t)tw0 + r < m ? t[(uint32_t)tw0 + r] : (uint8_t)0;
                    DG_AL_WAIT_LOADS();
                }
                tcur = (int)s_tw[tr & 127u];
                int pin = dg_al_dpp<DG_DPP_WAVE_SHL1, 0xf>(DG_AL_BIG, P[0]);
                int tin = dg_al_dpp<DG_DPP_WAVE_SHL1, 0xf>(0, T[0]);
                if (lane == 63) { pin = DG_AL_BIG; tin = newc; }
                P[0] = P[1]; T[0] = T[1];
                P[1] = pin; T[1] = tin;
            }
            qc = qcn;
            if ((i & 63u) == 0 && i >= 64u) {
                const uint32_t r = i + 64u + (uint32_t)lane;
                s_qw[r & 127u] = r < n ? q[r] : (uint8_t)0;
                DG_AL_WAIT_LOADS();
            }
            qcn = (int)s_qw[i & 127u];
        }
    };
    // row i.  EDGE: the band hangs over an end of t on this row (columns < 0 or > m are not cells)
    auto row = [&](const uint32_t i, auto first_row, auto edge_row) {
        constexpr bool FIRST = decltype(first_row)::value, EDGE = decltype(edge_row)::value;
        const int left = dg_al_dpp<DG_DPP_WAVE_SHR1, 0xf>(DG_AL_BIG, P[C - 1]);
        int A[C];
        uint32_t dbits = 0;
        int lm = DG_AL_BIG;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int j = jl + c;
            const bool valid = (!EDGE || (uint32_t)j <= m) && kb + c >= off;
            int best;
            if constexpr (FIRST) best = j == 0 ? 0 : DG_AL_BIG;
            else {
                const int dg = (c == 0 ? left : P[c - 1]) + (T[c] == qc ? DG_AL_MATCH : DG_AL_MISMATCH);
                const int up = P[c] + DG_AL_INS;
                const bool ins = up < dg;
                best = ins ? up : dg;
                dbits |= ins ? 1u << c : 0u;
            }
            best = valid ? best : DG_AL_BIG;
            A[c] = best;
            const int x = best - DG_AL_DEL * (kb + c);
            lm = x < lm ? x : lm;
        }
        int incl = lm, v;
        v = dg_al_dpp<DG_DPP_ROW_SHR(1), 0xf>(0x7fffffff, incl); incl = v < incl ? v : incl;
        v = dg_al_dpp<DG_DPP_ROW_SHR(2), 0xf>(0x7fffffff, incl); incl = v < incl ? v : incl;
        v = dg_al_dpp<DG_DPP_ROW_SHR(4), 0xf>(0x7fffffff, incl); incl = v < incl ? v : incl;
        v = dg_al_dpp<DG_DPP_ROW_SHR(8), 0xf>(0x7fffffff, incl); incl = v < incl ? v : incl;
        v = dg_al_dpp<DG_DPP_BCAST15, 0xa>(0x7fffffff, incl); incl = v < incl ? v : incl;
        v = dg_al_dpp<DG_DPP_BCAST31, 0xc>(0x7fffffff, incl); incl = v < incl ? v : incl;
        int pm = dg_al_dpp<DG_DPP_WAVE_SHR1, 0xf>(0x7fffffff, incl);
        uint32_t word = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int j = jl + c;
            const bool valid = (!EDGE || (uint32_t)j <= m) && kb + c >= off;
            const int x = A[c] - DG_AL_DEL * (kb + c);
            const bool del = j > 0 && pm < x;
            const int sc = del ? pm + DG_AL_DEL * (kb + c) : A[c];
            const uint32_t d = del ? 2u : (dbits >> c) & 1u;
            pm = x < pm ? x : pm;
            P[c] = valid ? sc : DG_AL_BIG;
            word |= d << (2 * c);
        }
        if (lane == 0) word = shift;                        // (its cells lie in front of the band)
        dirs[(uint64_t)i * 64ull + (uint64_t)lane] = (uint8_t)word;
    };
    row(0u, std::true_type{}, std::true_type{});
    for (uint32_t i = 1; i <= n && !lost; i++) {
        advance(i);
        if (lost) break;
        if (lo >= 0 && lo + B - 1 <= (int)m) row(i, std::false_type{}, std::false_type{});
        else row(i, std::false_type{}, std::true_type{});
    }
    // (n, m) must be a reachable cell of the last row
    const int kend = (int)m - lo + off;
    bool fail = lost || kend < off || kend >= 64 * C;
    if (!fail) {
        int fin = (kend & 1) ? P[1] : P[0];
        fin = __builtin_amdgcn_readlane(fin, kend / C);
        fail = fin >= DG_AL_LIM;
    }
    if (fail) { if (lane == 0) p.aln_len[a] = DG_AL_RETRY; return; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // ---- walk back (uniform), rows of directions through LDS ----
    uint32_t i = n, j = m, len = 0;
    int r0 = (int)n + 1, codes = 0;
    const uint32_t cap = n + m;
    bool bad = false;
    while (i > 0 || j > 0) {
        if ((int)i < r0) {
            __syncthreads();
            r0 = (int)i >= DG_AL_ROWS - 1 ? (int)i - (DG_AL_ROWS - 1) : 0;
            for (uint32_t x = lane; x < ((uint32_t)((int)i - r0) + 1u) * 64u; x += 64) s_dir[x] = dirs[(uint64_t)r0 * 64ull + x];
            __syncthreads();
        }
        const int k = (int)j - lo + off;
        if (k < off + DG_AL_MARGIN || k > off + B - 1 - DG_AL_MARGIN || len >= cap) { bad = true; break; }
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_dir[((int)i - r0) * 64 + k / C]);
        const uint32_t d = (w >> (2 * (k % C))) & 3u;
        if (d == 3u) { bad = true; break; }
        codes = (uint32_t)lane == (len & 63u) ? (int)d : codes;
        len++;
        if ((len & 63u) == 0) path[len - 64u + (uint32_t)lane] = (uint8_t)codes;
        if (d != 2u) {
            if (i == 0) { bad = true; break; }
            lo -= (int)__builtin_amdgcn_readfirstlane((int)s_dir[((int)i - r0) * 64]);      // s_i
            i--;
        }
        if (d != 1u) {
            if (j == 0) { bad = true; break; }
            j--;
        }
    }
    if (bad) { if (lane == 0) p.aln_len[a] = DG_AL_RETRY; return; }
    if ((uint32_t)lane < (len & 63u)) path[(len & ~63u) + (uint32_t)lane] = (uint8_t)codes;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    uint32_t iq = n, jt = m;
    for (uint32_t s0 = 0; s0 < len; s0 += 64) {
        const uint32_t s = s0 + (uint32_t)lane;
        const bool on = s < len;
        const uint32_t d = on ? path[s] : 3u;
        const bool uq = on && d != 2u, ut = on && d != 1u;
        const unsigned long long mq = __ballot(uq), mt = __ballot(ut);
        const unsigned long long lt = (1ull << lane) - 1ull;
        const uint32_t myi = iq - (uint32_t)__popcll(mq & lt), myj = jt - (uint32_t)__popcll(mt & lt);
        if (on) {
            qo[len - 1u - s] = uq ? q[myi - 1u] : (uint8_t)'-';
            to[len - 1u - s] = ut ? t[myj - 1u] : (uint8_t)'-';
        }
        iq -= (uint32_t)__popcll(mq); jt -= (uint32_t)__popcll(mt);
    }
    if (lane == 0) p.aln_len[a] = len;
}

// SimpleAligner.cpp:57-58 for the records of the '-' strand: both aligned strings reverse-complemented in place
// (Alignment.cpp:15-26: upper-case A, C, G, T are complemented, everything else, the gaps too, only moves)
__device__ __forceinline__ uint8_t dg_al_comp(uint8_t ch) {
    return ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch;
}
__global__ __launch_bounds__(64) void k_align_revcomp(uint8_t *qaln, uint8_t *taln, const uint64_t *out_off, const uint32_t *aln_len,
                                                      const uint32_t *idx) {
    const uint32_t a = idx[blockIdx.x];
    const uint32_t len = aln_len[a];
    for (int w = 0; w < 2; w++) {
        uint8_t *sq = (w ? taln : qaln) + out_off[a];
        for (uint32_t x = threadIdx.x; x < (len + 1u) / 2u; x += 64) {
            const uint32_t y = len - 1u - x;
            const uint8_t lo = sq[x], hi = sq[y];
            sq[x] = dg_al_comp(hi); sq[y] = dg_al_comp(lo);       // (x == y: the middle one, complemented once)
        }
    }
}
