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
// j = i * tlen / qlen (the role of GuidedAlign's band around the SDP chain).  The tests hold a bit-exact
// CPU twin of it.
//
// One wave per alignment.  A row of the band (<= 961 cells) lives in LDS; lane l owns C = ceil(B / 64)
// consecutive cells.  The dependency on the cell to the left inside a row (deletions) is a running
// minimum, S[k] = min_{k' <= k} (A[k'] - 5 k') + 5 k with A = min(diagonal, insertion), so a row is C
// sequential steps per lane plus one wave-wide prefix minimum instead of B sequential steps.  Two bits
// of direction per cell go to HBM as one coalesced 256-byte store per row; the walk back from
// (qlen, tlen) reads them through LDS, 64 rows at a time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DG_AL_MATCH (-5)
#define DG_AL_MISMATCH 6
#define DG_AL_INS 4
#define DG_AL_DEL 5
#define DG_AL_INF (1 << 28)
#define DG_AL_MAXW 480u
#define DG_AL_ROWS 64            // rows of directions staged in LDS during the walk back

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

struct DgAlignParams {
    const uint8_t *q, *t;          // sequence blobs
    const uint64_t *q_off, *t_off;
    const uint32_t *q_len, *t_len;
    const uint64_t *out_off;       // per alignment: room for q_len + t_len columns in qaln / taln
    uint8_t *qaln, *taln;          // written from the BACK of each alignment's room
    uint32_t *aln_len;
    uint32_t *dirs;                // direction words: 64 per row
    const uint64_t *dir_off;       // per alignment, in rows
    uint32_t first, n;             // alignments [first, first + n) of the arrays
};

__global__ __launch_bounds__(64) void k_align_banded(DgAlignParams p) {
    __shared__ int32_t s_row[2][1024];
    __shared__ uint32_t s_dir[DG_AL_ROWS * 64];
    const uint32_t a = p.first + blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t n = p.q_len[a], m = p.t_len[a];
    const uint8_t *q = p.q + p.q_off[a], *t = p.t + p.t_off[a];
    uint8_t *qo = p.qaln + p.out_off[a], *to = p.taln + p.out_off[a];
    const uint32_t cap = n + m;
    if (n == 0 || m == 0) {
        // all gaps, in order: query columns first
        for (uint32_t i = lane; i < n; i += 64) { qo[i] = q[i]; to[i] = '-'; }
        for (uint32_t j = lane; j < m; j += 64) { qo[n + j] = '-'; to[n + j] = t[j]; }
        if (lane == 0) p.aln_len[a] = cap;
        return;
    }
    const uint32_t W = dg_align_halfwidth(n, m), B = 2u * W + 1u;
    const uint32_t C = (B + 63u) / 64u;                     // cells per lane (<= 16)
    uint32_t *dirs = p.dirs + p.dir_off[a] * 64ull;
    const uint32_t k0 = (uint32_t)lane * C;
    for (uint32_t k = lane; k < 1024; k += 64) { s_row[0][k] = DG_AL_INF; s_row[1][k] = DG_AL_INF; }
    __syncthreads();
    int64_t cp = 0;
    for (uint32_t i = 0; i <= n; i++) {
        const int64_t ci = (int64_t)((uint64_t)i * m / n);
        const int32_t *prev = s_row[(i & 1u) ^ 1u];
        int32_t *cur = s_row[i & 1u];
        const int shift = (int)(ci - cp);
        const uint8_t qc = i ? q[i - 1] : 0;
        // A = min(diagonal, insertion) for the lane's cells, then the running minimum over the row
        int32_t A[16];
        uint32_t dbits = 0;                                 // bit k: 1 = insertion wins over the diagonal
        int32_t run = DG_AL_INF;                            // min over this lane's cells so far of (A[k] - DEL k)
        int32_t X[16];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            A[c] = DG_AL_INF; X[c] = DG_AL_INF;
            if ((uint32_t)c < C) {
                const uint32_t k = k0 + (uint32_t)c;
                const int64_t j = ci - (int64_t)W + (int64_t)k;
                if (k < B && j >= 0 && j <= (int64_t)m) {
                    int32_t best = DG_AL_INF;
                    if (i == 0 && j == 0) best = 0;
                    if (i > 0) {
                        const int64_t kd = (int64_t)k + shift - 1, ku = kd + 1;
                        if (j > 0 && kd >= 0 && kd < (int64_t)B) {
                            const int32_t pv = prev[kd];
                            if (pv < DG_AL_INF) best = pv + (qc == t[j - 1] ? DG_AL_MATCH : DG_AL_MISMATCH);
                        }
                        if (ku >= 0 && ku < (int64_t)B) {
                            const int32_t pv = prev[ku];
                            if (pv < DG_AL_INF && pv + DG_AL_INS < best) { best = pv + DG_AL_INS; dbits |= 1u << c; }
                        }
                    }
                    A[c] = best;
                    if (best < DG_AL_INF) X[c] = best - DG_AL_DEL * (int32_t)k;
                }
            }
        }
        // exclusive prefix minimum of the lanes' minima
#pragma unroll
        for (int c = 0; c < 16; c++) if ((uint32_t)c < C && X[c] < run) run = X[c];
        int32_t incl = run;
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t up = __shfl_up(incl, o);
            if (lane >= o && up < incl) incl = up;
        }
        int32_t P = __shfl_up(incl, 1);
        if (lane == 0) P = DG_AL_INF;
        uint32_t word = 0;
#pragma unroll
        for (int c = 0; c < 16; c++) {
            if ((uint32_t)c < C) {
                const uint32_t k = k0 + (uint32_t)c;
                const int64_t j = ci - (int64_t)W + (int64_t)k;
                const bool valid = k < B && j >= 0 && j <= (int64_t)m;
                uint32_t d = 3u;
                int32_t S = DG_AL_INF;
                if (valid) {
                    // deletion (left) wins only when strictly better than diagonal / insertion
                    if (j > 0 && P < X[c]) { S = P + DG_AL_DEL * (int32_t)k; d = 2u; }
                    else if (A[c] < DG_AL_INF) { S = A[c]; d = (dbits >> c) & 1u; }
                    if (i == 0 && j == 0) d = 3u;
                    if (X[c] < P) P = X[c];
                }
                if (k < 1024u) cur[k] = S;
                word |= d << (2 * c);
            }
        }
        dirs[(uint64_t)i * 64ull + (uint64_t)lane] = word;
        cp = ci;
        __syncthreads();
    }
    // ---- walk back from (n, m): directions through LDS, DG_AL_ROWS rows at a time ----
    uint32_t i = n, j = m, len = 0;
    __shared__ uint32_t s_state[3];
    while (i > 0 || j > 0) {
        const uint32_t r1 = i, r0 = i >= DG_AL_ROWS - 1 ? i - (DG_AL_ROWS - 1) : 0u;     // rows [r0, r1]
        for (uint32_t x = lane; x < (r1 - r0 + 1u) * 64u; x += 64) s_dir[x] = dirs[(uint64_t)r0 * 64ull + x];
        __syncthreads();
        if (lane == 0) {
            while ((i > 0 || j > 0) && i >= r0) {
                const int64_t ci = (int64_t)((uint64_t)i * m / n);
                const uint32_t k = (uint32_t)((int64_t)j - (ci - (int64_t)W));
                const uint32_t d = (s_dir[(i - r0) * 64u + k / C] >> (2u * (k % C))) & 3u;
                uint8_t qb, tb;
                if (d == 0u) { qb = q[--i]; tb = t[--j]; }
                else if (d == 1u) { qb = q[--i]; tb = '-'; }
                else if (d == 2u) { qb = '-'; tb = t[--j]; }
                else { i = 0; j = 0; break; }               // cannot happen: (0, 0) is inside the band
                len++;
                qo[cap - len] = qb; to[cap - len] = tb;
                if (i < r0) break;
            }
            s_state[0] = i; s_state[1] = j; s_state[2] = len;
        }
        __syncthreads();
        i = s_state[0]; j = s_state[1]; len = s_state[2];
        __syncthreads();
    }
    if (lane == 0) p.aln_len[a] = len;
}
