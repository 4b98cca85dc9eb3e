// k_align2.hip.h -- EXPERIMENT (make experiments, DAGCON_ALIGN2=1), measured and not adopted.
//
// Bit-equal to k_align_adapt and to the CPU twin on the whole suite at its first run -- and slower where it was meant
// to help: 3,840 pairs of 50 kb (64 targets x 60x) 86 ms against 60.  Half the waves (1,920 on 1,024 SIMDs) carry
// longer dependent chains each (four cells a lane, the per-half scalars in vector registers), and with fewer than two
// waves per SIMD nothing hides them: the kernel is bound by the latency of one wave's row, not by issue slots.  It
// would pay where twice as many pairs are in flight (direction rows of 32 bytes instead of 64 would allow that inside
// the same buffer): not built.
#pragma once
#include "../k_align.hip.h"

// ---- the band that follows the alignment, TWO pairs per wave (round 3) -----------------------------------------------
// k_align_adapt issues 234 instructions per row of 113 cells, and most of them do not care how many cells a lane holds:
// two DPP minima and a prefix minimum over the lanes, the band's shift, window bookkeeping, the loop.  Here a pair takes
// HALF a wave -- 32 lanes x 4 cells, the same 128 cell slots -- and a wave aligns two pairs side by side: the reductions
// stop at the half's edge (row shifts inside the 16-lane rows, one row broadcast inside the half; no broadcast across it),
// what was scalar per pair (band position, shift, windows' cursors, lengths) lives in vector registers, uniform over the
// half, and the two halves run through their rows in lock step (the pairs of a launch are sorted by length).  Per pair
// the arithmetic is k_align_adapt's, cell for cell and tie for tie: the CPU twin does not know the difference.
__global__ __launch_bounds__(64) void k_align_adapt2(DgAlignParams p) {
    constexpr int C = 4, W = DG_AL_WA, B = 2 * W + 1, off = 32 * C - B;
    __shared__ uint8_t s_dir[2][DG_AL_ROWS * 32];
    __shared__ uint8_t s_qw[2][128], s_tw[2][128];
    const int lane = threadIdx.x, half = lane >> 5, l = lane & 31;
    const uint32_t slot = 2u * blockIdx.x + (uint32_t)half;
    const bool have = slot < p.n;
    const uint32_t a = p.idx[have ? slot : 2u * blockIdx.x];
    const uint32_t n = have ? p.q_len[a] : 0u, m = have ? p.t_len[a] : 0u;
    const uint8_t *q = p.q + p.q_off[a], *t = p.t + p.t_off[a];
    uint8_t *qo = p.qaln + p.out_off[a], *to = p.taln + p.out_off[a];
    bool act = have;                                        // the half still has rows to do
    if (have && (n == 0 || m == 0)) {
        for (uint32_t i = l; i < n; i += 32) { qo[i] = q[i]; to[i] = '-'; }
        for (uint32_t j = l; j < m; j += 32) { qo[n + j] = '-'; to[n + j] = t[j]; }
        if (l == 0) p.aln_len[a] = n + m;
        act = false;
    }
    const bool pair_ok = act;
    uint8_t *dirs = reinterpret_cast<uint8_t *>(p.dirs + p.dir_off[a] * 64ull);
    uint8_t *path = dirs + ((((uint64_t)n + 1ull) * 64ull + 255ull) & ~255ull);
    uint8_t *sq = s_qw[half], *st = s_tw[half], *sd = s_dir[half];
    int P[C], T[C];
    int jl = -W - off + l * C;                              // column of the lane's first cell
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int j = jl + c;
        P[c] = DG_AL_BIG;
        T[c] = (act && j >= 1 && j <= (int)m) ? (int)t[j - 1] : 0;
    }
    const int tw0 = W;
    uint32_t tr = 0;
    if (act) {
        for (int x = l; x < 128; x += 32) {
            st[x] = (uint32_t)(tw0 + x) < m ? t[tw0 + x] : (uint8_t)0;
            sq[x] = (uint32_t)x < n ? q[x] : (uint8_t)0;
        }
    }
    DG_AL_WAIT_LOADS();
    __syncthreads();
    int tcur = (int)st[0];
    int qc = -1, qcn = (int)sq[0];
    int lo = -W;                                            // column of the band's first cell
    bool lost = false;
    const int kb = l * C;
    uint32_t shift = 0;
    const uint32_t nmax = (uint32_t)max(__shfl((int)(pair_ok ? n : 0u), 0), __shfl((int)(pair_ok ? n : 0u), 32));
    // minimum over the half, in its last lane: row shifts inside the rows of 16, then row 0 -> 1 and row 2 -> 3
#define DG_AL2_HALF_MIN(X)                                                                  \
    do {                                                                                    \
        int v_;                                                                             \
        v_ = dg_al_dpp<DG_DPP_ROW_SHR(1), 0xf>(0x7fffffff, X); X = v_ < X ? v_ : X;         \
        v_ = dg_al_dpp<DG_DPP_ROW_SHR(2), 0xf>(0x7fffffff, X); X = v_ < X ? v_ : X;         \
        v_ = dg_al_dpp<DG_DPP_ROW_SHR(4), 0xf>(0x7fffffff, X); X = v_ < X ? v_ : X;         \
        v_ = dg_al_dpp<DG_DPP_ROW_SHR(8), 0xf>(0x7fffffff, X); X = v_ < X ? v_ : X;         \
        v_ = dg_al_dpp<DG_DPP_BCAST15, 0xa>(0x7fffffff, X); X = v_ < X ? v_ : X;            \
    } while (0)
    // the band moves on to row i: where the previous row's best cell is (its first one) says how far
    auto advance = [&](const uint32_t i) {
        int mn = P[0];
#pragma unroll
        for (int c = 1; c < C; c++) mn = P[c] < mn ? P[c] : mn;
        DG_AL2_HALF_MIN(mn);
        // (the half's minimum sits in its last lane: two scalar reads and a select, no trip through the LDS crossbar)
        const int best0 = __builtin_amdgcn_readlane(mn, 31), best1 = __builtin_amdgcn_readlane(mn, 63);
        const int best = half ? best1 : best0;
        if (act && best >= DG_AL_LIM) { lost = true; act = false; }
        int am = 1 << 20;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const unsigned long long bb = __ballot(act && P[c] == best);
            const uint32_t b0 = (uint32_t)bb, b1 = (uint32_t)(bb >> 32);
            const int k0 = b0 ? C * (__ffs((int)b0) - 1) + c : 1 << 20, k1 = b1 ? C * (__ffs((int)b1) - 1) + c : 1 << 20;   // (scalar)
            const int k = half ? k1 : k0;
            am = k < am ? k : am;
        }
        int sft = am - off + 1 - W;
        sft = sft < 0 ? 0 : sft > 2 ? 2 : sft;
        shift = act ? (uint32_t)sft : 0u;
#pragma unroll
        for (uint32_t s = 0; s < 2u; s++) {
            const bool dos = s < shift;                     // (uniform over the half)
            const int newc = tcur;
            if (dos) {
                lo++; jl++; tr++;
                if ((tr & 63u) == 0 && tr >= 64u) {
                    for (uint32_t x = (uint32_t)l; x < 64u; x += 32u) {
                        const uint32_t r = tr + 64u + x;
                        st[r & 127u] = (uint32_t)tw0 + r < m ? t[(uint32_t)tw0 + r] : (uint8_t)0;
                    }
                    DG_AL_WAIT_LOADS();
                }
                tcur = (int)st[tr & 127u];
            }
            int pin = dg_al_dpp<DG_DPP_WAVE_SHL1, 0xf>(DG_AL_BIG, P[0]);
            int tin = dg_al_dpp<DG_DPP_WAVE_SHL1, 0xf>(0, T[0]);
            if (l == 31) { pin = DG_AL_BIG; tin = newc; }
            if (dos) {
#pragma unroll
                for (int c = 0; c + 1 < C; c++) { P[c] = P[c + 1]; T[c] = T[c + 1]; }
                P[C - 1] = pin; T[C - 1] = tin;
            }
        }
        qc = qcn;
        if ((i & 63u) == 0 && i >= 64u) {
            if (act) {
                for (uint32_t x = (uint32_t)l; x < 64u; x += 32u) {
                    const uint32_t r = i + 64u + x;
                    sq[r & 127u] = r < n ? q[r] : (uint8_t)0;
                }
            }
            DG_AL_WAIT_LOADS();
        }
        qcn = (int)sq[i & 127u];
    };
    // row i.  EDGE: a band hangs over an end of t on this row (columns < 0 or > m are not cells)
    auto row = [&](const uint32_t i, auto first_row, auto edge_row) {
        constexpr bool FIRST = decltype(first_row)::value, EDGE = decltype(edge_row)::value;
        int left = dg_al_dpp<DG_DPP_WAVE_SHR1, 0xf>(DG_AL_BIG, P[C - 1]);
        if (l == 0) left = DG_AL_BIG;                       // (the other half's last cell is nobody's neighbour)
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
        int incl = lm;
        DG_AL2_HALF_MIN(incl);
        int pm = dg_al_dpp<DG_DPP_WAVE_SHR1, 0xf>(0x7fffffff, incl);
        if (l == 0) pm = 0x7fffffff;
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
            if (act) P[c] = valid ? sc : DG_AL_BIG;
            word |= d << (2 * c);
        }
        if (l == 0) word = shift;                           // (its cells lie in front of the band)
        if (act) dirs[(uint64_t)i * 64ull + (uint64_t)l] = (uint8_t)word;
    };
    row(0u, std::true_type{}, std::true_type{});
    for (uint32_t i = 1; i <= nmax; i++) {
        if (act && i > n) act = false;                      // the shorter pair of the two is through
        if (!__ballot(act)) break;
        advance(i);
        const bool edge = act && !(lo >= 0 && lo + B - 1 <= (int)m);
        if (__ballot(edge)) row(i, std::false_type{}, std::true_type{});
        else row(i, std::false_type{}, std::false_type{});
    }
#undef DG_AL2_HALF_MIN
    // (n, m) must be a reachable cell of the last row
    const int kend = (int)m - lo + off;
    bool fail = !pair_ok || lost || kend < off || kend >= 32 * C;
    {
        const int ke = fail ? off : kend;
        int sel = P[0];
#pragma unroll
        for (int c = 1; c < C; c++) sel = (ke & (C - 1)) == c ? P[c] : sel;
        const int fin = __shfl(sel, ke / C, 32);
        fail = fail || fin >= DG_AL_LIM;
    }
    if (pair_ok && fail && l == 0) p.aln_len[a] = DG_AL_RETRY;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    // ---- walk back, each half its own pair (the halves step in lock step until the shorter path ends) ----
    uint32_t i = n, j = m, len = 0;
    int r0 = (int)n + 1, codes = 0;
    const uint32_t cap = n + m;
    bool bad = false, go = pair_ok && !fail;
    while (go && (i > 0 || j > 0)) {
        if ((int)i < r0) {
            r0 = (int)i >= DG_AL_ROWS - 1 ? (int)i - (DG_AL_ROWS - 1) : 0;
            const uint32_t nb = ((uint32_t)((int)i - r0) + 1u) * 32u;
            for (uint32_t x = (uint32_t)l; x < nb; x += 32u) sd[x] = dirs[((uint64_t)r0 + (x >> 5)) * 64ull + (x & 31u)];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        const int k = (int)j - lo + off;
        if (k < off + DG_AL_MARGIN || k > off + B - 1 - DG_AL_MARGIN || len >= cap) { bad = true; break; }
        const uint32_t w = (uint32_t)sd[((int)i - r0) * 32 + k / C];
        const uint32_t d = (w >> (2 * (k % C))) & 3u;
        if (d == 3u) { bad = true; break; }
        codes = (uint32_t)l == (len & 31u) ? (int)d : codes;
        len++;
        if ((len & 31u) == 0) path[len - 32u + (uint32_t)l] = (uint8_t)codes;
        if (d != 2u) {
            if (i == 0) { bad = true; break; }
            lo -= (int)sd[((int)i - r0) * 32];               // s_i
            i--;
        }
        if (d != 1u) {
            if (j == 0) { bad = true; break; }
            j--;
        }
    }
    if (go && bad) { if (l == 0) p.aln_len[a] = DG_AL_RETRY; go = false; }
    if (go && (uint32_t)l < (len & 31u)) path[(len & ~31u) + (uint32_t)l] = (uint8_t)codes;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    uint32_t iq = n, jt = m;
    const uint32_t lenx = go ? len : 0u;
    for (uint32_t s0 = 0; s0 < lenx; s0 += 32) {
        const uint32_t s = s0 + (uint32_t)l;
        const bool on = s < lenx;
        const uint32_t d = on ? path[s] : 3u;
        const bool uq = on && d != 2u, ut = on && d != 1u;
        const uint32_t mq = (uint32_t)(__ballot(uq) >> (32 * half)), mt = (uint32_t)(__ballot(ut) >> (32 * half));
        const uint32_t lt = (1u << l) - 1u;
        const uint32_t myi = iq - (uint32_t)__popc(mq & lt), myj = jt - (uint32_t)__popc(mt & lt);
        if (on) {
            qo[lenx - 1u - s] = uq ? q[myi - 1u] : (uint8_t)'-';
            to[lenx - 1u - s] = ut ? t[myj - 1u] : (uint8_t)'-';
        }
        iq -= (uint32_t)__popc(mq); jt -= (uint32_t)__popc(mt);
    }
    if (go && l == 0) p.aln_len[a] = len;
}

