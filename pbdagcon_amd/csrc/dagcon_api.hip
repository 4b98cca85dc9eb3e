// dagcon_api.hip -- host side of the C ABI (include/dagcon.h).
//
// Owns the HIP stream, the HBM arenas and the launch sequence of the hot path
//   a1 k_norm_*             | a2 k_carve, k_groups, k_emit, k_lists |
//   b  k_merge              | c  k_bestpath
// There is no CPU fallback anywhere in this file: without a HIP device
// dagcon_create fails with DAGCON_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/dagcon.h"
#include "dagcon_dev.h"
#include "k_build.hip.h"
#include "k_merge.hip.h"
#ifdef DG_EXPERIMENTS
#include "experiments/k_align2.hip.h"        // the following band with two pairs per wave: exact, 86 ms against 60
#include "experiments/k_emit2.hip.h"         // addAln with a thread per column: exact, and 15.6 ms of build against 12.4
#include "experiments/k_merge_tile.hip.h"     // dropped experiments: `make experiments` only, never in the shipped library
#endif
#include "k_merge_q.hip.h"
#include "k_bestpath.hip.h"
#include "k_align.hip.h"

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// ---- how many pieces the merge / bestPath sweeps of a batch are cut into (host arithmetic only: exported as
// dagcon_debug_plan so that a CPU test can sweep it; every grid size derived from it is > 0) ----
#define DQ_KMAX 52u      // reads per target up to which the row sweep (k_merge_q, rows of 8 lanes) beats the wave sweep (k_merge):
                         // 600 targets x 6 kb at 40x / 50x / 60x / 70x: 6.5 / 8.4 / 11.6 / 20.0 ms against 7.1 / 8.4 / 9.7 / 11.2 (tools/kmax_probe.py)
struct DgPlanIn { uint32_t T; uint64_t n_alns, sum_bb; uint32_t gcuts, max_segments, min_segment_len, seg_env, merge_q; };
struct DgPlan { uint32_t seg_max, seg_min, use_q, bp_max; };
static DgPlan dg_plan_pieces(const DgPlanIn &in) {
    DgPlan pl;
    const uint32_t T = in.T;
    // shortest stretch worth a worker: 768 positions when that already fills the chip, shorter (down to 192)
    // for small batches, whose waves would otherwise be few and long
    pl.seg_min = in.min_segment_len;
    if (!pl.seg_min) pl.seg_min = (uint32_t)std::min<uint64_t>(768, std::max<uint64_t>(192, in.sum_bb / 8192));
    // merge workers per target: about one chip's worth of resident waves (8 per SIMD x 1024
    // SIMDs) over the batch, never fewer than 8 nor more than 256 per target
    if (in.max_segments) pl.seg_max = in.max_segments > 64u ? 64u : in.max_segments;
    else if (in.seg_env) pl.seg_max = in.seg_env;
    else if (in.gcuts) pl.seg_max = 64;      // the worklist of k_cuts2 is taken by ticket: the finer its entries the better
                                             // the balance (config-5 shape, 1,000 targets: 8 / 32 / 64 pieces 54 / 34 / 31 ms)
    else { uint32_t sm = T ? 8192u / T : 8u; pl.seg_max = sm < 8u ? 8u : sm > 256u ? 256u : sm; }
    if (in.gcuts && !in.min_segment_len) pl.seg_min = 256;
    // k_merge_q (DQ_ROWS segments per wave, DQ_WAVES waves per SIMD) for full-span batches big enough to fill the chip with
    // it: as many pieces as go (<= 256 per target) with its waves filling the chip a whole number of times -- a last round
    // that is a third full costs as much as a full one (configs[1]: 36 / 49 / 56 / 64 pieces 20.6 / 17.4 / 19.2 / 18.3 ms)
    pl.use_q = 0;
    if (in.merge_q && !in.gcuts && pl.seg_max != 1) {
        const uint32_t slots = 1024u * DQ_WAVES;
        if (in.max_segments || in.seg_env) pl.use_q = 1;                            // (the caller's number of pieces)
        // a row holds 4 + 4 list entries in its one-look path and 8 in the generic one: past ~50 reads per target
        // too many visits outgrow it (DQ_KMAX)
        else if (T && in.n_alns <= (uint64_t)DQ_KMAX * T) {
            // pieces a target can give: up to 256, one per 128 positions of the average backbone
            const uint64_t avail = std::min<uint64_t>(256, std::max<uint64_t>(1, in.sum_bb / T / 128));
            const uint64_t k = (uint64_t)T * avail / DQ_ROWS / slots;                     // whole rounds at that many pieces
            if (k >= 1 || (uint64_t)T * avail / DQ_ROWS * 10u >= 6u * slots) {            // (or one round six tenths full)
                // (very many short targets -- more targets than a round has rows: unless every target gets at
                // least two pieces the wave-per-segment kernel keeps the batch)
                const uint64_t sm = std::min<uint64_t>(avail, std::max<uint64_t>(k, 1) * slots * DQ_ROWS / T);
                if (sm >= 2) { pl.seg_max = (uint32_t)sm; pl.use_q = 1; }
            }
        }
    }
    if (pl.seg_max < 1) pl.seg_max = 1;
    if (pl.use_q && !in.min_segment_len) pl.seg_min = 128;                         // (its pieces are a quarter of a wave's work)
    // bestPath is swept in three times as many pieces: its waves are light (one piece = one
    // sequential sweep when that is asked for)
    // (more than 64 of them only where 64 per target leave the chip short of waves; never on the partial-span path)
    pl.bp_max = pl.seg_max == 1 ? 1u : std::min(in.gcuts || T >= 256u ? 64u : (uint32_t)DG_BP_PIECES, 3u * pl.seg_max);
    if (pl.bp_max < 1) pl.bp_max = 1;
    return pl;
}

struct Ctx {
    dagcon_opts opts;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    std::string err;
    bool uploaded = false, ran = false, fetched = false;

    // host copy of the filtered batch
    uint32_t T = 0, A = 0;
    int emit2 = 0;                                 // addAln with a thread per column (k_emit2.hip.h; DAGCON_EMIT2)
    uint32_t max_len = 0, bs_stride = 0;
    DevBuf d_matK, d_bbstart;
    int bp_lane = 1, bl_stk = -1;                  // full-span bestPath: a row of eight lanes per piece (k_bp_sweep_l; DAGCON_BP_LANE=0: a wave per piece, k_bp_sweep; 2: rows whatever the batch size); DAGCON_BP_LANE_STACK: test knob
    int bp_fused = 1;                              // partial-span bestPath: one (A, B) sweep + vertex-parallel kernels (DAGCON_BP_FUSED=0: three sweeps)
    uint32_t align_dropped = 0;                    // records of the last dagcon_align / dagcon_consensus_pre the band could not align
    int emit_scan = 1;                             // k_emit takes the prefix over the reads itself (DAGCON_EMIT_SCAN=0: k_groups, as for deeper targets)
    int nf2 = 1;                                   // k_norm_finish2 (a wave per chunk) instead of k_norm_finish (DAGCON_NF2=0)
    int poison = 0;                                // DAGCON_POISON (tests): arenas nobody clears are filled with 0xEE bytes before every run (bits 1, 2, 4); 8: every buffer the kernels fill
    int align2 = 0;                                // (make experiments) k_align_adapt2: two pairs per wave, DAGCON_ALIGN2=1
    int fold = 1;                                  // duplicate insertion chains folded by k_emit (DAGCON_FOLD=0: never)
    int list_q = 0;                                // (make experiments) partial-span worklist by rows, k_merge_list_q: DAGCON_MERGE_LIST_Q=1
    int merge_q = 1, use_q = 0;                    // k_merge_q: eight segments per wave (DAGCON_MERGE_Q=0: never); this batch
    uint32_t max_k = 0, max_tlen = 0;
    uint64_t sum_len = 0, sum_bb = 0, mat_cells = 0, blob_bytes = 0;
    bool have_bb = false;
    std::vector<uint32_t> h_tlen, h_aln_len, h_aln_start, h_aln_tgt;
    std::vector<uint64_t> h_aln_begin, h_aln_off, h_mat_base, h_bbv_base, h_bb_off, h_matc_base;
    std::vector<uint32_t> h_matc_stride;
    uint64_t matc_cells = 0;
    std::vector<uint8_t> h_tactive;
    std::vector<uint32_t> h_ch_base, h_ch_aln;      // chunk tables of k_norm_*
    std::vector<uint64_t> h_norm_off;               // column buffer of each alignment
    std::vector<uint32_t> h_ck_base;                // first k_emit checkpoint of each alignment
    uint64_t n_ckpt = 0;
    uint32_t emit_shift = 9;                        // 512 backbone positions per k_emit wave
    uint32_t n_chunks = 0;
    uint64_t tmp_main = 0, tmp_cap = 0;

    // device buffers
    DevBuf d_tfail, d_q, d_t, d_aln_off, d_aln_len, d_aln_start, d_aln_tgt, d_tlen, d_aln_begin, d_tactive,
        d_bb, d_bb_off, d_mat_base, d_bbv_base, d_matc_base, d_matc_stride;
    DevBuf d_nmis, d_norm_off, d_n_lo, d_n_hi, d_n_start, d_n_ins, d_n_del, d_norm;
    DevBuf d_ch_aln, d_ch_base, d_ch_k0, d_ch_next, d_ch_w, d_ch_tb, d_ch_flag, d_ch_src, d_ch_out, d_ch_adv,
        d_n_lb, d_norm_tmp, d_ckpt, d_ck_base;
    DevBuf d_node_base, d_n_nodes, d_pool_base, d_pool_size, d_pool_top, d_t_nins;
    DevBuf d_matA, d_matD, d_matC, d_cov, d_gcount, d_gbase, d_bid;
    DevBuf d_nodes, d_best, d_queue, d_score, d_cns_tmp, d_bp_tt, d_score_b;
    DevBuf d_pool, d_stk, d_cuts, d_cuts_bp, d_bp_stat, d_bp_len, d_nextcut, d_tile_list, d_rd, d_pro_state, d_sh_cnt, d_seg_done, d_wl_first, d_queue0, d_bp_end, d_bp_ab, d_defer, d_cns_tmp0;
    DevBuf d_al[14];                                // dagcon_align: blobs, offsets, outputs, directions, launch order, widths
    DevBuf d_cns, d_cns_off, d_cns_len, d_seg_first, d_n_seg, d_seg_r0, d_seg_r1, d_st;

    uint64_t norm_cap = 0, node_cap = 0, pool_cap = 0, cns_cap = 0, seg_cap = 0;
    uint32_t stk_words = 4096, growth_pct = 100, seg_max = 8, bp_max = 16, seg_env = 0, seg_min = 768;    // (scratch per target and segment: grown x4 and re-run on DG_E_STACK)
    uint32_t sh_log = 16;                           // slots per segment behind enter's / exit's list (x2 on DG_E_LOG_OVF)
    bool full_span = false;                         // (nearly) every alignment of the batch covers its whole target
    uint32_t gcuts = 1;                             // partial-span cuts: prologue + worklist + epilogue (DAGCON_GCUTS=0: off)
    uint32_t tile_pos = 0, tile_words = 0, tile_ny = 0, tile_list_cap = 0, list_grid = 8192;   // LDS tiles (tile_pos 0: off)
    double ins_per_pos = -1.0;                      // inserted vertices per backbone position, from the last run
    uint64_t expected_workers = 0;                  // merge workers the batch will probably run (prefetch on / off)

    DgStatus h_st;
    dagcon_timings tm;

    // results (host)
    std::vector<uint64_t> r_seg_begin, r_seq_off, r_cns_off, r_seg_first;
    std::vector<int32_t> r_range0, r_range1, r_tmp0, r_tmp1;
    std::vector<uint32_t> r_seq_len, r_cns_len, r_n_seg, r_tfail;
    std::vector<int32_t> r_status;
    char *r_blob = nullptr;             // page-locked: the consensus blob comes back at PCIe speed
    size_t r_blob_cap = 0;

    // debug dump storage
    std::vector<uint8_t> g_base, g_deleted, g_backbone;
    std::vector<int32_t> g_weight, g_cov, g_bbpos, g_out_dst, g_out_cnt, g_in_src;
    std::vector<uint32_t> g_out_begin, g_in_begin;
};

int fail(Ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                    \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail((c), DAGCON_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                        hipGetErrorString(_e), __FILE__, __LINE__);                        \
    } while (0)

int ensure(Ctx *c, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return DAGCON_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 16 + 256;
    const bool dbg = getenv("DAGCON_ALLOC_TIMING") != nullptr;
    const double t0 = dbg ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0;
    hipError_t e = hipMalloc(&b.p, want);
    if (dbg) {
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
        if (dt > 0.005) fprintf(stderr, "dagcon: hipMalloc(%.1f MB) took %.1f ms\n", want / 1e6, dt * 1e3);
    }
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, DAGCON_ERR_WORKSPACE, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return DAGCON_OK;
}

#define ENSURE(c, buf, bytes)                                \
    do {                                                     \
        int _r = ensure((c), (buf), (size_t)(bytes));        \
        if (_r != DAGCON_OK) return _r;                      \
    } while (0)

template <typename T>
int upload_vec(Ctx *c, DevBuf &b, const std::vector<T> &v) {
    ENSURE(c, b, v.size() * sizeof(T));
    if (!v.empty()) HIPCHK(c, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return DAGCON_OK;
}

void free_buf(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
}

int ensure_arenas(Ctx *c) {
    ENSURE(c, c->d_norm, c->norm_cap * sizeof(uint16_t));
    ENSURE(c, c->d_nodes, c->node_cap * sizeof(DgNode));
    ENSURE(c, c->d_best, c->node_cap * 4);
    ENSURE(c, c->d_queue, c->node_cap * 4);
    ENSURE(c, c->d_score, c->node_cap * 8);
    ENSURE(c, c->d_bp_tt, c->node_cap * 4);
    if (c->gcuts && c->bp_fused) ENSURE(c, c->d_score_b, c->node_cap * 4);
    ENSURE(c, c->d_cns_tmp, c->node_cap);
    ENSURE(c, c->d_pool, c->pool_cap * 4);
    ENSURE(c, c->d_stk, std::max<uint64_t>((uint64_t)c->T * std::max(c->bp_max, c->seg_max), (c->tile_pos || c->gcuts) ? c->list_grid : 0) * c->stk_words * 4);
    if (c->tile_pos) ENSURE(c, c->d_nextcut, c->sum_bb * 4);
    if (c->tile_pos || c->gcuts) ENSURE(c, c->d_tile_list, (4ull + 3ull * c->tile_list_cap) * 4);
    if (c->gcuts) {
        ENSURE(c, c->d_rd, (uint64_t)c->A * 16 + 16);
        ENSURE(c, c->d_pro_state, (uint64_t)c->T * 16 + 16);
        ENSURE(c, c->d_sh_cnt, (uint64_t)c->T * (2 + 2 * DG_SH_MAX) * 4 + 16);
        ENSURE(c, c->d_seg_done, (uint64_t)c->tile_list_cap * (DG_SH_MAX + 1) * 4 + 16);
        ENSURE(c, c->d_wl_first, (uint64_t)c->T * 4 + 16);
        ENSURE(c, c->d_queue0, c->node_cap * 4);
        ENSURE(c, c->d_bp_end, (uint64_t)c->T * c->bp_max * 4 + 16);
        ENSURE(c, c->d_bp_ab, (uint64_t)c->T * c->bp_max * 16 + 16);
        ENSURE(c, c->d_defer, (uint64_t)c->T * (DG_DEFER_MAX + 1) * 4 + 16);
        ENSURE(c, c->d_cns_tmp0, c->node_cap);
    }
    ENSURE(c, c->d_cuts, (uint64_t)c->T * (c->seg_max + 2) * 4);
    ENSURE(c, c->d_cuts_bp, (uint64_t)c->T * (c->bp_max + 2) * 4);
    ENSURE(c, c->d_bp_stat, (uint64_t)c->T * c->bp_max * 8);
    ENSURE(c, c->d_bp_len, (uint64_t)c->T * c->bp_max * 4);
    ENSURE(c, c->d_cns, c->cns_cap);
    ENSURE(c, c->d_seg_r0, c->seg_cap * 4);
    ENSURE(c, c->d_seg_r1, c->seg_cap * 4);
    return DAGCON_OK;
}

void fill_params(Ctx *c, DgParams &p) {
    memset(&p, 0, sizeof p);
    p.q = (const uint8_t *)c->d_q.p; p.t = (const uint8_t *)c->d_t.p;
    p.aln_off = (const uint64_t *)c->d_aln_off.p;
    p.aln_len = (const uint32_t *)c->d_aln_len.p;
    p.aln_start = (const uint32_t *)c->d_aln_start.p;
    p.aln_tgt = (const uint32_t *)c->d_aln_tgt.p;
    p.tlen = (const uint32_t *)c->d_tlen.p;
    p.aln_begin = (const uint64_t *)c->d_aln_begin.p;
    p.tactive = (const uint8_t *)c->d_tactive.p; p.tfail = (uint32_t *)c->d_tfail.p;
    p.bb = c->have_bb ? (const uint8_t *)c->d_bb.p : nullptr;
    p.bb_off = (const uint64_t *)c->d_bb_off.p;
    p.mat_base = (const uint64_t *)c->d_mat_base.p;
    p.matc_base = (const uint64_t *)c->d_matc_base.p; p.matc_stride = (const uint32_t *)c->d_matc_stride.p;
    p.bbv_base = (const uint64_t *)c->d_bbv_base.p;
    p.T = c->T; p.A = c->A;
    p.trim = c->opts.trim; p.min_len = c->opts.min_len;
    p.min_weight = c->opts.min_weight < 0 ? (int32_t)c->opts.min_cov : c->opts.min_weight;
    p.flags = c->opts.flags;
    p.max_k = c->max_k; p.max_tlen = c->max_tlen;
    p.nmis = (uint32_t *)c->d_nmis.p; p.norm_off = (uint64_t *)c->d_norm_off.p;
    p.n_lo = (uint32_t *)c->d_n_lo.p; p.n_hi = (uint32_t *)c->d_n_hi.p;
    p.n_start = (uint32_t *)c->d_n_start.p; p.n_ins = (uint32_t *)c->d_n_ins.p;
    p.n_del = (uint32_t *)c->d_n_del.p;
    p.norm = (uint16_t *)c->d_norm.p; p.norm_cap = c->norm_cap;
    p.ch_aln = (const uint32_t *)c->d_ch_aln.p; p.ch_base = (const uint32_t *)c->d_ch_base.p;
    p.n_chunks = c->n_chunks;
    p.ch_k0 = (uint32_t *)c->d_ch_k0.p; p.ch_next = (uint32_t *)c->d_ch_next.p;
    p.ch_w = (uint32_t *)c->d_ch_w.p; p.ch_tb = (uint32_t *)c->d_ch_tb.p;
    p.ch_flag = (uint32_t *)c->d_ch_flag.p; p.ch_src = (uint64_t *)c->d_ch_src.p;
    p.ch_out = (uint32_t *)c->d_ch_out.p; p.ch_adv = (uint32_t *)c->d_ch_adv.p;
    p.n_lb = (uint32_t *)c->d_n_lb.p; p.norm_tmp = (uint16_t *)c->d_norm_tmp.p;
    p.tmp_main = c->tmp_main; p.tmp_cap = c->tmp_cap;
    p.ckpt = (uint32_t *)c->d_ckpt.p; p.ck_base = (const uint32_t *)c->d_ck_base.p; p.emit_shift = c->emit_shift;
    p.node_base = (uint64_t *)c->d_node_base.p; p.n_nodes = (uint32_t *)c->d_n_nodes.p;
    p.pool_base = (uint64_t *)c->d_pool_base.p; p.pool_size = (uint32_t *)c->d_pool_size.p;
    p.pool_top = (uint32_t *)c->d_pool_top.p; p.t_nins = (uint32_t *)c->d_t_nins.p;
    p.matA = (uint32_t *)c->d_matA.p; p.matD = (uint32_t *)c->d_matD.p; p.matC = (uint32_t *)c->d_matC.p;
    p.cov = (int32_t *)c->d_cov.p; p.gcount = (uint32_t *)c->d_gcount.p;
    p.gbase = (uint32_t *)c->d_gbase.p; p.bid = (uint32_t *)c->d_bid.p;
    p.nodes = (DgNode *)c->d_nodes.p; p.best = (int32_t *)c->d_best.p;
    p.queue = (int32_t *)c->d_queue.p; p.score = (float2 *)c->d_score.p; p.bp_tt = (float *)c->d_bp_tt.p;
    p.cns_tmp = (uint8_t *)c->d_cns_tmp.p; p.node_cap = c->node_cap;
    p.pool = (uint32_t *)c->d_pool.p; p.pool_cap = c->pool_cap;
    p.stk = (int32_t *)c->d_stk.p; p.stk_words = c->stk_words; p.growth_pct = c->growth_pct;
    // the prefetch wave pays while the chip has idle wave slots; past ~1.5 workers per SIMD the
    // workers hide each other's latency and it only takes issue slots from them
#ifdef DG_EXPERIMENTS
    { const char *e = getenv("DAGCON_PF_AHEAD"); p.pf_ahead = e ? (uint32_t)atoi(e) : (c->expected_workers >= 4096 ? 0u : 48u); }
#else
    p.pf_ahead = 0;
#endif
    p.emit2 = c->emit2 ? 1u : 0u; p.matK = (uint8_t *)c->d_matK.p; p.bbstart = (uint32_t *)c->d_bbstart.p; p.bs_stride = c->bs_stride;
    p.bp_fused = c->bp_fused ? 1u : 0u; p.score_b = (float *)c->d_score_b.p;
    // (rows pay where there are pieces enough to fill the chip with them, eight to a wave: 64 targets x 50 kb x 60x, 16,384
    // pieces: 2.9 ms by rows against 2.3 by waves; configs[1], 147,000 pieces: 3.8 against 4.7.  DAGCON_BP_LANE=2: always)
    p.bp_lane = c->bp_lane >= 2 || (c->bp_lane && (uint64_t)c->T * c->bp_max >= 32768ull) ? 1u : 0u; p.bl_stk = c->bl_stk >= 0 && c->bl_stk < DG_BL_STK ? (uint32_t)c->bl_stk : (uint32_t)DG_BL_STK;
    p.emit_scan = (c->emit_scan && c->max_k <= 64u && !c->emit2) ? 1u : 0u;
    p.fold = (c->fold && !(c->opts.flags & DAGCON_FLAG_STOP_AFTER_BUILD)) ? 1u : 0u;
    p.q_kmax = c->use_q && !c->opts.max_segments && !c->seg_env && c->max_k > DQ_KMAX ? DQ_KMAX : 0u;
    p.bp_seg_min = (c->seg_min + 2u) / 3u;
    p.seg_max = c->seg_max; p.seg_min = c->seg_min; p.cuts = (uint32_t *)c->d_cuts.p; p.bp_max = c->bp_max; p.cuts_bp = (uint32_t *)c->d_cuts_bp.p; p.bp_stat = (float *)c->d_bp_stat.p; p.bp_len = (uint32_t *)c->d_bp_len.p;
    p.gcuts = c->gcuts; p.sh_log = c->sh_log;
    p.rd_s = (uint32_t *)c->d_rd.p; p.rd_e = p.rd_s + c->A; p.rd_lead = p.rd_e + c->A; p.rd_trail = p.rd_lead + c->A;
    p.pro_state = (uint32_t *)c->d_pro_state.p; p.sh_cnt = (uint32_t *)c->d_sh_cnt.p;
    p.queue0 = (int32_t *)c->d_queue0.p; p.bp_end = (uint32_t *)c->d_bp_end.p; p.bp_ab = (float *)c->d_bp_ab.p;
    p.defer = (uint32_t *)c->d_defer.p; p.cns_tmp0 = (uint8_t *)c->d_cns_tmp0.p;
    p.seg_done = (uint32_t *)c->d_seg_done.p; p.wl_first = (uint32_t *)c->d_wl_first.p;
    p.nextcut = (uint32_t *)c->d_nextcut.p; p.tile_pos = c->tile_pos; p.tile_words = c->tile_words;
    p.tile_list = (uint32_t *)c->d_tile_list.p; p.tile_list_cap = c->tile_list_cap;
    p.cns = (uint8_t *)c->d_cns.p; p.cns_cap = c->cns_cap;
    p.cns_off = (uint64_t *)c->d_cns_off.p; p.cns_len = (uint32_t *)c->d_cns_len.p;
    p.seg_first = (uint64_t *)c->d_seg_first.p; p.n_seg = (uint32_t *)c->d_n_seg.p;
    p.seg_r0 = (int32_t *)c->d_seg_r0.p; p.seg_r1 = (int32_t *)c->d_seg_r1.p;
    p.seg_cap = c->seg_cap;
    p.st = (DgStatus *)c->d_st.p;
}

// stage a1: count, chunked normalizeGaps + trimAln, and the sequential kernel for what is left
void launch_normalize(Ctx *c, const DgParams &p) {
    hipStream_t s = c->stream;
    if (c->A == 0) return;
    (void)hipMemsetAsync(c->d_ckpt.p, 0xFF, c->n_ckpt * 4, s);
    hipLaunchKernelGGL((k_norm_chunk<DG_NW, 64, false>), dim3((c->n_chunks + 63) / 64), dim3(64), 0, s, p);
    hipLaunchKernelGGL((k_norm_chunk<DG_NW_BIG, 32, true>), dim3((c->n_chunks + 31) / 32), dim3(32), 0, s, p);
    hipLaunchKernelGGL(k_norm_scan, dim3((c->A + 63) / 64), dim3(64), 0, s, p);
    if (c->nf2) hipLaunchKernelGGL(k_norm_finish2, dim3((c->n_chunks + 3) / 4), dim3(256), 0, s, p);    // a wave per chunk
    else hipLaunchKernelGGL(k_norm_finish, dim3((c->n_chunks + 63) / 64), dim3(64), 0, s, p);          // a lane per chunk (DAGCON_NF2=0)
    hipLaunchKernelGGL(k_normalize_slow, dim3((c->A + 63) / 64), dim3(64), 0, s, p);
}

int launch_all(Ctx *c) {
    int r = ensure_arenas(c);
    if (r != DAGCON_OK) return r;
    DgParams p;
    fill_params(c, p);
    hipStream_t s = c->stream;
    if (c->poison & 8) {
        // every buffer the kernels themselves fill (nothing the host uploaded), before the memsets below: whoever reads an
        // entry of them that THIS run has not written finds 0xEE bytes, in a fresh process as in one that re-uses its memory
        DevBuf *work[] = {&c->d_nmis, &c->d_n_lo, &c->d_n_hi, &c->d_n_start, &c->d_n_ins, &c->d_n_del, &c->d_ch_k0, &c->d_ch_next, &c->d_ch_w,
                          &c->d_ch_tb, &c->d_ch_flag, &c->d_ch_src, &c->d_ch_out, &c->d_ch_adv, &c->d_n_lb, &c->d_norm_tmp, &c->d_ckpt,
                          &c->d_node_base, &c->d_n_nodes, &c->d_pool_base, &c->d_pool_size, &c->d_pool_top, &c->d_t_nins, &c->d_cov, &c->d_gcount,
                          &c->d_gbase, &c->d_bid, &c->d_best, &c->d_queue, &c->d_score, &c->d_cns_tmp, &c->d_bp_tt, &c->d_stk, &c->d_cuts,
                          &c->d_cuts_bp, &c->d_bp_stat, &c->d_bp_len, &c->d_nextcut, &c->d_rd, &c->d_pro_state, &c->d_sh_cnt, &c->d_wl_first,
                          &c->d_queue0, &c->d_bp_end, &c->d_bp_ab, &c->d_defer, &c->d_cns_tmp0, &c->d_cns, &c->d_cns_off, &c->d_seg_first,
                          &c->d_seg_r0, &c->d_seg_r1, &c->d_tile_list, &c->d_seg_done};
        for (DevBuf *b : work)
            if (b->p && b->cap) HIPCHK(c, hipMemsetAsync(b->p, 0xEE, b->cap, s));
    }
    HIPCHK(c, hipMemsetAsync(c->d_st.p, 0, sizeof(DgStatus), s));
    HIPCHK(c, hipMemsetAsync(c->d_tfail.p, 0, (size_t)c->T * 4 + 4, s));
    HIPCHK(c, hipMemsetAsync(c->d_cns_len.p, 0, (size_t)c->T * 4, s));
    HIPCHK(c, hipMemsetAsync(c->d_n_seg.p, 0, (size_t)c->T * 4, s));
    if (c->matc_cells) HIPCHK(c, hipMemsetAsync(c->d_matC.p, 0, c->matc_cells * 4, s));
    if (c->poison) {
        // what no kernel is supposed to read before it has been written in THIS run: a process that re-uses its
        // arenas (another context's freed memory, the batch before) finds old cells there, not the zeros of a fresh one
        if ((c->poison & 1) && c->d_matA.p) { HIPCHK(c, hipMemsetAsync(c->d_matA.p, 0xEE, c->d_matA.cap, s)); HIPCHK(c, hipMemsetAsync(c->d_matD.p, 0xEE, c->d_matD.cap, s)); }
        if ((c->poison & 2) && c->d_nodes.p) { HIPCHK(c, hipMemsetAsync(c->d_nodes.p, 0xEE, c->d_nodes.cap, s)); HIPCHK(c, hipMemsetAsync(c->d_pool.p, 0xEE, c->d_pool.cap, s)); }
        if ((c->poison & 4) && c->d_score_b.p) HIPCHK(c, hipMemsetAsync(c->d_score_b.p, 0xEE, c->d_score_b.cap, s));
        if ((c->poison & 4) && c->d_norm.p) HIPCHK(c, hipMemsetAsync(c->d_norm.p, 0xEE, c->d_norm.cap, s));
    }
    HIPCHK(c, hipEventRecord(c->ev[0], s));
    launch_normalize(c, p);
    HIPCHK(c, hipEventRecord(c->ev[1], s));
    // (matA / matD are not cleared: k_emit writes every cell of every row)
    hipLaunchKernelGGL(k_carve, dim3(1), dim3(1024), 0, s, p);
    if (c->T > 0) {
        const uint32_t rows4 = (c->max_tlen + 2 + 4 * DG_LPW - 1) / (4 * DG_LPW);   // 4 waves x DG_LPW positions per block
        if (c->gcuts && c->A > 0) hipLaunchKernelGGL(k_readspan, dim3((c->A + 63) / 64), dim3(64), 0, s, p);   // (before matC becomes prefix sums)
        if (p.emit_scan) hipLaunchKernelGGL(k_gsum, dim3(c->T, (c->max_tlen + 2 + 255) / 256), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(k_groups, dim3(c->T, (c->max_tlen + 2 + 31) / 32), dim3(256), 0, s, p);
        hipLaunchKernelGGL(k_gscan, dim3(c->T), dim3(1024), 0, s, p);
#ifdef DG_EXPERIMENTS
        if (c->A > 0 && c->emit2) {
            hipLaunchKernelGGL(k_blockscan, dim3(c->A), dim3(64), 0, s, p);
            hipLaunchKernelGGL(k_emit2, dim3((c->bs_stride + 3u) / 4u, c->A), dim3(256), 0, s, p);
            if (p.fold) hipLaunchKernelGGL(k_dedupe, dim3(c->T, rows4), dim3(256), 0, s, p);
        } else
#endif
        if (c->A > 0)
            hipLaunchKernelGGL(k_emit, dim3(c->T, (c->max_k + DG_ERPW - 1) / DG_ERPW, ((c->max_tlen + 2) >> c->emit_shift) + 1),
                               dim3(64), 0, s, p);
        const size_t lds = (size_t)4 * 2 * (c->max_k + 2) * sizeof(int32_t);
        if (lds > 65536)
            HIPCHK(c, hipFuncSetAttribute((const void *)k_lists, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_lists, dim3(c->T, rows4), dim3(256), lds, s, p);
    }
    HIPCHK(c, hipEventRecord(c->ev[2], s));
    if (c->T > 0 && !(c->opts.flags & DAGCON_FLAG_STOP_AFTER_BUILD)) {
        if (!c->gcuts || c->tile_pos) hipLaunchKernelGGL(k_cuts, dim3(c->T), dim3(64), 0, s, p);      // (k_cuts2 makes its own, bestPath's too)
        if (c->gcuts && !c->tile_pos) {
            // partial-span cuts: enter and what hangs on it first, then the segments as a worklist, exit last
            HIPCHK(c, hipMemsetAsync(c->d_tile_list.p, 0, 16, s));
            HIPCHK(c, hipMemsetAsync(c->d_seg_done.p, 0, (size_t)c->tile_list_cap * (DG_SH_MAX + 1) * 4, s));
            hipLaunchKernelGGL(k_merge_pro, dim3(c->T), dim3(64), 0, s, p);
            hipLaunchKernelGGL(k_cuts2, dim3(c->T), dim3(64), 0, s, p);
            // the worklist, a wave per entry (make experiments: eight entries per wave, one per row of eight lanes, k_merge_list_q)
#ifdef DG_EXPERIMENTS
            if (c->merge_q && c->list_q && c->max_k <= DQ_KMAX && (uint64_t)c->T * c->seg_max >= DQ_ROWS) {
                const uint64_t grid = std::min<uint64_t>(1024ull * DQ_WAVES, (uint64_t)c->T * c->seg_max / DQ_ROWS);
                hipLaunchKernelGGL(k_merge_list_q, dim3((uint32_t)grid), dim3(64), 0, s, p);
            } else
#endif
            hipLaunchKernelGGL(k_merge_list, dim3(c->list_grid), dim3(64), 0, s, p);
            hipLaunchKernelGGL(k_merge_fin, dim3(c->T), dim3(64), 0, s, p);
        }
#ifdef DG_EXPERIMENTS
        else if (c->tile_pos) {
            HIPCHK(c, hipMemsetAsync(c->d_tile_list.p, 0, 16, s));
            HIPCHK(c, hipFuncSetAttribute((const void *)k_merge_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(c->tile_words * 4)));
            hipLaunchKernelGGL(k_cutmap, dim3(c->T), dim3(1024), 0, s, p);
            hipLaunchKernelGGL(k_merge_tile, dim3(c->T, c->tile_ny), dim3(DG_T_LANES), c->tile_words * 4, s, p);
            hipLaunchKernelGGL(k_merge_list, dim3(c->list_grid), dim3(64), 0, s, p);
        } else if (p.pf_ahead) hipLaunchKernelGGL(k_merge<true>, dim3(c->T * c->seg_max), dim3(128), 0, s, p);
#endif
        else if (c->use_q) {
            hipLaunchKernelGGL(k_merge_q, dim3((c->T * c->seg_max + DQ_ROWS - 1u) / DQ_ROWS), dim3(64), 0, s, p);
            // (the few deep targets of a shallow batch: the same cuts, a wave per segment)
            if (p.q_kmax) hipLaunchKernelGGL(k_merge<false>, dim3(c->T * c->seg_max), dim3(64), 0, s, p);
        }
        else hipLaunchKernelGGL(k_merge<false>, dim3(c->T * c->seg_max), dim3(64), 0, s, p);
    }
    HIPCHK(c, hipEventRecord(c->ev[3], s));
    if (c->T > 0 && !(c->opts.flags & (DAGCON_FLAG_STOP_AFTER_BUILD | DAGCON_FLAG_STOP_AFTER_MERGE))) {
        hipLaunchKernelGGL(k_bp_terms, dim3(c->T, 16), dim3(256), 0, s, p);
        if (c->gcuts && !c->tile_pos) {
            // partial-span pileups: the pieces of k_cuts2, three sweeps (A, B, absolute), see dg_bp_sweep
            hipLaunchKernelGGL(k_bp_xtree, dim3(c->T), dim3(64), 0, s, p);
            if (p.bp_fused) {
                // one sweep for (A, B), then vertex-parallel kernels for the absolute scores and the choices
                hipLaunchKernelGGL(k_bp_sweep_ab, dim3(c->T * c->bp_max), dim3(64), 0, s, p);
                hipLaunchKernelGGL(k_bp_comb, dim3(c->T), dim3(64), 0, s, p);
                hipLaunchKernelGGL(k_bp_abs, dim3(c->T * c->bp_max), dim3(256), 0, s, p);
                hipLaunchKernelGGL(k_bp_choose, dim3(c->T * c->bp_max), dim3(256), 0, s, p);
            } else {
                hipLaunchKernelGGL(k_bp_sweep_g<0>, dim3(c->T * c->bp_max), dim3(64), 0, s, p);
                hipLaunchKernelGGL(k_bp_reset_def, dim3(c->T), dim3(64), 0, s, p);
                hipLaunchKernelGGL(k_bp_sweep_g<1>, dim3(c->T * c->bp_max), dim3(64), 0, s, p);
                hipLaunchKernelGGL(k_bp_comb, dim3(c->T), dim3(64), 0, s, p);
            }
            hipLaunchKernelGGL(k_bp_sweep_g<2>, dim3(c->T * c->bp_max), dim3(64), 0, s, p);     // (fused: whole-target sweeps only)
            hipLaunchKernelGGL(k_bp_defer, dim3(c->T), dim3(64), 0, s, p);
            hipLaunchKernelGGL(k_bp_walk_g, dim3(c->T * c->bp_max), dim3(64), 0, s, p);
        } else {
            // a lane per piece first; the wave-per-piece sweep then takes the pieces a lane gave up (deep recursion)
            if (p.bp_lane) hipLaunchKernelGGL(k_bp_sweep_l, dim3((c->T * c->bp_max + 7u) / 8u), dim3(64), 0, s, p);
            hipLaunchKernelGGL(k_bp_sweep, dim3(c->T * c->bp_max), dim3(64), 0, s, p);
            hipLaunchKernelGGL(k_bp_check, dim3(c->T), dim3(64), 0, s, p);
            if (p.bp_lane) hipLaunchKernelGGL(k_bp_walk_r, dim3((c->T * c->bp_max + 7u) / 8u), dim3(64), 0, s, p);
            else hipLaunchKernelGGL(k_bp_walk, dim3(c->T * c->bp_max), dim3(64), 0, s, p);
        }
        hipLaunchKernelGGL(k_bp_join, dim3(c->T), dim3(64), 0, s, p);
    }
    HIPCHK(c, hipEventRecord(c->ev[4], s));
    HIPCHK(c, hipGetLastError());
    return DAGCON_OK;
}

}  // namespace

extern "C" {

int dagcon_abi_version(void) { return DAGCON_ABI_VERSION; }

void dagcon_default_opts(dagcon_opts *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->min_cov = 6; o->min_len = 500; o->trim = 50; o->min_weight = -1; o->device = 0; o->flags = 0;
}

const char *dagcon_last_error(const dagcon_ctx *ctx) {
    return ctx ? reinterpret_cast<const Ctx *>(ctx)->err.c_str() : "null context";
}

int dagcon_create(const dagcon_opts *opts, dagcon_ctx **out) {
    if (!opts || !out) return DAGCON_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DAGCON_ERR_NO_DEVICE;
    if (opts->device < 0 || opts->device >= ndev) return DAGCON_ERR_NO_DEVICE;
    if (opts->flags & ~DAGCON_FLAGS_ALL) return DAGCON_ERR_UNSUPPORTED;   // (internal bits start at 8: never from outside)
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, opts->device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return DAGCON_ERR_NO_DEVICE;                      // the code object is gfx950 only
    }
    Ctx *c = new Ctx();
    c->opts = *opts;
    c->device = opts->device;
    if (const char *e = getenv("DAGCON_EMIT_SHIFT")) {      // test knob: k_emit stretches of 1 << v positions
        const int v = atoi(e);
        if (v >= 4 && v <= 20) c->emit_shift = (uint32_t)v;
    }
    if (const char *e = getenv("DAGCON_MERGE_SEGS")) {      // tuning knob: 1 = one worker per target
        const int v = atoi(e);
        if (v >= 1 && v <= 64) c->seg_env = (uint32_t)v;
    }
    if (const char *e = getenv("DAGCON_FOLD")) c->fold = atoi(e) != 0;
    if (const char *e = getenv("DAGCON_ALIGN2")) c->align2 = atoi(e) != 0;
    if (const char *e = getenv("DAGCON_POISON")) c->poison = atoi(e);
    if (const char *e = getenv("DAGCON_NF2")) c->nf2 = atoi(e) != 0;
    if (const char *e = getenv("DAGCON_EMIT_SCAN")) c->emit_scan = atoi(e) != 0;
    if (const char *e = getenv("DAGCON_BP_FUSED")) c->bp_fused = atoi(e) != 0;
    if (const char *e = getenv("DAGCON_BP_LANE")) c->bp_lane = atoi(e);
    if (const char *e = getenv("DAGCON_BP_LANE_STACK")) c->bl_stk = atoi(e);
#ifdef DG_EXPERIMENTS
    if (const char *e = getenv("DAGCON_EMIT2")) c->emit2 = atoi(e) != 0;
#endif
    if (const char *e = getenv("DAGCON_MERGE_LIST_Q")) c->list_q = atoi(e) != 0;
    if (const char *e = getenv("DAGCON_MERGE_Q")) c->merge_q = atoi(e) != 0;     // eight segments per wave (k_merge_q.hip.h)
    memset(&c->tm, 0, sizeof c->tm);
    memset(&c->h_st, 0, sizeof c->h_st);
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return DAGCON_ERR_NO_DEVICE;
    }
    for (auto &e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete c; return DAGCON_ERR_HIP; }
    if (ensure(c, c->d_st, sizeof(DgStatus)) != DAGCON_OK) { delete c; return DAGCON_ERR_WORKSPACE; }
    *out = reinterpret_cast<dagcon_ctx *>(c);
    return DAGCON_OK;
}

void dagcon_destroy(dagcon_ctx *ctx) {
    if (!ctx) return;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->r_blob) (void)hipHostFree(c->r_blob);
    DevBuf *all[] = {&c->d_q, &c->d_t, &c->d_aln_off, &c->d_aln_len, &c->d_aln_start, &c->d_aln_tgt,
                     &c->d_tlen, &c->d_aln_begin, &c->d_tactive, &c->d_tfail, &c->d_bb, &c->d_bb_off, &c->d_mat_base, &c->d_matc_base, &c->d_matc_stride,
                     &c->d_bbv_base, &c->d_nmis, &c->d_norm_off, &c->d_n_lo, &c->d_n_hi, &c->d_n_start, &c->d_ch_aln, &c->d_ch_base, &c->d_ch_k0, &c->d_ch_next, &c->d_ch_w, &c->d_ch_tb, &c->d_ch_flag, &c->d_ch_src, &c->d_ch_out, &c->d_ch_adv, &c->d_n_lb, &c->d_norm_tmp, &c->d_ckpt, &c->d_ck_base,
                     &c->d_n_ins, &c->d_n_del, &c->d_norm, &c->d_node_base, &c->d_n_nodes,
                     &c->d_pool_base, &c->d_pool_size, &c->d_pool_top, &c->d_t_nins, &c->d_matA, &c->d_matD,
                     &c->d_matC, &c->d_cov, &c->d_gcount, &c->d_gbase, &c->d_bid, &c->d_nodes,
                     &c->d_best, &c->d_queue, &c->d_score, &c->d_cns_tmp, &c->d_bp_tt, &c->d_score_b, &c->d_matK, &c->d_bbstart, &c->d_pool, &c->d_stk, &c->d_cuts, &c->d_cuts_bp, &c->d_bp_stat, &c->d_bp_len, &c->d_nextcut, &c->d_tile_list, &c->d_rd, &c->d_pro_state, &c->d_sh_cnt, &c->d_seg_done, &c->d_wl_first, &c->d_queue0, &c->d_bp_end, &c->d_bp_ab, &c->d_defer, &c->d_cns_tmp0, &c->d_cns,
                     &c->d_cns_off, &c->d_cns_len, &c->d_seg_first, &c->d_n_seg, &c->d_seg_r0, &c->d_seg_r1,
                     &c->d_st};
    for (DevBuf *b : all) free_buf(*b);
    for (DevBuf &b : c->d_al) free_buf(b);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// dev_q / dev_t: the blobs are on the device already (dagcon_consensus_pre: the aligner's output), b->qstr / tstr unused
static int upload_impl(dagcon_ctx *ctx, const dagcon_batch *b, const void *dev_q, const void *dev_t) {
    if (!ctx || !b) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    c->uploaded = c->ran = c->fetched = false;
    const uint32_t T = b->n_targets;
    if (T && (!b->tlen || !b->aln_begin)) return fail(c, DAGCON_ERR_INVALID_ARG, "tlen/aln_begin is NULL");
    const uint64_t A_all = T ? b->aln_begin[T] : 0;
    if (A_all && (!b->aln_start || !b->aln_off || !b->aln_len || ((!b->qstr || !b->tstr) && !(dev_q && dev_t))))
        return fail(c, DAGCON_ERR_INVALID_ARG, "alignment arrays are NULL");
    if (b->backbone && !b->backbone_off) return fail(c, DAGCON_ERR_INVALID_ARG, "backbone_off is NULL");
    HIPCHK(c, hipSetDevice(c->device));

    c->T = T;
    c->h_tlen.assign(b->tlen, b->tlen + T);
    c->h_aln_begin.assign(T + 1, 0);
    c->h_tactive.assign(T, 0);
    c->h_mat_base.assign(T, 0);
    c->h_matc_base.assign(T, 0); c->h_matc_stride.assign(T, 0); c->matc_cells = 0;
    c->h_bbv_base.assign(T, 0);
    c->h_bb_off.assign(T, 0);
    c->h_aln_len.clear(); c->h_aln_start.clear(); c->h_aln_tgt.clear(); c->h_aln_off.clear();
    c->max_len = 0;
    c->max_k = 0; c->max_tlen = 0; c->sum_len = 0; c->sum_bb = 0; c->mat_cells = 0;
    c->have_bb = b->backbone != nullptr;
    uint64_t bb_bytes = 0, n_whole = 0;
    const uint64_t min_cov = c->opts.min_cov;
    for (uint32_t t = 0; t < T; t++) {
        const uint64_t ab = b->aln_begin[t], ae = b->aln_begin[t + 1];
        if (ae < ab) return fail(c, DAGCON_ERR_INVALID_ARG, "aln_begin not monotone at target %u", t);
        const uint64_t k_all = ae - ab;
        // main.cpp:66-72 (Reader) and :118 (Consensus): groups below min_cov are dropped
        const bool active = k_all > 0 && k_all >= min_cov;
        c->h_aln_begin[t] = c->h_aln_len.size();
        if (!active) continue;
        if (b->tlen[t] > 0x3FFFFFFFu) return fail(c, DAGCON_ERR_UNSUPPORTED, "tlen of target %u too large", t);
        c->h_tactive[t] = 1;
        for (uint64_t a = ab; a < ae; a++) {
            const uint32_t len = b->aln_len[a];
            if (b->aln_off[a] > b->blob_bytes || len > b->blob_bytes - b->aln_off[a])
                return fail(c, DAGCON_ERR_INVALID_ARG, "alignment %llu runs past the blob", (unsigned long long)a);
            if (len < c->opts.min_len) continue;       // main.cpp:132
            c->h_aln_len.push_back(len);
            c->h_aln_start.push_back(b->aln_start[a]);
            c->h_aln_off.push_back(b->aln_off[a]);
            c->h_aln_tgt.push_back(t);
            c->sum_len += len;
            c->max_len = std::max(c->max_len, len);
            // (a read that spans the target begins at its first base and has a column per target base; necessary, not
            // sufficient -- a read that ends early and inserts a lot passes too: the batch is then exact all the same, with
            // fewer cuts than it could have)
            n_whole += len >= b->tlen[t] && b->aln_start[a] == 1u;
        }
        const uint64_t k = c->h_aln_len.size() - c->h_aln_begin[t];
        if (k > DAGCON_MAX_COVERAGE)
            return fail(c, DAGCON_ERR_UNSUPPORTED, "target %u has %llu alignments (max %u)", t,
                        (unsigned long long)k, DAGCON_MAX_COVERAGE);
        c->max_k = std::max<uint32_t>(c->max_k, (uint32_t)k);
        c->max_tlen = std::max(c->max_tlen, b->tlen[t]);
        if ((uint64_t)b->tlen[t] + 2 > 4ull * 65535ull)
            return fail(c, DAGCON_ERR_UNSUPPORTED, "tlen of target %u exceeds %u", t, 4u * 65535u - 2u);
        c->h_mat_base[t] = c->mat_cells;
        c->mat_cells += ((uint64_t)b->tlen[t] + 2) * k;
        c->h_matc_stride[t] = (b->tlen[t] + 2 + 7) & ~7u;      // matC is [read][position], rows 32-byte aligned
        c->h_matc_base[t] = c->matc_cells;
        c->matc_cells += (uint64_t)c->h_matc_stride[t] * k;
        c->h_bbv_base[t] = c->sum_bb;                      // multiple of 4: 16-byte loads of bid[]
        c->sum_bb += ((uint64_t)b->tlen[t] + 2 + 3) & ~3ull;
        if (c->have_bb) {
            c->h_bb_off[t] = b->backbone_off[t];
            bb_bytes = std::max<uint64_t>(bb_bytes, b->backbone_off[t] + b->tlen[t]);
        }
    }
    // cuts for partial-span pileups (prologue + worklist + epilogue): where the reads are full-span the cut
    // vertices every read passes through are the same ones, found without that machinery
    c->full_span = n_whole == (uint64_t)c->h_aln_len.size();
    // shortest stretch worth a worker: 768 positions when that already fills the chip, shorter (down to 192)
    // for small batches, whose waves would otherwise be few and long
    c->gcuts = c->full_span ? 0u : 1u;
    if (const char *e = getenv("DAGCON_GCUTS")) c->gcuts = atoi(e) ? 1u : 0u;
    {
        DgPlanIn pi;
        pi.T = T; pi.n_alns = c->h_aln_len.size(); pi.sum_bb = c->sum_bb; pi.gcuts = c->gcuts;
        pi.max_segments = c->opts.max_segments; pi.min_segment_len = c->opts.min_segment_len;
        pi.seg_env = c->seg_env; pi.merge_q = c->merge_q ? 1u : 0u;
        const DgPlan pl = dg_plan_pieces(pi);
        c->seg_max = pl.seg_max; c->seg_min = pl.seg_min; c->use_q = (int)pl.use_q; c->bp_max = pl.bp_max;
    }
    if (const char *e = getenv("DAGCON_BP_SEGS")) { const int v = atoi(e); if (v >= 1 && v <= 64) c->bp_max = (uint32_t)v; }
    // scratch per (target, piece): 4096 words where that is cheap, less for batches of very many
    // targets (2 GB in all at most; a piece that needs more raises DG_E_STACK: grown x4, re-run)
    {
        const uint64_t pieces = std::max<uint64_t>(1, (uint64_t)T * std::max(c->bp_max, c->seg_max));
        const uint32_t fit = (uint32_t)std::min<uint64_t>(4096, (512ull << 20) / pieces);
        const uint32_t base = std::max(256u, fit);
        if (c->stk_words < base || (uint64_t)c->stk_words * pieces > (1024ull << 20)) c->stk_words = base;
    }
    if (c->gcuts) c->tile_list_cap = std::max<uint32_t>(c->tile_list_cap, (uint32_t)std::min<uint64_t>((uint64_t)T * c->seg_max + 64, 0x0FFFFFFFull));
#ifdef DG_EXPERIMENTS
    // LDS tiles for mergeNodes: positions per tile from the LDS budget and the expected size of a
    // position's share of the graph (exact after the first run of a shape)
    {
        // (opt-in: exact, but at configs[1] the lanes' dependent LDS chains at two waves per CU take 128 ms
        // where the wave-per-segment kernel takes 22: DESIGN.md, "tried and dropped")
        int on = 0;
        if (const char *e = getenv("DAGCON_TILES")) on = atoi(e);
        uint32_t kb = 80;
        if (const char *e = getenv("DAGCON_TILE_KB")) { const int v = atoi(e); if (v >= 16 && v <= 160) kb = (uint32_t)v; }
        c->tile_pos = 0;
        if (on && T && c->max_k >= 1 && c->seg_max != 1) {
            c->tile_words = kb * 256u;
            const double ipp = c->ins_per_pos >= 0 ? c->ins_per_pos : (double)c->sum_len / 10.0 / (double)std::max<uint64_t>(c->sum_bb, 1);
            const double wpp = 9.5 * (1.0 + ipp) + 3.0 * ipp + 3.0 * dg_capb(c->max_k);
            const double room = ((double)c->tile_words - 1200.0) * 0.9;
            uint32_t G = (uint32_t)std::max(8.0, room / wpp);
            if (const char *e = getenv("DAGCON_TILE_POS")) { const int v = atoi(e); if (v >= 2) G = (uint32_t)v; }
            c->tile_pos = G;
            c->tile_ny = (c->max_tlen + 2 + G - 1) / G;
            c->tile_list_cap = std::max<uint32_t>(c->tile_list_cap, (uint32_t)std::min<uint64_t>((uint64_t)T * c->tile_ny + 16, 0x0FFFFFFFull));
        }
    }
#endif
    c->h_aln_begin[T] = c->h_aln_len.size();
    if (c->h_aln_len.size() > 0xFFFFFFF0ull) return fail(c, DAGCON_ERR_UNSUPPORTED, "too many alignments");
    c->A = (uint32_t)c->h_aln_len.size();
    c->blob_bytes = b->blob_bytes;
    // cut vertices need every read to span them: with full-span reads a target is swept in seg_max
    // pieces, with partial spans in a few
    c->expected_workers = c->full_span ? (uint64_t)T * c->seg_max : (uint64_t)T * 3;
    // windows of DG_NCH input columns: the units of the chunked normalizeGaps
    c->h_ch_base.assign((size_t)c->A + 1, 0);
    c->h_ch_aln.clear();
    for (uint32_t a = 0; a < c->A; a++) {
        const uint32_t nw = std::max<uint32_t>(1u, (c->h_aln_len[a] + DG_NCH - 1) / DG_NCH);
        c->h_ch_base[a] = (uint32_t)c->h_ch_aln.size();
        if (c->h_ch_aln.size() + nw > 0xFFFFFFF0ull) return fail(c, DAGCON_ERR_UNSUPPORTED, "too many alignment columns");
        c->h_ch_aln.insert(c->h_ch_aln.end(), nw, a);
    }
    // column buffers: an alignment normalises to at most 2 columns per input column (every mismatch
    // becomes two); offsets are multiples of 8 columns (16-byte pieces)
    c->h_norm_off.assign((size_t)c->A, 0);
    {
        uint64_t top = 0;
        for (uint32_t a = 0; a < c->A; a++) { c->h_norm_off[a] = top; top += (2ull * c->h_aln_len[a] + 7ull) & ~7ull; }
        c->norm_cap = std::max<uint64_t>(c->norm_cap, top + 64);
    }
    c->h_ck_base.assign((size_t)c->A, 0);
    c->n_ckpt = 0;
    for (uint32_t a = 0; a < c->A; a++) {
        c->h_ck_base[a] = (uint32_t)c->n_ckpt;
        c->n_ckpt += (((uint64_t)c->h_tlen[c->h_aln_tgt[a]] + 2) >> c->emit_shift) + 1;
        if (c->n_ckpt > 0xFFFFFFF0ull) return fail(c, DAGCON_ERR_UNSUPPORTED, "too many alignment columns");
    }
    c->h_ch_base[c->A] = (uint32_t)c->h_ch_aln.size();
    c->n_chunks = (uint32_t)c->h_ch_aln.size();
    c->tmp_main = (2ull * b->blob_bytes + 8ull * c->n_chunks + 15ull) & ~7ull;
    c->tmp_cap = c->tmp_main + std::max<uint64_t>(c->tmp_main / 16, 1ull << 20);

    // inputs -> HBM
    ENSURE(c, c->d_q, b->blob_bytes);
    ENSURE(c, c->d_t, b->blob_bytes);
    if (b->blob_bytes) {
        HIPCHK(c, hipMemcpyAsync(c->d_q.p, dev_q ? dev_q : b->qstr, b->blob_bytes, dev_q ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_t.p, dev_t ? dev_t : b->tstr, b->blob_bytes, dev_t ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    }
    if (c->have_bb) {
        ENSURE(c, c->d_bb, bb_bytes);
        if (bb_bytes) HIPCHK(c, hipMemcpyAsync(c->d_bb.p, b->backbone, bb_bytes, hipMemcpyHostToDevice, c->stream));
    }
    int r;
    if ((r = upload_vec(c, c->d_aln_off, c->h_aln_off))) return r;
    if ((r = upload_vec(c, c->d_aln_len, c->h_aln_len))) return r;
    if ((r = upload_vec(c, c->d_aln_start, c->h_aln_start))) return r;
    if ((r = upload_vec(c, c->d_aln_tgt, c->h_aln_tgt))) return r;
    if ((r = upload_vec(c, c->d_tlen, c->h_tlen))) return r;
    if ((r = upload_vec(c, c->d_aln_begin, c->h_aln_begin))) return r;
    if ((r = upload_vec(c, c->d_tactive, c->h_tactive))) return r;
    if ((r = upload_vec(c, c->d_bb_off, c->h_bb_off))) return r;
    if ((r = upload_vec(c, c->d_mat_base, c->h_mat_base))) return r;
    if ((r = upload_vec(c, c->d_matc_base, c->h_matc_base))) return r;
    if ((r = upload_vec(c, c->d_matc_stride, c->h_matc_stride))) return r;
    if ((r = upload_vec(c, c->d_bbv_base, c->h_bbv_base))) return r;
    if ((r = upload_vec(c, c->d_ch_base, c->h_ch_base))) return r;
    if ((r = upload_vec(c, c->d_ch_aln, c->h_ch_aln))) return r;
    if ((r = upload_vec(c, c->d_ck_base, c->h_ck_base))) return r;
    if ((r = upload_vec(c, c->d_norm_off, c->h_norm_off))) return r;
    ENSURE(c, c->d_ckpt, c->n_ckpt * 4);

    // work arrays whose size the host knows
    const size_t A4 = (size_t)c->A * 4, T4 = (size_t)T * 4;
    ENSURE(c, c->d_nmis, A4);
    ENSURE(c, c->d_n_lo, A4); ENSURE(c, c->d_n_hi, A4); ENSURE(c, c->d_n_start, A4);
    ENSURE(c, c->d_n_ins, A4); ENSURE(c, c->d_n_del, A4); ENSURE(c, c->d_n_lb, A4);
    {
        const size_t C4 = (size_t)c->n_chunks * 4;
        ENSURE(c, c->d_ch_k0, C4); ENSURE(c, c->d_ch_next, C4); ENSURE(c, c->d_ch_w, C4); ENSURE(c, c->d_ch_tb, C4);
        ENSURE(c, c->d_ch_flag, C4); ENSURE(c, c->d_ch_src, 2 * C4); ENSURE(c, c->d_ch_out, C4); ENSURE(c, c->d_ch_adv, C4);
        ENSURE(c, c->d_norm_tmp, c->tmp_cap * sizeof(uint16_t));
    }
    ENSURE(c, c->d_node_base, (size_t)T * 8); ENSURE(c, c->d_n_nodes, T4);
    ENSURE(c, c->d_pool_base, (size_t)T * 8); ENSURE(c, c->d_pool_size, T4); ENSURE(c, c->d_pool_top, T4);
    ENSURE(c, c->d_t_nins, T4); ENSURE(c, c->d_tfail, T4 + 4);
    if (c->emit2) {
        // cells as [read][position] rows (k_emit2.hip.h); the chains' keys; backbone position per 64-column block
        ENSURE(c, c->d_matA, c->matc_cells * 4 + 256); ENSURE(c, c->d_matD, c->matc_cells * 4 + 256);
        ENSURE(c, c->d_matK, c->matc_cells + 256);
        c->bs_stride = (uint32_t)((2ull * c->max_len + 63ull) / 64ull + 1ull);
        ENSURE(c, c->d_bbstart, (uint64_t)std::max<uint32_t>(c->A, 1u) * c->bs_stride * 4);
    } else { ENSURE(c, c->d_matA, c->mat_cells * 4); ENSURE(c, c->d_matD, c->mat_cells * 4); }
    ENSURE(c, c->d_matC, c->matc_cells * 4 + 256);
    ENSURE(c, c->d_cov, c->sum_bb * 4); ENSURE(c, c->d_gcount, c->sum_bb * 4);
    ENSURE(c, c->d_gbase, c->sum_bb * 4); ENSURE(c, c->d_bid, c->sum_bb * 4);
    ENSURE(c, c->d_cns_off, (size_t)T * 8); ENSURE(c, c->d_cns_len, T4);
    ENSURE(c, c->d_seg_first, (size_t)T * 8); ENSURE(c, c->d_n_seg, T4);

    // first guesses for the data-dependent arenas; a run that finds them too
    // small records the exact need on the device and is repeated once.
    c->node_cap = std::max<uint64_t>(c->node_cap, c->sum_bb + c->sum_len / 7 + 1024);
    c->pool_cap = std::max<uint64_t>(c->pool_cap, 8ull * c->node_cap + 80ull * c->sum_bb + 1024ull * T);
    c->cns_cap = std::max<uint64_t>(c->cns_cap, c->sum_bb + c->sum_bb / 4 + 1024);
    c->seg_cap = std::max<uint64_t>(c->seg_cap, (uint64_t)T * 4 + 1024);
    if ((r = ensure_arenas(c))) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->uploaded = true;
    c->tm.reruns = 0;
    return DAGCON_OK;
}

int dagcon_upload(dagcon_ctx *ctx, const dagcon_batch *b) { return upload_impl(ctx, b, nullptr, nullptr); }

int dagcon_run(dagcon_ctx *ctx) {
    if (!ctx) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c->uploaded) return fail(c, DAGCON_ERR_STATE, "dagcon_run before dagcon_upload");
    HIPCHK(c, hipSetDevice(c->device));
    int r = launch_all(c);
    if (r != DAGCON_OK) return r;
    c->ran = true; c->fetched = false;
    // debugging aid (tools/bp_pieces.py): DAGCON_DUMP=<target>:<path> leaves that target's merged graph, its bestPath cuts,
    // scores and choices in a file -- N, bp_max, pool words, then cuts row, records, pool, (score, final) pairs, best[]
    if (const char *e = getenv("DAGCON_DUMP")) {
        const uint32_t t = (uint32_t)atoi(e);
        const char *path = strchr(e, ':');
        if (path && t < c->T) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            uint64_t nb = 0, pb = 0;
            uint32_t hdr[4] = {0, c->bp_max, 0, c->seg_max};
            (void)hipMemcpy(&nb, (uint64_t *)c->d_node_base.p + t, 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&pb, (uint64_t *)c->d_pool_base.p + t, 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&hdr[0], (uint32_t *)c->d_n_nodes.p + t, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&hdr[2], (uint32_t *)c->d_pool_top.p + t, 4, hipMemcpyDeviceToHost);
            std::vector<uint32_t> cuts(c->bp_max + 2), pool(hdr[2]), best(hdr[0]);
            std::vector<DgNode> nd(hdr[0]);
            std::vector<float> sc(2 * (size_t)hdr[0]);
            (void)hipMemcpy(cuts.data(), (uint32_t *)c->d_cuts_bp.p + (uint64_t)t * (c->bp_max + 2), cuts.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(nd.data(), (DgNode *)c->d_nodes.p + nb, nd.size() * sizeof(DgNode), hipMemcpyDeviceToHost);
            (void)hipMemcpy(pool.data(), (uint32_t *)c->d_pool.p + pb, pool.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(sc.data(), (float *)c->d_score.p + 2 * nb, sc.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(best.data(), (uint32_t *)c->d_best.p + nb, best.size() * 4, hipMemcpyDeviceToHost);
            if (FILE *f = fopen(path + 1, "wb")) {
                fwrite(hdr, 4, 4, f); fwrite(cuts.data(), 4, cuts.size(), f); fwrite(nd.data(), sizeof(DgNode), nd.size(), f);
                fwrite(pool.data(), 4, pool.size(), f); fwrite(sc.data(), 4, sc.size(), f); fwrite(best.data(), 4, best.size(), f);
                // (partial-span batches: the merge's worklist -- (target, first vertex, last vertex) triples -- behind it)
                uint32_t nl = 0;
                std::vector<uint32_t> wl;
                if (c->gcuts && c->d_tile_list.p) {
                    (void)hipMemcpy(&nl, c->d_tile_list.p, 4, hipMemcpyDeviceToHost);
                    if (nl > c->tile_list_cap) nl = c->tile_list_cap;
                    wl.resize(3 * (size_t)nl);
                    if (nl) (void)hipMemcpy(wl.data(), (uint32_t *)c->d_tile_list.p + 4, wl.size() * 4, hipMemcpyDeviceToHost);
                }
                fwrite(&nl, 4, 1, f);
                if (nl) fwrite(wl.data(), 4, wl.size(), f);
                fclose(f);
            }
        }
    }
    return DAGCON_OK;
}

int dagcon_sync(dagcon_ctx *ctx) {
    if (!ctx) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DAGCON_OK;
}

// device -> host on the context's own stream.  (hipMemcpy would go through the null stream, and the stream is
// non-blocking so that a second context on the same GPU is not serialised against this one's copies.)
static hipError_t d2h(Ctx *c, void *dst, const void *src, size_t bytes) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);
}

static int read_status(Ctx *c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, d2h(c, &c->h_st, c->d_st.p, sizeof(DgStatus)));
    return DAGCON_OK;
}

int dagcon_fetch(dagcon_ctx *ctx, dagcon_results *res) {
    if (!ctx || !res) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c->ran) return fail(c, DAGCON_ERR_STATE, "dagcon_fetch before dagcon_run");
    HIPCHK(c, hipSetDevice(c->device));
    int r;
    for (int attempt = 0;; attempt++) {
        if ((r = read_status(c))) return r;
        const uint32_t f = c->h_st.err_flags;
        if (f == 0) break;
        if (f & DG_E_TARGET_MASK)       // (target-level failures never set the batch flag: see DgParams::tfail)
            return fail(c, DAGCON_ERR_INTERNAL, "unexpected batch-level flag 0x%x", f);
        if (attempt >= 6) return fail(c, DAGCON_ERR_WORKSPACE, "workspace still too small after %d re-runs (flags 0x%x)", attempt, f);
        if (f & DG_E_NORM_OVF) c->norm_cap = c->h_st.norm_top + 1024;
        if (f & DG_E_NODE_OVF) c->node_cap = c->h_st.node_need + 1024;
        if (f & DG_E_POOL_OVF) c->pool_cap = c->h_st.pool_need + 1024;
        if (f & DG_E_POOL_TGT) c->growth_pct *= 3;
        if (f & DG_E_STACK) c->stk_words *= 4;
        if (f & DG_E_LIST_OVF) c->tile_list_cap *= 4;
        if (f & DG_E_LOG_OVF) c->sh_log *= 2;
        if (f & DG_E_OUT_OVF) {
            c->cns_cap = std::max<uint64_t>(c->cns_cap, c->h_st.cns_top + 1024);
            c->seg_cap = std::max<uint64_t>(c->seg_cap, c->h_st.seg_top + 1024);
        }
        c->tm.reruns++;
        if ((r = launch_all(c))) return r;
    }
    // timings
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[4])); c->tm.ms_total = ms;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->tm.ms_normalize = ms;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->tm.ms_build = ms;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); c->tm.ms_merge = ms;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[3], c->ev[4])); c->tm.ms_bestpath = ms;

    const uint32_t T = c->T;
    // per-target outcome (ABI 2): a failure is confined to its target
    c->r_tfail.assign(T, 0); c->r_status.assign(T, DAGCON_OK);
    if (T) HIPCHK(c, d2h(c, c->r_tfail.data(), c->d_tfail.p, (size_t)T * 4));
    uint32_t n_failed = 0;
    c->err.clear();
    for (uint32_t t = 0; t < T; t++) {
        const uint32_t f = c->r_tfail[t];
        if (!f) continue;
        const int code = (f & (DG_E_BADCHAR | DG_E_NONCONF)) ? DAGCON_ERR_NONCONFORMING
                       : (f & DG_E_TOO_BIG) ? DAGCON_ERR_UNSUPPORTED : DAGCON_ERR_INTERNAL;
        c->r_status[t] = code;
        if (!n_failed++) {
            if (f & DG_E_BADCHAR) fail(c, code, "target %u: an alignment holds a byte outside printable ASCII", t);
            else if (f & DG_E_NONCONF) fail(c, code, "target %u: an alignment (after the min_len filter) leaves the backbone: start < 1 or target bases past tlen", t);
            else if (f & DG_E_TOO_BIG) fail(c, code, "target %u too large (more than 2^25 - 3 vertices or 2^30 pool words)", t);
            else fail(c, code, "device invariant violated in target %u", t);
        }
    }
    const uint64_t nseg = c->h_st.seg_top, nb = c->h_st.cns_top;
    c->r_cns_off.assign(T, 0); c->r_cns_len.assign(T, 0); c->r_seg_first.assign(T, 0); c->r_n_seg.assign(T, 0);
    c->r_tmp0.assign(nseg, 0); c->r_tmp1.assign(nseg, 0);
    if (c->r_blob_cap < nb + 1) {
        if (c->r_blob) (void)hipHostFree(c->r_blob);
        c->r_blob = nullptr; c->r_blob_cap = 0;
        const size_t want = (size_t)(nb + 1) + (size_t)(nb / 8) + 4096;
        HIPCHK(c, hipHostMalloc((void **)&c->r_blob, want, hipHostMallocDefault));
        c->r_blob_cap = want;
    }
    c->r_blob[nb] = 0;
    const bool full = !(c->opts.flags & (DAGCON_FLAG_STOP_AFTER_BUILD | DAGCON_FLAG_STOP_AFTER_MERGE));
    if (T && full) {
        HIPCHK(c, d2h(c, c->r_cns_off.data(), c->d_cns_off.p, (size_t)T * 8));
        HIPCHK(c, d2h(c, c->r_cns_len.data(), c->d_cns_len.p, (size_t)T * 4));
        HIPCHK(c, d2h(c, c->r_seg_first.data(), c->d_seg_first.p, (size_t)T * 8));
        HIPCHK(c, d2h(c, c->r_n_seg.data(), c->d_n_seg.p, (size_t)T * 4));
        if (nseg) {
            HIPCHK(c, d2h(c, c->r_tmp0.data(), c->d_seg_r0.p, nseg * 4));
            HIPCHK(c, d2h(c, c->r_tmp1.data(), c->d_seg_r1.p, nseg * 4));
        }
        if (nb) HIPCHK(c, d2h(c, c->r_blob, c->d_cns.p, nb));
    }
    c->r_seg_begin.assign(T + 1, 0);
    c->r_range0.clear(); c->r_range1.clear(); c->r_seq_off.clear(); c->r_seq_len.clear();
    uint64_t bases = 0;
    for (uint32_t t = 0; t < T; t++) {
        c->r_seg_begin[t] = c->r_range0.size();
        if (!full || !c->h_tactive[t] || c->r_tfail[t]) continue;
        for (uint32_t i = 0; i < c->r_n_seg[t]; i++) {
            const uint64_t s = c->r_seg_first[t] + i;
            const int32_t r0 = c->r_tmp0[s], r1 = c->r_tmp1[s];
            c->r_range0.push_back(r0); c->r_range1.push_back(r1);
            c->r_seq_off.push_back(c->r_cns_off[t] + (uint64_t)r0);
            c->r_seq_len.push_back((uint32_t)(r1 - r0));
            bases += (uint64_t)(r1 - r0);
        }
    }
    c->r_seg_begin[T] = c->r_range0.size();
    c->tm.consensus_bases = bases;
    c->tm.algorithmic_bytes = 2ull * c->sum_len + bases;
    c->tm.n_alignments = c->A;
    c->tm.n_columns = c->h_st.n_columns;
    c->tm.n_nodes = c->h_st.node_need;
    c->tm.merge_segments = c->h_st.n_mseg;
    if (c->sum_bb && c->h_st.node_need >= c->sum_bb) c->ins_per_pos = (double)(c->h_st.node_need - c->sum_bb) / (double)c->sum_bb;
    res->n_targets = T;
    res->n_segments = c->r_range0.size();
    res->seg_begin = c->r_seg_begin.data();
    res->range0 = c->r_range0.data(); res->range1 = c->r_range1.data();
    res->seq_off = c->r_seq_off.data(); res->seq_len = c->r_seq_len.data();
    res->seq_blob = c->r_blob; res->seq_bytes = nb;
    res->target_status = c->r_status.data(); res->n_failed = n_failed;
    c->fetched = true;
    return DAGCON_OK;
}

// diagnostic builds (-DDG_STAMPS) only: raw device counters of the last run
int dagcon_debug_counters(dagcon_ctx *ctx, unsigned long long *out8) {
    if (!ctx || !out8) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    DgStatus st;
    HIPCHK(c, d2h(c, &st, c->d_st.p, sizeof st));
    for (int i = 0; i < 16; i++) out8[i] = st.dbg[i];
    return DAGCON_OK;
}

uint32_t dagcon_align_dropped(dagcon_ctx *ctx) {
    return ctx ? reinterpret_cast<Ctx *>(ctx)->align_dropped : 0u;
}

// host arithmetic only (no device, no context): the pieces a batch of that shape would be cut into
int dagcon_debug_plan(uint32_t n_targets, uint64_t n_alignments, uint64_t sum_positions, uint32_t partial_span,
                      uint32_t max_segments, uint32_t min_segment_len, uint32_t out4[4]) {
    if (!out4) return DAGCON_ERR_INVALID_ARG;
    DgPlanIn pi;
    pi.T = n_targets; pi.n_alns = n_alignments; pi.sum_bb = sum_positions; pi.gcuts = partial_span ? 1u : 0u;
    pi.max_segments = max_segments; pi.min_segment_len = min_segment_len; pi.seg_env = 0; pi.merge_q = 1;
    const DgPlan pl = dg_plan_pieces(pi);
    out4[0] = pl.seg_max; out4[1] = pl.seg_min; out4[2] = pl.use_q; out4[3] = pl.bp_max;
    return DAGCON_OK;
}

int dagcon_get_timings(dagcon_ctx *ctx, dagcon_timings *out) {
    if (!ctx || !out) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c->ran) return fail(c, DAGCON_ERR_STATE, "no run to report");
    if (!c->fetched) {
        // timings of a run that has been synchronised but not fetched
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[4])); c->tm.ms_total = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->tm.ms_normalize = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->tm.ms_build = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); c->tm.ms_merge = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[3], c->ev[4])); c->tm.ms_bestpath = ms;
    }
    *out = c->tm;
    return DAGCON_OK;
}

int dagcon_consensus(dagcon_ctx *ctx, const dagcon_batch *batch, dagcon_results *results) {
    int r = dagcon_upload(ctx, batch);
    if (r != DAGCON_OK) return r;
    if ((r = dagcon_run(ctx)) != DAGCON_OK) return r;
    return dagcon_fetch(ctx, results);
}

static int normalize_impl(Ctx *c, dagcon_ctx *ctx, uint32_t n, const uint32_t *aln_start,
                          const uint64_t *aln_off, const uint32_t *aln_len, const char *qstr,
                          const char *tstr, uint64_t blob_bytes, const uint64_t *out_off, char *qout,
                          char *tout, uint32_t *out_len, uint32_t *out_start) {
    // one pseudo target (tlen 0) that holds every alignment; only the a1
    // kernels run, with the graph stage's conformity check switched off
    std::vector<uint32_t> tl(1, 0u);
    std::vector<uint64_t> ab = {0, n};
    dagcon_batch b;
    memset(&b, 0, sizeof b);
    b.n_targets = 1; b.tlen = tl.data(); b.aln_begin = ab.data();
    b.aln_start = aln_start; b.aln_off = aln_off; b.aln_len = aln_len;
    b.qstr = qstr; b.tstr = tstr; b.blob_bytes = blob_bytes;
    int r = dagcon_upload(ctx, &b);
    if (r != DAGCON_OK) return r;
    for (int attempt = 0;; attempt++) {
        DgParams p;
        fill_params(c, p);
        p.flags |= DG_F_A1_ONLY;
        HIPCHK(c, hipMemsetAsync(c->d_st.p, 0, sizeof(DgStatus), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_tfail.p, 0, (size_t)c->T * 4 + 4, c->stream));
        launch_normalize(c, p);
        HIPCHK(c, hipGetLastError());
        if ((r = read_status(c))) return r;
        if ((c->h_st.err_flags & DG_E_NORM_OVF) && attempt < 3) {
            c->norm_cap = c->h_st.norm_top + 1024;
            if ((r = ensure_arenas(c))) return r;
            continue;
        }
        break;
    }
    if (c->h_st.err_flags & DG_E_BADCHAR)
        return fail(c, DAGCON_ERR_NONCONFORMING, "alignment %u holds a byte outside printable ASCII", c->h_st.bad_aln);
    if (c->h_st.err_flags) return fail(c, DAGCON_ERR_INTERNAL, "normalize failed (flags 0x%x)", c->h_st.err_flags);
    std::vector<uint64_t> noff(n);
    std::vector<uint32_t> lo(n), hi(n), st(n);
    if (n) {
        HIPCHK(c, d2h(c, noff.data(), c->d_norm_off.p, (size_t)n * 8));
        HIPCHK(c, d2h(c, lo.data(), c->d_n_lo.p, (size_t)n * 4));
        HIPCHK(c, d2h(c, hi.data(), c->d_n_hi.p, (size_t)n * 4));
        HIPCHK(c, d2h(c, st.data(), c->d_n_start.p, (size_t)n * 4));
    }
    std::vector<uint16_t> cols;
    for (uint32_t a = 0; a < n; a++) {
        const uint32_t m = hi[a] - lo[a];
        cols.resize(m);
        if (m) HIPCHK(c, d2h(c, cols.data(), (const uint16_t *)c->d_norm.p + noff[a] + lo[a], (size_t)m * 2));
        for (uint32_t i = 0; i < m; i++) {
            qout[out_off[a] + i] = (char)(cols[i] & 0xff);
            tout[out_off[a] + i] = (char)(cols[i] >> 8);
        }
        out_len[a] = m;
        out_start[a] = st[a];
    }
    return DAGCON_OK;
}

int dagcon_normalize(dagcon_ctx *ctx, uint32_t n, const uint32_t *aln_start, const uint64_t *aln_off,
                     const uint32_t *aln_len, const char *qstr, const char *tstr, uint64_t blob_bytes,
                     uint32_t trim, uint32_t flags, const uint64_t *out_off, char *qout, char *tout,
                     uint32_t *out_len, uint32_t *out_start) {
    if (!ctx) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (n && (!aln_start || !aln_off || !aln_len || !qstr || !tstr || !out_off || !qout || !tout || !out_len || !out_start))
        return fail(c, DAGCON_ERR_INVALID_ARG, "NULL argument");
    if (n > DAGCON_MAX_COVERAGE)
        return fail(c, DAGCON_ERR_UNSUPPORTED, "dagcon_normalize takes at most %u alignments per call", DAGCON_MAX_COVERAGE);
    const dagcon_opts saved = c->opts;
    c->opts.min_cov = 0; c->opts.min_len = 0; c->opts.trim = trim;
    c->opts.flags = flags & DAGCON_FLAG_RAW_ALIGNMENTS;
    const int r = normalize_impl(c, ctx, n, aln_start, aln_off, aln_len, qstr, tstr, blob_bytes, out_off,
                                 qout, tout, out_len, out_start);
    c->opts = saved;
    c->uploaded = false; c->ran = false;
    return r;
}

// the -a stage on the device: aligned strings left in c->d_al[7] / [8] at out_off[a], their lengths in aln_len (host)
static int align_device(Ctx *c, uint32_t n, const uint64_t *q_off, const uint32_t *q_len,
                        const uint64_t *t_off, const uint32_t *t_len, const char *q_blob, uint64_t q_bytes,
                        const char *t_blob, uint64_t t_bytes, const uint64_t *out_off, uint32_t *aln_len, uint64_t *out_bytes_ret) {
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t out_bytes = 0;
    std::vector<uint64_t> dir_off(n);
    for (uint32_t a = 0; a < n; a++) {
        if (q_off[a] > q_bytes || q_len[a] > q_bytes - q_off[a] || t_off[a] > t_bytes || t_len[a] > t_bytes - t_off[a])
            return fail(c, DAGCON_ERR_INVALID_ARG, "pair %u runs past its blob", a);
        if ((uint64_t)q_len[a] + t_len[a] > 0x7FFFFFF0ull) return fail(c, DAGCON_ERR_UNSUPPORTED, "pair %u too long", a);
        out_bytes = std::max<uint64_t>(out_bytes, out_off[a] + (uint64_t)q_len[a] + t_len[a]);
    }
    DevBuf &dq = c->d_al[0], &dt = c->d_al[1], &dqo = c->d_al[2], &dto = c->d_al[3], &dql = c->d_al[4], &dtl = c->d_al[5],
           &doo = c->d_al[6], &dqa = c->d_al[7], &dta = c->d_al[8], &dlen = c->d_al[9], &ddir = c->d_al[10], &ddo = c->d_al[11];
    ENSURE(c, dq, q_bytes); ENSURE(c, dt, t_bytes);
    ENSURE(c, dqo, (size_t)n * 8); ENSURE(c, dto, (size_t)n * 8); ENSURE(c, dql, (size_t)n * 4); ENSURE(c, dtl, (size_t)n * 4);
    ENSURE(c, doo, (size_t)n * 8); ENSURE(c, dqa, out_bytes); ENSURE(c, dta, out_bytes); ENSURE(c, dlen, (size_t)n * 4);
    ENSURE(c, ddo, (size_t)n * 8);
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(dq.p, q_blob, q_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dt.p, t_blob, t_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dqo.p, q_off, (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dto.p, t_off, (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dql.p, q_len, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dtl.p, t_len, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(doo.p, out_off, (size_t)n * 8, hipMemcpyHostToDevice, s));
    // Two passes (k_align.hip.h): every pair in the narrow band first; the pairs whose path came near an edge of it
    // (DG_AL_RETRY) again in the full band.  Inside a pass: groups of as many pairs as fit the direction budget
    // (one wave per pair and ~1 us per row: what counts is how many pairs are in flight; but a hipMalloc of tens
    // of GB takes seconds on this platform, so 32 GB at most, a quarter of the free memory), and inside a group
    // one launch per kernel instance (cells per lane).
    uint64_t budget_rows = (6ull << 30) / 256ull;
    {
        size_t mfree = 0, mtotal = 0;
        if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess) {
            const uint64_t have = (uint64_t)mfree + (uint64_t)ddir.cap;      // (the buffer of the last call is ours to reuse)
            budget_rows = std::min<uint64_t>(16ull << 30, std::max<uint64_t>(1ull << 30, have / 4)) / 256ull;
        }
    }
    if (const char *e = getenv("DAGCON_ALIGN_GB")) { const long long v = atoll(e); if (v >= 1 && v <= 200) budget_rows = ((uint64_t)v << 30) / 256ull; }
    if (const char *e = getenv("DAGCON_ALIGN_ROWS")) { const long long v = atoll(e); if (v >= 1) budget_rows = (uint64_t)v; }   // test knob
    const bool t_dbg = getenv("DAGCON_ALIGN_TIMING") != nullptr;
    if (t_dbg) HIPCHK(c, hipStreamSynchronize(s));
    double t_grp = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    DevBuf &didx = c->d_al[12], &dhw = c->d_al[13];
    ENSURE(c, didx, (size_t)n * 4); ENSURE(c, dhw, (size_t)n * 4);
    std::vector<uint32_t> todo(n), halfw(n), order(n);
    for (uint32_t a = 0; a < n; a++) todo[a] = a;
    DgAlignParams ap;
    ap.q = (const uint8_t *)dq.p; ap.t = (const uint8_t *)dt.p;
    ap.q_off = (const uint64_t *)dqo.p; ap.t_off = (const uint64_t *)dto.p;
    ap.q_len = (const uint32_t *)dql.p; ap.t_len = (const uint32_t *)dtl.p;
    ap.out_off = (const uint64_t *)doo.p; ap.qaln = (uint8_t *)dqa.p; ap.taln = (uint8_t *)dta.p;
    ap.aln_len = (uint32_t *)dlen.p; ap.dir_off = (const uint64_t *)ddo.p; ap.halfw = (const uint32_t *)dhw.p;
    // the band that follows the alignment first (k_align_adapt): every pair long enough for a static band wider than it
    {
        std::vector<uint32_t> ad, rest;
        for (uint32_t a = 0; a < n; a++) (dg_align_halfwidth_first(q_len[a], t_len[a]) > DG_AL_WA ? ad : rest).push_back(a);
        if (getenv("DAGCON_ALIGN_STATIC")) { rest.insert(rest.end(), ad.begin(), ad.end()); ad.clear(); }      // test knob
        std::stable_sort(ad.begin(), ad.end(), [&](uint32_t x, uint32_t y) { return q_len[x] > q_len[y]; });
        // groups of equal size (a small last one would run at the latency of its longest pair)
        uint64_t all_rows = 0;
        for (uint32_t a : ad) all_rows += dg_align_rows_adapt(q_len[a], t_len[a]);
        const uint64_t ngrp = std::max<uint64_t>(1, (all_rows + budget_rows - 1) / budget_rows);
        const uint64_t grp_rows = std::min<uint64_t>(budget_rows, all_rows / ngrp + 1 + (all_rows / ngrp) / 64);
        size_t first = 0;
        while (first < ad.size()) {
            uint64_t rows = 0;
            size_t cnt = 0;
            while (first + cnt < ad.size()) {
                const uint32_t a = ad[first + cnt];
                const uint64_t r = dg_align_rows_adapt(q_len[a], t_len[a]);
                if (cnt && rows + r > grp_rows) break;
                dir_off[a] = rows;
                rows += r; cnt++;
            }
            ENSURE(c, ddir, rows * 256ull);
            ap.dirs = (uint32_t *)ddir.p;
            HIPCHK(c, hipMemcpyAsync(ddo.p, dir_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
            HIPCHK(c, hipMemcpyAsync((uint32_t *)didx.p + first, ad.data() + first, cnt * 4, hipMemcpyHostToDevice, s));
            ap.idx = (const uint32_t *)didx.p + first; ap.n = (uint32_t)cnt; ap.first_pass = 1u;
#ifdef DG_EXPERIMENTS
            if (c->align2) hipLaunchKernelGGL(k_align_adapt2, dim3((uint32_t)((cnt + 1) / 2)), dim3(64), 0, s, ap);      // two pairs per wave
            else
#endif
            hipLaunchKernelGGL(k_align_adapt, dim3((uint32_t)cnt), dim3(64), 0, s, ap);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(s));
            if (t_dbg) {
                const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
                fprintf(stderr, "dagcon_align: following band, group of %zu pairs, %.1f MB of directions: %.2f ms\n", cnt, rows * 256.0 / 1e6, (now - t_grp) * 1e3);
                t_grp = now;
            }
            first += cnt;
        }
        if (!ad.empty()) {
            HIPCHK(c, d2h(c, aln_len, dlen.p, (size_t)n * 4));
            size_t back = 0;
            for (uint32_t a : ad) if (aln_len[a] == DG_AL_RETRY) { rest.push_back(a); back++; }
            if (t_dbg) fprintf(stderr, "dagcon_align: %zu of %zu pairs go on to the static bands\n", back, ad.size());
        }
        std::sort(rest.begin(), rest.end());
        todo.swap(rest);
    }
    for (int pass = 0; pass < 2 && !todo.empty(); pass++) {
        // (a pair whose first band is the full one already is final in the first pass: its width says so)
        for (uint32_t a : todo) halfw[a] = pass == 0 ? dg_align_halfwidth_first(q_len[a], t_len[a]) : dg_align_halfwidth(q_len[a], t_len[a]);
        HIPCHK(c, hipMemcpyAsync(dhw.p, halfw.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        static const uint32_t kinds[6] = {2, 4, 6, 8, 12, 16};
        size_t first = 0;
        while (first < todo.size()) {
            uint64_t rows = 0;
            size_t cnt = 0;
            while (first + cnt < todo.size()) {
                const uint32_t a = todo[first + cnt];
                const uint64_t r = dg_align_rows(q_len[a], t_len[a], dg_align_cells(halfw[a]));
                if (cnt && rows + r > budget_rows) break;
                dir_off[a] = rows;
                rows += r; cnt++;
            }
            ENSURE(c, ddir, rows * 256ull);
            ap.dirs = (uint32_t *)ddir.p;
            HIPCHK(c, hipMemcpyAsync(ddo.p, dir_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
            // the group's pairs by kernel instance, the long ones first inside each; two launches per instance when the
            // pass is the first one: pairs whose narrow band IS the full band are final at once
            size_t fill = 0;
            for (int k = 0; k < 6; k++) {
                for (int fin = 0; fin < 2; fin++) {
                    const size_t k0 = fill;
                    for (size_t x = 0; x < cnt; x++) {
                        const uint32_t a = todo[first + x];
                        const bool is_final = pass == 1 || halfw[a] == dg_align_halfwidth(q_len[a], t_len[a]);
                        if (dg_align_cells(halfw[a]) == kinds[k] && (int)is_final == fin) order[first + fill++] = a;
                    }
                    const uint32_t nk = (uint32_t)(fill - k0);
                    if (!nk) continue;
                    std::stable_sort(order.begin() + first + k0, order.begin() + first + fill,
                                     [&](uint32_t x, uint32_t y) { return q_len[x] > q_len[y]; });
                    HIPCHK(c, hipMemcpyAsync((uint32_t *)didx.p + first + k0, order.data() + first + k0, (size_t)nk * 4, hipMemcpyHostToDevice, s));
                    ap.idx = (const uint32_t *)didx.p + first + k0; ap.n = nk; ap.first_pass = fin ? 0u : 1u;
                    switch (kinds[k]) {
                        case 2: hipLaunchKernelGGL(k_align_band<2>, dim3(nk), dim3(64), 0, s, ap); break;
                        case 4: hipLaunchKernelGGL(k_align_band<4>, dim3(nk), dim3(64), 0, s, ap); break;
                        case 6: hipLaunchKernelGGL(k_align_band<6>, dim3(nk), dim3(64), 0, s, ap); break;
                        case 8: hipLaunchKernelGGL(k_align_band<8>, dim3(nk), dim3(64), 0, s, ap); break;
                        case 12: hipLaunchKernelGGL(k_align_band<12>, dim3(nk), dim3(64), 0, s, ap); break;
                        default: hipLaunchKernelGGL(k_align_band<16>, dim3(nk), dim3(64), 0, s, ap); break;
                    }
                    HIPCHK(c, hipGetLastError());
                }
            }
            HIPCHK(c, hipStreamSynchronize(s));       // (the direction buffer and the offsets are reused by the next group)
            if (t_dbg) {
                const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
                fprintf(stderr, "dagcon_align: pass %d, group of %zu pairs, %.1f MB of directions: %.2f ms\n", pass, cnt, rows * 256.0 / 1e6, (now - t_grp) * 1e3);
                t_grp = now;
            }
            first += cnt;
        }
        if (pass == 0) {
            HIPCHK(c, d2h(c, aln_len, dlen.p, (size_t)n * 4));
            std::vector<uint32_t> again;
            for (uint32_t a : todo) if (aln_len[a] == DG_AL_RETRY) again.push_back(a);
            if (t_dbg) fprintf(stderr, "dagcon_align: %zu of %u pairs go to the full band\n", again.size(), n);
            todo.swap(again);
        }
    }
    HIPCHK(c, d2h(c, aln_len, dlen.p, (size_t)n * 4));
    uint32_t dropped = 0;
    for (uint32_t a = 0; a < n; a++) {
        if ((uint64_t)aln_len[a] > (uint64_t)q_len[a] + t_len[a]) return fail(c, DAGCON_ERR_INTERNAL, "pair %u: alignment longer than its room", a);
        // the band could not connect the corners (sequences of very different lengths, indels beyond the widest band):
        // length 0, and the record then falls to the min_len filter -- the reference's SDPAlign always returns something
        dropped += aln_len[a] == 0 && (q_len[a] || t_len[a]);
    }
    c->align_dropped = dropped;                   // (the call succeeds: dagcon_align_dropped reports them)
    *out_bytes_ret = out_bytes;
    return DAGCON_OK;
}

int dagcon_align(dagcon_ctx *ctx, uint32_t n, const uint64_t *q_off, const uint32_t *q_len,
                 const uint64_t *t_off, const uint32_t *t_len, const char *q_blob, uint64_t q_bytes,
                 const char *t_blob, uint64_t t_bytes, const uint64_t *out_off, char *qaln, char *taln,
                 uint32_t *aln_len) {
    if (!ctx) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (n == 0) return DAGCON_OK;
    if (!q_off || !q_len || !t_off || !t_len || !q_blob || !t_blob || !out_off || !qaln || !taln || !aln_len)
        return fail(c, DAGCON_ERR_INVALID_ARG, "NULL argument");
    uint64_t out_bytes = 0;
    int r = align_device(c, n, q_off, q_len, t_off, t_len, q_blob, q_bytes, t_blob, t_bytes, out_off, aln_len, &out_bytes);
    if (r != DAGCON_OK) return r;
    HIPCHK(c, d2h(c, qaln, c->d_al[7].p, out_bytes));
    HIPCHK(c, d2h(c, taln, c->d_al[8].p, out_bytes));
    return DAGCON_OK;
}

// main.cpp:117-145 with -a in one call: every record re-aligned (SimpleAligner.cpp:25-63), start / end / strand as
// SimpleAligner.cpp:51-62, then the usual path; the aligned strings never leave the device
int dagcon_consensus_pre(dagcon_ctx *ctx, const dagcon_pre_batch *b, dagcon_results *results) {
    if (!ctx || !b || !results) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    const uint32_t T = b->n_targets;
    if (T && (!b->tlen || !b->rec_begin)) return fail(c, DAGCON_ERR_INVALID_ARG, "tlen/rec_begin is NULL");
    const uint64_t n64 = T ? b->rec_begin[T] : 0;
    if (n64 > 0xFFFFFFF0ull) return fail(c, DAGCON_ERR_UNSUPPORTED, "too many records");
    const uint32_t n = (uint32_t)n64;
    if (n && (!b->tstart || !b->strand || !b->q_off || !b->q_len || !b->t_off || !b->t_len || !b->q_blob || !b->t_blob))
        return fail(c, DAGCON_ERR_INVALID_ARG, "record arrays are NULL");
    std::vector<uint64_t> out_off(n);
    std::vector<uint32_t> alen(n, 0), start(n);
    uint64_t tot = 0;
    for (uint32_t a = 0; a < n; a++) { out_off[a] = tot; tot += ((uint64_t)b->q_len[a] + b->t_len[a] + 15ull) & ~15ull; }
    uint64_t out_bytes = 0;
    if (n) {
        int r = align_device(c, n, b->q_off, b->q_len, b->t_off, b->t_len, b->q_blob, b->q_bytes, b->t_blob, b->t_bytes,
                             out_off.data(), alen.data(), &out_bytes);
        if (r != DAGCON_OK) return r;
    }
    // SimpleAligner.cpp:51-62 (the alignment is global: GenomicTBegin() = 0, GenomicTEnd() = |tseq|)
    std::vector<uint32_t> rc_list;
    for (uint32_t g = 0; g < T; g++) {
        if (b->rec_begin[g + 1] < b->rec_begin[g] || b->rec_begin[g + 1] > n64) return fail(c, DAGCON_ERR_INVALID_ARG, "rec_begin not monotone at target %u", g);
        for (uint64_t a = b->rec_begin[g]; a < b->rec_begin[g + 1]; a++) {
            uint32_t st = b->tstart[a];
            const uint32_t en = st + b->t_len[a];
            if (b->strand[a] == '-') { st = b->tlen[g] - en; if (alen[a]) rc_list.push_back((uint32_t)a); }
            start[a] = st + 1u;
        }
    }
    if (!rc_list.empty()) {
        DevBuf &didx = c->d_al[12];
        HIPCHK(c, hipMemcpyAsync(didx.p, rc_list.data(), rc_list.size() * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_align_revcomp, dim3((uint32_t)rc_list.size()), dim3(64), 0, c->stream,
                           (uint8_t *)c->d_al[7].p, (uint8_t *)c->d_al[8].p, (const uint64_t *)c->d_al[6].p,
                           (const uint32_t *)c->d_al[9].p, (const uint32_t *)didx.p);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));       // (rc_list is a local)
    }
    dagcon_batch db;
    memset(&db, 0, sizeof db);
    db.n_targets = T; db.tlen = b->tlen; db.aln_begin = b->rec_begin;
    db.aln_start = start.data(); db.aln_off = out_off.data(); db.aln_len = alen.data();
    db.blob_bytes = n ? out_bytes : 0;
    int r = upload_impl(ctx, &db, n ? c->d_al[7].p : nullptr, n ? c->d_al[8].p : nullptr);
    if (r != DAGCON_OK) return r;
    if ((r = dagcon_run(ctx)) != DAGCON_OK) return r;
    return dagcon_fetch(ctx, results);
}

int dagcon_host_alloc(dagcon_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return DAGCON_OK;
}

void dagcon_host_free(dagcon_ctx *ctx, void *p) {
    if (!ctx || !p) return;
    (void)hipSetDevice(reinterpret_cast<Ctx *>(ctx)->device);
    (void)hipHostFree(p);
}

int dagcon_debug_graph(dagcon_ctx *ctx, uint32_t target, dagcon_graph_dump *out) {
    if (!ctx || !out) return DAGCON_ERR_INVALID_ARG;
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c->ran) return fail(c, DAGCON_ERR_STATE, "dagcon_debug_graph before dagcon_run");
    if (target >= c->T) return fail(c, DAGCON_ERR_INVALID_ARG, "target out of range");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t nb = 0, pb = 0;
    uint32_t N = 0, psz = 0;
    HIPCHK(c, d2h(c, &nb, (uint64_t *)c->d_node_base.p + target, 8));
    HIPCHK(c, d2h(c, &pb, (uint64_t *)c->d_pool_base.p + target, 8));
    HIPCHK(c, d2h(c, &N, (uint32_t *)c->d_n_nodes.p + target, 4));
    HIPCHK(c, d2h(c, &psz, (uint32_t *)c->d_pool_top.p + target, 4));
    if (!c->h_tactive[target]) N = 0;
    std::vector<DgNode> nd(N);
    std::vector<uint32_t> pool(psz);
    std::vector<int32_t> cov(c->h_tlen[target] + 2, 0);
    c->g_weight.assign(N, 0); c->g_cov.assign(N, 0);
    if (N) {
        HIPCHK(c, d2h(c, nd.data(), (DgNode *)c->d_nodes.p + nb, (size_t)N * sizeof(DgNode)));
        const uint32_t nbb = c->h_tlen[target] + 2;
        HIPCHK(c, d2h(c, cov.data(), (int32_t *)c->d_cov.p + c->h_bbv_base[target], (size_t)nbb * 4));
        if (psz) HIPCHK(c, d2h(c, pool.data(), (uint32_t *)c->d_pool.p + pb, (size_t)psz * 4));
    }
    c->g_base.assign(N, 0); c->g_deleted.assign(N, 0); c->g_backbone.assign(N, 0); c->g_bbpos.assign(N, 0);
    c->g_out_begin.assign(N + 1, 0); c->g_in_begin.assign(N + 1, 0);
    c->g_out_dst.clear(); c->g_out_cnt.clear(); c->g_in_src.clear();
    uint32_t nbb_seen = 0;
    for (uint32_t v = 0; v < N; v++) {
        c->g_base[v] = nd[v].base;
        c->g_weight[v] = nd[v].weight;
        c->g_deleted[v] = (nd[v].flags & DG_NF_DELETED) ? 1 : 0;
        c->g_backbone[v] = (nd[v].flags & DG_NF_BACKBONE) ? 1 : 0;
        if (c->g_backbone[v]) { c->g_bbpos[v] = (int32_t)nbb_seen; c->g_cov[v] = cov[nbb_seen]; nbb_seen++; }
        else c->g_bbpos[v] = nd[v].bbpos;
        c->g_out_begin[v] = (uint32_t)c->g_out_dst.size();
        c->g_in_begin[v] = (uint32_t)c->g_in_src.size();
        for (uint32_t i = 0; i < nd[v].out_len; i++) {
            c->g_out_dst.push_back((int32_t)pool[nd[v].out_off + 2 * i]);
            c->g_out_cnt.push_back((int32_t)pool[nd[v].out_off + 2 * i + 1]);
        }
        for (uint32_t i = 0; i < nd[v].in_len; i++) c->g_in_src.push_back((int32_t)pool[nd[v].in_off + i]);
    }
    c->g_out_begin[N] = (uint32_t)c->g_out_dst.size();
    c->g_in_begin[N] = (uint32_t)c->g_in_src.size();
    out->n_nodes = N;
    out->base = c->g_base.data(); out->weight = c->g_weight.data(); out->coverage = c->g_cov.data();
    out->deleted = c->g_deleted.data(); out->backbone = c->g_backbone.data(); out->bbpos = c->g_bbpos.data();
    out->out_begin = c->g_out_begin.data(); out->out_dst = c->g_out_dst.data(); out->out_count = c->g_out_cnt.data();
    out->in_begin = c->g_in_begin.data(); out->in_src = c->g_in_src.data();
    return DAGCON_OK;
}

}  // extern "C"
