// dagcon_dev.h -- device-side data layout shared by the kernels and the host API.
//
// One batch = T independent targets.  Everything lives in a handful of HBM
// arenas owned by the context; per-target and per-alignment arrays hold
// offsets into them.
//
// Vertex ids are target-local int32, numbered in BACKBONE-POSITION ORDER:
//   for p = 0 .. blen+1:   [inserted vertices whose _bbMap is p, in (read, column) order]
//                          [backbone vertex p]            (p = 0 enter '^', p = blen+1 exit '$')
// so id(backbone p) = bid[p] = gbase[p] + gcount[p], enter = 0, exit = N-1, and every edge of
// the freshly built graph runs from a lower id to a higher id.  Ids are only names: what the
// reference's results depend on is the ORDER of each vertex's out- and in-list (SURVEY
// Appendix A.1/A.2), which is kept exactly.  Position order makes the FIFO sweep of mergeNodes
// and the reverse sweep of bestPath walk memory monotonically (cache lines are reused, not
// re-fetched), which is what a latency-bound pointer chase needs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DG_GAP '-'

// DgParams::flags: bits 0..2 are dagcon_opts.flags; internal bits from 8 up
#define DG_F_RAW      1u
#define DG_F_A1_ONLY  8u   // dagcon_normalize: no graph follows, skip the backbone conformity check
#define DG_F_RESWEEP  16u  // DAGCON_FLAG_DEBUG_RESWEEP

// status bits (DgStatus::err_flags)
#define DG_E_BADCHAR     0x001u  // byte outside printable ASCII in an alignment string
#define DG_E_NONCONF     0x002u  // alignment leaves the backbone (start<1 or runs past tlen)
#define DG_E_NORM_OVF    0x004u  // normalised-column arena too small (rerun with norm_top)
#define DG_E_NODE_OVF    0x008u  // vertex arena too small (rerun with node_need)
#define DG_E_POOL_OVF    0x010u  // adjacency pool arena too small (rerun with pool_need)
#define DG_E_POOL_TGT    0x020u  // one target outgrew its pool share (rerun with larger growth factor)
#define DG_E_STACK       0x040u  // mergeInNodes recursion scratch exhausted
#define DG_E_INTERNAL    0x080u  // invariant violated (empty list dereference, list > 65535)
#define DG_E_OUT_OVF     0x100u  // output arena too small
#define DG_E_TOO_BIG     0x200u  // a target has more than 2^25-2 vertices
#define DG_E_LOG_OVF     0x800u  // a segment appended more entries to enter's / exit's list than its slots hold (rerun with more)
#define DG_E_LIST_OVF    0x400u  // more tiles handed to k_merge_list than its list holds (rerun with a longer one)

// failures of one target (its input, or an invariant of its graph): recorded in DgParams::tfail,
// the batch goes on; everything else is a capacity problem of the whole batch (grow and re-run)
#define DG_E_TARGET_MASK (DG_E_BADCHAR | DG_E_NONCONF | DG_E_INTERNAL | DG_E_TOO_BIG)

struct DgStatus {
    uint32_t err_flags;
    uint32_t bad_aln;
    uint32_t bad_target;
    uint32_t n_mseg;               // segments of the merge / bestPath sweeps, all targets (k_cuts)
    unsigned long long norm_top;   // bump cursor into the column arena (uint16 units)
    unsigned long long node_need;  // exact vertex count of the batch (set by carve)
    unsigned long long pool_need;  // exact pool words of the batch (set by carve)
    unsigned long long cns_top;    // bump cursor into the consensus blob
    unsigned long long seg_top;    // bump cursor into the segment arrays
    unsigned long long n_columns;  // normalised, trimmed columns
    unsigned long long ovf_top;    // bump cursor into the re-run region of the chunk scratch (uint16 units)
    unsigned long long dbg[16];     // diagnostic build only (DG_STAMPS): cycle / visit counters of target 0
};

// One 32-byte record per vertex.  The first 16 bytes are what a neighbour needs to know
// about it (one dwordx4 gather); the second 16 bytes say where its own lists live.
// out entry i: pool[out_off + 2i] = dst, pool[out_off + 2i + 1] = count
// in  entry i: pool[in_off + i]   = src          (edge counts live on the out side only)
struct __attribute__((aligned(16))) DgNode {
    uint16_t out_len, in_len;
    uint8_t base, flags;
    uint16_t pad;
    int32_t weight;
    int32_t pending;               // in-edges whose source is not processed yet (merge);
                                   // out-edges whose target is not scored yet (bestPath)
    uint32_t out_off, in_off;
    uint16_t out_cap, in_cap;
    int32_t bbpos;                 // _bbMap: absent key (enter, exit) reads as 0
};
#define DG_NF_BACKBONE 1u
#define DG_NF_DELETED  2u
#define DG_NF_DEFER    8u   // bestPath on partial-span pileups: scored by k_bp_defer, after the segments' sweeps
#define DG_NF_SHARED   4u   // during k_merge_list: a list of this vertex is shared by the segments' workers (DgGraph::sh)

// arrival cell: (target base << 25) | (source id + 1); deletion: id field all ones; DG_CELL_DUP (k_build.hip.h): a match
// whose in-edge an earlier read's folded chain already brings
// departure cell: (target id + 1) | (reads folded into this one's chain << 25)
#define DG_CELL_ID(c)   ((c) & 0x1FFFFFFu)
#define DG_CELL_DEL     0x1FFFFFFu
#define DG_CELL_BASE(c) ((uint8_t)((c) >> 25))
#define DG_MAX_NODES    0x1FFFFFDu
#define DG_EMIT_SEG     512u   // backbone positions per k_emit wave: default of DgParams::emit_shift (1 << 9)
#define DG_CK_NONE      0xFFFFFFFFu
#define DG_DEFER_MAX    64u
#define DG_SH_MAX       8u            // shared out-lists per target (enter + vertices of the prologue that reach into several segments)
#define DG_TOMB         0xFFFFFFFFu   // erased entry of a shared list (enter's out-list, exit's in-list)

struct DgParams {
    // ---- inputs (resident in HBM after dagcon_upload) ----
    const uint8_t *q, *t;
    const uint64_t *aln_off;
    const uint32_t *aln_len, *aln_start, *aln_tgt;
    const uint32_t *tlen;
    const uint64_t *aln_begin;     // [T+1], indices into the (filtered) alignment arrays
    const uint8_t *tactive;        // [T] 1 = build a graph (main.cpp:66-72,118)
    uint32_t *tfail;               // [T] DG_E_* bits of a failure confined to the target (bad input, broken
                                   // invariant): later kernels skip it, the other targets of the batch complete
    const uint8_t *bb;             // optional backbone blob
    const uint64_t *bb_off;
    const uint64_t *mat_base;      // [T] offset into matA/matD/matC: (tlen+2) * K cells
    const uint64_t *bbv_base;      // [T] offset into the per-position arrays: tlen+2 cells
    uint32_t T, A;
    uint32_t trim, min_len;
    int32_t min_weight;
    uint32_t flags;
    uint32_t max_k, max_tlen;
    // ---- per alignment work arrays ----
    uint32_t *nmis;
    uint64_t *norm_off;
    uint32_t *n_lo, *n_hi, *n_start, *n_ins, *n_del;
    uint16_t *norm;                // column arena: low byte q, high byte t
    uint64_t norm_cap;
    // ---- chunked normalizeGaps (k_norm_*): window c of alignment a is chunk ch_base[a] + c ----
    const uint32_t *ch_aln;        // [n_chunks] alignment of the chunk
    const uint32_t *ch_base;       // [A + 1]
    uint32_t n_chunks;
    uint32_t *ch_k0;               // first input column, DG_CH_NONE: no chunk starts in this window
    uint32_t *ch_next;             // window (of the alignment) the next chunk starts in
    uint32_t *ch_w, *ch_tb;        // columns written, of which with a target base
    uint32_t *ch_flag;             // 1 = the alignment goes to k_normalize_slow
    uint64_t *ch_src;              // where the chunk's columns are, in norm_tmp
    uint32_t *ch_out, *ch_adv;     // after k_norm_scan: column index / target bases in front, DG_CH_NONE: no chunk
    uint32_t *n_lb;                // [A] target bases trimAln took off the left end
    uint16_t *norm_tmp;            // chunk scratch: 2 columns per input column, then the re-run region
    // column of alignment a where backbone position s << emit_shift begins (its insertion run first),
    // at ckpt[ck_base[a] + s]; DG_CK_NONE where the read has no column at that position
    uint32_t *ckpt;
    const uint32_t *ck_base;       // [A]
    uint32_t emit_shift;           // log2 of the positions per k_emit wave (>= 4: whole 16-position batches)
    uint64_t tmp_main, tmp_cap;    // uint16 units: start of the re-run region, end of the scratch
    // ---- per target work arrays ----
    uint64_t *node_base;
    uint32_t *n_nodes;
    uint64_t *pool_base;
    uint32_t *pool_size, *pool_top;
    uint32_t *t_nins;              // inserted vertices of the target
    // ---- per backbone position (bbv_base indexed) ----
    uint32_t *gcount;              // inserted vertices whose _bbMap is p
    uint32_t *gbase;               // id of the first vertex of group p
    uint32_t *bid;                 // id of backbone vertex p
    int32_t *cov;                  // coverage (AlnGraphBoost.cpp:76,87)
    // ---- matrices [position][read] ----
    uint32_t *matA, *matD;         // arrival / departure
    uint8_t *matK;                 // (p.emit2) [read][position], a byte: key of the short insertion chain a match column closes (k_dedupe)
    uint32_t *bbstart;             // (p.emit2) [A][bs_stride]: backbone position at the start of every 64-column block (k_blockscan)
    uint32_t bs_stride;
    uint32_t emit2;                // 1: addAln with a thread per column (k_emit2.hip.h): matA / matD are [read][position] like matC
    uint32_t *matC;                // [read][position] (row stride matc_stride): insertion run length in front of
                                   // the position, then its exclusive prefix over reads
    const uint64_t *matc_base;     // [T] offset of the target's K rows
    const uint32_t *matc_stride;   // [T] (tlen + 2) rounded up to a multiple of 4 cells
    // ---- vertex arena ----
    DgNode *nodes;
    int32_t *best, *queue;
    float2 *score;                 // (best-path score, 1 = final)
    float *bp_tt;                  // per vertex: what an edge into it subtracts (k_bp_terms)
    float *score_b;                // per vertex (p.gcuts, p.bp_fused): B, best path to exit that does not pass the piece's upper cut
    uint32_t bp_fused;             // 1: partial-span bestPath as one (A, B) sweep + vertex-parallel kernels (k_bp_sweep_ab)
    uint32_t bp_seg_min;           // k_cuts2: shortest bestPath piece, in backbone positions (see fill_params)
    uint32_t bp_lane;              // 1: full-span bestPath with a lane per piece (k_bp_sweep_l); k_bp_sweep then takes the pieces it gave up
    uint32_t bl_stk;               // k_bp_sweep_l: entries of a lane's evaluation stack (<= DG_BL_STK; a test knob below that)
    uint8_t *cns_tmp;
    uint64_t node_cap;
    uint32_t *pool;
    uint64_t pool_cap;
    int32_t *stk;                  // per-target scratch, stk_words each
    uint32_t stk_words;
    uint32_t growth_pct;           // pool growth region as % of the initial adjacency words
    uint32_t pf_ahead;             // vertices the prefetch wave runs ahead of the sweep (0 = off)
    uint32_t emit_scan;            // 1 (every target has at most 64 reads): matC keeps the insertion run lengths, k_gsum adds them up
                                   // per position (gcount) and k_emit takes the prefix over the reads itself (DPP): no k_groups
    uint32_t fold;                 // 1: k_emit folds duplicate insertion chains as it builds (dg_emit_fold); 0 with
                                   // DAGCON_FLAG_STOP_AFTER_BUILD (the dump is then addAln's graph itself) and DAGCON_FOLD=0
    uint32_t q_kmax;               // k_merge_q takes the targets of at most this many reads, k_merge the deeper ones (0: no split)
    uint32_t seg_max;              // most segments a target's merge sweep is split into (k_cuts)
    uint32_t seg_min;              // shortest backbone stretch worth a worker of its own
    uint32_t *cuts;                // [T][seg_max + 2]: segment count, first vertex of each segment
    uint32_t bp_max;               // most segments of a target's bestPath sweep (finer than the merge's:
    uint32_t *cuts_bp;             //   its waves are light), [T][bp_max + 2] like cuts
    float *bp_stat;                // [T][seg_max][2]: largest |score| of the segment, score of its first vertex
    uint32_t *bp_len;              // [T][seg_max]: vertices of the best path inside the segment (enter / exit excluded)
    uint32_t *bp_end;              // [T][bp_max] (p.gcuts): where a piece of the walk ended (k_bp_walk)
    float *bp_ab;                  // [T][bp_max][4] (p.gcuts): A, B of the segment's first vertex, absolute score of its upper cut
    uint32_t *defer;               // [T][DG_DEFER_MAX + 1] (p.gcuts): count, then the vertices k_bp_defer scores
    uint8_t *cns_tmp0;             // (p.gcuts) the walk's first piece, from enter to the first cut it meets
    // ---- cuts for partial-span pileups (k_readspan, k_merge_pro, k_cuts2, k_merge_list, k_merge_fin) ----
    uint32_t gcuts;                // 1: mergeNodes runs as prologue + worklist of segments + epilogue
    uint32_t *rd_s, *rd_e;         // [A] first / last backbone position a read consumes (0 / 0: the read is empty)
    uint32_t *rd_lead, *rd_trail;  // [A] insertion run in front of the first / behind the last position
    int32_t *queue0;               // FIFO of the prologue and of the segment that resumes it (it also holds the vertices
                                   // the prologue visits beyond that segment: a stretch of `queue` would be too short)
    uint32_t *pro_state;           // [T][4]: queue head, queue tail of the prologue, vertices it visited, 1 = no cuts allowed
    uint32_t *sh_cnt;              // [T][2 + 2 * DG_SH_MAX]: shared out-lists n, entries of in[exit], then n x (vertex, entries of its out-list)
    uint32_t *seg_done;            // [tile_list_cap][DG_SH_MAX + 1]: slots of its stretches a worklist entry has used (last: in[exit])
    uint32_t sh_log;               // slots a segment has behind either shared list
    uint32_t *wl_first;            // [T]: worklist index of the target's first segment
    // ---- LDS tiles (k_cutmap, k_merge_tile, k_merge_list) ----
    uint32_t *nextcut;             // [bbv_base + p]: smallest cut position >= p (k_cutmap)
    uint32_t tile_pos;             // backbone positions per tile (0: tiles are not used)
    uint32_t tile_words;           // LDS words of a tile's image
    uint32_t *tile_list;           // [0] entries, [1] tiles that gave up, then (target, first, last) triples
    uint32_t tile_list_cap;
    // ---- outputs ----
    uint8_t *cns;
    uint64_t cns_cap;
    uint64_t *cns_off;
    uint32_t *cns_len;
    uint64_t *seg_first;
    uint32_t *n_seg;
    int32_t *seg_r0, *seg_r1;
    uint64_t seg_cap;
    DgStatus *st;
};

// slots a backbone vertex gets for each of its two lists before it has to move to the
// growth region (K = alignments of the target)
__host__ __device__ inline uint32_t dg_capb(uint32_t k) { return k + 2u < 12u ? k + 2u : 12u; }
