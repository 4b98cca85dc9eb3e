// dagcon_dev.h -- device-side data layout shared by the kernels and the host API.
//
// One batch = T independent targets.  Everything lives in a handful of HBM
// arenas owned by the context; per-target and per-alignment arrays hold
// offsets into them.  Vertex ids are target-local int32:
//     0            enter '^'          (AlnGraphBoost.cpp:26-28)
//     1 .. blen    backbone
//     blen+1       exit '$'
//     blen+2 ..    inserted vertices, numbered per alignment in column order
// Vertex ids are only names: what the reference's results depend on is the
// ORDER of each vertex's out- and in-list, which is
// kept exactly.
#pragma once
#include <stdint.h>

#define DG_GAP '-'

// DgParams::flags: bits 0..2 are dagcon_opts.flags; internal bits from 8 up
#define DG_F_RAW      1u
#define DG_F_A1_ONLY  8u   // dagcon_normalize: no graph follows, skip the backbone conformity check

// status bits (Status::err_flags)
#define DG_E_BADCHAR     0x001u  // byte outside printable ASCII in an alignment string
#define DG_E_NONCONF     0x002u  // alignment leaves the backbone (start<1 or runs past tlen)
#define DG_E_NORM_OVF    0x004u  // normalised-column arena too small (rerun with norm_need)
#define DG_E_NODE_OVF    0x008u  // vertex arena too small (rerun with node_need)
#define DG_E_POOL_OVF    0x010u  // adjacency pool arena too small (rerun with pool_need)
#define DG_E_POOL_TGT    0x020u  // one target outgrew its pool share (rerun with larger growth factor)
#define DG_E_STACK       0x040u  // mergeInNodes recursion scratch exhausted
#define DG_E_INTERNAL    0x080u  // invariant violated (empty list dereference, list > 65535)
#define DG_E_OUT_OVF     0x100u  // output arena too small

struct DgStatus {
    uint32_t err_flags;
    uint32_t bad_aln;
    uint32_t bad_target;
    uint32_t pad;
    unsigned long long norm_top;   // bump cursor into the column arena (uint16 units)
    unsigned long long node_need;  // exact vertex count of the batch (set by carve)
    unsigned long long pool_need;  // exact pool words of the batch (set by carve)
    unsigned long long cns_top;    // bump cursor into the consensus blob
    unsigned long long seg_top;    // bump cursor into the segment arrays
    unsigned long long n_columns;  // normalised, trimmed columns
};

// Per-vertex record gathered when a vertex is looked at as somebody's neighbour.
struct DgHot {
    uint16_t out_len, in_len;
    uint8_t base, flags;
    uint16_t pad;
};
#define DG_NF_BACKBONE 1u
#define DG_NF_DELETED  2u

// Where a vertex's own adjacency lists live in the target's pool.
// out entry i: pool[out_off + 2i] = dst, pool[out_off + 2i + 1] = count
// in  entry i: pool[in_off + i]   = src          (edge counts live on the out side only)
struct DgLists {
    uint32_t out_off, in_off;
    uint16_t out_cap, in_cap;
    uint32_t pad;
};

struct DgParams {
    // ---- inputs (resident in HBM after dagcon_upload) ----
    const uint8_t *q, *t;
    const uint64_t *aln_off;
    const uint32_t *aln_len, *aln_start, *aln_tgt;
    const uint32_t *tlen;
    const uint64_t *aln_begin;     // [T+1], indices into the (filtered) alignment arrays
    const uint8_t *tactive;        // [T] 1 = build a graph (main.cpp:66-72,118)
    const uint8_t *bb;             // optional backbone blob
    const uint64_t *bb_off;
    const uint64_t *mat_base;      // [T] offset into matA/matD: (tlen+2) * K cells
    const uint64_t *bbv_base;      // [T] offset into cov/bvote: tlen+2 cells
    uint32_t T, A;
    uint32_t trim, min_len;
    int32_t min_weight;
    uint32_t flags;
    uint32_t max_k, max_tlen;
    // ---- per alignment work arrays ----
    uint32_t *nmis;
    uint64_t *norm_off;
    uint32_t *n_lo, *n_hi, *n_start, *n_ins, *n_del, *ins_base;
    uint16_t *norm;                // column arena: low byte q, high byte t
    uint64_t norm_cap;
    // ---- per target work arrays ----
    uint64_t *node_base;
    uint32_t *n_nodes;
    uint64_t *pool_base;
    uint32_t *pool_size, *pool_top;
    // ---- arenas ----
    uint32_t *matA, *matD;         // arrival / departure matrices [vertex][read]
    int32_t *cov;                  // backbone coverage
    uint32_t *bvote;               // (read+1)<<8 | base : last writer wins (AlnGraphBoost.cpp:79,90)
    DgHot *hot;
    DgLists *lists;
    int32_t *weight, *bbpos, *pending, *best, *queue;
    float *score;
    uint8_t *cns_tmp;
    uint64_t node_cap;
    uint32_t *pool;
    uint64_t pool_cap;
    int32_t *stk;                  // per-target scratch, stk_words each
    uint32_t stk_words;
    uint32_t growth_pct;           // pool growth region as % of the initial adjacency words
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
