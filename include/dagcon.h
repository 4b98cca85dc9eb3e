/*
 * dagcon.h -- C ABI of the MI355X-native DAGCon consensus engine.
 *
 * This is the drop-in boundary for the hot path of verdurin/pbdagcon.  The
 * reference has no FFI: its seam is the body of the Consensus worker,
 *
 *     src/cpp/main.cpp:130-138   (pbdagcon)     src/cpp/dazcon.cpp:76-89 (dazcon)
 *
 *         AlnGraphBoost ag(tlen | backbone);
 *         for each alignment: if (|qstr| < minLen) continue;
 *                             normalizeGaps; trimAln(trim); ag.addAln;
 *         ag.mergeNodes();
 *         ag.consensus(seqs, minWeight, minLen);
 *
 * One call of dagcon_consensus() replaces that sequence for a whole batch of
 * independent targets.  Everything crosses the boundary as plain pointers and
 * sizes (structure-of-arrays blobs); no C++ or torch types.  INTEGRATION.md
 * shows the binding a maintainer of the reference would add.
 *
 * All functions return DAGCON_OK (0) or a negative dagcon_status; no
 * exceptions cross the boundary.  A context is single-owner: one host thread
 * drives it; work inside is asynchronous on the context's HIP stream.
 */
#ifndef DAGCON_H
#define DAGCON_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAGCON_ABI_VERSION 2   /* 2: per-target status in dagcon_results, dagcon_host_alloc/free */

typedef enum dagcon_status {
    DAGCON_OK = 0,
    DAGCON_ERR_INVALID_ARG = -1,    /* NULL pointer, inconsistent sizes */
    DAGCON_ERR_NO_DEVICE = -2,      /* HIP runtime / gfx950 device unavailable: never falls back to CPU */
    DAGCON_ERR_HIP = -3,            /* a HIP call failed, see dagcon_last_error */
    DAGCON_ERR_NONCONFORMING = -4,  /* an alignment would drive the reference into undefined
                                       behaviour (AlnGraphBoost.cpp:64-107: start < 1, target bases
                                       past tlen, bytes outside printable ASCII) */
    DAGCON_ERR_UNSUPPORTED = -5,    /* more alignments per target than DAGCON_MAX_COVERAGE, a target too
                                       large for 25-bit vertex ids, an undefined bit in dagcon_opts.flags */
    DAGCON_ERR_WORKSPACE = -6,      /* device workspace could not be grown */
    DAGCON_ERR_INTERNAL = -7,       /* device-side invariant violated */
    DAGCON_ERR_STATE = -8           /* call sequence error (run before upload, ...) */
} dagcon_status;

#define DAGCON_MAX_COVERAGE 4094u   /* alignments per target that pass the min_len filter */

/* dagcon_opts.flags */
#define DAGCON_FLAG_RAW_ALIGNMENTS 1u /* skip normalizeGaps/trimAln: feed strings to addAln as they
                                         are (what test/cpp/AlnGraphBoostTest.cpp:11-47 does) */
#define DAGCON_FLAG_STOP_AFTER_BUILD 2u /* debug: stop after addAln (no merge, no consensus) so that
                                           dagcon_debug_graph shows the graph before mergeNodes */
#define DAGCON_FLAG_STOP_AFTER_MERGE 4u /* debug: stop after mergeNodes */
#define DAGCON_FLAG_DEBUG_RESWEEP 16u   /* debug: treat the segmented bestPath sweep as inexact, so that
                                           every target takes the one-piece re-sweep (tests only) */
#define DAGCON_FLAGS_ALL (DAGCON_FLAG_RAW_ALIGNMENTS | DAGCON_FLAG_STOP_AFTER_BUILD | \
                          DAGCON_FLAG_STOP_AFTER_MERGE | DAGCON_FLAG_DEBUG_RESWEEP)
                                        /* dagcon_create refuses any other bit (DAGCON_ERR_UNSUPPORTED) */

/* Mirrors ProgramOpts (src/cpp/ProgramOpts.hpp:8-36) for this path. */
typedef struct dagcon_opts {
    uint32_t min_cov;    /* -c: targets with fewer alignments are skipped (main.cpp:66-72,118) */
    uint32_t min_len;    /* -m: alignment pre-filter on raw |qstr| (main.cpp:132) and minimum
                            emitted segment length (AlnGraphBoost.cpp:359,371) */
    uint32_t trim;       /* -t: trimAln length (main.cpp:134) */
    int32_t  min_weight; /* consensus minWeight; <0 means "= min_cov" as main.cpp:261,279 does */
    int32_t  device;     /* HIP device ordinal */
    uint32_t flags;
    uint32_t max_segments; /* workers per target for mergeNodes (bestPath: three times as many): a
                              target is swept in up to this many pieces, split at backbone vertices
                              every read passes through (the result does not depend on it).
                              0 = automatic, 1 = one sequential sweep per target, at most 64 */
    uint32_t min_segment_len; /* shortest backbone stretch given a worker of its own; 0 = default (768; down to 192 for small batches) */
} dagcon_opts;

/* Defaults of pbdagcon (main.cpp:181-211): -c 6 -m 500 -t 50. */
void dagcon_default_opts(dagcon_opts *o);

/*
 * A batch of independent targets.  Alignment a of target t is
 * a in [aln_begin[t], aln_begin[t+1]); its strings are qstr[aln_off[a] ..
 * aln_off[a]+aln_len[a]) and the same range of tstr (equal lengths, as
 * Alignment.hpp:35-37 requires).  aln_start is 1-based (Alignment.hpp:21-22).
 * Order of alignments inside a target is the order the reference would call
 * addAln in; it is semantics (adjacency order).
 */
typedef struct dagcon_batch {
    uint32_t n_targets;
    const uint32_t *tlen;        /* [n_targets] backbone length (Alignment::tlen) */
    const uint64_t *aln_begin;   /* [n_targets+1] */
    const uint32_t *aln_start;   /* [n_alns] */
    const uint64_t *aln_off;     /* [n_alns] byte offset into qstr / tstr */
    const uint32_t *aln_len;     /* [n_alns] */
    const char *qstr;            /* query blob  */
    const char *tstr;            /* target blob */
    uint64_t blob_bytes;         /* size of each blob */
    const char *backbone;        /* optional: real backbone bases (dazcon.cpp:76); NULL = 'N'
                                    backbone filled in by the reads (main.cpp:130) */
    const uint64_t *backbone_off;/* [n_targets] offsets into backbone when it is given */
} dagcon_batch;

/*
 * Consensus segments, CnsResult (AlnGraphBoost.hpp:63-66) for every target.
 * Segments of target t are s in [seg_begin[t], seg_begin[t+1]).  range0/range1
 * index the target's consensus string (quirk Q5), seq is seq_blob[seq_off[s] ..
 * +seq_len[s]).  The arrays are owned by the context and stay valid until the
 * next dagcon_upload / dagcon_consensus / dagcon_destroy on it.
 */
typedef struct dagcon_results {
    uint32_t n_targets;
    uint64_t n_segments;
    const uint64_t *seg_begin;   /* [n_targets+1] */
    const int32_t *range0;       /* [n_segments] */
    const int32_t *range1;       /* [n_segments] */
    const uint64_t *seq_off;     /* [n_segments] */
    const uint32_t *seq_len;     /* [n_segments] */
    const char *seq_blob;
    uint64_t seq_bytes;
    /* ABI 2.  A failure is confined to its target, as in the reference, where an assert or the
     * undefined behaviour behind it hits one worker's one target (AlnGraphBoost.cpp:71-72):
     * target_status[t] is DAGCON_OK or the dagcon_status that target alone failed with
     * (DAGCON_ERR_NONCONFORMING, DAGCON_ERR_UNSUPPORTED: too large, DAGCON_ERR_INTERNAL); a failed
     * target has no segments, every other target of the batch is complete and exact.  The call
     * that filled the struct still returns DAGCON_OK; dagcon_last_error describes the first failure. */
    const int32_t *target_status; /* [n_targets] */
    uint32_t n_failed;            /* targets whose status is not DAGCON_OK */
} dagcon_results;

/* Per-stage device time of the last dagcon_run, from HIP events on the
 * context's stream (milliseconds), plus the algorithmic byte count SURVEY.md
 * section 8(d) defines. */
typedef struct dagcon_timings {
    float ms_total;          /* first kernel start -> last kernel end */
    float ms_normalize;      /* stage a1: count + normalizeGaps + trimAln */
    float ms_build;          /* stage a2: carve + init + emit + adjacency lists */
    float ms_merge;          /* stage b : mergeNodes */
    float ms_bestpath;       /* stage c : bestPath + consensus segmentation */
    uint64_t algorithmic_bytes; /* sum(|q|+|t|) over alignments passing min_len + output bases */
    uint64_t consensus_bases;   /* sum of emitted segment lengths */
    uint64_t n_alignments;      /* alignments that passed min_len and reached the device */
    uint64_t n_columns;         /* normalised, trimmed columns threaded into graphs */
    uint64_t n_nodes;           /* graph vertices before merging, all targets */
    uint32_t reruns;            /* times the batch was re-run after growing the workspace */
    uint32_t merge_segments;    /* pieces the targets of the batch were swept in, all targets */
} dagcon_timings;

typedef struct dagcon_ctx dagcon_ctx;

int  dagcon_abi_version(void);
int  dagcon_create(const dagcon_opts *opts, dagcon_ctx **out);
void dagcon_destroy(dagcon_ctx *ctx);
const char *dagcon_last_error(const dagcon_ctx *ctx);

/* The drop-in call: replaces main.cpp:130-138 for every target of the batch.
 * Equivalent to dagcon_upload + dagcon_run + dagcon_fetch. */
int dagcon_consensus(dagcon_ctx *ctx, const dagcon_batch *batch, dagcon_results *results);

/* The same work in three steps, so that a caller can keep inputs resident in
 * HBM and time the device path alone (bench.py does). */
int dagcon_upload(dagcon_ctx *ctx, const dagcon_batch *batch); /* host filter + H2D, synchronous */
int dagcon_run(dagcon_ctx *ctx);                               /* enqueue all kernels, asynchronous */
int dagcon_sync(dagcon_ctx *ctx);                              /* wait for the stream */
int dagcon_fetch(dagcon_ctx *ctx, dagcon_results *results);    /* sync + status check + D2H */
int dagcon_get_timings(dagcon_ctx *ctx, dagcon_timings *out);  /* after dagcon_sync / dagcon_fetch */

/*
 * Page-locked host memory for the input blobs (qstr / tstr / backbone): a caller that parses
 * alignment records straight into such a buffer gets its dagcon_upload at link speed instead of
 * through the driver's staging copies.  Optional: any host memory is accepted by dagcon_upload.
 */
int  dagcon_host_alloc(dagcon_ctx *ctx, size_t bytes, void **out);
void dagcon_host_free(dagcon_ctx *ctx, void *p);

/*
 * Unit-level entry point for stage a1 alone: normalizeGaps (Alignment.cpp:
 * 131-217) followed by trimAln (Alignment.cpp:219-242) on n alignments, on
 * the device.  Inputs use the batch blob layout; outputs are written into
 * qout/tout at out_off[a] (capacity 2*aln_len[a] each), with out_len[a] and
 * the trimmed start in out_start[a].  trim = 0 gives normalizeGaps alone;
 * flags = DAGCON_FLAG_RAW_ALIGNMENTS gives trimAln alone (strings as given).
 */
int dagcon_normalize(dagcon_ctx *ctx, uint32_t n, const uint32_t *aln_start,
                     const uint64_t *aln_off, const uint32_t *aln_len,
                     const char *qstr, const char *tstr, uint64_t blob_bytes,
                     uint32_t trim, uint32_t flags, const uint64_t *out_off, char *qout, char *tout,
                     uint32_t *out_len, uint32_t *out_start);

/*
 * The `-a` stage (main.cpp:127-128, SimpleAligner.cpp:25-63): re-aligns n (query, target) pairs of
 * UNALIGNED sequences, as the .pre format carries them (Alignment.cpp:82-112), on the device.
 * Pair a is q_blob[q_off[a] .. +q_len[a]) against t_blob[t_off[a] .. +t_len[a]).  The aligned strings
 * (equal lengths, '-' for gaps) are written to qaln / taln at out_off[a] (room for q_len[a] + t_len[a]
 * columns each), their length to aln_len[a].  The alignment is global: in the terms of
 * SimpleAligner.cpp:52-53 GenomicTBegin() = 0 and GenomicTEnd() = t_len[a]; the caller applies
 * SimpleAligner.cpp:51-62 (start / end / reverse complement) itself.
 * blasr_libcpp is not in the reference tree: this stage is pinned to the reference only by its one
 * known-answer test (test/cpp/SimpleAlignerTest.cpp:8-21); everything else is parity-unpinned.
 */
int dagcon_align(dagcon_ctx *ctx, uint32_t n, const uint64_t *q_off, const uint32_t *q_len,
                 const uint64_t *t_off, const uint32_t *t_len, const char *q_blob, uint64_t q_bytes,
                 const char *t_blob, uint64_t t_bytes, const uint64_t *out_off, char *qaln, char *taln,
                 uint32_t *aln_len);

/*
 * What main.cpp:117-145 does with -a, in one call: the records of a batch of targets as the .pre format carries
 * them (Alignment.cpp:82-112: tstart is Alignment::start as parsePre leaves it, q / t the unaligned sequences) are
 * re-aligned as by dagcon_align, start / end / strand handled as SimpleAligner.cpp:51-62 does (start = tstart,
 * end = start + t_len; '-': start = tlen - end and both strings reverse-complemented; start += 1), then
 * filtered, normalised, trimmed and threaded as by dagcon_consensus.  The aligned strings stay on the device.
 * Same parity statement as dagcon_align.
 */
typedef struct dagcon_pre_batch {
    uint32_t n_targets;
    const uint32_t *tlen;        /* [n_targets] */
    const uint64_t *rec_begin;   /* [n_targets + 1]: records of target g are rec_begin[g] .. rec_begin[g + 1] - 1 */
    const uint32_t *tstart;      /* [n_rec] */
    const char *strand;          /* [n_rec] '+' or '-' */
    const uint64_t *q_off;       /* [n_rec] query sequence: q_blob[q_off .. + q_len) */
    const uint32_t *q_len;
    const uint64_t *t_off;       /* [n_rec] target sequence (in the read's orientation): t_blob[t_off .. + t_len) */
    const uint32_t *t_len;
    const char *q_blob;
    uint64_t q_bytes;
    const char *t_blob;
    uint64_t t_bytes;
} dagcon_pre_batch;
int dagcon_consensus_pre(dagcon_ctx *ctx, const dagcon_pre_batch *batch, dagcon_results *results);

/*
 * Debug / parity aid: adjacency of one target's graph as left by the last
 * dagcon_run (after mergeNodes), in list order.  Vertex ids are in backbone
 * position order: the inserted vertices whose _bbMap is p (in read, column
 * order), then backbone vertex p, for p = 0 .. tlen+1.  Node arrays have n_nodes
 * entries; out lists are CSR (out_begin[n_nodes+1], out_dst, out_count), in
 * lists likewise.  Buffers are owned by the context (valid until next call).
 */
typedef struct dagcon_graph_dump {
    uint32_t n_nodes;
    const uint8_t *base;
    const int32_t *weight;
    const int32_t *coverage;     /* meaningful for backbone vertices */
    const uint8_t *deleted;
    const uint8_t *backbone;     /* 1 for enter / backbone / exit vertices */
    const int32_t *bbpos;        /* backbone position (backbone vertices), _bbMap (inserted vertices) */
    const uint32_t *out_begin;
    const int32_t *out_dst;
    const int32_t *out_count;
    const uint32_t *in_begin;
    const int32_t *in_src;
} dagcon_graph_dump;
int dagcon_debug_graph(dagcon_ctx *ctx, uint32_t target, dagcon_graph_dump *out);

/* Diagnostic builds (-DDG_STAMPS) only: sixteen raw device counters of the last
 * run (in-kernel cycle stamps of target 0); all zero in the shipped build. */
int dagcon_debug_counters(dagcon_ctx *ctx, unsigned long long *out8);

/* Records of the last dagcon_align / dagcon_consensus_pre on this context whose corners the widest band could not
 * connect (sequences of very different lengths, indels of hundreds of bases): their alignment has length 0 and the
 * min_len filter then drops them, where the reference's SDPAlign + GuidedAlign (SimpleAligner.cpp:35-48) always
 * returns something.  The calls succeed; a caller that cares asks here (the pbdagcon host warns on stderr). */
uint32_t dagcon_align_dropped(dagcon_ctx *ctx);

/* Host arithmetic only (no device, no context): the pieces dagcon_upload would cut the merge and bestPath
 * sweeps of a batch of this shape into -- out4 = {pieces per target (merge), shortest stretch worth a
 * piece, 1 when the four-segments-per-wave merge kernel takes the batch, pieces per target (bestPath)}.
 * sum_positions = sum over active targets of tlen + 2 rounded up to a multiple of 4.  Every launch grid
 * derived from these is non-empty: pieces >= 1 for every input (tests/test_abi.py sweeps it). */
int dagcon_debug_plan(uint32_t n_targets, uint64_t n_alignments, uint64_t sum_positions, uint32_t partial_span,
                      uint32_t max_segments, uint32_t min_segment_len, uint32_t out4[4]);

#ifdef __cplusplus
}
#endif
#endif /* DAGCON_H */
