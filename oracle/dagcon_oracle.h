/*
 * dagcon_oracle.h -- CPU restatement of the pbdagcon consensus hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker.  The product path (pbdagcon_amd/) never
 * links, imports or falls back to this code.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Parity pinning:
 *   - og_normalize_gaps / og_trim_aln / og_parse_m5 are checked against the
 *     reference's own Alignment.cpp compiled in place (oracle/_ref, see
 *     oracle/Makefile) and against test/cpp/AlignmentTest.cpp vectors.
 *   - the graph part (AlnGraphBoost.cpp) needs Boost.Graph, which is absent
 *     from this image, so it cannot be built here; it is pinned by the
 *     reference's own known-answer tests (test/cpp/AlnGraphBoostTest.cpp:11-57)
 *     and cross-checked against an independent literal Python model
 *     (oracle/pymodel.py).
 */
#ifndef DAGCON_ORACLE_H
#define DAGCON_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Alignment.cpp ---------------------------------------------------- */

/* Alignment.cpp:15-26 revComp.  In place; only upper-case ACGT complemented. */
void og_revcomp(char *seq, size_t len);

/* Alignment.cpp:131-217 normalizeGaps.  qout/tout need capacity 2*len+1.
 * Returns the normalised length (NUL terminated outputs). */
size_t og_normalize_gaps(const char *q, const char *t, size_t len, int push,
                         char *qout, char *tout);

/* Alignment.cpp:219-242 trimAln.  Returns loffs/roffs of the kept window and
 * the number of target bases dropped on the left (added to start). */
void og_trim_aln(const char *t, size_t len, int trim_len,
                 size_t *loffs, size_t *roffs, uint32_t *lbases);

/* Alignment.cpp:44-80 parseM5 (groupByTarget selectable).  Output strings are
 * malloc'd; caller frees with og_free_parsed.  Returns 0 on an empty line
 * (record left untouched, as the reference does), 1 otherwise. */
typedef struct og_parsed {
    char *id, *sid, *qstr, *tstr;
    uint32_t tlen, start;
    char strand;
} og_parsed;
int og_parse_m5(const char *line, size_t len, int group_by_target, og_parsed *out);
void og_free_parsed(og_parsed *p);

/* Alignment.cpp:82-112 parsePre: "qid tid strand tlen tstart tend qseq tseq" (the two sequences
 * are UNALIGNED substrings; main.cpp:243-246 selects this parser together with -a).  `end` goes to
 * *end_out.  Returns 0 on an empty line, -1 on fewer than 8 fields (the reference indexes
 * fields[7]: undefined behaviour), 1 otherwise. */
int og_parse_pre(const char *line, size_t len, og_parsed *out, uint32_t *end_out);

/* ---- the -a re-aligner (SimpleAligner.cpp:25-63) ------------------------
 * blasr_libcpp (SDPAlign + GuidedAlign) is not in the tree: PARITY UNPINNED except for the
 * reference's one known-answer test (test/cpp/SimpleAlignerTest.cpp:8-21), which this restatement
 * reproduces: a global alignment that minimises blasr's distance score (SimpleAligner.cpp:10-23:
 * match -5, mismatch +6 from SMRTDistanceMatrix, insertion 4, deletion 5), ties resolved
 * diagonal first, then insertion (gap in the target), then deletion, inside a band of half-width
 * og_align_halfwidth(qlen, tlen) around the length-scaled diagonal j = i * tlen / qlen.  Narrower bands are tried
 * first and stand when the path keeps 8 cells away from their edges: a band of 56 cells to either side of a centre
 * that follows the best cell of the row before (pairs long enough for the static bands to be wider than that), then
 * the static band of og_align_halfwidth_first(), then the full one.
 * Outputs (capacity qlen + tlen + 1 each) get the aligned strings; returns their length. */
uint32_t og_align_halfwidth(uint32_t qlen, uint32_t tlen);
uint32_t og_align_halfwidth_first(uint32_t qlen, uint32_t tlen);   /* the band tried first (see og_banded_align) */
size_t og_banded_align(const char *q, uint32_t qlen, const char *t, uint32_t tlen, char *qaln, char *taln);
/* SimpleAligner.cpp:51-62: what align() does to start / end / strings once the aligner has
 * produced (queryStr, targetStr, GenomicTBegin = 0, GenomicTEnd = tlen of the record's tstr).
 * qaln / taln are rewritten in place (reverse-complemented for the '-' strand). */
void og_simple_aligner_finish(char *qaln, char *taln, size_t n, uint32_t tseq_len, uint32_t tlen,
                              char strand, uint32_t *start, uint32_t *end);

/* ---- AlnGraphBoost.cpp ------------------------------------------------ */

typedef struct og_graph og_graph;

/* AlnGraphBoost.cpp:16-39 (backbone string) / :41-62 (length only, 'N'). */
og_graph *og_graph_new_seq(const char *backbone, size_t blen);
og_graph *og_graph_new_len(size_t blen);
void og_graph_free(og_graph *g);

/* AlnGraphBoost.cpp:64-107 addAln (+ :109-127 addEdge).  Literal: no
 * validation, exactly the three column branches of the reference. */
void og_add_aln(og_graph *g, uint32_t start, const char *q, const char *t, size_t len);

/* AlnGraphBoost.cpp:129-273 mergeNodes/mergeInNodes/mergeOutNodes/markForReaper.
 * Returns 0, or -1 if a state the reference would treat as undefined
 * behaviour was hit (empty edge list dereference). */
int og_merge_nodes(og_graph *g);

/* AlnGraphBoost.cpp:375-459 bestPath.  Writes node ids of the path
 * (including enter/exit); returns path length.  *path is malloc'd. */
size_t og_best_path(og_graph *g, int32_t **path);

/* AlnGraphBoost.cpp:285-325 consensus(int minWeight): longest run.  malloc'd. */
char *og_consensus_longest(og_graph *g, int min_weight);

typedef struct og_segment { int32_t range0, range1; char *seq; } og_segment;
/* AlnGraphBoost.cpp:327-373 consensus(vector<CnsResult>&, minWeight, minLen).
 * Returns the number of segments; *segs is malloc'd (free with og_free_segments). */
size_t og_consensus_all(og_graph *g, int min_weight, size_t min_len, og_segment **segs);
void og_free_segments(og_segment *segs, size_t n);

/* AlnGraphBoost.cpp:468-487 danglingNodes. */
int og_dangling_nodes(og_graph *g);

/* Introspection for tests (live graph statistics / adjacency dump). */
size_t og_num_nodes(const og_graph *g);
size_t og_num_live_nodes(const og_graph *g);
size_t og_num_live_edges(const og_graph *g);
/* Per node: base, weight, coverage, deleted, backbone, bbmap. */
void og_node_info(const og_graph *g, size_t v, char *base, int *weight, int *coverage,
                  int *deleted, int *backbone, int64_t *bbmap);
/* Out/in adjacency in list order.  Returns degree; fills up to cap entries. */
size_t og_out_edges(const og_graph *g, size_t v, int32_t *dst, int32_t *count, size_t cap);
size_t og_in_edges(const og_graph *g, size_t v, int32_t *src, int32_t *count, size_t cap);

/* ---- main.cpp:117-148: one target, end to end -------------------------- */

typedef struct og_opts {
    uint32_t min_len;     /* -m, main.cpp:132,138 */
    uint32_t trim;        /* -t, main.cpp:134 */
    int32_t  min_weight;  /* = -c, main.cpp:261,279 (quirk Q1) */
} og_opts;

/* Replays main.cpp:130-138 for one target.  backbone may be NULL ('N'
 * backbone, pbdagcon) or tlen chars (dazcon.cpp:76).  Alignments that would
 * drive the reference into undefined behaviour (start<1, target bases running
 * past tlen, |q|!=|t| is the caller's problem) make it return -2 with
 * *bad_aln set.  Returns number of segments (>=0) on success. */
long og_consensus_target(uint32_t tlen, const char *backbone, size_t n_alns,
                         const uint32_t *starts, const char *const *qstrs,
                         const char *const *tstrs, const size_t *lens,
                         const og_opts *opts, og_segment **segs, long *bad_aln);

/* Same, reading alignments from flat blobs (the C-ABI batch layout) so that
 * bench.py can time the port without per-string Python overhead. */
long og_consensus_target_blob(uint32_t tlen, const char *backbone, size_t n_alns,
                              const uint32_t *starts, const uint64_t *offs,
                              const uint32_t *lens, const char *qblob,
                              const char *tblob, const og_opts *opts,
                              og_segment **segs, long *bad_aln);

#ifdef __cplusplus
}
#endif
#endif
