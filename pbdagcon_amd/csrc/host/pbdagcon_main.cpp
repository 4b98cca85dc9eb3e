// pbdagcon_main.cpp -- `pbdagcon`-compatible command line on top of the C ABI.
//
// Replaces the Reader -> N x Consensus -> Writer thread pipeline of the reference
// (src/cpp/main.cpp:37-176, 227-288) with: parse the BLASR -m 5 stream, group
// consecutive records by target id (BlasrM5AlnProvider.cpp:34-55), hand batches of
// independent targets to dagcon_consensus() (include/dagcon.h), print FASTA records in
// input order (main.cpp:141-143).  Same flags and defaults as main.cpp:178-225.
//
// The parser thread fills one batch while a second thread has the previous one on the GPU
// (context creation included, so HIP start-up hides behind the first batch's parsing).
//
// Differences that are deliberate and documented in DESIGN.md:
//   * output order is input order (the reference's is nondeterministic for -j >= 2, Q3);
//   * -j is the number of host threads that index the text and copy the strings (the consensus
//     itself is the GPU's), -j 1 does not deadlock (Q2);
//   * -a (.pre input, every record re-aligned first: main.cpp:127-128, 243-246) runs this build's own
//     banded aligner on the GPU (dagcon_align): blasr_libcpp is absent, the stage is pinned to the
//     reference by its one known-answer test only (test/cpp/SimpleAlignerTest.cpp:8-21);
//   * blank lines are skipped (the reference duplicates the previous record, Q9);
//   * a missing input file is an error on stderr, exit 1 (the reference is silent, Q11).
#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../../include/dagcon.h"

namespace {

struct Opts {
    unsigned threads = 4, min_cov = 6, min_len = 500, trim = 50;
    bool align = false, verbose = false, dump = false;
    std::vector<int> devices{0};       // --devices: one consensus worker (thread + context) per GPU
    int pinned = -1;                   // --pinned 0|1: page-locked blobs (-1: when the input is several batches long)
    unsigned contexts = 0;             // --contexts N: consensus workers per GPU (0: two when the input is several batches long)
    unsigned polish = 0;               // --polish N (with -a): N more rounds with the consensus as the new backbone
    size_t batch_targets = 512;        // parsing and the GPU still overlap on mid-size inputs; 256 left the GPU a third less efficient (4,000 targets: 0.79 -> 0.67 s)
    size_t batch_bytes = 512ull << 20;  // (of strings: 80 targets of 50 kb x 60x fill the chip; what a context holds, and has to
                                       // allocate on its first batch, grows with it)
    size_t slab_bytes = 0;             // test hook: text indexed per round (0 = automatic)
    std::string input;
};

void usage(FILE *f) {
    fprintf(f,
            "USAGE: pbdagcon [-j <int>] [-c <uint>] [-m <uint>] [-t <uint>] [-a] [-v] <input>\n"
            "  PBDAGCON is a tool that implements DAGCon (Directed Acyclic Graph Consensus); this build\n"
            "  runs the consensus on an MI355X through libdagcon_hip.so.\n"
            "  -j, --threads       host threads for parsing (default 4); the consensus runs on the GPU\n"
            "  -c, --min-coverage  minimum alignments per target, also the minimum node weight (default 6)\n"
            "  -m, --min-length    minimum alignment / consensus length (default 500)\n"
            "  -t, --trim          trim alignments on either side (default 50)\n"
            "  -a, --align         input is .pre (qid tid strand tlen tstart tend qseq tseq): align the sequences first\n"
            "                      (this build's own banded GLOBAL aligner on the GPU; the reference's blasr SDPAlign(Local) +\n"
            "                      GuidedAlign is not in its tree: parity unpinned beyond its one SimpleAligner known-answer test)\n"
            "  -v, --verbose       per-target progress on stderr\n"
            "  --polish N          with -a: N more rounds, each with the previous round's consensus as the backbone the reads\n"
            "                      are re-aligned to (README.md:14-15 of the reference: 'the new consensus can be used as a new\n"
            "                      backbone sequence to iteratively improve the consensus quality')\n"
            "  --devices LIST      GPUs to use, e.g. 0,1,2,3 (default 0): one consensus worker per GPU, batches of\n"
            "                      targets dealt round-robin, records still printed in input order\n"
            "  --contexts N        consensus workers (thread + context) per GPU, 1..4: a batch's upload and formatting run\n"
            "                      beside another batch's kernels (default: 2 for inputs of several batches, else 1)\n"
            "  <input>             BLASR -m 5 file (.pre with -a) sorted by target, or - for stdin\n"
            "  version 0.3 (dagcon-mi355x)\n");
}

bool parse_uint(const char *s, unsigned *out) {
    char *e = nullptr;
    errno = 0;
    unsigned long v = strtoul(s, &e, 10);
    if (errno || !e || *e || v > 0xFFFFFFFFul) return false;
    *out = (unsigned)v;
    return true;
}

int parse_args(int argc, char **argv, Opts &o) {
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](unsigned *dst) {
            if (i + 1 >= argc || !parse_uint(argv[i + 1], dst)) { fprintf(stderr, "PARSE ERROR: %s needs an unsigned integer\n", a.c_str()); return false; }
            i++;
            return true;
        };
        if (a == "-j" || a == "--threads") { if (!need(&o.threads)) return 2; }
        else if (a == "-c" || a == "--min-coverage") { if (!need(&o.min_cov)) return 2; }
        else if (a == "-m" || a == "--min-length") { if (!need(&o.min_len)) return 2; }
        else if (a == "-t" || a == "--trim") { if (!need(&o.trim)) return 2; }
        else if (a == "-a" || a == "--align") o.align = true;
        else if (a == "-v" || a == "--verbose") o.verbose = true;
        else if (a == "--dump-parsed") o.dump = true;            // test hook: parser only, no GPU
        else if (a == "--slab-bytes") { unsigned v = 0; if (!need(&v)) return 2; o.slab_bytes = v; }   // test hook
        else if (a == "--batch-targets") { unsigned v = 0; if (!need(&v) || !v) return 2; o.batch_targets = v; }
        else if (a == "--polish") { if (!need(&o.polish)) return 2; }
        else if (a == "--contexts") { if (!need(&o.contexts) || !o.contexts || o.contexts > 4) { fprintf(stderr, "PARSE ERROR: --contexts takes 1..4\n"); return 2; } }
        else if (a == "--pinned") { unsigned v = 0; if (!need(&v)) return 2; o.pinned = v ? 1 : 0; }
        else if (a == "--devices") {
            if (i + 1 >= argc) { fprintf(stderr, "PARSE ERROR: --devices needs a list such as 0,1,2\n"); return 2; }
            o.devices.clear();
            const char *p = argv[++i];
            while (*p) {
                char *e = nullptr;
                const long v = strtol(p, &e, 10);
                if (e == p || v < 0 || v > 1023) { fprintf(stderr, "PARSE ERROR: bad --devices list\n"); return 2; }
                o.devices.push_back((int)v);
                p = *e == ',' ? e + 1 : e;
                if (*e && *e != ',') { fprintf(stderr, "PARSE ERROR: bad --devices list\n"); return 2; }
            }
            if (o.devices.empty()) { fprintf(stderr, "PARSE ERROR: --devices list is empty\n"); return 2; }
        }
        else if (a == "-h" || a == "--help") { usage(stdout); exit(0); }
        else if (a == "--version") { printf("pbdagcon  version: 0.3\n"); exit(0); }
        else if (a == "-" || a[0] != '-') {
            if (!o.input.empty()) { fprintf(stderr, "PARSE ERROR: more than one input\n"); return 2; }
            o.input = a;
        } else { fprintf(stderr, "PARSE ERROR: unknown argument %s\n", a.c_str()); return 2; }
    }
    if (o.input.empty()) { fprintf(stderr, "PARSE ERROR: required argument missing: input\n"); usage(stderr); return 2; }
    return 0;
}

// Alignment.cpp:15-26: only upper-case ACGT are complemented, then the string is reversed
void revcomp_into(char *dst, const char *s, size_t n) {
    for (size_t i = 0; i < n; i++) {
        char c = s[n - 1 - i];
        dst[i] = c == 'T' ? 'A' : c == 'G' ? 'C' : c == 'A' ? 'T' : c == 'C' ? 'G' : c;
    }
}

// istringstream >> uint32_t on a token (Alignment.cpp:63-66)
uint32_t tok_u32(const char *s, size_t n) {
    uint64_t v = 0;
    size_t i = 0;
    bool any = false, over = false, neg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    for (; i < n && s[i] >= '0' && s[i] <= '9'; i++) {
        if (!over) v = v * 10 + (uint64_t)(s[i] - '0');
        if (v > 0xFFFFFFFFull) over = true;
        any = true;
    }
    if (!any) return 0;
    if (over) return 0xFFFFFFFFu;
    return neg ? (uint32_t)(0u - (uint32_t)v) : (uint32_t)v;
}

// string blob of a batch: page-locked (dagcon_host_alloc) when a context offers it, else malloc
struct Blob {
    char *p = nullptr;
    size_t cap = 0, n = 0;
    dagcon_ctx *owner = nullptr;       // context the page-locked block came from (nullptr: malloc)
    void release() {
        if (p) { if (owner) dagcon_host_free(owner, p); else free(p); }
        p = nullptr; cap = 0; owner = nullptr;
    }
    bool resize(size_t bytes, dagcon_ctx *pin) {
        if (bytes > cap) {
            release();
            const size_t want = bytes + bytes / 8 + 4096;
            void *q = nullptr;
            if (pin && dagcon_host_alloc(pin, want, &q) == DAGCON_OK) { p = (char *)q; owner = pin; }
            else { p = (char *)malloc(want); owner = nullptr; }
            if (!p) return false;
            cap = want;
        }
        n = bytes;
        return true;
    }
    char *data() { return p; }
    size_t size() const { return n; }
};

struct Batch {
    std::vector<std::string> ids;
    std::vector<uint32_t> tlen, start, len, len2;          // len2 / off2: the target sequence of a .pre record
    std::vector<uint64_t> begin{0}, off, off2;
    std::vector<char> strand;
    Blob q, t;
    unsigned long long seq = 0;        // position in the input: records are printed in this order
    std::string out;                   // the batch's FASTA records
    void clear() { ids.clear(); tlen.clear(); start.clear(); len.clear(); len2.clear(); begin.assign(1, 0); off.clear(); off2.clear(); strand.clear(); q.n = 0; t.n = 0; out.clear(); }
};

bool g_timing = false;                                    // PBDAGCON_TIMING
std::mutex g_tmu;
double g_t_upload = 0, g_t_run = 0, g_t_fetch = 0;
double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// one batch through the device; the records go to b.out (main.cpp:141-143), warnings to stderr
int flush(dagcon_ctx *ctx, Batch &b, const Opts &o, Blob *scratch) {
    if (b.ids.empty()) return 0;
    if (b.begin.back() != b.start.size()) b.begin.push_back(b.start.size());
    dagcon_batch db;
    memset(&db, 0, sizeof db);
    db.n_targets = (uint32_t)b.ids.size();
    db.tlen = b.tlen.data(); db.aln_begin = b.begin.data();
    db.aln_start = b.start.data(); db.aln_off = b.off.data(); db.aln_len = b.len.data();
    db.qstr = b.q.data(); db.tstr = b.t.data(); db.blob_bytes = b.q.size();
    // -a: SimpleAligner on every record first (main.cpp:127-128)
    std::vector<uint64_t> ooff;
    std::vector<uint32_t> alen, nstart;
    char *qa = nullptr, *ta = nullptr;                  // -a: the aligned strings, in the worker's page-locked scratch
    dagcon_results r;
    int rc = DAGCON_OK;
    bool have_results = false;
    if (o.align && !o.polish) {
        // main.cpp:117-145 with -a in one call: the aligned strings stay on the device
        const size_t A = b.start.size();
        dagcon_pre_batch pb;
        memset(&pb, 0, sizeof pb);
        pb.n_targets = db.n_targets; pb.tlen = b.tlen.data(); pb.rec_begin = b.begin.data();
        pb.tstart = b.start.data(); pb.strand = b.strand.data();
        pb.q_off = b.off.data(); pb.q_len = b.len.data(); pb.t_off = b.off2.data(); pb.t_len = b.len2.data();
        pb.q_blob = b.q.data(); pb.q_bytes = b.q.size(); pb.t_blob = b.t.data(); pb.t_bytes = b.t.size();
        const double ta0 = wall();
        rc = dagcon_consensus_pre(ctx, &pb, &r);
        if (g_timing) fprintf(stderr, "pbdagcon timing: -a batch of %zu records: dagcon_consensus_pre %.3f\n", A, wall() - ta0);
        if (rc != DAGCON_OK) {
            fprintf(stderr, "pbdagcon: alignment / consensus failed (%d): %s\n", rc, dagcon_last_error(ctx));
            return 1;
        }
        if (const uint32_t nd = dagcon_align_dropped(ctx))
            fprintf(stderr, "pbdagcon: warning: %u of %zu records could not be aligned inside the widest band and were dropped\n", nd, A);
        have_results = true;
    } else if (o.align) {
        const double ta0 = wall();
        const size_t A = b.start.size();
        ooff.resize(A); alen.assign(A, 0); nstart.resize(A);
        uint64_t tot = 0;
        for (size_t a = 0; a < A; a++) { ooff[a] = tot; tot += (uint64_t)b.len[a] + b.len2[a]; }
        if (!scratch[0].resize(tot + 1, ctx) || !scratch[1].resize(tot + 1, ctx)) { fprintf(stderr, "pbdagcon: out of memory\n"); return 1; }
        qa = scratch[0].data(); ta = scratch[1].data();
        const double ta1 = wall();
        int rc = dagcon_align(ctx, (uint32_t)A, b.off.data(), b.len.data(), b.off2.data(), b.len2.data(), b.q.data(), b.q.size(),
                              b.t.data(), b.t.size(), ooff.data(), &qa[0], &ta[0], alen.data());
        const double ta2 = wall();
        if (g_timing) fprintf(stderr, "pbdagcon timing: -a batch of %zu records: buffers %.3f  dagcon_align %.3f\n", A, ta1 - ta0, ta2 - ta1);
        if (rc != DAGCON_OK) {
            fprintf(stderr, "pbdagcon: alignment failed (%d): %s\n", rc, dagcon_last_error(ctx));
            return 1;
        }
        if (const uint32_t nd = dagcon_align_dropped(ctx))
            fprintf(stderr, "pbdagcon: warning: %u of %zu records could not be aligned inside the widest band and were dropped\n", nd, A);
        size_t g = 0;
        for (size_t a = 0; a < A; a++) {
            while (b.begin[g + 1] <= a) g++;
            // SimpleAligner.cpp:51-62 (the alignment is global: GenomicTBegin() = 0, GenomicTEnd() = |tseq|)
            uint32_t start = b.start[a];
            const uint32_t end = start + b.len2[a];
            if (b.strand[a] == '-') {
                start = b.tlen[g] - end;
                std::string tmp(alen[a], 0);
                revcomp_into(&tmp[0], &qa[ooff[a]], alen[a]); memcpy(&qa[ooff[a]], tmp.data(), alen[a]);
                revcomp_into(&tmp[0], &ta[ooff[a]], alen[a]); memcpy(&ta[ooff[a]], tmp.data(), alen[a]);
            }
            nstart[a] = start + 1;
        }
        db.aln_start = nstart.data(); db.aln_off = ooff.data(); db.aln_len = alen.data();
        db.qstr = qa; db.tstr = ta; db.blob_bytes = tot;
    }
    if (have_results) {
    } else if (g_timing) {                                // the three steps of dagcon_consensus, timed apart
        const double t0 = wall();
        rc = dagcon_upload(ctx, &db);
        const double t1 = wall();
        if (rc == DAGCON_OK) rc = dagcon_run(ctx);
        if (rc == DAGCON_OK) rc = dagcon_sync(ctx);
        const double t2 = wall();
        if (rc == DAGCON_OK) rc = dagcon_fetch(ctx, &r);
        const double t3 = wall();
        std::lock_guard<std::mutex> lk(g_tmu);
        g_t_upload += t1 - t0; g_t_run += t2 - t1; g_t_fetch += t3 - t2;
    } else rc = dagcon_consensus(ctx, &db, &r);
    if (rc != DAGCON_OK) {
        fprintf(stderr, "pbdagcon: consensus failed (%d): %s\n", rc, dagcon_last_error(ctx));
        return 1;
    }
    // --polish: the consensus becomes the backbone, the reads are aligned to it again, N times.  The
    // reference names this use (README.md:14-15) and leaves it to the caller; nothing of it is in its
    // C++ sources, so there is no reference behaviour to match: the steps are this build's own
    // (longest segment as the new backbone, dagcon_align of every read against the stretch of it the
    // read covered before, unaligned target flanks stripped, real-backbone consensus as dazcon.cpp:76).
    // Every round: (1) the longest segment is the new backbone; it covers about positions
    // [trim + range0, trim + range1) of the previous one.  (2) From its previous alignment each read is
    // clipped to what lies over that stretch (plus a margin) and the stretch of the new backbone it covers is
    // estimated; (3) dagcon_align, global over the two pieces, backbone bases in front of / behind the read
    // stripped; (4) real-backbone consensus as dazcon.cpp:76 does.
    if (o.align && o.polish) {
        const size_t A = b.start.size();
        const uint32_t T = db.n_targets;
        const uint32_t pad = 64;
        std::vector<uint32_t> cur_start(A), cur_len(A), cur_qbase(A, 0);   // qbase: the read's base the current alignment begins with
        std::vector<uint64_t> cur_off(A);
        std::string cur_q(qa, db.blob_bytes + 1), cur_t(ta, db.blob_bytes + 1);     // the reads' last alignments, per record
        for (size_t a = 0; a < A; a++) { cur_start[a] = db.aln_start[a]; cur_off[a] = db.aln_off[a]; cur_len[a] = db.aln_len[a]; }
        std::vector<uint64_t> p_qoff(A), p_toff(A), p_ooff(A), p_begin, p_bboff;
        std::vector<uint32_t> p_qlen(A), p_tlen(A), p_alen(A), p_tl, w0(A);
        std::string p_q, p_t, p_qa, p_ta, p_bb, fwd;
        std::vector<uint32_t> k_start, k_len; std::vector<uint64_t> k_off;
        for (unsigned round = 1; round <= o.polish; round++) {
            std::vector<std::string> bb(T);
            std::vector<int32_t> r0(T, 0), r1(T, 0);
            for (uint32_t g = 0; g < T; g++) {
                uint32_t best = 0;                           // (the first of the longest ones, as AlnGraphBoost.cpp:309,319 breaks ties)
                for (uint64_t sg = r.seg_begin[g]; sg < r.seg_begin[g + 1]; sg++)
                    if (r.seq_len[sg] > best) { best = r.seq_len[sg]; bb[g].assign(r.seq_blob + r.seq_off[sg], r.seq_len[sg]); r0[g] = r.range0[sg]; r1[g] = r.range1[sg]; }
            }
            p_q.clear(); p_t.clear();
            uint64_t tot = 0;
            size_t g = 0;
            for (size_t a = 0; a < A; a++) {
                while (b.begin[g + 1] <= a) g++;
                const std::string &B = bb[g];
                p_qoff[a] = p_q.size(); p_toff[a] = p_t.size(); p_ooff[a] = tot; p_qlen[a] = 0; p_tlen[a] = 0; w0[a] = 0;
                if (B.empty() || cur_len[a] == 0) { cur_len[a] = 0; continue; }     // (no consensus, or nothing left of the read)
                // the new backbone lies over [lo_t, hi_t) of the previous one, margins included (0-based)
                const int64_t org = (int64_t)o.trim + r0[g];
                const int64_t lo_t = org - pad, hi_t = (int64_t)o.trim + r1[g] + pad;
                int64_t tpos = (int64_t)cur_start[a] - 1;
                uint32_t qpos = 0, qlo = 0, qhi = 0;
                int64_t t_first = -1, t_last = -1;
                for (uint32_t i = 0; i < cur_len[a]; i++) {
                    const char qc = cur_q[cur_off[a] + i], tc = cur_t[cur_off[a] + i];
                    const bool in = tpos >= lo_t && tpos < hi_t;
                    if (qc != '-') { if (tpos < lo_t) qlo = qpos + 1; if (in) qhi = qpos + 1; qpos++; }
                    if (in && tc != '-') { if (t_first < 0) t_first = tpos; t_last = tpos; }
                    if (tc != '-') tpos++;
                }
                if (qhi <= qlo || t_first < 0) { cur_len[a] = 0; continue; }
                const int64_t a0 = std::max<int64_t>(0, std::min<int64_t>(t_first - org - pad, (int64_t)B.size()));
                const int64_t a1 = std::max<int64_t>(a0, std::min<int64_t>(t_last + 1 - org + pad, (int64_t)B.size()));
                w0[a] = (uint32_t)a0; p_tlen[a] = (uint32_t)(a1 - a0);
                p_t.append(B, (size_t)a0, (size_t)(a1 - a0));
                // the read in the target's orientation, clipped
                fwd.resize(b.len[a]);
                if (b.strand[a] == '-') revcomp_into(&fwd[0], b.q.data() + b.off[a], b.len[a]);
                else memcpy(&fwd[0], b.q.data() + b.off[a], b.len[a]);
                p_qlen[a] = qhi - qlo;
                p_q.append(fwd, cur_qbase[a] + qlo, qhi - qlo);
                cur_qbase[a] += qlo;
                tot += (uint64_t)p_qlen[a] + p_tlen[a];
            }
            p_qa.assign(tot + 1, 0); p_ta.assign(tot + 1, 0);
            if (p_q.empty()) p_q.push_back(0);
            if (p_t.empty()) p_t.push_back(0);
            rc = dagcon_align(ctx, (uint32_t)A, p_qoff.data(), p_qlen.data(), p_toff.data(), p_tlen.data(), p_q.data(), p_q.size(),
                              p_t.data(), p_t.size(), p_ooff.data(), &p_qa[0], &p_ta[0], p_alen.data());
            if (rc != DAGCON_OK) { fprintf(stderr, "pbdagcon: alignment failed (%d): %s\n", rc, dagcon_last_error(ctx)); return 1; }
            // global over the two pieces: backbone bases in front of / behind the read are not part of its alignment
            k_start.clear(); k_off.clear(); k_len.clear();
            p_begin.assign(1, 0);
            g = 0;
            for (size_t a = 0; a < A; a++) {
                while (b.begin[g + 1] <= a) { g++; p_begin.push_back(k_start.size()); }
                if (cur_len[a] == 0) continue;
                uint32_t n = p_alen[a], lead = 0;
                uint64_t off = p_ooff[a];
                while (n && p_qa[off] == '-') { off++; n--; lead++; }
                while (n && p_qa[off + n - 1] == '-') n--;
                cur_start[a] = w0[a] + lead + 1u; cur_off[a] = off; cur_len[a] = n;
                if (n) { k_start.push_back(cur_start[a]); k_off.push_back(off); k_len.push_back(n); }
            }
            while (p_begin.size() < (size_t)T + 1) p_begin.push_back(k_start.size());
            cur_q.swap(p_qa); cur_t.swap(p_ta);            // (the next round clips against these)
            p_tl.assign(T, 0); p_bboff.assign(T, 0); p_bb.clear();
            for (uint32_t t2 = 0; t2 < T; t2++) { p_tl[t2] = (uint32_t)bb[t2].size(); p_bboff[t2] = p_bb.size(); p_bb += bb[t2]; }
            if (p_bb.empty()) p_bb.push_back('N');
            memset(&db, 0, sizeof db);
            db.n_targets = T; db.tlen = p_tl.data(); db.aln_begin = p_begin.data();
            db.aln_start = k_start.data(); db.aln_off = k_off.data(); db.aln_len = k_len.data();
            db.qstr = cur_q.data(); db.tstr = cur_t.data(); db.blob_bytes = cur_q.size();
            db.backbone = p_bb.data(); db.backbone_off = p_bboff.data();
            rc = dagcon_consensus(ctx, &db, &r);
            if (rc != DAGCON_OK) { fprintf(stderr, "pbdagcon: consensus failed (%d): %s\n", rc, dagcon_last_error(ctx)); return 1; }
        }
    }
    char head[64];
    for (uint32_t g = 0; g < r.n_targets; g++) {
        if (o.verbose)
            fprintf(stderr, "Consensus calling: %s Alignments: %llu\n", b.ids[g].c_str(),
                    (unsigned long long)(b.begin[g + 1] - b.begin[g]));
        // a failure is confined to its target (the reference's assert hits one worker's one target,
        // AlnGraphBoost.cpp:71-72): warn, go on with the rest
        if (r.target_status[g] != DAGCON_OK)
            fprintf(stderr, "pbdagcon: warning: target %s skipped (%s)\n", b.ids[g].c_str(),
                    r.target_status[g] == DAGCON_ERR_NONCONFORMING ? "an alignment leaves the backbone or holds a non-printable byte"
                    : r.target_status[g] == DAGCON_ERR_UNSUPPORTED ? "too large" : "internal error");
        for (uint64_t s = r.seg_begin[g]; s < r.seg_begin[g + 1]; s++) {
            // main.cpp:141-143  ">%s/%d_%d\n%s\n"
            b.out += '>'; b.out += b.ids[g];
            snprintf(head, sizeof head, "/%d_%d\n", r.range0[s], r.range1[s]);
            b.out += head;
            b.out.append(r.seq_blob + r.seq_off[s], r.seq_len[s]);
            b.out += '\n';
        }
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    Opts o;
    if (int rc = parse_args(argc, argv, o)) return rc;
    // PBDAGCON_TIMING=1: where the wall time of the run went, on stderr (seconds)
    const bool timing = getenv("PBDAGCON_TIMING") != nullptr;
    g_timing = timing;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_main = now();
    double t_index = 0, t_fill = 0, t_wait = 0, t_create = 0, t_flush = 0, t_print = 0, t_parse_end = 0, t_joined = 0;
    // ---- input: mmap a file, or slurp stdin ----
    const char *data = nullptr;
    size_t size = 0;
    std::string slurp;
    void *map = nullptr;
    if (o.input == "-") {
        char buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, stdin)) > 0) slurp.append(buf, n);
        data = slurp.data(); size = slurp.size();
    } else {
        int fd = open(o.input.c_str(), O_RDONLY);
        if (fd < 0) { fprintf(stderr, "pbdagcon: error opening file: %s\n", o.input.c_str()); return 1; }
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); return 1; }
        size = (size_t)st.st_size;
        if (size) {
            map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) { fprintf(stderr, "pbdagcon: mmap failed\n"); close(fd); return 1; }
            data = (const char *)map;
        }
        close(fd);
    }

    // ---- consensus workers: one thread + context per GPU (the reference starts its N consensus
    // workers itself too, main.cpp:251-274); batches are taken in input order from one queue, their
    // records are printed in input order by whoever completes the next one in line ----
    // (two contexts per GPU: one batch's host-to-device copy, host preparation and record formatting go on
    // beside the other's kernels)
    const unsigned per_dev = o.contexts ? o.contexts : (size > 512ull << 20 ? 2u : 1u);
    std::vector<int> worker_dev;
    for (unsigned k = 0; k < per_dev; k++) for (int d : o.devices) worker_dev.push_back(d);
    const size_t ndev = worker_dev.size();
    const size_t nbuf = ndev + 1;                          // the parser fills one while the others are on GPUs
    std::vector<Batch> bufs(nbuf);
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Batch *> free_list, work;                   // work: FIFO
    for (auto &x : bufs) free_list.push_back(&x);
    std::vector<Batch *> done;                              // completed, waiting for their turn to print
    unsigned long long next_seq = 0, print_seq = 0;
    bool stop = false;
    int worker_status = 0;
    dagcon_ctx *pin_ctx = nullptr;                          // first context up: page-locked blobs come from it
    bool want_pin = o.pinned == 1 || (o.pinned < 0 && size > 2 * o.batch_bytes);
    std::vector<std::thread> workers;
    if (!o.dump) {
        for (size_t w = 0; w < ndev; w++) workers.emplace_back([&, w] {
            dagcon_ctx *ctx = nullptr;
            dagcon_opts dopt;
            dagcon_default_opts(&dopt);
            dopt.min_cov = o.min_cov; dopt.min_len = o.min_len; dopt.trim = o.trim;
            dopt.min_weight = (int32_t)o.min_cov;          // main.cpp:261,279 (quirk Q1)
            dopt.device = worker_dev[w];
            const double tc0 = now();
            int rc = dagcon_create(&dopt, &ctx);
            if (w == 0) t_create = now() - tc0;
            if (rc != DAGCON_OK) {
                fprintf(stderr, "pbdagcon: no usable MI355X as device %d (dagcon_create = %d); there is no CPU fallback\n", worker_dev[w], rc);
                std::lock_guard<std::mutex> lk(mu);
                worker_status = 1;
                cv.notify_all();
                return;
            }
            { std::lock_guard<std::mutex> lk(mu); if (!pin_ctx) pin_ctx = ctx; }
            Blob scratch[2];
            for (;;) {
                Batch *b = nullptr;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !work.empty() || stop; });
                    if (work.empty()) break;
                    b = work.front(); work.erase(work.begin());
                }
                const double tf0 = now();
                const int st = flush(ctx, *b, o, scratch);
                const double tf = now() - tf0;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    t_flush += tf;
                    if (st) worker_status = st;
                    done.push_back(b);
                    // print what is next in line (this batch and any that were waiting on it)
                    for (bool again = true; again;) {
                        again = false;
                        for (size_t i = 0; i < done.size(); i++) {
                            if (done[i]->seq != print_seq) continue;
                            Batch *d = done[i];
                            done.erase(done.begin() + i);
                            { const double tp0 = now(); fwrite(d->out.data(), 1, d->out.size(), stdout); t_print += now() - tp0; }
                            d->clear();
                            free_list.push_back(d);
                            print_seq++;
                            again = true;
                            break;
                        }
                    }
                }
                cv.notify_all();
            }
            // (blobs that were page-locked through this context are released before it goes)
            {
                std::unique_lock<std::mutex> lk(mu);
                for (auto &x : bufs) { if (x.q.owner == ctx) x.q.release(); if (x.t.owner == ctx) x.t.release(); }
                if (pin_ctx == ctx) pin_ctx = nullptr;
            }
            scratch[0].release(); scratch[1].release();
            dagcon_destroy(ctx);
        });
    }
    Batch *bp = nullptr;
    // a free batch buffer for the parser (waits for a worker to finish one)
    auto acquire = [&]() -> Batch * {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !free_list.empty() || worker_status; });
        if (free_list.empty()) return nullptr;
        Batch *b = free_list.back(); free_list.pop_back();
        return b;
    };
    // hands the filled batch to the workers and takes the next free buffer
    auto submit = [&]() -> int {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (worker_status) return worker_status;
            bp->seq = next_seq++;
            work.push_back(bp);
        }
        cv.notify_all();
        const double tw0 = now();
        bp = acquire();
        t_wait += now() - tw0;
        return bp ? 0 : 1;
    };

    // ---- parse (Alignment.cpp:44-80) and group by target id (BlasrM5AlnProvider.cpp:34-55) ----
    // 1. index: o.threads threads split the text at line starts and record where every field of
    //    every record is (nothing is copied);
    // 2. in file order: records are grouped into targets and targets into batches;
    // 3. per batch the same threads copy (or reverse-complement) the strings into the blobs.
    struct Rec {
        const char *id, *name, *q, *t;
        uint32_t idl, namel, len, tlen, start, tl;   // tl: length of the target sequence (.pre); len: of the query string
        char strand;
    };
    struct Part { std::vector<Rec> recs; int err = 0; unsigned long long err_rec = 0; int err_nf = 0; };
    const unsigned nthr = std::max(1u, std::min(o.threads, 64u));
    // the text is taken a slab at a time, so that the first batch reaches the GPU before the
    // whole file has been indexed; the records of a slab's last (possibly unfinished) target
    // are carried into the next slab
    const size_t slab_bytes = o.slab_bytes ? o.slab_bytes : std::max<size_t>(o.batch_bytes, 256u << 20);
    size_t slab_pos = 0, unmapped = 0;
    unsigned long long n_rec_before = 0;
    bool had_error = false;
    int status = 0;
    std::vector<Rec> carry;
    std::vector<Part> parts(nthr);
    std::vector<const Rec *> recs;
    auto index_slab = [&](size_t s0, size_t s1) {
        std::vector<size_t> cut(nthr + 1, s1);
        cut[0] = s0;
        for (unsigned k = 1; k < nthr; k++) {
            size_t p0 = std::max(cut[k - 1], s0 + (size_t)((unsigned long long)(s1 - s0) * k / nthr));
            if (p0 > s0 && p0 < s1) {
                const char *nl = (const char *)memchr(data + p0 - 1, '\n', s1 - (p0 - 1));
                p0 = nl ? (size_t)(nl - data) + 1 : s1;
            }
            cut[k] = std::min(p0, s1);
        }
        auto index = [&](unsigned k) {
            Part &pt = parts[k];
            pt.recs.clear(); pt.err = 0;
            size_t pos = cut[k];
            const size_t stop = cut[k + 1];
            while (pos < stop) {
                const char *line = data + pos;
                const char *nl = (const char *)memchr(line, '\n', size - pos);
                size_t ll = nl ? (size_t)(nl - line) : size - pos;
                pos += ll + (nl ? 1 : 0);
                if (ll && line[ll - 1] == '\r') ll--;
                const char *f[19];
                size_t fl[19];
                int nf = 0;
                size_t i = 0;
                while (i < ll && nf < 19) {
                    while (i < ll && line[i] == ' ') i++;
                    if (i >= ll) break;
                    const char *sp = (const char *)memchr(line + i, ' ', ll - i);   // fields 16..18 are ~tlen chars each
                    const size_t j = sp ? (size_t)(sp - line) : ll;
                    f[nf] = line + i; fl[nf] = j - i; nf++;
                    i = j;
                }
                if (nf == 0) continue;                          // blank line
                if (o.align) {
                    // Alignment.cpp:82-112 parsePre: qid tid strand tlen tstart tend qseq tseq
                    if (nf < 8) { pt.err = 1; pt.err_rec = pt.recs.size() + 1; pt.err_nf = nf; return; }
                    Rec r;
                    r.id = f[1]; r.idl = (uint32_t)fl[1];
                    r.name = f[0]; r.namel = (uint32_t)fl[0];
                    r.strand = f[2][0];
                    r.tlen = tok_u32(f[3], fl[3]);
                    r.start = tok_u32(f[4], fl[4]);             // (SimpleAligner.cpp:61 adds the 1)
                    r.q = f[6]; r.len = (uint32_t)fl[6];
                    r.t = f[7]; r.tl = (uint32_t)fl[7];
                    pt.recs.push_back(r);
                    continue;
                }
                if (nf < 19) { pt.err = 1; pt.err_rec = pt.recs.size() + 1; pt.err_nf = nf; return; }
                if (fl[16] != fl[18]) { pt.err = 2; pt.err_rec = pt.recs.size() + 1; return; }
                Rec r;
                r.id = f[5]; r.idl = (uint32_t)fl[5];
                r.name = f[0]; r.namel = (uint32_t)fl[0];
                r.q = f[16]; r.t = f[18]; r.len = (uint32_t)fl[16]; r.tl = r.len;
                r.tlen = tok_u32(f[6], fl[6]);
                r.start = tok_u32(f[7], fl[7]) + 1;             // Alignment.cpp:65-66
                r.strand = f[9][0];
                pt.recs.push_back(r);
            }
        };
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nthr; k++) th.emplace_back(index, k);
        index(0);
        for (auto &x : th) x.join();
        // records in file order, up to the first malformed one
        recs.clear();
        for (const Rec &r : carry) recs.push_back(&r);
        for (unsigned k = 0; k < nthr; k++) {
            for (const Rec &r : parts[k].recs) recs.push_back(&r);
            n_rec_before += parts[k].recs.size();
            if (parts[k].err == 1) {
                fprintf(stderr, "pbdagcon: format error: record %llu has %d fields, %d expected\n", n_rec_before + 1, parts[k].err_nf, o.align ? 8 : 19);
                had_error = true; break;
            }
            if (parts[k].err == 2) {
                fprintf(stderr, "pbdagcon: format error: record %llu: query and target strings differ in length\n", n_rec_before + 1);
                had_error = true; break;
            }
        }
    };
    // copies the strings of records [r0, r1) into batch b, whose offsets are set already
    auto fill_strings = [&](Batch &b, size_t r0, size_t r1, size_t bytes, size_t bytes2) {
        dagcon_ctx *pin = nullptr;
        if (want_pin) { std::lock_guard<std::mutex> lk(mu); pin = pin_ctx; }
        if (!b.q.resize(bytes, pin) || !b.t.resize(bytes2, pin)) { fprintf(stderr, "pbdagcon: out of memory\n"); exit(1); }
        auto work = [&](unsigned k) {
            for (size_t x = r0 + k; x < r1; x += nthr) {
                const Rec &r = *recs[x];
                char *dq = b.q.data() + b.off[x - r0], *dt = b.t.data() + b.off2[x - r0];
                if (o.align) {                                    // .pre: sequences as they are (Alignment.cpp:112)
                    memcpy(dq, r.q, r.len);
                    memcpy(dt, r.t, r.tl);
                } else if (r.strand == '-') {                            // Alignment.cpp:69-75: start is NOT flipped (Q6)
                    revcomp_into(dq, r.q, r.len);
                    revcomp_into(dt, r.t, r.len);
                } else {
                    memcpy(dq, r.q, r.len);
                    memcpy(dt, r.t, r.len);
                }
            }
        };
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nthr; k++) th.emplace_back(work, k);
        work(0);
        for (auto &x : th) x.join();
    };
    bp = &bufs[0];
    if (!o.dump) bp = acquire();
    if (!bp) status = 1;
#define b (*bp)
    while (status == 0 && !had_error && (slab_pos < size || !carry.empty())) {
        size_t s1 = std::min(size, slab_pos + slab_bytes);
        if (s1 < size) {                                   // a slab ends at a line end
            const char *nl = (const char *)memchr(data + s1, '\n', size - s1);
            s1 = nl ? (size_t)(nl - data) + 1 : size;
        }
        { const double t0 = now(); index_slab(slab_pos, s1); t_index += now() - t0; }
        slab_pos = s1;
        const bool eof = slab_pos >= size || had_error;
        // all but the last target of the slab (it may go on in the next one)
        size_t n_use = recs.size();
        if (!eof) {
            while (n_use > 0 && recs[n_use - 1]->idl == recs.back()->idl &&
                   memcmp(recs[n_use - 1]->id, recs.back()->id, recs.back()->idl) == 0) n_use--;
        }
        size_t rb = 0;                                   // first record of the batch being formed
        size_t bytes = 0, bytes2 = 0;
        for (size_t x = 0; x <= n_use && status == 0; x++) {
            const bool last = x == n_use;
            const bool new_target = !last && (x == rb || recs[x]->idl != recs[x - 1]->idl ||
                                              memcmp(recs[x]->id, recs[x - 1]->id, recs[x]->idl) != 0);
            // a batch is closed when it is full, and at the end of the slab's usable records once it
            // holds something (at the end of the input whatever it holds)
            if (last || (new_target && x > rb && (b.ids.size() >= o.batch_targets || bytes + bytes2 >= o.batch_bytes))) {   // (-a: the t strings count too)
                if (x > rb) {
                    b.begin.push_back(b.start.size());
                    { const double t0 = now(); fill_strings(b, rb, x, bytes, bytes2); t_fill += now() - t0; }
                    if (o.dump) {
                        for (size_t y = rb; y < x; y++) {
                            const Rec &r = *recs[y];
                            const size_t o0 = b.off[y - rb];
                            size_t g = 0;
                            while (b.begin[g + 1] <= y - rb) g++;
                            printf("%.*s\t%u\t%u\t%c\t%.*s\t%.*s\t%.*s\n", (int)r.idl, r.id, b.tlen[g], r.start, r.strand,
                                   (int)r.namel, r.name, (int)r.len, b.q.data() + o0, (int)r.tl, b.t.data() + b.off2[y - rb]);
                        }
                        b.clear();
                    } else status = submit();
                }
                rb = x; bytes = 0; bytes2 = 0;
                if (last || status) break;
            }
            const Rec &r = *recs[x];
            if (new_target) {
                if (x > rb) b.begin.push_back(b.start.size());
                b.ids.emplace_back(r.id, r.idl);
                b.tlen.push_back(r.tlen);
            }
            b.start.push_back(r.start);
            b.off.push_back(bytes); b.off2.push_back(bytes2);
            b.len.push_back(r.len); b.len2.push_back(r.tl);
            b.strand.push_back(r.strand);
            bytes += r.len; bytes2 += r.tl;
        }
        // the unfinished target's records wait for the next slab
        std::vector<Rec> next_carry;
        for (size_t x = n_use; x < recs.size(); x++) next_carry.push_back(*recs[x]);
        carry.swap(next_carry);
        // the text in front of the first carried record is finished with: its page-table entries go now, while the
        // GPU works, instead of all at once at the end (0.3 s for 26 GB of text)
        if (map) {
            const size_t dead = carry.empty() ? slab_pos : (size_t)(carry[0].name - data);
            const size_t upto = dead & ~(size_t)((2u << 20) - 1);
            if (upto > unmapped) {
                // MADV_DONTNEED takes the address-space lock shared (munmap takes it exclusively and would stall the
                // workers' page faults and the driver's pinning): the threads drop a share of the range each
                const size_t n2m = (upto - unmapped) >> 21;
                auto drop = [&](unsigned k) {
                    const size_t a = unmapped + ((n2m * k / nthr) << 21), e = unmapped + ((n2m * (k + 1) / nthr) << 21);
                    if (e > a) madvise((char *)map + a, e - a, MADV_DONTNEED);
                };
                std::vector<std::thread> th;
                for (unsigned k = 1; k < nthr; k++) th.emplace_back(drop, k);
                drop(0);
                for (auto &x : th) x.join();
                unmapped = upto;
            }
        }
        if (eof && carry.empty()) break;
        if (eof) slab_pos = size;
    }
    if (had_error) status = 1;
    t_parse_end = now();
#undef b
    if (!o.dump) {
        {
            std::unique_lock<std::mutex> lk(mu);
            // every submitted batch printed (or a worker gave up)
            cv.wait(lk, [&] { return print_seq == next_seq || worker_status; });
            stop = true;
            if (worker_status && !status) status = worker_status;
        }
        cv.notify_all();
        for (auto &w : workers) w.join();          // (every worker destroys its context: the device memory goes back in order,
                                                   // so that the next process's large hipMalloc does not wait for it)
        // Every record is printed and the GPU is released: what is left is host-side tidying -- page-locked blobs, the
        // mapping of the input, the HIP runtime's own exit handlers (0.1 - 0.3 s at 1,000 targets) -- which the kernel
        // does for a process that ends, at once.  PBDAGCON_TEARDOWN=1 keeps the orderly way.
        if (!getenv("PBDAGCON_TEARDOWN") && !status) {
            fflush(stdout);
            if (timing)
                fprintf(stderr, "pbdagcon timing: total %.3f = parse loop %.3f (index %.3f  fill %.3f  wait-for-buffer %.3f) + drain %.3f, no host teardown | "
                        "worker 0: create %.3f; all workers: flush %.3f (upload %.3f  run %.3f  fetch %.3f)  print %.3f\n",
                        now() - t_main, t_parse_end - t_main, t_index, t_fill, t_wait, now() - t_parse_end, t_create, t_flush,
                        g_t_upload, g_t_run, g_t_fetch, t_print);
            fflush(stderr);
            _exit(0);
        }
    }
    t_joined = now();
    for (auto &x : bufs) { x.q.release(); x.t.release(); }
    if (map) munmap(map, size);
    fflush(stdout);
    if (timing)
        fprintf(stderr, "pbdagcon timing: total %.3f = parse loop %.3f (index %.3f  fill %.3f  wait-for-buffer %.3f) + drain %.3f + teardown %.3f | "
                "worker 0: create %.3f; all workers: flush %.3f (upload %.3f  run %.3f  fetch %.3f)  print %.3f\n",
                now() - t_main, t_parse_end - t_main, t_index, t_fill, t_wait, t_joined - t_parse_end, now() - t_joined, t_create, t_flush,
                g_t_upload, g_t_run, g_t_fetch, t_print);
    return status;
}
