// dazcon_main.cpp -- `dazcon`-compatible command line on top of the C ABI.
//
// The reference's dazcon (src/cpp/dazcon.cpp) reads DALIGNER overlaps (.las) and a DAZZ_DB
// database (.db) through the DALIGNER / DAZZ_DB C sources and recomputes every alignment with
// DALIGNER's Compute_Trace_PTS (DazAlnProvider.cpp:304-369).  Those libraries are not in the tree
// (empty submodules): the binary formats and the trace-point realigner are PARITY UNPINNED and
// are not rebuilt here.  What IS first-party in that front end is restated in this file, pinned by
// the reference's own known-answer tests (test/cpp/TargetHitTest.cpp:4-77):
//
//   TargetHit::add / belongs / computeOvlScore     DazAlnProvider.cpp:165-211
//   Target::addRecord / sortHits / getAlignments    DazAlnProvider.cpp:264-369 (hit selection, -m, -x, -o)
//   decodeAlignment                                 DazAlnProvider.cpp:383-417 (trace -> alignment strings)
//   nextTarget's filters                            DazAlnProvider.cpp:79-117  (target list, min coverage)
//   Consensus / record format                       dazcon.cpp:61-107 (real backbone, -t 10, ">%s/%d/%d_%d")
//
// with the reference's flags and defaults (dazcon.cpp:122-192).  Round 3: -a <file>.las and -s <file>.db are read in
// their binary layouts (daz_io.h: PARITY UNPINNED, tested by round trip against this build's own writer), and every
// overlap of a .las is aligned between its end points by the device aligner (dagcon_align) where the reference runs
// DALIGNER's Compute_Trace_PTS.  Otherwise -s names a text file that carries what
// the .db holds (the reads) and -a one that carries what the .las holds after Read_Overlap /
// Compute_Trace_PTS (overlap records with their trace points, or with alignment strings already
// decoded); INTEGRATION.md gives the layout:
//
//   -s file:  one read per line          <read id, 1-based>  <sequence ACGT>
//   -a file:  one overlap per line, sorted by A-read like a .las
//       O <aread> <bread> <flags> <abpos> <aepos> <bbpos> <bepos> <diffs> <tstr> <qstr>
//       R <aread> <bread> <flags> <abpos> <aepos> <bbpos> <bepos> <diffs> <trace: comma separated ints, or ->
//     (read numbers 1-based as dazcon prints them; R records are decoded by decodeAlignment from the
//     two reads of the -s file, B complemented when flags & 1, COMP(), as DazAlnProvider.cpp:343-347 does)
//
// Deliberate differences: records are printed in input order (Q3); the header's second field, an
// uninitialised int in the reference (dazcon.cpp:62, Q4), counts from 0 per run; std::sort's order
// among equal scores is unspecified in the reference, here equal scores keep their input order.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <numeric>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/dagcon.h"
#include "daz_io.h"

namespace {

struct Opts {
    int threads = 4;
    unsigned min_cov = 6, min_len = 500, trim = 10, max_hits = 85;
    bool sort_cov = false, proper = false, verbose = false, dump_hits = false, dump_alns = false;
    std::string aln_file, seq_file;
    std::set<int> targets;
    size_t batch_targets = 512;
    int device = 0;
};

void usage(FILE *f) {
    fprintf(f,
            "USAGE: dazcon -a <overlaps> -s <reads> [-j <int>] [-c <uint>] [-l <uint>] [-t <uint>] [-m <uint>] [-x] [-o] [-v] [targets ...]\n"
            "  PBI consensus module (DAGCon over daligner-style overlaps); the consensus runs on an MI355X.\n"
            "  -a, --align-file    overlaps: a DALIGNER .las file (name ends in .las), or the text layout of INTEGRATION.md\n"
            "  -s, --seq-file      reads: a DAZZ_DB database (name ends in .db; its .idx / .bps beside it), or the text layout\n"
            "  -j, --threads       accepted for compatibility (default 4); the consensus is the GPU's\n"
            "  -c, --min-coverage  minimum coverage for correction (default 6)\n"
            "  -l, --min-len       minimum length for correction (default 500)\n"
            "  -t, --trim          trim alignments on either side (default 10)\n"
            "  -m, --max-hit       maximum number of hits to pass to consensus (default 85)\n"
            "  -x, --coverage-sort sort hits by coverage\n"
            "  -o, --only-proper-overlaps  use only overlaps that align to the ends\n"
            "  -v, --verbose\n"
            "  targets             limit consensus to these target ids\n"
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
        auto need_s = [&](std::string *dst) {
            if (i + 1 >= argc) { fprintf(stderr, "PARSE ERROR: %s needs a value\n", a.c_str()); return false; }
            *dst = argv[++i];
            return true;
        };
        unsigned u = 0;
        if (a == "-j" || a == "--threads") { if (!need(&u)) return 1; o.threads = (int)u; }
        else if (a == "-c" || a == "--min-coverage") { if (!need(&o.min_cov)) return 1; }
        else if (a == "-l" || a == "--min-len") { if (!need(&o.min_len)) return 1; }
        else if (a == "-t" || a == "--trim") { if (!need(&o.trim)) return 1; }
        else if (a == "-m" || a == "--max-hit") { if (!need(&o.max_hits)) return 1; }
        else if (a == "-a" || a == "--align-file") { if (!need_s(&o.aln_file)) return 1; }
        else if (a == "-s" || a == "--seq-file") { if (!need_s(&o.seq_file)) return 1; }
        else if (a == "-x" || a == "--coverage-sort") o.sort_cov = true;
        else if (a == "-o" || a == "--only-proper-overlaps") o.proper = true;
        else if (a == "-v" || a == "--verbose") o.verbose = true;
        else if (a == "--dump-hits") o.dump_hits = true;          // test hook: hit selection only, no GPU
        else if (a == "--dump-alns") o.dump_alns = true;          // test hook: the alignments handed to the consensus, no GPU
        else if (a == "--device") { if (!need(&u)) return 1; o.device = (int)u; }
        else if (a == "--batch-targets") { if (!need(&u) || !u) return 1; o.batch_targets = u; }
        else if (a == "-h" || a == "--help") { usage(stdout); exit(0); }
        else if (a == "--version") { printf("dazcon  version: 0.3\n"); exit(0); }
        else if (a[0] != '-') {
            char *e = nullptr;
            const long v = strtol(a.c_str(), &e, 10);
            if (!e || *e) { fprintf(stderr, "PARSE ERROR: target ids are integers\n"); return 1; }
            o.targets.insert((int)v);
        } else { fprintf(stderr, "PARSE ERROR: unknown argument %s\n", a.c_str()); return 1; }
    }
    // dazcon.cpp:148-156: both files are required
    if (o.aln_file.empty() || o.seq_file.empty()) { fprintf(stderr, "PARSE ERROR: required arguments missing: -a and -s\n"); usage(stderr); return 1; }
    return 0;
}

// ---- what DALIGNER's Overlap / Path carry (align.h), as far as the first-party code reads them ----
struct Path { int abpos = 0, aepos = 0, bbpos = 0, bepos = 0, diffs = 0; };
struct Record {
    int aread = 0, bread = 0;          // 0-based
    unsigned flags = 0;
    Path path;
    bool decoded = false;              // O record: strings given
    bool realign = false;              // .las record: aligned between its end points on the device (see daz_io.h)
    std::string tstr, qstr;
    std::vector<int> trace;            // R record: what Compute_Trace_PTS left in path.trace
};

// DazAlnProvider.cpp:165-211
struct TargetHit {
    float ovlScore = 0.0f, covScore = 0.0f;
    int aread = -1, bread = 0;
    unsigned flags = 0;
    int alen = 0, blen = 0;
    std::vector<Record> records;
    bool belongs(const Record &r) const { return aread == r.aread && bread == r.bread && flags == r.flags; }   // :165-169
    void add(Record &&rec) {                                                                                    // :171-188
        if (records.empty()) { records.push_back(std::move(rec)); return; }
        const Path &prev = records.back().path;
        const int prevLen = prev.aepos - prev.abpos;
        const Path &curr = rec.path;
        const int currLen = curr.aepos - curr.abpos;
        if (curr.abpos > prev.aepos) records.push_back(std::move(rec));
        else if (currLen > prevLen) { records.pop_back(); records.push_back(std::move(rec)); }
    }
    void computeOvlScore(bool proper) {                                                                          // :190-211
        int ahlen = 0, bhlen = 0, diff = 0;
        for (const Record &rec : records) {
            const Path &p = rec.path;
            ahlen += p.aepos - p.abpos;
            bhlen += p.bepos - p.bbpos;
            diff += std::abs(ahlen - bhlen) + p.diffs;
        }
        ovlScore = (1 - diff / (float)ahlen) * ahlen;
        if (proper) {
            const Path &f = records.front().path, &b = records.back().path;
            if (f.abpos != 0 && b.bbpos != 0) ovlScore = 0.0f;
            if (f.aepos != alen && b.bepos != blen) ovlScore = 0.0f;
        }
    }
};

float invertedSum(float x, unsigned int y) { return x + 1 / (float)y; }                                         // :379-381

struct Target {
    int id = -1, length = 0;
    std::vector<TargetHit> hits;
    std::vector<unsigned> coverage;
    // :264-283
    void addRecord(Record &&rec, int blen, bool proper) {
        if (!hits.empty() && hits.back().belongs(rec)) {
            hits.back().add(std::move(rec));
            hits.back().computeOvlScore(proper);
            return;
        }
        TargetHit hit;
        hit.aread = rec.aread; hit.bread = rec.bread; hit.flags = rec.flags;
        hit.alen = length; hit.blen = blen;
        hit.add(std::move(rec));
        hit.computeOvlScore(proper);
        hits.push_back(std::move(hit));
    }
    // :285-302 (std::sort there: the order among equal scores is unspecified; stable here)
    void sortHits(bool sortCov) {
        std::stable_sort(hits.begin(), hits.end(), [](const TargetHit &l, const TargetHit &r) { return l.ovlScore > r.ovlScore; });
        if (!sortCov) return;
        coverage.assign((size_t)length, 0u);
        for (TargetHit &hit : hits) {
            for (const Record &rec : hit.records) {
                auto beg = coverage.begin() + rec.path.abpos, end = coverage.begin() + rec.path.aepos;
                std::for_each(beg, end, [](unsigned &x) { ++x; });
                hit.covScore = std::accumulate(beg, end, 0.0, invertedSum);   // (the last record's range wins, :296)
            }
        }
        std::stable_sort(hits.begin(), hits.end(), [](const TargetHit &l, const TargetHit &r) { return l.covScore > r.covScore; });
    }
};

const char ToU[8] = {'A', 'C', 'G', 'T', '.', '[', ']', '-'};                                                   // :18

int code_of(char c) { return c == 'A' || c == 'a' ? 0 : c == 'C' || c == 'c' ? 1 : c == 'G' || c == 'g' ? 2 : c == 'T' || c == 't' ? 3 : 4; }

// decodeAlignment (:383-417).  a and b are whole reads here (the reference loads windows of them with a
// border and offsets the pointers back, :324-347); past their ends the reference reads the buffer's
// terminator, code 4 = '.'.
void decodeAlignment(const std::string &a, const std::string &b, const Record &rec, std::string &tstr, std::string &qstr) {
    auto A = [&](int i) { return i >= 0 && i < (int)a.size() ? ToU[code_of(a[(size_t)i])] : '.'; };
    auto B = [&](int j) { return j >= 0 && j < (int)b.size() ? ToU[code_of(b[(size_t)j])] : '.'; };
    int i = rec.path.abpos, j = rec.path.bbpos;
    for (int p : rec.trace) {
        if (p < 0) {
            p = -p;
            while (i != p && i < (int)a.size() + 2) { tstr += A(i++); qstr += B(j++); }
            tstr += ToU[7]; qstr += B(j++);
        } else {
            while (j != p && j < (int)b.size() + 2) { tstr += A(i++); qstr += B(j++); }
            tstr += A(i++); qstr += ToU[7];
        }
    }
    const int p = rec.path.aepos;
    while (i <= p) { tstr += A(i++); qstr += B(j++); }
}

std::string complement_seq(const std::string &s) {        // DAZZ_DB Complement_Seq: reverse complement
    std::string r(s.rbegin(), s.rend());
    for (char &c : r) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
    return r;
}

struct Aln { unsigned start; std::string q, t; bool job = false; int aread = 0, bread = 0; unsigned flags = 0; Path path; };
struct TargetData { int id; std::string seq; std::vector<Aln> alns; };

}  // namespace

int main(int argc, char **argv) {
    Opts o;
    if (int rc = parse_args(argc, argv, o)) return rc;
    auto ends_with = [](const std::string &x, const char *suf) { const size_t n = strlen(suf); return x.size() >= n && x.compare(x.size() - n, n, suf) == 0; };
    const bool las_in = ends_with(o.aln_file, ".las"), db_in = ends_with(o.seq_file, ".db");
    // ---- reads (-s) ----
    std::vector<std::string> reads;       // index = 0-based read id
    if (db_in) {
        // Open_DB + Trim_DB (DazAlnProvider.cpp:34-41); every read as ToU[Load_Subread(...)] would give it (:125-130)
        daz::Db db;
        std::string err;
        if (!db.open(o.seq_file, &err)) { fprintf(stderr, "dazcon: %s\n", err.c_str()); return 1; }
        db.trim();
        reads.resize(db.size());
        for (size_t i = 0; i < db.size(); i++) reads[i] = db.read(i);
    } else {
        std::ifstream in(o.seq_file);
        if (!in) { fprintf(stderr, "dazcon: error opening sequence file: %s\n", o.seq_file.c_str()); return 1; }
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            std::istringstream ss(line);
            long id; std::string seq;
            if (!(ss >> id >> seq) || id < 1) { fprintf(stderr, "dazcon: bad line in %s\n", o.seq_file.c_str()); return 1; }
            if ((size_t)id > reads.size()) reads.resize((size_t)id);
            reads[(size_t)id - 1] = seq;
        }
    }
    // ---- overlaps (-a), grouped by A-read as DazAlnProvider::nextTarget does (:79-117) ----
    std::ifstream in;
    if (!las_in) {
        in.open(o.aln_file);
        if (!in) { fprintf(stderr, "dazcon: error opening alignment file: %s\n", o.aln_file.c_str()); return 1; }
    }
    std::vector<TargetData> out_targets;
    size_t n_jobs = 0;
    Target trg;
    auto finish_target = [&]() -> int {
        if (trg.id < 0) return 0;
        const int tid = trg.id + 1;
        if (!o.targets.empty() && !o.targets.count(tid)) return 0;                  // :93
        // getAlignments (:304-369)
        trg.sortHits(o.sort_cov);
        const size_t nh = trg.hits.size() > o.max_hits ? o.max_hits : trg.hits.size();
        TargetData td;
        td.id = tid;
        for (size_t h = 0; h < nh; h++) {
            const TargetHit &hit = trg.hits[h];
            if (o.dump_hits)
                printf("%d\t%d\t%u\t%.9g\t%.9g\t%zu\n", tid, hit.bread + 1, hit.flags, (double)hit.ovlScore, (double)hit.covScore, hit.records.size());
            for (const Record &rec : hit.records) {
                Aln al;
                al.start = (unsigned)rec.path.abpos + 1;                                 // :358
                if (rec.decoded) { al.t = rec.tstr; al.q = rec.qstr; }
                else if (rec.realign) { al.job = true; al.aread = rec.aread; al.bread = rec.bread; al.flags = rec.flags; al.path = rec.path; n_jobs++; }
                else {
                    if ((size_t)rec.bread >= reads.size() || reads[(size_t)rec.bread].empty()) {
                        fprintf(stderr, "dazcon: read %d is not in %s\n", rec.bread + 1, o.seq_file.c_str());
                        return 1;
                    }
                    const std::string &bs = reads[(size_t)rec.bread];
                    decodeAlignment(reads[(size_t)trg.id], (rec.flags & 1u) ? complement_seq(bs) : bs, rec, al.t, al.q);
                }
                td.alns.push_back(std::move(al));
            }
        }
        if (td.alns.size() < o.min_cov) { for (const Aln &al : td.alns) n_jobs -= al.job; return 0; }   // :98-101
        td.seq = reads[(size_t)trg.id];                                                // :119-132 (the A-read itself)
        out_targets.push_back(std::move(td));
        return 0;
    };
    // a record joins its target (DazAlnProvider::nextTarget, :79-117); `where` names it in messages
    auto take = [&](Record &&r, const std::string &where) -> int {
        if (r.aread < 0 || (size_t)r.aread >= reads.size() || reads[(size_t)r.aread].empty()) {
            fprintf(stderr, "dazcon: read %d is not in %s\n", r.aread + 1, o.seq_file.c_str());
            return 1;
        }
        if (r.path.abpos < 0 || r.path.aepos < r.path.abpos || r.path.aepos > (int)reads[(size_t)r.aread].size()) {
            fprintf(stderr, "dazcon: %s: A interval outside the read\n", where.c_str());
            return 1;
        }
        if (r.realign) {
            if (r.bread < 0 || (size_t)r.bread >= reads.size() || r.path.bbpos < 0 || r.path.bepos < r.path.bbpos ||
                r.path.bepos > (int)reads[(size_t)r.bread].size()) {
                fprintf(stderr, "dazcon: %s: B read or interval outside the database\n", where.c_str());
                return 1;
            }
        }
        if (r.aread != trg.id) {                                                     // :90: the A-read changes
            if (int rc = finish_target()) return rc;
            trg = Target();
            trg.id = r.aread;                                                        // firstRecord (:229-249)
            trg.length = (int)reads[(size_t)r.aread].size();
        }
        const int blen = (size_t)r.bread < reads.size() ? (int)reads[(size_t)r.bread].size() : 0;
        trg.addRecord(std::move(r), blen, o.proper);
        return 0;
    };
    if (las_in) {
        // Read_Overlap / Read_Trace (DazAlnProvider.cpp:134-140); the trace points are read and passed over: the
        // alignment between the overlap's end points is the device aligner's (daz_io.h)
        daz::LasReader las;
        std::string err;
        if (!las.open(o.aln_file, &err)) { fprintf(stderr, "dazcon: %s\n", err.c_str()); return 1; }
        daz::Overlap ov;
        std::vector<uint16_t> tr;
        while (las.next(&ov, &tr, &err)) {
            Record r;
            r.aread = ov.aread; r.bread = ov.bread; r.flags = ov.flags; r.realign = true;
            r.path.abpos = ov.abpos; r.path.aepos = ov.aepos; r.path.bbpos = ov.bbpos; r.path.bepos = ov.bepos; r.path.diffs = ov.diffs;
            if (int rc = take(std::move(r), "overlap " + std::to_string(las.seen))) return rc;
        }
        if (!err.empty()) { fprintf(stderr, "dazcon: %s\n", err.c_str()); return 1; }
        if (int rc = finish_target()) return rc;
    } else {
        std::string line;
        unsigned long long ln = 0;
        while (std::getline(in, line)) {
            ln++;
            if (line.empty()) continue;
            std::istringstream ss(line);
            char kind;
            Record r;
            long ar, br;
            if (!(ss >> kind >> ar >> br >> r.flags >> r.path.abpos >> r.path.aepos >> r.path.bbpos >> r.path.bepos >> r.path.diffs) ||
                (kind != 'O' && kind != 'R') || ar < 1 || br < 1) {
                fprintf(stderr, "dazcon: format error in %s line %llu\n", o.aln_file.c_str(), ln);
                return 1;
            }
            r.aread = (int)ar - 1; r.bread = (int)br - 1;
            if (kind == 'O') {
                r.decoded = true;
                if (!(ss >> r.tstr >> r.qstr) || r.tstr.size() != r.qstr.size()) {
                    fprintf(stderr, "dazcon: format error in %s line %llu: alignment strings\n", o.aln_file.c_str(), ln);
                    return 1;
                }
            } else {
                std::string tr;
                ss >> tr;
                if (tr != "-" && !tr.empty()) {
                    const char *p = tr.c_str();
                    while (*p) {
                        char *e = nullptr;
                        r.trace.push_back((int)strtol(p, &e, 10));
                        if (e == p) { fprintf(stderr, "dazcon: format error in %s line %llu: trace\n", o.aln_file.c_str(), ln); return 1; }
                        p = *e == ',' ? e + 1 : e;
                    }
                }
            }
            if (int rc = take(std::move(r), "line " + std::to_string(ln))) return rc;
        }
        if (int rc = finish_target()) return rc;
    }
    if (o.dump_hits) return 0;
    if (o.dump_alns && !n_jobs) {
        for (const TargetData &td : out_targets)
            for (const Aln &al : td.alns) printf("%d\t%u\t%s\t%s\n", td.id, al.start, al.t.c_str(), al.q.c_str());
        return 0;
    }

    // ---- Consensus (dazcon.cpp:61-107) on the device, batches of targets, real backbones ----
    dagcon_ctx *ctx = nullptr;
    dagcon_opts dopt;
    dagcon_default_opts(&dopt);
    dopt.min_cov = o.min_cov; dopt.min_len = o.min_len; dopt.trim = o.trim;
    dopt.min_weight = (int32_t)o.min_cov;          // dazcon.cpp:89
    dopt.device = o.device;
    int rc = dagcon_create(&dopt, &ctx);
    if (rc != DAGCON_OK) {
        fprintf(stderr, "dazcon: no usable MI355X as device %d (dagcon_create = %d); there is no CPU fallback\n", o.device, rc);
        return 1;
    }
    if (n_jobs) {
        // the overlaps of a .las: A[abpos, aepos) against B[bbpos, bepos) (B complemented when COMP(flags), its
        // coordinates are the complement's: DazAlnProvider.cpp:333-347), global between the end points, in groups of
        // at most 256 MB of sequence
        std::vector<Aln *> jobs;
        for (TargetData &td : out_targets) for (Aln &al : td.alns) if (al.job) jobs.push_back(&al);
        size_t j0 = 0;
        uint32_t dropped = 0;
        while (j0 < jobs.size()) {
            std::string qb, tb;
            std::vector<uint64_t> qo, to, oo;
            std::vector<uint32_t> ql, tl;
            uint64_t room = 0;
            size_t j1 = j0;
            for (; j1 < jobs.size() && qb.size() + tb.size() < (256u << 20); j1++) {
                const Aln &al = *jobs[j1];
                const std::string &bs = reads[(size_t)al.bread];
                const std::string bq = (al.flags & daz::COMP_FLAG) ? complement_seq(bs) : bs;
                qo.push_back(qb.size()); ql.push_back((uint32_t)(al.path.bepos - al.path.bbpos));
                qb.append(bq, (size_t)al.path.bbpos, (size_t)(al.path.bepos - al.path.bbpos));
                to.push_back(tb.size()); tl.push_back((uint32_t)(al.path.aepos - al.path.abpos));
                tb.append(reads[(size_t)al.aread], (size_t)al.path.abpos, (size_t)(al.path.aepos - al.path.abpos));
                oo.push_back(room); room += (uint64_t)ql.back() + tl.back();
            }
            std::string qa((size_t)room + 1, '\0'), ta((size_t)room + 1, '\0');
            std::vector<uint32_t> alen(j1 - j0, 0);
            rc = dagcon_align(ctx, (uint32_t)(j1 - j0), qo.data(), ql.data(), to.data(), tl.data(), qb.data(), qb.size(), tb.data(), tb.size(),
                              oo.data(), &qa[0], &ta[0], alen.data());
            if (rc != DAGCON_OK) { fprintf(stderr, "dazcon: alignment failed (%d): %s\n", rc, dagcon_last_error(ctx)); dagcon_destroy(ctx); return 1; }
            dropped += dagcon_align_dropped(ctx);
            for (size_t j = j0; j < j1; j++) {
                jobs[j]->q.assign(qa, (size_t)oo[j - j0], alen[j - j0]);
                jobs[j]->t.assign(ta, (size_t)oo[j - j0], alen[j - j0]);
            }
            j0 = j1;
        }
        if (dropped) fprintf(stderr, "dazcon: warning: %u overlaps could not be aligned inside the widest band and were dropped\n", dropped);
        if (o.dump_alns) {
            for (const TargetData &td : out_targets)
                for (const Aln &al : td.alns) printf("%d\t%u\t%s\t%s\n", td.id, al.start, al.t.c_str(), al.q.c_str());
            dagcon_destroy(ctx);
            return 0;
        }
    }
    int fake_well_counter = 0;                     // dazcon.cpp:62 reads it uninitialised (Q4)
    int status = 0;
    for (size_t t0 = 0; t0 < out_targets.size() && !status; t0 += o.batch_targets) {
        const size_t t1 = std::min(out_targets.size(), t0 + o.batch_targets);
        std::vector<uint32_t> tlen, start, len;
        std::vector<uint64_t> begin{0}, off, bb_off;
        std::string q, t, bb;
        for (size_t x = t0; x < t1; x++) {
            const TargetData &td = out_targets[x];
            tlen.push_back((uint32_t)td.seq.size());
            bb_off.push_back(bb.size());
            bb += td.seq;
            for (const Aln &al : td.alns) {
                start.push_back(al.start); off.push_back(q.size()); len.push_back((uint32_t)al.q.size());
                q += al.q; t += al.t;
            }
            begin.push_back(start.size());
        }
        dagcon_batch db;
        memset(&db, 0, sizeof db);
        db.n_targets = (uint32_t)(t1 - t0);
        db.tlen = tlen.data(); db.aln_begin = begin.data(); db.aln_start = start.data();
        db.aln_off = off.data(); db.aln_len = len.data(); db.qstr = q.data(); db.tstr = t.data(); db.blob_bytes = q.size();
        db.backbone = bb.data(); db.backbone_off = bb_off.data();
        dagcon_results r;
        rc = dagcon_consensus(ctx, &db, &r);
        if (rc != DAGCON_OK) { fprintf(stderr, "dazcon: consensus failed (%d): %s\n", rc, dagcon_last_error(ctx)); status = 1; break; }
        for (uint32_t g = 0; g < r.n_targets; g++) {
            const TargetData &td = out_targets[t0 + g];
            if (o.verbose) fprintf(stderr, "(0) calling: %d Alignments: %zu\n", td.id, td.alns.size());
            if (r.target_status[g] != DAGCON_OK)
                fprintf(stderr, "dazcon: warning: target %d skipped (non-conforming alignment or internal error %d)\n", td.id, r.target_status[g]);
            for (uint64_t s = r.seg_begin[g]; s < r.seg_begin[g + 1]; s++) {
                // dazcon.cpp:92-97  ">%s/%d/%d_%d\n%s\n"
                printf(">%d/%d/%d_%d\n", td.id, fake_well_counter, r.range0[s], r.range1[s]);
                fwrite(r.seq_blob + r.seq_off[s], 1, r.seq_len[s], stdout);
                fputc('\n', stdout);
                ++fake_well_counter;
            }
        }
    }
    dagcon_destroy(ctx);
    fflush(stdout);
    return status;
}
