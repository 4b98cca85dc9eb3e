// daz_io.h -- readers for the two binary inputs of the reference's dazcon: a DALIGNER .las file of overlaps and a
// DAZZ_DB database (.db stub + hidden .idx / .bps files).
//
// PARITY UNPINNED.  The reference reads them through DALIGNER's and DAZZ_DB's own C sources (Open_DB, Trim_DB,
// Load_Subread, Read_Overlap, Read_Trace: DazAlnProvider.cpp:34-66, 134-139, 322-351), which are empty submodules here,
// and it holds no fixture at this boundary.  What is written below follows what the reference's own code shows of the
// layout -- the .las header is an int64 overlap count and an int trace spacing, trace elements are one byte up to
// TRACE_XOVR = 125 and two above it (DazAlnProvider.cpp:49-63) -- and, for the rest, the published on-disk layout of
// those libraries (Overlap minus its leading pointer, 40 bytes; HITS_DB / HITS_READ images in the .idx; four bases a byte,
// first base in the top bits, in the .bps).  It is tested by round trip against a writer of this build's own
// (tests/daz_files.py) -- never against a file DALIGNER or DAZZ_DB wrote.
//
// The reference then recomputes each overlap's alignment inside the trace-point panels (Compute_Trace_PTS,
// DazAlnProvider.cpp:349) -- also absent.  dazcon_main.cpp aligns the overlap's two intervals with the device aligner
// (dagcon_align, the -a stage of pbdagcon) instead: same endpoints, this build's own banded alignment between them.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace daz {

constexpr int TRACE_XOVR = 125;        // DALIGNER align.h; DazAlnProvider.cpp:56
constexpr uint32_t COMP_FLAG = 0x1;    // COMP(flags): B is complemented (DazAlnProvider.cpp:333)
constexpr int DB_BEST = 0x0800;        // DAZZ_DB DB.h: the read of its well that Trim_DB keeps when not `all`

struct Overlap {                       // DALIGNER's Overlap without the pointer in front: the 40 bytes Read_Overlap reads
    int32_t tlen, diffs, abpos, bbpos, aepos, bepos;
    uint32_t flags;
    int32_t aread, bread;
    int32_t pad;
};
static_assert(sizeof(Overlap) == 40, "on-disk overlap record");

struct LasReader {
    FILE *f = nullptr;
    int64_t novl = 0, seen = 0;
    int32_t tspace = 0;
    int tbytes = 1;
    bool open(const std::string &path, std::string *err) {
        f = fopen(path.c_str(), "rb");
        if (!f) { *err = "Open failed: " + path; return false; }                              // DazAlnProvider.cpp:43-46
        if (fread(&novl, sizeof(int64_t), 1, f) != 1) { *err = "Failed to read novl"; return false; }      // :48-49
        if (fread(&tspace, sizeof(int32_t), 1, f) != 1) { *err = "Failed to read tspace"; return false; }  // :52-53
        tbytes = tspace <= TRACE_XOVR ? 1 : 2;                                                // :55-62
        if (novl < 0 || tspace <= 0) { *err = "not a .las file: " + path; return false; }
        return true;
    }
    // one overlap and its trace points (pairs: differences, B bases, per panel of tspace A bases)
    bool next(Overlap *o, std::vector<uint16_t> *trace, std::string *err) {
        if (seen >= novl) return false;
        if (fread(o, sizeof(Overlap), 1, f) != 1) { *err = "truncated .las (overlap record)"; return false; }
        if (o->tlen < 0 || o->tlen > (1 << 28)) { *err = "corrupt .las (trace length)"; return false; }
        trace->resize((size_t)o->tlen);
        if (tbytes == 1) {
            std::vector<uint8_t> b((size_t)o->tlen);
            if (o->tlen && fread(b.data(), 1, b.size(), f) != b.size()) { *err = "truncated .las (trace)"; return false; }
            for (size_t i = 0; i < b.size(); i++) (*trace)[i] = b[i];
        } else if (o->tlen && fread(trace->data(), 2, trace->size(), f) != trace->size()) { *err = "truncated .las (trace)"; return false; }
        seen++;
        return true;
    }
    ~LasReader() { if (f) fclose(f); }
};

// images of DAZZ_DB's HITS_DB and HITS_READ as they sit in the .idx (LP64: pointers are 8 bytes, padded as the
// compiler pads the C structs)
struct HitsDb {
    int32_t ureads, treads, cutoff, all;
    float freq[4];
    int32_t maxlen, pad0;
    int64_t totlen;
    int32_t nreads, trimmed, part, ufirst, tfirst, pad1;
    uint64_t path;
    int32_t loaded, pad2;
    uint64_t bases, reads, tracks;
};
static_assert(sizeof(HitsDb) == 112, "HITS_DB image");
struct HitsRead {
    int32_t origin, rlen, fpulse, pad0;
    int64_t boff, coff;
    int32_t flags, pad1;
};
static_assert(sizeof(HitsRead) == 40, "HITS_READ image");

struct Db {
    HitsDb hdr;
    std::vector<HitsRead> reads;       // after trim(): the reads daligner numbered
    std::vector<uint8_t> bps;
    static std::string hidden(const std::string &db_path, const char *ext) {
        const size_t slash = db_path.find_last_of('/');
        const std::string dir = slash == std::string::npos ? "" : db_path.substr(0, slash + 1);
        std::string root = slash == std::string::npos ? db_path : db_path.substr(slash + 1);
        if (root.size() > 3 && root.compare(root.size() - 3, 3, ".db") == 0) root.resize(root.size() - 3);
        return dir + "." + root + ext;
    }
    bool open(const std::string &db_path, std::string *err) {                                 // Open_DB, whole database
        FILE *f = fopen(db_path.c_str(), "r");
        if (!f) { *err = "Failed to open DB"; return false; }                                 // DazAlnProvider.cpp:35-37
        fclose(f);
        const std::string idx = hidden(db_path, ".idx"), bp = hidden(db_path, ".bps");
        f = fopen(idx.c_str(), "rb");
        if (!f) { *err = "Failed to open DB (" + idx + ")"; return false; }
        bool ok = fread(&hdr, sizeof hdr, 1, f) == 1 && hdr.ureads >= 0 && hdr.ureads < (1 << 30);
        if (ok) { reads.resize((size_t)hdr.ureads); ok = hdr.ureads == 0 || fread(reads.data(), sizeof(HitsRead), reads.size(), f) == reads.size(); }
        fclose(f);
        if (!ok) { *err = "corrupt DB index " + idx; return false; }
        f = fopen(bp.c_str(), "rb");
        if (!f) { *err = "Failed to open DB (" + bp + ")"; return false; }
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        bps.resize(n > 0 ? (size_t)n : 0);
        ok = bps.empty() || fread(bps.data(), 1, bps.size(), f) == bps.size();
        fclose(f);
        if (!ok) { *err = "cannot read " + bp; return false; }
        for (const HitsRead &r : reads)
            if (r.rlen < 0 || r.boff < 0 || (uint64_t)r.boff + ((uint64_t)r.rlen + 3) / 4 > bps.size()) { *err = "corrupt DB index (read beyond the bases file)"; return false; }
        return true;
    }
    // Trim_DB (DazAlnProvider.cpp:41): the reads shorter than the cutoff go, and unless `all` every read but the best
    // of its well; the survivors are renumbered -- those are the numbers a .las carries
    void trim() {
        if (hdr.trimmed) return;
        if (hdr.cutoff <= 0 && hdr.all) return;
        const int need = hdr.all ? 0 : DB_BEST;
        std::vector<HitsRead> keep;
        for (const HitsRead &r : reads)
            if ((r.flags & need) == need && r.rlen >= hdr.cutoff) keep.push_back(r);
        reads.swap(keep);
        hdr.trimmed = 1;
    }
    size_t size() const { return reads.size(); }
    // Load_Subread(db, i, 0, rlen, buf, 0) followed by ToU (DazAlnProvider.cpp:125-130): the read as A, C, G, T
    std::string read(size_t i) const {
        const HitsRead &r = reads[i];
        std::string s((size_t)r.rlen, 'A');
        for (int32_t k = 0; k < r.rlen; k++) s[(size_t)k] = "ACGT"[(bps[(size_t)r.boff + (size_t)k / 4] >> (6 - 2 * (k & 3))) & 3];
        return s;
    }
};

}  // namespace daz
