// ref_glue.cpp -- extern "C" handles onto the reference's own Alignment.cpp.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with
// /root/reference/src/cpp/Alignment.cpp, read in place (never copied), into
// oracle/_ref/libref_alignment.so by oracle/Makefile.  It exists to pin the
// oracle's restatement of normalizeGaps / trimAln / parseM5 / revComp against
// the real code.  AlnGraphBoost.cpp cannot be built the same way: it needs
// Boost.Graph, which this image does not have.
#include <cstring>
#include <istream>
#include <sstream>
#include <string>
#include "Alignment.hpp"

extern "C" {

// normalizeGaps (Alignment.cpp:131-217).  Outputs need capacity 2*len+1.
size_t ref_normalize_gaps(const char* q, const char* t, size_t len, int push,
                          char* qout, char* tout) {
    dagcon::Alignment a;
    a.qstr.assign(q, len);
    a.tstr.assign(t, len);
    dagcon::Alignment n = normalizeGaps(a, push != 0);
    std::memcpy(qout, n.qstr.data(), n.qstr.size());
    std::memcpy(tout, n.tstr.data(), n.tstr.size());
    qout[n.qstr.size()] = 0;
    tout[n.tstr.size()] = 0;
    return n.qstr.size();
}

// trimAln (Alignment.cpp:219-242).  In place on NUL-terminated buffers;
// returns the new length and updates *start.
size_t ref_trim_aln(char* q, char* t, size_t len, int trim_len, uint32_t* start) {
    dagcon::Alignment a;
    a.qstr.assign(q, len);
    a.tstr.assign(t, len);
    a.start = *start;
    trimAln(a, trim_len);
    std::memcpy(q, a.qstr.data(), a.qstr.size());
    std::memcpy(t, a.tstr.data(), a.tstr.size());
    q[a.qstr.size()] = 0;
    t[a.tstr.size()] = 0;
    *start = a.start;
    return a.qstr.size();
}

// parseM5 (Alignment.cpp:44-80) through operator>> on one line.
// Buffers must be large enough for the line.  Returns 1.
int ref_parse_m5(const char* line, size_t len, int group_by_target,
                 char* id, char* sid, char* qstr, char* tstr,
                 uint32_t* tlen, uint32_t* start, char* strand) {
    dagcon::Alignment::groupByTarget = group_by_target != 0;
    dagcon::Alignment::parse = parseM5;
    std::istringstream in(std::string(line, len));
    dagcon::Alignment a;
    in >> a;
    std::strcpy(id, a.id.c_str());
    std::strcpy(sid, a.sid.c_str());
    std::strcpy(qstr, a.qstr.c_str());
    std::strcpy(tstr, a.tstr.c_str());
    *tlen = a.tlen;
    *start = a.start;
    *strand = a.strand;
    dagcon::Alignment::groupByTarget = true;
    return 1;
}

// parsePre (Alignment.cpp:82-112) through operator>> on one line.
int ref_parse_pre(const char* line, size_t len, char* id, char* sid, char* qstr, char* tstr,
                  uint32_t* tlen, uint32_t* start, uint32_t* end, char* strand) {
    dagcon::Alignment::parse = parsePre;
    std::istringstream in(std::string(line, len));
    dagcon::Alignment a;
    in >> a;
    dagcon::Alignment::parse = parseM5;
    std::strcpy(id, a.id.c_str());
    std::strcpy(sid, a.sid.c_str());
    std::strcpy(qstr, a.qstr.c_str());
    std::strcpy(tstr, a.tstr.c_str());
    *tlen = a.tlen;
    *start = a.start;
    *end = a.end;
    *strand = a.strand;
    return 1;
}

// revComp (Alignment.cpp:15-26).  In place.
void ref_revcomp(char* seq, size_t len) {
    std::string s(seq, len);
    std::string r = revComp(s);
    std::memcpy(seq, r.data(), len);
}

}  // extern "C"
