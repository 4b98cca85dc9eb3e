#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run in the build container).

 a1_vectors.json   inputs and outputs of the REFERENCE's own normalizeGaps /
                   trimAln / parseM5 (src/cpp/Alignment.cpp compiled in place into
                   oracle/_ref/libref_alignment.so) on seeded random alignment
                   strings and on the reference's .m5 fixture lines.  These pin
                   stage a1 and the parser to the real code.
 kat_graph.json    the reference's own known-answer tests for the graph stages
                   (test/cpp/AlnGraphBoostTest.cpp:11-57), as data.
 config1.json      BASELINE configs[0] (1 kb x 20x, seed 1): input batch digest and
                   the oracle's FASTA.  The graph part of the reference cannot be
                   built here (Boost.Graph absent), so this one is an oracle
                   self-consistency fixture, pinned to the reference only through
                   kat_graph.json -- DESIGN.md says so.

The fixtures are data (inputs + expected outputs); no reference source text is
stored.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402


def main():
    oracle.build()
    assert oracle.ref_lib() is not None, "needs /root/reference (oracle/_ref)"
    rng = np.random.default_rng(20261004)
    vec = []
    alphabets = [b"ACGT-", b"AC-", b"ACGT-.", b"ACGTN-", b"A-"]
    for i in range(400):
        n = int(rng.integers(1, 90))
        al = alphabets[i % len(alphabets)]
        q = bytes(al[j] for j in rng.integers(0, len(al), n))
        t = bytes(al[j] for j in rng.integers(0, len(al), n))
        qn, tn = oracle.ref_normalize_gaps(q, t)
        trim = int(rng.integers(0, 12))
        start = int(rng.integers(1, 50))
        qt, tt, st = oracle.ref_trim_aln(qn, tn, start, trim)
        vec.append(dict(q=q.decode(), t=t.decode(), qn=qn.decode(), tn=tn.decode(),
                        trim=trim, start=start, qt=qt.decode(), tt=tt.decode(), start_t=st))
    # realistic ones: synthetic reads (sub/ins/del) including homopolymer runs
    from util import random_target
    for i in range(60):
        alns, _ = random_target(rng, int(rng.integers(20, 300)), 1, alphabet=[b"ACGT", b"AC"][i % 2],
                                sub=0.05, ins=0.15, dele=0.08)
        s, q, t = alns[0]
        qn, tn = oracle.ref_normalize_gaps(q, t)
        trim = int(rng.integers(0, 30))
        qt, tt, st = oracle.ref_trim_aln(qn, tn, s, trim)
        vec.append(dict(q=q.decode(), t=t.decode(), qn=qn.decode(), tn=tn.decode(),
                        trim=trim, start=s, qt=qt.decode(), tt=tt.decode(), start_t=st))
    parsed = []
    for fn in ("basic.m5", "parsequery.m5"):
        for line in open(f"/root/reference/test/cpp/{fn}", "rb").read().split(b"\n"):
            if not line:
                continue
            for gbt in (True, False):
                r = oracle.ref_parse_m5(line, gbt)
                parsed.append(dict(line=line.decode(), group_by_target=gbt,
                                   **{k: (v.decode() if isinstance(v, bytes) else v) for k, v in r.items()}))
    json.dump(dict(source="reference Alignment.cpp via oracle/_ref/libref_alignment.so",
                   vectors=vec, parsed=parsed), open(os.path.join(HERE, "a1_vectors.json"), "w"), indent=0)

    # .pre lines (Alignment.cpp:82-112 parsePre) through the reference's own parser
    pre = []
    for i in range(40):
        n, m = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        q = bytes(b"ACGT"[j] for j in rng.integers(0, 4, n)).decode()
        t = bytes(b"ACGTN"[j] for j in rng.integers(0, 5, m)).decode()
        tlen = int(rng.integers(m, 5000))
        ts = int(rng.integers(0, tlen - m + 1))
        sep = " " if i % 5 else "  "                 # empty fields are skipped (Alignment.cpp:89-92)
        line = sep.join([f"q{i}/0_{n}", f"t{i % 7}", "+-"[i % 2], str(tlen), str(ts), str(ts + m), q, t])
        r = oracle.ref_parse_pre(line.encode())
        pre.append(dict(line=line, **{k: (v.decode() if isinstance(v, bytes) else v) for k, v in r.items()}))
    json.dump(dict(source="reference Alignment.cpp parsePre via oracle/_ref/libref_alignment.so", parsed=pre),
              open(os.path.join(HERE, "pre_vectors.json"), "w"), indent=0)

    kat = dict(
        raw_consensus=dict(backbone="ATATTAGGC", start=1, expected="ATATAGCCGGC", alignments=[
            dict(t="ATATTA---GGC", q="ATAT-AGCCGGC"), dict(t="ATATTA-GGC", q="ATAT-ACGGC"),
            dict(t="AT-ATTA--GGC", q="ATCAT--CCGGC"), dict(t="ATATTA--G-GC", q="ATAT-ACCGAG-"),
            dict(t="ATATTA---GGC", q="ATAT-AGCCGGC")]),
        dangling_nodes=dict(blen=12, start=0, t="C-GCGGA-T-G-", q="CCGCGG-G-A-T", expected=False),
        normalize=[dict(q="CAC", t="CGC", qn="C-AC", tn="CG-C"),
                   dict(q="-C--CGT", t="CCGAC-T", qn="CCG--T", tn="CCGACT"),
                   dict(q="ATAT-AGCCGGC", t="ATATTA---GGC", qn="ATAT-AGCCGGC", tn="ATATTAG--G-C"),
                   dict(q="CAACAT", t="C-A-AT", qn="CAACAT", tn="CAA--T")],
        trim=dict(t="ACG-TCA-GCA", q="AC-C-C-T---", start=1, cases=[
            dict(trim=0, start=1, t="ACG-TCA-GCA", q="AC-C-C-T---"),
            dict(trim=3, start=4, t="-TCA-", q="C-C-T"), dict(trim=4, start=5, t="C", q="C"),
            dict(trim=5, start=6, t="", q=""), dict(trim=500, start=None, t="", q="")]),
        simple_aligner=dict(start=765, end=826, tlen=2092, strand="-",
                            tstr="ACAGAGATGCAAGGTAAAGTACAATTGAAAAACTAACCTCTTCCAGCGAGACTTATAGCGA",
                            qstr="ACAGAAGATGAAGGTAAATACAATGAAAAAACTACCTCGGTTCCAGCGAGAACTATAGCGA",
                            expected_tstr="TCGCTATAAGT-CTCGCTGGAA--GAGGTTAGTTTTT-CAATTGTACTTTACCTTGCATCT-CTGT",
                            expected_start=1267),
        source="test/cpp/AlnGraphBoostTest.cpp:11-57, test/cpp/AlignmentTest.cpp:19-143, "
               "test/cpp/SimpleAlignerTest.cpp:8-21, src/tests/test_aligngraph.py:50-54")
    json.dump(kat, open(os.path.join(HERE, "kat_graph.json"), "w"), indent=1)

    from pbdagcon_amd import synth
    from util import oracle_batch
    b = synth.make_batch(1, 1000, 20, seed=1)
    segs = oracle_batch(b, 6, 500, 50)[0]
    fasta = "".join(f">{b.ids[0]}/{r0}_{r1}\n{s.decode()}\n" for r0, r1, s in segs)
    json.dump(dict(workload="configs[0]: 1 target x 1000 bp x 20x, seed 1, -c 6 -m 500 -t 50",
                   input_sha256=hashlib.sha256(b.qstr.tobytes() + b.tstr.tobytes()).hexdigest(),
                   n_columns=int(b.qstr.size), fasta=fasta,
                   pinned_by="oracle (graph stages of the reference are not buildable here); "
                             "the oracle itself is pinned by kat_graph.json"),
              open(os.path.join(HERE, "config1.json"), "w"), indent=1)
    print("wrote a1_vectors.json kat_graph.json config1.json")


if __name__ == "__main__":
    main()
