"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Integer / byte / index work: everything must be bit-exact.  The one floating
point quantity (bestPath scores, fp32) only decides comparisons; its values
are exact multiples of 0.5, so there is no tolerance anywhere in this file.
"""
import numpy as np
import pytest

import oracle
from oracle import pymodel
from pbdagcon_amd import capi, synth
from pbdagcon_amd.consensus import Alignment, AlnGraphBoost, normalizeGaps, trimAln
from util import batch_from_targets, oracle_batch, random_target

pytestmark = pytest.mark.gpu


def _digest(results):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "make_large_hashes", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_large_hashes.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.digest(results)


def _golden(name):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "large_hashes.json")
    return json.load(open(path))[name]["sha256"]


# ---- reference KATs, through the device -------------------------------------

def test_normalize_kat_device():
    """test/cpp/AlignmentTest.cpp:19-47 + src/tests/test_aligngraph.py:50-54."""
    for q, t, qe, te in [
        (b"CAC", b"CGC", b"C-AC", b"CG-C"),
        (b"-C--CGT", b"CCGAC-T", b"CCG--T", b"CCGACT"),
        (b"ATAT-AGCCGGC", b"ATATTA---GGC", b"ATAT-AGCCGGC", b"ATATTAG--G-C"),
        (b"CAACAT", b"C-A-AT", b"CAACAT", b"CAA--T"),
    ]:
        b = normalizeGaps(Alignment(start=1, qstr=q, tstr=t))
        assert (b.qstr, b.tstr) == (qe, te)


def test_trim_kat_device():
    """test/cpp/AlignmentTest.cpp:80-143."""
    t, q = b"ACG-TCA-GCA", b"AC-C-C-T---"
    for trim, start, te, qe in [(0, 1, t, q), (3, 4, b"-TCA-", b"C-C-T"), (4, 5, b"C", b"C"),
                                (5, 6, b"", b""), (500, None, b"", b"")]:
        a = Alignment(start=1, qstr=q, tstr=t, strand="-")
        trimAln(a, trim)
        assert (a.tstr, a.qstr) == (te, qe)
        if start is not None:
            assert a.start == start


def test_raw_consensus_kat_device():
    """test/cpp/AlnGraphBoostTest.cpp:11-47: expected ATATAGCCGGC."""
    ag = AlnGraphBoost(b"ATATTAGGC")
    for t, q in [(b"ATATTA---GGC", b"ATAT-AGCCGGC"), (b"ATATTA-GGC", b"ATAT-ACGGC"),
                 (b"AT-ATTA--GGC", b"ATCAT--CCGGC"), (b"ATATTA--G-GC", b"ATAT-ACCGAG-"),
                 (b"ATATTA---GGC", b"ATAT-AGCCGGC")]:
        ag.addAln(Alignment(id="target", tlen=9, start=1, qstr=q, tstr=t))
    ag.mergeNodes()
    assert ag.consensus() == b"ATATAGCCGGC"


# ---- differential tests against the oracle -----------------------------------

def _check_batch(ctx_factory, batch, min_cov=0, min_len=0, trim=0, min_weight=-1, graphs=()):
    ctx = ctx_factory(min_cov=min_cov, min_len=min_len, trim=trim, min_weight=min_weight)
    got = ctx.consensus(batch)
    exp = oracle_batch(batch, min_cov, min_len, trim, min_weight)
    assert len(got) == len(exp)
    for t, (g, e) in enumerate(zip(got, exp)):
        assert g == e, f"target {t}: consensus differs"
    return ctx, got


def _oracle_graph(batch, t, min_len, trim, merge):
    """Oracle graph of target t plus the oracle-id -> device-id map.  The oracle numbers
    inserted vertices in (read, column) order after the backbone (the reference's
    add_vertex order); the device numbers vertices in backbone-position order: the inserted
    vertices whose _bbMap is p in (read, column) order, then backbone vertex p."""
    blen = int(batch.tlen[t])
    g = oracle.Graph(blen=blen) if batch.backbone is None else oracle.Graph(
        backbone=batch.backbone[int(batch.backbone_off[t]):int(batch.backbone_off[t]) + blen].tobytes())
    ins_pos = []
    for s, q, tt in batch.target_alignments(t):
        if len(q) < min_len:
            continue
        qn, tn = oracle.normalize_gaps(q, tt)
        qn, tn, s2 = oracle.trim_aln(qn, tn, s, trim)
        g.add_aln(s2, qn, tn)
        bb = s2
        for a, b in zip(qn, tn):
            if a == b or a == 0x2D:
                bb += 1
            elif b == 0x2D:
                ins_pos.append(bb)
    if merge:
        assert g.merge_nodes() == 0
    gcount = [0] * (blen + 2)
    for p in ins_pos:
        gcount[p] += 1
    gbase, acc = [], 0
    for p in range(blen + 2):
        gbase.append(acc)
        acc += gcount[p] + 1
    o2d = [gbase[p] + gcount[p] for p in range(blen + 2)]
    seen = [0] * (blen + 2)
    for p in ins_pos:
        o2d.append(gbase[p] + seen[p])
        seen[p] += 1
    return g.adjacency(), o2d


def _check_graphs(ctx, batch, trim, merge, targets=None):
    for t in (range(batch.n_targets) if targets is None else targets):
        got = ctx.debug_graph(t)
        exp, o2d = _oracle_graph(batch, t, 0, trim, merge)
        assert len(got) == len(exp) == len(o2d), f"target {t}: vertex count"
        assert sorted(o2d) == list(range(len(got)))
        blen = int(batch.tlen[t])
        for o, (eb, ew, ec, ed, eoe, eie) in enumerate(exp):
            g = got[o2d[o]]
            assert g["deleted"] == ed, f"target {t} vertex {o}: deleted flag"
            assert g["backbone"] == (o < blen + 2)
            if ed:
                continue
            assert (g["base"], g["weight"]) == (eb, ew), f"target {t} vertex {o}: base/weight {g} vs {exp[o]}"
            if o < blen + 2:
                assert g["coverage"] == ec, f"target {t} vertex {o}: coverage"
            assert g["out"] == [(o2d[d], c) for d, c in eoe], f"target {t} vertex {o}: out list"
            assert g["inn"] == [o2d[s] for s, _ in eie], f"target {t} vertex {o}: in list"


@pytest.mark.parametrize("merge", [False, True])
def test_graph_adjacency_matches_oracle(gpu_ctx_factory, merge):
    """Stage a2 (and b) leave exactly the reference's graph: same vertices, same
    weights / coverage / bases, same adjacency ORDER and edge counts."""
    rng = np.random.default_rng(7)
    targets = []
    for i in range(24):
        tl = int(rng.integers(5, 60))
        alph = [b"ACGT", b"AC", b"A"][i % 3]
        alns, bb = random_target(rng, tl, int(rng.integers(1, 9)), alphabet=alph, dots=(i % 5 == 0))
        targets.append((tl, alns, bb))
    batch = batch_from_targets(targets)
    flags = capi.FLAG_STOP_AFTER_MERGE if merge else capi.FLAG_STOP_AFTER_BUILD
    ctx = gpu_ctx_factory(min_cov=0, min_len=0, trim=2, min_weight=0, flags=flags)
    ctx.consensus(batch)
    _check_graphs(ctx, batch, 2, merge)


def test_emit_stretches_on_adversarial_pileups(gpu_ctx_factory, monkeypatch):
    """k_emit threads a read into the graph in stretches of backbone positions, one wave each,
    entering at recorded columns and looking back / ahead for the neighbouring vertex.  With
    16-position stretches every kind of column lands on a stretch edge: deletion runs across
    it, insertion runs in front of it, reads that start or end on it; the built graph and the
    consensus are the oracle's.  Then the production stretch length on longer targets."""
    monkeypatch.setenv("DAGCON_EMIT_SHIFT", "4")
    rng = np.random.default_rng(41)
    targets = []
    for i in range(40):
        tl = int(rng.integers(20, 200))
        alph = [b"ACGT", b"AC", b"A"][i % 3]
        alns, bb = random_target(rng, tl, int(rng.integers(1, 10)), alphabet=alph, sub=0.05,
                                 ins=float(rng.uniform(0.05, 0.3)), dele=float(rng.uniform(0.05, 0.35)),
                                 dots=(i % 6 == 0), full_span=(i % 2 == 0))
        targets.append((tl, alns, bb))
    batch = batch_from_targets(targets)
    ctx = gpu_ctx_factory(min_cov=0, min_len=0, trim=1, min_weight=0, flags=capi.FLAG_STOP_AFTER_BUILD)
    ctx.consensus(batch)
    _check_graphs(ctx, batch, 1, False)
    _check_batch(gpu_ctx_factory, batch, min_cov=0, min_len=0, trim=0, min_weight=0)
    _check_batch(gpu_ctx_factory, batch, min_cov=2, min_len=10, trim=3, min_weight=1)
    monkeypatch.delenv("DAGCON_EMIT_SHIFT")
    targets = []
    for i in range(3):
        tl = int(rng.integers(700, 1700))
        alns, bb = random_target(rng, tl, 7, alphabet=b"ACGT", sub=0.03, ins=0.12, dele=0.2, full_span=(i != 1))
        targets.append((tl, alns, bb))
    batch = batch_from_targets(targets)
    ctx = gpu_ctx_factory(min_cov=0, min_len=0, trim=3, min_weight=0, flags=capi.FLAG_STOP_AFTER_BUILD)
    ctx.consensus(batch)
    _check_graphs(ctx, batch, 3, False)


def test_segmented_sweeps_are_exact(gpu_ctx_factory):
    """mergeNodes and bestPath are split at backbone vertices every read passes through
    (k_cuts) and the pieces are swept concurrently.  1, 3, 8 and 32 pieces per target all give
    the oracle's consensus, also with the one-piece bestPath re-sweep forced: full-span reads
    (cuts everywhere) and partial spans (cuts only where all reads overlap, often none)."""
    full = synth.make_batch(6, 6000, 30, seed=31000)
    part = synth.make_batch(6, 0, 24, seed=32000, min_span=0.5,
                            tlens=np.array([5000, 7000, 9000, 3000, 12000, 6000]))
    for batch, trim in ((full, 50), (part, 10)):
        exp = oracle_batch(batch, 6, 500, trim)
        nseg = {}
        for ms, flags in ((1, 0), (3, 0), (8, 0), (32, 0), (8, capi.FLAG_DEBUG_RESWEEP)):
            ctx = gpu_ctx_factory(min_cov=6, min_len=500, trim=trim, max_segments=ms, flags=flags,
                                  min_segment_len=768)
            assert ctx.consensus(batch) == exp, f"max_segments={ms} flags={flags}"
            nseg[ms] = ctx.timings()["merge_segments"]
        assert nseg[1] == batch.n_targets
        if batch is full:
            assert nseg[8] == 6 * 7 and nseg[3] == 6 * 3       # 6000 / 768 = 7 stretches wanted and found
        assert nseg[32] >= nseg[8] >= nseg[3] >= nseg[1]
        # the default: shorter stretches (down to 192 positions) when the batch is too small to fill the chip
        ctx = gpu_ctx_factory(min_cov=6, min_len=500, trim=trim)
        assert ctx.consensus(batch) == exp
        assert ctx.timings()["merge_segments"] >= nseg[32]


@pytest.mark.parametrize("seed", [11, 12])
def test_segments_on_adversarial_little_pileups(gpu_ctx_factory, seed):
    """Cuts every few bases on tiny-alphabet pileups: merge groups, turned-around edges and
    long insertion runs right next to the cut vertices; graph after merge and consensus."""
    rng = np.random.default_rng(seed)
    targets = []
    for i in range(120):
        tl = int(rng.integers(20, 160))
        alph = [b"AC", b"ACGT", b"A"][i % 3]
        alns, bb = random_target(rng, tl, int(rng.integers(2, 10)), alphabet=alph,
                                 sub=float(rng.uniform(0, 0.1)), ins=float(rng.uniform(0, 0.25)),
                                 dele=float(rng.uniform(0, 0.12)), full_span=(i % 4 != 0))
        targets.append((tl, alns, bb))
    batch = batch_from_targets(targets)
    full = batch_from_targets([x for i, x in enumerate(targets) if i % 4 != 0])      # (k_cuts instead of k_cuts2)
    for kw in [dict(min_cov=0, min_len=0, trim=0, min_weight=0),
               dict(min_cov=2, min_len=10, trim=2, min_weight=1)]:
        for b in (batch, full):
            exp = oracle_batch(b, kw["min_cov"], kw["min_len"], kw["trim"], kw["min_weight"])
            ctx = gpu_ctx_factory(max_segments=16, min_segment_len=4, **kw)
            assert ctx.consensus(b) == exp
            assert ctx.timings()["merge_segments"] > 3 * b.n_targets       # the cuts were really used
    # and the merged graph itself, vertex by vertex
    ctx = gpu_ctx_factory(min_cov=0, min_len=0, trim=0, min_weight=0, max_segments=16, min_segment_len=4,
                          flags=capi.FLAG_STOP_AFTER_MERGE)
    ctx.consensus(batch)
    for t in range(0, batch.n_targets, 5):
        got = ctx.debug_graph(t)
        exp, o2d = _oracle_graph(batch, t, 0, 0, True)
        assert len(got) == len(exp)
        for o, (eb, ew, ec, ed, eoe, eie) in enumerate(exp):
            g = got[o2d[o]]
            assert g["deleted"] == ed
            if ed:
                continue
            assert (g["base"], g["weight"]) == (eb, ew)
            assert g["out"] == [(o2d[d], c) for d, c in eoe], f"target {t} vertex {o}: out list"
            assert g["inn"] == [o2d[s] for s, _ in eie], f"target {t} vertex {o}: in list"


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_small_targets(gpu_ctx_factory, seed):
    """Hundreds of adversarial little pileups: tiny alphabets, ragged spans, long
    insertion runs, dots for gaps; several option settings."""
    rng = np.random.default_rng(seed)
    targets = []
    for i in range(150):
        tl = int(rng.integers(1, 120))
        alph = [b"ACGT", b"AC", b"ACGTN", b"A"][i % 4]
        alns, bb = random_target(rng, tl, int(rng.integers(0, 14)), alphabet=alph,
                                 sub=float(rng.uniform(0, 0.1)), ins=float(rng.uniform(0, 0.3)),
                                 dele=float(rng.uniform(0, 0.2)), dots=(i % 7 == 0),
                                 full_span=(i % 3 == 0))
        targets.append((tl, alns, bb))
    batch = batch_from_targets(targets)
    for kw in [dict(min_cov=0, min_len=0, trim=0, min_weight=0),
               dict(min_cov=3, min_len=10, trim=3, min_weight=-1),
               dict(min_cov=1, min_len=0, trim=1, min_weight=2)]:
        _check_batch(gpu_ctx_factory, batch, **kw)


def test_backbone_given_dazcon_style(gpu_ctx_factory):
    """dazcon.cpp:76: real backbone bases (AlnGraphBoost.cpp:16-39)."""
    rng = np.random.default_rng(11)
    targets = []
    for i in range(40):
        tl = int(rng.integers(20, 200))
        alns, bb = random_target(rng, tl, int(rng.integers(2, 10)), alphabet=b"ACGT")
        targets.append((tl, alns, bb))
    batch = batch_from_targets(targets, with_backbone=True)
    _check_batch(gpu_ctx_factory, batch, min_cov=0, min_len=0, trim=1, min_weight=0)
    _check_batch(gpu_ctx_factory, batch, min_cov=2, min_len=15, trim=5, min_weight=1)


def test_config1_shape(gpu_ctx_factory):
    """BASELINE configs[0]: single 1 kb backbone, 20x reads, default options."""
    batch = synth.make_batch(1, 1000, 20, seed=1)
    _, got = _check_batch(gpu_ctx_factory, batch, min_cov=6, min_len=500, trim=50)
    assert len(got[0]) >= 1


def test_midsize_synthetic(gpu_ctx_factory):
    """64 targets x 2 kb x 24x and mixed lengths / partial spans (config-5 shape, scaled)."""
    b1 = synth.make_batch(64, 2000, 24, seed=500)
    _check_batch(gpu_ctx_factory, b1, min_cov=6, min_len=500, trim=50)
    tl = np.random.default_rng(3).integers(600, 4000, 48)
    b2 = synth.make_batch(48, 0, 16, seed=900, min_span=0.6, tlens=tl)
    _check_batch(gpu_ctx_factory, b2, min_cov=6, min_len=500, trim=10)


# ---- edge cases ---------------------------------------------------------------

def test_empty_and_filtered(gpu_ctx_factory):
    ctx = gpu_ctx_factory(min_cov=6, min_len=500, trim=50)
    # empty batch
    empty = capi.HostBatch(np.zeros(0, np.uint32), np.zeros(1, np.uint64), np.zeros(0, np.uint32),
                           np.zeros(0, np.uint64), np.zeros(0, np.uint32), b"", b"")
    assert ctx.consensus(empty) == []
    # a target below min_cov, a target with no alignments, a normal one
    rng = np.random.default_rng(5)
    a1, bb1 = random_target(rng, 700, 3, full_span=True)
    a3, bb3 = random_target(rng, 900, 9, full_span=True)
    batch = batch_from_targets([(700, a1, bb1), (50, [], b"A" * 50), (900, a3, bb3)])
    got = ctx.consensus(batch)
    assert got[0] == [] and got[1] == []
    assert got == oracle_batch(batch, 6, 500, 50)


def test_all_alignments_below_min_len(gpu_ctx_factory):
    """Every alignment is dropped by main.cpp:132 but the group still passes
    main.cpp:118: the graph is the bare 'N' backbone."""
    rng = np.random.default_rng(6)
    alns, bb = random_target(rng, 30, 7, full_span=True)
    batch = batch_from_targets([(30, alns, bb)])
    for mw in (0, 1, 2):
        _check_batch(gpu_ctx_factory, batch, min_cov=2, min_len=100, trim=0, min_weight=mw)


def test_alignment_trimmed_to_nothing(gpu_ctx_factory):
    """trimAln can leave empty strings (AlignmentTest.cpp:126-141); addAln then
    only adds enter->exit (AlnGraphBoost.cpp:106)."""
    rng = np.random.default_rng(8)
    alns, bb = random_target(rng, 40, 6, full_span=True)
    short = (5, b"ACGT", b"ACGT")
    batch = batch_from_targets([(40, alns + [short, short], bb)])
    _check_batch(gpu_ctx_factory, batch, min_cov=0, min_len=0, trim=3, min_weight=0)
    _check_batch(gpu_ctx_factory, batch, min_cov=0, min_len=0, trim=30, min_weight=0)


def test_nonconforming_is_rejected(gpu_ctx_factory):
    ctx = gpu_ctx_factory(min_cov=0, min_len=0, trim=0, min_weight=0)
    # start = 0 aliases the enter vertex (AlnGraphBoostTest.cpp:49-57 relies on UB)
    with pytest.raises(capi.DagconError) as e:
        ctx.consensus(batch_from_targets([(12, [(0, b"CCGCGG-G-A-T", b"C-GCGGA-T-G-")], b"N" * 12)]))
    assert e.value.code == -4
    # target bases run past tlen
    with pytest.raises(capi.DagconError) as e:
        ctx.consensus(batch_from_targets([(3, [(1, b"ACGTA", b"ACGTA")], b"NNN")]))
    assert e.value.code == -4
    # a byte outside printable ASCII
    with pytest.raises(capi.DagconError) as e:
        ctx.consensus(batch_from_targets([(5, [(1, b"AC\x01TA", b"ACGTA")], b"NNNNN")]))
    assert e.value.code == -4
    # ... anywhere in a long alignment (the 16-byte path of the chunked normalize), in either string,
    # just outside the range on both sides; and in raw mode (k_normalize_slow)
    good = bytes(b"ACGT"[i % 4] for i in range(3000))
    for pos, byte, in_q in ((1500, 0x20, True), (1501, 0x7F, False), (2999, 0x80, True), (7, 0xFF, False)):
        bad = bytearray(good); bad[pos] = byte
        q, t = (bytes(bad), good) if in_q else (good, bytes(bad))
        for flags in (0, capi.FLAG_RAW_ALIGNMENTS):
            c2 = gpu_ctx_factory(min_cov=0, min_len=0, trim=0, min_weight=0, flags=flags)
            with pytest.raises(capi.DagconError) as e:
                c2.consensus(batch_from_targets([(3000, [(1, q, t)], b"N" * 3000)]))
            assert e.value.code == -4, (pos, byte, in_q, flags)
    assert e.value.code == -4
    # the context is still usable afterwards
    ok = ctx.consensus(batch_from_targets([(5, [(1, b"ACGTA", b"ACGTA")], b"NNNNN")]))
    assert ok[0] == [(0, 5, b"ACGTA")]


def test_deep_coverage_more_than_one_wave(gpu_ctx_factory):
    """More than 64 alignments per target: adjacency rows span several waves."""
    rng = np.random.default_rng(12)
    alns, bb = random_target(rng, 80, 150, alphabet=b"ACGT", full_span=False)
    alns2, bb2 = random_target(rng, 60, 70, alphabet=b"AC", full_span=True)
    batch = batch_from_targets([(80, alns, bb), (60, alns2, bb2)])
    _check_batch(gpu_ctx_factory, batch, min_cov=6, min_len=10, trim=2)


def test_long_gap_runs_take_the_redo_path(gpu_ctx_factory):
    """Gap runs longer than k_norm_chunk's LDS windows (an insertion or deletion of hundreds of
    columns, homopolymers a gap slides through) take its second pass (512-column window) or
    k_normalize_slow; same answer."""
    rng = np.random.default_rng(21)
    tl = 900
    alns, bb = random_target(rng, tl, 8, alphabet=b"ACGT", full_span=True, sub=0.02, ins=0.08, dele=0.04)
    bbs = bytearray(bb)
    bbs[300:520] = b"A" * 220                      # long homopolymer in the backbone
    bb = bytes(bbs)
    extra = []
    for k in range(6):
        q, t = bytearray(), bytearray()
        for i in range(tl):
            if k % 3 == 0 and i == 200:            # 300-column insertion (second pass of k_norm_chunk);
                n_ins = 300 if k == 0 else 700     # 700 columns: beyond its window too, k_normalize_slow
                ins = bytes(b"ACGT"[j] for j in rng.integers(0, 4, n_ins))
                q += ins; t += b"-" * n_ins
            if k % 3 == 1 and 600 <= i < 850:      # 250-column deletion
                q.append(0x2D); t.append(bb[i]); continue
            if k % 3 == 2 and i == 299:            # an 'A' inserted in front of the homopolymer:
                q += b"A" * 3; t += b"-" * 3       # the push slides it through 220 columns
            q.append(bb[i]); t.append(bb[i])
        extra.append((1, bytes(q), bytes(t)))
    alns2 = [(1, bytes(bb[i] if rng.random() > 0.05 else 0x2D for i in range(tl)), bb) for _ in range(3)]
    batch = batch_from_targets([(tl, alns + extra + alns2, bb)])
    _check_batch(gpu_ctx_factory, batch, min_cov=0, min_len=0, trim=5, min_weight=0)
    _check_batch(gpu_ctx_factory, batch, min_cov=6, min_len=500, trim=50)
    # the a1 entry point alone, against the oracle strings
    ctx = gpu_ctx_factory(min_cov=0, min_len=0, trim=0, min_weight=0)
    got = ctx.normalize(extra, trim=7)
    for (s, q, t), (gs, gq, gt) in zip(extra, got):
        qn, tn = oracle.normalize_gaps(q, t)
        assert (gq, gt, gs) == oracle.trim_aln(qn, tn, s, 7)


def test_chunked_normalize_boundaries(gpu_ctx_factory, monkeypatch):
    """normalizeGaps runs in chunks of ~512 input columns that start cold (k_norm_chunk).
    Gaps in flight across chunk starts: an insertion in front of a long dinucleotide repeat
    slides through every chunk start inside it (the chunk in front is run again with the next
    one taken in), small alphabets keep many gaps moving, and trimAln's window ends inside,
    at and across chunk edges."""
    rng = np.random.default_rng(77)
    alns = []
    # (AC)^n backbone stretch with an 'AC' / 'A' / 'CA' insertion in front of it, at several phases
    for ins, at, rep_len in [(b"AC", 1000, 2600), (b"A", 1020, 1500), (b"CACA", 2040, 3000), (b"AC", 30, 5000)]:
        left = bytes(b"ACGT"[j] for j in rng.integers(0, 4, at))
        right = bytes(b"ACGT"[j] for j in rng.integers(0, 4, 700))
        rep = b"AC" * (rep_len // 2)
        t = left + b"G" + b"-" * len(ins) + rep + b"T" + right
        q = left + b"G" + ins + rep + b"T" + right
        alns.append((1, q, t))
        # the same with a deletion in the query instead (gaps in the other string)
        alns.append((1, left + b"G" + b"-" * len(ins) + rep + b"T" + right, left + b"G" + ins + rep + b"T" + right))
    # long random alignments over 2- and 4-letter alphabets, high indel rates
    for i in range(24):
        alph = [b"AC", b"ACGT", b"A"][i % 3]
        tl = int(rng.integers(1500, 6000))
        a1, _ = random_target(rng, tl, 1, alphabet=alph, sub=0.05, ins=0.15, dele=0.10, ins_ext=0.4, full_span=True,
                              dots=(i % 5 == 0))
        alns.append(a1[0])
    ctx = gpu_ctx_factory(min_cov=0, min_len=0, trim=0, min_weight=0)
    for trim in (0, 1, 50, 1023, 1024, 1100, 3000):
        got = ctx.normalize(alns, trim=trim)
        for (s, q, t), (gs, gq, gt) in zip(alns, got):
            qn, tn = oracle.normalize_gaps(q, t)
            assert (gq, gt, gs) == oracle.trim_aln(qn, tn, s, trim), f"trim {trim}"
    # and through the graph: matC / counts of the chunked finish against the oracle consensus
    tl = 4000
    ta, bb = random_target(rng, tl, 12, alphabet=b"AC", sub=0.04, ins=0.12, dele=0.08, ins_ext=0.4, full_span=True)
    batch = batch_from_targets([(tl, ta, bb)])
    _check_batch(gpu_ctx_factory, batch, min_cov=0, min_len=0, trim=0, min_weight=0)
    _check_batch(gpu_ctx_factory, batch, min_cov=4, min_len=100, trim=1030)
    # k_emit's entry columns next to chunk edges: 16-position stretches put one on every edge
    monkeypatch.setenv("DAGCON_EMIT_SHIFT", "4")
    _check_batch(gpu_ctx_factory, batch, min_cov=0, min_len=0, trim=0, min_weight=0)
    _check_batch(gpu_ctx_factory, batch, min_cov=4, min_len=100, trim=1030)


def test_idempotent_reruns_same_context(gpu_ctx_factory):
    """Same context, same resident batch, run twice: identical output (no state leaks
    between runs, workspace reuse is clean)."""
    batch = synth.make_batch(8, 1500, 20, seed=77)
    ctx = gpu_ctx_factory(min_cov=6, min_len=500, trim=50)
    ctx.upload(batch)
    ctx.run(); a = ctx.fetch()
    ctx.run(); b = ctx.fetch()
    assert a == b == oracle_batch(batch, 6, 500, 50)


def test_cli_end_to_end(tmp_path):
    """pbdagcon-compatible command line: .m5 text in, FASTA out, byte-identical to the records
    the reference would print (main.cpp:141-143) for the oracle's segments, in input order;
    a group below -c and a '-' strand record included; stdin ('-') gives the same bytes."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "pbdagcon_amd", "bin", "pbdagcon")
    batch = synth.make_batch(5, 1500, 12, seed=31)
    lines = synth.to_m5(batch).decode().splitlines()
    keep = [ln for i, ln in enumerate(lines) if not (24 <= i < 33)]      # target 2 keeps 3 < 6 alignments
    m5 = ("\n".join(keep) + "\n").encode()
    path = tmp_path / "in.m5"
    path.write_bytes(m5)
    out = subprocess.run([cli, "-c", "6", "-m", "500", "-t", "50", "-j", "1", str(path)],
                         capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    exp = []
    for t in range(batch.n_targets):
        alns = [a for i, a in enumerate(batch.target_alignments(t)) if not (24 <= 12 * t + i < 33)]
        if len(alns) < 6:
            continue
        for r0, r1, s in oracle.consensus_target(int(batch.tlen[t]), alns, 500, 50, 6):
            exp.append(b">%s/%d_%d\n%s\n" % (batch.ids[t].encode(), r0, r1, s))
    assert out.stdout == b"".join(exp) and len(exp) == 4
    out2 = subprocess.run([cli, "-"], input=m5, capture_output=True, timeout=300)
    assert out2.returncode == 0 and out2.stdout == out.stdout
    # several batches in flight (a new batch per slab of text), targets carried across slab edges
    out3 = subprocess.run([cli, "-j", "3", "--slab-bytes", "20000", str(path)], capture_output=True, timeout=300)
    assert out3.returncode == 0 and out3.stdout == out.stdout


def test_full_size_configs1_properties(gpu_ctx_factory):
    """BASELINE configs[1] at full size (1,000 targets x 10 kb x 40x, what bench.py times), through
    properties that do not need the oracle on all of it: a second run of the same context gives
    the same records; the records do not depend on how many pieces a target is swept in nor by which
    kernel (dozens per target four to a wave, 8 per target a wave each, one sequential sweep); every target yields one record spanning the trimmed
    backbone; a sample of targets equals the oracle's output."""
    import hashlib
    batch = synth.make_batch(1000, 10000, 40, seed=1000)
    ctx = gpu_ctx_factory(min_cov=6, min_len=500, trim=50)
    ctx.upload(batch)
    ctx.run(); r1 = ctx.fetch()
    ctx.run(); r2 = ctx.fetch()
    assert r1 == r2
    assert ctx.timings()["merge_segments"] > 40000        # k_merge_q: dozens of pieces per target, four to a wave
    import os
    os.environ["DAGCON_MERGE_Q"] = "0"                    # ... and a wave per piece (k_merge), 8 per target
    try:
        old = gpu_ctx_factory(min_cov=6, min_len=500, trim=50)
        assert old.consensus(batch) == r1
        assert old.timings()["merge_segments"] == 8000
    finally:
        del os.environ["DAGCON_MERGE_Q"]
    seq = gpu_ctx_factory(min_cov=6, min_len=500, trim=50, max_segments=1)
    r3 = seq.consensus(batch)
    assert seq.timings()["merge_segments"] == 1000
    assert hashlib.sha256(repr(r3).encode()).digest() == hashlib.sha256(repr(r1).encode()).digest()
    for t, segs in enumerate(r1):
        assert len(segs) == 1, f"target {t}"
        r0, r1_, s = segs[0]
        assert r0 == 0 and r1_ == len(s) and 9700 <= len(s) <= 10100, f"target {t}: {r0} {r1_} {len(s)}"
        assert set(s) <= set(b"ACGT")
    # whole-batch parity: the digest of all 1,000 targets' segments equals the one the CPU oracle
    # produced in the build container (tests/golden/make_large_hashes.py), and a sample live
    assert _digest(r1) == _golden("configs1_1000x10kx40")
    sample = list(range(0, 1000, 97))
    sub = batch.select(sample)
    assert oracle_batch(sub, 6, 500, 50) == [r1[t] for t in sample]


def test_config3_and_config5_shapes(gpu_ctx_factory):
    """BASELINE configs[2] / configs[4] shapes from the aligned strings inward: 50 kb x 60x
    (long target, more than 64... no: 60 reads, deep pools) and 20 kb x 30x with mixed target
    lengths and partial spans, -t 10 as dazcon does, real backbone bases given."""
    b3 = synth.make_batch(2, 50000, 60, seed=7000)
    _, r3 = _check_batch(gpu_ctx_factory, b3, min_cov=8, min_len=500, trim=50)
    assert _digest(r3) == _golden("config3_2x50kx60")
    tl = np.random.default_rng(5).integers(2000, 40000, 6)
    b5 = synth.make_batch(6, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True)
    _, r5 = _check_batch(gpu_ctx_factory, b5, min_cov=6, min_len=500, trim=10)
    assert _digest(r5) == _golden("config5_6xmixedx30_partial")
    # the config-5 shape at the size tools/shapes.py times (400 mixed-length targets, partial spans):
    # whole batch against the committed oracle digest
    tl = np.random.default_rng(5).integers(2000, 40000, 400)
    b5 = synth.make_batch(400, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True)
    ctx = gpu_ctx_factory(min_cov=6, min_len=500, trim=10)
    assert _digest(ctx.consensus(b5)) == _golden("config5_400xmixedx30_partial")


def test_production_coverage_caps(gpu_ctx_factory):
    """m4topre caps a target at 75 alignments, dazcon -m at 85: more than one
    wave of reads per target through every kernel."""
    b = synth.make_batch(3, 3000, 85, seed=4100)
    _check_batch(gpu_ctx_factory, b, min_cov=8, min_len=500, trim=50)
    b = synth.make_batch(2, 2500, 75, seed=4200, min_span=0.7)
    _check_batch(gpu_ctx_factory, b, min_cov=8, min_len=500, trim=50)


def test_limits_are_reported(gpu_ctx_factory):
    """More alignments per target than DAGCON_MAX_COVERAGE is refused up front, not mis-computed."""
    ctx = gpu_ctx_factory(min_cov=0, min_len=0, trim=0, min_weight=0)
    n = capi.MAX_COVERAGE + 1
    alns = [(1, b"ACGT", b"ACGT")] * n
    with pytest.raises(capi.DagconError) as e:
        ctx.consensus(batch_from_targets([(4, alns, b"ACGT")]))
    assert e.value.code == -5


def test_ragged_batch_and_workspace_growth(gpu_ctx_factory):
    """Targets of very different sizes in one batch, on a fresh context whose first workspace
    guess is too small for insertion-rich input: the batch is re-run with the exact sizes the
    device recorded and the answer is the same as on a warm context."""
    rng = np.random.default_rng(17)
    targets = []
    for tl, k in [(30, 3), (4000, 25), (5, 9), (900, 40), (12, 1), (2500, 12)]:
        alns, bb = random_target(rng, tl, k, alphabet=b"ACGT", ins=0.35, dele=0.05, sub=0.02, full_span=(tl < 100))
        targets.append((tl, alns, bb))
    batch = batch_from_targets(targets)
    ctx = gpu_ctx_factory(min_cov=1, min_len=0, trim=2, min_weight=1)
    a = ctx.consensus(batch)
    first = ctx.timings()["reruns"]
    b = ctx.consensus(batch)
    assert a == b == oracle_batch(batch, 1, 0, 2, 1)
    assert ctx.timings()["reruns"] == 0 and first >= 0


# ---- ABI 2: failures are confined to their target ------------------------------

def test_one_bad_target_does_not_take_the_batch_down(gpu_ctx_factory):
    """The reference's assert / undefined behaviour on a non-conforming alignment hits one worker's
    one target (AlnGraphBoost.cpp:71-72).  Here: 50 targets, one of which holds an alignment that
    runs past tlen, one a non-printable byte, one that starts behind the target; dagcon_consensus returns DAGCON_OK,
    target_status names the three, and the other 47 are bit-exact."""
    rng = np.random.default_rng(91)
    targets = []
    for i in range(50):
        tl = int(rng.integers(600, 1500))
        alns, bb = random_target(rng, tl, int(rng.integers(7, 12)), alphabet=b"ACGT", full_span=(i % 2 == 0))
        targets.append((tl, alns, bb))
    good = batch_from_targets(targets)
    exp = oracle_batch(good, 6, 500, 10)
    bad = list(targets)
    tl, alns, bb = bad[7]
    s0, q0, t0 = alns[3]
    bad[7] = (tl, alns[:3] + [(tl - 100, q0, t0)] + alns[4:], bb)             # target bases past tlen
    tl, alns, bb = bad[23]
    s0, q0, t0 = alns[0]
    bad[23] = (tl, [(s0, q0[:40] + b"\x07" + q0[41:], t0)] + alns[1:], bb)    # a byte outside 33..126
    tl, alns, bb = bad[49]
    s0, q0, t0 = alns[-1]
    bad[49] = (tl, alns[:-1] + [(tl + 5, q0, t0)], bb)                        # starts behind the target
    batch = batch_from_targets(bad)
    ctx = gpu_ctx_factory(min_cov=6, min_len=500, trim=10)
    with pytest.raises(capi.DagconError) as e:                                # strict callers still get an error
        ctx.consensus(batch)
    assert e.value.code == -4
    got = ctx.consensus(batch, strict=False)
    assert list(np.flatnonzero(ctx.target_status)) == [7, 23, 49]
    assert all(ctx.target_status[t] == -4 for t in (7, 23, 49))
    for t in range(50):
        if t in (7, 23, 49):
            assert got[t] == []
        else:
            assert got[t] == exp[t], f"target {t}"
    # the context is clean afterwards
    assert ctx.consensus(good) == exp and not ctx.target_status.any()


def test_flags_are_checked_and_host_alloc(gpu_ctx_factory):
    """dagcon_create refuses undefined flag bits (bit 8 is an internal one: DG_F_A1_ONLY would switch the
    conformity check off); dagcon_host_alloc gives page-locked blobs that dagcon_upload takes as they are."""
    with pytest.raises(capi.DagconError) as e:
        capi.Context(min_cov=0, min_len=0, trim=0, flags=8)
    assert e.value.code == -5
    with pytest.raises(capi.DagconError) as e:
        capi.Context(flags=1 << 20)
    assert e.value.code == -5
    batch = synth.make_batch(4, 1200, 12, seed=5)
    ctx = gpu_ctx_factory(min_cov=6, min_len=500, trim=50)
    pinned = ctx.pin_batch(batch)
    assert ctx.consensus(pinned) == ctx.consensus(batch) == oracle_batch(batch, 6, 500, 50)
    # an offset that wraps: refused, not read
    import copy
    b2 = copy.copy(batch)
    b2.aln_off = batch.aln_off.copy(); b2.aln_off[3] = np.uint64(2**64 - 16)
    with pytest.raises(capi.DagconError) as e:
        ctx.consensus(b2)
    assert e.value.code == -1


def test_very_deep_coverage(gpu_ctx_factory):
    """DAGCON_MAX_COVERAGE is what the header says: thousands of alignments on one target (k_lists
    with more than 64 KB of dynamic LDS, the multi-wave rows of k_groups / k_emit / k_lists / k_merge)."""
    rng = np.random.default_rng(33)
    for k, tl in ((2100, 24), (capi.MAX_COVERAGE, 12)):
        alns, bb = random_target(rng, tl, k, alphabet=b"ACGT", sub=0.05, ins=0.08, dele=0.05, full_span=False)
        batch = batch_from_targets([(tl, alns, bb)])
        _check_batch(gpu_ctx_factory, batch, min_cov=6, min_len=0, trim=0, min_weight=-1)
        _check_batch(gpu_ctx_factory, batch, min_cov=0, min_len=0, trim=1, min_weight=0)


def test_cli_several_workers_and_a_bad_target(tmp_path):
    """--devices: one consensus worker (thread + context) per listed GPU, batches dealt to them in
    input order, records printed in input order (here the same GPU twice: the box has one).  A
    target with a non-conforming alignment is reported on stderr and skipped; exit code 0."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "pbdagcon_amd", "bin", "pbdagcon")
    batch = synth.make_batch(12, 1200, 10, seed=77)
    m5 = synth.to_m5(batch)
    path = tmp_path / "in.m5"
    path.write_bytes(m5)
    one = subprocess.run([cli, "-j", "2", str(path)], capture_output=True, timeout=300)
    assert one.returncode == 0, one.stderr.decode()
    exp = b"".join(b">%s/%d_%d\n%s\n" % (batch.ids[t].encode(), r0, r1, s)
                   for t, segs in enumerate(oracle_batch(batch, 6, 500, 50)) for r0, r1, s in segs)
    assert one.stdout == exp and exp.count(b">") == 12
    two = subprocess.run([cli, "-j", "2", "--devices", "0,0", "--batch-targets", "2", "--pinned", "1", str(path)],
                         capture_output=True, timeout=300)
    assert two.returncode == 0 and two.stdout == exp, two.stderr.decode()
    # --contexts: several workers per GPU (what long inputs get by default), text taken a small slab at a time so
    # that consumed text is dropped while later batches are still being parsed
    three = subprocess.run([cli, "-j", "3", "--contexts", "3", "--batch-targets", "1", "--slab-bytes", "40000", str(path)],
                           capture_output=True, timeout=300, env=dict(os.environ, PBDAGCON_TIMING="1"))
    assert three.returncode == 0 and three.stdout == exp, three.stderr.decode()
    assert b"pbdagcon timing: total" in three.stderr
    # break one alignment of target 5: tStart beyond the target
    lines = m5.decode().splitlines()
    f = lines[53].split(" ")
    f[7] = str(1150)
    lines[53] = " ".join(f)
    path.write_text("\n".join(lines) + "\n")
    bad = subprocess.run([cli, "--devices", "0,0", "--batch-targets", "3", str(path)], capture_output=True, timeout=300)
    assert bad.returncode == 0
    assert b"skipped" in bad.stderr and batch.ids[5].encode() in bad.stderr
    keep = b"".join(rec for rec in [b">" + r for r in exp.split(b">")[1:]] if not rec.startswith(b">" + batch.ids[5].encode()))
    assert bad.stdout == keep


def test_bench_two_ranks_share_the_gpu():
    """bench.py --gpus 2 starts two ranks itself; with --backend gloo they share the one GPU of the box:
    every rank checks its shard against the oracle, rank 0 checks the gathered FASTA against the
    SHA-256 each rank took of its own part."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                          "--warmup", "1", "--targets", "24", "--tlen", "2000", "--coverage", "16", "--no-legs",
                          "--cpu-sample", "24"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["fasta_gather_ok"] is True and line["bit_exact_vs_oracle"] is True
    assert line["targets_verified"] == 24 + 16


def test_bench_streams_its_shard_in_batches_two_ranks():
    """configs[3] as specified, rehearsed on the one GPU of the box: two gloo ranks, each streams ITS shard of the global
    target space (3 batches of 24 targets) through two contexts, FASTA gathered per super-batch; every generated batch
    is checked against the oracle target by target (--stream-verify -1), the re-used batch against the batch it
    re-uses, the gathered payload against every rank's size + SHA-256."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--stream-batches", "3", "--stream-distinct", "2", "--gather-every", "2", "--targets", "24",
                          "--tlen", "2000", "--coverage", "16", "--stream-verify", "-1"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["targets_total"] == 144 and line["targets_done"] == 144
    assert line["fasta_gather_ok"] is True and line["bit_exact_vs_oracle"] is True and line["reused_batches_identical"] is True
    assert line["targets_verified"] == 2 * 2 * 24 and line["fasta_records"] == 144 and line["gather_rounds"] == 2
    assert line["value"] > 0


# ---- the -a stage: .pre records re-aligned on the device ------------------------

def _mutate(rng, t, sub=0.03, ins=0.08, dele=0.05):
    q = bytearray()
    for c in t:
        u = rng.random()
        if u < dele:
            continue
        q.append(c if u > dele + sub else b"ACGT"[rng.integers(0, 4)])
        while rng.random() < ins:
            q.append(b"ACGT"[rng.integers(0, 4)])
    return bytes(q)


def test_align_kat_and_twin(gpu_ctx_factory, monkeypatch):
    """dagcon_align against its CPU twin (oracle.banded_align), bit for bit: the reference's one
    known-answer test (test/cpp/SimpleAlignerTest.cpp:8-21), empty and one-base sequences, pairs of
    very different lengths (band edges), pairs of several thousand to a hundred thousand bases (band
    narrower than the matrix, every kernel instance, directions staged through LDS in many rounds), a
    band that does not connect the corners, and several launch groups."""
    import json
    import os
    k = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_graph.json")))["simple_aligner"]
    ctx = gpu_ctx_factory()
    (qa, ta), = ctx.align([(k["qstr"].encode(), k["tstr"].encode())])
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    assert ta.translate(rc)[::-1].decode() == k["expected_tstr"]
    assert (qa, ta) == oracle.banded_align(k["qstr"].encode(), k["tstr"].encode())
    rng = np.random.default_rng(44)
    pairs = [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"A", b"A"), (b"A", b"C"), (b"ACGTACGT", b"A"), (b"G", b"ACGTACGG")]
    for i in range(60):
        t = bytes(b"ACGT"[j] for j in rng.integers(0, 4, int(rng.integers(1, 300))))
        q = _mutate(rng, t) if i % 3 else bytes(b"ACGT"[j] for j in rng.integers(0, 4, int(rng.integers(1, 300))))
        pairs.append((q, t))
    for n in (1500, 4000, 9000):
        t = bytes(b"ACGT"[j] for j in rng.integers(0, 4, n))
        pairs.append((_mutate(rng, t), t))
        pairs.append((_mutate(rng, t, ins=0.2), t[: n // 2]))       # the band does not reach the corner cleanly
    # every kernel instance (2 .. 16 cells per lane: the band is 2 W + 1 cells on 64 lanes)
    for n in (20000, 45000, 110000):
        t = bytes(b"ACGT"[j] for j in rng.integers(0, 4, n))
        pairs.append((_mutate(rng, t), t))
    # a target far longer than the query: the band jumps more than its width from row to row and never
    # connects the corners (the twin then returns nothing: its walk stops at the unreachable corner)
    t = bytes(b"ACGT"[j] for j in rng.integers(0, 4, 5000))
    pairs.append((t[:3], t))
    pairs.append((t[:40], t[:3000]))
    pairs.append((t, t[:40]))
    exp = [oracle.banded_align(q, t) for q, t in pairs]
    assert exp[-3] == (b"", b"")
    assert ctx.align(pairs) == exp
    monkeypatch.setenv("DAGCON_ALIGN_ROWS", "5000")                 # several launch groups
    assert ctx.align(pairs) == exp
    # the static bands alone (every kernel instance of k_align_band; by default the band that follows the
    # alignment takes the long pairs and the static ones only see the short pairs and the fall-backs)
    monkeypatch.delenv("DAGCON_ALIGN_ROWS")
    monkeypatch.setenv("DAGCON_ALIGN_STATIC", "1")
    monkeypatch.setenv("OG_NO_ADAPTIVE", "1")
    exp_static = [oracle.banded_align(q, t) for q, t in pairs]
    assert ctx.align(pairs) == exp_static
    # (on these pairs the two agree wherever the following band stood)
    assert sum(a != b for a, b in zip(exp, exp_static)) <= 2


def test_cli_pre_input_with_align(tmp_path):
    """pbdagcon -a file.pre: parsePre (Alignment.cpp:82-112), every record re-aligned
    (main.cpp:127-128, SimpleAligner.cpp:25-63: start / end arithmetic, '-' strand reverse
    complement), then the usual path; against the same steps on the CPU (twin aligner + oracle)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "pbdagcon_amd", "bin", "pbdagcon")
    rng = np.random.default_rng(8)
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    lines, exp = [], []
    for ti in range(4):
        tlen = int(rng.integers(1500, 2500))
        target = bytes(b"ACGT"[j] for j in rng.integers(0, 4, tlen))
        alns = []
        for r in range(10 if ti != 2 else 3):                       # target 2 stays below -c 6
            s = int(rng.integers(0, tlen // 4)); e = int(rng.integers(3 * tlen // 4, tlen + 1))
            strand = b"+-"[r % 2:r % 2 + 1]
            # m4topre.py:194-206 hands over the target substring in the read's orientation
            tseq = target[s:e] if strand == b"+" else target[s:e].translate(rc)[::-1]
            qseq = _mutate(rng, tseq)
            tstart = s if strand == b"+" else tlen - e
            lines.append(b" ".join([b"q%d_%d" % (ti, r), b"t%d" % ti, strand, b"%d" % tlen, b"%d" % tstart,
                                    b"%d" % (tstart + len(tseq)), qseq, tseq]))
            st, en, qa, ta = oracle.simple_align(tstart, tlen, strand, qseq, tseq)
            alns.append((st, qa, ta))
        if len(alns) >= 6:
            for r0, r1, s_ in oracle.consensus_target(tlen, alns, 500, 50, 6):
                exp.append(b">t%d/%d_%d\n%s\n" % (ti, r0, r1, s_))
    path = tmp_path / "in.pre"
    path.write_bytes(b"\n".join(lines) + b"\n")
    out = subprocess.run([cli, "-a", "-j", "2", str(path)], capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    assert out.stdout == b"".join(exp) and len(exp) == 3


def test_consensus_pre_keeps_strings_on_the_device(gpu_ctx_factory):
    """dagcon_consensus_pre (align, SimpleAligner.cpp:51-62 finish with the '-' strand reverse-complemented on
    the device, then the usual path, nothing copied back in between) against the same steps on the CPU."""
    rng = np.random.default_rng(77)
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    targets, exp = [], []
    for ti in range(5):
        tlen = int(rng.integers(900, 2600))
        target = bytes(b"ACGT"[j] for j in rng.integers(0, 4, tlen))
        recs, alns = [], []
        for r in range(9 if ti != 3 else 2):                        # target 3 stays below min_cov
            s = int(rng.integers(0, tlen // 4)); e = int(rng.integers(3 * tlen // 4, tlen + 1))
            strand = b"+-"[(r + ti) % 2:(r + ti) % 2 + 1]
            tseq = target[s:e] if strand == b"+" else target[s:e].translate(rc)[::-1]
            qseq = _mutate(rng, tseq)
            tstart = s if strand == b"+" else tlen - e
            recs.append((tstart, strand, qseq, tseq))
            st, en, qa, ta = oracle.simple_align(tstart, tlen, strand, qseq, tseq)
            alns.append((st, qa, ta))
        targets.append((tlen, recs))
        exp.append(oracle.consensus_target(tlen, alns, 500, 50, 6) if len(alns) >= 6 else [])
    targets.append((1200, []))                                      # a target without records
    exp.append([])
    ctx = gpu_ctx_factory(min_cov=6, min_len=500, trim=50)
    assert ctx.consensus_pre(targets) == exp
    assert ctx.consensus_pre([]) == []
    # a record that leaves its target (tstart + |tseq| > tlen) fails that target alone
    tl0, recs0 = targets[0]
    bad = [(tl0, [(tl0 - 10,) + recs0[0][1:]] + recs0[1:])] + targets[1:]
    got = ctx.consensus_pre(bad, strict=False)
    assert got[0] == [] and got[1:] == exp[1:]
    assert ctx.target_status[0] == -4 and not ctx.target_status[1:].any()


def _polish_twin(tlen, recs, rounds, trim, min_cov, min_len, pad=64):
    """The steps of `pbdagcon -a --polish N` (csrc/host/pbdagcon_main.cpp) composed on the CPU:
    recs = [(tstart, strand, qseq, tseq, q_fwd)] of one target -> segments after `rounds` rounds."""
    alns, cur = [], []
    for tstart, strand, qseq, tseq, q_fwd in recs:
        st, en, qa, ta = oracle.simple_align(tstart, tlen, strand, qseq, tseq)
        alns.append((st, qa, ta)); cur.append([st, qa, ta, 0])
    segs = oracle.consensus_target(tlen, alns, min_len, trim, min_cov) if len(alns) >= min_cov else []
    for _ in range(rounds):
        bb, r0, r1 = b"", 0, 0
        for a0, a1, sq in segs:
            if len(sq) > len(bb):
                bb, r0, r1 = sq, a0, a1
        alns = []
        for k, (tstart, strand, qseq, tseq, q_fwd) in enumerate(recs):
            st, qa, ta, qb = cur[k]
            if not bb or not qa:
                cur[k] = [0, b"", b"", 0]
                continue
            org = trim + r0
            lo_t, hi_t = org - pad, trim + r1 + pad
            tpos, qpos, qlo, qhi, t_first, t_last = st - 1, 0, 0, 0, -1, -1
            for qc, tc in zip(qa, ta):
                inside = lo_t <= tpos < hi_t
                if qc != 0x2D:
                    if tpos < lo_t:
                        qlo = qpos + 1
                    if inside:
                        qhi = qpos + 1
                    qpos += 1
                if inside and tc != 0x2D:
                    if t_first < 0:
                        t_first = tpos
                    t_last = tpos
                if tc != 0x2D:
                    tpos += 1
            if qhi <= qlo or t_first < 0:
                cur[k] = [0, b"", b"", 0]
                continue
            w0 = max(0, min(t_first - org - pad, len(bb)))
            w1 = max(w0, min(t_last + 1 - org + pad, len(bb)))
            qa2, ta2 = oracle.banded_align(q_fwd[qb + qlo:qb + qhi], bb[w0:w1])
            lead = len(qa2) - len(qa2.lstrip(b"-"))
            n = len(qa2.rstrip(b"-"))
            qa2, ta2 = qa2[lead:n], ta2[lead:n]
            cur[k] = [w0 + lead + 1, qa2, ta2, qb + qlo]
            if qa2:
                alns.append((w0 + lead + 1, qa2, ta2))
        segs = oracle.consensus_target(len(bb), alns, min_len, trim, min_cov, backbone=bb) if (bb and len(alns) >= min_cov) else []
    return segs


def test_cli_polish_rounds(tmp_path):
    """pbdagcon -a --polish N (the reference names the use in README.md:14-15 and has no code
    for it): every round takes the longest consensus segment as the new backbone and aligns the reads to it
    again.  The command line against the same steps composed on the CPU (twin aligner + oracle), byte for
    byte; and the point of it: the consensus gets closer to the sequence the reads were drawn from."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "pbdagcon_amd", "bin", "pbdagcon")
    rng = np.random.default_rng(21)
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    trim, min_cov, min_len = 20, 6, 500

    def infix_distance(c, t):
        """edit distance of c to the best-matching stretch of t"""
        prev = np.zeros(len(t) + 1, dtype=np.int64)
        tt = np.frombuffer(t, np.uint8)
        for ch in c:
            sub = prev[:-1] + (tt != ch)
            cur = np.empty_like(prev)
            cur[0] = prev[0] + 1
            np.minimum(sub, prev[1:] + 1, out=cur[1:])
            # (insertions along the row: a running minimum of cur[j-1] + 1)
            cur = np.minimum.accumulate(cur - np.arange(cur.size)) + np.arange(cur.size)
            prev = cur
        return int(prev.min())

    lines, targets = [], []
    for ti in range(3):
        truth = bytes(b"ACGT"[j] for j in rng.integers(0, 4, int(rng.integers(1500, 2000))))
        backbone = _mutate(rng, truth, sub=0.04, ins=0.06, dele=0.06)        # the seed read: as noisy as the others
        tlen = len(backbone)
        recs = []
        for r in range(24):
            s = 0 if r % 3 else int(rng.integers(0, tlen // 5))
            e = tlen if r % 3 else int(rng.integers(4 * tlen // 5, tlen + 1))
            ts, te = int(s * len(truth) / tlen), int(e * len(truth) / tlen)   # about the same stretch of the truth
            strand = b"+-"[r % 2:r % 2 + 1]
            q_fwd = _mutate(rng, truth[ts:te], sub=0.03, ins=0.07, dele=0.05)
            tseq_fwd = backbone[s:e]
            qseq = q_fwd if strand == b"+" else q_fwd.translate(rc)[::-1]
            tseq = tseq_fwd if strand == b"+" else tseq_fwd.translate(rc)[::-1]
            tstart = s if strand == b"+" else tlen - e
            lines.append(b" ".join([b"q%d_%d" % (ti, r), b"t%d" % ti, strand, b"%d" % tlen, b"%d" % tstart,
                                    b"%d" % (tstart + len(tseq)), qseq, tseq]))
            recs.append((tstart, strand, qseq, tseq, q_fwd))
        targets.append((truth, tlen, recs))
    path = tmp_path / "in.pre"
    path.write_bytes(b"\n".join(lines) + b"\n")
    err = {}
    for rounds in (0, 1, 2):
        got = subprocess.run([cli, "-a", "-t", str(trim), "--polish", str(rounds), str(path)], capture_output=True, timeout=600)
        assert got.returncode == 0, got.stderr.decode()
        exp, e = [], 0.0
        for ti, (truth, tlen, recs) in enumerate(targets):
            segs = _polish_twin(tlen, recs, rounds, trim, min_cov, min_len)
            assert segs, f"--polish {rounds}: target {ti} lost its consensus"
            exp += [b">t%d/%d_%d\n%s\n" % (ti, a0, a1, sq) for a0, a1, sq in segs]
            best = max((sq for _, _, sq in segs), key=len)
            e += infix_distance(best, truth) / len(best)
        assert got.stdout == b"".join(exp), f"--polish {rounds}"
        err[rounds] = e
    assert err[2] < err[0] and err[1] < err[0], err


# ---- round 3: the parity holes round 2 exposed ------------------------------------------------

@pytest.mark.parametrize("seed", [417, 463])
def test_a_cut_vertex_that_a_read_start_could_merge_with(seed, monkeypatch):
    """tools/stress.py seeds 417 and 463, round 4 (26 - 30 targets of 0.7 - 9 kb at span 0.6, -t 300) with max_segments
    64 and min_segment_len 4 / 64: round 2's generalized cuts took a backbone vertex v for a cut although a read started at
    v's successor with a leading insertion of v's base -- mergeInNodes(successor) then unites v with that vertex, from the
    worker BEHIND the cut, while the worker in front is still to run mergeInNodes(v) (k_cuts2, condition (5)).  With
    pieces that short the two workers met: a merged graph that differed from run to run (and from the oracle's in every
    run), bestPath stuck or out of bounds on it -- one run in six ended in a GPU fault or DAGCON_ERR_INTERNAL.  Here: the
    merged graph of every target against the oracle's, vertex by vertex and list order included, three runs over, with
    every buffer the kernels fill poisoned; then the consensus of the setting."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress
    monkeypatch.setenv("DAGCON_POISON", "15")
    b, desc, min_cov, min_len, trim, kws = stress.make_round(seed, 4)
    oracle_graphs = {}
    for rep in range(3):
        ctx = capi.Context(min_cov=min_cov, min_len=min_len, trim=trim, flags=capi.FLAG_STOP_AFTER_MERGE, **kws[2])
        try:
            ctx.consensus(b, strict=False)
            assert ctx.timings()["merge_segments"] > 20 * b.n_targets           # the short pieces were really used
            for t in range(b.n_targets):
                if ctx.target_status[t] != 0:
                    continue
                got = ctx.debug_graph(t)
                if t not in oracle_graphs:
                    oracle_graphs[t] = _oracle_graph(b, t, min_len, trim, True)
                exp, o2d = oracle_graphs[t]
                assert len(got) == len(exp)
                for o, (eb, ew, ec, ed, eoe, eie) in enumerate(exp):
                    g = got[o2d[o]]
                    assert g["deleted"] == ed, (rep, t, o2d[o])
                    if ed:
                        continue
                    assert g["weight"] == ew and [(d, c) for d, c in g["out"]] == [(o2d[d], c) for d, c in eoe], (rep, t, o2d[o])
                    assert list(g["inn"]) == [o2d[x[0]] if isinstance(x, tuple) else o2d[x] for x in eie], (rep, t, o2d[o])
        finally:
            ctx.close()
    exp = oracle_batch(b, min_cov, min_len, trim)
    for rep in range(3):
        ctx = capi.Context(min_cov=min_cov, min_len=min_len, trim=trim, **kws[2])
        try:
            assert ctx.consensus(b) == exp, f"{desc} {kws[2]} run {rep}"
        finally:
            ctx.close()


@pytest.mark.parametrize("rnd", [0, 7, 8, 23])
def test_stress_rounds_that_caught_the_exit_tree_bug(rnd, monkeypatch):
    """tools/stress.py seed 17, rounds 0, 7, 8, 23, verbatim (its generator, make_round): partial spans down to 0.3 at
    65x with -t 300, adversarial pileups with pieces of 4 / 64 positions.  Round 2's first partial-span bestPath
    passed the whole suite and came out 34 bases short on round 0 (the tree of vertices that lead to exit and nowhere
    else, k_bp_xtree).  Every setting of the campaign: default, one piece, 64 pieces, and k_emit stretches of 16 / 64."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress
    b, desc, min_cov, min_len, trim, kws = stress.make_round(17, rnd)
    exp = oracle_batch(b, min_cov, min_len, trim)
    for kw in kws:
        for shift in (None, "4", "6"):
            if shift is None:
                monkeypatch.delenv("DAGCON_EMIT_SHIFT", raising=False)
            else:
                monkeypatch.setenv("DAGCON_EMIT_SHIFT", shift)
            ctx = capi.Context(min_cov=min_cov, min_len=min_len, trim=trim, **kw)
            try:
                got = ctx.consensus(b)
            finally:
                ctx.close()
            bad = [t for t in range(len(exp)) if got[t] != exp[t]]
            assert not bad, f"{desc} {kw} shift={shift} -c {min_cov} -m {min_len} -t {trim}: targets {bad[:8]}"


def test_exit_tree_reads_end_all_over_the_backbone(gpu_ctx_factory):
    """mergeInNodes(exit) (AlnGraphBoost.cpp:162-215 on `$`) unites the vertices reads END with, wherever on the
    backbone they lie, and its recursion their predecessors: reads that end at many different positions behind equal
    trailing insertions ('A', 'CA', 'GCA', 'TGCA') grow one little tree into which edges run from all over the
    backbone (k_bp_xtree scores it before the pieces are swept; a piece's sweep treats an edge into it as an escape
    like an edge to exit).  Merged graph vertex by vertex, consensus under several piece counts.  (No BACKBONE vertex is
    ever reaped, here or in 6,000 random adversarial pileups searched with the Python model: the ctor's edge keeps a
    backbone vertex first in its successor's in-list and gives it a second out-edge wherever a read leaves it.)"""
    rng = np.random.default_rng(99)
    tails = [b"A", b"CA", b"GCA", b"TGCA", b"A", b"CA"]
    targets = []
    for i in range(6):
        tl = int(rng.integers(1500, 4000))
        bb = bytes(b"ACGT"[k] for k in rng.integers(0, 4, tl))
        alns = []
        for r in range(int(rng.integers(24, 50))):
            s0 = int(rng.integers(0, tl // 3))
            e0 = int(rng.integers(s0 + tl // 4, tl + 1))
            q, t = bytearray(), bytearray()
            for p in range(s0, e0):
                u = rng.random()
                if u < 0.04:
                    q.append(0x2D); t.append(bb[p])
                else:
                    q.append(bb[p]); t.append(bb[p])
                if rng.random() < 0.06:
                    q.append(b"ACGT"[rng.integers(0, 4)]); t.append(0x2D)
            if q[-1] != t[-1]:                       # end on a match column, then the trailing insertion
                q[-1] = t[-1]
            tail = tails[int(rng.integers(0, len(tails)))]
            q += tail; t += b"-" * len(tail)
            alns.append((s0 + 1, bytes(q), bytes(t)))
        targets.append((tl, alns, bb))
    batch = batch_from_targets(targets, with_backbone=True)
    exp = oracle_batch(batch, 4, 200, 0)
    assert sum(len(s) for s in exp) >= 6
    pieces = {}
    for kw in (dict(), dict(max_segments=1), dict(max_segments=64, min_segment_len=4), dict(max_segments=64, min_segment_len=64),
               dict(max_segments=8, min_segment_len=300)):
        ctx = gpu_ctx_factory(min_cov=4, min_len=200, trim=0, **kw)
        got = ctx.consensus(batch)
        assert got == exp, f"{kw}: targets {[t for t in range(len(exp)) if got[t] != exp[t]]}"
        pieces[tuple(kw.items())] = ctx.timings()["merge_segments"]
    assert max(pieces.values()) > 10 * batch.n_targets            # the partial-span cuts were really used
    ctx = gpu_ctx_factory(min_cov=4, min_len=200, trim=0, min_weight=0, max_segments=64, min_segment_len=16,
                          flags=capi.FLAG_STOP_AFTER_MERGE)
    ctx.consensus(batch)
    _check_graphs(ctx, batch, 0, True)
    # the tree is really there: the exit vertex has fewer in-edges than reads ended, and some vertex that leads to
    # exit alone has in-edges from several backbone positions
    g = ctx.debug_graph(0)
    exit_in = g[-1]["inn"]
    assert len(exit_in) < len(targets[0][1])
    fan = max(len({g[s]["bbpos"] for s in g[v]["inn"]}) for v in exit_in if not g[v]["backbone"])
    assert fan >= 3, fan


def test_one_deep_target_in_a_shallow_batch(gpu_ctx_factory):
    """Routing is per target, not by the batch's average coverage: 999 targets at 40x and ONE at 500x (10 kb each).
    The shallow ones are swept four segments to a wave (k_merge_q: a row holds 8 + 8 list entries), the deep one --
    whose lists would push nearly every visit onto the literal single-lane path there (830 ms against
    88 at 250x) -- a wave per segment (k_merge) beside them.  Records identical to the same targets run apart; the deep
    target and a sample of the shallow ones against the oracle.  What the deep target costs is in the message: at 500x
    no backbone vertex is passed by every read (0.96^500), so it has no cut and is ONE sequential sweep (measured
    0.63 s, lists longer than a wave on the literal path) -- the shallow targets finish beside it in their usual time."""
    from util import concat_batches
    sh_a = synth.make_batch(500, 10000, 40, seed=1000)
    sh_b = synth.make_batch(499, 10000, 40, seed=1000, first_target=500)
    deep = synth.make_batch(1, 10000, 500, seed=555000)
    shallow = concat_batches([sh_a, sh_b])
    mixed = concat_batches([sh_a, deep, sh_b])
    ctx = gpu_ctx_factory(min_cov=6, min_len=500, trim=50)
    r_sh = ctx.consensus(shallow)
    ms_sh = ctx.timings()["ms_total"]
    r_mx = ctx.consensus(mixed)
    r_mx = ctx.consensus(mixed)                  # (the second run: arenas grown)
    tm = ctx.timings()
    assert r_mx[:500] == r_sh[:500] and r_mx[501:] == r_sh[500:]
    assert oracle_batch(deep, 6, 500, 50) == [r_mx[500]]
    sample = [0, 250, 499, 501, 750, 999]
    assert oracle_batch(mixed.select(sample), 6, 500, 50) == [r_mx[t] for t in sample]
    assert tm["ms_total"] < 2.0 * ms_sh + 1500.0, \
        f"mixed batch {tm['ms_total']:.1f} ms (merge {tm['ms_merge']:.1f}) against {ms_sh:.1f} ms for the 999 shallow targets alone"
    print(f"mixed-coverage batch: {tm['ms_total']:.1f} ms (merge {tm['ms_merge']:.1f} ms); shallow alone {ms_sh:.1f} ms")


def test_align_reports_the_records_it_drops(gpu_ctx_factory):
    """A pair whose corners no band connects (a 12-base read against a 9 kb target: the band moves 750 columns a row) comes back with length 0, as
    from the CPU twin; where the reference's SDPAlign + GuidedAlign would return something (SimpleAligner.cpp:35-48) the
    record is then dropped by the min_len filter -- dagcon_align_dropped says how many (the CLI warns)."""
    import ctypes
    rng = np.random.default_rng(5)
    t_long = bytes(b"ACGT"[k] for k in rng.integers(0, 4, 9000))
    ok_t = bytes(b"ACGT"[k] for k in rng.integers(0, 4, 700))
    pairs = [(ok_t[:650], ok_t), (t_long[:12], t_long), (ok_t, ok_t)]
    ctx = gpu_ctx_factory(min_cov=0, min_len=0, trim=0)
    got = ctx.align(pairs)
    exp = [oracle.banded_align(q, t) for q, t in pairs]
    assert got == exp
    assert len(got[1][0]) == 0 and len(got[0][0]) > 0 and len(got[2][0]) == 700
    ctx.L.dagcon_align_dropped.restype = ctypes.c_uint32
    ctx.L.dagcon_align_dropped.argtypes = [ctypes.c_void_p]
    assert ctx.L.dagcon_align_dropped(ctx.h) == 1
    ctx.align(pairs[:1])
    assert ctx.L.dagcon_align_dropped(ctx.h) == 0


@pytest.mark.parametrize("seed,rnd", [(205, 5), (201, 3)])
def test_nothing_reads_what_this_run_did_not_write(seed, rnd, monkeypatch):
    """Two rounds of tools/stress.py that a context on RE-USED memory got wrong in round 3 (an internal error in one, a
    memory fault in the other) and a fresh process got right: the fold left a victim's cell in enter's departure row
    unwritten -- that row leaves a batch only where a cell is not 0 -- and nobody clears the matrices, so the cell was
    whatever the arena held before.  DAGCON_POISON fills the arenas no kernel clears (cells, vertex records, slot pool,
    columns) with 0xEE before every run: anything read before it is written in the same run shows at once.  All nine
    settings of the campaign, reads with leading insertions, -t 0, targets below -c among the others."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress
    monkeypatch.setenv("DAGCON_POISON", "7")
    b, desc, min_cov, min_len, trim, kws = stress.make_round(seed, rnd)
    exp = oracle_batch(b, min_cov, min_len, trim)
    for kw in kws:
        for shift in (None, "4", "6"):
            if shift is None:
                monkeypatch.delenv("DAGCON_EMIT_SHIFT", raising=False)
            else:
                monkeypatch.setenv("DAGCON_EMIT_SHIFT", shift)
            ctx = capi.Context(min_cov=min_cov, min_len=min_len, trim=trim, **kw)
            try:
                for _ in range(2):                      # (the second run on the same arenas)
                    got = ctx.consensus(b)
                    assert got == exp, f"{desc} {kw} shift={shift}: targets {[t for t in range(len(exp)) if got[t] != exp[t]][:8]}"
            finally:
                ctx.close()


@pytest.mark.parametrize("knob", ["DAGCON_FOLD", "DAGCON_EMIT_SCAN", "DAGCON_NF2", "DAGCON_BP_FUSED", "DAGCON_BP_LANE",
                                  "DAGCON_BP_LANE_STACK"])
def test_round3_paths_and_their_checkers_agree(knob, monkeypatch):
    """Every round-3 path has the path it replaced behind a switch -- the fold (duplicate chains merged by the sweep
    instead), the wave's own prefix over the reads (k_groups instead), k_norm_finish2 (a lane per chunk instead), the
    one-sweep partial-span bestPath (three sweeps instead), the full-span bestPath with a lane per piece (a wave per
    piece instead; DAGCON_BP_LANE_STACK = 0 / 1: a lane gives its piece up at the first / second turned-around edge in
    a row, so that the wave-per-piece sweep takes over half-done pieces).  Both sides of each switch against the oracle: full-span and
    partial-span synthetic pileups, adversarial little ones (leading insertions, tiny alphabets), a deep target (more
    than a wave of reads: the k_groups path whatever the switch says)."""
    rng = np.random.default_rng(33)
    batches = [(synth.make_batch(8, 4000, 36, seed=9100), dict(min_cov=6, min_len=500, trim=50)),
               (synth.make_batch(8, 0, 28, seed=9200, min_span=0.4, tlens=rng.integers(1500, 9000, 8), with_backbone=True),
                dict(min_cov=6, min_len=300, trim=10)),
               (synth.make_batch(2, 3000, 90, seed=9300), dict(min_cov=6, min_len=500, trim=50))]
    targets = []
    for i in range(40):
        tl = int(rng.integers(30, 400))
        alns, bb = random_target(rng, tl, int(rng.integers(3, 24)), alphabet=[b"AC", b"ACGT", b"A"][i % 3],
                                 ins=float(rng.uniform(0.05, 0.3)), dele=float(rng.uniform(0.02, 0.2)), full_span=(i % 2 == 0))
        targets.append((tl, alns, bb))
    batches.append((batch_from_targets(targets), dict(min_cov=0, min_len=0, trim=0, min_weight=0)))
    for val in ("0", "1"):
        monkeypatch.setenv(knob, val)
        if knob == "DAGCON_BP_LANE" and val == "1": monkeypatch.setenv(knob, "2")     # (2: whatever the batch size)
        if knob == "DAGCON_BP_LANE_STACK": monkeypatch.setenv("DAGCON_BP_LANE", "2")
        for b, kw in batches:
            exp = oracle_batch(b, kw["min_cov"], kw["min_len"], kw["trim"], kw.get("min_weight"))
            ctx = capi.Context(**kw)
            try:
                got = ctx.consensus(b)
            finally:
                ctx.close()
            assert got == exp, f"{knob}={val}: targets {[t for t in range(len(exp)) if got[t] != exp[t]][:8]}"
