"""`.pre` input and the `-a` stage, CPU side: the restatement of parsePre against the
reference's own parser, the re-aligner's CPU twin against the reference's one known-answer test,
the CLI's .pre parser."""
import json
import os
import subprocess

import numpy as np

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_parse_pre_matches_reference_vectors():
    """Alignment.cpp:82-112 through tests/golden/pre_vectors.json (made by the reference's parsePre)."""
    oracle.build()
    vec = json.load(open(os.path.join(GOLD, "pre_vectors.json")))["parsed"]
    assert len(vec) == 40
    for v in vec:
        r = oracle.parse_pre(v["line"].encode())
        for k in ("id", "sid", "qstr", "tstr", "strand"):
            assert r[k].decode() == v[k], (k, v["line"])
        assert (r["tlen"], r["start"], r["end"]) == (v["tlen"], v["start"], v["end"])
    assert oracle.parse_pre(b"") is None and oracle.parse_pre(b"a b c") == -1


def test_parse_pre_against_live_reference_build():
    if oracle.ref_lib() is None:
        import pytest
        pytest.skip("oracle/_ref not present")
    rng = np.random.default_rng(5)
    for i in range(200):
        n, m = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        q = bytes(b"ACGTacgtN-"[j] for j in rng.integers(0, 10, n))
        t = bytes(b"ACGTN"[j] for j in rng.integers(0, 5, m))
        line = b" ".join([b"q%d" % i, b"t%d" % (i % 3), b"+-"[i % 2:i % 2 + 1], b"%d" % rng.integers(1, 9999),
                          b"%d" % rng.integers(0, 999), b"%d" % rng.integers(0, 9999), q, t])
        assert oracle.parse_pre(line) == oracle.ref_parse_pre(line)


def test_simple_aligner_kat():
    """test/cpp/SimpleAlignerTest.cpp:8-21: the only pin the reference holds on the blasr boundary."""
    oracle.build()
    k = json.load(open(os.path.join(GOLD, "kat_graph.json")))["simple_aligner"]
    start, end, q, t = oracle.simple_align(k["start"], k["tlen"], k["strand"].encode(), k["qstr"].encode(), k["tstr"].encode())
    assert t.decode() == k["expected_tstr"] and start == k["expected_start"]
    assert end == k["end"]


def test_banded_align_is_a_global_alignment():
    """Properties of the twin (parity unpinned beyond the KAT): the strings spell the inputs, no
    all-gap column, optimal against a full dynamic programme when the band covers the matrix."""
    oracle.build()
    rng = np.random.default_rng(9)

    def full_dp(q, t):
        n, m = len(q), len(t)
        S = np.zeros((n + 1, m + 1), dtype=np.int64)
        S[:, 0] = 4 * np.arange(n + 1); S[0, :] = 5 * np.arange(m + 1)
        for i in range(1, n + 1):
            for j in range(1, m + 1):
                S[i, j] = min(S[i - 1, j - 1] + (-5 if q[i - 1] == t[j - 1] else 6), S[i - 1, j] + 4, S[i, j - 1] + 5)
        return int(S[n, m])

    def score(qa, ta):
        s = 0
        for a, b in zip(qa, ta):
            s += 4 if b == 0x2D else 5 if a == 0x2D else (-5 if a == b else 6)
        return s

    for i in range(60):
        n = int(rng.integers(0, 50))
        t = bytes(b"ACGT"[j] for j in rng.integers(0, 4, int(rng.integers(0, 50))))
        q = bytearray()
        for c in t[:n] if i % 2 else bytes(b"ACGT"[j] for j in rng.integers(0, 4, n)):
            u = rng.random()
            if u < 0.1:
                continue
            q.append(c if u > 0.2 else b"ACGT"[rng.integers(0, 4)])
            if rng.random() < 0.1:
                q.append(b"ACGT"[rng.integers(0, 4)])
        q = bytes(q)
        qa, ta = oracle.banded_align(q, t)
        assert len(qa) == len(ta)
        assert qa.replace(b"-", b"") == q and ta.replace(b"-", b"") == t
        assert all(not (a == 0x2D and b == 0x2D) for a, b in zip(qa, ta))
        if q and t and oracle.align_halfwidth(len(q), len(t)) >= max(len(q), len(t)):
            assert score(qa, ta) == full_dp(q, t)


def test_cli_parses_pre_like_the_reference(tmp_path):
    """pbdagcon -a --dump-parsed (parser only, no GPU) against the reference-made vectors."""
    cli = os.path.join(ROOT, "pbdagcon_amd", "bin", "pbdagcon")
    if not os.path.exists(cli):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pbdagcon_amd", "csrc"), "all"])
    vec = json.load(open(os.path.join(GOLD, "pre_vectors.json")))["parsed"]
    path = tmp_path / "in.pre"
    path.write_text("\n".join(v["line"] for v in vec) + "\n")
    for extra in ([], ["--slab-bytes", "300", "-j", "3"]):
        out = subprocess.run([cli, "-a", "--dump-parsed", *extra, str(path)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        got = [ln.split("\t") for ln in out.stdout.splitlines()]
        assert len(got) == len(vec)
        for v, g in zip(vec, got):
            # (records of one target carry one tlen: the dump prints the group's)
            assert [g[0], g[2], g[3], g[4], g[5], g[6]] == [v["id"], str(v["start"]), v["strand"], v["sid"], v["qstr"], v["tstr"]]
    bad = tmp_path / "bad.pre"
    bad.write_text("q t + 10 0 3\n")
    assert subprocess.run([cli, "-a", "--dump-parsed", str(bad)], capture_output=True).returncode == 1


def test_following_band_agrees_with_the_static_band(monkeypatch):
    """The band that follows the alignment (tried first on pairs of more than ~1.1 kb) finds, on reads at ~15 % error
    without long indels, the alignment the static band finds; a pair with a 120-base deletion makes it give up (its
    path would hug the band's edge) and the static band decides."""
    oracle.build()
    rng = np.random.default_rng(21)

    def mutate(t, ins=0.10, dele=0.04, sub=0.01):
        q = bytearray()
        for c in t:
            u = rng.random()
            if u < dele:
                continue
            q.append(b"ACGT"[rng.integers(0, 4)] if u < dele + sub else c)
            while rng.random() < ins:
                q.append(b"ACGT"[rng.integers(0, 4)])
        return bytes(q)

    pairs = []
    for n in (1500, 3000, 6000, 9000):
        t = bytes(b"ACGT"[j] for j in rng.integers(0, 4, n))
        pairs.append((mutate(t), t))
    t = bytes(b"ACGT"[j] for j in rng.integers(0, 4, 5000))
    pairs.append((mutate(t[:2000] + t[2120:]), t))                  # 120 target bases the read does not have
    follow = [oracle.banded_align(q, t) for q, t in pairs]
    monkeypatch.setenv("OG_NO_ADAPTIVE", "1")
    static = [oracle.banded_align(q, t) for q, t in pairs]
    assert follow == static
    for (qa, ta), (q, t) in zip(follow, pairs):
        assert qa.replace(b"-", b"") == q and ta.replace(b"-", b"") == t
