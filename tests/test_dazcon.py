"""dazcon front end: the first-party arithmetic of DazAlnProvider.cpp restated in
pbdagcon_amd/bin/dazcon, against the reference's TargetHitTest known answers and a Python model;
the .las / .db readers and DALIGNER's trace-point realigner are parity-unpinned and not rebuilt."""
import json
import os
import subprocess

import numpy as np
import pytest

import daz_model as dm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "pbdagcon_amd", "bin", "dazcon")


def _write(tmp_path, reads, lines):
    s, a = tmp_path / "reads.txt", tmp_path / "ovl.txt"
    s.write_text("".join(f"{i} {seq}\n" for i, seq in sorted(reads.items())))
    a.write_text("\n".join(lines) + "\n")
    return ["-s", str(s), "-a", str(a)]


def _cli():
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pbdagcon_amd", "csrc"), "all"])
    return CLI


def test_target_hit_kats(tmp_path):
    """test/cpp/TargetHitTest.cpp:4-77: ovlScore 6986 / 3770 / (3770 then) 4721."""
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_dazcon.json")))
    rng = np.random.default_rng(1)
    seq = lambda n: "".join("ACGT"[k] for k in rng.integers(0, 4, n))
    for case in kat["cases"]:
        reads = {1: seq(9000), 2: seq(9000)}
        for upto in range(1, len(case["paths"]) + 1):
            lines = [f"O 1 2 0 {p['abpos']} {p['aepos']} {p['bbpos']} {p['bepos']} {p['diffs']} A A" for p in case["paths"][:upto]]
            out = subprocess.run([_cli(), *_write(tmp_path, reads, lines), "--dump-hits", "-c", "0"], capture_output=True, text=True)
            assert out.returncode == 0, out.stderr
            f = out.stdout.split()
            # EXPECT_FLOAT_EQ: within 4 units in the last place (the float expression gives 4721.00049)
            want = np.float32(case["scores"][upto - 1])
            assert abs(np.float32(f[3]) - want) <= 4 * np.spacing(want) and int(f[5]) == upto, case["name"]
            # and the model gives the same float, bit for bit
            h = dm.group_hits([dict(aread=0, bread=1, flags=0, **p) for p in case["paths"][:upto]], 9000, {1: 9000})
            assert f"{float(h[0].ovl):.9g}" == f[3]


@pytest.mark.parametrize("flags", [[], ["-x"], ["-o"], ["-m", "3"], ["-x", "-m", "4", "-o"]])
def test_hit_selection_matches_model(tmp_path, flags):
    """Grouping of records into hits, scores, -o, the sort (and the coverage sort of -x), -m."""
    rng = np.random.default_rng(len(" ".join(flags)) + 3)
    alen = 3000
    reads = {1: "A" * alen}
    recs = []
    for b in range(2, 14):
        blen = int(rng.integers(500, 4000))
        reads[b] = "C" * blen
        fl = int(rng.integers(0, 2))
        pos = int(rng.integers(0, 300)) if rng.random() < 0.7 else 0
        for _ in range(int(rng.integers(1, 4))):
            ln = int(rng.integers(200, 1500))
            ab, ae = pos, min(alen, pos + ln)
            bb = 0 if rng.random() < 0.3 else int(rng.integers(0, 50))
            be = blen if rng.random() < 0.3 else min(blen, bb + ln + int(rng.integers(-20, 20)))
            recs.append(dict(aread=0, bread=b - 1, flags=fl, abpos=ab, aepos=ae, bbpos=bb, bepos=be, diffs=int(rng.integers(0, 200))))
            pos = ae + int(rng.integers(-100, 200))
            if pos >= alen - 10:
                break
    lines = [f"O 1 {r['bread'] + 1} {r['flags']} {r['abpos']} {r['aepos']} {r['bbpos']} {r['bepos']} {r['diffs']} A A" for r in recs]
    out = subprocess.run([_cli(), *_write(tmp_path, reads, lines), "--dump-hits", "-c", "0", *flags], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    hits = dm.group_hits(recs, alen, {b - 1: len(s) for b, s in reads.items()}, proper="-o" in flags)
    hits = dm.sort_hits(hits, alen, "-x" in flags)
    if "-m" in flags:
        hits = hits[:int(flags[flags.index("-m") + 1])]
    exp = [f"1\t{h.bread + 1}\t{h.flags}\t{float(h.ovl):.9g}\t{float(h.cov):.9g}\t{len(h.records)}" for h in hits]
    assert out.stdout.splitlines() == exp


def test_decode_alignment_and_filters(tmp_path):
    """decodeAlignment (DazAlnProvider.cpp:383-417) on trace records, B complemented for COMP
    overlaps; the target list and the minimum coverage of nextTarget (:79-117)."""
    rng = np.random.default_rng(12)
    reads, lines, model = dm.synth_dataset(rng, n_targets=3, tlen=(300, 500), n_b=(7, 9), raw_fraction=1.0)
    args = _write(tmp_path, reads, lines)
    out = subprocess.run([_cli(), *args, "--dump-alns"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = [ln.split("\t") for ln in out.stdout.splitlines()]
    exp = []
    for aid, recs in model.items():
        hits = dm.sort_hits(dm.group_hits(recs, len(reads[aid]), {r["bread"]: len(reads[r["bread"] + 1]) for r in recs}), len(reads[aid]), False)
        for h in hits:
            for r in h.records:
                t, q = dm.decode(reads[aid], r["b_oriented"], r)
                assert (t, q) == (r["tstr"], r["qstr"])           # the model inverts the generator
                exp.append([str(aid), str(r["abpos"] + 1), t, q])
    assert got == exp
    first = next(iter(model))
    only = subprocess.run([_cli(), *args, "--dump-alns", str(first)], capture_output=True, text=True)
    assert [ln.split("\t")[0] for ln in only.stdout.splitlines()] == [str(first)] * len(model[first])
    none = subprocess.run([_cli(), *args, "--dump-alns", "-c", "50"], capture_output=True, text=True)
    assert none.returncode == 0 and none.stdout == ""


def test_flags_and_errors(tmp_path):
    assert subprocess.run([_cli()], capture_output=True).returncode == 1                      # -a and -s are required
    assert subprocess.run([_cli(), "-a", "x"], capture_output=True).returncode == 1
    assert subprocess.run([_cli(), "-a", "x", "-s", str(tmp_path / "nope")], capture_output=True).returncode == 1
    assert "0.3" in subprocess.run([_cli(), "--version"], capture_output=True, text=True).stdout
    s = tmp_path / "r.txt"; s.write_text("1 ACGT\n2 ACGT\n")
    a = tmp_path / "o.txt"; a.write_text("O 1 2 0 0 9 0 4 0 ACGT ACGT\n")                    # A interval outside the read
    assert subprocess.run([_cli(), "-a", str(a), "-s", str(s), "--dump-hits"], capture_output=True).returncode == 1


@pytest.mark.gpu
def test_dazcon_end_to_end(tmp_path):
    """reads + overlaps -> dazcon -> FASTA: real backbone (dazcon.cpp:76), -t 10 default, the
    record format of dazcon.cpp:92-97 with the well counter from 0; against the oracle on the
    alignments the model says the front end must produce."""
    import oracle
    rng = np.random.default_rng(3)
    reads, lines, model = dm.synth_dataset(rng, n_targets=4, tlen=(1500, 2600), n_b=(8, 14))
    out = subprocess.run([_cli(), *_write(tmp_path, reads, lines), "--batch-targets", "3"], capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    exp, well = [], 0
    for aid, recs in model.items():
        hits = dm.sort_hits(dm.group_hits(recs, len(reads[aid]), {r["bread"]: len(reads[r["bread"] + 1]) for r in recs}), len(reads[aid]), False)
        alns = [(r["abpos"] + 1, r["qstr"].encode(), r["tstr"].encode()) for h in hits for r in h.records]
        for r0, r1, s in oracle.consensus_target(len(reads[aid]), alns, 500, 10, 6, backbone=reads[aid].encode()):
            exp.append(b">%d/%d/%d_%d\n%s\n" % (aid, well, r0, r1, s))
            well += 1
    assert out.stdout == b"".join(exp) and len(exp) >= 4
