"""Host logic on CPU: synthetic generator, sharding, the FASTA gather over a
2-rank gloo group, the .m5 reader."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_is_deterministic_and_shard_independent():
    from pbdagcon_amd import synth
    a = synth.make_batch(6, 400, 8, seed=42)
    b = synth.make_batch(6, 400, 8, seed=42)
    assert np.array_equal(a.qstr, b.qstr) and np.array_equal(a.tstr, b.tstr)
    # a target's data depends only on its global index, not on the shard it is generated in
    c = synth.make_batch(3, 400, 8, seed=42, first_target=3)
    for t in range(3):
        assert c.target_alignments(t) == a.target_alignments(3 + t)
    assert c.ids[0] == a.ids[3]


def test_synth_shape():
    from pbdagcon_amd import synth
    b = synth.make_batch(2, 2000, 10, seed=5, with_backbone=True)
    for t in range(2):
        for s, q, tt in b.target_alignments(t):
            assert len(q) == len(tt) and s == 1
            tb = bytes(c for c in tt if c != 0x2D)
            o = int(b.backbone_off[t])
            assert tb == b.backbone[o:o + 2000].tobytes()       # target side spells the backbone
            assert not any(x == 0x2D and y == 0x2D for x, y in zip(q, tt))
    p = synth.make_batch(4, 1000, 6, seed=9, min_span=0.6)
    for t in range(4):
        for s, q, tt in p.target_alignments(t):
            nt = sum(1 for c in tt if c != 0x2D)
            assert nt >= 600 and s - 1 + nt <= 1000


def test_shard_ranges_cover_and_balance():
    from pbdagcon_amd.shard import shard_ranges
    rng = np.random.default_rng(0)
    w = rng.integers(1, 1000, 1000)
    for world in (1, 2, 3, 8):
        r = shard_ranges(w, world)
        assert r[0][0] == 0 and r[-1][1] == 1000
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        loads = [w[a:b].sum() for a, b in r]
        assert max(loads) - min(loads) <= 2 * w.max()
    assert shard_ranges([], 4) == [(0, 0)] * 4


WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
from pbdagcon_amd import synth
from pbdagcon_amd.shard import shard_ranges, gather_fasta
from util import oracle_batch
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
full = synth.make_batch(7, 300, 8, seed=3)
w = [int(full.aln_len[int(full.aln_begin[t]):int(full.aln_begin[t+1])].sum()) for t in range(7)]
lo, hi = shard_ranges(w, world)[rank]
mine = full.select(range(lo, hi))
# the compute leg of this CPU test is the checker; on the GPU box the same
# plumbing carries the HIP results (bench.py)
res = oracle_batch(mine, 4, 100, 10)
def fasta(b, rs):
    return b"".join(b">%s/%d_%d\n%s\n" % (b.ids[t].encode(), r0, r1, s) for t, segs in enumerate(rs) for r0, r1, s in segs)
got = gather_fasta(fasta(mine, res), dist, torch)
if rank == 0:
    exp = fasta(full, oracle_batch(full, 4, 100, 10))
    assert got == exp, "gathered FASTA differs from the single-rank result"
    assert len(exp) > 0
    print("GATHER_OK", len(got))
else:
    assert got is None
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_gather(tmp_path):
    """world_size 2 on CPU: contiguous shards + the gather reproduce the
    single-rank FASTA byte for byte."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
        env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GATHER_OK" in out.stdout


def _cli():
    path = os.path.join(ROOT, "pbdagcon_amd", "bin", "pbdagcon")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pbdagcon_amd", "csrc"), "all"])
    return path


def test_cli_parser_matches_reference_parser(tmp_path):
    """The C++ front end's -m 5 parser (pbdagcon --dump-parsed, no GPU involved) against the
    oracle's parseM5, which tests/test_oracle.py pins to the reference's own Alignment.cpp:
    '+' and '-' strands, repeated spaces, the reference's two fixture shapes."""
    import oracle
    from pbdagcon_amd import synth
    oracle.build()
    b = synth.make_batch(3, 120, 5, seed=11)
    m5 = synth.to_m5(b).decode().splitlines()
    # flip some records to the '-' strand (strings are stored reverse-complemented there)
    lines = []
    for i, ln in enumerate(m5):
        f = ln.split(" ")
        if i % 3 == 1:
            f[9] = "-"
        if i % 4 == 2:
            f[4] = f[4] + " "            # double space: empty fields are skipped (Alignment.cpp:50-53)
        lines.append(" ".join(f))
    lines.insert(0, "id 3 0 3 +  ref 3 0 3 + -40645 8129 0 0 0 254 CAC |-| CGC")
    lines.insert(1, "id 3 0 3 +  ref 3 0 3 - -40645 8129 0 0 0 254 GGCCAATT |-| AATTGGCC")
    path = tmp_path / "in.m5"
    path.write_text("\n".join(lines) + "\n\n")
    out = subprocess.run([_cli(), "--dump-parsed", str(path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    got = [ln.split("\t") for ln in out.stdout.splitlines()]
    assert len(got) == len(lines)
    for ln, g in zip(lines, got):
        e = oracle.parse_m5(ln.encode(), True)
        assert g == [e["id"].decode(), str(e["tlen"]), str(e["start"]), e["strand"].decode(),
                     e["sid"].decode(), e["qstr"].decode(), e["tstr"].decode()]
    # the text is indexed a slab at a time by -j threads; a target that straddles slab edges is
    # carried over: tiny slabs, odd thread counts, same lines out
    for slab, j in ((200, 3), (1000, 1), (5000, 7)):
        out2 = subprocess.run([_cli(), "--dump-parsed", "--slab-bytes", str(slab), "-j", str(j), str(path)],
                              capture_output=True, text=True, timeout=60)
        assert out2.returncode == 0 and out2.stdout == out.stdout, (slab, j)
    assert got[0][5:] == ["CAC", "CGC"] and got[1][5:] == ["AATTGGCC", "GGCCAATT"]   # AlignmentTest.cpp:49-63


def test_cli_flags_and_errors(tmp_path):
    cli = _cli()
    assert subprocess.run([cli], capture_output=True).returncode == 2                 # input is required
    assert subprocess.run([cli, "-a", "x.pre"], capture_output=True).returncode == 1  # -a: .pre input (a missing file is an error)
    assert subprocess.run([cli, "--dump-parsed", str(tmp_path / "nope.m5")], capture_output=True).returncode == 1
    bad = tmp_path / "bad.m5"
    bad.write_text("only three fields\n")
    assert subprocess.run([cli, "--dump-parsed", str(bad)], capture_output=True).returncode == 1
    v = subprocess.run([cli, "--version"], capture_output=True, text=True)
    assert "0.3" in v.stdout


def _bench(*args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`bench.py --gpus 2` with nothing in the environment starts two ranks (child processes,
    gloo on CPU here), splits the target index space with shard_ranges and verifies the gathered
    FASTA against the SHA-256 every rank took of its own part (--rehearse: records are
    fabricated, no device).  The reference starts its N consensus workers itself too
    (main.cpp:251-274)."""
    import json
    out = _bench("--gpus", "2", "--backend", "gloo", "--rehearse", "--targets", "37")
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["fasta_gather_ok"] is True
    assert line["targets_total"] == 74 and line["records"] == 74 and line["shard"] == [0, 37]
    # three ranks, and the driver's way (ranks from the environment of torch.distributed.run)
    out = _bench("--gpus", "3", "--backend", "gloo", "--rehearse", "--targets", "5")
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert out.returncode == 0 and line["n_gpus"] == 3 and line["fasta_gather_ok"] is True and line["records"] == 15
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rehearse",
         "--targets", "11"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["fasta_gather_ok"] is True and line["records"] == 22


def test_bench_streams_the_global_target_space_over_ranks():
    """configs[3] as specified (main.cpp:251-274: N workers drain ONE queue of all targets): `bench.py --gpus 3
    --stream-batches 4` gives every rank its shard of the global target space, each streams it in batches and the
    FASTA is gathered on rank 0 per super-batch; --rehearse fabricates the records (no device).  Also a total that
    does not divide (ragged last batches, ranks with different batch counts)."""
    import json
    out = _bench("--gpus", "3", "--backend", "gloo", "--rehearse", "--stream-batches", "4", "--targets", "8",
                 "--tlen", "300", "--coverage", "5", "--gather-every", "3")
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["fasta_gather_ok"] is True and line["backend"] == "gloo"
    assert line["config"]["targets_total"] == 96 and line["fasta_records"] == 96 and line["targets_done"] == 96
    assert line["config"]["shards"] == [[0, 32], [32, 64], [64, 96]] and line["gather_rounds"] == 2
    out = _bench("--gpus", "3", "--backend", "gloo", "--rehearse", "--targets-total", "101", "--targets", "8",
                 "--tlen", "300", "--coverage", "5")
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["fasta_gather_ok"] is True and line["fasta_records"] == 101 and line["targets_done"] == 101
    assert [b - a for a, b in line["config"]["shards"]] == [34, 34, 33]
    out = _bench("--rehearse", "--stream-batches", "3", "--targets", "5")           # one rank: no process group
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert out.returncode == 0 and line["fasta_gather_ok"] is True and line["fasta_records"] == 15


def test_bench_without_a_gpu_fails_loudly():
    """The measured path has no CPU stand-in: without a device bench.py ends with an error."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    out = _bench("--steps", "1", "--warmup", "0", "--targets", "2", "--tlen", "600", "--coverage", "8")
    assert out.returncode != 0


def test_shard_ranges_cover_and_balance():
    import numpy as np
    from pbdagcon_amd.shard import shard_ranges
    rng = np.random.default_rng(3)
    for n, world in ((1000, 8), (7, 3), (3, 8), (0, 2), (100000, 8)):
        w = rng.integers(1, 100, n).astype(float)
        r = shard_ranges(w, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == n
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        if n >= 100 * world:
            tot = [w[a:b].sum() for a, b in r]
            assert max(tot) <= 1.05 * w.sum() / world + 100


def test_cli_flag_parsing_devices():
    cli = _cli()
    assert subprocess.run([cli, "--devices"], capture_output=True).returncode == 2
    assert subprocess.run([cli, "--devices", "0,x", "f.m5"], capture_output=True).returncode == 2
    assert subprocess.run([cli, "--devices", "", "f.m5"], capture_output=True).returncode == 2
