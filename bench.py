#!/usr/bin/env python3
"""bench.py -- consensus bases/s of the DAGCon hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path (normalise -> build -> merge -> best path)
over one batch of synthetic targets that is already resident in HBM, results
brought back to the host (and, for N > 1, the FASTA payload gathered on rank 0
over RCCL); the host-side conversion / gather of one step overlaps the device work of
the next, all K steps complete inside the timed region.  Workload at every N: BASELINE.json configs[1] per GPU -- 1,000
targets x 10 kb backbone x 40x coverage, sub/ins/del = 1 %/10 %/4 %, pbdagcon
defaults -c 6 -m 500 -t 50 -- so scaling is weak (targets are independent; each
rank owns a contiguous shard of the target index space, pbdagcon_amd/shard.py,
and no collective is on the data path).

`--gpus N` with no WORLD_SIZE in the environment starts the N ranks itself (child
processes, before this process has touched a GPU; the reference spawns its N
consensus workers itself too, main.cpp:251-274); under torch.distributed.run
the ranks are taken from the environment.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      : dominant kernel of the device pipeline -- algorithmic bytes of
                  one launch / its HIP-event duration, against the 8 TB/s HBM peak
  cpu_baseline  : the CPU oracle (a port, oracle/dagcon_oracle.c) timed on this
                  host's cores over a bounded sample of the same workload; its
                  segments double as the whole-batch parity check
  h2d_inclusive : the same batch with the host->device copy of the strings inside
                  the clock (dagcon_consensus on a warm context); never `value`
  two_contexts  : the timed workload again with two contexts in flight on the GPU (inputs
                  resident): what a caller that keeps two batches in flight gets
  streamed      : the same with every batch's host->device copy inside the clock
  e2e           : .m5 text of the workload on tmpfs -> the pbdagcon command line ->
                  FASTA (file to FASTA, process start included)
  e2e_pre       : configs[2]'s surface: .pre text of 64 targets x 50 kb x 60x ->
                  pbdagcon -a (re-alignment on the device) -> FASTA

Other modes (not the driver's line):
  --stream-batches B   B batches of --targets targets through ONE GPU, two contexts
                       in flight (upload of one overlaps the kernels of the other):
                       the N = 1 point of configs[3]
  --rehearse           launcher / sharding / gather rehearsal without a device:
                       every rank fabricates records for its shard (CPU tests)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
PROFILE_DIR = os.path.join(ROOT, "profiles", "r03")


def fasta_bytes(batch, results):
    """main.cpp:141-143 record format: >id/range0_range1\\nseq\\n"""
    out = []
    for t, segs in enumerate(results):
        tid = batch.ids[t]
        for r0, r1, seq in segs:
            out.append(b">%s/%d_%d\n%s\n" % (tid.encode(), r0, r1, seq))
    return b"".join(out)


def cpu_baseline(batch, sample_targets, opts, cores, faithful=False):
    """Time the oracle on `sample_targets` targets with `cores` threads (the C
    call releases the GIL: this is the reference's N-consensus-thread layout,
    main.cpp:259-263, with one whole target per task).  Returns the rate and the
    segments themselves (the parity check uses them)."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle
    oracle.build()
    oracle.lib()
    n = min(sample_targets, batch.n_targets)

    def one(t):
        a0, a1 = int(batch.aln_begin[t]), int(batch.aln_begin[t + 1])
        return oracle.consensus_target_blob(
            int(batch.tlen[t]), batch.aln_start[a0:a1].copy(), batch.aln_off[a0:a1].copy(),
            batch.aln_len[a0:a1].copy(), batch.qstr, batch.tstr, opts["min_len"], opts["trim"],
            opts["min_cov"], None, faithful=faithful)

    one(0)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        segs = list(ex.map(one, range(n)))
    dt = time.perf_counter() - t0
    bases = sum(r1 - r0 for s in segs for r0, r1, _ in s)
    return bases / dt, n, dt, segs


# ------------------------------------------------------------------ launcher
def launch_ranks(args):
    """Start args.gpus ranks of this script as child processes (one per GPU, torch.distributed
    rendezvous on 127.0.0.1) and relay rank 0's line.  Nothing in this process has touched HIP."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def sha(b):
    return hashlib.sha256(b).digest()


def verify_gather(gathered, sizes, digests):
    """rank 0: the gathered payload is the concatenation, in rank order, of exactly the parts
    the ranks hashed before sending."""
    if gathered is None or sum(sizes) != len(gathered):
        return False
    pos = 0
    for n, d in zip(sizes, digests):
        if sha(gathered[pos:pos + n]) != d:
            return False
        pos += n
    return True


def exchange_digests(part, dist, torch, dev):
    d = torch.frombuffer(bytearray(sha(part)), dtype=torch.uint8).to(dev)
    out = [torch.zeros(32, dtype=torch.uint8, device=dev) for _ in range(dist.get_world_size())]
    dist.all_gather(out, d)
    return [bytes(x.cpu().numpy().tobytes()) for x in out]


# ------------------------------------------------------------------ rehearsal (no device)
def rehearse(args, rank, world):
    """Launcher + shard_ranges + gather, with fabricated records: what the CPU tests run."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from pbdagcon_amd.shard import gather_fasta, shard_ranges
    if world > 1:
        dist.init_process_group("gloo")
    total = args.targets * world
    lo, hi = shard_ranges(np.full(total, args.tlen * args.coverage, dtype=np.float64), world)[rank]
    part = b"".join(b">t%07d/0_%d\n%s\n" % (t, args.tlen, hashlib.md5(b"%d" % t).hexdigest().encode() * (1 + t % 3))
                    for t in range(lo, hi))
    whole = b"".join(b">t%07d/0_%d\n%s\n" % (t, args.tlen, hashlib.md5(b"%d" % t).hexdigest().encode() * (1 + t % 3))
                     for t in range(total))
    if world > 1:
        gathered, sizes = gather_fasta(part, dist, torch, None, return_sizes=True)
        digests = exchange_digests(part, dist, torch, torch.device("cpu"))
    else:
        gathered, sizes, digests = part, [len(part)], [sha(part)]
    if rank == 0:
        ok = verify_gather(gathered, sizes, digests) and gathered == whole
        print(json.dumps({"rehearse": True, "n_gpus": world, "shard": [lo, hi], "targets_total": total,
                          "fasta_gather_ok": bool(ok), "records": gathered.count(b">")}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------ streaming: configs[3]
def raw_snapshot(raw):
    """What a dagcon_results struct points at, copied out (the context's host buffers are its own until its next fetch)."""
    import ctypes as C
    import numpy as np
    T, S = raw.n_targets, raw.n_segments
    blob = C.string_at(raw.seq_blob, raw.seq_bytes) if raw.seq_bytes else b""
    if not S:
        return T, [0] * (T + 1), [], [], [], [], blob
    return (T, np.ctypeslib.as_array(raw.seg_begin, (T + 1,)).tolist(), np.ctypeslib.as_array(raw.range0, (S,)).tolist(),
            np.ctypeslib.as_array(raw.range1, (S,)).tolist(), np.ctypeslib.as_array(raw.seq_off, (S,)).tolist(),
            np.ctypeslib.as_array(raw.seq_len, (S,)).tolist(), blob)


def snapshot_fasta(ids, snap):
    """FASTA records (main.cpp:141-143) of a snapshot."""
    T, begin, r0, r1, off, ln, blob = snap
    out = []
    for t in range(T):
        for s in range(begin[t], begin[t + 1]):
            out.append(b">%s/%d_%d\n" % (ids[t], r0[s], r1[s]))
            out.append(blob[off[s]:off[s] + ln[s]])
            out.append(b"\n")
    return b"".join(out)


def snapshot_results(snap):
    T, begin, r0, r1, off, ln, blob = snap
    return [[(r0[s], r1[s], blob[off[s]:off[s] + ln[s]]) for s in range(begin[t], begin[t + 1])] for t in range(T)]


def stream_core(ctxs, batches, B, on_result=None, sync=None):
    """B batches (round-robin over `batches`, or batches(i) when it is callable) through the two contexts
    `ctxs`: an uploader thread one batch ahead of the thread that waits for results.  on_result(i, raw)
    sees every batch's dagcon_results before its context is handed back.  Returns (seconds, consensus
    bases, summed device ms, first results)."""
    import threading
    import queue
    import torch
    get = batches if callable(batches) else (lambda i: batches[i % len(batches)])
    ready = [queue.Queue(), queue.Queue()]
    free = [threading.Semaphore(1), threading.Semaphore(1)]
    err = []

    def uploader():
        try:
            for i in range(B):
                k = i & 1
                free[k].acquire()
                ctxs[k].upload(get(i))
                ctxs[k].run()
                ready[k].put(i)
        except Exception as e:                     # hand the failure to the waiting thread
            err.append(e)
            for q in ready:
                q.put(-1)

    th = threading.Thread(target=uploader)
    (sync or torch.cuda.synchronize)()
    t0 = time.perf_counter()
    th.start()
    bases = 0
    dev_ms = 0.0
    first = None
    for i in range(B):
        k = i & 1
        if ready[k].get() < 0:
            th.join()
            raise err[0]
        raw = ctxs[k].fetch_raw()
        tm = ctxs[k].timings()
        dev_ms += tm["ms_total"]
        bases += tm["consensus_bases"]
        if first is None:
            first = ctxs[k].results_to_py(raw)
        if on_result is not None:
            on_result(i, raw)
        free[k].release()
    th.join()
    (sync or torch.cuda.synchronize)()
    return time.perf_counter() - t0, bases, dev_ms, first


def stream_plan(total, per_batch, world, tlen, coverage):
    """The global target space [0, total) cut into one contiguous shard per rank (shard_ranges: the reference's
    N workers drain one queue of all targets, main.cpp:251-274; here the queue is dealt up front) and every shard
    into batches of per_batch targets.  Returns [(lo, hi, [(first, n), ...])] per rank."""
    import numpy as np
    from pbdagcon_amd.shard import shard_ranges
    plan = []
    for lo, hi in shard_ranges(np.full(total, float(tlen) * coverage), world):
        plan.append((lo, hi, [(f, min(per_batch, hi - f)) for f in range(lo, hi, per_batch)]))
    return plan


def target_ids(first, n, tlen):
    return [b"t%07d/0_%d" % (first + k, tlen) for k in range(n)]


def fake_records(first, n, tlen):
    return b"".join(b">t%07d/0_%d/0_%d\n%s\n" % (t, tlen, tlen, hashlib.md5(b"%d" % t).hexdigest().encode() * (1 + t % 3))
                    for t in range(first, first + n))


def stream_worker(args, rank, world, local_rank, quiet=False):
    """configs[3] as specified: --targets-total targets (default --targets x --stream-batches per rank) over
    `world` ranks.  Every rank takes its shard of the GLOBAL target space, streams it in batches of --targets
    through two contexts on its own GPU (upload of one batch beside the kernels of the other, host->device
    copy of every batch inside the clock), and every --gather-every batches the FASTA made so far is gathered
    on rank 0 (RCCL when the group is nccl).  Rank 0 checks the gathered payload against the size and SHA-256
    every rank kept of its own part.  Only the first --stream-distinct batches of a shard are generated
    (100 x 0.9 GB of synthetic strings would time the generator); later batches re-use their strings under
    their own target ids, and their sequences must come out identical to the batch they re-use.
    Returns the line (rank 0) or None."""
    import numpy as np
    dist = None
    rehearsal = args.rehearse
    if not rehearsal:
        import torch
        if args.backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl" and not rehearsal:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    on_gpu_group = dist is not None and dist.get_backend() == "nccl"
    red_dev = None
    if dist is not None:
        red_dev = torch.device("cuda", local_rank) if on_gpu_group else torch.device("cpu")
    from pbdagcon_amd.shard import gather_fasta
    opts = dict(min_cov=6, min_len=500, trim=50)
    total = args.targets_total or args.targets * args.stream_batches * world
    plan = stream_plan(total, args.targets, world, args.tlen, args.coverage)
    lo, hi, mine = plan[rank]
    B = len(mine)
    n_rounds = max(-(-len(p[2]) // args.gather_every) for p in plan)      # gathers every rank takes part in
    nd = max(1, min(args.stream_distinct, B))
    thr = min(16, len(os.sched_getaffinity(0)))

    ctxs, distinct = [], []
    if not rehearsal:
        from pbdagcon_amd import capi, synth
        ctxs = [capi.Context(device=local_rank, **opts) for _ in range(2)]
        for j in range(nd):
            f, n = mine[j]
            distinct.append(ctxs[0].pin_batch(synth.make_batch(n, args.tlen, args.coverage, seed=1000, first_target=f, threads=thr)))
        for c in ctxs:                              # warm both contexts (arena allocation, first-use growth)
            c.upload(distinct[0]); c.run(); c.fetch()

    def batch_of(i):
        f, n = mine[i]
        src = distinct[i % nd]
        return src if n == src.n_targets else src.select(range(n))

    parts = []                      # this rank's FASTA, one piece per batch
    seq_digest = {}                 # digest of the sequences of a distinct batch (ids left out)
    reuse_ok = [True]
    keep_first = {}

    def seqs_only(fa):
        return hashlib.sha256(b"\n".join(fa.split(b"\n")[1::2])).digest()       # (every record is a header line and a sequence line)

    gathered_parts = [[] for _ in range(world)]
    state = {"done": 0, "round": 0, "sent": 0}
    my_hash = hashlib.sha256()
    my_size = [0]

    def gather_round():
        payload = b"".join(parts[state["sent"]:])
        state["sent"] = len(parts)
        my_hash.update(payload)
        my_size[0] += len(payload)
        state["round"] += 1
        if dist is None:
            gathered_parts[0].append(payload)
            return
        out, sizes = gather_fasta(payload, dist, torch, local_rank if on_gpu_group else None, return_sizes=True)
        if rank == 0:
            pos = 0
            for r, n in enumerate(sizes):
                gathered_parts[r].append(out[pos:pos + n])
                pos += n

    all_ids = [target_ids(f, n, args.tlen) for f, n in mine]

    # the thread that fetches only copies a batch's results out and hands the context back; records are formatted,
    # digested and (every --gather-every batches) gathered by a writer thread, in order -- the reference's Writer
    # (main.cpp:164-175) beside its consensus workers
    import queue
    import threading
    wq = queue.Queue()
    werr = []

    def writer():
        try:
            while True:
                item = wq.get()
                if item is None:
                    return
                i, snap = item
                f, n = mine[i]
                fa = snapshot_fasta(all_ids[i], snap)
                parts.append(fa)
                if n == distinct[i % nd].n_targets:
                    d = seqs_only(fa)
                    if i < nd:
                        seq_digest[i] = d
                        keep_first[i] = snapshot_results(snap) if args.stream_verify else None
                    elif seq_digest.get(i % nd) != d:
                        reuse_ok[0] = False
                state["done"] += 1
                if state["done"] % args.gather_every == 0:
                    gather_round()
        except Exception as e:
            werr.append(e)

    def on_result(i, raw):
        wq.put((i, raw_snapshot(raw)))

    def fence():
        if dist is not None:
            dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    bases, dev_ms = 0, 0.0
    if rehearsal:
        for i, (f, n) in enumerate(mine):
            parts.append(fake_records(f, n, args.tlen))
            state["done"] += 1
            if state["done"] % args.gather_every == 0:
                gather_round()
    elif B:
        wt = threading.Thread(target=writer)
        wt.start()
        try:
            _, bases, dev_ms, _ = stream_core(ctxs, batch_of, B, on_result)
        finally:
            wq.put(None)
            wt.join()
        if werr:
            raise werr[0]
    while state["round"] < n_rounds:                 # the tail, and the rounds a shorter shard sits out with nothing to send
        gather_round()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        bt = torch.tensor([bases, hi - lo], dtype=torch.int64, device=red_dev)
        dist.all_reduce(bt)
        bases_all, targets_all = int(bt[0].item()), int(bt[1].item())
    else:
        bases_all, targets_all = bases, hi - lo

    # ---- outside the clock: parity of the distinct batches against the oracle, the gather against the digests ----
    verified, n_verified = True, 0
    if args.stream_verify and not rehearsal:
        from util import oracle_batch
        for j in range(nd):
            k = distinct[j].n_targets if args.stream_verify < 0 else min(args.stream_verify, distinct[j].n_targets)
            exp = oracle_batch(distinct[j].select(range(k)), **opts)
            verified = verified and exp == keep_first[j][:k]
            n_verified += k
    gather_ok = None
    if dist is not None:
        flag = torch.tensor([1 if (verified and reuse_ok[0]) else 0, n_verified], dtype=torch.int64, device=red_dev)
        mn, sm = flag.clone(), flag.clone()
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        dist.all_reduce(sm)
        all_ok, n_verified = bool(mn[0].item()), int(sm[1].item())
        me = torch.frombuffer(bytearray(my_hash.digest() + my_size[0].to_bytes(8, "little")), dtype=torch.uint8).to(red_dev)
        outs = [torch.zeros(40, dtype=torch.uint8, device=red_dev) for _ in range(world)]
        dist.all_gather(outs, me)
        claims = [bytes(x.cpu().numpy().tobytes()) for x in outs]
    else:
        all_ok = verified and reuse_ok[0]
        claims = [my_hash.digest() + my_size[0].to_bytes(8, "little")]
    line = None
    if rank == 0:
        whole = [b"".join(g) for g in gathered_parts]            # rank order = global target order (contiguous shards)
        gather_ok = all(sha(w) == c[:32] and len(w) == int.from_bytes(c[32:], "little") for w, c in zip(whole, claims))
        payload = b"".join(whole)
        if rehearsal:
            gather_ok = gather_ok and payload == fake_records(0, total, args.tlen)
        records = payload.count(b">")
        text_bytes = 0 if rehearsal else sum(int(b.qstr.size) * 2 for b in distinct) / nd * B
        line = {
            "metric": "consensus bases/sec (whole node)", "value": bases_all / dt, "unit": "bases/s", "n_gpus": world,
            "steps": B, "warmup": 1, "ms_per_step": dt / max(B, 1) * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"configs[3]: {total} targets x {args.tlen} bp x {args.coverage}x over {world} GPU(s): every rank "
                                   f"streams its shard of the global target space in batches of {args.targets} through two contexts "
                                   f"(the first {nd} batches of a shard are generated, later ones re-use their strings under their own ids), "
                                   f"host->device copy of every batch INSIDE the clock, FASTA gathered on rank 0 every {args.gather_every} batches",
                       "targets_total": total, "batches_rank0": B, "shards": [[p[0], p[1]] for p in plan]},
            "targets_done": targets_all, "targets_per_s": targets_all / dt,
            "h2d_GBps_rank0": text_bytes / dt / 1e9, "device_ms_per_batch_rank0": dev_ms / max(B, 1),
            "fasta_bytes": len(payload), "fasta_records": records, "gather_rounds": n_rounds,
            "fasta_gather_ok": bool(gather_ok), "bit_exact_vs_oracle": bool(all_ok) if args.stream_verify and not rehearsal else None,
            "targets_verified": n_verified, "reused_batches_identical": bool(all_ok) if not rehearsal else None,
            "backend": "none" if dist is None else dist.get_backend(),
            "batch0_sha256": hashlib.sha256(parts[0]).hexdigest() if parts else None,
        }
        if rehearsal:
            line["rehearse"] = True
        if not quiet:
            print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()
    return line


# ------------------------------------------------------------------ legs outside the timed region
def write_m5(batch, path):
    """The batch as BLASR -m 5 text (Alignment.cpp:44-80 field layout), '+' strand."""
    import numpy as np
    with open(path, "wb") as f:
        for t in range(batch.n_targets):
            tid = batch.ids[t]
            for k, (start, q, tt) in enumerate(batch.target_alignments(t)):
                qa, ta = np.frombuffer(q, np.uint8), np.frombuffer(tt, np.uint8)
                nq, nt = int(np.count_nonzero(qa != 45)), int(np.count_nonzero(ta != 45))
                match = np.where(qa == ta, np.uint8(124), np.uint8(42)).tobytes()
                f.write(b"q%07d_%d/0_%d %d 0 %d + %s %d %d %d + -1000 0 0 0 0 254 " % (
                    t, k, nq, nq, nq, tid.encode(), int(batch.tlen[t]), start - 1, start - 1 + nt))
                f.write(q); f.write(b" "); f.write(match); f.write(b" "); f.write(tt); f.write(b"\n")
    return os.path.getsize(path)


def e2e_leg(batch, n_targets, expect_fasta, repeat=1):
    """file -> FASTA through pbdagcon_amd/bin/pbdagcon on a slice of the workload; the .m5 text sits on
    tmpfs so that the number is the front end + device path, not a disk.  repeat > 1: the same text that many
    times over (target ids repeat, never side by side: every copy is its own set of groups, BlasrM5AlnProvider.cpp:45-50),
    so that process start-up is a small part of the run."""
    exe = os.path.join(ROOT, "pbdagcon_amd", "bin", "pbdagcon")
    if not os.path.exists(exe):
        return None
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    path = os.path.join(tmpdir, f"dagcon_bench_{os.getpid()}.m5")
    try:
        sub = batch.select(range(min(n_targets, batch.n_targets)))
        size = write_m5(sub, path)
        if repeat > 1:
            blk = open(path, "rb").read()
            with open(path, "ab") as f:
                for _ in range(repeat - 1):
                    f.write(blk)
            del blk
            size *= repeat
            expect_fasta = expect_fasta * repeat
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            out = subprocess.run([exe, "-j", str(min(16, len(os.sched_getaffinity(0)))), path], capture_output=True)
            dt = time.perf_counter() - t0
            if out.returncode != 0:
                return {"error": out.stderr.decode()[-300:]}
            if best is None or dt < best[0]:
                best = (dt, out.stdout)
        dt, fasta = best
        bases = sum(len(l) for l in fasta.split(b"\n") if l and not l.startswith(b">"))
        return {"value": bases / dt, "unit": "bases/s", "targets": sub.n_targets * repeat, "wall_s": dt,
                "text_GBps": size / dt / 1e9, "m5_bytes": size,
                "fasta_identical_to_device_path": fasta == expect_fasta,
                "what": "pbdagcon_amd/bin/pbdagcon <file.m5 on tmpfs> -> FASTA, process start, parse, upload, kernels, "
                        "formatting all inside the clock; best of 2"}
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


def e2e_pre_leg(n_targets, tlen, coverage, opts):
    """configs[2]'s input surface: .pre records (unaligned query / target substrings, Alignment.cpp:82-112) of
    n_targets x tlen x coverage on tmpfs -> `pbdagcon -a` (every record re-aligned on the device, then the usual
    path) -> FASTA.  The aligner is parity-unpinned beyond the reference's one KAT."""
    from pbdagcon_amd import synth
    exe = os.path.join(ROOT, "pbdagcon_amd", "bin", "pbdagcon")
    if not os.path.exists(exe):
        return None
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    path = os.path.join(tmpdir, f"dagcon_bench_{os.getpid()}.pre")
    try:
        b = synth.make_batch(n_targets, tlen, coverage, seed=7000, threads=min(16, len(os.sched_getaffinity(0))))
        with open(path, "wb") as f:
            for t in range(b.n_targets):
                tid = b.ids[t].encode()
                for k, (start, q, tt) in enumerate(b.target_alignments(t)):
                    ts = tt.replace(b"-", b"")
                    f.write(b"q%07d_%d %s + %d %d %d %s %s\n" % (t, k, tid, int(b.tlen[t]), start - 1, start - 1 + len(ts),
                                                                  q.replace(b"-", b""), ts))
        size = os.path.getsize(path)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            out = subprocess.run([exe, "-a", "-c", str(opts["min_cov"]), "-m", str(opts["min_len"]), "-t", str(opts["trim"]),
                                  "-j", str(min(16, len(os.sched_getaffinity(0)))), path], capture_output=True)
            dt = time.perf_counter() - t0
            if out.returncode != 0:
                return {"error": out.stderr.decode()[-300:]}
            if best is None or dt < best[0]:
                best = (dt, out.stdout)
        dt, fasta = best
        bases = sum(len(l) for l in fasta.split(b"\n") if l and not l.startswith(b">"))
        # outside the clock: the first and the last target through the CPU twin of the aligner (SimpleAligner.cpp:25-63
        # restated: parity unpinned beyond the reference's one KAT) + the oracle's consensus; their records must be in the
        # FASTA byte for byte
        import oracle
        from concurrent.futures import ThreadPoolExecutor
        checked, same = 0, True
        for t in sorted({0, b.n_targets - 1}):
            recs = b.target_alignments(t)

            def one(rec):
                start, q, tt = rec
                ts = tt.replace(b"-", b"")
                st, en, qa, ta = oracle.simple_align(start - 1, int(b.tlen[t]), b"+", q.replace(b"-", b""), ts)
                return (st, qa, ta)

            with ThreadPoolExecutor(max_workers=min(16, len(os.sched_getaffinity(0)))) as ex:
                alns = list(ex.map(one, recs))
            exp = b"".join(b">%s/%d_%d\n%s\n" % (b.ids[t].encode(), r0, r1, sq) for r0, r1, sq in
                           oracle.consensus_target(int(b.tlen[t]), alns, opts["min_len"], opts["trim"], opts["min_cov"]))
            same = same and len(exp) > 0 and exp in fasta
            checked += 1
        return {"value": bases / dt, "unit": "bases/s", "targets": n_targets, "tlen": tlen, "coverage": coverage, "wall_s": dt,
                "text_GBps": size / dt / 1e9, "pre_bytes": size, "fasta_checked": checked, "fasta_identical_to_twin_and_oracle": bool(same),
                "what": "pbdagcon_amd/bin/pbdagcon -a <file.pre on tmpfs> -> FASTA (config-3 shape), process start, parse, "
                        "re-alignment of every record, consensus, formatting all inside the clock; best of 2; two targets "
                        "re-done on the CPU (twin aligner + oracle) outside the clock"}
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


def config5_leg(local_rank):
    """The dazcon path's shape from the decoded strings inward (BASELINE configs[4]; its .las/.db surface is not built):
    400 targets of 2-40 kb, 30 reads each spanning >= 60 % of their target, real backbone, -t 10.  Device time of the
    second run on a warm context; all 400 targets against the digest the CPU oracle produced (tests/golden/large_hashes.json)."""
    import importlib.util
    import numpy as np
    from pbdagcon_amd import capi, synth
    spec = importlib.util.spec_from_file_location("make_large_hashes", os.path.join(ROOT, "tests", "golden", "make_large_hashes.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "large_hashes.json")))["config5_400xmixedx30_partial"]["sha256"]
    tl = np.random.default_rng(5).integers(2000, 40000, 400)
    b5 = synth.make_batch(400, 0, 30, seed=8000, min_span=0.6, tlens=tl, with_backbone=True, threads=min(16, len(os.sched_getaffinity(0))))
    ctx = capi.Context(device=local_rank, min_cov=6, min_len=500, trim=10)
    try:
        ctx.upload(b5); ctx.run(); ctx.fetch()
        best = None
        for _ in range(3):
            ctx.run(); res = ctx.fetch(); tm = ctx.timings()
            if best is None or tm["ms_total"] < best["ms_total"]:
                best = tm
    finally:
        ctx.close()
    bases = sum(len(sq) for segs in res for _, _, sq in segs)
    return {"value": bases / (best["ms_total"] * 1e-3), "unit": "bases/s", "targets": 400, "consensus_bases": bases,
            "ms": best["ms_total"], "stage_ms": {k: best[k] for k in ("ms_normalize", "ms_build", "ms_merge", "ms_bestpath")},
            "pieces": best["merge_segments"], "roofline_frac": best["algorithmic_bytes"] / (best["ms_total"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "digest_matches_oracle": m.digest(res) == golden,
            "what": "config-5 shape (400 targets x 2-40 kb x 30x, spans >= 60 %, real backbone, -c 6 -m 500 -t 10), inputs "
                    "resident, device pipeline of one batch (best of 3), all 400 targets against the committed oracle digest"}


def load_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this round (collected in
    separate --pmc passes, tools/pmc_summary.py); None when there is none for this workload."""
    try:
        pmc = json.load(open(os.path.join(PROFILE_DIR, "pmc_hbm_traffic.json")))
        key = [k for k in pmc["kernels"] if k.startswith(kernel)][0]
        return pmc["kernels"][key]["hbm_bytes_per_launch_raw"], pmc.get("source", "profiles/r03/pmc_hbm_traffic.json")
    except Exception:
        return None, None


# ------------------------------------------------------------------ the driver's line
def worker(args, rank, world, local_rank):
    import numpy as np
    import torch
    dist = None
    if args.backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)   # rehearsal: ranks share GPUs
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    n_gpus = world
    red_dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")

    from pbdagcon_amd import capi, synth
    from pbdagcon_amd.shard import gather_fasta, shard_ranges
    opts = dict(min_cov=6, min_len=500, trim=50)
    # contiguous shards of the global target index space, balanced by alignment bytes; a target's data
    # depends only on its global index
    total = args.targets * world
    lo, hi = shard_ranges(np.full(total, float(args.tlen) * args.coverage), world)[rank]
    batch = synth.make_batch(hi - lo, args.tlen, args.coverage, seed=1000,
                             first_target=lo, threads=min(16, len(os.sched_getaffinity(0))))
    ctx = capi.Context(device=local_rank, **opts)
    ctx.upload(batch)                       # inputs resident in HBM from here on

    def gather(res):
        if dist is None:
            return None
        return gather_fasta(fasta_bytes(batch, res), dist, torch, local_rank, return_sizes=True)

    def run_steps(n, collect=None):
        """n passes of the hot path over the batch, each with its results brought to the host
        (dagcon_run + dagcon_fetch).  While pass i+1 is on the device the host turns pass i's
        results into records and (N > 1) gathers their FASTA over RCCL; the last pass's records and
        gather are done before this returns."""
        res = gathered = None
        if n:
            ctx.run()
        for i in range(n):
            raw = ctx.fetch_raw()
            if collect is not None:
                collect(ctx.timings())
            if i + 1 < n:
                ctx.run()
            res = ctx.results_to_py(raw)
            gathered = gather(res)
        return res, gathered

    run_steps(args.warmup)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stage = {k: [] for k in ("ms_total", "ms_normalize", "ms_build", "ms_merge", "ms_bestpath")}
    fence()
    t0 = time.perf_counter()

    def collect(tm):
        for k in stage:
            stage[k].append(tm[k])

    res, gathered = run_steps(args.steps, collect)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    tm = ctx.timings()
    bases_rank = tm["consensus_bases"]
    if dist is not None:
        bt = torch.tensor([bases_rank], dtype=torch.int64, device=red_dev)
        dist.all_reduce(bt)
        bases_all = int(bt.item())
    else:
        bases_all = bases_rank
    value = bases_all * args.steps / dt

    # ---- everything below is outside the timed region ----
    my_fasta = fasta_bytes(batch, res)
    cores = min(len(os.sched_getaffinity(0)), 16)
    cpu = None
    verified, verified_targets = None, 0
    if not args.no_verify or not args.no_cpu:
        # at N = 1 rank 0 runs the oracle over its whole shard (that run is also the cpu_baseline leg); at N > 1
        # every rank checks a sample of its shard
        n_chk = args.cpu_sample if rank == 0 and n_gpus == 1 else min(64 if rank == 0 else 16, batch.n_targets)
        v, n, cdt, segs = cpu_baseline(batch, n_chk, opts, cores)
        verified = segs == res[:n]
        verified_targets = n
        cpu = (v, n, cdt)
    gather_ok = None
    if dist is not None:
        flag = torch.tensor([1 if verified in (True, None) else 0, verified_targets], dtype=torch.int64, device=red_dev)
        mn = flag.clone()
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        sm = flag.clone()
        dist.all_reduce(sm)
        all_verified, verified_total = bool(mn[0].item()), int(sm[1].item())
        digests = exchange_digests(my_fasta, dist, torch, red_dev)
        if rank == 0:
            payload, sizes = gathered
            gather_ok = verify_gather(payload, sizes, digests) and payload.count(b">") >= total
    else:
        all_verified, verified_total = verified, verified_targets

    if rank == 0:
        ms = {k: sum(v) / len(v) for k, v in stage.items()}
        dom = max(("ms_normalize", "ms_build", "ms_merge", "ms_bestpath"), key=lambda k: ms[k])
        dom_kernel = {"ms_normalize": "k_norm_chunk", "ms_build": "k_emit", "ms_merge": "k_merge_q", "ms_bestpath": "k_bp_sweep"}[dom]
        alg = tm["algorithmic_bytes"]
        achieved = alg / (ms[dom] * 1e-3) / 1e9
        config1 = args.targets == 1000 and args.tlen == 10000 and args.coverage == 40
        traffic, traffic_src = load_traffic(dom_kernel) if config1 else (None, None)
        line = {
            "metric": "consensus bases/sec (whole node)",
            "value": value,
            "unit": "bases/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"configs[1]: {args.targets} targets x {args.tlen} bp backbone x {args.coverage}x "
                            "per GPU, .m5-layout alignment strings resident in HBM, -c 6 -m 500 -t 50",
                "targets_per_gpu": args.targets, "tlen": args.tlen, "coverage": args.coverage,
                "parallelism": f"target-sharded x{n_gpus} (contiguous ranges, shard_ranges), no data-path collective",
            },
            "bases_per_gpu_per_s": value / n_gpus,
            "roofline": {
                "bound": "hbm", "kernel": f"stage {dom[3:]} ({dom_kernel} and its helpers: the longest stage of the pipeline)",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg,
                "kernel_ms": ms[dom],
                "pipeline_ms": ms["ms_total"],
                "pipeline_frac": alg / (ms["ms_total"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "note": "the kernels are instruction-issue / latency bound, not bandwidth bound (DESIGN.md section 4); "
                        "the HBM roofline is the bound the path would have if they were not",
            },
            "stage_ms": {k: ms[k] for k in ("ms_normalize", "ms_build", "ms_merge", "ms_bestpath")},
            "merge_segments": tm["merge_segments"],
            "bytes_per_base": alg / max(bases_rank, 1),
            "bit_exact_vs_oracle": bool(all_verified) if all_verified is not None else None,
            "targets_verified": verified_total,
            "fasta_gather_ok": gather_ok,
            "fasta_sha256": hashlib.sha256(my_fasta).hexdigest() if n_gpus == 1 else None,
        }
        if cpu is not None and not args.no_cpu and n_gpus == 1:      # (the CPU legs belong to the N = 1 line)
            v, n, cdt = cpu
            line["cpu_baseline"] = {
                "value": v, "unit": "bases/s", "cores": cores, "kind": "port",
                "sample": f"first {n} targets of the same workload, oracle/dagcon_oracle.c, "
                          f"{cores} threads, one target per task, {cdt:.1f} s wall",
            }
            line["gpu_over_cpu"] = value / n_gpus / v
            # second flavour (SURVEY 8d / BASELINE.md 3): the same algorithm on the reference's kind of
            # containers (std::map, std::list edge properties, per-vertex vectors, std::string)
            vf, nf, fdt, segf = cpu_baseline(batch, max(cores * 6, 32), opts, cores, faithful=True)
            line["cpu_baseline_faithful"] = {
                "value": vf, "unit": "bases/s", "cores": cores, "kind": "port",
                "sample": f"first {nf} targets, oracle/cpu_faithful.cpp (reference-style containers), "
                          f"{cores} threads, {fdt:.1f} s wall",
                "same_segments_as_device": segf == res[:nf],
            }
            line["gpu_over_cpu_faithful"] = value / n_gpus / vf
        def leg(name, fn):
            """A leg outside the timed region must never take the line down with it: its key says what went wrong."""
            try:
                out = fn()
                if isinstance(out, dict) and name is None:
                    line.update(out)
                elif name is not None:
                    line[name] = out
            except Exception as e:      # noqa: BLE001
                line[name or "legs_error"] = {"error": f"{type(e).__name__}: {e}"[:300]}

        if not args.no_legs and n_gpus == 1:
            state = {}

            def leg_h2d():
                # host->device copy inside the clock: dagcon_consensus on the warm context (pageable blobs),
                # then with the blobs in page-locked memory (dagcon_host_alloc)
                t1 = time.perf_counter()
                r2 = ctx.consensus(batch)
                d1 = time.perf_counter() - t1
                pinned = ctx.pin_batch(batch)
                state["pinned"] = pinned
                t1 = time.perf_counter()
                r3 = ctx.consensus(pinned)
                d2 = time.perf_counter() - t1
                h2d_bytes = 2 * int(batch.qstr.size)
                return {
                    "value": bases_rank / d2, "unit": "bases/s", "ms": d2 * 1e3, "ms_pageable": d1 * 1e3,
                    "value_pageable": bases_rank / d1, "h2d_bytes": h2d_bytes,
                    "same_results": r2 == res and r3 == res,
                    "what": "dagcon_consensus (host filter + H2D of the strings + kernels + D2H) on a warm context, one "
                            "batch, nothing overlapped; `value` with the blobs page-locked by dagcon_host_alloc",
                }

            def leg_two():
                # the same batch again and again through TWO contexts (what the CLI does on long inputs): one batch's
                # upload and result copy run beside the other's kernels, every upload inside the clock
                pinned = state.get("pinned") or ctx.pin_batch(batch)
                h2d_bytes = 2 * int(batch.qstr.size)
                ctx2 = capi.Context(device=local_rank, **opts)
                try:
                    ctx2.upload(pinned); ctx2.run(); ctx2.fetch()
                    nb = 8
                    sdt, sbases, sdev, sfirst = stream_core([ctx, ctx2], [pinned], nb)
                    # ... and with the inputs resident (what `value` times, but with two batches in flight)
                    nr = 8
                    both = [ctx, ctx2]
                    ctx.upload(pinned)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    inflight = []
                    rbases = 0
                    for i in range(nr):
                        c2 = both[i & 1]
                        if len(inflight) == 2:
                            cf = inflight.pop(0); cf.fetch_raw(); rbases += cf.timings()["consensus_bases"]
                        c2.run(); inflight.append(c2)
                    for cf in inflight:
                        cf.fetch_raw(); rbases += cf.timings()["consensus_bases"]
                    torch.cuda.synchronize()
                    rdt = time.perf_counter() - t1
                finally:
                    ctx2.close()
                return {
                    "two_contexts": {
                        "value": rbases / rdt, "unit": "bases/s", "steps": nr, "ms_per_step": rdt / nr * 1e3,
                        "what": "inputs resident as for `value`, but two contexts in flight on one GPU (own non-blocking stream "
                                "each): one batch's memory-bound kernels and tails run beside the other's issue-bound ones",
                    },
                    "streamed": {
                        "value": sbases / sdt, "unit": "bases/s", "batches": nb, "ms_per_batch": sdt / nb * 1e3,
                        "h2d_GBps": h2d_bytes * nb / sdt / 1e9, "same_results": sfirst == res,
                        "what": "8 batches of this workload through two contexts in flight on one GPU, page-locked blobs, "
                                "host->device copy of every batch and the result copy inside the clock",
                    },
                }

            def leg_c3():
                # the N = 1 point of configs[3]: 12 batches of this workload's size streamed through two contexts,
                # uploads inside the clock; batch 0 is the batch `value` was measured on
                ctx.close()
                a3 = argparse.Namespace(**vars(args))
                a3.stream_batches, a3.stream_distinct, a3.targets_total, a3.gather_every = 12, 2, 0, 4
                a3.stream_verify, a3.rehearse = 64, False
                c3 = stream_worker(a3, 0, 1, local_rank, quiet=True)
                c3["batch0_identical_to_value_run"] = c3.pop("batch0_sha256") == hashlib.sha256(my_fasta).hexdigest()
                return c3

            first_fasta = fasta_bytes(batch.select(range(min(args.e2e_targets, batch.n_targets))), res[:args.e2e_targets])
            leg("h2d_inclusive", leg_h2d)
            leg(None, leg_two)
            leg("e2e", lambda: e2e_leg(batch, args.e2e_targets, first_fasta))
            if config1:
                leg("e2e_pre", lambda: e2e_pre_leg(64, 50000, 60, dict(min_cov=8, min_len=500, trim=50)))
                leg("e2e_4000", lambda: e2e_leg(batch, args.e2e_targets, first_fasta, repeat=4))
                leg("config5_shape", lambda: config5_leg(local_rank))
            leg("configs3_n1", leg_c3)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--targets", type=int, default=1000, help="targets per GPU (configs[1]: 1000)")
    ap.add_argument("--tlen", type=int, default=10000)
    ap.add_argument("--coverage", type=int, default=40)
    ap.add_argument("--cpu-sample", type=int, default=1000, help="targets timed on the CPU oracle (1000 = ~17 s of CPU work)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the h2d_inclusive and e2e legs")
    ap.add_argument("--e2e-targets", type=int, default=1000)
    ap.add_argument("--backend", default="nccl",
                    help="process-group backend; 'gloo' lets several ranks rehearse on one GPU")
    ap.add_argument("--stream-batches", type=int, default=0)
    ap.add_argument("--stream-distinct", type=int, default=4, help="batches of a shard that are really generated")
    ap.add_argument("--targets-total", type=int, default=0,
                    help="configs[3]: size of the GLOBAL target space shared out over the ranks (100000); "
                         "default --targets x --stream-batches per rank")
    ap.add_argument("--gather-every", type=int, default=4, help="batches per FASTA gather on rank 0 (a super-batch)")
    ap.add_argument("--stream-verify", type=int, default=64,
                    help="targets of every generated batch checked against the oracle (-1: all, 0: none)")
    ap.add_argument("--rehearse", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.stream_batches or args.targets_total:
        if not args.stream_batches:
            args.stream_batches = -(-args.targets_total // (args.targets * world))
        stream_worker(args, rank, world, local_rank)
        return 0
    if args.rehearse:
        return rehearse(args, rank, world)
    return worker(args, rank, world, local_rank)


if __name__ == "__main__":
    sys.exit(main())
