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
PROFILE_DIR = os.path.join(ROOT, "profiles", "r02")


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


# ------------------------------------------------------------------ streaming (N = 1 point of configs[3])
def stream_core(ctxs, batches, B):
    """B batches (round-robin over `batches`) through the two contexts `ctxs`: an uploader thread one batch ahead
    of the thread that waits for results.  Returns (seconds, consensus bases, summed device ms, first results)."""
    import threading
    import queue
    import torch
    nd = len(batches)
    ready = [queue.Queue(), queue.Queue()]
    free = [threading.Semaphore(1), threading.Semaphore(1)]

    def uploader():
        for i in range(B):
            k = i & 1
            free[k].acquire()
            ctxs[k].upload(batches[i % nd])
            ctxs[k].run()
            ready[k].put(i)

    th = threading.Thread(target=uploader)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th.start()
    bases = 0
    dev_ms = 0.0
    first = None
    for i in range(B):
        k = i & 1
        ready[k].get()
        raw = ctxs[k].fetch_raw()
        tm = ctxs[k].timings()
        dev_ms += tm["ms_total"]
        bases += tm["consensus_bases"]
        if first is None:
            first = ctxs[k].results_to_py(raw)
        free[k].release()
    th.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, bases, dev_ms, first


def stream_batches(args):
    """B batches of args.targets targets through one GPU: two contexts on two streams, an uploader
    thread one batch ahead of the thread that waits for results, inputs in page-locked host memory
    (dagcon_host_alloc) so the copy runs at link speed.  The batches come from `--stream-distinct`
    distinct synthetic batches used round-robin (generating 100 x 0.9 GB on the host would time
    the generator)."""
    import threading
    import queue
    import torch
    from pbdagcon_amd import capi, synth
    torch.cuda.set_device(0)
    opts = dict(min_cov=6, min_len=500, trim=50)
    B = args.stream_batches
    nd = max(1, min(args.stream_distinct, B))
    thr = min(16, len(os.sched_getaffinity(0)))
    ctxs = [capi.Context(device=0, **opts) for _ in range(2)]
    batches = []
    for i in range(nd):
        b = synth.make_batch(args.targets, args.tlen, args.coverage, seed=1000, first_target=i * args.targets, threads=thr)
        batches.append(ctxs[0].pin_batch(b))
    # warm both contexts (arena allocation, first-use growth)
    for c in ctxs:
        c.upload(batches[0]); c.run(); c.fetch()
    dt, bases, dev_ms, first = stream_core(ctxs, batches, B)
    text_bytes = sum(int(b.qstr.size) * 2 for b in batches) / nd * B
    line = {
        "metric": "consensus bases/sec (whole node)", "value": bases / dt, "unit": "bases/s", "n_gpus": 1,
        "steps": B, "warmup": 1, "ms_per_step": dt / B * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"configs[3] at N=1: {B} batches x {args.targets} targets x {args.tlen} bp x {args.coverage}x "
                               f"streamed through one GPU ({nd} distinct batches round-robin), host->device copy of every "
                               "batch INSIDE the clock, page-locked input blobs, two contexts in flight",
                   "targets_total": B * args.targets},
        "h2d_GBps": text_bytes / dt / 1e9,
        "device_ms_per_batch": dev_ms / B,
        "targets_per_s": B * args.targets / dt,
    }
    print(json.dumps(line), flush=True)
    for c in ctxs:
        c.close()
    return 0


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


def e2e_leg(batch, n_targets, expect_fasta):
    """file -> FASTA through pbdagcon_amd/bin/pbdagcon on a slice of the workload; the .m5 text sits on
    tmpfs so that the number is the front end + device path, not a disk."""
    exe = os.path.join(ROOT, "pbdagcon_amd", "bin", "pbdagcon")
    if not os.path.exists(exe):
        return None
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    path = os.path.join(tmpdir, f"dagcon_bench_{os.getpid()}.m5")
    try:
        sub = batch.select(range(min(n_targets, batch.n_targets)))
        size = write_m5(sub, path)
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
        return {"value": bases / dt, "unit": "bases/s", "targets": sub.n_targets, "wall_s": dt,
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
        return {"value": bases / dt, "unit": "bases/s", "targets": n_targets, "tlen": tlen, "coverage": coverage, "wall_s": dt,
                "text_GBps": size / dt / 1e9, "pre_bytes": size,
                "what": "pbdagcon_amd/bin/pbdagcon -a <file.pre on tmpfs> -> FASTA (config-3 shape), process start, parse, "
                        "re-alignment of every record, consensus, formatting all inside the clock; best of 2"}
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


def load_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this round (collected in
    separate --pmc passes, tools/pmc_summary.py); None when there is none for this workload."""
    try:
        pmc = json.load(open(os.path.join(PROFILE_DIR, "pmc_hbm_traffic.json")))
        key = [k for k in pmc["kernels"] if k.startswith(kernel)][0]
        return pmc["kernels"][key]["hbm_bytes_per_launch_raw"], pmc.get("source", "profiles/r02/pmc_hbm_traffic.json")
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
        if not args.no_legs and n_gpus == 1:
            # host->device copy inside the clock: dagcon_consensus on the warm context (pageable blobs),
            # then with the blobs in page-locked memory (dagcon_host_alloc)
            t1 = time.perf_counter()
            r2 = ctx.consensus(batch)
            d1 = time.perf_counter() - t1
            pinned = ctx.pin_batch(batch)
            t1 = time.perf_counter()
            r3 = ctx.consensus(pinned)
            d2 = time.perf_counter() - t1
            h2d_bytes = 2 * int(batch.qstr.size)
            line["h2d_inclusive"] = {
                "value": bases_rank / d2, "unit": "bases/s", "ms": d2 * 1e3, "ms_pageable": d1 * 1e3,
                "value_pageable": bases_rank / d1, "h2d_bytes": h2d_bytes,
                "same_results": r2 == res and r3 == res,
                "what": "dagcon_consensus (host filter + H2D of the strings + kernels + D2H) on a warm context, one "
                        "batch, nothing overlapped; `value` with the blobs page-locked by dagcon_host_alloc",
            }
            if n_gpus == 1:
                # the same batch again and again through TWO contexts (what the CLI does on long inputs): one batch's
                # upload and result copy run beside the other's kernels, every upload inside the clock
                ctx2 = capi.Context(device=local_rank, **opts)
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
                line["two_contexts"] = {
                    "value": rbases / rdt, "unit": "bases/s", "steps": nr, "ms_per_step": rdt / nr * 1e3,
                    "what": "inputs resident as for `value`, but two contexts in flight on one GPU (own non-blocking stream "
                            "each): one batch's memory-bound kernels and tails run beside the other's issue-bound ones",
                }
                ctx2.close()
                line["streamed"] = {
                    "value": sbases / sdt, "unit": "bases/s", "batches": nb, "ms_per_batch": sdt / nb * 1e3,
                    "h2d_GBps": h2d_bytes * nb / sdt / 1e9, "same_results": sfirst == res,
                    "what": "8 batches of this workload through two contexts in flight on one GPU, page-locked blobs, "
                            "host->device copy of every batch and the result copy inside the clock",
                }
            line["e2e"] = e2e_leg(batch, args.e2e_targets, fasta_bytes(batch.select(range(min(args.e2e_targets, batch.n_targets))),
                                                                      res[:args.e2e_targets]))
            if n_gpus == 1 and config1:
                line["e2e_pre"] = e2e_pre_leg(64, 50000, 60, dict(min_cov=8, min_len=500, trim=50))
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
    ap.add_argument("--stream-distinct", type=int, default=4)
    ap.add_argument("--rehearse", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse:
        return rehearse(args, rank, world)
    if args.stream_batches:
        return stream_batches(args)
    return worker(args, rank, world, local_rank)


if __name__ == "__main__":
    sys.exit(main())
