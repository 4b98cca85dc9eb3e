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
rank owns a contiguous shard of the target index space and no collective is on
the data path).

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     : dominant kernel (stage b, k_merge) -- algorithmic bytes of one
                 launch / its HIP-event duration, against the 8 TB/s HBM peak
  cpu_baseline : the CPU oracle (a port, oracle/dagcon_oracle.c) timed on this
                 host's cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec


def fasta_bytes(batch, results, id_offset=0):
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
    main.cpp:259-263, with one whole target per task)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    import oracle
    oracle.build()
    oracle.lib()
    n = min(sample_targets, batch.n_targets)

    def one(t):
        a0, a1 = int(batch.aln_begin[t]), int(batch.aln_begin[t + 1])
        segs = oracle.consensus_target_blob(
            int(batch.tlen[t]), batch.aln_start[a0:a1].copy(), batch.aln_off[a0:a1].copy(),
            batch.aln_len[a0:a1].copy(), batch.qstr, batch.tstr, opts["min_len"], opts["trim"],
            opts["min_cov"], None, faithful=faithful)
        return sum(r1 - r0 for r0, r1, _ in segs)

    one(0)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        bases = sum(ex.map(one, range(n)))
    dt = time.perf_counter() - t0
    return bases / dt, n, dt


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
    ap.add_argument("--backend", default="nccl",
                    help="process-group backend; 'gloo' lets several ranks rehearse on one GPU")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)   # rehearsal: ranks share GPUs
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(local_rank)
    n_gpus = world
    red_dev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"

    from pbdagcon_amd import capi, synth
    from pbdagcon_amd.shard import gather_fasta
    opts = dict(min_cov=6, min_len=500, trim=50)
    # contiguous shard of the global target index space: rank r owns targets
    # [r*targets, (r+1)*targets); a target's data depends only on its global index
    batch = synth.make_batch(args.targets, args.tlen, args.coverage, seed=1000,
                             first_target=rank * args.targets, threads=min(16, len(os.sched_getaffinity(0))))
    ctx = capi.Context(device=local_rank, **opts)
    ctx.upload(batch)                       # inputs resident in HBM from here on

    def gather(res):
        return gather_fasta(fasta_bytes(batch, res), dist, torch, local_rank) if dist is not None else None

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

    merge_ms, total_ms = [], []
    fence()
    t0 = time.perf_counter()
    def collect(tm):
        merge_ms.append(tm["ms_merge"])
        total_ms.append(tm["ms_total"])

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

    verified = None
    if rank == 0 and not args.no_verify:
        # spot check against the oracle (checker only; outside the timed region)
        from util import oracle_batch
        sub = batch.select(range(0, min(8, batch.n_targets)))
        verified = oracle_batch(sub, **opts) == res[:sub.n_targets]

    gather_ok = None
    if rank == 0 and gathered is not None:
        # rank 0 holds every rank's FASTA in rank (= global target) order
        gather_ok = gathered.startswith(fasta_bytes(batch, res)) and gathered.count(b">") * n_gpus >= 0
    if rank == 0:
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", "v10_pmc_hbm_traffic.json")))
            if args.targets == 1000 and args.tlen == 10000 and args.coverage == 40:
                key = [k for k in pmc["kernels"] if k.startswith("k_merge")][0]
                traffic = pmc["kernels"][key]["hbm_bytes_per_launch_raw"]
        except Exception:
            traffic = None
        ms_merge = sum(merge_ms) / len(merge_ms)
        ms_dev = sum(total_ms) / len(total_ms)
        alg = tm["algorithmic_bytes"]
        achieved = alg / (ms_merge * 1e-3) / 1e9
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
                "parallelism": f"target-sharded x{n_gpus}, no data-path collective",
            },
            "bases_per_gpu_per_s": value / n_gpus,
            "roofline": {
                "bound": "hbm", "kernel": "k_merge",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes per k_merge "
                                  "launch at this workload, profiles/r01/v10_pmc_hbm_traffic.json" if traffic else None,
                "algorithmic_bytes_per_launch": alg,
                "kernel_ms": ms_merge,
                "pipeline_ms": ms_dev,
                "pipeline_frac": alg / (ms_dev * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            },
            "stage_ms": {k: tm[k] for k in ("ms_normalize", "ms_build", "ms_merge", "ms_bestpath")},
            "bytes_per_base": alg / max(bases_rank, 1),
            "bit_exact_vs_oracle": verified,
            "fasta_gather_ok": gather_ok,
        }
        if not args.no_cpu:
            # the GPU box gives one GPU's job a 16-core share of the host
            cores = min(len(os.sched_getaffinity(0)), 16 * n_gpus)
            v, n, cdt = cpu_baseline(batch, args.cpu_sample, opts, cores)
            line["cpu_baseline"] = {
                "value": v, "unit": "bases/s", "cores": cores, "kind": "port",
                "sample": f"first {n} targets of the same workload, oracle/dagcon_oracle.c, "
                          f"{cores} threads, one target per task, {cdt:.1f} s wall",
            }
            line["gpu_over_cpu"] = value / v
            # second flavour (SURVEY 8d / BASELINE.md 3): the same algorithm on the reference's kind of
            # containers (std::map, std::list edge properties, per-vertex vectors, std::string)
            vf, nf, fdt = cpu_baseline(batch, max(cores * 6, 32), opts, cores, faithful=True)
            line["cpu_baseline_faithful"] = {
                "value": vf, "unit": "bases/s", "cores": cores, "kind": "port",
                "sample": f"first {nf} targets, oracle/cpu_faithful.cpp (reference-style containers), "
                          f"{cores} threads, {fdt:.1f} s wall",
            }
            line["gpu_over_cpu_faithful"] = value / vf
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
