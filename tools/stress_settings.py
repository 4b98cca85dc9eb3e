"""One round of tools/stress.py, every (context setting, emit shift) pair in a process of its own: which of them fails.
    python tools/stress_settings.py <seed> <round> [VAR=VALUE ...]
(with a third argument 'one <k> <shift>' it is the child: runs that pair and exits 0 / 1)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
seed, rnd = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "one":
    import stress
    from pbdagcon_amd import capi
    from util import oracle_batch
    k, shift = int(sys.argv[4]), sys.argv[5]
    b, desc, min_cov, min_len, trim, kws = stress.make_round(seed, rnd)
    if shift != "-": os.environ["DAGCON_EMIT_SHIFT"] = shift
    exp = oracle_batch(b, min_cov, min_len, trim)
    ctx = capi.Context(min_cov=min_cov, min_len=min_len, trim=trim, **kws[k])
    reps = int(os.environ.get("STRESS_REPS", "1"))
    ok = True
    for _ in range(reps):
        try:
            got = ctx.consensus(b)
        except capi.DagconError as e:
            got = str(e)
        ok = ok and got == exp
    print("timings", {kk: round(v, 2) for kk, v in ctx.timings().items() if kk.startswith("ms_")}, flush=True)
    ctx.close()
    sys.exit(0 if ok else 1)
for kv in sys.argv[3:]:
    kk, v = kv.split("="); os.environ[kk] = v
for k in range(3):
    for shift in ("-", "4", "6"):
        r = subprocess.run([sys.executable, __file__, str(seed), str(rnd), "one", str(k), shift], capture_output=True, text=True, timeout=300)
        tail = (r.stdout + r.stderr).strip().splitlines()[-2:]
        print(f"setting {k} shift {shift}: rc {r.returncode}", " | ".join(tail)[:300], flush=True)
