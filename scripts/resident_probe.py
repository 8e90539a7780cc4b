"""XCD-resident iteration (k_resident) against the two-kernel iteration, on the GPU box: same results bit for bit?
how fast?  One line per case: single pairs of every BASELINE config (10k), the 5k config 2, and the 64-pair headline
batch; `RESIDENT` is switched per context with cvo_ctx_set_option."""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from unified_cvo_amd import CvoGPU  # noqa: E402

NP = int(os.environ.get("PROBE_PAIRS", "64"))
REPS = int(os.environ.get("PROBE_REPS", "3"))
from unified_cvo_amd import build as _hipbuild  # noqa: E402
RESIDENT_LIB = _hipbuild.build_resident()  # k_resident is not part of the default library
opts = {k[4:]: v for k, v in os.environ.items() if k.startswith("RES_")}   # RES_RESIDENT_BLOCKS=12 -> option


def run_single(name, builder, kw, max_it=0):
    P, a, b, init = builder(**kw)
    out = {}
    for mode in ("two-kernel", "resident"):
        gpu = CvoGPU(params=P, library=RESIDENT_LIB)
        if mode == "resident":
            gpu.set_option(os.environ.get("PROBE_ALT", "RESIDENT"), "1")
        for k, v in opts.items():
            gpu.set_option(k, v)
        da, db = gpu.upload(a), gpu.upload(b)
        kwargs = dict(max_iterations=max_it) if max_it else {}
        gpu.align(da, db, init, max_iterations=50)
        best = None
        for _ in range(REPS):
            r = gpu.align(da, db, init, **kwargs)
            if best is None or r.seconds < best.seconds:
                best = r
        out[mode] = best
        if mode == "resident" and "PHASE_TICKS" in opts:
            tk, nb = gpu.debug_resident_ticks()   # (summed over the warm-up and the timed calls)
            n, nt = max(tk[8], 1), max(tk[14], 1)
            names = ["wait head", "rows1", "arrive A", "wait twist", "rows2", "arrive B"]
            tnames = ["wait arrivals A", "twist_finalize", "wait arrivals B", "update until the head leaves", "rest of the update"]
            print("    %d row blocks per pair; row block 0, us per iteration: " % nb +
                  ", ".join(f"{nm} {tk[q] / n / 100.0:.2f}" for q, nm in enumerate(names)) + "; tail block: " +
                  ", ".join(f"{nm} {tk[9 + q] / nt / 100.0:.2f}" for q, nm in enumerate(tnames)), flush=True)
        gpu.close()
    t, r = out["two-kernel"], out["resident"]
    same = (t.iterations == r.iterations) and np.array_equal(t.transform, r.transform)
    print(f"{name:34s} iterations {r.iterations:5d}: two-kernel {t.seconds*1e6/max(t.iterations,1):6.2f} us/it, resident "
          f"{r.seconds*1e6/max(r.iterations,1):6.2f} us/it  ({'bit-identical' if same else 'DIFFERENT: max|d| = %.3g, iterations %d vs %d' % (np.max(np.abs(t.transform - r.transform)), t.iterations, r.iterations)})",
          flush=True)


def run_batch(n_pairs, n=10000, max_it=0):
    cs = [cases.config2(n=n, pair_id=p) for p in range(n_pairs)]
    P = cs[0][0]
    out = {}
    for mode in ("two-kernel", "resident"):
        gpu = CvoGPU(params=P, library=RESIDENT_LIB)
        if mode == "resident":
            gpu.set_option("RESIDENT", "1")
        for k, v in opts.items():
            gpu.set_option(k, v)
        clouds = gpu.upload_many([c[1] for c in cs] + [c[2] for c in cs])
        s, t = clouds[:n_pairs], clouds[n_pairs:]
        inits = [c[3] for c in cs]
        kwargs = dict(max_iterations=max_it) if max_it else {}
        gpu.align_batch(s, t, inits, **kwargs)
        best, res = 1e9, None
        for _ in range(REPS):
            t0 = time.perf_counter()
            res = gpu.align_batch(s, t, inits, **kwargs)
            best = min(best, time.perf_counter() - t0)
        builds, iters, cand = gpu.debug_list_builds()
        out[mode] = (best, res, builds, iters)
        gpu.close()
    (tt, tr, tb, ti), (rt, rr, rb, ri) = out["two-kernel"], out["resident"]
    same = all(a.iterations == b.iterations and np.array_equal(a.transform, b.transform) for a, b in zip(tr, rr))
    print(f"batch of {n_pairs} x {n}: two-kernel {tt*1e3:7.2f} ms ({n_pairs/tt:7.1f} align/s), resident {rt*1e3:7.2f} ms "
          f"({n_pairs/rt:7.1f} align/s); list builds {tb} / {rb}; {'bit-identical' if same else 'DIFFERENT'}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["single", "batch"]
    if "single" in what:
        run_single("config2 shape 10k x 10k xyz", cases.config2, dict(n=10000))
        run_single("config2 5k x 5k xyz", cases.config2, dict(n=5000))
        run_single("config3 10k colour", cases.config3, dict(n=10000))
        run_single("config4 10k semantic warm start", cases.config4, dict(n=10000))
    if "batch" in what:
        run_batch(NP)
    if "batch16" in what:
        run_batch(16)
