#!/usr/bin/env python
"""One-stop performance probe for the optimisation loop (MI355X): the numbers VERDICT r1 asks to move.

  * headline batch (64 x 10k x 10k geometric): ms for the first 16 / 64 / 256 iterations and for the whole align
    (early phase share), list builds, waits;
  * single pair in flight: us per iteration for the 10k geometric shape, config 3 (colour) and config 4 (semantic,
    warm start) - what frame-to-frame tracking pays;
  * with CVO_PHASE_TICKS=1: per-block phase stamps of k_assoc / k_coeff and the update tail (stderr of the library).

usage: perf_probe.py [out.json]      (set CVO_PHASE_TICKS=1 for the stamps)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from unified_cvo_amd import CvoGPU  # noqa: E402

out = {}
P = cases.load_params("geometric_gpu")
NP = int(os.environ.get("PROBE_PAIRS", "64"))
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
src, tgt = both[:NP], both[NP:]
inits = [a[3] for a in pairs]
gpu.align_batch(src, tgt, inits, max_iterations=64)
batch = {}
for it in (16, 64, 256, 2000):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        res = gpu.align_batch(src, tgt, inits, max_iterations=it)
        best = min(best, time.perf_counter() - t0)
    builds, iters, cand = gpu.debug_list_builds()
    batch[str(it)] = {"ms": round(best * 1e3, 3), "device_ms": round(res[0].seconds * 1e3, 3), "list_builds": builds,
                      "candidates_per_row_iteration": round(cand / max(iters, 1) / 10000.0, 3)}
    print(f"[probe] batch of {NP}: {it} iterations {best*1e3:.2f} ms, builds {builds} ({builds/NP:.1f} per pair), "
          f"{cand / max(iters, 1) / 10000.0:.2f} candidates per row and iteration", file=sys.stderr)
out["batch"] = batch
out["align_per_s"] = NP / (batch["2000"]["ms"] * 1e-3)
if os.environ.get("CVO_PHASE_TICKS"):
    gpu.align_batch(src, tgt, inits, max_iterations=1000)
    print("[probe] phase stamps, 16-pair sub-batch at iteration 1000:", file=sys.stderr)
    a, c = gpu.debug_time_kernels(10)
    out["batch_alone_us"] = {"k_assoc": a * 1e3, "k_coeff": c * 1e3}
for h in both:
    h.free()
gpu.close()

single = {}
for name, builder, kw in (("geo10k", cases.config2, dict(n=10000)), ("config3", cases.config3, dict(n=10000)),
                          ("config4", cases.config4, dict(n=10000)), ("config1", cases.config1, {})):
    Pc, a, b, init = builder(**kw)
    g = CvoGPU(params=Pc)
    da, db = g.upload(a), g.upload(b)
    mi = 3000 if name == "config1" else 0
    g.align(da, db, init, max_iterations=50)
    best = None
    for _ in range(3):
        r = g.align(da, db, init, max_iterations=mi)
        if best is None or r.seconds < best.seconds:
            best = r
    builds, iters, cand = g.debug_list_builds()
    single[name] = {"iterations": best.iterations, "us_per_iter": round(best.seconds * 1e6 / max(best.iterations, 1), 3),
                    "align_ms": round(best.seconds * 1e3, 3), "list_builds": builds}
    print(f"[probe] single pair {name}: {best.iterations} iterations, {best.seconds*1e6/max(best.iterations,1):.2f} us/iteration, "
          f"{builds} list builds", file=sys.stderr)
    if os.environ.get("CVO_PHASE_TICKS") and name in ("geo10k", "config4"):
        g.align(da, db, init, max_iterations=max(best.iterations // 2, 1))
        print(f"[probe] phase stamps, single pair {name}:", file=sys.stderr)
        g.debug_time_kernels(10)
    g.close()
out["single_pair"] = single
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
