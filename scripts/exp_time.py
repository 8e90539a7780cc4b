"""A/B timing of library builds on the GPU box: the headline batch (64 x 10k x 10k, 2000 iterations) and one 10k pair,
for the product library and every experiment build named on the command line (unified_cvo_amd/build.py:
build_variant -> lib/libcvo_hip_<name>.so), interleaved so that box-to-box and minute-to-minute drift cancels; prints
min / median ms and whether the poses are bit-identical to the product library's.
usage: python scripts/exp_time.py [name ...]      (EXP_REPS=5, EXP_ROUNDS=3, EXP_ITERS=0 -> run to completion)"""
import hashlib
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from unified_cvo_amd import CvoGPU, build as B  # noqa: E402

names = ["product"] + sys.argv[1:]
REPS = int(os.environ.get("EXP_REPS", "5"))
ROUNDS = int(os.environ.get("EXP_ROUNDS", "3"))
ITERS = int(os.environ.get("EXP_ITERS", "0"))
NP = int(os.environ.get("EXP_PAIRS", "64"))
P = cases.load_params("geometric_gpu")
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
ctx = {}
for nm in names:
    lib = None if nm == "product" else os.path.join(B.LIBDIR, f"libcvo_hip_{nm}.so")
    gpu = CvoGPU(params=P, library=lib)
    both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
    ctx[nm] = (gpu, both[:NP], both[NP:])
    gpu.align_batch(both[:NP], both[NP:], [a[3] for a in pairs], max_iterations=64)
inits = [a[3] for a in pairs]
tb = {nm: [] for nm in names}
ts = {nm: [] for nm in names}
sig = {}
for rnd in range(ROUNDS):
    for nm in names:
        gpu, s, t = ctx[nm]
        for _ in range(REPS):
            t0 = time.perf_counter()
            r = gpu.align_batch(s, t, inits, max_iterations=ITERS)
            tb[nm].append(time.perf_counter() - t0)
        h = hashlib.sha256()
        for x in r:
            h.update(np.ascontiguousarray(x.transform).tobytes())
            h.update(str(x.iterations).encode())
        sig[nm] = h.hexdigest()[:12]
        for _ in range(2):
            r1 = gpu.align(s[0], t[0], inits[0], max_iterations=ITERS)
            ts[nm].append(r1.seconds / max(r1.iterations, 1))
for nm in names:
    b = np.array(tb[nm]) * 1e3
    print(f"{nm:14s} batch {b.min():7.2f} min {np.median(b):7.2f} median ms ({NP / b.min() * 1e3:7.1f} align/s)   single pair "
          f"{min(ts[nm]) * 1e6:6.2f} us/iteration   poses {sig[nm]} {'(identical)' if sig[nm] == sig['product'] else 'DIFFERENT'}", flush=True)
