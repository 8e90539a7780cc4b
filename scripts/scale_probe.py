import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
P = cases.load_params("geometric_gpu")
NP = 128
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
src, tgt = both[:NP], both[NP:]
inits = [a[3] for a in pairs]
for n in (1, 4, 8, 16, 32, 64, 128):
    gpu.align_batch(src[:n], tgt[:n], inits[:n], max_iterations=64)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); gpu.align_batch(src[:n], tgt[:n], inits[:n]); best = min(best, time.perf_counter() - t0)
    g, ppl = gpu.debug_last_geometry()
    print(f"{n:4d} pairs ({g} streams x {ppl}): {best*1e3:8.2f} ms  {best*1e6/2000:7.2f} us/iteration  {n/best:8.1f} align/s", flush=True)
