"""Scan cost probe: tiles per k_scan launch and launch time at several points of the optimiser trajectory."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import cases
from unified_cvo_amd import CvoGPU
B = int(os.environ.get("PROBE_PAIRS", "16"))
n = 10000
P = cases.load_params("geometric_gpu")
gpu = CvoGPU(params=P)
pairs = [cases.config2(n=n, pair_id=p) for p in range(B)]
src = [gpu.upload(q[1]) for q in pairs]
tgt = [gpu.upload(q[2]) for q in pairs]
inits = [q[3] for q in pairs]
for it in [int(x) for x in os.environ.get("PROBE_ITERS", "50,500,0").split(",")]:
    kw = dict(max_iterations=it) if it > 0 else {}
    res = gpu.align_batch(src, tgt, inits, **kw)
    t0, rpt, tpt = gpu.debug_scan_stats()
    reps = 10
    ms = gpu.debug_time_scan(reps)
    t1, _, _ = gpu.debug_scan_stats()
    ng, ppl = gpu.debug_last_geometry()
    per_launch = (t1 - t0) / ((reps + 1) * ng)
    print(f"iters={res[0].iterations} groups={ng} ppl={ppl} scan={ms*1e3:.1f}us tiles/launch={per_launch:.0f} "
          f"frac={per_launch*rpt*tpt/(n*n*ppl):.4f} avg_frac_run={t0*rpt*tpt/(float(n)*n*B*max(res[0].iterations,1)):.4f} "
          f"cand={gpu.debug_last_candidates()} builds,iters,cand_evals={gpu.debug_list_builds()} secs={res[0].seconds:.4f}", flush=True)
