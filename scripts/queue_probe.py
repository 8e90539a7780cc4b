"""Throughput of the batch queue on ONE GPU: (a) 512 queued 10k x 10k geometric pairs (the 8-GPU headline's whole work
list) through 64 / 96 / 128 in-flight slots, next to eight back-to-back cvo_align_batch calls of 64; (b) a mixed queue -
warm starts of the semantic configuration (~300 iterations) among cold starts (thousands) - against its
iteration-weighted ideal.  Every pose is compared with the fixed-batch / solo result.  usage: queue_probe.py [out.json]"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from unified_cvo_amd import CvoGPU

out = {}
NQ = int(os.environ.get("QUEUE_PAIRS", "512"))
P = cases.load_params("geometric_gpu")
pairs = [cases.config2(n=10000, pair_id=p) for p in range(64)]
gpu = CvoGPU(params=P, library=os.environ.get("CVO_LIB") or None)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
src, tgt, inits = both[:64], both[64:], [a[3] for a in pairs]
ref = gpu.align_batch(src, tgt, inits)
t0 = time.perf_counter()
for _ in range(NQ // 64):
    gpu.align_batch(src, tgt, inits)
t_fixed = time.perf_counter() - t0
out["fixed_batches_of_64"] = {"pairs": NQ, "s": round(t_fixed, 4), "align_per_s": round(NQ / t_fixed, 1)}
print(f"[queue] {NQ // 64} x cvo_align_batch(64): {t_fixed*1e3:.1f} ms, {NQ / t_fixed:.1f} align/s", flush=True)
for slots in (64, 96, 128, 192):
    best = None
    for _ in range(2):
        q = gpu.open_queue(slots, 10000, 10000)
        t0 = time.perf_counter()
        for k in range(NQ):
            q.submit(src[k % 64], tgt[k % 64], inits[k % 64])
        res = []
        while q.pending():
            res.extend(q.poll(wait=2))
        dt = time.perf_counter() - t0
        st = q.stats()
        q.close()
        if best is None or dt < best[0]:
            best = (dt, st)
    same = all(np.array_equal(r.transform, ref[k % 64].transform) and r.iterations == ref[k % 64].iterations for k, r in enumerate(res))
    out[f"queue_{slots}"] = {"pairs": NQ, "slots": slots, "s": round(best[0], 4), "align_per_s": round(NQ / best[0], 1),
                             "bit_identical_to_fixed_batch": bool(same), **best[1]}
    print(f"[queue] {NQ} pairs through {slots} slots: {best[0]*1e3:.1f} ms, {NQ / best[0]:.1f} align/s, poses "
          f"{'identical' if same else 'DIFFERENT'}, {best[1]}", flush=True)
for h in both:
    h.free()
gpu.close()

# ---- mixed queue: the geometric configuration, three pairs in four stop after 300 iterations (warm-started tracking
# frames), the fourth runs its 2000 (a cold start)
g = CvoGPU(params=P, library=os.environ.get("CVO_LIB") or None)
pm = cases.config2(n=10000, pair_id=3)
da, db, cold = g.upload(pm[1]), g.upload(pm[2]), pm[3]
warm = cold
lim = 2000
solo_w = g.align(da, db, warm, max_iterations=300)
solo_c = g.align(da, db, cold, max_iterations=lim)
kinds = [(k % 4 == 0) for k in range(256)]   # every fourth pair is a long one
total_iters = sum(solo_c.iterations if c else solo_w.iterations for c in kinds)
# the ideal: the same number of pair-iterations at the rate of a UNIFORM fixed batch of the long problem (64 in flight)
gb = g.align_batch([da] * 64, [db] * 64, [cold] * 64, max_iterations=lim)
t0 = time.perf_counter()
gb = g.align_batch([da] * 64, [db] * 64, [cold] * 64, max_iterations=lim)
t_uniform = time.perf_counter() - t0
rate_uniform = 64 * solo_c.iterations / t_uniform
for slots in (64, 128):
    q = g.open_queue(slots, 10000, 10000, max_iterations=lim)
    t0 = time.perf_counter()
    for c in kinds:
        q.submit(da, db, cold, 0 if c else 300)
    res = []
    while q.pending():
        res.extend(q.poll(wait=2))
    dt = time.perf_counter() - t0
    st = q.stats()
    q.close()
    same = all(np.array_equal(r.transform, (solo_c if c else solo_w).transform) and r.iterations == (solo_c if c else solo_w).iterations
               for r, c in zip(res, kinds))
    eff = (total_iters / dt) / rate_uniform
    out[f"mixed_{slots}"] = {"pairs": len(kinds), "short_iterations": solo_w.iterations, "long_iterations": solo_c.iterations,
                             "s": round(dt, 4), "pair_iterations_per_s": round(total_iters / dt), "uniform_batch_pair_iterations_per_s": round(rate_uniform),
                             "fraction_of_iteration_weighted_ideal": round(eff, 3), "bit_identical_to_solo": bool(same), **st}
    print(f"[queue] mixed ({solo_w.iterations} / {solo_c.iterations} iterations, 1 cold in 4) through {slots} slots: {dt*1e3:.1f} ms = "
          f"{eff:.3f} of the iteration-weighted ideal ({rate_uniform/1e6:.2f} M pair-iterations/s in a uniform batch of 64), poses "
          f"{'identical' if same else 'DIFFERENT'}, {st}", flush=True)
g.close()
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
