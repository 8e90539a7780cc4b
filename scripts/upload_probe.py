"""Cost of bringing one step's clouds (2 x 64 clouds of 10k points) onto the device: wall time and host CPU time of
cvo_cloud_upload_many with 1 / 2 / 16 host threads, spatial ordering on the device (default) and on the host
(CVO_ORDER=host), and the PCIe-inclusive pipeline (upload of batch k + 1 while batch k is solved) with 2 threads - what a
rank gets on the 16-CPU box at 8 ranks.  GPU box."""
import os
import sys
import threading
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from unified_cvo_amd import CvoGPU  # noqa: E402

P = cases.load_params("geometric_gpu")
NP = 64
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
hosts = [a[1] for a in pairs] + [a[2] for a in pairs]
inits = [a[3] for a in pairs]
for order in ("device", "host"):
    gpu = CvoGPU(params=P)
    if order == "host":
        gpu.set_option("ORDER", "host")
    cl = gpu.upload_many(hosts, threads=16)
    gpu.align_batch(cl[:NP], cl[NP:], inits, max_iterations=64)
    t0 = time.perf_counter()
    gpu.align_batch(cl[:NP], cl[NP:], inits)
    t_res = time.perf_counter() - t0
    for h in cl:
        h.free()
    for threads in (1, 2, 4, 16):
        best_w, best_c = 1e9, 1e9
        for _ in range(3):
            w0, c0 = time.perf_counter(), time.process_time()
            cl = gpu.upload_many(hosts, threads=threads)
            w, c = time.perf_counter() - w0, time.process_time() - c0
            best_w, best_c = min(best_w, w), min(best_c, c)
            for h in cl:
                h.free()
        print(f"ordering on the {order:6s}: upload_many of {len(hosts)} clouds, {threads:2d} threads: {best_w*1e3:7.2f} ms wall, "
              f"{best_c*1e3:7.2f} ms host CPU ({best_c*1e3/len(hosts):.3f} ms per cloud)", flush=True)
    # pipeline with 2 upload threads
    nxt = {}

    def prefetch():
        nxt["c"] = gpu.upload_many(hosts, threads=2)

    cur = gpu.upload_many(hosts, threads=2)
    n_pipe = 6
    t0 = time.perf_counter()
    for _ in range(n_pipe):
        th = threading.Thread(target=prefetch)
        th.start()
        gpu.align_batch(cur[:NP], cur[NP:], inits)
        th.join()
        for h in cur:
            h.free()
        cur = nxt["c"]
    t = (time.perf_counter() - t0) / n_pipe
    print(f"ordering on the {order:6s}: resident step {t_res*1e3:.2f} ms; PCIe-inclusive pipeline with 2 upload threads {t*1e3:.2f} ms per step "
          f"= {t_res / t:.3f} of the resident rate", flush=True)
    gpu.close()
