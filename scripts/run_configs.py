"""BASELINE.json configs 1-4 end to end on one MI355X next to the CPU oracle: iterations, timing, pose parity,
inner_product / function_angle parity.  Writes profiles/<round>/configs.json (argv[1], default gpurun_out/configs.json)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from unified_cvo_amd import CvoGPU  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "configs.json")
po.set_num_threads(int(os.environ.get("ORACLE_THREADS", "16")))
rows = []
for name, builder, kw, oracle_cap in (
        ("config1 demo 523x1080 (cvo_outdoor_params, geometric-only, fixed 1000 iterations)", cases.config1, {}, 1000),
        ("config1 demo 523x1080 (to its own eps_2 stop)", cases.config1, {}, 0),
        ("config2 5k x 5k xyz (cvo_geometric_params_gpu)", cases.config2, dict(n=5000), 0),
        ("config2-shape 10k x 10k xyz", cases.config2, dict(n=10000), 0),
        ("config3 10k x 10k + 5-ch colour (cvo_intensity_params_gpu, HEAD side + overrides)", cases.config3, dict(n=10000), 0),
        ("config4 10k x 10k + 19-class semantics (cvo_semantic_params_img_gpu0, warm start)", cases.config4, dict(n=10000), 0),
        ("clustered street scene 10k x 10k xyz (NOT a BASELINE config: synth.scene_pair, density varies > 100x, rows on the K cap)",
         cases.scene, dict(n=10000), 0)):
    P, src, tgt, init = builder(**kw)
    gpu = CvoGPU(params=P)
    ds, dt = gpu.upload(src), gpu.upload(tgt)
    opt = dict(max_iterations=oracle_cap) if oracle_cap else {}
    gpu.align(ds, dt, init, max_iterations=20)  # warm-up (graph capture, code load)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        g = gpu.align(ds, dt, init, **opt)
        times.append(time.perf_counter() - t0)
    op = po.params_from(P)
    ox, oy = po.Cloud.from_pointcloud(src), po.Cloud.from_pointcloud(tgt)
    o = po.align(op, ox, oy, init, max_iterations=oracle_cap)
    final_state = np.linalg.inv(g.transform.astype(np.float64)).astype(np.float32)
    ip_g, ip_o = gpu.inner_product_gpu(ds, dt, final_state, P.ell_init), po.inner_product(op, ox, oy, final_state, P.ell_init)
    fa_g, fa_o = gpu.function_angle(ds, dt, final_state, P.ell_init, False), po.function_angle(op, ox, oy, final_state, P.ell_init, False)
    row = dict(config=name, N=src.num_points(), M=tgt.num_points(), gpu_iterations=g.iterations, oracle_iterations=o["iterations"],
               gpu_ret=g.ret, oracle_ret=o["ret"], gpu_align_ms_median=float(np.median(times) * 1e3),
               gpu_loop_ms=g.seconds * 1e3, gpu_us_per_iter=g.seconds * 1e6 / max(g.iterations, 1),
               oracle_align_ms=o["seconds"] * 1e3, oracle_us_per_iter=o["seconds"] * 1e6 / max(o["iterations"], 1),
               oracle_threads=po.num_threads(), pose_max_abs_diff=cases.max_abs_diff(g.transform, o["transform"]),
               inner_product_gpu=ip_g, inner_product_oracle=ip_o, function_angle_gpu=fa_g, function_angle_oracle=fa_o)
    rows.append(row)
    print(json.dumps(row), flush=True)
    gpu.close()
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(rows, open(out_path, "w"), indent=1)
