"""The headline workload (64 resident 10k x 10k geometric pairs, 2000 iterations) on the C++ host of the multi-GPU mode
(host/cvo_align_sharded --bench: cvo::CvoGPUSharded, RCCL communicator alive) next to cvo_align_batch through the
Python binding, and what the hardware-queue contract is worth: the same C++ run with the
GPU_MAX_HW_QUEUES hint (cvo_process_hint_hw_queues) switched off.  GPU box; usage: python scripts/cpp_host_bench.py > profiles/r4/cpp_host_bench.txt"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from test_cpp_host import _write_xyz_pcd  # noqa: E402

NP = 64
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
tmp = tempfile.mkdtemp()
args = []
for p, (_, src, tgt, _) in enumerate(pairs):
    _write_xyz_pcd(os.path.join(tmp, f"s{p}.pcd"), src.device_arrays()[0])
    _write_xyz_pcd(os.path.join(tmp, f"t{p}.pcd"), tgt.device_arrays()[0])
    args += [os.path.join(tmp, f"s{p}.pcd"), os.path.join(tmp, f"t{p}.pcd")]
shard = os.path.join(ROOT, "host", "cvo_align_sharded")
yaml = os.path.join(cases.CONFIGS, "geometric_gpu.yaml")
base = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "CVO_NO_HW_QUEUE_HINT")}
for label, env in (("no environment variable set (the host calls cvo_process_hint_hw_queues() before its first HIP call)", base),
                   ("CVO_NO_HW_QUEUE_HINT=1 (HIP's default of 4 hardware queues)", dict(base, CVO_NO_HW_QUEUE_HINT="1")),
                   ("GPU_MAX_HW_QUEUES=8 exported by the caller", dict(base, GPU_MAX_HW_QUEUES="8"))):
    r = subprocess.run([shard, "--bench", "7", yaml, "0", "1"] + args, text=True, env=env, capture_output=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("bench ")]
    adv = [l[:120] for l in r.stdout.splitlines() if l.startswith("advice ")]
    print(f"C++ host, {label}:\n    {line[0] if line else r.stderr[-400:]}" + (f"\n    {adv[0]} ..." if adv else ""), flush=True)

from unified_cvo_amd import CvoGPU  # noqa: E402
gpu = CvoGPU(params=pairs[0][0])
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
inits = [np.eye(4, dtype=np.float32)] * NP
gpu.align_batch(both[:NP], both[NP:], inits)
ts = []
for _ in range(7):
    t0 = time.perf_counter()
    gpu.align_batch(both[:NP], both[NP:], inits)
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"cvo_align_batch through the Python binding (no RCCL in the process): min {min(ts):.3f} median {sorted(ts)[3]:.3f} ms "
      f"({NP / min(ts) * 1e3:.1f} align/s)")
