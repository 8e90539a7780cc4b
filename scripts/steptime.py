import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, torch.distributed as dist, numpy as np, gc
if os.environ.get("NOGC"): gc.disable()
import cases
from unified_cvo_amd import CvoGPU, sharding
use_dist = "TORCHELASTIC_RUN_ID" in os.environ
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if use_dist:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
P = cases.load_params("geometric_gpu"); gpu = CvoGPU(params=P)
pairs = [cases.config2(n=10000, pair_id=p) for p in range(64)]
src = [gpu.upload(q[1]) for q in pairs]; tgt = [gpu.upload(q[2]) for q in pairs]; inits = [q[3] for q in pairs]
pose_buf = torch.zeros(64, 16, dtype=torch.float32, device=dev)
for w in range(int(os.environ.get('PREWARM','0'))):
    t0=time.perf_counter(); gpu.align_batch(src, tgt, inits, max_iterations=48); print(f'prewarm {1e3*(time.perf_counter()-t0):.1f} ms', flush=True)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = gpu.align_batch(src, tgt, inits); t1 = time.perf_counter()
    gpu.poses_to_device(pose_buf.data_ptr(), 64); t2 = time.perf_counter()
    status = torch.tensor([r.ret for r in res], dtype=torch.int32, device=dev); t3 = time.perf_counter()
    poses, stat = sharding.gather_poses(pose_buf, status, 64, 1, 0); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"dist={use_dist} align_batch {1e3*(t1-t0):.1f} ms (device loop {res[0].seconds*1e3:.1f}) poses_to_device {1e3*(t2-t1):.2f} tensor {1e3*(t3-t2):.2f} gather {1e3*(t4-t3):.2f}", flush=True)
if use_dist: dist.destroy_process_group()
