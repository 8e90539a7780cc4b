import gc, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np, cases
from unified_cvo_amd import CvoGPU
P, src, tgt, init = cases.config2(n=4000)
def cycle():
    gpu = CvoGPU(params=P)
    ds, dt = gpu.upload(src), gpu.upload(tgt)
    r = gpu.align_batch([ds]*8, [dt]*8, [init]*8, max_iterations=40)
    ds.free(); dt.free(); gpu.close(); del gpu, ds, dt; gc.collect()
cycle(); torch.cuda.synchronize()
f0 = torch.cuda.mem_get_info()[0]
for k in range(1, 121):
    cycle()
    if k % 20 == 0:
        torch.cuda.synchronize(); print(k, (f0 - torch.cuda.mem_get_info()[0]) >> 20, "MiB")
