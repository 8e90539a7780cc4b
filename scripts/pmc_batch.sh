# PMC passes over one batched align (64 x 10k x 10k, 2000 iterations): instruction mix, SQ cycle breakdown, L2 / L1.
# usage (GPU box): [PROBE_ITERS=64] [PROBE_PAIRS=64] bash scripts/pmc_batch.sh OUTDIR   -> OUTDIR/pmc_batch.json
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/${1:-gpurun_out/pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one_batch.py <<PY
import os, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import cases
from unified_cvo_amd import CvoGPU
P = cases.load_params("geometric_gpu")
NP = int(os.environ.get("PROBE_PAIRS", "64"))
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
r = gpu.align_batch(both[:NP], both[NP:], [a[3] for a in pairs], max_iterations=int(os.environ.get("PROBE_ITERS", "0")))
print(r[0].iterations, r[0].seconds)
PY
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/pb$i -o p -- python /tmp/one_batch.py > /tmp/pb$i.out 2> /tmp/pb$i.log || tail -5 /tmp/pb$i.log
done
python $R/scripts/summarize_pmc.py $O/pmc_batch.json /tmp/pb1 /tmp/pb2 /tmp/pb3 /tmp/pb4 /tmp/pb5 /tmp/pb6 > $O/pmc_batch.txt
