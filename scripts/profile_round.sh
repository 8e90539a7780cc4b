# The round's profile set, run on the GPU box (gpurun): bench line, rocprofv3 kernel trace of the same command, PMC
# passes (one counter group per pass, no trace domains), every BASELINE config next to the oracle.
#   ROUND=r4 bash scripts/profile_round.sh      -> gpurun_out/r4_prof/ ; then scripts/install_profiles.py r4 (here)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${ROUND:-r5}_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# PMC: the headline batch itself (one cvo_align_batch of 64 x 10k x 10k, 2000 iterations; launches of 16 pairs), not
# the whole bench harness - counter collection serialises every dispatch
cat > /tmp/one_batch.py <<PY
import os, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import cases
from unified_cvo_amd import CvoGPU
P = cases.load_params("geometric_gpu")
pairs = [cases.config2(n=10000, pair_id=p) for p in range(64)]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
r = gpu.align_batch(both[:64], both[64:], [a[3] for a in pairs])
print(r[0].iterations, r[0].seconds)
PY
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc$i -o p -- python /tmp/one_batch.py > /tmp/pmc$i.json 2> /tmp/pmc$i.log || tail -3 /tmp/pmc$i.log
done
python $R/scripts/summarize_pmc.py $O/pmc_summary_raw.json /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 /tmp/pmc4 /tmp/pmc5 /tmp/pmc6 > $O/pmc_summary.txt
# the kernel trace of the bench command itself (timed steps only: production kernels); its per-kernel averages and the PMC
# traffic of THIS build go into profiles/kernel_traffic.json before the bench run, which reports them (roofline.kernel is
# chosen by this trace)
CVO_KERNEL_CLOCK=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-single-pair --no-pipeline --no-extra-legs > $O/bench_under_rocprof.json 2> $O/rocprof.log
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/scripts/install_profiles.py ${ROUND:-r5} --traffic
timeout 900 python $R/bench.py --gpus 1 --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.log
cd $R && timeout 600 python scripts/run_configs.py $O/configs.json > $O/configs.log 2>&1
# the same bench command under torchrun with ONE rank (what the driver does for N > 1, at N = 1): RCCL cost per step
cd $R && timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-single-pair --no-pipeline --no-extra-legs > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.log
# the C++ host of the multi-GPU mode on the headline workload, with and without the library's hardware-queue hint
timeout 600 python $R/scripts/cpp_host_bench.py > $O/cpp_host_bench.txt 2> $O/cpp_host_bench.log
# what an upload costs (ordering on the device vs on the host) and the PCIe-inclusive pipeline with 2 upload threads
timeout 600 python $R/scripts/upload_probe.py > $O/upload_probe.txt 2>&1
# where latency-bound turns into throughput-bound: the batch at 1 ... 128 pairs
timeout 600 python $R/scripts/scale_probe.py > $O/scale_probe.txt 2>&1
# the early phase: first 16 / 64 / 256 iterations of the batch, single pairs
timeout 600 python $R/scripts/perf_probe.py $O/perf_probe.json > $O/perf_probe.txt 2>&1
# the clustered street scene (not a BASELINE config): us per iteration at 3000 / 10k points, and its kernels
timeout 300 python $R/scripts/scene_probe.py 2>&1 | grep -v "chunk [0-9]" > $O/scene_probe.txt
GRAFT_REPO_ROOT=$R timeout 300 bash $R/scripts/scene_profile.sh > /dev/null 2>&1; cp $R/gpurun_out/scene_prof/kernel_stats.csv $O/scene_kernel_stats.csv
# round 5: the batch queue (512 pairs through 64 ... 192 slots, mixed queue), overlap queries, row-class limit, one rank's
# host budget on two CPUs
timeout 600 python $R/scripts/queue_probe.py $O/queue_probe.json > $O/queue_probe.txt 2>&1
timeout 300 python $R/scripts/overlap_probe.py > $O/overlap_probe.txt 2>&1
timeout 600 python $R/scripts/rowmax_probe.py > $O/rowmax_probe.txt 2>&1
timeout 900 bash $R/scripts/rank_rehearsal.sh gpurun_out/${ROUND:-r5}_prof/rehearsal > $O/rank_rehearsal.txt 2>&1
# run-to-run reproducibility under load
timeout 900 python $R/scripts/stress_repeat.py ${STRESS_BATCH:-100} ${STRESS_SINGLE:-40} > $O/stress.txt 2>&1
tail -3 $O/bench_n1.log
