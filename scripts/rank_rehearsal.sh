# Rehearsal of ONE of eight ranks' host budget on a 16-CPU box (BASELINE.json configs[4]: 8 ranks x 64 pairs): the bench
# and the C++ host of the multi-GPU mode pinned to two CPUs, next to the unpinned run.  What this cannot show is the
# fabric (RCCL over xGMI between eight devices); everything a rank does on its own GPU and its two host threads it can.
#   bash scripts/rank_rehearsal.sh OUTDIR        (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/${1:-gpurun_out/rehearsal}
mkdir -p $O
cd $R
for cpus in "0-15" "0-1"; do
  tag=$(echo $cpus | tr - _)
  taskset -c $cpus timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-single-pair --no-extra-legs \
      > $O/bench_cpus_$tag.json 2> $O/bench_cpus_$tag.log
  python - <<PY
import json
d = json.loads(open("$O/bench_cpus_$tag.json").read().strip().splitlines()[-1])
print("bench.py pinned to CPUs $cpus: %.1f align/s, %.2f ms/step, host threads per rank %s, PCIe-inclusive pipeline %s align/s" % (
    d["value"], d["ms_per_step"], d["config"]["host_threads_per_rank"], round(d["pcie_inclusive"].get("value", 0), 1)))
PY
done
for cpus in "0-15" "0-1"; do
  echo "C++ host (cvo_align_sharded --bench, one device, RCCL communicator alive) pinned to CPUs $cpus:"
  taskset -c $cpus timeout 600 python scripts/cpp_host_bench.py 2>&1 | grep -v "^\[cvo\]"
done
