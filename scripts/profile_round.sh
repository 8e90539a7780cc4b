set -x
R=/root/repo
O=$R/gpurun_out/${ROUND:-r2}_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --gpus 1 --steps 2 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-single-pair > $O/bench_under_rocprof.json 2> $O/rocprof.log
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc$i -o p -- python $R/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-single-pair > /tmp/pmc$i.json 2> /tmp/pmc$i.log
done
python $R/scripts/summarize_pmc.py $O/pmc_summary_raw.json /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 /tmp/pmc4
cd $R && timeout 900 python scripts/run_configs.py $O/configs.json > $O/configs.log 2>&1
tail -3 $O/bench_n1.log
