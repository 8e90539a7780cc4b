# Round 6: one pair in flight for every shape (us per iteration, how many iterations adopted the speculative update), the
# run-to-run stress and the mixed-batch soak (tagged partials / speculation choose a PATH by timing, never a result).
mkdir -p gpurun_out/r6
O=gpurun_out/r6/validate.txt
: > $O
for c in config2 config3 config4 scene demo; do CVO_VERBOSE=1 python scripts/single_sweep.py $c 2>&1 | grep -E "us/it|speculative" | tail -2 >> $O; done
python scripts/stress_repeat.py 100 40 2>&1 | tail -8 >> $O
python scripts/soak_batch.py 150 2>&1 | tail -4 >> $O
cat $O
