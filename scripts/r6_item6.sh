# Round 6, VERDICT item 6 + 9: the queue at 64 ... 192 slots with the product library and the 8-byte-ELL experiment build,
# admission shares of the mixed queue, the A/B at 64 pairs, the early phase, and the one-device rehearsal of bench.py's N>1 branch.
# (the experiment build first, here: python -c "from unified_cvo_amd import build; build.build_variant(\"ell8\", [\"-DCVO_ELL8\"])")
mkdir -p gpurun_out/r6
O=gpurun_out/r6/queue_probe.txt
echo "== product library" > $O
python scripts/queue_probe.py gpurun_out/r6/queue_product.json 2>&1 | grep queue >> $O
echo "== -DCVO_ELL8 (8-byte ELL entries)" >> $O
CVO_LIB=unified_cvo_amd/lib/libcvo_hip_ell8.so python scripts/queue_probe.py 2>&1 | grep queue >> $O
for a in 1 2 8; do echo "== QUEUE_ADMIT=$a (a settled sub-batch takes newcomers when 1/$a of its slots are free)" >> $O; CVO_QUEUE_ADMIT=$a QUEUE_PAIRS=64 python scripts/queue_probe.py 2>&1 | grep -i "mixed" >> $O; done
echo "== exp_time: product vs ell8 at 64 pairs" >> $O
EXP_ROUNDS=2 EXP_REPS=4 python scripts/exp_time.py ell8 2>&1 | tail -2 >> $O
echo "== early phase" >> $O
python scripts/early_sweep.py 2>&1 | tail -1 >> $O
echo "== rehearsal: torchrun, 2 ranks on device 0, gloo" > gpurun_out/r6/rehearsal.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --pairs-per-gpu 16 --rehearse-one-device --no-cpu-baseline --no-single-pair --no-pipeline --no-extra-legs > gpurun_out/r6/rehearsal.json 2>> gpurun_out/r6/rehearsal.txt
tail -3 gpurun_out/r6/rehearsal.txt
cat $O
