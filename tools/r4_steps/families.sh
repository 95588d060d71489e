# every model family's goldens on the GPU (forward fixtures + train-mode losses / gradients), all three modes
rm -f gpurun_out/parity_errors.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -s -k "(train_losses_and_grads or eval_forward) and (d1_128_eval or d2_ or d3_ or d5_ or d6_)" > $OUT/families.log 2>&1; echo "families rc=$?" | tee -a $OUT/rc.txt
grep -E "^FAILED|passed|failed|^E  +(Assertion|assert)" $OUT/families.log | cut -c1-300 | tail -30
cp gpurun_out/parity_errors.txt $OUT/ 2>/dev/null
