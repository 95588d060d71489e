timeout 600 python -m pytest tests/test_gpu_model.py -q -s -k "train_losses_and_grads and d4_" > $OUT/d4train.log 2>&1; echo "d4train rc=$?" | tee -a $OUT/rc.txt
grep -E "^FAILED|passed|failed|^E  +(Assertion|assert)|worst grad-norm rel err \(" $OUT/d4train.log | cut -c1-260 | tail -12
