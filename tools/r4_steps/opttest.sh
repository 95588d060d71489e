timeout 300 python -m pytest tests/test_gpu_post_loss.py -q -k "clip_adamw" > $OUT/opttest.log 2>&1; echo "opttest rc=$?" | tee -a $OUT/rc.txt; tail -3 $OUT/opttest.log
