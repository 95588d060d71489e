timeout 600 python -m pytest tests/test_gpu_model.py -q -s -k "dense" > $OUT/dense.log 2>&1; echo "dense rc=$?" | tee -a $OUT/rc.txt; grep "dense detections\|passed\|failed" $OUT/dense.log | tail -8
