timeout 600 python -m pytest tests/test_gpu_model.py -q -s -k "complete_lists" > $OUT/d4dets.log 2>&1; echo "d4dets rc=$?" | tee -a $OUT/rc.txt; grep -v "^$" $OUT/d4dets.log | tail -12
