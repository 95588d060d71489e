timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "largest_family" > $OUT/largest.log 2>&1; echo "largest rc=$?" | tee -a $OUT/rc.txt
grep -E "^D6|^FAILED|passed|failed|^E  +(Assertion|assert)|Error" $OUT/largest.log | cut -c1-300 | tail -12
