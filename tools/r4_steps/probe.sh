timeout 900 python -m pytest tests/test_gpu_rccl.py -q -x > $OUT/rccl_tests.log 2>&1; echo "rccl tests rc=$?" | tee -a $OUT/rc.txt; tail -4 $OUT/rccl_tests.log | cut -c1-300
