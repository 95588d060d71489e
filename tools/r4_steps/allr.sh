for v in 0 1 0 1; do echo "ALLR=$v"; EFFDET_WGRAD_SPLIT_ALLR=$v timeout 200 python tools/kbench_split.py --which wgrad --reps 10 2>/dev/null | grep "split wgrad"; done | tee $OUT/allr.txt
EFFDET_WGRAD_SPLIT_ALLR=1 timeout 300 python -m pytest tests/test_gpu_split.py -q > $OUT/allr_tests.log 2>&1; echo "allr tests rc=$?" | tee -a $OUT/rc.txt; tail -2 $OUT/allr_tests.log
