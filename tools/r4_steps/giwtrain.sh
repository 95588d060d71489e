timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_determinism.py tests/test_gpu_conv.py -q -x -k "train or fused or determin or bitwise or batched_unpack or per_image" > $OUT/giw_tests.log 2>&1; echo "giw tests rc=$?" | tee -a $OUT/rc.txt; tail -4 $OUT/giw_tests.log
bash tools/ab_env.sh EFFDET_GATE_IN_WEIGHTS_TRAIN 0 1 0 1 2>&1 | tee $OUT/ab_giw.txt
