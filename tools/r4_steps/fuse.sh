timeout 600 python -m pytest tests/test_gpu_backbone_ops.py -q -x -k "fused_expand" > $OUT/fuse_tests.log 2>&1; rc=$?; echo "fuse op tests rc=$rc" | tee -a $OUT/rc.txt; tail -12 $OUT/fuse_tests.log
if [ $rc -eq 0 ]; then
  timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_determinism.py -q -x -k "eval or detections or batch or determin" > $OUT/fuse_model_tests.log 2>&1; echo "fuse model tests rc=$?" | tee -a $OUT/rc.txt; tail -4 $OUT/fuse_model_tests.log
  for v in 0 1; do EFFDET_FUSE_EXPAND_DW=$v timeout 300 python tools/infer_bench.py --reps 20 > $OUT/infer_fuse$v.log 2>&1; tail -1 $OUT/infer_fuse$v.log; done
  for v in 0 1; do EFFDET_FUSE_EXPAND_DW=$v timeout 300 python tools/infer_bench.py --network efficientdet-d4 --batch 8 --size 1024 --reps 10 > $OUT/infer_d4_fuse$v.log 2>&1; tail -1 $OUT/infer_d4_fuse$v.log; done
fi
