timeout 600 python -m pytest tests/test_gpu_post_loss.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -q -k "nms or detect or detections" > $OUT/nms_tests.log 2>&1; echo "nms tests rc=$?" | tee -a $OUT/rc.txt; tail -6 $OUT/nms_tests.log
for v in 1 0; do
  EFFDET_NMS_V1=$v timeout 300 python tools/infer_bench.py --network efficientdet-d0 --batch 32 --size 512 --reps 10 > $OUT/infer_v1_$v.log 2>&1; tail -3 $OUT/infer_v1_$v.log
done
EFFDET_NMS_ROUND=4096 timeout 300 python tools/infer_bench.py --network efficientdet-d0 --batch 32 --size 512 --reps 10 > $OUT/infer_r4096.log 2>&1; tail -2 $OUT/infer_r4096.log
EFFDET_NMS_ROUND=1024 timeout 300 python tools/infer_bench.py --network efficientdet-d0 --batch 32 --size 512 --reps 10 > $OUT/infer_r1024.log 2>&1; tail -2 $OUT/infer_r1024.log
for v in 1 0; do
  EFFDET_NMS_V1=$v timeout 300 python tools/infer_bench.py --network efficientdet-d4 --batch 8 --size 1024 --reps 5 > $OUT/infer_d4_v1_$v.log 2>&1; tail -2 $OUT/infer_d4_v1_$v.log
done
