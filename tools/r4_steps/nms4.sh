timeout 600 python -m pytest tests/test_gpu_post_loss.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_model.py -q -k "nms or detect or detections" > $OUT/nms_tests.log 2>&1; echo "nms tests rc=$?" | tee -a $OUT/rc.txt; tail -3 $OUT/nms_tests.log
for v in 1 2; do EFFDET_NMS_CROSS=$v timeout 300 python tools/infer_bench.py --reps 20 > $OUT/infer_cross$v.log 2>&1; echo "cross=$v"; tail -1 $OUT/infer_cross$v.log; done
for v in 1 2; do EFFDET_NMS_CROSS=$v timeout 300 python tools/infer_bench.py --network efficientdet-d4 --batch 8 --size 1024 --reps 10 > $OUT/infer_d4_cross$v.log 2>&1; echo "cross=$v"; tail -1 $OUT/infer_d4_cross$v.log; done
timeout 200 python tools/graph_nms_probe.py > $OUT/graph_nms_probe.txt 2>&1; cat $OUT/graph_nms_probe.txt
