timeout 600 python -m pytest tests/test_gpu_post_loss.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_model.py -q -k "nms or detect or detections" > $OUT/nms_tests.log 2>&1; echo "nms tests rc=$?" | tee -a $OUT/rc.txt; tail -3 $OUT/nms_tests.log
timeout 300 python tools/infer_bench.py --network efficientdet-d0 --batch 32 --size 512 --reps 20 > $OUT/infer_d0.log 2>&1; tail -1 $OUT/infer_d0.log
for r in 1024 4096; do EFFDET_NMS_ROUND=$r timeout 300 python tools/infer_bench.py --reps 10 --no-graph > $OUT/infer_d0_r$r.log 2>&1; tail -1 $OUT/infer_d0_r$r.log; done
timeout 300 python tools/infer_bench.py --network efficientdet-d4 --batch 8 --size 1024 --reps 10 > $OUT/infer_d4.log 2>&1; tail -1 $OUT/infer_d4.log
EFFDET_NMS_ROUND=2048 timeout 300 python tools/infer_bench.py --network efficientdet-d4 --batch 8 --size 1024 --reps 10 --no-graph > $OUT/infer_d4_r2048.log 2>&1; tail -1 $OUT/infer_d4_r2048.log
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt_infer -o kt -- python $GRAFT_REPO_ROOT/tools/infer_bench.py --no-graph --reps 5 > $GRAFT_REPO_ROOT/$OUT/kt_infer.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ['OUT'] + '/kt_infer/**/*kernel_stats.csv', recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0]))):
        if any(k in r['Name'] for k in ('nms_', 'rs_', 'decode', 'gather_dets')):
            print('%-60s calls %5s total %9.1f us avg %8.1f' % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3))
PY
