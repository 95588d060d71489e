timeout 600 python tools/graph_nms_probe.py > $OUT/graph_nms_probe.txt 2>&1; cat $OUT/graph_nms_probe.txt
timeout 600 python -m pytest tests/test_gpu_post_loss.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -q -k "nms or detect or detections" > $OUT/nms_tests.log 2>&1; echo "nms tests rc=$?" | tee -a $OUT/rc.txt; tail -3 $OUT/nms_tests.log
timeout 300 python tools/infer_bench.py --network efficientdet-d0 --batch 32 --size 512 --reps 10 --no-graph > $OUT/infer_d0.log 2>&1; tail -1 $OUT/infer_d0.log
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt_infer -o kt -- python $GRAFT_REPO_ROOT/tools/infer_bench.py --no-graph --reps 5 > $GRAFT_REPO_ROOT/$OUT/kt_infer.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ['OUT'] + '/kt_infer/**/*kernel_stats.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:28]:
        print('%-70s calls %5s total %9.1f us avg %8.1f' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3))
PY
