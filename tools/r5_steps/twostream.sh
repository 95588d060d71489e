B1="python bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline"
for v in 0 1 0 1; do
  EFFDET_HEAD_TWO_STREAMS=$v timeout 300 $B1 > $OUT/ts_$v.log 2> $OUT/ts_$v.err; python - <<PY
import json
l=[x for x in open('$OUT/ts_$v.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('two_streams=$v', d['value'], d['ms_per_step'], d['config']['launch'], d['config']['graph_self_check']['replay_vs_eager_rel_to_update'] if d['config'].get('graph_self_check') else None)
else:
    print('two_streams=$v no line'); print(open('$OUT/ts_$v.err').read()[-1500:])
PY
done
EFFDET_HEAD_TWO_STREAMS=1 timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_pipeline.py tests/test_gpu_determinism.py -q -x -k "train_losses or replay_is_the_eager or determinism or bitwise or fwd_exact" > $OUT/ts_tests.log 2>&1; echo "ts tests rc=$?"; tail -3 $OUT/ts_tests.log
