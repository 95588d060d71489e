B1="python bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline"
for v in 0 1 0 1 0 1; do
  EFFDET_GATE_IN_WEIGHTS_TRAIN=$v timeout 300 $B1 > $OUT/giw.log 2> $OUT/giw.err
  python - <<PY
import json
l=[x for x in open('$OUT/giw.log') if x.startswith('{')]
d=json.loads(l[-1]); print('giw_train=$v %8.2f img/s %7.3f ms' % (d['value'], d['ms_per_step']))
PY
done
EFFDET_GATE_IN_WEIGHTS_TRAIN=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_determinism.py tests/test_gpu_pipeline.py -q -x -k "train_losses or fwd_exact or determinism or bitwise or replay_is or gate_in" > $OUT/giw_tests.log 2>&1; echo "giw tests rc=$?"; tail -3 $OUT/giw_tests.log
