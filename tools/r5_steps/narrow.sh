B1="python bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline"
for v in 64 128 64 128; do
  EFFDET_IGEMM_NARROW=$v timeout 300 $B1 > $OUT/n.log 2> $OUT/n.err
  python - <<PY
import json
l=[x for x in open('$OUT/n.log') if x.startswith('{')]
d=json.loads(l[-1]); print('narrow=$v train %8.2f img/s %7.3f ms' % (d['value'], d['ms_per_step']))
PY
done
for v in 64 128 64 128; do EFFDET_IGEMM_NARROW=$v python tools/infer_bench.py --dtype f32 --reps 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120; done
for v in 64 128; do EFFDET_IGEMM_NARROW=$v python tools/infer_bench.py --dtype f32_bf16x3 --reps 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120; EFFDET_IGEMM_NARROW=$v python tools/infer_bench.py --network efficientdet-d4 --batch 8 --size 1024 --dtype f32_bf16x3 --reps 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120; done
