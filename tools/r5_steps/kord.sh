B1="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference"
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in 0 64 0 64 256; do
  EFFDET_F32_KORD=$v timeout 300 $B1 --no-roofline > $R/$OUT/kord_$v.log 2> $R/$OUT/kord_$v.err
  python - <<PY
import json
l=[x for x in open('$R/$OUT/kord_$v.log') if x.startswith('{')]
d=json.loads(l[-1]); print('f32_kord=$v', d['value'], d['ms_per_step'])
PY
done
for v in 0 64; do
  EFFDET_F32_KORD=$v timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_kord_$v -o pmc -- $B1 --no-roofline --no-graph --steps 2 --warmup 1 > $R/$OUT/pmc_kord_$v.log 2>&1
  python $R/tools/pmc_fetch.py $R/$OUT/pmc_kord_$v "conv_igemm_kernel<float, 128, 2, 8, 0, 2" "f32_kord=$v" 
  rm -rf $R/$OUT/pmc_kord_$v
done
cd $R
