B1="python bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline"
run() {
  env "$@" timeout 300 $B1 > $OUT/knob.log 2> $OUT/knob.err
  python - "$*" <<PY
import json, sys
l=[x for x in open('$OUT/knob.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('%-48s %8.2f img/s %7.3f ms  check %s' % (sys.argv[1], d['value'], d['ms_per_step'], (d['config'].get('graph_self_check') or {}).get('replay_vs_eager_rel_to_update')))
else:
    print(sys.argv[1], 'FAILED', open('$OUT/knob.err').read()[-300:])
PY
}
run A=base
run EFFDET_GATE_IN_WEIGHTS_TRAIN=1
run EFFDET_SPLIT_PERS=1
run EFFDET_SPLIT_PERS=3
run EFFDET_SPLIT_PERS=0
run A=base
run EFFDET_SPLIT_KORD=0
run EFFDET_SPLIT_KORD=2
run EFFDET_IGEMM_DEEP=1
run EFFDET_IGEMM_NARROW=0
run EFFDET_EXPAND_Z_ONLY=0
run EFFDET_CONV_PW=0
run EFFDET_BIFPN_WGRAD_GROUP=0
run A=base
