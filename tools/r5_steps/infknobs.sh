run() { for net in d0 d4; do
  if [ $net = d0 ]; then A="--dtype f32 --reps 10"; else A="--network efficientdet-d4 --batch 8 --size 1024 --dtype f32 --reps 10"; fi
  r=$(env "$@" python tools/infer_bench.py $A 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/.*forward \([0-9.]*\) ms.img.*one-graph detect \([0-9.]*\).*/fwd \1 graph-detect \2/')
  printf "%-34s %s %s\n" "$*" $net "$r"; done; }
run A=base
run EFFDET_IGEMM_NARROW_K=8
run EFFDET_IGEMM_NARROW_K=32
run EFFDET_IGEMM_NARROW=512
run EFFDET_IGEMM_DEEP=0
run EFFDET_FUSE_EXPAND_DW=0
run EFFDET_FUSE_CIN=16,24,32,40
run EFFDET_GATE_IN_WEIGHTS=0
run EFFDET_CONV_PW=0
run A=base
