#!/bin/bash
# Round-6 evidence in ONE GPU call: rocprofv3 kernel-trace stats of the train step in the headline mode (and, with ALL=1, the other modes),
# the three PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE) of the headline mode, the step trace, and
# (INFER=1) kernel stats + PMC passes of the D0 / D4 inference legs in the headline's forward arithmetic.
# Every rocprofv3 run is `timeout`-wrapped and asks for csv (the rocpd writer has hung after finalisation).
OUT=${1:-gpurun_out/prof_r06}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline --no-graph"
PMCS=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE")
HX=f32_hf16x3_bwd_bf16x3
MODES="$HX"; [ -n "$ALL" ] && MODES="$HX f32_bwd_bf16x3 f32 f32_bf16x3 bf16"
for dt in $MODES; do
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_$dt -o kt -- $B --dtype $dt --steps 10 --warmup 3 > $R/$OUT/kt_$dt.log 2>&1
done
dt=$HX
for c in "${PMCS[@]}"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_${dt}_$n -o pmc -- $B --dtype $dt --steps 2 --warmup 1 > $R/$OUT/pmc_${dt}_$n.log 2>&1
done
python $R/tools/pmc_summary.py $R/$OUT/pmc_$dt.json $dt $R/$OUT/pmc_${dt}_FETCH_SIZE $R/$OUT/pmc_${dt}_WRITE_SIZE $R/$OUT/pmc_${dt}_SQ_VALU_MFMA_BUSY_CYCLES > $R/$OUT/pmc_${dt}_summary.log 2>&1
rm -rf $R/$OUT/pmc_${dt}_FETCH_SIZE $R/$OUT/pmc_${dt}_WRITE_SIZE $R/$OUT/pmc_${dt}_SQ_VALU_MFMA_BUSY_CYCLES
python $R/tools/step_trace.py $(find $R/$OUT/kt_$HX -name "*kernel_trace.csv" | head -1) > $R/$OUT/step_trace_$HX.txt 2>&1
if [ -n "$INFER" ]; then
for net in d0 d4; do
  if [ $net = d0 ]; then I="python $R/tools/infer_bench.py --dtype f32_hf16x3 --no-graph"; else I="python $R/tools/infer_bench.py --network efficientdet-d4 --batch 8 --size 1024 --dtype f32_hf16x3 --no-graph"; fi
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_infer_$net -o kt -- $I --reps 10 > $R/$OUT/kt_infer_$net.log 2>&1
  for c in "${PMCS[@]}"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout 150 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_infer_${net}_$n -o pmc -- $I --reps 2 > $R/$OUT/pmc_infer_${net}_$n.log 2>&1
  done
  python $R/tools/pmc_summary.py $R/$OUT/pmc_infer_${net}_$HX.json $HX $R/$OUT/pmc_infer_${net}_FETCH_SIZE $R/$OUT/pmc_infer_${net}_WRITE_SIZE $R/$OUT/pmc_infer_${net}_SQ_VALU_MFMA_BUSY_CYCLES "python tools/infer_bench.py ($net, f16x3 head + exact-fp32 trunk, eager: forward + detect + post-processing passes)" > $R/$OUT/pmc_infer_${net}_summary.log 2>&1
  rm -rf $R/$OUT/pmc_infer_${net}_FETCH_SIZE $R/$OUT/pmc_infer_${net}_WRITE_SIZE $R/$OUT/pmc_infer_${net}_SQ_VALU_MFMA_BUSY_CYCLES
done
fi
find $R/$OUT -name "*kernel_trace.csv" -delete
cd $R; du -sh $OUT; cat $OUT/pmc_*_summary.log | head -30; head -14 $OUT/step_trace_$HX.txt
