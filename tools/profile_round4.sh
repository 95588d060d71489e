#!/bin/bash
# Round-4 evidence in ONE GPU call: rocprofv3 kernel-trace stats of the train step in the three arithmetic modes and of the D0 / D4
# inference legs, the three PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE) of EVERY mode, the step
# trace of the headline mode.  Every rocprofv3 run is `timeout`-wrapped and asks for csv (the rocpd writer has hung after finalisation).
OUT=${1:-gpurun_out/prof_r04}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline --no-graph"
for dt in f32_bf16x3 bf16 f32; do
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_$dt -o kt -- $B --dtype $dt --steps 10 --warmup 3 > $R/$OUT/kt_$dt.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout 150 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_${dt}_$n -o pmc -- $B --dtype $dt --steps 2 --warmup 1 > $R/$OUT/pmc_${dt}_$n.log 2>&1
  done
  python $R/tools/pmc_summary.py $R/$OUT/pmc_$dt.json $dt $R/$OUT/pmc_${dt}_FETCH_SIZE $R/$OUT/pmc_${dt}_WRITE_SIZE $R/$OUT/pmc_${dt}_SQ_VALU_MFMA_BUSY_CYCLES > $R/$OUT/pmc_${dt}_summary.log 2>&1
  rm -rf $R/$OUT/pmc_${dt}_FETCH_SIZE $R/$OUT/pmc_${dt}_WRITE_SIZE $R/$OUT/pmc_${dt}_SQ_VALU_MFMA_BUSY_CYCLES
done
python $R/tools/step_trace.py $(find $R/$OUT/kt_f32_bf16x3 -name "*kernel_trace.csv" | head -1) > $R/$OUT/step_trace_bf16x3.txt 2>&1
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_infer_d0 -o kt -- python $R/tools/infer_bench.py --no-graph --reps 10 > $R/$OUT/kt_infer_d0.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_infer_d4 -o kt -- python $R/tools/infer_bench.py --network efficientdet-d4 --batch 8 --size 1024 --no-graph --reps 10 > $R/$OUT/kt_infer_d4.log 2>&1
find $R/$OUT -name "*kernel_trace.csv" -delete
cd $R; du -sh $OUT; cat $OUT/pmc_*_summary.log | head -20; head -12 $OUT/step_trace_bf16x3.txt
