#!/bin/bash
# One GPU call: rocprofv3 kernel-trace stats of the train step in the three arithmetic modes + the three PMC passes of the
# headline mode.  Every rocprofv3 run is wrapped in `timeout`: the profiled python process has been seen to hang at exit AFTER
# rocprofv3 finalised its output (the files are complete by then).
OUT=${1:-gpurun_out/prof_r03}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import torch" 2>/dev/null
B="python bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline --no-graph"
for dt in f32_bf16x3 bf16 f32; do
  timeout 110 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$dt -o kt -- $B --dtype $dt --steps 10 --warmup 3 > $OUT/kt_$dt.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 110 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o pmc -- $B --dtype f32_bf16x3 --steps 2 --warmup 1 > $OUT/pmc_$n.log 2>&1
done
find $OUT -name "*.csv" | head -30
du -sh $OUT
