#!/bin/bash
# ordered launch list of one headline train step (rocprofv3 kernel trace of eager launches)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/kt_list; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python $R/bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline --no-graph --steps 4 --warmup 2 > $OUT/run.log 2>&1
python $R/tools/step_trace.py $(find $OUT -name "*kernel_trace.csv" | head -1) --list 0 500 > $R/gpurun_out/step_list.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
