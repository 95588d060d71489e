#!/bin/bash
# One GPU-box visit of round 4: tools/r4_call.sh TAG step [step ...]; every step has its own timeout, output under gpurun_out/TAG/.
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
B1="python bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference"
for s in "$@"; do
  t0=$(date +%s)
  case $s in
    tests)    timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "tests rc=$?" | tee -a $OUT/rc.txt; tail -5 $OUT/pytest.log ;;
    newtests) timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_conv.py::test_integration_md_stub_runs -q -s > $OUT/newtests.log 2>&1; echo "newtests rc=$?" | tee -a $OUT/rc.txt; tail -15 $OUT/newtests.log ;;
    parity)   rm -f gpurun_out/parity_errors.txt; timeout 900 python -m pytest tests/test_gpu_model.py -q -s -k "eval_forward or train_losses or non_square" > $OUT/parity.log 2>&1; echo "parity rc=$?" | tee -a $OUT/rc.txt; tail -5 $OUT/parity.log; cp gpurun_out/parity_errors.txt $OUT/ 2>/dev/null ;;
    outliers) timeout 600 python tools/grad_outliers.py > $OUT/grad_outliers.log 2>&1; echo "outliers rc=$?" | tee -a $OUT/rc.txt; cp gpurun_out/grad_outliers.txt $OUT/ 2>/dev/null; tail -30 $OUT/grad_outliers.log ;;
    mall)     hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_bw tools/probe/mall_bw.hip > /dev/null 2>&1 && timeout 120 /tmp/mall_bw > $OUT/mall_bw.txt 2>&1; cat $OUT/mall_bw.txt ;;
    mfma)     hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/probe/mfma_peak.hip > /dev/null 2>&1 && timeout 120 /tmp/mfma_peak > $OUT/mfma_peak.txt 2>&1; cat $OUT/mfma_peak.txt ;;
    chunk)    timeout 300 python tools/chunk_probe.py 4 > $OUT/chunk_probe.txt 2>&1; tail -8 $OUT/chunk_probe.txt ;;
    ddp1)     timeout 600 python bench.py --ddp-single --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline > $OUT/bench_ddp1.log 2> $OUT/bench_ddp1.err; echo "ddp1 rc=$?" | tee -a $OUT/rc.txt; tail -c 1500 $OUT/bench_ddp1.log; tail -5 $OUT/bench_ddp1.err ;;
    bench)    timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.txt; tail -c 3000 $OUT/bench.log; tail -3 $OUT/bench.err ;;
    bench1)   timeout 300 $B1 > $OUT/bench1.log 2> $OUT/bench1.err; echo "bench1 rc=$?" | tee -a $OUT/rc.txt; tail -c 1200 $OUT/bench1.log ;;
    hostc)    timeout 400 python tools/host_contention.py > $OUT/host_contention.txt 2>&1; tail -8 $OUT/host_contention.txt ;;
    *)        if [ -f "tools/r4_steps/$s.sh" ]; then OUT=$OUT bash tools/r4_steps/$s.sh; else echo "unknown step $s"; fi ;;
  esac
  echo "[$s: $(( $(date +%s) - t0 )) s]" | tee -a $OUT/rc.txt
done
find $OUT -name "*kernel_trace.csv" -size +40M -delete 2>/dev/null
