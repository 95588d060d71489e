#!/bin/bash
# One GPU-box visit of round 5: tools/r5_call.sh TAG step [step ...]; every step has its own timeout, output under gpurun_out/TAG/.
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
B1="python bench.py --no-cpu-baseline --no-extra-modes --no-d4 --no-inference"
for s in "$@"; do
  t0=$(date +%s)
  case $s in
    tests)    timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "tests rc=$?" | tee -a $OUT/rc.txt; tail -40 $OUT/pytest.log ;;
    mixed)    rm -f gpurun_out/parity_errors.txt; timeout 900 python -m pytest tests/test_gpu_conv.py::test_conv_second_output_in_the_split_layout tests/test_gpu_model.py -q -s -k "second_output or fwd_exact or (train_losses and f32_bwd_bf16x3)" > $OUT/mixed.log 2>&1; echo "mixed rc=$?" | tee -a $OUT/rc.txt; grep -v "^$" $OUT/mixed.log | tail -40; cp gpurun_out/parity_errors.txt $OUT/parity_mixed.txt 2>/dev/null ;;
    parity)   rm -f gpurun_out/parity_errors.txt; timeout 1200 python -m pytest tests/test_gpu_model.py -q -s -k "eval_forward or train_losses or non_square or fwd_exact" > $OUT/parity.log 2>&1; echo "parity rc=$?" | tee -a $OUT/rc.txt; tail -5 $OUT/parity.log; cp gpurun_out/parity_errors.txt $OUT/ 2>/dev/null ;;
    ddp1)     timeout 600 python bench.py --ddp-single --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline > $OUT/bench_ddp1.log 2> $OUT/bench_ddp1.err; echo "ddp1 rc=$?" | tee -a $OUT/rc.txt; tail -c 2500 $OUT/bench_ddp1.log; tail -8 $OUT/bench_ddp1.err ;;
    ddp2gloo) EFFDET_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --batch 8 --steps 5 --warmup 2 --no-roofline > $OUT/bench_ddp2gloo.log 2> $OUT/bench_ddp2gloo.err; echo "ddp2gloo rc=$?" | tee -a $OUT/rc.txt; tail -c 2000 $OUT/bench_ddp2gloo.log; tail -8 $OUT/bench_ddp2gloo.err ;;
    bench)    timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.txt; tail -c 3000 $OUT/bench.log; tail -3 $OUT/bench.err ;;
    bench1)   timeout 300 $B1 > $OUT/bench1.log 2> $OUT/bench1.err; echo "bench1 rc=$?" | tee -a $OUT/rc.txt; tail -c 4000 $OUT/bench1.log; tail -3 $OUT/bench1.err ;;
    *)        if [ -f "tools/r5_steps/$s.sh" ]; then OUT=$OUT bash tools/r5_steps/$s.sh; else echo "unknown step $s"; fi ;;
  esac
  echo "[$s: $(( $(date +%s) - t0 )) s]" | tee -a $OUT/rc.txt
done
find $OUT -name "*kernel_trace.csv" -size +40M -delete 2>/dev/null
