timeout 600 python -m pytest tests/test_gpu_rccl.py -q > $OUT/rccl.log 2>&1; rc=$?; echo "rccl rc=$rc" | tee -a $OUT/rc.txt; tail -12 $OUT/rccl.log
cp gpurun_out/rccl_world1.json $OUT/ 2>/dev/null
if [ $rc -eq 0 ]; then
  timeout 600 python bench.py --ddp-single --no-cpu-baseline --no-extra-modes --no-d4 --no-inference --no-roofline > $OUT/bench_ddp1.log 2> $OUT/bench_ddp1.err; echo "ddp1 rc=$?" | tee -a $OUT/rc.txt; tail -c 1200 $OUT/bench_ddp1.log; tail -3 $OUT/bench_ddp1.err | cut -c1-300
fi
