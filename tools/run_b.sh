mkdir -p gpurun_out/r02b
timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "big" -x > gpurun_out/r02b/big_tests.log 2>&1; echo "bigtests rc=$?"; tail -15 gpurun_out/r02b/big_tests.log
timeout 300 python tools/kbench2.py > gpurun_out/r02b/kbench2.log 2>&1; echo "kbench2 rc=$?"; cat gpurun_out/r02b/kbench2.log
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model.py -q -k "packed or consumer or differentiable or d1_128_train" > gpurun_out/r02b/fixed_tests.log 2>&1; echo "fixed rc=$?"; tail -5 gpurun_out/r02b/fixed_tests.log
bash tools/gpu_round.sh r02b stats pmc
