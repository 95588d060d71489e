mkdir -p gpurun_out/r3r
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_determinism.py tests/test_gpu_split.py -q -x > gpurun_out/r3r/pytest.log 2>&1; tail -3 gpurun_out/r3r/pytest.log
bash tools/ab_env.sh EFFDET_UNPACK_BATCH
timeout 300 python tools/copy_trace.py > gpurun_out/r3r/copy_trace.txt 2>&1; tail -5 gpurun_out/r3r/copy_trace.txt
