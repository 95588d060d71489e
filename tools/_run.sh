mkdir -p gpurun_out/r3y
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_determinism.py tests/test_gpu_backbone_ops.py -q -x > gpurun_out/r3y/pytest.log 2>&1; tail -3 gpurun_out/r3y/pytest.log
bash tools/ab_env.sh EFFDET_WGRAD_THIN 0 1 0 1
