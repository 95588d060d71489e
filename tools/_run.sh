mkdir -p gpurun_out/r3v
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bifpn_ops.py tests/test_gpu_model.py tests/test_gpu_determinism.py -q -x > gpurun_out/r3v/pytest.log 2>&1; tail -3 gpurun_out/r3v/pytest.log
bash tools/ab_env.sh EFFDET_BIFPN_WGRAD_GROUP 0 1 0 1
