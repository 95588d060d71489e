timeout 300 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -2
bash tools/ab_env.sh EFFDET_WGRAD_THIN 2 1 2 1
