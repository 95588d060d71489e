mkdir -p gpurun_out/r3u
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_backbone_ops.py tests/test_gpu_model.py tests/test_gpu_determinism.py -q -x > gpurun_out/r3u/pytest.log 2>&1; tail -3 gpurun_out/r3u/pytest.log
bash tools/ab_env.sh EFFDET_HIP_LIB tools/ab/lib_base.so efficientdet/pytorch_amd/libeffdet_hip.so
bash tools/ab_env.sh EFFDET_UNPACK_BATCH 0 1 0 1
