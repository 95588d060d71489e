mkdir -p gpurun_out/r3C
timeout 900 python -m pytest tests/test_gpu_backbone_ops.py tests/test_gpu_model.py tests/test_gpu_determinism.py -q > gpurun_out/r3C/pytest.log 2>&1; tail -3 gpurun_out/r3C/pytest.log
bash tools/ab_env.sh EFFDET_HIP_LIB tools/ab/lib_base.so efficientdet/pytorch_amd/libeffdet_hip.so tools/ab/lib_base.so efficientdet/pytorch_amd/libeffdet_hip.so
