mkdir -p gpurun_out/r3D
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_determinism.py tests/test_gpu_pipeline.py -q -x > gpurun_out/r3D/pytest.log 2>&1; tail -3 gpurun_out/r3D/pytest.log
bash tools/ab_env.sh EFFDET_STEM_LINK 0 1 0 1
