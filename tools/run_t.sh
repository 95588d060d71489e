python -m pytest tests/test_gpu_pipeline.py -q -x -k "graphed" > gpurun_out/graph_test.log 2>&1; echo rc=$?
grep -v Warning gpurun_out/graph_test.log | grep -n "Error\|error\|Fatal\|fault\|assert\|rc=" | head -20
head -60 gpurun_out/graph_test.log | cut -c1-300
