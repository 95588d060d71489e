timeout 300 python tools/ns_state_probe.py > $OUT/ns_state_probe.txt 2>&1; tail -6 $OUT/ns_state_probe.txt
timeout 300 python -m pytest tests/test_gpu_model.py -q -s -k "non_square and 384" > $OUT/ns384_alone.log 2>&1; grep "worst gradient" $OUT/ns384_alone.log
