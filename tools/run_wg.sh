# bf16 weight-gradient kernel on the head shapes (tools/kbench.py) + its numerics tests
for c in "256 256" "64 256" "256 720"; do set -- $c; echo -n "wgrad bf16 $1->$2: "; python tools/kbench.py --dtype bf16 --which wgrad --cin $1 --cout $2 --reps 20 2>/dev/null | tail -1; done
python -m pytest tests/test_gpu_conv.py -q -k "wgrad" 2>&1 | tail -1
