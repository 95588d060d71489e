mkdir -p gpurun_out/r3p
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "pw or pointwise or skinny" > gpurun_out/r3p/pytest.log 2>&1; tail -3 gpurun_out/r3p/pytest.log
for z in 0 1 0 1; do
EFFDET_CONV_PW=$z timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-d4 --no-inference 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); ks = d.get('kernels', d.get('roofline', {}))
        print('conv_pw', $z, d['value'], d['ms_per_step'])
"
done
