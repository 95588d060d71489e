EFFDET_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --batch 8 --no-cpu-baseline --no-roofline > $OUT/bench_n2_gloo.log 2> $OUT/bench_n2_gloo.err; echo "n2 gloo rc=$?" | tee -a $OUT/rc.txt; tail -c 900 $OUT/bench_n2_gloo.log; grep -i "error\|Traceback" $OUT/bench_n2_gloo.err | head -5
timeout 600 python bench.py --ddp-single --no-cpu-baseline --no-extra-modes --no-d4 --no-inference > $OUT/bench_ddp1.log 2> $OUT/bench_ddp1.err; echo "ddp1 rc=$?" | tee -a $OUT/rc.txt; python - <<PY
import json
l=[x for x in open('$OUT/bench_ddp1.log') if x.startswith('{')][-1]; d=json.loads(l)
print(d['value'], d['ms_per_step'], d['host_ms_per_step'], d['config']['ddp_graph'], d['roofline']['frac'], d['roofline']['traffic_source'])
PY
