# real failure path of the supervisor: 2 RCCL ranks on ONE GPU (RCCL refuses: "Duplicate GPU detected") -> both attempts must fail FAST, rc != 0, no hang
t0=$(date +%s)
EFFDET_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --batch 4 --steps 3 --warmup 1 --no-roofline > $OUT/failpath.log 2> $OUT/failpath.err; echo "failpath rc=$? in $(( $(date +%s) - t0 )) s" | tee -a $OUT/rc.txt
grep "bench\]\|bench rank\|every attempt" $OUT/failpath.err | cut -c1-200 | head -8; cat $OUT/failpath.log | head -3
