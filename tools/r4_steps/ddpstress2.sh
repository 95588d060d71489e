# the scenario that aborted once: a 2-rank torchrun job first, the captured world_size-1 DDP bench right behind it -- three times over
for i in 1 2 3; do
  EFFDET_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 2 --steps 5 --warmup 2 --batch 8 --no-cpu-baseline --no-roofline > $OUT/n2_$i.log 2> $OUT/n2_$i.err; echo "n2 gloo $i rc=$?" | tee -a $OUT/rc.txt
  timeout 300 python bench.py --ddp-single --no-cpu-baseline --no-extra-modes --no-d4 --no-inference > $OUT/ddp2_$i.log 2> $OUT/ddp2_$i.err
  echo "ddp-single behind it $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $OUT/ddp2_$i.log | head -1) watchdog-errors=$(grep -c 'watchdog thread terminated with exception' $OUT/ddp2_$i.err)" | tee -a $OUT/rc.txt
done
