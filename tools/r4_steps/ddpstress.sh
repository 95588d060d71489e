# the captured DDP step over a world_size-1 RCCL group, N times in a row (roofline leg included): every run must end rc 0
for i in ${STRESS_RUNS:-1 2 3 4 5 6 7 8}; do
  timeout 300 python bench.py --ddp-single --no-cpu-baseline --no-extra-modes --no-d4 --no-inference > $OUT/ddp_stress$i.log 2> $OUT/ddp_stress$i.err
  echo "ddp-single run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $OUT/ddp_stress$i.log | head -1) $(grep -o '"captured": [a-z]*' $OUT/ddp_stress$i.log | head -1)" | tee -a $OUT/rc.txt
done
