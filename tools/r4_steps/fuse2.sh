for v in 0 1; do EFFDET_FUSE_EXPAND_DW=$v timeout 300 python tools/infer_bench.py --reps 20 > $OUT/infer_fuse$v.log 2>&1; tail -1 $OUT/infer_fuse$v.log; done
for c in "16,24" "16,24,32" ; do EFFDET_FUSE_CIN=$c timeout 300 python tools/infer_bench.py --network efficientdet-d4 --batch 8 --size 1024 --reps 10 > $OUT/infer_d4_fuse.log 2>&1; echo "D4 FUSE_CIN=$c"; tail -1 $OUT/infer_d4_fuse.log; done
EFFDET_FUSE_EXPAND_DW=0 timeout 300 python tools/infer_bench.py --network efficientdet-d4 --batch 8 --size 1024 --reps 10 > $OUT/infer_d4_fuse0.log 2>&1; tail -1 $OUT/infer_d4_fuse0.log
