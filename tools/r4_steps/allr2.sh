bash tools/ab_env.sh EFFDET_WGRAD_SPLIT_ALLR 0 -1 0 -1 2>&1 | tee $OUT/ab_allr.txt
