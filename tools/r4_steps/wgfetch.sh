R=$GRAFT_REPO_ROOT
for n in 0 14 32 56 112; do
  lab="splits=$n"; [ $n = 0 ] && lab="splits=default(28)"
  EFFDET_WGRAD_FORCE_SPLITS=$n timeout 200 python tools/kbench_split.py --which wgrad --shapes 256:256 --reps 10 2>/dev/null | grep "split wgrad" | sed "s/^/$lab  /" | tee -a $OUT/wgfetch.txt
  (cd /tmp && EFFDET_WGRAD_FORCE_SPLITS=$n timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_wg_$n -o p -- python $R/tools/kbench_split.py --which wgrad --shapes 256:256 --reps 3 > /dev/null 2>&1)
  python tools/pmc_fetch.py $OUT/pmc_wg_$n conv_wgrad_split_kernel "$lab" | tee -a $OUT/wgfetch.txt
  rm -rf $OUT/pmc_wg_$n
done
