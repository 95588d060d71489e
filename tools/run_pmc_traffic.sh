# HBM traffic of ONE head-tower conv launch (tools/kbench2.py shape 0) per arithmetic: usage tools/run_pmc_traffic.sh TAG
TAG=${1:-traffic}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for mode in "" "--f32 f32" "--f32 bf16x3"; do
  n=$(echo "m$mode" | tr -d ' -')
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/${n}_$c -o run -- python /root/repo/tools/kbench2.py --reps 3 --variants 0 --only 0 $mode > $OUT/${n}_$c.log 2>&1
  done
  echo "== mode '$mode' (FETCH_SIZE is in KiB and counts 128-B requests as 64 B on gfx950: x2)"
  python /root/repo/tools/pmc_kernel.py $OUT/${n}_FETCH_SIZE $OUT/${n}_WRITE_SIZE --match conv_igemm | grep -E "launches|FETCH_SIZE|WRITE_SIZE"
  rm -rf $OUT/${n}_FETCH_SIZE $OUT/${n}_WRITE_SIZE
done
