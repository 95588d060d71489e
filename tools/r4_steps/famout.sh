timeout 600 python tools/grad_outliers.py ${FAMCASES:-d2_128_train d3_128_train d5_128_train d6_128_train} > $OUT/grad_outliers_families.log 2>&1; echo "famout rc=$?" | tee -a $OUT/rc.txt
cp gpurun_out/grad_outliers.txt $OUT/grad_outliers_families.txt 2>/dev/null; cat $OUT/grad_outliers_families.txt | cut -c1-200
