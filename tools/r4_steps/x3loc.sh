timeout 600 python tools/x3_locate.py ${X3CASES:-d2_128_train d3_128_train} > $OUT/x3_locate.txt 2>&1; echo "x3loc rc=$?" | tee -a $OUT/rc.txt; cut -c1-200 $OUT/x3_locate.txt | tail -120
