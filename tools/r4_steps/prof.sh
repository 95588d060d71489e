bash tools/profile_round4.sh $OUT/prof > $OUT/prof.log 2>&1; echo "prof rc=$?" | tee -a $OUT/rc.txt; tail -25 $OUT/prof.log
