timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/rc.txt; tail -1 $OUT/smoke.log
