#!/bin/bash
# Round-end GPU visit: the whole -m gpu suite, smoke(), the default bench line, then the profile passes (tools/profile_round.sh).
OUT=gpurun_out/${1:-final}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "tests rc=$?" | tee $OUT/rc.txt; tail -4 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/rc.txt
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.txt; tail -c 2500 $OUT/bench.log
bash tools/profile_round.sh $OUT/prof > $OUT/prof.log 2>&1; echo "prof rc=$?" | tee -a $OUT/rc.txt
